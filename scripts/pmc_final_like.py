"""SQ counters of the closest-hit kernel on a final_scene-like scene, for each tree placement (diagnostic):
    python scripts/pmc_final_like.py <outdir> <boxes_per_side> <cloud_spheres> [spp]
Runs scripts/gpu_final_like.py once per counter group under rocprofv3 (--pmc with --kernel-trace only) and prints, per kernel
name, dispatch time and the wave-cycle split: waiting (s_waitcnt), waiting to issue, executing vector instructions, lane
utilisation, LDS instructions per vector instruction and the LDS bank-conflict share.  gpu_final_like.py renders with the default
placement, with the float tree through L2, and with the default again; the kernel names tell the placements apart
(k_intersect<.., true> in its last template argument = the 58 B tree in LDS walked without a stack)."""
import collections
import csv
import glob
import os
import re
import subprocess
import sys

out, nb, nc = sys.argv[1], sys.argv[2], sys.argv[3]
spp = sys.argv[4] if len(sys.argv) > 4 else "32"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("TMPDIR", "/tmp")
GROUPS = {"sq": ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU",
                 "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_LDS"],
          "lds": ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_WAIT_INST_LDS"]}
os.makedirs(out, exist_ok=True)
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for g, ctrs in GROUPS.items():
    d = os.path.join(out, g)
    subprocess.run(["rm", "-rf", d])
    cmd = ["rocprofv3", "--pmc"] + ctrs + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
           os.path.join(root, "scripts", "gpu_final_like.py"), nb, nc, spp, "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=root)
    open(os.path.join(out, g + ".log"), "w").write(r.stdout + r.stderr)
    trace = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not trace or not cc:
        print("no output for group", g)
        continue
    name_of, us_of = {}, {}
    for row in csv.DictReader(open(trace[0])):
        name_of[row["Dispatch_Id"]] = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void rt::", "").replace("rt::", "")
        us_of[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
    seen = set()
    for row in csv.DictReader(open(cc[0])):
        n = name_of.get(row["Dispatch_Id"])
        if n is None:
            continue
        tot[n][row["Counter_Name"]] += float(row["Counter_Value"])
        if (g, row["Dispatch_Id"]) not in seen:
            seen.add((g, row["Dispatch_Id"]))
            tot[n]["us:" + g] += us_of[row["Dispatch_Id"]]
for n, v in sorted(tot.items(), key=lambda kv: -kv[1].get("us:sq", 0.0)):
    if not n.startswith("k_intersect"):
        continue
    wc = max(v["SQ_WAVE_CYCLES"], 1.0)
    print(f"{n}\n   dispatches {v['us:sq'] / 1e3:8.2f} ms   waiting {v['SQ_WAIT_ANY'] / wc:.2f}  waiting to issue {v['SQ_WAIT_INST_ANY'] / wc:.2f}  "
          f"vector {v['SQ_ACTIVE_INST_VALU'] / wc:.2f}  lanes {v['SQ_THREAD_CYCLES_VALU'] / max(64 * v['SQ_ACTIVE_INST_VALU'], 1):.2f}  "
          f"LDS active {v['SQ_ACTIVE_INST_LDS'] / wc:.2f}\n   vector instructions {v['SQ_INSTS_VALU']:.3e}  LDS instructions {v['SQ_INSTS_LDS']:.3e}  "
          f"loads {v['SQ_INSTS_VMEM_RD']:.3e}  scalar {v['SQ_INSTS_SALU']:.3e}  LDS bank-conflict cycles / LDS cycles {v['SQ_LDS_BANK_CONFLICT'] / max(v['SQ_LDS_IDX_ACTIVE'], 1):.2f}  "
          f"waiting on LDS issue {v['SQ_WAIT_INST_LDS'] / wc:.2f}")
