"""Per-dispatch HBM traffic and wave statistics of one render (diagnostic):
    python scripts/pmc_probe.py <outdir> [lib.so] [spp] [scene]
Runs scripts/gpu_depth_probe.py under rocprofv3 several times (FETCH_SIZE, WRITE_SIZE, SQ, LDS and instruction-cache groups in separate --pmc passes,
--kernel-trace only, as MI355X_MICROARCH.md prescribes) and prints, for the LAST render of the run, one line per kernel launch:
duration, FETCH_SIZE x2 (gfx950 wide-load correction) + WRITE_SIZE, the resulting GB/s, the wave-cycle split, and the miss
rate and request rate of the instruction cache."""
import collections
import csv
import glob
import os
import re
import subprocess
import sys

out = sys.argv[1]
lib = sys.argv[2] if len(sys.argv) > 2 else ""
spp = sys.argv[3] if len(sys.argv) > 3 else "128"
scene = sys.argv[4] if len(sys.argv) > 4 else "sphere_scene"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("TMPDIR", "/tmp")
if lib:
    os.environ["RTOW_GPU_LIB"] = os.path.abspath(lib)
GROUPS = {"fetch": ["FETCH_SIZE"], "write": ["WRITE_SIZE"],
          "sq": ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU",
                 "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_LDS"],
          "lds": ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU"],
          # instruction fetch: requests / misses of the instruction cache (64 KB shared by a pair of CUs; k_shade is ~45 KB of code)
          "icache": ["SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQC_ICACHE_MISSES_DUPLICATE", "SQ_IFETCH"]}
os.makedirs(out, exist_ok=True)


def short(n):
    m = re.match(r"void rt::(k_\w+)<([^>]*)>", n)
    return (m.group(1) + "<" + m.group(2).replace(" ", "") + ">") if m else n.split("(")[0].replace("rt::", "")


data = {}
for g, ctrs in GROUPS.items():
    d = os.path.join(out, g)
    subprocess.run(["rm", "-rf", d])
    cmd = ["rocprofv3", "--pmc"] + ctrs + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                                           os.path.join(root, "scripts", "gpu_depth_probe.py"), spp, "0", scene]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=root)
    open(os.path.join(out, g + ".log"), "w").write(r.stdout + r.stderr)
    trace = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not trace or not cc:
        print("no output for group", g, file=sys.stderr)
        continue
    rows = sorted(csv.DictReader(open(trace[0])), key=lambda r: int(r["Start_Timestamp"]))
    disp = [(r["Dispatch_Id"], short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
    last = max(i for i, x in enumerate(disp) if x[1] == "k_init_counts")
    disp = disp[last:]
    vals = collections.defaultdict(dict)
    for r in csv.DictReader(open(cc[0])):
        vals[r["Dispatch_Id"]][r["Counter_Name"]] = vals[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    data[g] = [(name, us, vals.get(did, {})) for did, name, us in disp]

n = min(len(v) for v in data.values())
print("%-3s %-34s %9s %9s %9s %7s | %6s %6s %6s %6s %6s %6s | %7s %9s" % ("#", "kernel", "us", "fetchMB", "writeMB", "TB/s", "wait", "winst", "valu", "lane", "lds", "ldscf", "icmiss", "icreq/us"))
tot = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0])
for i in range(n):
    name, us, _ = data["fetch"][i]
    f = data["fetch"][i][2].get("FETCH_SIZE", 0.0) * 1024 * 2
    w = data["write"][i][2].get("WRITE_SIZE", 0.0) * 1024
    sq = data["sq"][i][2]
    ld = data["lds"][i][2]
    ic = data["icache"][i][2] if "icache" in data else {}
    wc = max(sq.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    t = tot[name]
    t[0] += us; t[1] += f; t[2] += w; t[3] += 1
    if i < 40:
        print("%-3d %-34s %9.1f %9.1f %9.1f %7.2f | %6.2f %6.2f %6.2f %6.2f %6.2f %6.2f | %7.4f %9.0f" % (
            i, name[:34], us, f / 1e6, w / 1e6, (f + w) / us / 1e6, sq.get("SQ_WAIT_ANY", 0) / wc, sq.get("SQ_WAIT_INST_ANY", 0) / wc,
            sq.get("SQ_ACTIVE_INST_VALU", 0) / wc, sq.get("SQ_THREAD_CYCLES_VALU", 0) / max(64 * sq.get("SQ_ACTIVE_INST_VALU", 0), 1),
            sq.get("SQ_ACTIVE_INST_LDS", 0) / wc, ld.get("SQ_LDS_BANK_CONFLICT", 0) / max(ld.get("SQ_LDS_IDX_ACTIVE", 0), 1),
            ic.get("SQC_ICACHE_MISSES", 0) / max(ic.get("SQC_ICACHE_REQ", 0), 1), ic.get("SQC_ICACHE_REQ", 0) / max(us, 1e-9)))
print()
for name, (us, f, w, c) in sorted(tot.items(), key=lambda x: -x[1][0]):
    print("%-34s calls %3d  %9.2f ms  fetch %8.2f GB  write %8.2f GB  %6.2f TB/s" % (name[:34], c, us / 1e3, f / 1e9, w / 1e9, (f + w) / max(us, 1e-9) / 1e6))
