"""VALU issue rate of the two trace kernels from PMC counters — the bound that actually limits this path
(SURVEY.md 8(d) predicted a VALU/latency-bound kernel; roofline.frac against HBM only says how far from the
memory roof it is).  One rocprofv3 --pmc pass (SQ counters only, --kernel-trace) of bench.py:

  issue_frac = SQ_INSTS_VALU / (kernel seconds x 1024 SIMDs x f_clk / 4)
               a wave64 VALU instruction occupies its 16-lane SIMD for 4 cycles; 256 CUs x 4 SIMDs; f_clk 2.4 GHz
  lane_util  = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)     share of lanes doing work in an issued instruction

Run on the GPU box:  python scripts/collect_valu.py profiles/round1/valu_config2.json [bench args]
bench.py reports the file's numbers as roofline.valu (it cannot profile itself)."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

out_json = sys.argv[1]
bench_args = sys.argv[2:] or ["--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("TMPDIR", "/tmp")
KERNELS = ("k_intersect", "k_shade")
COUNTERS = ["SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"]
N_SIMD, F_CLK = 256 * 4, 2.4e9
d = tempfile.mkdtemp(prefix="pmc_valu_", dir=os.path.join(root, "gpurun_out"))
cmd = ["rocprofv3", "--pmc"] + COUNTERS + ["--kernel-trace", "--output-format", "csv", "-d", d, "--",
                                           sys.executable, os.path.join(root, "bench.py")] + bench_args
r = subprocess.run(cmd, capture_output=True, text=True, cwd=root)
bench_line = None
for line in r.stdout.splitlines():
    if line.startswith("{") and '"metric"' in line:
        bench_line = json.loads(line)
tot = collections.defaultdict(float)
secs = collections.defaultdict(float)
launches = collections.defaultdict(int)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = next((k for k in KERNELS if k in row["Kernel_Name"]), None)
        if k is not None:
            tot[(k, row["Counter_Name"])] += float(row["Counter_Value"])
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = next((k for k in KERNELS if k in row["Kernel_Name"]), None)
        if k is not None:
            secs[k] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9
            launches[k] += 1
res = {"command": "bench.py " + " ".join(bench_args), "n_simd": N_SIMD, "f_clk_hz": F_CLK, "kernels": {},
       "note": "issue_frac = SQ_INSTS_VALU / (seconds * n_simd * f_clk / 4); lane_util = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU); "
               "kernel seconds from the --kernel-trace timestamps of the same (counter-collecting, hence slightly slower) run; "
               "issue_frac above 1 marks the limit of the 4-cycles-per-instruction model (the counter includes instructions issued "
               "under an empty EXEC mask in divergent code): read it as saturated"}
for k in KERNELS:
    insts = tot[(k, "SQ_INSTS_VALU")]
    res["kernels"][k] = {
        "launches": launches[k], "seconds": secs[k], "insts_valu": insts,
        "issue_frac": insts / max(secs[k] * N_SIMD * F_CLK / 4.0, 1e-30),
        "lane_util": tot[(k, "SQ_THREAD_CYCLES_VALU")] / max(64.0 * tot[(k, "SQ_ACTIVE_INST_VALU")], 1e-30),
        "wait_frac_of_wave_cycles": tot[(k, "SQ_WAIT_ANY")] / max(tot[(k, "SQ_WAVE_CYCLES")], 1e-30),
    }
if bench_line:
    res["bench_config"] = bench_line.get("config")
os.makedirs(os.path.dirname(os.path.abspath(out_json)), exist_ok=True)
json.dump(res, open(out_json, "w"), indent=1)
print(json.dumps(res))
