"""VALU issue rate of the two trace kernels from PMC counters — the companion roofline of this path (roofline.frac
against HBM says how far from the memory roof it is; this says how far from the vector-issue roof).  One rocprofv3 --pmc
pass (SQ counters only, --kernel-trace) of bench.py:

  issue_frac = SQ_INSTS_VALU / (kernel seconds x peak wave-instructions per second)
               peak = the MEASURED v_mul_f32 / v_add_f32 rate of scripts/micro/mul_rate.hip at 8 waves per SIMD
               (profiles/round2/valu_peak.json: 71.5 T lane-ops/s = 2.14 cycles per wave64 instruction on the 1024
               SIMD-32s at the 2.34 GHz the chip holds under that load; MI355X_MICROARCH.md: 2 cycles, 2.4 GHz = 78.6 T).
               Round 1 divided by a 4-cycle model and read 1.13 "saturated"; against the measured peak the same
               counters are 0.57 / 0.38.  Integer multiplies, compares+selects and rcp/sqrt issue at half / half / quarter
               that rate (same file), so a kernel made of them saturates below 1.
  lane_util  = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)     share of lanes doing work in an issued instruction

Run on the GPU box:  python scripts/collect_valu.py profiles/round2/valu_config2.json [bench args]
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
bench_args = sys.argv[2:] or ["--steps", "1", "--warmup", "0", "--timed-only"]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("TMPDIR", "/tmp")
KERNELS = ("k_intersect", "k_shade")
COUNTERS = ["SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"]
peak_file = os.path.join(root, "profiles", "round2", "valu_peak.json")
PEAK_LANE_OPS = json.load(open(peak_file))["rates"]["v_mul_f32/v_add_f32"]["rate"][3] * 1e12  # measured, 8 waves per SIMD
PEAK_WAVE_INSTS = PEAK_LANE_OPS / 64.0
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
res = {"command": "bench.py " + " ".join(bench_args), "peak_wave_insts_per_s": PEAK_WAVE_INSTS,
       "peak_source": os.path.relpath(peak_file, root), "kernels": {},
       "note": "issue_frac = SQ_INSTS_VALU / (seconds * peak_wave_insts_per_s), peak = measured v_mul_f32/v_add_f32 issue rate at 8 waves "
               "per SIMD (scripts/micro/mul_rate.hip); lane_util = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU); kernel seconds "
               "from the --kernel-trace timestamps of the same (counter-collecting, hence slightly slower) run"}
for k in KERNELS:
    insts = tot[(k, "SQ_INSTS_VALU")]
    res["kernels"][k] = {
        "launches": launches[k], "seconds": secs[k], "insts_valu": insts,
        "issue_frac": insts / max(secs[k] * PEAK_WAVE_INSTS, 1e-30),
        "lane_util": tot[(k, "SQ_THREAD_CYCLES_VALU")] / max(64.0 * tot[(k, "SQ_ACTIVE_INST_VALU")], 1e-30),
        "wait_frac_of_wave_cycles": tot[(k, "SQ_WAIT_ANY")] / max(tot[(k, "SQ_WAVE_CYCLES")], 1e-30),
    }
if bench_line:
    res["bench_config"] = bench_line.get("config")
    res["library_build_id"] = bench_line.get("library_build_id")
os.makedirs(os.path.dirname(os.path.abspath(out_json)), exist_ok=True)
json.dump(res, open(out_json, "w"), indent=1)
print(json.dumps(res))
