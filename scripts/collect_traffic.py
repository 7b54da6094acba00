"""HBM traffic of the trace step from PMC counters, as MI355X_MICROARCH.md §HBM prescribes:
FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (TCC has 4 slots; they do not fit
one pass) with --kernel-trace only; units are KiB... rocprofv3 reports KB (x1024 here); on gfx950
FETCH_SIZE reads exactly 1/2 of a wide coalesced (16 B/lane) streaming read, so it is doubled;
WRITE_SIZE is exact for 16 B/lane streaming stores.  The x2 was calibrated in round 3 for the access patterns of the trace
kernels (coalesced 4-32 B, gathers of 4-32 B out of scattered lines): it is exact for all of them — the L2 fetches whole
128 B lines with one request each (profiles/round3/fetch_calibration.json).

Run on the GPU box:  python scripts/collect_traffic.py profiles/round1/traffic.json [bench args]
Writes the JSON that bench.py reports as roofline.traffic (per launch, like roofline.achieved).
"""
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
tot = collections.defaultdict(float)
launches = collections.defaultdict(int)
bench_line = None
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    d = tempfile.mkdtemp(prefix="pmc_", dir=os.path.join(root, "gpurun_out"))
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
           sys.executable, os.path.join(root, "bench.py")] + bench_args
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=root)
    for line in r.stdout.splitlines():
        if line.startswith("{") and '"metric"' in line:
            bench_line = json.loads(line)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        seen = set()
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            k = next((k for k in KERNELS if k in name), None)
            if k is None or row["Counter_Name"] != counter:
                continue
            tot[(k, counter)] += float(row["Counter_Value"])
            if counter == "FETCH_SIZE" and row["Dispatch_Id"] not in seen:
                seen.add(row["Dispatch_Id"])
                launches[k] += 1
# FETCH_SIZE x2: calibrated on this chip for streams AND gathers (scripts/micro/fetch_gather.hip,
# profiles/round3/fetch_calibration.json): every L2 miss is ONE request for the whole 128 B line, tallied at 64 B, whether the
# wave wanted 4, 16 or 2 x 16 B of it.  WRITE_SIZE: exact for coalesced stores; a scattered 8-32 B store to one line counts 32 B.
res = {"command": "bench.py " + " ".join(bench_args), "kernels": {}, "fetch_correction": 2.0,
       "fetch_correction_source": "profiles/round3/fetch_calibration.json (lines * 128 B / FETCH_SIZE = 2.000 for all ten read patterns)",
       "library_build_id": (bench_line or {}).get("library_build_id"),
       "note": "FETCH_SIZE x2 = 128 B lines fetched by the L2 (gfx950 tallies the 128 B request at 64 B), WRITE_SIZE x1; KB x 1024; "
               "separate --pmc passes"}
total_bytes = 0.0
for k in KERNELS:
    fb = tot[(k, "FETCH_SIZE")] * 1024.0 * 2.0
    wb = tot[(k, "WRITE_SIZE")] * 1024.0
    res["kernels"][k] = {"launches": launches[k], "fetch_bytes": fb, "write_bytes": wb}
    total_bytes += fb + wb
# a "launch" is the logical one of bench.py / RtStats: one kernel, one depth, all shards (issued as two concurrent
# half-grid launches on two streams, so rocprof sees twice as many dispatches)
n_logical = (bench_line or {}).get("roofline", {}).get("launches") or max(launches["k_shade"], 1)
res["trace_step_bytes_total"] = total_bytes
res["trace_step_dispatches"] = launches["k_shade"] + launches["k_intersect"]
res["trace_step_launches"] = n_logical
res["trace_step_bytes_per_launch"] = total_bytes / n_logical
if bench_line:
    res["bench_config"] = bench_line.get("config")
    res["algorithmic_bytes_per_launch"] = bench_line["roofline"]["bytes_per_launch"]
    res["traffic_over_algorithmic"] = res["trace_step_bytes_per_launch"] / max(bench_line["roofline"]["bytes_per_launch"], 1)
os.makedirs(os.path.dirname(os.path.abspath(out_json)), exist_ok=True)
json.dump(res, open(out_json, "w"), indent=1)
print(json.dumps(res))
