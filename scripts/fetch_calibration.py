"""Calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on this gfx950 against known byte counts, per access pattern
(scripts/micro/fetch_gather.hip; MI355X_MICROARCH.md §HBM gives the x2 only for wide coalesced streaming reads):
    python scripts/fetch_calibration.py profiles/round3/fetch_calibration.json
Separate --pmc passes with --kernel-trace only (TCC has 4 counter slots).  For every variant: requested bytes, touched
128 B lines (= every line of the 4 GiB buffer, once), what FETCH_SIZE / WRITE_SIZE report, the raw request counters behind
them, the duration, and the factors  lines*128 / FETCH_SIZE  and  requested / FETCH_SIZE."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

out_json = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(root, "scripts", "micro", "fetch_gather")
os.environ.setdefault("TMPDIR", "/tmp")
REPS = 2
PASSES = {"fetch": ["FETCH_SIZE"], "write": ["WRITE_SIZE"],
          "rdreq": ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_BUBBLE_sum"],
          "wrreq": ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"],
          "l2": ["TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"]}
variants = None
per = collections.defaultdict(dict)
for tag, ctrs in PASSES.items():
    d = os.path.join(root, "gpurun_out", "fetchcal_" + tag)
    subprocess.run(["rm", "-rf", d])
    r = subprocess.run(["rocprofv3", "--pmc"] + ctrs + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", exe, str(REPS)],
                       capture_output=True, text=True, cwd=root)
    try:
        j = json.loads(r.stdout[r.stdout.index("{"):r.stdout.rindex("}") + 1])
    except ValueError:
        print("pass", tag, "gave no JSON:", r.stdout[-500:], r.stderr[-2000:], file=sys.stderr)
        continue
    variants = variants or j["variants"]
    trace = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not trace or not cc:
        print("pass", tag, "gave no csv", r.stderr[-2000:], file=sys.stderr)
        continue
    rows = sorted((x for x in csv.DictReader(open(trace[0])) if "rd_" in x["Kernel_Name"] or "wr_" in x["Kernel_Name"]),
                  key=lambda x: int(x["Start_Timestamp"]))
    assert len(rows) == REPS * len(j["variants"]), (len(rows), len(j["variants"]))
    vals = collections.defaultdict(dict)
    for x in csv.DictReader(open(cc[0])):
        vals[x["Dispatch_Id"]][x["Counter_Name"]] = vals[x["Dispatch_Id"]].get(x["Counter_Name"], 0.0) + float(x["Counter_Value"])
    for k, v in enumerate(j["variants"]):
        row = rows[k * REPS + REPS - 1]  # the last repetition of the variant
        per[v["name"]]["us_" + tag] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
        per[v["name"]].update(vals.get(row["Dispatch_Id"], {}))
        per[v["name"]]["kernel"] = row["Kernel_Name"]
res = {"buffer_bytes": None, "variants": []}
for v in variants or []:
    p = per[v["name"]]
    lines = 1 << 25
    e = dict(v)
    e["lines_touched"] = lines
    e["line_bytes"] = lines * 128
    e.update({k: p[k] for k in p if k != "kernel"})
    fs = p.get("FETCH_SIZE", 0.0) * 1024.0
    ws = p.get("WRITE_SIZE", 0.0) * 1024.0
    if v["name"].startswith("rd_") and fs:
        e["FETCH_SIZE_bytes"] = fs
        e["lines128_over_FETCH_SIZE"] = lines * 128 / fs
        e["requested_over_FETCH_SIZE"] = v["requested_bytes"] / fs
        e["bytes_per_rdreq_if_lines_are_fetched_whole"] = lines * 128 / max(p.get("TCC_EA0_RDREQ_sum", 0.0), 1.0)
    if v["name"].startswith("wr_") and ws:
        e["WRITE_SIZE_bytes"] = ws
        e["requested_over_WRITE_SIZE"] = v["requested_bytes"] / ws
    res["variants"].append(e)
    print(json.dumps(e))
os.makedirs(os.path.dirname(os.path.abspath(out_json)), exist_ok=True)
json.dump(res, open(out_json, "w"), indent=1)
