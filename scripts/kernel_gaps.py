"""Idle time between consecutive kernels of each HIP queue in one timed frame of bench.py (diagnostic):
    python scripts/kernel_gaps.py <outdir>     runs rocprofv3 --kernel-trace on `bench.py --steps 1 --warmup 1 --timed-only`"""
import collections
import csv
import glob
import os
import subprocess
import sys

out = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("TMPDIR", "/tmp")
subprocess.run(["rm", "-rf", out])
r = subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable, os.path.join(root, "bench.py"),
                    "--steps", "1", "--warmup", "1", "--timed-only"], capture_output=True, text=True, cwd=root)
f = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda x: int(x["Start_Timestamp"]))
# the last frame: from the last k_init_counts on
last = max(i for i, x in enumerate(rows) if "k_init_counts" in x["Kernel_Name"])
rows = rows[last:]
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(x["End_Timestamp"]) for x in rows)
print(f"frame: {len(rows)} dispatches, {(t1 - t0) / 1e6:.3f} ms from the first start to the last end")
byq = collections.defaultdict(list)
for x in rows:
    byq[x["Queue_Id"]].append(x)
for q, xs in byq.items():
    busy = sum(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for x in xs)
    gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(xs, xs[1:])]
    pos = [g for g in gaps if g > 0]
    print(f"queue {q}: {len(xs)} dispatches, busy {busy / 1e6:.3f} ms, gaps {sum(pos) / 1e6:.3f} ms in {len(pos)} gaps "
          f"(median {sorted(pos)[len(pos) // 2] / 1e3 if pos else 0:.1f} us, max {max(pos) / 1e3 if pos else 0:.1f} us), overlapping starts {sum(1 for g in gaps if g <= 0)}")
