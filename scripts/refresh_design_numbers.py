"""Rewrites the generated block of DESIGN.md section 6 (between `<!-- numbers:begin -->` and `<!-- numbers:end -->`) and the
headline sentence of README.md from profiles/round6 (run after scripts/gpu_round_profiles.sh + copying its files there), so
that no current number in those files is typed by hand."""
import json
import os
import re

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(root, "profiles", "round6") + "/"


def sp(x):
    return f"{x:,.0f}".replace(",", " ")


b2 = json.load(open(P + "bench_config2.json"))
t2 = json.load(open(P + "traffic_config2.json")) if os.path.exists(P + "traffic_config2.json") else None
v2 = json.load(open(P + "valu_config2.json"))["kernels"] if os.path.exists(P + "valu_config2.json") else None
cb = b2["cpu_baseline"]
block = (f"| round 6: the grid walk read in its ISA (single-layer form, raw `v_min`, the cannot-happen exits as cold blocks, the shard of a claim as scalar state: `k_intersect_grid` -9 %, the frame -3 % on one box; the other experiments on the kernels filed, section 9); all device memory of a context in one range grown by a helper thread; wrappers and media nest without limit.  Leases of this round: 43.92 / 44.03 / 44.04 / 44.33 / 45.50 / 46.03 ms per frame before the grid work, 43.40 with the single-layer form, 43.51 with all of it — boxes differ by more than the change; the committed line is the last one taken, the A/Bs on one box are in `profiles/round6/ab_grid_walk_*` | **{sp(b2['value'])}** | **{b2['ms_per_step']:.2f}** | "
         f"{100 * b2['roofline']['frac']:.1f} % |\n"
         f"| CPU oracle, stream order + BVH, {cb['cores']} host cores (EPYC 9575F), the faster of the portable and the `-march=native` build | "
         f"{cb['value']:.1f} | — | — |\n\n"
         f"(`profiles/round6/bench_config2.json`: `python bench.py --steps 20 --warmup 5`, build `{b2['library_build_id']}`; CPU builds probed: "
         + "; ".join(f"{k.split(' (')[0]}: {v} Mray/s at 1 spp" for k, v in cb.get("builds_probed", {}).items()) + ".)\n\n"
         f"`roofline.frac` = algorithmic bytes of the trace step / its device time / 8 TB/s = {sp(b2['roofline']['achieved'])} GB/s / 8 000; whole path "
         f"(96 B / ray + 24 B / path) {b2['whole_path']['hbm_frac']:.3f}.  The kernels are bound by vector issue (section 4), so the honest companion is the VALU "
         f"figure of the bench line (`roofline.valu`")
if v2:
    block += ": " + ", ".join(f"`{k}` issues {x['issue_frac']:.2f} of the measured peak at {x['lane_util']:.2f} lane utilisation" for k, x in v2.items())
block += ") and the instruction count.  "
if t2:
    block += (f"Measured traffic (PMC, FETCH x 2 + WRITE): {sp((b2['roofline'].get('traffic') or t2['trace_step_bytes_per_launch']) / 1e6)} MB per launch = "
              f"{t2['traffic_over_algorithmic']:.3f} x the algorithmic bytes.  ")
block += (f"Handed to the host as the reference's output is (f32 frame + flipped RGB8 through `rt_render` into page-locked memory): "
          f"{sp(b2['value_host_inclusive'])} Mray/s ({100 * (b2['value_host_inclusive'] / b2['value'] - 1):+.1f} %; `value_host_inclusive`, never `value`).  ")
ff, fq = b2.get("first_frame") or {}, b2.get("first_frame_quiet_device") or {}
if ff.get("first_frame_ms") and fq.get("first_frame_ms"):
    def parts(f):
        pm = f["parts_ms"]
        return (f"{f['first_frame_ms']:.0f} ms (`rt_ctx_create` {pm['rt_ctx_create']:.0f}, host scene build {pm['scene_build_host']:.0f}, `rt_scene_upload` {pm['rt_scene_upload']:.0f}, "
                f"first `rt_render` {pm['first_rt_render']:.0f} in {f['first_render_slices']} slice(s) — {f['first_render_device_ms']:.1f} ms on the device against "
                f"{f['second_render_device_ms']:.1f} for the second frame, which takes {pm['second_rt_render']:.0f} ms in all — slowest chunk of the pool "
                f"{f['first_render_parts_ms'].get('pool_slowest_chunk_ms', 0):.0f} ms)")
    block += (f"What the reference's own timer covers — one frame per process (`main.rs:62-129`), here with `rt_prepare` before the scene is built — in a process of its own, "
              f"twice: before the bench process has touched the GPU `first_frame_ms` = {parts(fq)}; after the timed steps, beside the bench process and its "
              f"{ff['alloc_bytes'] / 1e9:.0f} GB, {parts(ff)}.  `alloc_bytes` = {ff['alloc_bytes'] / 1e9:.1f} GB.  (One request in a few waits 3-6 s inside the driver on this machine, "
              f"section 2: a first frame that meets it on one of the first chunks of its pool takes that long, as in `profiles/round6/first_chunk_stalls.txt`; met later it costs slices.)\n")
else:
    block += "\n"
p = os.path.join(root, "DESIGN.md")
s = open(p).read()
s = re.sub(r"<!-- numbers:begin -->.*<!-- numbers:end -->", lambda m: "<!-- numbers:begin -->\n" + block + "<!-- numbers:end -->", s, flags=re.S)
open(p, "w").write(s)
rp = os.path.join(root, "README.md")
r = open(rp).read()
r = re.sub(r"runs at [0-9.]+ Gray/s, [0-9.]+ ms per frame \(boxes differ by about ±2 %; ≈ [0-9]+× the 16-core CPU restatement\)",
           f"runs at {b2['value'] / 1e3:.1f} Gray/s, {b2['ms_per_step']:.1f} ms per frame (boxes differ by about ±2 %; ≈ {b2['value'] / cb['value']:.0f}× the 16-core CPU restatement)", r)
open(rp, "w").write(r)
print(block)
