#!/bin/bash
# Growth of the pool beside a tenant that holds 53 GB in its own pool (the bench process), by chunk size: first-frame child with
# 128 MB and 512 MB chunks, alternating.  Output: gpurun_out/r6/chunk_size_beside_tenant.txt
out=gpurun_out/r6; mkdir -p $out; log=$out/chunk_size_beside_tenant.txt; : > $log
python - <<'PY' &
import time, sys
sys.path.insert(0, ".")
import ray_tracing_in_one_weekend_amd as rt
rt.register_default_images()
scene = rt.Scene.build("sphere_scene", 16 / 9)
r = rt.Renderer(0)
r.upload(scene)
for _ in range(3):
    r.render(scene.camera, rt.make_params(1920, 1080, 256, max_depth=50))
print("tenant holds its pool", flush=True)
time.sleep(75)
r.close()
PY
tenant=$!
sleep 12
for k in 1 2 3 4; do
  for lib in "" build/librtow_chunk512.so; do
    RTOW_GPU_LIB=$lib timeout -k 10 120 python bench.py --first-frame-child --config 2 | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p = d['parts_ms']; q = d['first_render_parts_ms']
print('${lib:-128 MB chunks (product)}: first_frame %.0f ms = ctx %.0f + scene %.0f + upload %.0f + first render %.0f (%d slices), second render %.0f (%d slices); pool %d MB after the first frame, slowest chunk %.0f ms' % (
      d['first_frame_ms'], p['rt_ctx_create'], p['scene_build_host'], p['rt_scene_upload'], p['first_rt_render'], d['first_render_slices'], p['second_rt_render'], d['second_render_slices'], q['pool_mapped_mb'], q['pool_slowest_chunk_ms']))" >> $log
  done
done
wait $tenant
cat $log
