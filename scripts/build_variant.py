"""Experiment builds of the GPU library: python scripts/build_variant.py name -DX=1 [-DY=2 ...]  ->  build/librtow_<name>.so
(the product flags of build.py plus the given defines; scripts/gpu_ab.py takes the files).  Knobs of experiments are
compile-time defines here, never environment variables read by the library."""
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracing_in_one_weekend_amd import build as b

name, defs = sys.argv[1], sys.argv[2:]
out = os.path.join(b.ROOT, "build", f"librtow_{name}.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
b._run([hipcc] + b.HIPCC_FLAGS + defs + [f'-DRT_BUILD_ID="{name}"', "-o", out, b.gpu_sources()[0]])
print(out)
