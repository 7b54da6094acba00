"""Instruction mix per kernel of one 128-spp render (diagnostic): scalar / branch / LDS / vector-memory instructions next to the
VALU count, and the cycles the scalar and vector pipes were busy:  python scripts/pmc_instmix.py <outdir> [spp] [scene]"""
import collections
import csv
import glob
import os
import re
import subprocess
import sys

out = sys.argv[1]
spp = sys.argv[2] if len(sys.argv) > 2 else "128"
scene = sys.argv[3] if len(sys.argv) > 3 else "sphere_scene"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("TMPDIR", "/tmp")
GROUPS = {"mix": ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES"],
          "busy": ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INST_CYCLES_SALU", "SQ_WAIT_INST_ANY"],
          "valu": ["SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_CVT", "SQ_THREAD_CYCLES_VALU", "SQ_INST_CYCLES_VMEM"]}


def short(n):
    m = re.match(r"void rt::(k_\w+)<([^>]*)>", n)
    return (m.group(1) + "<" + m.group(2).replace(" ", "") + ">") if m else n.split("(")[0].replace("rt::", "")


tot = collections.defaultdict(lambda: collections.defaultdict(float))
for g, ctrs in GROUPS.items():
    d = os.path.join(out, g)
    subprocess.run(["rm", "-rf", d])
    cmd = ["rocprofv3", "--pmc"] + ctrs + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                                           os.path.join(root, "scripts", "gpu_depth_probe.py"), spp, "0", scene]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=root)
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not cc:
        print("no output for", g, r.stderr[-800:], file=sys.stderr)
        continue
    for row in csv.DictReader(open(cc[0])):
        tot[short(row["Kernel_Name"])][row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in sorted(tot.items(), key=lambda x: -x[1].get("SQ_INSTS_VALU", 0))[:6]:
    valu = max(v.get("SQ_INSTS_VALU", 0), 1)
    print(f"{k[:44]:44s} VALU {valu / 1e9:7.2f} G  per VALU: SALU {v.get('SQ_INSTS_SALU', 0) / valu:.3f} branch {v.get('SQ_INSTS_BRANCH', 0) / valu:.3f} "
          f"LDS {v.get('SQ_INSTS_LDS', 0) / valu:.3f} SMEM {v.get('SQ_INSTS_SMEM', 0) / valu:.3f} VMEM rd {v.get('SQ_INSTS_VMEM_RD', 0) / valu:.4f} wr {v.get('SQ_INSTS_VMEM_WR', 0) / valu:.4f}")
    wc = max(v.get("SQ_WAVE_CYCLES", 0), 1)
    print(f"{'':44s} of wave-cycles: VALU active {v.get('SQ_ACTIVE_INST_VALU', 0) / wc:.3f} scalar active {v.get('SQ_ACTIVE_INST_SCA', 0) / wc:.3f} LDS active {v.get('SQ_ACTIVE_INST_LDS', 0) / wc:.3f} "
          f"any {v.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f} wait-inst {v.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}; busy cycles {v.get('SQ_BUSY_CYCLES', 0) / 1e9:.2f} G, SALU inst cycles {v.get('SQ_INST_CYCLES_SALU', 0) / 1e9:.2f} G")
    print(f"{'':44s} VALU kinds: int32 {v.get('SQ_INSTS_VALU_INT32', 0) / valu:.3f} fma {v.get('SQ_INSTS_VALU_FMA_F32', 0) / valu:.3f} mul {v.get('SQ_INSTS_VALU_MUL_F32', 0) / valu:.3f} "
          f"add {v.get('SQ_INSTS_VALU_ADD_F32', 0) / valu:.3f} trans {v.get('SQ_INSTS_VALU_TRANS_F32', 0) / valu:.3f} cvt {v.get('SQ_INSTS_VALU_CVT', 0) / valu:.3f}")
