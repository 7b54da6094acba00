"""Summarises hipcc's -Rpass-analysis=kernel-resource-usage remarks: python scripts/kernel_resources.py build.log
(registers, scratch and waves/SIMD of every kernel; the numbers DESIGN.md quotes come from here)."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
for b in blocks:
    name = b.split("\n")[0].split(" [")[0]
    try:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    except OSError:
        pass

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"
    print("%-72s VGPR %4s AGPR %3s SGPR %3s scratch %4s occ %2s LDS %6s" % (name[-72:], g("VGPRs"), g("AGPRs"), g("SGPRs"),
          g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
