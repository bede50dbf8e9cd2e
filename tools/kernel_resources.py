#!/usr/bin/env python3
"""Print VGPR / AGPR / spill / LDS / occupancy per kernel of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
extra = sys.argv[2:]
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + root + "/include", "-I" + root + "/koemorph_amd/csrc",
       "-c", "-Rpass-analysis=kernel-resource-usage", src, "-o", "/dev/null"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
for line in out.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = m.group(1); print(); print(subprocess.run(["c++filt", cur], capture_output=True, text=True).stdout.strip()[:90]); continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|TotalSGPRs): (\d+)", line)
    if m and cur:
        print(f"   {m.group(1)}: {m.group(2)}", end="")
print()
