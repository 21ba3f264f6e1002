#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS / occupancy table of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage).

usage: tools/kernel_resources.py kofft_amd/csrc/k_complex_f64.hip [substring filter ...]"""
import re
import subprocess
import sys
from pathlib import Path

src = Path(sys.argv[1]).resolve()
filters = sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
       "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", str(src)]
import os
cmd[1:1] = os.environ.get("EXTRA_HIPFLAGS", "").split()
res = subprocess.run(cmd, capture_output=True, text=True, cwd=src.parent)
txt = res.stderr
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
names = [b.split("\n")[0].strip() for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch':>7} {'occ':>4} {'LDS':>7}  kernel")
for b, d in zip(blocks, dem):
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    d = d.replace("void kofft::", "").replace("kofft::", "")
    if filters and not all(f in d for f in filters):
        continue
    print(f"{g('VGPRs'):5d} {g('AGPRs'):5d} {g('SGPRs'):5d} {g('ScratchSize .bytes/lane.'):7d} {g('Occupancy .waves/SIMD.'):4d} {g('LDS Size .bytes/block.'):7d}  {d[:170]}")
