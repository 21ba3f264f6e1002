#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel in a device-only assembly file (`hipcc --cuda-device-only -S`), from the
.amdhsa metadata.  usage: kernel_regs.py file.s [substring]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    def f(key):
        m = re.search(r"\.%s:\s+(\S+)" % key, blk)
        return m.group(1) if m else "?"
    name = f("name")
    if pat and pat not in name:
        continue
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except OSError:
        pass
    agpr = re.match(r"\s*(\d+)", blk).group(1)
    print(f"vgpr {f('vgpr_count'):>4} agpr {agpr:>3} sgpr {f('sgpr_count'):>3} scratch {f('private_segment_fixed_size'):>5} spill {f('vgpr_spill_count'):>3}  {name[:150]}")
