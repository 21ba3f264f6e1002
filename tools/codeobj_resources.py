#!/usr/bin/env python3
"""Register / LDS / scratch figures of every kernel in libkofft_hip.so, read from the code objects' own metadata
(NT_AMDGPU_METADATA: what the loader uses), not from a profiler's trace columns.

usage: tools/codeobj_resources.py [substring filter ...]        (table on stdout)
       from codeobj_resources import kernel_table               (dict: demangled name -> figures)

`lds_static` is the code object's group_segment_fixed_size; the kernels here size their exchange buffers at launch (dynamic
LDS, the third launch argument), which no code object records -- DESIGN.md section 5 gives those per kernel."""
import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
LLVM = Path("/opt/rocm/lib/llvm/bin")


def _norm(name: str) -> str:
    return re.sub(r"\s+", "", name.replace("void ", ""))


def kernel_table(lib: Path = ROOT / "kofft_amd" / "lib" / "libkofft_hip.so") -> dict:
    import yaml

    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        work = Path(tmp) / lib.name
        shutil.copy(lib, work)  # llvm-objdump --offloading writes the bundles next to its input
        subprocess.run([str(LLVM / "llvm-objdump"), "--offloading", str(work)], capture_output=True, cwd=tmp, check=True)
        for co in sorted(Path(tmp).glob("*gfx950*")):
            notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], capture_output=True, text=True).stdout
            m = re.search(r"^\s*---\n(.*?)^\s*\.\.\.\s*$", notes, flags=re.S | re.M)
            if not m:
                continue
            meta = yaml.safe_load(m.group(1))
            kernels = meta.get("amdhsa.kernels", [])
            names = [k[".name"] for k in kernels]
            dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
            for k, d in zip(kernels, dem):
                out[_norm(d)] = {
                    "name": d, "vgpr": k.get(".vgpr_count"), "agpr": k.get(".agpr_count"), "sgpr": k.get(".sgpr_count"),
                    "scratch": k.get(".private_segment_fixed_size"), "lds_static": k.get(".group_segment_fixed_size"),
                    "vgpr_spill": k.get(".vgpr_spill_count"), "max_wg": k.get(".max_flat_workgroup_size"),
                }
    return out


def lookup(table: dict, trace_name: str):
    """Figures for a kernel as a rocprofv3 trace names it (demangled, possibly without the argument list)."""
    key = _norm(trace_name)
    if key in table:
        return table[key]
    head = key.split("(")[0]
    hits = [v for k, v in table.items() if k.split("(")[0] == head]
    return hits[0] if len(hits) == 1 else None


if __name__ == "__main__":
    filters = sys.argv[1:]
    print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch':>7} {'spill':>5} {'LDSst':>6} {'maxWG':>5}  kernel")
    for v in sorted(kernel_table().values(), key=lambda v: v["name"]):
        d = v["name"].replace("void kofft::", "").replace("kofft::", "")
        if filters and not all(f in d for f in filters):
            continue
        print(f"{v['vgpr']:5d} {v['agpr']:5d} {v['sgpr']:5d} {v['scratch']:7d} {v['vgpr_spill'] or 0:5d} {v['lds_static']:6d} {v['max_wg']:5d}  {d[:170]}")
