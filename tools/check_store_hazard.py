#!/usr/bin/env python3
"""ISA check of the 16-byte-store hazard (fft_device.hip.h: b128_store_guard): in the DISASSEMBLY of libkofft_hip.so, no VALU
instruction may write a data register of a store of more than 64 bits within the two wait states behind that store.

On gfx950 a `buffer_store_dwordx4` whose data registers the next (f64) VALU instruction overwrites stores the NEW value in some lanes --
also when its soffset is an SGPR, the form for which the compiler pads nothing.  The library pins `s_nop 1` behind every such store; this
script is what keeps it that way: a kernel that stores 16-byte values around the two store helpers shows up here (tests/test_isa_store_guard.py
runs it on the built library, no GPU needed).

usage: tools/check_store_hazard.py [libkofft_hip.so]      exit status 1 and a listing when a violation is found"""
import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
LLVM = Path("/opt/rocm/lib/llvm/bin")
WAIT_STATES = 2  # what LLVM inserts for the hazard it knows on gfx940+, and what b128_store_guard pins

STORE = re.compile(r"^\s*(?:buffer|global|flat|scratch)_store_(?:dwordx[34]|format_xyzw?|format_d16_xyzw)\s+(v\[\d+:\d+\]|v\d+)")
INSN = re.compile(r"^\s+([a-z_0-9]+)\s*(.*)$")


def vrange(tok: str):
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return int(m.group(1)), int(m.group(2))
    m = re.match(r"v(\d+)$", tok)
    return (int(m.group(1)),) * 2 if m else None


def disassemble(lib: Path):
    with tempfile.TemporaryDirectory() as tmp:
        work = Path(tmp) / lib.name
        shutil.copy(lib, work)
        subprocess.run([str(LLVM / "llvm-objdump"), "--offloading", str(work)], capture_output=True, cwd=tmp, check=True)
        for co in sorted(Path(tmp).glob("*gfx950*")):
            out = subprocess.run([str(LLVM / "llvm-objdump"), "-d", "--no-show-raw-insn", str(co)], capture_output=True, text=True, check=True).stdout
            yield co.name, [ln.split("//")[0].rstrip() for ln in out.splitlines()]


def check(lib: Path):
    stores, violations = 0, []
    for name, lines in disassemble(lib):
        func = "?"
        for i, ln in enumerate(lines):
            if ln.endswith(">:"):
                func = ln.split("<")[-1][:-2]
                continue
            m = STORE.match(ln)
            if not m:
                continue
            data = vrange(m.group(1))
            if data is None or data[1] - data[0] < 2:  # more than 64 bits of data
                continue
            stores += 1
            waited, j = 0, i + 1
            while waited < WAIT_STATES and j < len(lines):
                nxt = lines[j]
                j += 1
                if nxt.endswith(">:"):
                    break  # the next function
                mi = INSN.match(nxt)
                if not mi:
                    continue
                op, rest = mi.group(1), mi.group(2)
                if op == "s_nop":
                    waited += int(rest.strip() or 0, 0) + 1
                    continue
                if op.startswith("v_"):
                    dst = vrange(rest.split(",")[0].strip())
                    if dst and not (dst[1] < data[0] or dst[0] > data[1]):
                        violations.append((name, func, ln.strip(), nxt.strip(), waited))
                        break
                waited += 1
    return stores, violations


def main() -> int:
    lib = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "kofft_amd" / "lib" / "libkofft_hip.so"
    stores, violations = check(lib)
    print(f"{lib}: {stores} stores of more than 64 bits, {len(violations)} with a VALU write of their data registers inside {WAIT_STATES} wait states")
    for name, func, st, nx, w in violations[:40]:
        print(f"  {func[:100]}\n      {st}\n      {nx}      (after {w} wait state(s))")
    return 1 if violations else 0


if __name__ == "__main__":
    sys.exit(main())
