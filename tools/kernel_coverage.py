#!/usr/bin/env python3
"""Kernel instantiations of libkofft_hip.so that NO run under tools/kernel_coverage.sh launched.

usage: tools/kernel_coverage.py <dir with rocprofv3 --stats output>      (list on stdout: family counts, then every unlaunched kernel)

Names are compared after removing whitespace (the code object's demangled names against the profiler's)."""
import collections
import csv
import re
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from codeobj_resources import kernel_table  # noqa: E402


def norm(name: str) -> str:
    name = re.sub(r"\s+", "", name.replace("void ", ""))
    name = re.sub(r"\(.*$", "", name)  # the profiler drops the parameter list
    return name.replace("kofft::", "")


def launched(root: Path) -> collections.Counter:
    seen = collections.Counter()
    for f in root.rglob("*kernel_stats.csv"):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                seen[norm(row["Name"])] += int(row["Calls"])
    return seen


def main() -> None:
    root = Path(sys.argv[1])
    seen = launched(root)
    table = {norm(k): v for k, v in kernel_table().items()}
    missing = sorted(k for k in table if k not in seen)
    print(f"library: {len(table)} kernels, launched: {sum(1 for k in table if k in seen)}, never launched: {len(missing)}")
    fam = collections.Counter(re.match(r"[\w:]+", k).group(0) for k in missing)
    for f, c in fam.most_common():
        print(f"  {c:4d} {f}")
    print()
    for k in missing:
        print(k)
    stray = sorted(k for k in seen if k not in table and not k.startswith(("at::", "void at::", "__amd", "rccl", "nccl")))
    if stray:
        print("\nlaunched but not in the library (torch / runtime kernels):")
        for k in stray[:40]:
            print("  ", k)


if __name__ == "__main__":
    main()
