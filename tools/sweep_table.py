#!/usr/bin/env python3
"""profiles/<name>.md from one or two tools/sweep.py json files (the second: the same box with a knob set, for A/B columns).
usage: tools/sweep_table.py <name> <sweep.json> [--ab <label> <other.json> <n,n,...>] [--note "..."]"""
import json
import sys
from pathlib import Path

KINDS = ["c32", "c64", "rfft32", "irfft32", "stft", "rfft64", "irfft64"]


def load(p):
    rows = json.loads(Path(p).read_text())
    return {(r["kind"], r["n"]): r for r in rows}


def main():
    name, main_json = sys.argv[1], sys.argv[2]
    ab_label, ab, ab_ns = None, None, set()
    note = ""
    a = sys.argv[3:]
    while a:
        if a[0] == "--ab":
            ab_label, ab = a[1], load(a[2])
            ab_ns = {int(v) for v in a[3].split(",")}
            a = a[4:]
        elif a[0] == "--note":
            note = a[1]
            a = a[2:]
        else:
            raise SystemExit(f"unknown argument {a[0]}")
    d = load(main_json)
    ns = sorted({n for (_, n) in d})
    out = [f"# Size sweep `{name}` (tools/sweep.py, one box, ~512 MiB per launch)", "",
           "Fraction of the 8 TB/s HBM roofline from algorithmic bytes and HIP-event time (20 launches after 5 warm-ups); "
           "box-to-box spread +-5 % typically, up to 15-20 % seen in round 4 (STFT 1024: 0.57 on one box, 0.70-0.78 on others) -- compare columns of ONE file only." + (" " + note if note else ""), "",
           "| n | " + " | ".join(KINDS) + " |", "|---|" + "---|" * len(KINDS)]
    for n in ns:
        cells = []
        for k in KINDS:
            r = d.get((k, n))
            c = f"{r['frac']:.3f}" if r else ""
            if r and ab and n in ab_ns and (k, n) in ab:
                c += f" ({ab_label} {ab[(k, n)]['frac']:.3f})"
            cells.append(c)
        out.append(f"| {n} | " + " | ".join(cells) + " |")
    p = Path(__file__).resolve().parent.parent / "profiles" / f"{name}.md"
    p.write_text("\n".join(out) + "\n")
    (p.with_suffix(".json")).write_text(Path(main_json).read_text())
    print("\n".join(out))


if __name__ == "__main__":
    main()
