#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory (gpurun_out/prof_<tag>/) into profiles/<name>.md + .json.

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE are in KiB,
collected in separate --pmc passes; on gfx950 FETCH_SIZE reports exactly half of the bytes of a coalesced
streaming read, so it is doubled; WRITE_SIZE is exact for streaming stores."""
import csv
import glob
import json
import sys
from collections import defaultdict
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import csrc_sha16  # noqa: E402  (the hash bench.py compares profiles/traffic_*.json against)


def short(name: str) -> str:
    name = name.replace("void kofft::", "").replace("kofft::", "")
    return name if len(name) < 160 else name[:157] + "..."


def newest(pattern):
    """gpurun merges every run's files into the same local directory: keep only the latest run's file."""
    import os
    files = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return files[-1:]


def provenance(src: Path) -> dict:
    """The source / library hashes recorded by tools/profile.sh WHEN the counters were collected.  A directory without the record
    (collected by an older script) gets the current source hash and says so."""
    f = src / "provenance.json"
    if f.exists():
        return json.loads(f.read_text())
    return {"csrc_sha16": csrc_sha16(), "provenance": "hash taken at summary time: this directory has no provenance.json"}


def main():
    src, name, workload = Path(sys.argv[1]), sys.argv[2], sys.argv[3]
    out_dir = Path(__file__).resolve().parent.parent / "profiles"
    out_dir.mkdir(exist_ok=True)
    summary = {"source": str(src), "workload": workload, "kernels": [], "pmc": {}}
    lines = [f"# rocprofv3 summary `{name}` (workload `{workload}`)", "",
             "Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline ...`",
             "(PMC counters in separate passes: `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, `--pmc SQ_*`; tools/profile.sh)", "",
             "## Kernel stats (--kernel-trace --stats)", "", "| kernel | calls | avg us | min us | max us | % |", "|---|---|---|---|---|---|"]
    for f in newest(str(src / "trace" / "**" / "*_kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            k = {"name": short(r["Name"]), "ours": "kofft::" in r["Name"], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                 "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]), "pct": float(r["Percentage"])}
            if "kofft" in r["Name"] or k["pct"] > 1.0:
                summary["kernels"].append(k)
                lines.append(f"| `{k['name']}` | {k['calls']} | {k['avg_ns'] / 1e3:.1f} | {k['min_ns'] / 1e3:.1f} | "
                             f"{k['max_ns'] / 1e3:.1f} | {k['pct']:.2f} |")
    # per-dispatch geometry from the kernel trace; registers / scratch / static LDS from the code object's own metadata
    # (the trace's VGPR column is in allocation granules and its LDS column does not carry dynamic LDS: as printed in
    # rounds 1-2 those contradicted the design record)
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    try:
        from codeobj_resources import kernel_table, lookup
        table = kernel_table()
    except Exception as e:  # no library / no llvm tools on this box: say so rather than print the trace's columns
        table, lookup = None, None
        lines += ["", f"(code-object metadata unavailable: {type(e).__name__}: {e})"]
    summary["dispatch"] = []
    for f in newest(str(src / "trace" / "**" / "*_kernel_trace.csv")):
        seen = set()
        for r in csv.DictReader(open(f)):
            if "kofft" in r["Kernel_Name"] and r["Kernel_Name"] not in seen:
                seen.add(r["Kernel_Name"])
                res = lookup(table, r["Kernel_Name"]) if table else None
                regs = (f"VGPR {res['vgpr']}, AGPR {res['agpr']}, SGPR {res['sgpr']}, scratch {res['scratch']} B/lane, static LDS "
                        f"{res['lds_static']} B (exchange buffers are dynamic LDS, a launch argument: DESIGN.md 5)") if res else \
                       "registers: kernel not found in the code objects"
                lines += ["", f"Dispatch of `{short(r['Kernel_Name'])}`: grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}, "
                          f"workgroup {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))}; code object: {regs}"]
                summary["dispatch"].append({"kernel": short(r["Kernel_Name"]), "grid": r.get("Grid_Size_X", r.get("Grid_Size")),
                                            "workgroup": r.get("Workgroup_Size_X", r.get("Workgroup_Size")), "code_object": res})
    agg = defaultdict(list)
    per_kernel = defaultdict(lambda: defaultdict(list))
    steps_launched = {}
    clk_samples = []  # (GRBM_GUI_ACTIVE summed over the 8 XCDs) / 8 / the dispatch's own duration = shader clock under this load, GHz
    for tag in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_clk"):
        log = src / f"{tag}.log"
        if log.exists():  # the bench line of that pass says how many steps (C-ABI calls) it launched
            js = [ln for ln in log.read_text().splitlines() if ln.startswith("{")]
            if js:
                steps_launched[tag] = json.loads(js[-1]).get("launches_total")
        for f in newest(str(src / tag / "**" / "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if "kofft" in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    per_kernel[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r.get("End_Timestamp"):
                        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                        if dur > 20000:  # (kernels of a few microseconds: the counter's bracket is wider than the kernel)
                            clk_samples.append(float(r["Counter_Value"]) / 8.0 / dur)
    lines += ["", "## PMC (mean per launch of a kofft kernel)", "", "| counter | launches | mean |", "|---|---|---|"]
    for k, v in sorted(agg.items()):
        summary["pmc"][k] = sum(v) / len(v)
        lines.append(f"| {k} | {len(v)} | {sum(v) / len(v):.6g} |")
    if len(per_kernel) > 1:
        lines += ["", "Per kernel:", ""]
        summary["pmc_per_kernel"] = {}
        for kn, d in sorted(per_kernel.items()):
            summary["pmc_per_kernel"][kn] = {c: sum(v) / len(v) for c, v in d.items()}
            lines.append(f"* `{kn}`: " + ", ".join(f"{c} {sum(v) / len(v):.6g} (n={len(v)})" for c, v in sorted(d.items())))
    if "FETCH_SIZE" in agg and "WRITE_SIZE" in agg:
        # one bench step = one C-ABI call = possibly several kernels: sum over every kofft kernel of the pass, divided by the
        # steps the pass launched (falls back to the per-kernel mean when the log has no launches_total)
        nf = steps_launched.get("pmc_fetch") or len(agg["FETCH_SIZE"])
        nw = steps_launched.get("pmc_write") or len(agg["WRITE_SIZE"])
        fetch = sum(agg["FETCH_SIZE"]) / nf * 1024 * 2  # gfx950: half-counted coalesced reads (guide section HBM)
        write = sum(agg["WRITE_SIZE"]) / nw * 1024
        summary["hbm_bytes_per_step"] = fetch + write
        summary["hbm_read_bytes"] = fetch
        summary["hbm_write_bytes"] = write
        summary["kernels_per_step"] = len(agg["FETCH_SIZE"]) / nf
        lines += ["", f"HBM traffic per bench step ({len(agg['FETCH_SIZE']) / nf:.2f} kernel launches per step; guide's gfx950 correction: "
                  f"FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024): "
                  f"read {fetch / 1e9:.4f} GB + write {write / 1e9:.4f} GB = **{(fetch + write) / 1e9:.4f} GB**"]
        issue = {}
        tl = src / "trace_bench.log"
        if tl.exists():  # which FORM of the workload the counters belong to (config #2: in place / out of place)
            js = [ln for ln in tl.read_text().splitlines() if ln.startswith("{")]
            ftxt = (json.loads(js[-1]).get("config", {}).get("form") or "") if js else ""
            if ftxt:
                issue["form"] = "inplace" if ftxt.startswith("in place") else "oop"
        nsq = steps_launched.get("pmc_sq") or len(agg.get("SQ_INSTS_VALU", [])) or 1
        if "SQ_INSTS_VALU" in agg:  # per bench STEP, like the traffic (a step may be several kernels)
            issue["valu_insts_per_step"] = sum(agg["SQ_INSTS_VALU"]) / nsq
        if "SQ_LDS_IDX_ACTIVE" in agg:
            issue["lds_active_cycles_per_step"] = sum(agg["SQ_LDS_IDX_ACTIVE"]) / nsq
        if clk_samples:
            clk_samples.sort()
            # /opt/skills/guides/MI355X_MICROARCH.md ("DVFS give-back"): the quotient reads HIGH on dispatches shorter than ~0.3 ms (the
            # counter's bracket is a few microseconds wider than the kernel) and within 3 % on long ones: clamped to the 2400 MHz max clock
            issue["shader_clock_ghz_raw"] = clk_samples[len(clk_samples) // 2]
            issue["shader_clock_ghz"] = min(issue["shader_clock_ghz_raw"], 2.4)
            issue["shader_clock_from"] = "median over dispatches of GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration under this workload, clamped to the 2.4 GHz max clock"
        if issue:
            summary["issue"] = issue
            lines += ["", "Issue-side figures per bench step: " + ", ".join(f"{k} {v:.6g}" for k, v in issue.items() if not isinstance(v, str))]
        (out_dir / f"traffic_{workload}.json").write_text(json.dumps(
            {"workload": workload, "hbm_bytes_per_step": fetch + write, "read_bytes": fetch, "write_bytes": write,
             "kernels_per_step": len(agg["FETCH_SIZE"]) / nf, **issue, "from": f"profiles/{name}.md", **provenance(src)}) + "\n")
    if "SQ_LDS_BANK_CONFLICT" in agg:
        lines += ["", f"LDS bank-conflict cycles / LDS active cycles: {summary['pmc']['SQ_LDS_BANK_CONFLICT']:.0f} / "
                  f"{summary['pmc'].get('SQ_LDS_IDX_ACTIVE', 0):.0f}"]
    for logname in ("trace_bench.log",):
        p = src / logname
        if p.exists():
            js = [ln for ln in p.read_text().splitlines() if ln.startswith("{")]
            if js:
                b = json.loads(js[-1])
                summary["bench_under_profiler"] = {k: b[k] for k in ("value", "unit", "ms_per_step", "roofline")}
                lines += ["", f"bench.py under the profiler: {b['value']:.1f} {b['unit']}, kernel avg {b['roofline']['kernel_ms_avg']:.4f} ms "
                          f"(HIP events) -- compare with the --stats average above."]
                # the same fraction from both clocks: --stats average of the kofft kernels of one step against the HIP events
                alg = b["roofline"].get("algorithmic_bytes_per_launch")
                kk = [k for k in summary["kernels"] if k.get("ours")]  # every kernel of namespace kofft (istft_ola_kernel, bluestein_wg_kernel ...)
                steps_total = b.get("launches_total")
                if alg and kk and steps_total:
                    step_ns = sum(k["avg_ns"] * k["calls"] for k in kk) / steps_total
                    frac_stats = alg / (step_ns * 1e-9) / 8e12
                    summary["frac_from_stats"] = frac_stats
                    summary["frac_from_hip_events"] = b["roofline"]["frac"]
                    lines += ["", f"Roofline fraction (algorithmic {alg / 1e9:.4f} GB per step / 8 TB/s): **{frac_stats:.4f}** from the --stats "
                              f"averages ({step_ns / 1e3:.1f} us of kofft kernels per step, all {steps_total} launches incl. ramp and warm-up), "
                              f"**{b['roofline']['frac']:.4f}** from bench.py's HIP events (timed steps only)."]
                    iss = summary.get("issue", {})
                    if "valu_insts_per_step" in iss and "shader_clock_ghz" in iss:
                        # 4 cycles per wave64 VALU instruction on a SIMD16, 1024 SIMDs; one LDS pipe per CU, 256 CUs
                        cyc = iss["shader_clock_ghz"] * step_ns
                        valu = iss["valu_insts_per_step"] * 4 / (1024 * cyc)
                        lds = iss.get("lds_active_cycles_per_step", 0.0) / (256 * cyc)
                        hbm = summary.get("hbm_bytes_per_step", alg) / (step_ns * 1e-9) / 8e12
                        summary["issue_frac"] = {"valu": valu, "lds": lds, "hbm_traffic": hbm}
                        lines += ["", f"Which roofline binds (per step, {iss['shader_clock_ghz']:.3f} GHz measured): VALU issue **{valu:.3f}** "
                                  f"(SQ_INSTS_VALU x 4 / (1024 SIMDs x clock x time)), LDS **{lds:.3f}** (SQ_LDS_IDX_ACTIVE / (256 CUs x clock x time)), "
                                  f"HBM **{hbm:.3f}** (PMC traffic / time / 8 TB/s) -> bound: **{max((valu, 'valu'), (lds, 'lds'), (hbm, 'hbm'))[1]}**"]
    (out_dir / f"{name}.md").write_text("\n".join(lines) + "\n")
    (out_dir / f"{name}.json").write_text(json.dumps(summary, indent=1) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
