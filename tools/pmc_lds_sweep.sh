#!/bin/bash
# usage: tools/pmc_lds_sweep.sh [sweep args] -- LDS bank-conflict share of every kernel the size sweep launches (one PMC pass)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/lds_sweep
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/a -- python3 tools/sweep.py --mb 128 "$@" > $OUT/a.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "kofft" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].replace("void kofft::", "").replace("kofft::", "").replace("host::", "")[:150]][r["Counter_Name"]] += float(r["Counter_Value"])
rows = sorted(((d.get("SQ_LDS_BANK_CONFLICT", 0) / max(d.get("SQ_LDS_IDX_ACTIVE", 0), 1), k) for k, d in acc.items()), reverse=True)
for share, k in rows:
    if share > 0.005:
        print(f"{share:6.3f}  {k}")
print(len(rows), "kernels,", sum(1 for s, _ in rows if s > 0.005), "with conflicts above 0.5 % of their LDS cycles")
PY
