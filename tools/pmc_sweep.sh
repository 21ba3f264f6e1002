#!/bin/bash
# usage: tools/pmc_sweep.sh <tag> <sweep args...> -- SQ counters for one sweep point
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/sqs_$TAG
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/a -- python3 tools/sweep.py "$@" > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $OUT/b -- python3 tools/sweep.py "$@" > $OUT/b.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in "ab":
    for f in glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "kofft" not in k: continue
            acc[k[:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            print(k)
            for c, v in sorted(d.items()):
                print(f"   {c:34s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
