#!/bin/bash
# usage: tools/pmc_sq.sh <tag> [bench args...] -- SQ issue/wait breakdown of the bench kernel (two PMC passes)
# profiling is single-rank only: a multi-rank bench.py starts child processes, and under rocprofv3 (whose preloaded library has
# already initialised the GPU) that is an exec from a GPU-initialised process -- refused here and by bench.py itself
for a in "$@"; do case "$a" in --gpus|--gpus=*) echo "$0: --gpus is not allowed under the profiler (single rank only)" >&2; exit 2;; esac; done
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/sq_$TAG
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/a -- python3 bench.py --no-cpu-baseline --no-extra-workloads --no-twin --min-seconds 0 --steps 3 --warmup 1 --ramp-ms 50 "$@" > $OUT/a.log 2>&1
echo "a rc=$?" >> $OUT/a.log
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAVES --output-format csv -d $OUT/b -- python3 bench.py --no-cpu-baseline --no-extra-workloads --no-twin --min-seconds 0 --steps 3 --warmup 1 --ramp-ms 50 "$@" > $OUT/b.log 2>&1
echo "b rc=$?" >> $OUT/b.log
timeout -k 10 400 rocprofv3 --pmc SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM --output-format csv -d $OUT/c -- python3 bench.py --no-cpu-baseline --no-extra-workloads --no-twin --min-seconds 0 --steps 3 --warmup 1 --ramp-ms 50 "$@" > $OUT/c.log 2>&1
echo "c rc=$?" >> $OUT/c.log
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in "abc":
    for f in glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "kofft::" not in k: continue
            acc[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            print(k)
            for c, v in sorted(d.items()):
                print(f"   {c:34s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
