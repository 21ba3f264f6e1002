#!/bin/bash
# usage: tools/trace_only.sh <tag> [bench args...] -- one rocprofv3 --kernel-trace --stats pass of bench.py, per-kernel table on stdout
# profiling is single-rank only: a multi-rank bench.py starts child processes, and under rocprofv3 (whose preloaded library has
# already initialised the GPU) that is an exec from a GPU-initialised process -- refused here and by bench.py itself
for a in "$@"; do case "$a" in --gpus|--gpus=*) echo "$0: --gpus is not allowed under the profiler (single rank only)" >&2; exit 2;; esac; done
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/trace_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --no-extra-workloads --min-seconds 0 "$@" > $OUT/bench.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/trace/**/*_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "kofft" in r["Name"]:
            print(f"{float(r['AverageNs'])/1e3:10.1f} us avg  {int(r['Calls']):6d} calls  {float(r['Percentage']):6.2f} %  {r['Name'].replace('void kofft::','')[:110]}")
PY
grep "^{" $OUT/bench.log | python3 -c "import sys,json; [print('bench:', round(j['value'],1), j['unit'], 'frac', round(j['roofline']['frac'],4), round(j['ms_per_step'],3), 'ms') for j in map(json.loads, sys.stdin)]"
