#!/bin/bash
# usage: tools/exp_rfft.sh -- config 3 under prefetch depth / grid variations (run on the GPU box from the repo root)
run() { echo "== $*"; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-workloads --min-seconds 0.5 --workload rfft2048 --steps 10 --warmup 3 2>&1 | python -c "import sys,json; [print('  ', round(j['value'],1), j['unit'], round(j['roofline']['frac'],4), round(j['ms_per_step'],3), 'ms') for j in [json.loads(l) for l in sys.stdin if l.startswith('{')]]"; }
D1=KOFFT_HIP_LIB=$PWD/kofft_amd/lib_rfftd1/libkofft_hip.so
run $D1
run X=1
run $D1 KOFFT_HIP_PERSIST_GRID_PCT=200
run KOFFT_HIP_PERSIST_GRID_PCT=200
run $D1
run X=1
