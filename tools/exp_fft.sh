#!/bin/bash
# usage: tools/exp_fft.sh -- config 2 under grid variations (run on the GPU box from the repo root)
run() { echo "== $*"; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-workloads --min-seconds 0.5 --workload fft4096 --steps 20 --warmup 5 2>&1 | python -c "import sys,json; [print('  ', round(j['value'],1), j['unit'], round(j['roofline']['frac'],4), round(j['ms_per_step'],4), 'ms') for j in [json.loads(l) for l in sys.stdin if l.startswith('{')]]"; }
run X=1
run KOFFT_HIP_PERSIST_GRID_PCT=50
run KOFFT_HIP_PERSIST_GRID_PCT=75
run KOFFT_HIP_PERSIST_GRID_PCT=150
run X=1
run KOFFT_HIP_PERSIST_GRID_PCT=50
