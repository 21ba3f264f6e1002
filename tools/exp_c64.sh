#!/bin/bash
# usage: tools/exp_c64.sh -- config 5 under the large-n path's knobs (run on the GPU box from the repo root)
run() { echo "== $*"; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-workloads --min-seconds 0.5 --workload c64_2p20 --steps 6 --warmup 2 2>&1 | python -c "import sys,json; [print('  ', round(j['value'],1), 'GPoints/s', round(j['roofline']['frac'],4), round(j['ms_per_step'],3), 'ms') for j in [json.loads(l) for l in sys.stdin if l.startswith('{')]]"; }
run X=1
for mb in 256 384 512 768 1024 2048; do
  run KOFFT_HIP_BIG_CHUNK_MB=$mb KOFFT_HIP_BIG_MID_NT=1
done
run KOFFT_HIP_BIG_CHUNK_MB=256 KOFFT_HIP_BIG_MID_NT=0
run KOFFT_HIP_PERSIST_GRID_PCT=200
