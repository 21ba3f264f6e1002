#!/bin/bash
# usage: tools/bench_c64.sh [lib.so] -- config 5 bench line (value, frac, ms)
[ -n "${1:-}" ] && export KOFFT_HIP_LIB=$PWD/$1
timeout -k 10 200 python bench.py --no-cpu-baseline --workload c64_2p20 --steps 10 --warmup 3 2>&1 | python -c "import sys,json; [print(j['config']['workload'][:24], round(j['value'],1), round(j['roofline']['frac'],4), round(j['ms_per_step'],4)) for j in [json.loads(l) for l in sys.stdin if l.startswith('{')]]"
