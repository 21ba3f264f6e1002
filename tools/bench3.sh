#!/bin/bash
# usage: tools/bench3.sh [lib.so] -- the three f32 bench lines (value, frac, ms) with an optional library override
[ -n "${1:-}" ] && export KOFFT_HIP_LIB=$PWD/$1
for w in stft1024 rfft2048 fft4096; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w 2>&1 | python -c "import sys,json; [print(j['config']['workload'][:24], round(j['value'],1), round(j['roofline']['frac'],4), round(j['ms_per_step'],4)) for j in [json.loads(l) for l in sys.stdin if l.startswith('{')]]"
done
