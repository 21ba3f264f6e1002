#!/bin/bash
# usage: tools/exp_irfft.sh <n> <variant names...> -- irfft32 at one length under library variants (GPU box, repo root)
N=$1; shift
run() { echo "== $*"; env "$@" timeout -k 10 200 python tools/sweep.py --kinds irfft32 --only-n $N --out gpurun_out/exp_irfft.json 2>&1 | grep "^irfft32"; }
run X=1
for v in "$@"; do run KOFFT_HIP_LIB=$PWD/kofft_amd/lib_$v/libkofft_hip.so; done
run X=1
for v in "$@"; do run KOFFT_HIP_LIB=$PWD/kofft_amd/lib_$v/libkofft_hip.so; done
