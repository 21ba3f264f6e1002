#!/bin/bash
# usage: tools/profile_all.sh -- rocprofv3 kernel-trace + PMC passes of bench.py for the four BASELINE workloads (GPU box, repo root)
for w in fft4096 rfft2048 stft1024 c64_2p20; do
  echo "=== $w $(date +%T)"
  steps=50; [ $w = c64_2p20 ] && steps=10
  tools/profile.sh $w --workload $w --steps $steps > gpurun_out/prof_$w.log 2>&1
  echo "--- sq $w $(date +%T)"
  tools/pmc_sq.sh $w --workload $w > gpurun_out/sq_$w.txt 2>&1
done
echo done $(date +%T)
