#!/bin/bash
# usage: tools/profile_all.sh [round-tag]  -- rocprofv3 kernel-trace + PMC passes of bench.py for the four BASELINE workloads and the four
# SURVEY 8(f) rows, condensed into profiles/<round-tag>_<workload>.md/.json + profiles/traffic_<workload>.json (GPU box, repo root)
R=${1:-r04}
for w in fft4096 rfft2048 stft1024 c64_2p20 istft1024 magnitudes1024 fft2d_4096 bluestein1000; do
  echo "=== $w $(date +%T)"
  steps=50; [ $w = c64_2p20 ] && steps=10; [ $w = istft1024 ] && steps=20
  tools/profile.sh $w --workload $w --steps $steps > gpurun_out/prof_$w.log 2>&1
  python3 tools/summarize_profile.py gpurun_out/prof_$w ${R}_$w $w > gpurun_out/summ_$w.log 2>&1 || echo "summarize $w failed"
  case $w in fft4096|rfft2048|stft1024|c64_2p20)
    echo "--- sq $w $(date +%T)"
    tools/pmc_sq.sh $w --workload $w > gpurun_out/${R}_sq_$w.txt 2>&1;;
  esac
  # the raw traces are large: keep only the summaries
  rm -rf gpurun_out/prof_$w/trace gpurun_out/prof_$w/pmc_* gpurun_out/sq_$w
done
cp profiles/${R}_*.md profiles/${R}_*.json profiles/traffic_*.json gpurun_out/ 2>/dev/null
echo done $(date +%T)
