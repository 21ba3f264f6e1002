#!/bin/bash
# usage: tools_profile.sh <tag> [bench args...]  -- run on the GPU box from the repo root
# 1) kernel-trace stats  2) PMC passes (FETCH_SIZE / WRITE_SIZE / SQ issue + LDS / GRBM_GUI_ACTIVE = shader clock)  -- separate runs, as the guide prescribes
# profiling is single-rank only: a multi-rank bench.py starts child processes, and under rocprofv3 (whose preloaded library has
# already initialised the GPU) that is an exec from a GPU-initialised process -- refused here and by bench.py itself
for a in "$@"; do case "$a" in --gpus|--gpus=*) echo "$0: --gpus is not allowed under the profiler (single rank only)" >&2; exit 2;; esac; done
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
# provenance of the counters: the kernel sources and the library that actually ran, recorded NOW (summarize_profile.py copies these)
python3 -c "import bench, hashlib, json; print(json.dumps({'csrc_sha16': bench.csrc_sha16(), 'lib_sha16': hashlib.sha256(open('kofft_amd/lib/libkofft_hip.so','rb').read()).hexdigest()[:16]}))" > $OUT/provenance.json
cd $PWD
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --no-extra-workloads --no-twin --min-seconds 0 "$@" > $OUT/trace_bench.log 2>&1
echo "trace rc=$?" >> $OUT/trace_bench.log
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --no-cpu-baseline --no-extra-workloads --no-twin --min-seconds 0 --steps 3 --warmup 1 "$@" > $OUT/pmc_fetch.log 2>&1
echo "fetch rc=$?" >> $OUT/pmc_fetch.log
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --no-cpu-baseline --no-extra-workloads --no-twin --min-seconds 0 --steps 3 --warmup 1 "$@" > $OUT/pmc_write.log 2>&1
echo "write rc=$?" >> $OUT/pmc_write.log
timeout -k 10 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -- python3 bench.py --no-cpu-baseline --no-extra-workloads --no-twin --min-seconds 0 --steps 3 --warmup 1 "$@" > $OUT/pmc_sq.log 2>&1
echo "sq rc=$?" >> $OUT/pmc_sq.log
# the shader clock UNDER THIS LOAD: GRBM_GUI_ACTIVE (busy cycles per XCD, summed over the 8 XCDs) over each dispatch's own duration
timeout -k 10 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_clk -- python3 bench.py --no-cpu-baseline --no-extra-workloads --no-twin --min-seconds 0 --steps 3 --warmup 1 "$@" > $OUT/pmc_clk.log 2>&1
echo "clk rc=$?" >> $OUT/pmc_clk.log
find $OUT -name "*.csv" | head -50
