#!/bin/bash
# usage (GPU box, repo root): tools/kernel_coverage.sh <tag>
# Which kernel instantiations does the whole GPU suite launch?  The in-process GPU tests (everything but the tests that start child
# processes: under rocprofv3 that would be an exec from a GPU-initialised process) + the randomised soaks run under
# `rocprofv3 --kernel-trace --stats`; tools/kernel_coverage.py then lists what the library holds and nothing launched.
set -u
TAG=$1
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/cov_$TAG
mkdir -p $OUT
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tests -- python3 -m pytest tests -m gpu -q \
    --deselect tests/test_gpu_streams.py::test_bench_two_rank_rehearsal_carries_the_single_process_gather_ab \
    --deselect tests/test_gpu_streams.py::test_bench_single_rank_line_is_compact_and_complete \
    --ignore tests/test_gpu_cpp_mirror.py > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log
tail -3 $OUT/tests.log
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/soak -- python3 tools/soak_fuzz.py 100 140 > $OUT/soak.log 2>&1
echo "soak rc=$?" >> $OUT/soak.log
tail -2 $OUT/soak.log
python3 tools/kernel_coverage.py $OUT > $OUT/coverage.txt 2>&1
tail -5 $OUT/coverage.txt
# (the traces themselves are large: only the per-kernel stats tables and the coverage list are kept)
find $OUT -name "*kernel_trace.csv" -delete
