mkdir -p gpurun_out/pre_ab

for r in 1 2; do for v in new old; do if [ $v = new ]; then unset KOFFT_HIP_LIB; else export KOFFT_HIP_LIB=$PWD/kofft_amd/lib_old/libkofft_hip.so; fi
python3 tools/sweep.py --kinds rfft32 --min-n 65536 --max-n 4194304 --out gpurun_out/pre_ab/${v}_$r.json > gpurun_out/pre_ab/${v}_$r.log 2>&1
python3 tools/bench_bluestein.py f32 12345:4096 100003:512 1000003:64 > gpurun_out/pre_ab/blue_${v}_$r.log 2>&1; python3 tools/bench_bluestein.py f64 12345:2048 100003:256 > gpurun_out/pre_ab/blue64_${v}_$r.log 2>&1
done; done
grep -h "rfft" gpurun_out/pre_ab/new_2.log | tail -12
