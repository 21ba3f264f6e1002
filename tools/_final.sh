# the round's measurement pass on one box: bench line, profiles of all eight workloads, size sweep, 2-D A/B, soaks
mkdir -p gpurun_out/r06f
python3 bench.py > gpurun_out/r06f/bench_n1.json 2> gpurun_out/r06f/bench_n1.err; echo "bench rc=$?"
tools/profile_all.sh r06 > gpurun_out/r06f/profile_all.log 2>&1; echo "profiles rc=$?"
python3 tools/sweep.py --kinds c32,c64,rfft32,irfft32,stft,rfft64,irfft64 --out gpurun_out/r06f/sweep.json > gpurun_out/r06f/sweep.log 2>&1; echo "sweep rc=$?"
for e in 1 0; do echo "KOFFT_HIP_ND_FUSED=$e"; KOFFT_HIP_ND_FUSED=$e python3 tools/bench_nd.py f32 1x4096x4096 1x2048x4096 1x1024x4096 1x2048x2048 1x1024x1024 1x4096x1024 1x4096x2048 1x1024x2048 1x512x4096 1x8192x2048 1x8192x1024 1x8192x4096 256x256x256 2>&1 | grep -v amdgpu; done > gpurun_out/r06f/fft2d_ab.txt
mkdir -p gpurun_out/r06f/profiles && cp profiles/r06_* profiles/traffic_* gpurun_out/r06f/profiles/
echo done
