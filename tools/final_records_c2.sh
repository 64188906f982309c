set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 400 python bench.py > gpurun_out/final/bench.log 2>&1; grep "^{" gpurun_out/final/bench.log | tail -1 > gpurun_out/final/r2_bench.json
timeout 300 python bench.py --precision fp32 --no-alt > gpurun_out/final/bench_fp32.log 2>&1; grep "^{" gpurun_out/final/bench_fp32.log | tail -1 > gpurun_out/final/r2_bench_fp32.json
PPGS_AMD_STREAMS=2 timeout 300 python bench.py --no-cpu --no-alt > gpurun_out/final/bench_s2.log 2>&1; grep "^{" gpurun_out/final/bench_s2.log | tail -1 > gpurun_out/final/r2_bench_streams2.json
timeout 600 bash tests/prof.sh r2c > gpurun_out/final/prof.log 2>&1
timeout 120 python tools/time_frontend.py > gpurun_out/final/time_frontend.txt 2>&1
ls -la gpurun_out/final
timeout 600 python tools/bench_c3.py > gpurun_out/final/c3.log 2>&1
PPGS_BENCH_FORCE_DIST=1 timeout 600 python bench.py --workload c4 > gpurun_out/final/c4.log 2>&1
timeout 300 python tools/bench_streaming.py > gpurun_out/final/c5.log 2>&1
