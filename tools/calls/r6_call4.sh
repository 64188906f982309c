set -x
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_c4
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "q_rows_made or head_kernel_vs or c2_full_size or two_pipelines or 16bit_modes or fp16x2_mode_meets" > $out/qx_tests.log 2>&1; tail -n 25 $out/qx_tests.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step_stream_summed']; print(d['ms_per_step'], {a:round(b,4) for a,b in k.items()})"; }
for r in 1 2 3; do
  echo "QX two: $(python bench.py --steps 200 --warmup 20 --no-cpu --no-alt 2>/dev/null | line)" | tee -a $out/ab.txt
  echo "stored-Q two: $(PPGS_AMD_Q_IN_ATTN=0 python bench.py --allow-ablation --steps 200 --warmup 20 --no-cpu --no-alt 2>/dev/null | line)" | tee -a $out/ab.txt
  echo "QX one: $(PPGS_AMD_STREAMS=1 python bench.py --steps 200 --warmup 20 --no-cpu --no-alt 2>/dev/null | line)" | tee -a $out/ab.txt
  echo "stored-Q one: $(PPGS_AMD_Q_IN_ATTN=0 PPGS_AMD_STREAMS=1 python bench.py --allow-ablation --steps 200 --warmup 20 --no-cpu --no-alt 2>/dev/null | line)" | tee -a $out/ab.txt
done
