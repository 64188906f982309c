set -x
cd $GRAFT_REPO_ROOT
v=$1
out=gpurun_out/ab_$v.txt; : > $out
PPGS_AMD_LIB=ppgs_amd/libppgs_amd_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bf16 or layer or encoder or head_kernel or two_pipelines or c2_full or c3 or fp16" 2>&1 | tail -3 | tee -a $out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step_stream_summed']; print(d['ms_per_step'], [round(x,4) for x in d['repeat_blocks_ms_per_step']], {a:round(b,4) for a,b in k.items() if b})"; }
for r in 1 2 3; do
  for lib in ppgs_amd/libppgs_amd_$v.so ppgs_amd/libppgs_amd.so; do
    echo "$lib two: $(PPGS_AMD_LIB=$lib python bench.py --allow-ablation --steps 200 --warmup 20 --no-cpu --no-alt 2>/dev/null | line)" | tee -a $out
    echo "$lib one: $(PPGS_AMD_LIB=$lib PPGS_AMD_STREAMS=1 python bench.py --allow-ablation --steps 200 --warmup 20 --no-cpu --no-alt 2>/dev/null | line)" | tee -a $out
  done
done
