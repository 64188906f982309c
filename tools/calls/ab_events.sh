set -x
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_ev
mkdir -p $out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['repeat_blocks_ms_per_step'])"; }
for v in evdev evnofence; do
  PPGS_AMD_LIB=ppgs_amd/libppgs_amd_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_pipelines or c2_full_size or head_kernel_vs" 2>&1 | tail -2 | tee -a $out/ab.txt
done
for r in 1 2 3 4; do
  for v in "" _evdev _evnofence; do
    echo "lib$v: $(PPGS_AMD_LIB=ppgs_amd/libppgs_amd$v.so python bench.py --allow-ablation --steps 400 --warmup 20 --no-cpu --no-alt 2>/dev/null | line)" | tee -a $out/ab.txt
  done
done
