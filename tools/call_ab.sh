#!/bin/bash
# usage: tools/call_ab.sh <variant> [rounds]   -- alternate libppgs_amd_<variant>.so and libppgs_amd.so on one box
v=$1; n=${2:-3}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/ab_$v.txt; : > $out
PPGS_AMD_LIB=ppgs_amd/libppgs_amd_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bf16 or layer or encoder" 2>&1 | tail -3 | tee -a $out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step_stream_summed']; print(d['ms_per_step'], {a:round(b,4) for a,b in k.items()})"; }
for r in $(seq $n); do
  for lib in ppgs_amd/libppgs_amd_$v.so ppgs_amd/libppgs_amd.so; do
    echo "$lib two pipelines: $(PPGS_AMD_LIB=$lib python bench.py --steps 200 --warmup 20 --no-cpu --no-alt 2>/dev/null | line)" | tee -a $out
    echo "$lib one pipeline: $(PPGS_AMD_LIB=$lib PPGS_AMD_STREAMS=1 python bench.py --steps 200 --warmup 20 --no-cpu --no-alt 2>/dev/null | line)" | tee -a $out
  done
done
