#!/bin/bash
# quick A/B helper for gpurun: ms/step and per-kernel ms for env variants
#   usage: tests/bench_quick.sh "VAR=a VAR=b ..."   (one bench run per word)
for v in ${1:-PPGS_AMD_LIN_NT=0}; do
  echo "$v"
  env $v python bench.py --no-cpu --steps 30 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' ', round(d['value']/1e6,2),'Mfps', round(d['ms_per_step'],3),'ms', round(d['roofline']['achieved'],1),'TF', {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
done
