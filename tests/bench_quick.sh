#!/bin/bash
# quick A/B helper for gpurun: prints ms/step and per-kernel ms for env variants
for nt in 0 2 3; do
  echo "NT=$nt"
  PPGS_AMD_FFN_NT=$nt python bench.py --no-cpu --steps 30 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2),'Mfps', round(d['ms_per_step'],3),'ms', round(d['roofline']['achieved'],1),'TF', {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
done
