#!/bin/bash
# wav2vec2 body (16 x 499 frames, bf16) under rocprofv3 --kernel-trace, summarised per (kernel, grid) by
# tools/trace_by_grid.py -- gemm32_kernel runs three GEMM shapes per layer, which --stats would average together.
# usage: tests/prof_body.sh <tag>      -> gpurun_out/prof_<tag>_body/by_grid_{one,two}_pipelines.txt
tag=${1:-r5}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for n in 1 2; do
  name=$([ $n = 1 ] && echo one || echo two)
  out=$root/gpurun_out/prof_${tag}_body/s$n
  rm -rf $out; mkdir -p $out
  PPGS_AMD_W2V2_STREAMS=$n timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- python $root/tools/prof_w2v2_body.py > $out/run.log 2>&1
  { echo "# PPGS_AMD_W2V2_STREAMS=$n; wall time per forward:"; PPGS_AMD_W2V2_STREAMS=$n python $root/tools/time_w2v2_body.py 2>/dev/null | tail -n 1; python $root/tools/trace_by_grid.py $out/trace; } > $root/gpurun_out/prof_${tag}_body/by_grid_${name}_pipelines.txt 2>&1
  rm -rf $out/trace
done
cat $root/gpurun_out/prof_${tag}_body/by_grid_*.txt
