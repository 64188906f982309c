#!/bin/bash
# rocprofv3 kernel-trace stats of the OTHER configurations (tests/prof.sh does configs[1]):
#   c3: tools/bench_c3.py (w2v2fb: feature encoder + wav2vec2 body + hidden-512 PPG network)
#   c5: tools/bench_streaming.py (causal 64 x 160-frame chunks) and the batched KV-cached stream step
# usage: tests/prof_configs.sh <tag>      -> gpurun_out/prof_<tag>_{c3,c5,c5stream}/
tag=${1:-r5}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in c3 c5 c5stream; do
  out=$root/gpurun_out/prof_${tag}_$cfg
  mkdir -p $out
  case $cfg in
    c3) export PPGS_BENCH_C3_NATIVE_ONLY=1; cmd="python $root/tools/bench_c3.py" ;;
    c5) cmd="python $root/tools/bench_streaming.py --steps 100" ;;
    c5stream) cmd="python $root/tools/stream_profile.py 64 16" ;;
  esac
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- $cmd > $out/run.log 2>&1
  find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
done
ls $root/gpurun_out/prof_${tag}_*
