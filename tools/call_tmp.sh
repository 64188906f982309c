cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_call18; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q -x > $out/gputests.log 2>&1; tail -n 3 $out/gputests.log
for i in 1 2 3; do
  for k in 0 1; do
    PPGS_AMD_OUTCONV_KSPLIT=$k timeout 300 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu --no-alt --allow-ablation > $out/b.log 2>$out/err.log || tail -3 $out/err.log
    echo "ksplit=$k two pipelines: $(grep '^{' $out/b.log | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["kernel_ms_per_step_stream_summed"]["outconv_softmax"])')"
    PPGS_AMD_STREAMS=1 PPGS_AMD_OUTCONV_KSPLIT=$k timeout 300 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu --no-alt --allow-ablation > $out/b.log 2>$out/err.log || tail -3 $out/err.log
    echo "ksplit=$k one pipeline: $(grep '^{' $out/b.log | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["kernel_ms_per_step_stream_summed"]["outconv_softmax"])')"
  done
done | tee $out/ab.txt
