set -x
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_c2
mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q -x > $out/gputests.log 2>&1; tail -n 5 $out/gputests.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step_stream_summed']; print(d['ms_per_step'], {a:round(b,4) for a,b in k.items()})"; }
for r in 1 2 3; do
  echo "product two: $(python bench.py --steps 200 --warmup 20 --no-cpu --no-alt 2>/dev/null | line)" | tee -a $out/ab.txt
  echo "noQ-ablation two: $(PPGS_AMD_Q_IN_ATTN=1 python bench.py --allow-ablation --steps 200 --warmup 20 --no-cpu --no-alt 2>/dev/null | line)" | tee -a $out/ab.txt
  echo "product one: $(PPGS_AMD_STREAMS=1 python bench.py --steps 200 --warmup 20 --no-cpu --no-alt 2>/dev/null | line)" | tee -a $out/ab.txt
  echo "noQ-ablation one: $(PPGS_AMD_Q_IN_ATTN=1 PPGS_AMD_STREAMS=1 python bench.py --allow-ablation --steps 200 --warmup 20 --no-cpu --no-alt 2>/dev/null | line)" | tee -a $out/ab.txt
done
