set -x
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_c1
mkdir -p $out
( cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/tl2 -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-alt --steps 20 --warmup 5 --prewarm-s 0.2 > $GRAFT_REPO_ROOT/$out/tl2.log 2>&1
  PPGS_AMD_STREAMS=1 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/tl1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --allow-ablation --no-cpu --no-alt --steps 20 --warmup 5 --prewarm-s 0.2 > $GRAFT_REPO_ROOT/$out/tl1.log 2>&1 )
python tools/step_timeline.py $out/tl2 2 > $out/timeline_two.txt 2>&1
python tools/step_timeline.py $out/tl1 1 > $out/timeline_one.txt 2>&1
tail -3 $out/tl2.log $out/tl1.log
rm -rf $out/tl2 $out/tl1
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > $out/bench.log 2>&1; grep "^{" $out/bench.log | tail -1 > $out/bench.json
cat $out/timeline_two.txt | head -80
