# One gpurun call that produces every record of a round under gpurun_out/final_<tag>/ (copy what is to be judged
# into profiles/):   bash tools/final_records.sh r5
set -x
tag=${1:-r6}
cd $GRAFT_REPO_ROOT
out=gpurun_out/final_$tag
mkdir -p $out
timeout 1800 python -m pytest tests -m gpu -q > $out/gputests.log 2>&1; tail -n 3 $out/gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -n 5 $out/smoke.log
# the counter passes first: bench.py quotes roofline.traffic / mfma_busy_frac from the PMC summary under profiles/ whose
# header names the library it has loaded -- the summaries of THIS build are installed there (in this box's copy of the
# tree; copy them into the real profiles/ afterwards) before the bench lines are taken
timeout 900 bash tests/prof.sh $tag > $out/prof.log 2>&1
python tests/pmc_summary.py gpurun_out/prof_$tag/pmc*/bench_counter_collection.csv > $out/${tag}_pmc_summary.txt 2>$out/pmc_summary.err
find gpurun_out/prof_$tag/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${tag}_kernel_stats.csv
# the fp16x2 mode: kernel stats and the matrix-pipe counters of its step
( cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_x2/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --precision fp16x2 --no-cpu --no-alt --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/$out/prof_x2.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_x2/pmc1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --precision fp16x2 --no-cpu --no-alt --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/$out/prof_x2_pmc.log 2>&1 )
PMC_PRECISION=fp16x2 python tests/pmc_summary.py gpurun_out/prof_${tag}_x2/pmc1/bench_counter_collection.csv > $out/${tag}_pmc_summary_fp16x2.txt 2>>$out/pmc_summary.err
find gpurun_out/prof_${tag}_x2/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${tag}_kernel_stats_fp16x2.csv
round=$(echo $tag | sed 's/^\(r[0-9]*\).*/\1/')
cp $out/${tag}_pmc_summary.txt profiles/${round}_pmc_summary.txt
cp $out/${tag}_pmc_summary_fp16x2.txt profiles/${round}_pmc_summary_fp16x2.txt
# the driver's command first (a fresh process, 20 steps), then the default line
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.log 2>&1; grep "^{" $out/bench_driver.log | tail -1 > $out/${tag}_bench.json
timeout 400 python bench.py > $out/bench_default.log 2>&1; grep "^{" $out/bench_default.log | tail -1 > $out/${tag}_bench_1000steps.json
PPGS_AMD_STREAMS=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-alt > $out/bench_one.log 2>&1; grep "^{" $out/bench_one.log | tail -1 > $out/${tag}_one_pipeline_bench.json
timeout 300 python bench.py --precision fp32 --no-alt --no-cpu --steps 20 --warmup 5 > $out/bench_fp32.log 2>&1; grep "^{" $out/bench_fp32.log | tail -1 > $out/${tag}_bench_fp32.json
timeout 300 python bench.py --precision fp16x2 --no-alt --no-cpu --steps 20 --warmup 5 > $out/bench_x2.log 2>&1; grep "^{" $out/bench_x2.log | tail -1 > $out/${tag}_bench_fp16x2.json
# one pipeline under the profiler: the per-kernel whole-chip figures
( cd /tmp && export TMPDIR=/tmp && PPGS_AMD_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_one/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-alt --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/$out/prof_one.log 2>&1 )
find gpurun_out/prof_${tag}_one/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${tag}_one_pipeline_kernel_stats.csv
timeout 900 bash tests/prof_configs.sh $tag > $out/prof_configs.log 2>&1
for c in c3 c5 c5stream; do cp gpurun_out/prof_${tag}_$c/kernel_stats.csv $out/${tag}_kernel_stats_$c.csv; done
timeout 600 bash tests/prof_body.sh $tag > $out/prof_body.log 2>&1
for n in one two; do cp gpurun_out/prof_${tag}_body/by_grid_${n}_pipelines.txt $out/${tag}_body_by_grid_${n}_pipelines.txt; done
timeout 600 python tools/bench_c3.py > $out/c3.log 2>&1; grep "^{" $out/c3.log | tail -1 > $out/${tag}_bench_c3.json
timeout 600 python tools/bench_c3.py fp16x2 > $out/c3_x2.log 2>&1; grep "^{" $out/c3_x2.log | tail -1 > $out/${tag}_bench_c3_fp16x2.json
PPGS_BENCH_FORCE_DIST=1 timeout 600 python bench.py --workload c4 > $out/c4.log 2>&1; grep "^{" $out/c4.log | tail -1 > $out/${tag}_bench_c4.json
timeout 300 python tools/bench_streaming.py > $out/c5.log 2>&1; grep "^{" $out/c5.log | tail -1 > $out/${tag}_bench_c5.json
# the N-rank host path on the one GPU present (gloo; NOT a multi-GPU measurement): 8 launch loops through one GPU's queues
PPGS_BENCH_ALIAS_GPUS=1 timeout 600 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu --no-alt > $out/alias8.log 2>&1; grep "^{" $out/alias8.log | tail -1 > $out/${tag}_bench_alias8_dry_run.json
timeout 120 python tools/time_frontend.py > $out/time_frontend.txt 2>&1
# the step as a timeline: which kernels of the two pipelines overlap, where a queue idles (the profiler's own per-dispatch cost included)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_tl -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-alt --steps 20 --warmup 5 --prewarm-s 0.2 > $GRAFT_REPO_ROOT/$out/prof_tl.log 2>&1 )
python tools/step_timeline.py gpurun_out/prof_${tag}_tl 1 > $out/${tag}_step_timeline.txt 2>&1
timeout 200 python tools/two_stream_steps.py --streams 1 2 --steps 400 > $out/two_stream_steps.txt 2>&1
ls -la $out
# (the merge back takes <= 64 MiB: the traces stay on the box)
rm -rf gpurun_out/prof_${tag} gpurun_out/prof_${tag}_*
