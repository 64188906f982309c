#!/bin/bash
# rocprofv3 helper for gpurun: kernel-trace stats + two PMC passes.
# usage: tests/prof.sh <tag>
tag=${1:-r1}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-alt --steps 20 --warmup 5 > $out/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-alt --steps 3 --warmup 1 > $out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d $out/pmc2 -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-alt --steps 3 --warmup 1 > $out/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc3 -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-alt --steps 3 --warmup 1 > $out/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc4 -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-alt --steps 3 --warmup 1 > $out/pmc4.log 2>&1
find $out -type f | head -40
du -sh $out
