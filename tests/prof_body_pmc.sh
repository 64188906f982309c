#!/bin/bash
# rocprofv3 --pmc passes (own runs, --kernel-trace only) over the wav2vec2 body alone (16 x 499 frames, bf16, one
# pipeline so that per-kernel counters are not those of overlapping launches), summarised by tests/pmc_summary.py.
# usage: tests/prof_body_pmc.sh <tag>      -> gpurun_out/prof_<tag>_bodypmc/summary.txt
tag=${1:-r4}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/prof_${tag}_bodypmc
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export PPGS_AMD_W2V2_STREAMS=1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc1 -o b -- python $root/tools/prof_w2v2_body.py > $out/pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d $out/pmc2 -o b -- python $root/tools/prof_w2v2_body.py > $out/pmc2.log 2>&1
# (no FETCH_SIZE / WRITE_SIZE pass: with the TCC counters this workload did not finish in 12 minutes on the pool's boxes)
python $root/tests/pmc_summary.py $(find $out -name "*counter_collection.csv" | sort) > $out/summary.txt 2>&1
PPGS_AMD_W2V2_STREAMS=1 python $root/tools/time_w2v2_body.py 2>/dev/null | tail -n 1 | sed 's/^/# one pipeline, wall time per forward: /' >> $out/summary.txt
cat $out/summary.txt | cut -c1-400
