#!/bin/bash
# rocprofv3 passes for bench.py (scratch helper for gpurun).  $1 = tag.  Counters go in their own passes.
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 3 --breakdown > $OUT/bench.json 2> $OUT/bench_breakdown.txt
BENCH="python bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- $BENCH > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc1 -o pmc1 -- $BENCH > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $OUT/pmc2 -o pmc2 -- $BENCH > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- $BENCH > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc4 -o pmc4 -- $BENCH > $OUT/pmc4.log 2>&1
cat $OUT/bench_breakdown.txt | grep -v amdgpu; cat $OUT/bench.json
