#!/bin/bash
# rocprofv3 passes for bench.py (scratch helper for gpurun).  $1 = tag (e.g. r02), optional $2.. = extra bench.py arguments
# (e.g. "--arch vit_base_patch16_224 --index-rows 1000000" for BASELINE configs[3]).  Counters go in their own passes.
TAG=${1:-r02}; shift
EXTRA="$@"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 3 --breakdown $EXTRA $( [ -n "$EXTRA" ] && echo "--no-extras --no-cpu-baseline" ) > $OUT/bench.json 2> $OUT/bench_breakdown.txt
BENCH="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras $EXTRA"
echo "$BENCH" > $OUT/cmd.txt
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- $BENCH > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc1 -o pmc1 -- $BENCH > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $OUT/pmc2 -o pmc2 -- $BENCH > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- $BENCH > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc4 -o pmc4 -- $BENCH > $OUT/pmc4.log 2>&1
# keep only what tools/rocpd.py reads (the merged gpurun_out is capped at 64 MiB)
find $OUT -name "*.db" -size +30M -delete
grep -v amdgpu $OUT/bench_breakdown.txt | head -12; cut -c1-300 $OUT/bench.json
# summarise on the box and hand the text files back through gpurun_out/ (only that directory is merged into the checkout)
python tools/rocpd.py $TAG > /dev/null 2>&1
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG && cp profiles/${TAG}_* $GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG/
find $OUT -name "*.db" -delete
