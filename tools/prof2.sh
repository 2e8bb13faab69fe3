#!/bin/bash
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_cache
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCC|TCP|TA|TD|SQ|SQC|GRBM)_[A-Za-z0-9_]+" | sort -u > $OUT/counters.txt
wc -l $OUT/counters.txt
BENCH="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_WRITE_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_ATOMIC_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TA_BUSY_sum TA_TA_BUSY_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT/c$i -o c$i -- $BENCH > $OUT/c$i.log 2>&1
  tail -2 $OUT/c$i.log | cut -c1-200
done
ls $OUT
