#!/bin/bash
# Round 3, VERDICT r2 item 1: does the L2 hit rate of the traversal launches move with the derived node order (IDKPT_NODE_LAYOUT) and with the
# trace order of the bounce launches (IDKPT_TRACE_ORDER), and does the launch time follow?  One rocprofv3 --pmc pass (L2 / L1 request counters,
# --kernel-trace only beside it) and one --kernel-trace --stats pass per configuration of `python bench.py --steps 64 --warmup 32` (two full
# 32-sample batches per repetition), headline view and interior view.  Summaries -> gpurun_out/$TAG/summary/, copied to profiles/r03_*.
TAG=${1:-r03p}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
COMMON="--steps 64 --warmup 32 --repeats 2 --no-extras --no-cpu-baseline"
for VIEW in headline interior; do
  for CFG in "ref_queue:0:0" "couples_queue:1:0" "ref_order:0:2" "couples_order:1:2"; do
    NAME=${CFG%%:*}; REST=${CFG#*:}; LAY=${REST%%:*}; ORD=${REST#*:}
    export IDKPT_NODE_LAYOUT=$LAY IDKPT_TRACE_ORDER=$ORD
    timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/${VIEW}_${NAME}_l2 -o b -- python bench.py --view $VIEW $COMMON > $OUT/${VIEW}_${NAME}_l2.log 2>&1
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${VIEW}_${NAME}_stats -o b -- python bench.py --view $VIEW $COMMON > $OUT/${VIEW}_${NAME}_stats.log 2>&1
  done
done
unset IDKPT_NODE_LAYOUT IDKPT_TRACE_ORDER
python tools/prof_r03_order_summarize.py $OUT
