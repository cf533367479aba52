#!/bin/bash
# Round-2 artefacts on the current build: GPU suite, result tables, rocprofv3 kernel stats (headline + atrium) and the PMC passes of the headline command.
TAG=${1:-r02f}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q > $OUT/gpu_suite.log 2>&1; grep -n "passed\|failed" $OUT/gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
bash tools/run_r02_extra.sh $TAG/extra 2>&1 | tail -20
bash tools/run_r02_tables.sh $TAG/tables 2>&1 | tail -16
COMMON="--warmup 32 --repeats 2 --no-extras --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- python bench.py --steps 64 --warmup 32 --no-extras --no-cpu-baseline > $OUT/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_atrium -o b -- python bench.py --scene atrium --steps 64 --warmup 32 --no-extras --no-cpu-baseline > $OUT/stats_atrium.log 2>&1
run_pmc () { timeout 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -o b -- python bench.py $3 $COMMON > $OUT/$1.log 2>&1; }
for CFG in "head_s32:--steps 64" "head_s20:--steps 20"; do
  NAME=${CFG%%:*}; ARGS=${CFG#*:}
  run_pmc ${NAME}_fetch "FETCH_SIZE" "$ARGS"
  run_pmc ${NAME}_write "WRITE_SIZE" "$ARGS"
  run_pmc ${NAME}_l2 "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "$ARGS"
done
run_pmc head_s32_sq "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" "--steps 64"
run_pmc atrium_s32_l2 "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "--steps 64 --scene atrium"
run_pmc atrium_s32_sq "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" "--steps 64 --scene atrium"
python tools/prof_r02_summarize.py $OUT
find $OUT -name "*kernel_stats.csv" | head; ls $OUT/summary
# keep the merged output small: the raw per-dispatch tables are not needed
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +20M -delete
