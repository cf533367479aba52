#!/bin/bash
# Round-2 judged artefacts: rocprofv3 summaries of the bench command.
#   stats  : rocprofv3 --kernel-trace --stats of `python bench.py --steps 64 --warmup 32 --no-extras --no-cpu-baseline` (2 full 32-sample batches per repetition)
#   pmc    : separate --pmc passes (FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum | SQ lane-utilisation set) of the same command,
#            of the driver's command (--steps 20: one 20-sample batch per repetition) and of the interior view
# Collected with --kernel-trace only next to --pmc (gpurun refuses other trace domains beside counters).  Summaries -> gpurun_out/$TAG/summary/, copy to profiles/.
TAG=${1:-r02p}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
COMMON="--warmup 32 --repeats 2 --no-extras --no-cpu-baseline"
run_pmc () {   # name, counters, bench args
  timeout 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -o b -- python bench.py $3 $COMMON > $OUT/$1.log 2>&1
}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- python bench.py --steps 64 --warmup 32 --no-extras --no-cpu-baseline > $OUT/stats.log 2>&1
for CFG in "head_s32:--steps 64" "head_s20:--steps 20" "int_s32:--steps 64 --view interior"; do
  NAME=${CFG%%:*}; ARGS=${CFG#*:}
  run_pmc ${NAME}_fetch "FETCH_SIZE" "$ARGS"
  run_pmc ${NAME}_write "WRITE_SIZE" "$ARGS"
  run_pmc ${NAME}_l2 "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "$ARGS"
done
run_pmc head_s32_sq "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" "--steps 64"
python tools/prof_r02_summarize.py $OUT
