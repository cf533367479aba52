#!/bin/bash
# Judged artefacts of the final round-1 build: rocprofv3 summaries of the bench command (kernel stats + HBM-side PMC passes) and a
# FETCH_SIZE calibration on this kernel's own access pattern (independent random 64-B block gathers over 1 GiB: every block misses L2).
TAG=${1:-r01f}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
[ -x tools/ubench_lines.bin ] || hipcc --offload-arch=gfx950 -O3 tools/ubench_lines.hip -o tools/ubench_lines.bin
CMD="python bench.py --steps 64 --warmup 32 --no-cpu-baseline"
timeout 200 $CMD > $OUT/bench_plain.json 2> $OUT/bench_plain.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- $CMD > $OUT/bench_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o b -- $CMD > $OUT/bench_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o b -- $CMD > $OUT/bench_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/pmc_l2 -o b -- $CMD > $OUT/bench_l2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib -o c -- ./tools/ubench_lines.bin 24 > $OUT/calib.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc TCC_MISS_sum TCC_HIT_sum --output-format csv -d $OUT/calib2 -o c -- ./tools/ubench_lines.bin 24 > $OUT/calib2.log 2>&1
grep -h "mode" $OUT/calib.log | head -12
grep -h "^{" $OUT/bench_plain.json $OUT/bench_stats.log | cut -c1-160
