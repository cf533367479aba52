#!/bin/bash
# round 5, GPU call w: rocprofv3 kernel stats of the two bench commands on the final tree
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05w; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 50 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -o b -- python bench.py --no-pmc --no-extras --no-cpu-baseline > $OUT/stats_default.log 2>&1
timeout 45 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_driver -o b -- python bench.py --steps 20 --warmup 5 --no-pmc --no-extras --no-cpu-baseline > $OUT/stats_driver.log 2>&1
for k in driver default; do f=$(find $OUT/stats_$k -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r05_${k}_kernel_stats.csv; done
rm -rf $OUT/stats_default/*trace.csv $OUT/stats_driver/*trace.csv
head -4 $OUT/r05_default_kernel_stats.csv | cut -c1-150; grep -h "^{" $OUT/stats_default.log $OUT/stats_driver.log | cut -c1-120
