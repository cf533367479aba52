#!/bin/bash
# round 5, GPU call n: per-kernel times of the bench's timed region with the primary list tile-major (0) and pixel-major (8)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05n; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for pm in 0 8; do
  ( cd /tmp && IDKPT_GEN_PIXEL_MAJOR=$pm timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pm$pm -o b -- python $GRAFT_REPO_ROOT/bench.py --no-pmc --no-extras --no-cpu-baseline > $OUT/pm$pm.log 2>&1 )
  f=$(find $OUT/pm$pm -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/pm${pm}_kernel_stats.csv
done
for pm in 0 8; do echo "pm $pm"; head -12 $OUT/pm${pm}_kernel_stats.csv | cut -c1-60,200-400; done
