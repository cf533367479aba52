#!/bin/bash
# round 5, GPU call i: own-TLAS walk with the overlap rule — tests again, where the rule's threshold lies (clusters of growing overlap), kernel trace of the atrium as 87 BLASes
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05i; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_inst_tlas.py tests/test_gpu_instances.py tests/test_gpu_scene_updates.py -x -q -m gpu 2>&1 | tail -25 ) > $OUT/tests_inst.log
( timeout 900 python -m pytest tests/test_gpu_configscale.py -q -m gpu -k "atrium_per_mesh" 2>&1 | tail -15 ) > $OUT/tests_atrium.log
( timeout 1200 python tools/bench_inst_tlas.py 2> $OUT/bench_inst_tlas.err | tail -1 ) > $OUT/bench_inst_tlas.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_atrium87 -o atrium87 -- python $GRAFT_REPO_ROOT/tools/bench_inst_tlas.py --profile-atrium 2>&1 | tail -3 ) > $OUT/prof_atrium87.log
find $OUT/prof_atrium87 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/atrium87_kernel_stats.csv
tail -8 $OUT/tests_inst.log; tail -4 $OUT/tests_atrium.log; cat $OUT/bench_inst_tlas.json; tail -3 $OUT/prof_atrium87.log; head -12 $OUT/atrium87_kernel_stats.csv | cut -c1-200
