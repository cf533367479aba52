#!/bin/bash
# round 4, GPU call E: what bounds a frame traced alone?  instrumented per-step cycles at the current grid rule, the floor with few waves per CU, kernel timelines
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r04e
( IDKPT_LIB_PATH=$PWD/idkengine_amd/libidkpt_dev.so timeout 600 python tools/phase_profile.py 2>&1 | tail -8 ) > gpurun_out/r04e/phase_profile.txt
( SWEEP_TAG=r04e SWEEP_OPT=TRACE_WAVES:24,16,12,8,6,4,2,1 SWEEP_BATCHES=1 SWEEP_DEPTHS=2 IDKPT_GRID_RAYS_X4=0 timeout 600 python tools/sweep_r03.py headline 2>&1 | tail -10 ) > gpurun_out/r04e/waves_floor.txt
( SWEEP_TAG=r04e2 SWEEP_OPT=TRACE_WAVES:24,16,12,8,6,4,2,1 SWEEP_BATCHES=1 SWEEP_DEPTHS=2 timeout 600 python tools/sweep_r03.py headline 2>&1 | tail -10 ) > gpurun_out/r04e/waves_rule.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04e/sf -o t -- python $GRAFT_REPO_ROOT/tools/single_frame_profile.py 50 headline > $GRAFT_REPO_ROOT/gpurun_out/r04e/sf.log 2>&1 )
( cd /tmp && SHARD_MODS=8 SHARD_BANDS=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04e/shard8 -o t -- python $GRAFT_REPO_ROOT/tools/shard_small_batch.py 8 20 > $GRAFT_REPO_ROOT/gpurun_out/r04e/shard8.log 2>&1 )
for d in sf shard8; do F=$(find gpurun_out/r04e/$d -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" gpurun_out/r04e/${d}_kernel_stats.csv; rm -rf gpurun_out/r04e/$d; done
cat gpurun_out/r04e/phase_profile.txt gpurun_out/r04e/waves_floor.txt gpurun_out/r04e/waves_rule.txt; tail -n 2 gpurun_out/r04e/sf.log; tail -n 2 gpurun_out/r04e/shard8.log
