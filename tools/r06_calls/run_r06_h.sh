#!/bin/bash
# call h: the unified tree with its stack sized from the BLASes' RequiredStackSize: tests, bench_braid --quick (atrium + 3-part soup), 200 same-space fuzz seeds, the N = 8 shard projection (soup and atrium)
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06h; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_inst_tlas.py tests/test_gpu_instances.py -x -q > $O/tests.log 2>&1; echo "rc $?" >> $O/tests.log
tail -3 $O/tests.log
timeout 1500 python tools/bench_braid.py --quick > $O/bench_braid_quick.json 2> $O/bench_braid_quick.err; echo "rc $?"
FUZZ_SAME_SPACE=1 FUZZ_BLASES=2,14 timeout 900 python tools/fuzz_parity.py 200 70000 > $O/fuzz_same_space_200.log 2>&1; echo "rc $?" >> $O/fuzz_same_space_200.log
tail -2 $O/fuzz_same_space_200.log
SHARD_MODS=1,2,4,8 timeout 900 python tools/shard_small_batch.py 8 20 > $O/shard_small_batch_soup.txt 2>&1
SHARD_SCENE=atrium SHARD_MODS=1,2,4,8 timeout 900 python tools/shard_small_batch.py 8 20 > $O/shard_small_batch_atrium.txt 2>&1
cat $O/shard_small_batch_soup.txt $O/shard_small_batch_atrium.txt
