#!/bin/bash
# round 5, GPU call k: the sieved exact loop (k_trace_inst<P, EXACT>) behind the own-TLAS walk and as a main kernel (option inst_sieve): tests, the table, fuzz
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05k; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_inst_tlas.py tests/test_gpu_instances.py tests/test_gpu_scene_updates.py tests/test_gpu_configscale.py -x -q -m gpu 2>&1 | tail -25 ) > $OUT/tests_inst.log
( timeout 1500 python tools/bench_inst_tlas.py 2> $OUT/bench_inst_tlas.err | tail -1 ) > $OUT/bench_inst_tlas.json
( timeout 900 python tools/fuzz_parity.py 400 62000 2>&1 | grep -v ": OK" | tail -6 ) > $OUT/fuzz_400.log
tail -6 $OUT/tests_inst.log; cat $OUT/fuzz_400.log; tail -2 $OUT/bench_inst_tlas.err | cut -c1-300
