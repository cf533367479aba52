#!/bin/bash
# round 5, GPU call h: the instance loop through the library's own TLAS (kernels_trace_inst.hpp) — its tests, the instance / scene-update / bench-size tests around it, a fuzz run, first numbers
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05h; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_inst_tlas.py tests/test_gpu_instances.py -x -q -m gpu 2>&1 | tail -25 ) > $OUT/tests_inst.log
( timeout 900 python -m pytest tests/test_gpu_configscale.py -q -m gpu -k "atrium_per_mesh" 2>&1 | tail -15 ) > $OUT/tests_atrium.log
( timeout 900 python tools/fuzz_parity.py 400 60000 2>&1 | grep -v ": OK" | tail -8 ) > $OUT/fuzz_400.log
( timeout 900 python tools/bench_inst_tlas.py 2> $OUT/bench_inst_tlas.err | tail -1 ) > $OUT/bench_inst_tlas.json
tail -12 $OUT/tests_inst.log; tail -6 $OUT/tests_atrium.log; cat $OUT/fuzz_400.log; cat $OUT/bench_inst_tlas.json; tail -3 $OUT/bench_inst_tlas.err
