#!/bin/bash
# round 5, GPU call j: instance records (DScene::instRec) under the instance loop, the TLAS walk and the own-TLAS walk — the whole GPU suite, the own-TLAS table again, the 3-BLAS bench scenes
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05j; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
( timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 ) > $OUT/gpu_suite.log
( timeout 1200 python tools/bench_inst_tlas.py 2> $OUT/bench_inst_tlas.err | tail -1 ) > $OUT/bench_inst_tlas.json
( timeout 600 python tools/bench_multi.py 1000000 3 headline 2>&1 | tail -3 ) > $OUT/bench_multi3_headline.txt
( timeout 600 python tools/bench_multi.py 1000000 3 interior 2>&1 | tail -3 ) > $OUT/bench_multi3_interior.txt
( timeout 900 python tools/fuzz_parity.py 300 61000 2>&1 | grep -v ": OK" | tail -6 ) > $OUT/fuzz_300.log
tail -4 $OUT/gpu_suite.log; cat $OUT/bench_inst_tlas.json | cut -c1-600; cat $OUT/bench_multi3_headline.txt $OUT/bench_multi3_interior.txt | cut -c1-1500; cat $OUT/fuzz_300.log
