#!/bin/bash
# round 5, GPU call c: after the prune / split of the translation unit — smoke, the whole GPU suite, 300 fuzz seeds (now also drawing the wide option), random API sequences
TAG=r05c
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -3 ) > $OUT/smoke.log
( timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12 ) > $OUT/gpu_suite.log
( timeout 1200 python tools/fuzz_parity.py 300 30000 2>&1 | grep -v ": OK" | tail -8 ) > $OUT/fuzz_300.log
cat $OUT/smoke.log; tail -6 $OUT/gpu_suite.log; cat $OUT/fuzz_300.log
