#!/bin/bash
# round 5, GPU call d: RCCL transport self test + its tests, bench.py's own tests, and the default bench line with the new blocks (wide_nodes, multi_blas.atrium_per_mesh)
TAG=r05d
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_transport.py tests/test_gpu_bench.py tests/test_gpu_multi.py -x -q 2>&1 | tail -15 ) > $OUT/tests.log
( time timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
cat $OUT/tests.log; tail -3 $OUT/bench_default.time
python - <<PY
import json
try:
    d = json.loads([l for l in open("$OUT/bench_default.json") if l.startswith("{")][-1])
    print("bench", d["value"], d["ms_per_step"], d["traversed_mray_s"], d["roofline"]["frac"], d["transport"])
    for k in ("single_frame", "interior", "atrium", "multi_blas", "animated", "wide_nodes", "queries", "cpu_baseline"):
        print(k, json.dumps(d.get(k))[:1500])
except Exception as e:
    print("bench line unreadable:", e); print(open("$OUT/bench_default.err").read()[-2000:])
PY
