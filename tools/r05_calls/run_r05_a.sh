#!/bin/bash
# round 5, GPU call a: first run of the wide-node walk — its tests, the A/B against k_trace2 on the three bench views, the whole GPU suite, the phase profile, one bench line
TAG=r05a
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_wide.py -x -q 2>&1 | tail -25 ) > $OUT/test_wide.log
( timeout 900 python tools/wide_ab.py 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -12 ) > $OUT/wide_ab.log; cp gpurun_out/wide_ab.json $OUT/ 2>/dev/null
( timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 ) > $OUT/gpu_suite.log
( IDKPT_LIB_PATH=$PWD/idkengine_amd/libidkpt_dev.so PHASE_VARIANT=213 timeout 600 python tools/phase_profile.py 2>&1 | tail -8 ) > $OUT/phase_wide.log
( timeout 500 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err ); tail -3 $OUT/bench_default.err > $OUT/bench_default.errtail
cat $OUT/test_wide.log; cat $OUT/wide_ab.log; tail -4 $OUT/gpu_suite.log; cat $OUT/phase_wide.log
python - <<PY
import json
try:
    d = json.loads([l for l in open("$OUT/bench_default.json") if l.startswith("{")][-1])
    print("bench", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], {k: (d.get(k) or {}).get("mray_s") for k in ("single_frame", "interior")})
except Exception as e:
    print("bench line unreadable:", e)
PY
