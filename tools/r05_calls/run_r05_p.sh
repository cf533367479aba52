#!/bin/bash
# round 5, GPU call p: after switching the pixel-major list off for batches of different frames — the default bench line again (animated block), version / batching / sample tests
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05p; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_versions.py tests/test_gpu_batching.py tests/test_gpu_samples.py tests/test_gpu_scene_updates.py -x -q -m gpu 2>&1 | tail -4 ) > $OUT/tests.log
( time timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/err_default.log ) 2> $OUT/bench_default.time
( time timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/err_driver.log ) 2> $OUT/bench_driver.time
tail -2 $OUT/tests.log; python - <<'PY'
import json,os
for n in ("bench_default","bench_driver_cmd"):
    d=json.loads([l for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05p/"+n+".json") if l.startswith("{")][-1])
    print(n, d["value"], d["single_frame"]["mray_s"], [d["animated"][k]["mray_s"] for k in ("frames_in_flight_1","frames_in_flight_8","frames_in_flight_32")])
PY
