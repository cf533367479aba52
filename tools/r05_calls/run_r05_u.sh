#!/bin/bash
# round 5, GPU call u: the final tree (first bounce pixel-major on sparse views) — the whole GPU suite, the batch tests with both lists forced, fuzz, smoke, the two bench lines
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05u; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
( timeout 1200 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 ) > $OUT/gpu_suite.log
( IDKPT_GEN_PIXEL_MAJOR=2 IDKPT_BOUNCE_PIXEL_MAJOR=2 timeout 900 python -m pytest tests/test_gpu_batching.py tests/test_gpu_samples.py tests/test_gpu_versions.py tests/test_gpu_parity.py tests/test_gpu_inst_tlas.py tests/test_gpu_wide.py tests/test_gpu_multi.py tests/test_gpu_nocounters.py -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 ) > $OUT/tests_pm_forced.log
( timeout 600 python tools/fuzz_parity.py 300 130000 2>&1 | grep -v ": OK" | tail -3 ) > $OUT/fuzz_300.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2 ) > $OUT/smoke.log
( time timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/err_default.log ) 2> $OUT/bench_default.time
( time timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/err_driver.log ) 2> $OUT/bench_driver.time
cat $OUT/gpu_suite.log $OUT/tests_pm_forced.log $OUT/fuzz_300.log $OUT/smoke.log
python - <<'PY'
import json,os
for n in ("bench_default","bench_driver_cmd"):
    d=json.loads([l for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05u/"+n+".json") if l.startswith("{")][-1])
    print(n, d["value"], d["single_frame"]["mray_s"], d["roofline"]["frac"], d["interior"]["depth2"]["mray_s"], d["atrium"]["atrium_1000k_depth2"]["mray_s"], [d["animated"][k]["mray_s"] for k in ("frames_in_flight_1","frames_in_flight_8","frames_in_flight_32")], d["multi_blas"]["atrium_per_mesh"]["instance_loop"]["mray_s"])
PY
