#!/bin/bash
# call m (R06_TAG=r06m) / call r (R06_TAG=r06r, the final tree): verification of the round's tree: the whole GPU suite, smoke, both bench lines (driver's command and default), rocprofv3 kernel stats of both, fuzz
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${R06_TAG:-r06m}; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; echo "rc $?" >> $O/suite.log
tail -3 $O/suite.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc $?" >> $O/smoke.log; tail -2 $O/smoke.log
( time timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err ) 2>&1 | tail -3
( time timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_driver -o b -- python bench.py --steps 20 --warmup 5 --no-pmc --no-extras --no-cpu-baseline > $O/stats_driver.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -o b -- python bench.py --no-pmc --no-extras --no-cpu-baseline > $O/stats_default.log 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_atrium87 -o atrium87 -- python $GRAFT_REPO_ROOT/tools/bench_inst_tlas.py --profile-atrium 2>&1 | tail -3 ) > $O/prof_atrium87.log
timeout 1500 python tools/fuzz_parity.py 500 100000 > $O/fuzz_500.log 2>&1; echo "rc $?" >> $O/fuzz_500.log
tail -2 $O/fuzz_500.log
python - <<'PY'
import json, os
tag = os.environ.get("R06_TAG", "r06m")
for f in ("bench_driver_cmd", "bench_default"):
    try:
        d = json.loads(open("gpurun_out/%s/%s.json" % (tag, f)).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("value_one_frame"), d.get("value_eager"), d.get("value_traversed"), d["roofline"]["frac"])
    except Exception as e:
        print(f, "unreadable", e)
PY
