#!/bin/bash
# round 5, GPU call o (after the pixel-major primary list): the judged artefacts on the (near-)final tree — selftests of both N > 1 hosts, smoke(), the whole GPU suite, a fuzz soak, bench lines + rocprofv3 summaries
TAG=r05o
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
( timeout 400 python tools/scale_selftest.py --gpus 2 2>&1 | grep "selftest" | tail -14 ) > $OUT/selftest_group.txt
( IDKPT_BENCH_ONE_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 tools/scale_selftest.py 2>&1 | grep "selftest" | tail -8 ) > $OUT/selftest_ranks.txt
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -3 ) > $OUT/smoke.log
( timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -8 ) > $OUT/gpu_suite.log
( timeout 1800 python tools/fuzz_parity.py 800 90000 2>&1 | grep -v ": OK" | tail -6 ) > $OUT/fuzz_800.log
( FUZZ_BLASES=2,14 timeout 1800 python tools/fuzz_parity.py 500 95000 2>&1 | grep -v ": OK" | tail -6 ) > $OUT/fuzz_multi_500.log
( time timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/err_default.log ) 2> $OUT/bench_default.time
( time timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/err_driver.log ) 2> $OUT/bench_driver.time
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_driver -o b -- python bench.py --steps 20 --warmup 5 --no-pmc --no-extras --no-cpu-baseline > $OUT/stats_driver.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -o b -- python bench.py --no-pmc --no-extras --no-cpu-baseline > $OUT/stats_default.log 2>&1
timeout 200 python bench.py --gpus 2 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc > $OUT/group2_shared_gpu_driver_cmd.json 2>> $OUT/err.log
IDKPT_BENCH_ONE_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 2 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc > $OUT/ranks2_one_device_gloo.json 2>> $OUT/err.log
mkdir -p $OUT/summary
for k in driver default; do f=$(find $OUT/stats_$k -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/summary/r05_${k}_kernel_stats.csv; done
f=$(find $OUT/stats_driver -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python - "$f" > $OUT/summary/r05_bench_trace_launches.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_trace2" in r["Kernel_Name"] or "k_trace_fused" in r["Kernel_Name"] or "k_trace_wide" in r["Kernel_Name"]]
print("kernel,start_ns,duration_us")
for r in rows:
    print(f"\"{r['Kernel_Name'][:60]}\",{r['Start_Timestamp']},{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.1f}")
PY
cat $OUT/selftest_group.txt $OUT/selftest_ranks.txt $OUT/smoke.log; tail -3 $OUT/gpu_suite.log; cat $OUT/fuzz_800.log $OUT/fuzz_multi_500.log
for f in $OUT/bench_*.json $OUT/stats_driver.log $OUT/stats_default.log $OUT/group2*.json $OUT/ranks2*.json; do echo "$(basename $f): $(python -c "import json,sys; d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['repeats'], d['roofline']['avg_launch_us'], d['roofline']['frac'], (d.get('single_frame') or {}).get('mray_s'), (d.get('cpu_baseline') or {}).get('value'), d.get('transport'), d.get('rccl_ranks_seen'), ((d['config'].get('n_gpu') or {}).get('selftest')))" 2>/dev/null)"; done
