#!/bin/bash
# round 4, GPU call W: the two bench lines and the judged rocprofv3 summary with the final bench.py (adaptive repeats)
TAG=r04w
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
( time timeout 500 python bench.py > $OUT/bench_default.json 2> $OUT/err_default.log ) 2> $OUT/bench_default.time
( time timeout 500 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/err_driver.log ) 2> $OUT/bench_driver.time
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_driver -o b -- python bench.py --steps 20 --warmup 5 --no-pmc --no-extras --no-cpu-baseline > $OUT/stats_driver.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -o b -- python bench.py --no-pmc --no-extras --no-cpu-baseline > $OUT/stats_default.log 2>&1
mkdir -p $OUT/summary
for k in driver default; do f=$(find $OUT/stats_$k -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/summary/r04_${k}_kernel_stats.csv; done
f=$(find $OUT/stats_driver -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python - "$f" > $OUT/summary/r04_bench_trace_launches.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_trace2" in r["Kernel_Name"] or "k_trace_fused" in r["Kernel_Name"]]
print("kernel,start_ns,duration_us")
for r in rows:
    print(f"\"{r['Kernel_Name'][:60]}\",{r['Start_Timestamp']},{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.1f}")
PY
for f in $OUT/bench_*.json $OUT/stats_driver.log $OUT/stats_default.log; do echo "$(basename $f): $(python -c "import json,sys; d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['repeats'], d['repeat_ms'], d['roofline']['avg_launch_us'], d['roofline']['frac'])" 2>/dev/null)"; done
cat $OUT/bench_default.time $OUT/bench_driver.time | grep real
