#!/bin/bash
# Round-4 result tables and judged profiles (profiles/r04_*): the default bench line, the driver's command, the secondary configurations,
# the N-GPU flows on the one-GPU box, the shard projection, and rocprofv3 --kernel-trace --stats of the driver's command, the default command, the atrium and the
# 3-BLAS scene (the live --pmc passes of roofline.traffic / roofline.pmc are run by bench.py itself).
TAG=${1:-r04tab}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 500 python bench.py > $OUT/bench_default.json 2> $OUT/err.log
timeout 500 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2>> $OUT/err.log
S="--no-extras --no-cpu-baseline --no-pmc --repeats 3"
timeout 200 python bench.py $S --depth 5 > $OUT/sec_d5.json 2>> $OUT/err.log
timeout 200 python bench.py $S --depth 9 > $OUT/sec_d9.json 2>> $OUT/err.log
timeout 200 python bench.py $S --depth 2 --sort 1 > $OUT/sec_d2_sort.json 2>> $OUT/err.log
timeout 200 python bench.py $S --tris 260000 --depth 5 > $OUT/sec_260k_d5.json 2>> $OUT/err.log
timeout 300 python bench.py $S --tris 4000000 --width 3840 --height 2160 --depth 9 --steps 32 --warmup 32 > $OUT/sec_4m_4k_d9.json 2>> $OUT/err.log
timeout 300 python bench.py --no-extras --no-cpu-baseline --repeats 3 --view interior --depth 2 > $OUT/sec_interior_d2.json 2>> $OUT/err.log
timeout 200 python bench.py $S --view interior --depth 5 > $OUT/sec_interior_d5.json 2>> $OUT/err.log
timeout 300 python bench.py --no-extras --no-cpu-baseline --repeats 3 --scene atrium --depth 2 > $OUT/sec_atrium_1m_d2.json 2>> $OUT/err.log
timeout 200 python bench.py $S --scene atrium --tris 262000 --depth 5 > $OUT/sec_atrium_262k_d5.json 2>> $OUT/err.log
timeout 200 python bench.py $S --gpus 2 > $OUT/group2_shared_gpu.json 2>> $OUT/err.log
timeout 200 python bench.py $S --gpus 2 --steps 20 --warmup 5 > $OUT/group2_shared_gpu_driver_cmd.json 2>> $OUT/err.log
IDKPT_BENCH_ONE_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 2 --steps 20 --warmup 5 $S > $OUT/ranks2_one_device_gloo.json 2>> $OUT/err.log
( SHARD_MODS=1,2,4,8 SHARD_BANDS=8 timeout 600 python tools/shard_small_batch.py 8 20 2>&1 | tail -6 ) > $OUT/shard_small_batch.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_driver -o b -- python bench.py --steps 20 --warmup 5 --no-pmc --no-extras --no-cpu-baseline > $OUT/stats_driver.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -o b -- python bench.py --no-pmc --no-extras --no-cpu-baseline > $OUT/stats_default.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_atrium -o b -- python bench.py --no-pmc --no-extras --no-cpu-baseline --scene atrium > $OUT/stats_atrium.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_multi -o b -- python tools/bench_multi.py 1000000 3 headline > $OUT/stats_multi.log 2>&1
mkdir -p $OUT/summary
for k in driver default atrium multi; do f=$(find $OUT/stats_$k -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/summary/r04_${k}_kernel_stats.csv; done
f=$(find $OUT/stats_driver -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python - "$f" > $OUT/summary/r04_bench_trace_launches.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_trace2" in r["Kernel_Name"] or "k_trace_fused" in r["Kernel_Name"]]
print("kernel,start_ns,duration_us")
for r in rows:
    print(f"\"{r['Kernel_Name'][:60]}\",{r['Start_Timestamp']},{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.1f}")
PY
for f in $OUT/*.json; do echo "$(basename $f): $(python -c "import json,sys; d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step', d['config']['rays_per_step'], 'rays/step', 'L2 frac', d['roofline']['frac'], 'gather frac', (d['roofline'].get('gather_measured') or {}).get('frac'), 'n_gpus', d['n_gpus'], d['scaling'])" 2>/dev/null)"; done
cat $OUT/shard_small_batch.txt; tail -3 $OUT/err.log
