#!/bin/bash
# Round-2 result tables: the default bench line, the driver's command, and the secondary configurations.
TAG=${1:-r02t}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/err.log
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2>> $OUT/err.log
S="--no-extras --no-cpu-baseline --repeats 3"
timeout 200 python bench.py $S --depth 5 > $OUT/sec_d5.json 2>> $OUT/err.log
timeout 200 python bench.py $S --depth 5 --sort 1 > $OUT/sec_d5_sort.json 2>> $OUT/err.log
timeout 200 python bench.py $S --depth 9 > $OUT/sec_d9.json 2>> $OUT/err.log
timeout 200 python bench.py $S --depth 2 --sort 1 > $OUT/sec_d2_sort.json 2>> $OUT/err.log
timeout 200 python bench.py $S --tris 260000 --depth 5 > $OUT/sec_260k_d5.json 2>> $OUT/err.log
timeout 300 python bench.py $S --tris 4000000 --width 3840 --height 2160 --depth 9 --steps 32 --warmup 32 > $OUT/sec_4m_4k_d9.json 2>> $OUT/err.log
timeout 200 python bench.py $S --view interior --depth 2 > $OUT/sec_interior_d2.json 2>> $OUT/err.log
timeout 200 python bench.py $S --view interior --depth 5 > $OUT/sec_interior_d5.json 2>> $OUT/err.log
timeout 200 python bench.py $S --gpus 2 > $OUT/group2_shared_gpu.json 2>> $OUT/err.log
timeout 200 python tools/bench_animated.py > $OUT/animated.txt 2>> $OUT/err.log
for f in $OUT/*.json; do echo "$(basename $f): $(python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step', d['config']['rays_per_step'], 'rays/step', 'frac', d['roofline']['frac'])" 2>/dev/null)"; done
cat $OUT/animated.txt | tail -3
