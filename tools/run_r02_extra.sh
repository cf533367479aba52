#!/bin/bash
# Round-2 secondary numbers of DESIGN.md §9 that tools/run_r02_tables.sh does not cover (atrium lines with their roofline block, samples in flight, small frames)
TAG=${1:-r02x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
S="--no-extras --no-cpu-baseline --repeats 3"
timeout 200 python bench.py $S --scene atrium --depth 2 > $OUT/atrium_1m_d2.json 2> $OUT/err.log
timeout 200 python bench.py $S --scene atrium --depth 5 > $OUT/atrium_1m_d5.json 2>> $OUT/err.log
timeout 200 python bench.py $S --scene atrium --tris 262000 --depth 5 > $OUT/atrium_262k_d5.json 2>> $OUT/err.log
for B in 64 128; do for D in 2 9; do timeout 200 python bench.py $S --depth $D --batch $B --steps 128 --warmup 128 > $OUT/batch${B}_d$D.json 2>> $OUT/err.log; done; done
timeout 200 python bench.py $S --depth 9 --steps 128 --warmup 128 > $OUT/batch32_d9.json 2>> $OUT/err.log
timeout 200 python bench.py $S --gpus 2 --depth 5 > $OUT/group2_d5.json 2>> $OUT/err.log
timeout 200 python bench.py $S --gpus 2 --depth 9 > $OUT/group2_d9.json 2>> $OUT/err.log
timeout 200 python tools/small_frame_latency.py > $OUT/small_frames.txt 2>> $OUT/err.log
for f in $OUT/*.json; do echo "$(basename $f): $(python -c "import json,sys; d=json.load(open('$f')); r=d['roofline']; print(d['value'], d['unit'], d['ms_per_step'], 'ms/step', d['config']['rays_per_step'], 'rays/step', 'frac', r['frac'], r['bound'], 'gather frac', r['frac_of_gather_ceiling'])" 2>/dev/null)"; done
cat $OUT/small_frames.txt | tail -8
