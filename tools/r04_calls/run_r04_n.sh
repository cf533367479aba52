#!/bin/bash
# round 4, GPU call N: pooled leaves in the instance / TLAS / scene-version kernels (parity + numbers); the N > 1 bench paths on this one GPU; PMC of a lone frame; pooled phase profile
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r04n
( timeout 900 python -m pytest tests/test_gpu_instances.py tests/test_gpu_versions.py tests/test_gpu_scene_updates.py tests/test_gpu_glref_full.py tests/test_gpu_glref.py tests/test_gpu_multi.py tests/test_gpu_configscale.py -q -m gpu --maxfail=8 2>&1 | tail -8 ) > gpurun_out/r04n/tests.log
( timeout 600 python tools/fuzz_parity.py 120 10300 2>&1 | grep -v ": OK" | tail -10 ) > gpurun_out/r04n/fuzz.log
( timeout 600 python tools/bench_multi.py 1000000 3 headline > gpurun_out/r04n/multi_headline.json 2> gpurun_out/r04n/multi_headline.txt )
( timeout 600 python tools/bench_multi.py 1000000 3 interior > gpurun_out/r04n/multi_interior.json 2> gpurun_out/r04n/multi_interior.txt )
( timeout 600 python tools/bench_animated.py 1000000 64 1,8,32 2>&1 | tail -5 ) > gpurun_out/r04n/animated.txt
( timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc > gpurun_out/r04n/bench_group2.json 2> gpurun_out/r04n/bench_group2.err )
( IDKPT_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r04n/bench_ranks2.json 2> gpurun_out/r04n/bench_ranks2.err )
( IDKPT_LIB_PATH=$PWD/idkengine_amd/libidkpt_dev.so PHASE_VARIANT=116 timeout 600 python tools/phase_profile.py 2>&1 | tail -8 ) > gpurun_out/r04n/phase_profile_pooled.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04n/sfpmc -o c -- python $GRAFT_REPO_ROOT/tools/single_frame_profile.py 30 headline > $GRAFT_REPO_ROOT/gpurun_out/r04n/sfpmc.log 2>&1 )
python - <<'PY' > gpurun_out/r04n/single_frame_pmc.txt 2>&1
import csv, glob, collections
f = glob.glob("gpurun_out/r04n/sfpmc/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "k_trace2" in k:
        acc["primary" if "<true" in k.replace(" ", "") and "k_trace2<true" in k.replace(" ", "") or "k_trace2s<true" in k.replace(" ", "") else "bounce"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kind, c in acc.items():
    m = {n: sum(v[5:]) / max(1, len(v[5:])) for n, v in c.items()}
    h, mi, rq, ac = m.get("TCC_HIT_sum", 0), m.get("TCC_MISS_sum", 0), m.get("TCP_TCC_READ_REQ_sum", 0), m.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0)
    print(f"one frame at a time, {kind} launch (mean of the timed launches): L2 hit rate {h / max(1, h + mi):.3f}, L1 hit rate {1 - rq / max(1, ac):.3f}, L1 misses {rq * 64 / 1e6:.1f} MB, L2 misses {mi * 128 / 1e6:.1f} MB")
PY
rm -rf gpurun_out/r04n/sfpmc
tail -n 3 gpurun_out/r04n/tests.log; cat gpurun_out/r04n/fuzz.log gpurun_out/r04n/multi_headline.txt gpurun_out/r04n/multi_interior.txt gpurun_out/r04n/animated.txt; tail -n 2 gpurun_out/r04n/bench_group2.err gpurun_out/r04n/bench_ranks2.err; head -c 600 gpurun_out/r04n/bench_group2.json; echo; tail -n 1 gpurun_out/r04n/bench_ranks2.json | head -c 600; echo; cat gpurun_out/r04n/phase_profile_pooled.txt gpurun_out/r04n/single_frame_pmc.txt
