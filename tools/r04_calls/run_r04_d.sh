#!/bin/bash
# round 4, GPU call D: baseline of the multi-BLAS modes; kernel timelines of a frame traced alone and of one rank's share of an 8-GPU run
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r04d
( timeout 900 python tools/bench_multi.py 1000000 3 headline > gpurun_out/r04d/multi_headline.json 2> gpurun_out/r04d/multi_headline.txt )
( timeout 900 python tools/bench_multi.py 1000000 3 interior > gpurun_out/r04d/multi_interior.json 2> gpurun_out/r04d/multi_interior.txt )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04d/sf -o t -- python $GRAFT_REPO_ROOT/tools/single_frame_profile.py 50 headline > $GRAFT_REPO_ROOT/gpurun_out/r04d/sf.log 2>&1 )
( cd /tmp && SHARD_MODS=8 SHARD_BANDS=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04d/shard8 -o t -- python $GRAFT_REPO_ROOT/tools/shard_small_batch.py 8 20 > $GRAFT_REPO_ROOT/gpurun_out/r04d/shard8.log 2>&1 )
for d in sf shard8; do F=$(find gpurun_out/r04d/$d -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" gpurun_out/r04d/${d}_kernel_stats.csv; rm -rf gpurun_out/r04d/$d; done
tail -4 gpurun_out/r04d/multi_headline.txt gpurun_out/r04d/multi_interior.txt; tail -2 gpurun_out/r04d/sf.log gpurun_out/r04d/shard8.log
for d in sf shard8; do echo "== $d"; python - "$d" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(f"gpurun_out/r04d/{sys.argv[1]}_kernel_stats.csv")))
for r in rows[:16]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} total {float(r['TotalDurationNs'])/1e6:9.3f} ms avg {float(r['AverageNs'])/1e3:9.2f} us {r['Percentage']:>6s} %")
PY
done
