#!/bin/bash
# call t: counters of the unified tree's kernels on the atrium as 87 BLASes (one batch of 32 samples + 8 single frames: tools/bench_inst_tlas.py --profile-atrium)
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06t; mkdir -p $O; cd /tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAVES" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "FETCH_SIZE" "TA_BUSY_avr"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/p$i -o c -- python $GRAFT_REPO_ROOT/tools/bench_inst_tlas.py --profile-atrium > $O/p$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
OUTD=$O python - <<'PY' > $O/pmc_summary.txt
import csv, glob, collections, os
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.environ['OUTD'] + '/p*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if 'k_trace' in k:
            out[k[:80]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in out.items():
    print(k)
    for c, vals in sorted(d.items()):
        big = sorted(vals)[-1:]                      # the one batched launch (32 samples) is the largest dispatch
        print("   %-36s n=%3d  largest %.5g" % (c, len(vals), big[0]))
PY
cat $O/pmc_summary.txt
