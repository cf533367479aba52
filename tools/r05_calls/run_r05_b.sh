#!/bin/bash
# round 5, GPU call b: what bounds the wide-node walk — SQ / TCP / TCC counters of k_trace_wide beside k_trace2 on the interior view (every pixel traverses), and occupancy sweeps
TAG=r05b
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for W in 0 1; do
  i=0
  for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
             "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES" \
             "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "FETCH_SIZE" "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
    i=$((i+1))
    IDKPT_WIDE=$W VIEW=interior timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/w${W}p$i -o c -- python tools/profile_frame.py 1000000 2 16 8 > $OUT/w${W}p$i.log 2>&1
  done
done
OUTD=$OUT python - <<'PY' > $OUT/pmc_summary.txt
import csv, glob, collections, os
for w in (0, 1):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.environ['OUTD'] + '/w%dp*/**/*counter_collection.csv' % w, recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0]
            if 'k_trace' in k:
                out[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    print("== wide =", w)
    for k, d in out.items():
        print(k)
        for c, v in sorted(d.items()):
            big = sorted(v)[-4:]                         # the timed batches' launches (8 samples each) are the largest dispatches
            print("   %-36s n=%3d  mean of the 4 largest %.5g" % (c, len(v), sum(big) / len(big)))
PY
for VIEW in interior headline; do
  for W in 0 1; do for TW in 8 12 16 20 24; do
    echo "view $VIEW wide $W trace_waves $TW: $(IDKPT_WIDE=$W IDKPT_TRACE_WAVES=$TW VIEW=$VIEW timeout 120 python tools/profile_frame.py 1000000 2 64 32 2>/dev/null | tail -1)"
  done; done
done > $OUT/occupancy_sweep.txt
cat $OUT/pmc_summary.txt; cat $OUT/occupancy_sweep.txt
