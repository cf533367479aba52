#!/bin/bash
# TA / TCP / SQ-VMEM counters of the batched headline frame
TAG=${1:-pm}; BATCH=${2:-8}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
i=0
for SET in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_BUSY_avr" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" "TD_TD_BUSY_sum TD_TC_STALL_sum" "TCC_BUSY_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -o c -- python tools/profile_frame.py 1000000 2 $BATCH $BATCH > $OUT/p$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections, os
out = collections.defaultdict(dict)
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/%s/p*/c_counter_collection.csv' % os.environ.get('TAGX','pm'))):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if 'k_trace2' in k:
            out[k].setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
for k, d in out.items():
    print(k[:44])
    for c, v in sorted(d.items()): print("   %-40s %.4g (n=%d)" % (c, max(v), len(v)))
PY
