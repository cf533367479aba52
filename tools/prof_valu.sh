#!/bin/bash
# SQ issue-side counters of the batched headline frame: is the traversal kernel VALU-issue bound?
TAG=${1:-valu}; BATCH=${2:-32}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU" \
           "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -o c -- python tools/profile_frame.py 1000000 2 $BATCH $BATCH > $OUT/p$i.log 2>&1
done
TAGX=$TAG python - <<'PY'
import csv, glob, collections, os
out = collections.defaultdict(dict)
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/%s/p*/c_counter_collection.csv' % os.environ['TAGX'])):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:40]
        if 'k_trace2' in k or 'k_shade' in k or 'k_gen' in k:
            out[k].setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
for k, d in out.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s n=%d last=%.4g" % (c, len(v), v[-1]))
PY
