#!/bin/bash
# call j: the packet walk with its hand-written decision / push / pop block: tests, forced-packet suites, fuzz, A/B, counters (interior + atrium)
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06j; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_packet.py tests/test_gpu_queries.py -x -q > $O/tests.log 2>&1; echo "rc $?" >> $O/tests.log
tail -3 $O/tests.log
IDKPT_PACKET=2 timeout 1500 python -m pytest tests/test_gpu_batching.py tests/test_gpu_samples.py tests/test_gpu_parity.py tests/test_gpu_defer.py tests/test_gpu_fullsize.py -x -q > $O/suite_packet_forced.log 2>&1; echo "rc $?" >> $O/suite_packet_forced.log
tail -3 $O/suite_packet_forced.log
FUZZ_BLASES=1,1 timeout 1500 python tools/fuzz_parity.py 400 80000 > $O/fuzz_one_blas_400.log 2>&1; echo "rc $?" >> $O/fuzz_one_blas_400.log
tail -2 $O/fuzz_one_blas_400.log
AB_MODES=0,1,0,1 timeout 900 python tools/ab_packet.py interior interior_d5 atrium headline > $O/ab_packet.log 2>&1; echo "rc $?" >> $O/ab_packet.log
tail -6 $O/ab_packet.log
for V in interior atrium; do
  i=0
  for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
             "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAVES"; do
    i=$((i+1))
    IDKPT_PACKET=2 VIEW=$V timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/${V}_p2_$i -o c -- python tools/profile_frame.py 1000000 1 32 16 > $O/${V}_p2_$i.log 2>&1
  done
done
OUTD=$O python - <<'PY' > $O/pmc_summary.txt
import csv, glob, collections, os
for v in ("interior", "atrium"):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.environ['OUTD'] + '/%s_p2_*/**/*counter_collection.csv' % v, recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0]
            if 'k_trace_packet' in k:
                out[k[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    print("== view", v, "packet = 2")
    for k, d in out.items():
        print(k)
        for c, vals in sorted(d.items()):
            big = sorted(vals)[-2:]
            print("   %-36s n=%3d  mean of the 2 largest %.5g" % (c, len(vals), sum(big) / len(big)))
PY
cat $O/pmc_summary.txt
