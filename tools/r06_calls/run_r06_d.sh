#!/bin/bash
# call d: texture tests (sampler state, 8-bit formats, idkptUpdateTexture), the HIP path against the new glref fixtures, counters of the packet walk beside k_trace2
# (interior + atrium, one primary launch of 16 samples), the bench line with its new blocks (driver's command)
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06d; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_textures.py tests/test_gpu_glref.py tests/test_gpu_packet.py tests/test_gpu_boundary.py tests/test_gpu_multi.py -x -q > $O/tests.log 2>&1; echo "rc $?" >> $O/tests.log
tail -5 $O/tests.log
for V in interior atrium; do
for P in 0 2; do
  i=0
  for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
             "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAVES" \
             "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "FETCH_SIZE" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_DATA_READ_REQ" "TA_BUSY_avr SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    IDKPT_PACKET=$P VIEW=$V timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/${V}_p${P}_$i -o c -- python tools/profile_frame.py 1000000 1 32 16 > $O/${V}_p${P}_$i.log 2>&1
  done
done
done
OUTD=$O python - <<'PY' > $O/pmc_summary.txt
import csv, glob, collections, os
for v in ("interior", "atrium"):
  for p in (0, 2):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.environ['OUTD'] + '/%s_p%d_*/**/*counter_collection.csv' % (v, p), recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0]
            if 'k_trace' in k:
                out[k[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    print("== view", v, "packet =", p)
    for k, d in out.items():
        print(k)
        for c, vals in sorted(d.items()):
            big = sorted(vals)[-2:]                         # the timed batches' launches (16 samples each) are the largest dispatches
            print("   %-36s n=%3d  mean of the 2 largest %.5g" % (c, len(vals), sum(big) / len(big)))
PY
cat $O/pmc_summary.txt | head -80
( time timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err ) 2>&1 | tail -3
tail -c 3000 $O/bench_driver_cmd.json
