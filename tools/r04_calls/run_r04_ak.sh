#!/bin/bash
# round 4, GPU call AK: counters of k_trace2 MODE 1 (instance loop) / MODE 2 (TLAS) / MODE 0 on the 3-BLAS soup (tools/bench_multi.py): profiles/r04_multi_pmc.txt
TAG=r04ak
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/mem -o b -- python tools/bench_multi.py 1000000 3 headline > $OUT/mem.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $OUT/sq -o b -- python tools/bench_multi.py 1000000 3 headline > $OUT/sq.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq2 -o b -- python tools/bench_multi.py 1000000 3 headline > $OUT/sq2.log 2>&1
python - <<'PY' > $OUT/summary.txt
import csv, glob, os, collections, re
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04ak"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for kind in ("mem", "sq", "sq2"):
    for f in glob.glob(f"{out}/{kind}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            m = re.match(r"void k_trace2<(true|false), (true|false), 32, 1, false, 24, (\d), (\d+), false", k)
            if not m or m.group(2) == "true": continue                       # (the counting build's launches are one sample each)
            agg[(int(m.group(3)), m.group(1) == "true")][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = {0: "MODE 0 (one BLAS)", 1: "MODE 1 (instance loop)", 2: "MODE 2 (TLAS)"}
print("k_trace2 on soup-1M in 3 BLASes / in one BLAS, headline view, 32 samples in flight: per-launch means of the upper half of the launches (the 32-sample ones)")
for (mode, prim), d in sorted(agg.items()):
    def top(c):
        vs = sorted(d.get(c, [0.0])); vs = vs[len(vs) // 2:]; return sum(vs) / max(1, len(vs))
    l2h, l2m = top("TCC_HIT_sum"), top("TCC_MISS_sum")
    lanes = top("SQ_THREAD_CYCLES_VALU") / max(1.0, 64.0 * top("SQ_ACTIVE_INST_VALU"))
    print(f"{names[mode]:24s} {'primary' if prim else 'bounce ':8s}: L1 accesses {top('TCP_TOTAL_CACHE_ACCESSES_sum')/1e6:8.1f} M  L1 misses {top('TCP_TCC_READ_REQ_sum')/1e6:7.1f} M  L2 hit {l2h/max(1.0,l2h+l2m):.3f}  L2 misses {l2m/1e6:6.1f} M | VALU {top('SQ_INSTS_VALU')/1e6:8.1f} M  VMEM rd {top('SQ_INSTS_VMEM_RD')/1e6:6.1f} M  LDS {top('SQ_INSTS_LDS')/1e6:6.1f} M  SALU {top('SQ_INSTS_SALU')/1e6:8.1f} M | VALU lane utilisation {lanes:.3f}  waiting {top('SQ_WAIT_INST_ANY')/max(1.0,top('SQ_WAVE_CYCLES')):.3f} of wave cycles")
PY
cat $OUT/summary.txt; grep -h "per-visit" $OUT/mem.log | cut -c1-200
