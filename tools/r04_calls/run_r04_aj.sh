#!/bin/bash
# round 4, GPU call AJ: counters behind three measured negatives / positives of the round — plain vs quad records vs two parked leaves vs pooled leaves, 32 samples in flight
# (bench.py --steps 64 --warmup 32: two full batches per repetition): L1 / L2 requests and instruction counts per traversal launch, beside the launch time
TAG=r04aj
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
COMMON="--steps 64 --warmup 32 --repeats 2 --no-extras --no-cpu-baseline --no-pmc"
for CFG in "plain:IDKPT_LEAF_POOL=0" "pooled:IDKPT_LEAF_POOL=-1" "quad:IDKPT_QUAD=2" "park:IDKPT_PARK=7,IDKPT_LEAF_POOL=0"; do
  NAME=${CFG%%:*}; ENVS=${CFG#*:}
  ( export ${ENVS//,/ };
    timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/${NAME}_mem -o b -- python bench.py $COMMON > $OUT/${NAME}_mem.log 2>&1
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $OUT/${NAME}_sq -o b -- python bench.py $COMMON > $OUT/${NAME}_sq.log 2>&1 )
done
python - <<'PY' > $OUT/summary.txt
import csv, glob, os, collections, json
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04aj"
for name in ("plain", "pooled", "quad", "park"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for kind in ("mem", "sq"):
        for f in glob.glob(f"{out}/{name}_{kind}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "k_trace2" not in k: continue
                agg["primary" if "<true" in k or "ILb1E" in k else "bounce"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    line = None
    try: line = json.loads([l for l in open(f"{out}/{name}_mem.log") if l.startswith("{")][-1])
    except Exception: pass
    print(f"== {name}: " + (f"{line['value']} Mray/s under the counter pass, avg launch {line['roofline']['avg_launch_us']} us" if line else "no line"))
    for which, d in agg.items():
        big = {c: [v for v in vs] for c, vs in d.items()}
        def top(c):   # the launches of 32 samples are the large ones: mean of the upper half
            vs = sorted(big.get(c, [0.0])); vs = vs[len(vs) // 2:]; return sum(vs) / max(1, len(vs))
        acc, l1m, l2h, l2m = top("TCP_TOTAL_CACHE_ACCESSES_sum"), top("TCP_TCC_READ_REQ_sum"), top("TCC_HIT_sum"), top("TCC_MISS_sum")
        print(f"   {which:8s} L1 accesses {acc/1e6:9.1f} M  L1 misses -> L2 {l1m/1e6:9.1f} M  L2 hit {l2h/max(1.0,l2h+l2m):.3f}  L2 misses {l2m/1e6:8.1f} M | VALU {top('SQ_INSTS_VALU')/1e6:9.1f} M  VMEM rd {top('SQ_INSTS_VMEM_RD')/1e6:8.1f} M  LDS {top('SQ_INSTS_LDS')/1e6:8.1f} M  SALU {top('SQ_INSTS_SALU')/1e6:9.1f} M wave-instructions per launch")
PY
cat $OUT/summary.txt
