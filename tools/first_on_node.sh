#!/bin/bash
# First run on a real multi-GPU node (the build box has one GPU: DESIGN.md 6 "what has run where").  In order, stopping at the first failure:
#   1. tools/scale_selftest.py, one process / one multi-device context (idkptCreate(N): peer matrix, RCCL transport, N members == 1 device bit for bit)
#   2. tools/scale_selftest.py, one process per GPU over RCCL (rank -> device, ranks seen, sharded frame == 1 device bit for bit)
#   3. bench.py --gpus N --steps 20 --warmup 5 as the driver launches it (one rank per GPU), then as ONE process driving one N-device context
# and prints, per bench line: value, n_gpus, transport, rccl_ranks_seen, the per-rank launch sizes and the frame check (bits_equal).
#   usage: tools/first_on_node.sh [N = visible GPUs] [port = 29541]
set -u
cd "$(dirname "$0")/.."
N=${1:-$(python -c 'import torch; print(torch.cuda.device_count())')}
PORT=${2:-29541}
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=${FIRST_ON_NODE_OUT:-gpurun_out/first_on_node}; mkdir -p "$OUT"
echo "== $N GPU(s); outputs under $OUT"
python tools/scale_selftest.py --gpus "$N" 2>&1 | tee "$OUT/selftest_context.log" | grep "^\[selftest\]" | tail -4
grep -q "^\[selftest\] PASS" "$OUT/selftest_context.log" || { echo "== FAIL: one-context self test"; exit 1; }
if [ "$N" -gt 1 ]; then
  python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" tools/scale_selftest.py 2>&1 | tee "$OUT/selftest_ranks.log" | grep "^\[selftest\]" | tail -4
  grep -q "^\[selftest\] PASS" "$OUT/selftest_ranks.log" || { echo "== FAIL: rank-per-GPU self test"; exit 1; }
fi
summarise() {   # the bench line's multi-GPU fields
python - "$1" <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]
d = json.loads(line); g = d["config"].get("n_gpu") or {}
print(f"   value {d['value']} {d['unit']}  n_gpus {d['n_gpus']}  ms_per_step {d['ms_per_step']}  scaling {d['scaling']}")
print(f"   transport {d.get('transport')}  rccl_ranks_seen {d.get('rccl_ranks_seen')}  ranks {d.get('ranks')}")
for k in ("devices_used", "members_share_gpus", "peer_access", "ranks", "rank0_launches", "selftest"):
    if k in g: print(f"   {k}: {json.dumps(g[k])[:400]}")
ok = isinstance(g.get("selftest"), dict) and g["selftest"].get("bits_equal") is True
print("   frame of the N-GPU run == one device, bit for bit:", ok)
sys.exit(0 if ok else 2)
PY
}
if [ "$N" -gt 1 ]; then
  echo "== bench.py, one rank per GPU (the driver's command)"
  python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((PORT + 1)) bench.py --gpus "$N" --steps 20 --warmup 5 > "$OUT/bench_ranks.json" 2> "$OUT/bench_ranks.err" || { tail -5 "$OUT/bench_ranks.err"; echo "== FAIL: bench.py under torch.distributed.run"; exit 1; }
  summarise "$OUT/bench_ranks.json" || { echo "== FAIL: the sharded frame differs from one device"; exit 1; }
fi
echo "== bench.py, one process, one $N-device context"
python bench.py --gpus "$N" --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_context.json" 2> "$OUT/bench_context.err" || { tail -5 "$OUT/bench_context.err"; echo "== FAIL: bench.py --gpus $N in one process"; exit 1; }
summarise "$OUT/bench_context.json" || { echo "== FAIL: the N-device context's frame differs from one device"; exit 1; }
echo "== done"
