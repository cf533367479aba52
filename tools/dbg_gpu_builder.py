import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from idkengine_amd import scenes as S
from idkengine_amd.bvh import NativeBuilder, GpuBuilder
from idkengine_amd.pathtracer import PathTracer
nb = NativeBuilder()
class Cap:
    def __init__(s): s.calls = []
    def build_blas(s, p, t, r): s.calls.append((np.array(p, np.float32), np.array(t), r)); return nb.build_blas(p, t, r)
    def __getattr__(s, k): return getattr(nb, k)
cap = Cap(); S.soup_scene(int(sys.argv[1]) if len(sys.argv) > 1 else 200000, cap, seed=8)
positions, tris, refit = cap.calls[0]
boxes, cn, co = nb.core_arrays(positions, tris, refit)
pt = PathTracer(8, 8); gb = GpuBuilder(pt)
gn, go = gb.core_on_gpu(boxes)
print("n", len(boxes), "levels", gb.last_levels, "core ms", gb.last_core_ms)
print("order equal", np.array_equal(go, co), "nodes equal", gn.tobytes() == cn.tobytes())
cv = cn.view(np.uint32).reshape(-1, 8); gv = gn.view(np.uint32).reshape(-1, 8)
bad = np.nonzero((cv != gv).any(axis=1))[0]
print("differing nodes", len(bad), bad[:10])
for i in bad[:4]:
    print(i, "cpu", cn[i], "\n   gpu", gn[i])
# find the parent of the first differing node
if len(bad):
    i = bad[0]
    par = [p for p in range(1, len(cn)) if cn[p]["TriCount"] == 0 and cn[p]["TriStartOrChild"] in (i, i - 1) and cn[p]["TriStartOrChild"] != 0][:1]
    print("parent", par, cn[par[0]] if par else None, gn[par[0]] if par else None)
# ---- root decision by brute force (float32 emulation)
f = np.float32
mn, mx = boxes[:, 0:3], boxes[:, 4:7]
def key(v):
    u = v.view(np.uint32).astype(np.uint64)
    s = (u >> 31) & 1
    return np.where(s == 1, (~u) & 0xffffffff, u | 0x80000000).astype(np.uint64)
n = len(boxes)
for axis in range(3):
    k = key((mn[:, axis] + mx[:, axis]).astype(f))
    order = np.argsort(k, kind="stable")
    pmn = np.minimum.accumulate(mn[order], axis=0); pmx = np.maximum.accumulate(mx[order], axis=0)
    smn = np.minimum.accumulate(mn[order][::-1], axis=0)[::-1]; smx = np.maximum.accumulate(mx[order][::-1], axis=0)[::-1]
    def ha(a, b):
        d = (b - a).astype(f); x, y, z = d[:, 0], d[:, 1], d[:, 2]
        return ((x + y).astype(f).astype(np.float64) * z.astype(np.float64) + x.astype(np.float64) * y.astype(np.float64)).astype(f)
    lc = (ha(pmn, pmx) * np.arange(1, n + 1, dtype=f)).astype(f)
    rcst = (ha(smn, smx) * np.arange(n, 0, -1, dtype=f)).astype(f)
    cost = (lc[:-1] + rcst[1:]).astype(f)
    i = int(np.argmin(cost))
    print("axis", axis, "first-min split index", i + 1, "cost", cost[i], "ties", int((cost == cost[i]).sum()), "cost at 99796:", cost[99795], "at 99759:", cost[99758])
