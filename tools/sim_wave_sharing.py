"""Developer tool (CPU): simulate k_trace2's while-while wave schedule on primary rays and measure how many lane-steps of a node
step fetch the SAME node pair as the wave's first stepping lane (a wave-uniform fetch could go through the scalar cache instead of
the vector-memory path, which is what binds the kernel: DESIGN.md 5).  Not bit-exact (float32 numpy slab tests, no jitter / lens),
only the schedule matters.  Usage: python tools/sim_wave_sharing.py [atrium|interior|headline] [tris] [samples] [W H]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np  # noqa: E402
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd.bvh import NativeBuilder  # noqa: E402

WAVE = 64
REFILL_MIN, LEAF_MIN = int(os.environ.get("REFILL_MIN", 32)), int(os.environ.get("LEAF_MIN", 24))
CHUNK = int(os.environ.get("CHUNK", 0))     # 0: a refill takes exactly its n entries from the list; > 0: waves reserve CHUNK entries at a time


def primary_rays(cam, W, H, x0, y0, w, h, jitter):
    ys, xs = np.mgrid[y0:y0 + h, x0:x0 + w]
    ndc = np.stack([(xs + jitter[0]) / W * 2 - 1, (ys + jitter[1]) / H * 2 - 1], -1).astype(np.float64)
    ip = cam.inv_projection.reshape(4, 4).astype(np.float64); iv = cam.inv_view.reshape(4, 4).astype(np.float64)
    clip = np.concatenate([ndc, -np.ones_like(ndc[..., :1]), np.ones_like(ndc[..., :1])], -1)
    eye = clip @ ip; eye[..., 2] = -1.0; eye[..., 3] = 0.0
    wd = (eye @ iv)[..., :3]
    wd /= np.linalg.norm(wd, axis=-1, keepdims=True)
    return wd.astype(np.float32)


def box_hit(o, inv, bmin, bmax, T):
    t0 = (bmin - o) * inv; t1 = (bmax - o) * inv
    tsm = np.minimum(t0, t1); tbg = np.maximum(t0, t1)
    tn = np.maximum(tsm.max(-1), 0.0); tf = tbg.min(-1)
    return (tn <= tf) & (tn <= T), tn


def tri_hit(o, d, p0, p1, p2):
    e1 = p1 - p0; e2 = p2 - p0; r = o - p0
    n = np.cross(e1, e2); q = np.cross(r, d)
    with np.errstate(all="ignore"):
        inv = 1.0 / (d * n).sum(-1)
        t = (-n * r).sum(-1) * inv; by = (-q * e2).sum(-1) * inv; bz = (q * e1).sum(-1) * inv
    ok = (1.0 - by - bz >= 0) & (by >= 0) & (bz >= 0) & (t >= 0)
    return ok, t


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "atrium"
    ntris = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
    samples = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    W, H = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (1920, 1080)
    b = NativeBuilder()
    if which == "atrium":
        sc = S.atrium_scene(ntris, b); cam = S.atrium_camera(W, H)
    else:
        sc = S.soup_scene(ntris, b, seed=1); cam = S.Camera(W, H, position=(0.0, 0.0, 0.0) if which == "interior" else (0.0, 0.0, 25.0))
    nodes = sc.blas_nodes
    nmin = np.ascontiguousarray(nodes["Min"]); nmax = np.ascontiguousarray(nodes["Max"])
    nstart = nodes["TriStartOrChild"].astype(np.int64); ncount = nodes["TriCount"].astype(np.int64)
    tris = sc.blas_triangles; P = sc.vertex_positions
    tv = np.stack([P[tris["X"]], P[tris["Y"]], P[tris["Z"]]], 1)          # (T, 3, 3)
    print(f"{which}: {len(tris)} BLAS triangles, {len(nodes)} nodes", flush=True)
    # a screen window in the middle of the frame, tile-major list with the samples of a 16-tile group adjacent (k_gen_primary's order)
    tw, th = 16, 8                                                         # tiles in the window (one group = 16 tiles of a tile row)
    x0, y0 = (W // 2 - tw * 4) // 8 * 8, (H // 2 - th * 4) // 8 * 8
    rng = np.random.default_rng(0)
    dirs = []
    for ty in range(th):
        for s in range(samples):
            j = rng.uniform(0, 1, 2)
            d = primary_rays(cam, W, H, x0, y0 + ty * 8, tw * 8, 8, j)     # (8, tw*8, 3)
            d = d.reshape(8, tw, 8, 3).transpose(1, 0, 2, 3).reshape(tw * 64, 3)   # tile-major, lane = (y&7)*8 + (x&7)
            dirs.append(d)
    dirs = np.concatenate(dirs); N = len(dirs)
    origin = np.broadcast_to(cam.position.astype(np.float32), (N, 3))
    # root test
    with np.errstate(all="ignore"):
        invd = (1.0 / dirs).astype(np.float32)
    NW = 4                                                                 # waves simulated side by side (they pull from one list)
    active = np.zeros((NW, WAVE), bool); leaf = np.zeros((NW, WAVE), bool)
    top = np.zeros((NW, WAVE), np.int64); ray = np.zeros((NW, WAVE), np.int64)
    T = np.full((NW, WAVE), np.inf, np.float32)
    lf = np.zeros((NW, WAVE), np.int64); le = np.zeros((NW, WAVE), np.int64)
    stacks = [[[] for _ in range(WAVE)] for _ in range(NW)]
    head = 0
    cnext = [0] * NW; cend = [0] * NW
    st = dict(steps=0, lane_steps=0, leader_share=0, best_share=0, distinct=0, uniform_steps=0, hist=np.zeros(65, np.int64), lane_by_share=np.zeros(65, np.int64))
    thresholds = (2, 4, 8, 16, 32)
    cover = {k: [0, 0] for k in thresholds}                                # lanes covered by the scalar path / scalar steps taken, if used when share >= k
    done = np.zeros(NW, bool)
    while not done.all():
        for w in range(NW):
            if done[w]:
                continue
            idle = ~active[w]
            if (head < N or cnext[w] < cend[w]) and (idle.sum() >= REFILL_MIN or idle.all()):
                n = int(idle.sum()); lanes = np.nonzero(idle)[0]
                if CHUNK == 0:
                    take = min(n, N - head)
                    ids = np.arange(head, head + take); head += n
                else:
                    ids = list(range(cnext[w], min(cend[w], cnext[w] + n)))
                    cnext[w] += len(ids)
                    if len(ids) < n and head < N:
                        fresh = head; head += CHUNK
                        more = list(range(fresh, min(fresh + (n - len(ids)), N)))
                        cnext[w] = fresh + len(more); cend[w] = min(fresh + CHUNK, N)
                        ids += more
                    ids = np.array(ids, np.int64); take = len(ids)
                lanes = lanes[:take]
                ray[w, lanes] = ids; active[w, lanes] = True; leaf[w, lanes] = False; T[w, lanes] = np.inf
                ok, _ = box_hit(origin[ids], invd[ids], nmin[1], nmax[1], T[w, lanes])
                top[w, lanes] = np.where(ok, 2, 0)
                for ln in lanes:
                    stacks[w][ln] = []
            if not active[w].any():
                if head >= N and cnext[w] >= cend[w]:
                    done[w] = True
                continue
            # node phase
            while True:
                can = active[w] & ~leaf[w] & (top[w] != 0)
                if not can.any() or (active[w] & leaf[w]).sum() >= LEAF_MIN:
                    break
                lanes = np.nonzero(can)[0]; tp = top[w, lanes]
                vals, cnts = np.unique(tp, return_counts=True)
                lead = int((tp == tp[0]).sum())
                st["steps"] += 1; st["lane_steps"] += len(lanes); st["leader_share"] += lead; st["best_share"] += int(cnts.max()); st["distinct"] += len(vals)
                st["uniform_steps"] += int(len(vals) == 1); st["hist"][lead] += 1; st["lane_by_share"][lead] += lead
                for k in thresholds:
                    if lead >= k:
                        cover[k][0] += lead; cover[k][1] += 1
                rid = ray[w, lanes]; o = origin[rid]; iv = invd[rid]
                hl, tl = box_hit(o, iv, nmin[tp], nmax[tp], T[w, lanes]); hr, tr = box_hit(o, iv, nmin[tp + 1], nmax[tp + 1], T[w, lanes])
                lc, rc = ncount[tp], ncount[tp + 1]; ls, rs = nstart[tp], nstart[tp + 1]
                il, ir = hl & (lc > 0), hr & (rc > 0)
                anyleaf = il | ir
                lf[w, lanes] = np.where(anyleaf, np.where(il, ls, rs), lf[w, lanes]); le[w, lanes] = np.where(anyleaf, np.where(~ir, ls + lc, rs + rc), le[w, lanes])
                leaf[w, lanes] = anyleaf
                trl, trr = hl & (lc == 0), hr & (rc == 0)
                for i, ln in enumerate(lanes):
                    if trl[i] and trr[i]:
                        closer = tl[i] < tr[i]
                        top[w, ln] = ls[i] if closer else rs[i]; stacks[w][ln].append(rs[i] if closer else ls[i])
                    elif trl[i] or trr[i]:
                        top[w, ln] = ls[i] if trl[i] else rs[i]
                    else:
                        top[w, ln] = stacks[w][ln].pop() if stacks[w][ln] else 0
            # leaf phase
            for ln in np.nonzero(active[w] & leaf[w])[0]:
                r = ray[w, ln]; idx = np.arange(lf[w, ln], le[w, ln])
                ok, t = tri_hit(origin[r], dirs[r], tv[idx, 0], tv[idx, 1], tv[idx, 2])
                ok &= t < T[w, ln]
                if ok.any():
                    T[w, ln] = t[ok].min()
                leaf[w, ln] = False
            fin = active[w] & (top[w] == 0)
            active[w, fin] = False
    ls = st["lane_steps"]
    print(f"rays {N}, wave node steps {st['steps']}, lane-steps {ls} ({ls / st['steps']:.1f} lanes/step, {ls / N:.1f} steps/ray)")
    print(f"leader share {st['leader_share'] / ls:.3f} of lane-steps, best-node share {st['best_share'] / ls:.3f}, distinct nodes per step {st['distinct'] / st['steps']:.1f}, fully uniform steps {st['uniform_steps'] / st['steps']:.3f}")
    for k in thresholds:
        c, n = cover[k]
        print(f"  scalar path when leader share >= {k:2d}: covers {c / ls:.3f} of the lane-steps in {n / st['steps']:.3f} of the steps (vector lane-requests left: {1 - c / ls:.3f})")


if __name__ == "__main__":
    main()
