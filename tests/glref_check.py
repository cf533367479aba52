"""Checker shared by tests/test_glref.py (oracle vs the reference's llvmpipe outputs), tests/test_gpu_glref.py (HIP path vs
the same) and oracle/glref/make_vectors.py (summary).  The fixtures under tests/golden/glref/ are outputs of the
REFERENCE's own shaders; llvmpipe's float arithmetic is IEEE for + - * but its division, inverse square root and
transcendental functions are approximations of its own, so agreement is demanded to the tolerance north_star states
(1e-4 relative) — bit equality is reported, not required — and the handful of rays whose discrete decision (hit/miss at
an edge, Russian roulette, BSDF lobe) flips under a last-place difference are counted and bounded."""
import hashlib
import math
import numpy as np

REL_TOL = 1e-4            # BASELINE.json north_star tolerance
MAX_OUTLIER_FRAC = 0.003  # rays per stage allowed beyond REL_TOL or with a flipped discrete decision (observed: 0 - 0.1 %)
FIELDS = ("Origin", "Throughput", "Radiance", "PackedDirectionX", "PackedDirectionY", "PreviousIOROrTraverseCost")


def state_hash(rays, queue):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(rays).tobytes() + np.ascontiguousarray(queue, np.uint32).tobytes()).digest(), np.uint8)


def _records(fx_arr, dtype):
    """npz round trip keeps the structured dtype; be tolerant of a plain byte view."""
    a = np.asarray(fx_arr)
    return a if a.dtype.names else a.view(dtype)


def _compare_records(a, b):
    """a: candidate records, b: reference records (same length).  Returns (beyond-tolerance mask, bit-equal fraction, max rel)."""
    n = len(a)
    beyond = np.zeros(n, bool); eq_words = 0; words = 0; worst = 0.0
    if n == 0:
        return beyond, 1.0, 0.0
    for f in FIELDS:
        x = np.asarray(a[f], np.float32).reshape(n, -1); y = np.asarray(b[f], np.float32).reshape(n, -1)
        same = x.view(np.uint32) == y.view(np.uint32)
        eq_words += int(same.sum()); words += same.size
        with np.errstate(invalid="ignore", over="ignore"):
            rel = np.abs(x.astype(np.float64) - y.astype(np.float64)) / np.maximum(np.abs(y.astype(np.float64)), 1.0)
        rel = np.where(same, 0.0, rel)
        rel = np.where(np.isnan(rel), np.inf, rel)
        beyond |= (rel > REL_TOL).any(axis=1)
        fin = rel[np.isfinite(rel) & (rel <= REL_TOL)]
        if fin.size:
            worst = max(worst, float(fin.max()))
    return beyond, (eq_words / words if words else 1.0), worst


def _limit(n):
    return max(1, int(math.ceil(MAX_OUTLIER_FRAC * n)))


def check_case(fx, state_at, final, strict=True):
    """fx: loaded fixture; state_at(d) -> (ray records of the whole image, alive queue) of the implementation under test
    after a frame of RayDepth d (one sample); final: dict(image, counts, albedo, normal) of the case's own settings.
    Returns a report; with strict=True asserts the bounds."""
    depth = int(fx["depth"])
    rep = {"stages": []}
    # ---- FirstHit
    rays1, q1 = state_at(1)
    ref_rays = _records(fx["fh_rays"], rays1.dtype); ref_q = np.asarray(fx["fh_queue"], np.uint32)
    flips = np.setxor1d(q1, ref_q)
    keep = np.ones(len(rays1), bool); keep[flips] = False
    beyond, eq, worst = _compare_records(rays1[keep], ref_rays[keep])
    stage = {"stage": "FirstHit", "rays": int(len(rays1)), "flips": int(len(flips)), "beyond_tol": int(beyond.sum()), "bit_equal_words": round(eq, 4), "max_rel_within_tol": worst,
             "queue_identical": bool(np.array_equal(q1, ref_q))}
    rep["stages"].append(stage)
    if strict:
        assert len(flips) <= _limit(len(rays1)), stage
        assert beyond.sum() <= _limit(len(rays1)), stage
    # ---- NHit j from identical inputs
    prev = (rays1, q1)
    for j in range(1, depth):
        if f"in_hash_{j}" not in fx:
            break
        rin, qin = prev
        cur = state_at(j + 1)
        stale = not np.array_equal(state_hash(rin, qin), np.asarray(fx[f"in_hash_{j}"], np.uint8))
        if strict:
            assert not stale, f"bounce {j}: the state entering this bounce differs from the one the fixture's reference output was generated from (regenerate with oracle/glref/make_vectors.py)"
        ref_out = _records(fx[f"out_rays_{j}"], rin.dtype); ref_q = np.asarray(fx[f"out_queue_{j}"], np.uint32)
        flips = np.setxor1d(cur[1], ref_q)
        keep = ~np.isin(qin, flips)
        beyond, eq, worst = _compare_records(cur[0][qin][keep], ref_out[keep])
        stage = {"stage": f"NHit{j}", "rays": int(len(qin)), "flips": int(len(flips)), "beyond_tol": int(beyond.sum()), "bit_equal_words": round(eq, 4), "max_rel_within_tol": worst,
                 "queue_identical": bool(np.array_equal(cur[1], ref_q)), "stale_input": bool(stale)}
        rep["stages"].append(stage)
        if strict:
            assert len(flips) <= _limit(len(qin)), stage
            assert beyond.sum() <= _limit(len(qin)), stage
        prev = cur
    # ---- the reference's own whole frame (free run)
    img = np.asarray(final["image"], np.float32); ref_img = np.asarray(fx["free_image"], np.float32)
    ref_counts = [int(c) for c in fx["free_counts"]]
    counts = [int(c) for c in list(final["counts"])[1:1 + len(ref_counts)]]
    rel = np.abs(img.astype(np.float64) - ref_img) / np.maximum(np.abs(ref_img), 1.0)
    px_beyond = float((rel.max(axis=2) > REL_TOL).mean())
    free = {"counts": counts, "ref_counts": ref_counts, "counts_identical": counts == ref_counts,
            "pixels_bit_equal": round(float((img.view(np.uint32) == ref_img.view(np.uint32)).all(axis=2).mean()), 4), "pixels_beyond_tol": round(px_beyond, 5),
            "mean_abs_diff": float(np.abs(img - ref_img).mean()), "mean_ref": float(np.abs(ref_img).mean())}
    for k in ("albedo", "normal"):
        if final.get(k) is not None and ("free_" + k) in fx:
            a = np.asarray(final[k], np.float32); b = np.asarray(fx["free_" + k], np.float32)
            free[k + "_pixels_beyond_tol"] = round(float(((np.abs(a.astype(np.float64) - b) / np.maximum(np.abs(b), 1.0)).max(axis=2) > REL_TOL).mean()), 5)
    rep["free_run"] = free
    if strict:
        # alive counts may differ by the few flipped rays; a flip reshuffles every later slot-seeded RNG stream of its bounce, so pixel equality
        # is only demanded while the counts agree
        for a, b in zip(counts, ref_counts):
            assert abs(a - b) <= max(2, _limit(b) * 4), free
        if counts == ref_counts:
            assert px_beyond <= 0.005, free
            for k in ("albedo", "normal"):
                if k + "_pixels_beyond_tol" in free:
                    assert free[k + "_pixels_beyond_tol"] <= 0.005, free
        else:
            assert free["mean_abs_diff"] <= 0.05 * max(free["mean_ref"], 1e-3), free
    return rep


def check_query_hits(got, ref):
    """got: RayHit records of the implementation under test; ref: what the reference's TraceRay / TraceRayAny returned on llvmpipe.
    Hit/miss, triangle and instance identity must agree for every ray; T and the barycentrics of triangle hits are compared bit for
    bit (the traversal and Moeller-Trumbore use only + - * and one division whose result llvmpipe rounds like IEEE here: observed 100 %
    identical); sphere-light hits (sqrt) and misses (T = maxDist) to REL_TOL.  Fields the reference leaves undefined are skipped."""
    hit = ref["Hit"] != 0
    assert ((got["Hit"] != 0) == hit).all()
    tri = hit & (ref["TriangleId"] != 0xFFFFFFFF)
    assert (got["TriangleId"][hit] == ref["TriangleId"][hit]).all()
    assert (got["MeshTransformId"][hit] == ref["MeshTransformId"][hit]).all()
    for f in ("T", "BaryX", "BaryY"):
        assert (got[f][tri].view(np.uint32) == ref[f][tri].view(np.uint32)).all(), f
    light = hit & ~tri
    np.testing.assert_allclose(got["T"][light], ref["T"][light], rtol=REL_TOL)
    assert (got["T"][~hit].view(np.uint32) == ref["T"][~hit].view(np.uint32)).all()
    return {"rays": int(len(ref)), "hits": int(hit.sum()), "triangle_hits": int(tri.sum()), "light_hits": int(light.sum())}


def check_shadow_image(got, ref):
    """Visibility images of ShadowsRayTraced: sums of 0 / 1 / (1 - alpha) products over the samples, divided by the count."""
    got = np.asarray(got, np.float32); ref = np.asarray(ref, np.float32)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= REL_TOL
    assert (got.view(np.uint32) == ref.view(np.uint32)).mean() >= 0.999
