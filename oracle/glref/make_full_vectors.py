"""Mint tests/golden/glref_full/*.npz: the REFERENCE's own GLSL shaders (oracle/glref/glref.py on Mesa llvmpipe) on WHOLE frames at BASELINE size —
1920x1080 on the 1M-triangle scenes bench.py times (tests/golden/glref_cases.py FULL_CASES) — compared with the oracle ray by ray, stage by stage.
Test infrastructure; runs only in the build container (/root/reference + Mesa swrast).

    python oracle/glref/make_full_vectors.py [case ...]            (re)generate fixtures + summary
    python oracle/glref/make_full_vectors.py --check [case ...]    regenerate in memory, compare with the committed fixtures bit for bit
    python oracle/glref/make_full_vectors.py --updates [--check]   BLASRefit + Skinning shaders on the refittable 1M-triangle scene (config-5 size): updates_1m.npz

Method (the one of make_vectors.py): FirstHit is run by the reference on the whole frame; every bounce j is ONE NHit dispatch of the reference started from
the ORACLE's state after j-1 bounces, so every stage is compared from identical inputs.  All rays of a stage are compared here, under the gate of
tests/glref_check.py; the fixture keeps
  * `state_hash_<j>`: sha256 of the oracle's state (ray records + alive queue) after stage j — the state that was compared.  A test that finds the same hash
    on its candidate (the oracle again, or the HIP path, which equals it bit for bit) inherits this comparison of every ray;
  * `idx_<j>`, `ref_<j>`: the reference's records on every FULL_SAMPLE_STRIDE-th ray of the stage, for a direct comparison wherever the fixture travels;
  * `exc_ids_<j>`, `exc_ref_<j>`, `exc_cand_<j>`: EVERY ray on which the reference's run and the oracle differ beyond the gate (tolerance or alive decision),
    with both results; `exc_bf_<j>`: what a binary64 brute force over all triangles says about that ray (tests/c_driver/brute_force.c): [closest t, runner-up t,
    distance of the oracle's result, distance of the reference's result];
  * the reference's alive queue as a hash plus its symmetric difference with the oracle's (`ref_queue_hash_<j>`, `flips_<j>`).
tests/golden/glref_full/summary.json holds the whole-stage statistics (rays, flips, beyond tolerance, worst deviation)."""
import hashlib
import json
import os
import sys
import time
import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(ROOT, "tests", "golden", "glref_full")


def queue_hash(q):
    return np.frombuffer(hashlib.sha256(np.sort(np.asarray(q, np.uint32)).tobytes()).digest(), np.uint8)      # the alive SET (sorted ids)


def brute_force_exceptions(sc, origins, dirs, cand_pts, ref_pts):
    """binary64 closest hit of the given rays against every triangle; per ray [t closest, t runner-up, |candidate point - origin|, |reference point - origin|]."""
    import test_metamorphic as M
    tris, _ = M.world_triangles(sc)
    t, _, t2 = M.brute_force(tris, origins, dirs)
    return np.stack([t, t2, np.linalg.norm(cand_pts - origins, axis=1), np.linalg.norm(ref_pts - origins, axis=1)], 1)


def main(names, check=False):
    from oracle.glref import glref as G
    from oracle import oracle as O
    from idkengine_amd import gputypes as T
    from idkengine_amd.bvh import NativeBuilder
    import configs
    import glref_cases
    import glref_check
    os.makedirs(OUT, exist_ok=True)
    spath = os.path.join(OUT, "summary.json")
    summary = json.load(open(spath)) if os.path.exists(spath) else {}
    scenes = {}
    for name in names or list(glref_cases.FULL_CASES):
        skey, camf, w, h, ov = glref_cases.FULL_CASES[name]
        stride = glref_cases.FULL_SAMPLE_STRIDE.get(name, glref_cases.FULL_SAMPLE_STRIDE[None])
        if skey not in scenes:
            scenes[skey] = glref_cases.FULL_SCENES[skey](NativeBuilder())     # (the product builder: same bytes as the oracle's, tests/test_builder.py; minutes faster at this size)
        sc = scenes[skey]; cam = camf(w, h)
        st = configs.apply_settings(T.Settings.default(), ov)
        depth = int(st.RayDepth)
        t0 = time.time()

        def oracle_state(d):
            o = O.OraclePathTracer(sc, w, h); o.set_camera(cam); configs.apply_settings(o.settings, ov); o.settings.RayDepth = d; o.settings.SamplesPerPixel = 1
            o.render(); r, q, k = o.rays().copy(), o.alive_queue().copy(), o.alive_keys().copy(); o.close()
            return r, q, k
        out = {"mesa": np.frombuffer(G.gl().glref_info(), np.uint8), "depth": depth, "width": w, "height": h, "stride": stride}
        rep = {"stages": []}
        st1 = configs.apply_settings(T.Settings.default(), ov); st1.RayDepth = 1; st1.SamplesPerPixel = 1
        pt = G.ReferencePathTracer(sc, w, h, st1); pt.set_camera(cam); pt.render()
        ref_rays, ref_q = pt.rays(T.GpuWavefrontRay), np.asarray(pt.final_alive, np.uint32); pt.accumulated = 0
        cur = oracle_state(1)
        campos = np.asarray(cam.position, np.float64)

        def record_stage(j, ids, cand, ref, cand_q, ref_q, in_origins, in_dirs):
            """ids: ray ids compared in this stage (all pixels for FirstHit, the queue entering bounce j otherwise); cand / ref: records of those rays."""
            flips = np.setxor1d(cand_q, ref_q)
            beyond, eq, worst = glref_check._compare_records(cand, ref)
            exc = beyond | np.isin(ids, flips)
            pos = np.nonzero(exc)[0]
            out[f"state_hash_{j}"] = glref_check.state_hash(cur[0], cur[1])
            out[f"ref_queue_hash_{j}"] = queue_hash(ref_q); out[f"flips_{j}"] = flips.astype(np.uint32)
            samp = np.arange(0, len(ids), stride)
            out[f"idx_{j}"] = ids[samp].astype(np.uint32); out[f"ref_{j}"] = ref[samp]
            out[f"exc_ids_{j}"] = ids[pos].astype(np.uint32); out[f"exc_ref_{j}"] = ref[pos]; out[f"exc_cand_{j}"] = cand[pos]
            bf = np.zeros((0, 4))
            if 0 < len(pos) <= 2000:     # (a stage with more listed rays than that has a systematic difference: no point in asking the brute force about each)
                bf = brute_force_exceptions(sc, in_origins[pos], in_dirs[pos], cand["Origin"][pos].astype(np.float64), ref["Origin"][pos].astype(np.float64))
            out[f"exc_bf_{j}"] = bf
            stage = {"stage": "FirstHit" if j == 0 else f"NHit{j}", "rays": int(len(ids)), "flips": int(len(flips)), "beyond_tol": int(beyond.sum()), "exceptions": int(len(pos)),
                     "bit_equal_words": round(eq, 4), "max_rel_within_tol": worst,
                     "exceptions_where_oracle_is_the_binary64_closest_hit": int(sum(1 for r in bf if np.isfinite(r[0]) and abs(r[2] - r[0]) <= 3e-3 + 1e-4 * r[0])),
                     "exceptions_where_reference_is_the_binary64_closest_hit": int(sum(1 for r in bf if np.isfinite(r[0]) and abs(r[3] - r[0]) <= 3e-3 + 1e-4 * r[0]))}
            rep["stages"].append(stage)
            print("  ", name, json.dumps(stage), flush=True)

        # ---- FirstHit: all pixels; the primary ray of a pixel leaves the camera position (no lens in these cases) through the hit point
        ids = np.arange(w * h, dtype=np.uint32)
        d0 = cur[0]["Origin"].astype(np.float64) - campos; nrm = np.linalg.norm(d0, axis=1, keepdims=True); d0 = d0 / np.maximum(nrm, 1e-30)
        record_stage(0, ids, cur[0], ref_rays, cur[1], ref_q, np.broadcast_to(campos, d0.shape).copy(), d0)
        if st.Gpu.DoDebugBVHTraversal:      # the reference's own count of the traversal work of the frame's primary rays, and the oracle's counters for the same rays
            o = O.OraclePathTracer(sc, w, h); o.set_camera(cam); configs.apply_settings(o.settings, ov); o.enable_counters(True); o.render(); cst = o.stats(); o.close()
            out["ref_cost_sum"] = np.float64(ref_rays["PreviousIOROrTraverseCost"].astype(np.float64).sum())
            out["cand_cost_sum"] = np.float64(cur[0]["PreviousIOROrTraverseCost"].astype(np.float64).sum())
            out["cand_pairs"] = np.int64(cst["node_pair_visits"]); out["cand_tri_tests"] = np.int64(cst["triangle_tests"])
            rep["traversal_cost"] = {"reference_debugCost_sum": float(out["ref_cost_sum"]), "oracle_debugCost_sum": float(out["cand_cost_sum"]),
                                     "oracle_node_pair_visits": int(out["cand_pairs"]), "oracle_triangle_tests": int(out["cand_tri_tests"]),
                                     "pairs_plus_1.1_tests": float(out["cand_pairs"] + 1.1 * float(out["cand_tri_tests"])),
                                     "pixels_with_identical_cost": float((ref_rays["PreviousIOROrTraverseCost"].view(np.uint32) == cur[0]["PreviousIOROrTraverseCost"].view(np.uint32)).mean())}
            print("  ", name, json.dumps(rep["traversal_cost"]), flush=True)
        prev_out = None
        key_diffs = {}
        for j in range(1, depth):
            rin, qin = cur[0], cur[1]
            rout, qout = pt.run_nhit_from(rin, qin, j, sort_first=bool(st.DoRaySorting), keys=(cur[2] if (st.DoRaySorting and j > 1) else None)); prev_out = qout
            nxt = oracle_state(j + 1)
            if st.DoRaySorting and pt.last_out_keys is not None:
                # the keys the reference's dispatch cached for the next sort against the oracle's, ray by ray (a key is a TriangleId: a discrete result of the stage)
                kr = np.zeros(w * h, np.int64) - 1; kr[np.asarray(qout, np.int64)] = pt.last_out_keys
                ko = np.zeros(w * h, np.int64) - 1; ko[nxt[1].astype(np.int64)] = nxt[2]
                both = np.intersect1d(qout, nxt[1])
                kd = both[kr[both] != ko[both]]
                out[f"key_diff_ids_{j}"] = kd.astype(np.uint32); out[f"key_diff_ref_{j}"] = kr[kd].astype(np.uint32); out[f"key_diff_cand_{j}"] = ko[kd].astype(np.uint32)
                key_diffs[j] = int(len(kd))
            o_in = rin["Origin"][qin].astype(np.float64)
            d_in = glref_check.decode_unit_vec(rin["PackedDirectionX"][qin], rin["PackedDirectionY"][qin])
            cur = nxt
            record_stage(j, qin.astype(np.uint32), nxt[0][qin], rout[qin], nxt[1], np.asarray(qout, np.uint32), o_in, d_in)
            if j in key_diffs:
                rep["stages"][-1]["sort_keys_differing"] = key_diffs[j]
        pt.close()
        # ---- the reference's own whole frames, free-running (FirstHit, NHit, FinalDraw; two accumulated samples): only where no later bounce can shift RNG slots (RayDepth 2)
        if depth == 2 and not st.Gpu.DoDebugBVHTraversal:
            nfree = 2
            ptf = G.ReferencePathTracer(sc, w, h, st); ptf.set_camera(cam)
            o = O.OraclePathTracer(sc, w, h); o.set_camera(cam); configs.apply_settings(o.settings, ov)
            for _ in range(nfree):
                ptf.render(); o.render()
            rimg = np.asarray(ptf.image(0), np.float32).reshape(-1, 4); oimg = np.asarray(o.image(0), np.float32).reshape(-1, 4)
            counts_same = [int(c) for c in ptf.alive_counts] == [int(c) for c in list(o.stats()["alive_counts"])[1:1 + len(ptf.alive_counts)]]
            ptf.close(); o.close()
            rel = glref_check.pixel_rel_err(oimg, rimg)
            px = np.nonzero(rel > glref_check.REL_TOL)[0].astype(np.uint32)
            samp = np.arange(0, w * h, stride)
            out["free_samples"] = np.int64(nfree); out["free_image_hash"] = sha(oimg)
            out["free_idx"] = samp.astype(np.uint32); out["free_ref"] = rimg[samp]
            out["free_exc_px"] = px; out["free_exc_ref"] = rimg[px]; out["free_exc_cand"] = oimg[px]
            rep["free_run"] = {"samples": nfree, "pixels": int(w * h), "pixels_beyond_tol": int(len(px)), "alive_counts_identical": bool(counts_same),
                               "pixels_bit_equal": round(float((rimg.view(np.uint32) == oimg.view(np.uint32)).all(axis=1).mean()), 4),
                               "max_rel_within_tol": float(rel[rel <= glref_check.REL_TOL].max())}
            print("  ", name, json.dumps(rep["free_run"]), flush=True)
        rep["seconds"] = round(time.time() - t0, 1)
        path = os.path.join(OUT, name + ".npz")
        if check:       # regenerate in memory and demand the committed fixture bit for bit (tests/test_glref_full.py, `live`)
            fx = np.load(path)
            bad = [k for k in out if k != "mesa" and not (k in fx and np.asarray(out[k]).tobytes() == np.asarray(fx[k]).tobytes())] + [k for k in fx.files if k not in out]
            print(name, "reproduced" if not bad else f"DIFFERS in {bad}", flush=True)
            if bad:
                raise SystemExit(1)
            continue
        np.savez_compressed(path, **out)
        rep["fixture_bytes"] = os.path.getsize(path)
        rep["gate"] = {"rel_tol": glref_check.REL_TOL, "abs_floor": glref_check.ABS_FLOOR, "max_listed_exceptions_per_stage": glref_check.FULL_ALLOW.get(name), "reason": glref_check.FULL_EXCEPTION_REASON}
        summary[name] = rep
        json.dump(summary, open(spath, "w"), indent=1, sort_keys=True)
        print(name, "written", rep["fixture_bytes"], "bytes in", rep["seconds"], "s", flush=True)





# ---- config-5 size: Shaders/BLASRefit/compute.glsl and Shaders/Skinning/compute.glsl on the refittable 1M-triangle scene (3 M vertices) -------------------------
def full_update_inputs(builder):
    """The animated stand-in of BASELINE configs[5] at full size (deterministic; shared with the tests): refittable soup-1M, every vertex displaced, and the
    skinning inputs of tools/bench_animated.py's kind (all 3 M vertices, two joints, weights from the position)."""
    from idkengine_amd import scenes as S, gputypes as T
    sc = S.soup_scene(1000000, builder, seed=1, refittable=True)
    p = sc.vertex_positions
    moved = (p + np.sin(p[:, ::-1] * np.float32(1.7)).astype(np.float32) * np.float32(0.05)).astype(np.float32)
    n = len(p)
    un = np.zeros(n, T.GpuUnskinnedVertex)
    un["Position"] = p
    un["Normal"] = sc.vertices["Normal"][:n] if "Normal" in sc.vertices.dtype.names else 0
    un["Tangent"] = sc.vertices["Tangent"][:n] if "Tangent" in sc.vertices.dtype.names else 0
    un["JointIndices"][:, 0] = 0; un["JointIndices"][:, 1] = 1
    w0 = (np.float32(0.5) + np.float32(0.5) * np.sin(p[:, 1] * np.float32(0.3))).astype(np.float32)
    un["JointWeights"][:, 0] = w0; un["JointWeights"][:, 1] = np.float32(1.0) - w0
    joints = np.zeros((2, 3, 4), np.float32)
    c, s_ = np.float32(np.cos(0.05)), np.float32(np.sin(0.05))
    joints[0, :, :3] = [[c, 0, s_], [0, 1, 0], [-s_, 0, c]]; joints[0, :, 3] = (0.02, 0.0, -0.01)
    joints[1, :, :3] = np.eye(3); joints[1, :, 3] = (0.0, 0.03, 0.0)
    return sc, moved, un, joints


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def make_full_updates(check=False):
    from oracle.glref import glref as G
    from idkengine_amd.bvh import NativeBuilder
    t0 = time.time()
    sc, moved, un, joints = full_update_inputs(NativeBuilder())
    up = G.ReferenceSceneUpdates(sc)
    up.set_positions(moved)
    nodes = up.refit(0)
    out = {"refit_nodes_hash": sha(nodes), "refit_nodes_sample": nodes[::509], "node_count": np.int64(len(nodes))}
    up.set_positions(sc.vertex_positions)
    pos, prev, verts = up.skin(un, joints, 0, 0, 0, len(un))
    out["skin_positions_hash"] = sha(pos); out["skin_positions_sample"] = pos[::1021]
    out["skin_prev_is_input"] = np.array([np.array_equal(prev, sc.vertex_positions)])
    out["skin_normals_sample"] = verts["Normal"][::1021]; out["skin_tangents_sample"] = verts["Tangent"][::1021]
    up.close()
    path = os.path.join(OUT, "updates_1m.npz")
    if check:
        fx = np.load(path)
        bad = [k for k in out if not (k in fx and np.asarray(out[k]).tobytes() == np.asarray(fx[k]).tobytes())]
        print("updates_1m", "reproduced" if not bad else f"DIFFERS in {bad}", flush=True)
        if bad:
            raise SystemExit(1)
        return
    np.savez_compressed(path, **out)
    print("updates_1m written", os.path.getsize(path), "bytes in", round(time.time() - t0, 1), "s:", len(nodes), "nodes,", len(un), "vertices", flush=True)


if __name__ == "__main__":
    if "--updates" in sys.argv:
        make_full_updates(check="--check" in sys.argv)
    else:
        main([a for a in sys.argv[1:] if not a.startswith("--")], check="--check" in sys.argv)
