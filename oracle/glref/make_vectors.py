"""Mint tests/golden/glref/*.npz: outputs of the REFERENCE's own GLSL shaders (run by oracle/glref/glref.py on Mesa
llvmpipe) for every case of tests/golden/glref_cases.py.  Runs only in the build container (/root/reference + Mesa
swrast); the fixtures travel to the GPU box, this script's inputs do not.

    python oracle/glref/make_vectors.py [case ...]            (re)generate fixtures + summary
    python oracle/glref/make_vectors.py --check [case ...]    regenerate in memory, compare with the committed fixtures bit for bit
    python oracle/glref/make_vectors.py --queries [--check]   ray-query (TraceRay / TraceRayAny) and ShadowsRayTraced vectors
    python oracle/glref/make_vectors.py --extended            three larger live comparisons (no fixtures), JSON on stdout
    python oracle/glref/make_vectors.py --eight-bit           what llvmpipe does to RGBA8 / sRGB8 textures beside the specification's decode + float filter, JSON on stdout
    python oracle/glref/make_vectors.py --defect-d1           demonstrate reference defect D1 (glref.py ADAPTATIONS A7), JSON on stdout

Per case the fixture holds
  * stage vectors ("forced"): FirstHit's full output (ray records + alive queue), and for every bounce j the output of
    ONE NHit dispatch of the reference started from the ORACLE's state after j-1 bounces (sha256 of that input state is
    stored, so a drifting oracle is detected) — every bounce is compared from identical inputs;
  * free-run vectors: the reference's whole frame(s) (Result image, AOV images, alive counts per bounce), queue kept in
    the canonical order between dispatches (glref.py docstring).
A summary (bit-equal fractions, worst deviations, flipped decisions vs the oracle) goes to tests/golden/glref/summary.json.
"""
import hashlib
import json
import os
import sys
import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(ROOT, "tests", "golden", "glref")


def state_hash(rays, queue):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(rays).tobytes() + np.ascontiguousarray(queue, np.uint32).tobytes()).digest(), np.uint8)


def defect_d1():
    """Runs the reference's RaySorting exactly in the reference's host order (no A7) and reports what Reorder does."""
    from oracle.glref import glref as G
    from oracle import oracle as O
    from idkengine_amd import gputypes as T
    import configs
    import glref_cases
    fac, camf, w, h, ov = glref_cases.GLREF_CASES["presplit_sort_d4"]
    sc = fac(O.OracleBuilder()); cam = camf(w, h)
    st = configs.apply_settings(T.Settings.default(), ov); st.RayDepth = 3
    rep = {}
    for fix in (False, True):
        pt = G.ReferencePathTracer(sc, w, h, st, sort_count_fix=fix); pt.set_camera(cam)
        seen = {}
        inner = pt._ray_sorting

        def spy(count_arg=None):
            hdr = pt.header()
            new = 1 - hdr["pingpong"] if not fix else hdr["pingpong"]
            cnt = int(hdr["counts"][new])
            before = pt.alive_queue(cnt); keys = pt._u32(pt.b_keys, 0, cnt)
            item_count_seen_by_reorder = int(hdr["counts"][hdr["pingpong"]])
            inner(count_arg)
            after = pt.alive_queue(cnt)
            seen.update(alive=cnt, reorder_item_count=item_count_seen_by_reorder, dispatched_invocations=int(hdr["groups"][0]) * 32,
                        is_permutation=bool(np.array_equal(np.sort(before), np.sort(after))),
                        equals_stable_sort=bool(np.array_equal(after, before[np.argsort(keys, kind="stable")])))
        pt._ray_sorting = spy
        pt.render()
        rep["with_A7" if fix else "reference_order"] = seen
        pt.close()
    print(json.dumps(rep))


def eight_bit_report():
    """What llvmpipe makes of 8-bit textures, measured on the sampler wall of glref_cases._sampler_scene through the reference's FirstHit (AOV albedo = the texture tap):
    RGBA8 under GL_NEAREST is the specification's c / 255; RGBA8 under GL_LINEAR is filtered in 8-bit fixed point (gallivm's AoS path); sRGB8 is decoded by a polynomial.
    The oracle (and the HIP path) decode per texel as GL 4.6 2.3.5.1 / 8.24 write it and filter in float: the fixture pins the first, the other two are bounded here."""
    from oracle.glref import glref as G
    from oracle import oracle as O
    from idkengine_amd import gputypes as T
    import configs
    import glref_cases
    fac, camf, w, h, ov = glref_cases.GLREF_CASES["sampler_states_rgba8_d3"]
    B = O.OracleBuilder(); rep = {}
    for label, mf, srgb in (("rgba8_nearest", 1, False), ("rgba8_linear", 0, False), ("srgb8_nearest", 1, True), ("srgb8_linear", 0, True)):
        sc = fac(B); cam = camf(w, h)
        sc.textures = [T.TextureImage(t.data, t.wrap_s, t.wrap_t, mf, srgb=srgb) for t in sc.textures]
        st = configs.apply_settings(T.Settings.default(), ov)
        pt = G.ReferencePathTracer(sc, w, h, st); pt.set_camera(cam); pt.render()
        ref = pt.image(1)[..., :3].reshape(-1, 3).astype(np.float64); pt.close()
        o = O.OraclePathTracer(sc, w, h); o.set_camera(cam); configs.apply_settings(o.settings, ov); o.render()
        alb = o.image(1)[..., :3].reshape(-1, 3).astype(np.float64); o.close()
        d = np.abs(ref - alb); big = alb > 1e-2
        rep[label] = {"pixels": int(len(alb)), "textured_pixels": int((alb.sum(1) > 0).sum()), "max_abs": float(d.max()), "max_rel": float((d[big] / alb[big]).max()), "pixels_beyond_1e-4": int(((d > 1e-4 * np.maximum(alb, 1e-3)).any(1)).sum())}
    print(json.dumps(rep))


def query_scene(builder):
    from idkengine_amd import scenes as S
    sc = S.cornell_scene(builder, "mixed", True)
    sc.lights = S.make_lights([((0.0, 0.55, 0.2), 0.12, (20.0, 20.0, 20.0)), ((-0.5, -0.2, 0.6), 0.08, (5.0, 2.0, 2.0))])
    return sc


def shadow_scene(builder, variant):
    sc = query_scene(builder)
    if variant == "blend":   # short box alpha-blended, tall box alpha-tested: the continue-through-surface loop of the shader
        sc.materials["AlphaCutoff"][-2] = 2.0; sc.materials["BaseColorFactor"][-2] = (sc.materials["BaseColorFactor"][-2] & 0x00FFFFFF) | (0x60 << 24)
        sc.materials["AlphaCutoff"][-1] = 0.5; sc.materials["BaseColorFactor"][-1] = (sc.materials["BaseColorFactor"][-1] & 0x00FFFFFF) | (0x40 << 24)
    return sc


SHADOW_CONFIGS = [("mixed", 0), ("mixed", 1), ("blend", 0)]
SHADOW_PARAMS = [(0, 1, 0), (0, 4, 8), (1, 3, 6)]          # (light, RayTracingSamples, NoiseIndex = k * samples)
SHADOW_SIZE = (96, 80)


def update_inputs(builder):
    """Scene + inputs of the refit / skinning vectors (deterministic; shared with the tests)."""
    from idkengine_amd import scenes as S, gputypes as T
    sc = S.soup_scene(5000, builder, seed=12, refittable=True)
    rng = np.random.default_rng(3)
    moved = (sc.vertex_positions + np.sin(sc.vertex_positions[:, ::-1] * 1.7).astype(np.float32) * np.float32(0.05) + rng.normal(0, 0.01, sc.vertex_positions.shape)).astype(np.float32)
    n = 600
    un = np.zeros(n, T.GpuUnskinnedVertex)
    un["Position"] = sc.vertex_positions[:n]
    nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    tng = np.cross(nrm, rng.normal(size=(n, 3))); tng /= np.linalg.norm(tng, axis=1, keepdims=True)
    un["Normal"] = S.compress_sr11g11b10(nrm.astype(np.float32)); un["Tangent"] = S.compress_sr11g11b10(tng.astype(np.float32))
    un["JointIndices"] = rng.integers(0, 3, (n, 4)); wts = rng.uniform(0, 1, (n, 4)).astype(np.float32); un["JointWeights"] = wts / wts.sum(1, keepdims=True)
    joints = np.zeros((5, 3, 4), np.float32)                        # joint 0/1 unused by offset 2: JointMatricesOffset is exercised
    joints[2, :, :3] = np.eye(3); joints[2, :, 3] = (0.1, 0.0, -0.2)
    c, s_ = np.cos(0.3), np.sin(0.3); joints[3, :, :3] = [[c, 0, s_], [0, 1, 0], [-s_, 0, c]]; joints[3, :, 3] = (0, 0.3, 0)
    joints[4, :, :3] = np.diag([1.2, 0.8, 1.0]); joints[4, :, 3] = (-0.05, 0.02, 0.3)
    return sc, moved, un, joints, dict(input_offset=0, output_offset=40, joint_offset=2, count=n - 10)


def make_query_and_shadow_vectors(check=False):
    """queries.npz: the reference's TraceRay / TraceRayAny (BVHIntersect.glsl) on 4000 random rays x {closest, any} x {lights} x {TLAS};
    shadows.npz: Shaders/ShadowsRayTraced/compute.glsl on a G-buffer of the Cornell scene, 3 scene/TLAS configs x 3 parameter sets;
    updates.npz: Shaders/BLASRefit/compute.glsl on a displaced refittable soup and Shaders/Skinning/compute.glsl with offsets."""
    from oracle.glref import glref as G
    from oracle import oracle as O
    from idkengine_amd import scenes as S, gputypes as T
    B = O.OracleBuilder()
    out = {}
    sc = query_scene(B)
    rng = np.random.default_rng(11)
    rays = np.zeros(4000, T.RayQuery)
    rays["Origin"] = rng.uniform(-1.1, 1.1, (len(rays), 3)).astype(np.float32)
    d = rng.normal(size=(len(rays), 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays["Direction"] = d.astype(np.float32); rays["MaxDist"] = 3.4028235e+38
    rays["MaxDist"][::3] = rng.uniform(0.05, 2.0, len(rays[::3])).astype(np.float32)      # a third of the rays are range-limited
    out["rays"] = rays
    for tlas in (0, 1):
        rq = G.ReferenceRayQuery(sc, use_tlas=bool(tlas))
        for any_hit in (0, 1):
            for lights in (0, 1):
                out[f"hits_tlas{tlas}_any{any_hit}_lights{lights}"] = rq.trace(rays, bool(any_hit), bool(lights))
        rq.close()
    sh = {}
    w, h = SHADOW_SIZE
    cam = S.cornell_camera(w, h)
    prim = S.primary_ray_queries(cam, w, h)
    for variant, tlas in SHADOW_CONFIGS:
        scs = shadow_scene(B, variant)
        hits = O.trace_rays(scs, prim, use_tlas=bool(tlas))
        depth, normal = S.gbuffer_from_hits(scs, cam, w, h, prim, hits)
        sh[f"depth_{variant}_{tlas}"] = depth; sh[f"normal_{variant}_{tlas}"] = normal
        rs = G.ReferenceShadows(scs, use_tlas=bool(tlas))
        for light, samples, noise in SHADOW_PARAMS:
            p = T.ShadowParams.make(cam.inv_proj_view, w, h, light_index=light, samples=samples, noise_index=noise, jitter=(0.0005, -0.0003))
            sh[f"vis_{variant}_{tlas}_{light}_{samples}_{noise}"] = rs.trace(p, depth, normal, visibility=np.full((h, w), np.float32(-3.0)))
        rs.close()
    # ---- BLASRefit / Skinning
    sc_u, moved, un, joints, sk = update_inputs(B)
    up = G.ReferenceSceneUpdates(sc_u)
    up.set_positions(moved)
    upd = {"refit_nodes": up.refit(0)}
    pos, prev, verts = up.skin(un, joints, sk["input_offset"], sk["output_offset"], sk["joint_offset"], sk["count"])
    lo, hi = sk["output_offset"], sk["output_offset"] + sk["count"]
    upd["skin_positions"] = pos[lo:hi]; upd["skin_prev_positions"] = prev[lo:hi]; upd["skin_normals"] = verts["Normal"][lo:hi]; upd["skin_tangents"] = verts["Tangent"][lo:hi]
    upd["skin_untouched_ok"] = np.array([np.array_equal(pos[:lo], moved[:lo]) and np.array_equal(pos[hi:], moved[hi:])])
    up.close()
    failed = []
    for fname, data in (("queries.npz", out), ("shadows.npz", sh), ("updates.npz", upd)):
        path = os.path.join(OUT, fname)
        if check:
            fx = np.load(path)
            bad = [k for k in data if not (k in fx and np.asarray(data[k]).tobytes() == np.asarray(fx[k]).tobytes())] + [k for k in fx.files if k not in data]
            print(fname, "reproduced" if not bad else f"DIFFERS in {bad}", flush=True)
            failed += bad
        else:
            np.savez_compressed(path, **data)
            print(fname, "written", os.path.getsize(path), "bytes", flush=True)
    return failed


def extended():
    """Larger live comparisons (no fixtures): every bounce of three bigger frames, the reference's shaders on llvmpipe against the oracle from
    identical inputs.  JSON on stdout: per case the rays compared, flipped decisions, values beyond tolerance, worst deviation."""
    from oracle.glref import glref as G
    from oracle import oracle as O
    from idkengine_amd import scenes as S, gputypes as T
    import configs
    import glref_check
    B = O.OracleBuilder()
    cases = {
        "soup50k_256x144_d5": (lambda b: S.soup_scene(50000, b, seed=9), lambda w, h: S.Camera(w, h, position=(0, 0, 0), view_dir=(0.2, 0.1, -1), fovy_deg=90), 256, 144, dict(RayDepth=5)),
        "atrium40k_192x108_d4": (lambda b: S.atrium_scene(40000, b), S.atrium_camera, 192, 108, dict(RayDepth=4)),
        "lucy_256x320_d4_sort": (configs.lucy_scene, configs.lucy_camera, 256, 320, dict(RayDepth=4, DoRaySorting=1)),
    }
    rep = {}
    for name, (fac, camf, w, h, ov) in cases.items():
        sc = fac(B); cam = camf(w, h)
        st = configs.apply_settings(T.Settings.default(), ov); depth = int(st.RayDepth)

        def oracle_state(d):
            o = O.OraclePathTracer(sc, w, h); o.set_camera(cam); configs.apply_settings(o.settings, ov); o.settings.RayDepth = d; o.settings.SamplesPerPixel = 1
            o.render(); r, q = o.rays(), o.alive_queue(); o.close()
            return r, q
        st1 = configs.apply_settings(T.Settings.default(), ov); st1.RayDepth = 1
        pt = G.ReferencePathTracer(sc, w, h, st1); pt.set_camera(cam); pt.render()
        fh_rays, fh_q = pt.rays(T.GpuWavefrontRay), pt.final_alive; pt.accumulated = 0
        r1, q1 = oracle_state(1)
        beyond, eq, worst = glref_check._compare_records(r1, fh_rays)
        tot = {"rays": len(r1), "flips": int(len(np.setxor1d(fh_q, q1))), "beyond_tol": int(beyond.sum()), "max_rel": worst, "stages": 1}
        prev_out, prev = None, (r1, q1)
        for j in range(1, depth):
            rin, qin = prev
            if st.DoRaySorting and j > 1 and not np.array_equal(prev_out, qin):
                break
            rout, qout = pt.run_nhit_from(rin, qin, j, sort_first=bool(st.DoRaySorting)); prev_out = qout
            cur = oracle_state(j + 1)
            fl = np.setxor1d(cur[1], qout); keep = ~np.isin(qin, fl)
            beyond, eq, worst = glref_check._compare_records(cur[0][qin][keep], rout[qin][keep])
            tot["rays"] += len(qin); tot["flips"] += int(len(fl)); tot["beyond_tol"] += int(beyond.sum()); tot["max_rel"] = max(tot["max_rel"], worst); tot["stages"] += 1
            prev = cur
        pt.close()
        rep[name] = tot
    print(json.dumps(rep))


def summary_only(names):
    """Recomputes tests/golden/glref/summary.json from the COMMITTED fixtures and today's oracle (no llvmpipe needed): what the gate of
    tests/glref_check.py measures — pure relative errors, flips, pixels beyond tolerance — per case and stage."""
    from oracle import oracle as O
    from idkengine_amd import gputypes as T
    import configs
    import glref_cases
    import glref_check
    B = O.OracleBuilder()
    summary = {}
    for name in names or list(glref_cases.GLREF_CASES):
        fac, camf, w, h, ov = glref_cases.GLREF_CASES[name]
        sc = fac(B); cam = camf(w, h)
        st = configs.apply_settings(T.Settings.default(), ov)
        fx = np.load(os.path.join(OUT, name + ".npz"))

        def oracle_state(d):
            o = O.OraclePathTracer(sc, w, h); o.set_camera(cam); configs.apply_settings(o.settings, ov)
            o.settings.RayDepth = d; o.settings.SamplesPerPixel = 1
            o.render()
            r, q = o.rays(), o.alive_queue(); o.close()
            return r, q
        o = O.OraclePathTracer(sc, w, h); o.set_camera(cam); configs.apply_settings(o.settings, ov); o.render()
        rep = glref_check.check_case(fx, oracle_state, dict(image=o.image(0), counts=o.stats()["alive_counts"], albedo=o.image(1) if st.OutputAOVs else None,
                                                           normal=o.image(2) if st.OutputAOVs else None), strict=False, name=name)
        o.close()
        rep["gate"] = {"rel_tol": glref_check.REL_TOL, "abs_floor": glref_check.ABS_FLOOR, "max_outlier_frac": glref_check.MAX_OUTLIER_FRAC, "allow": glref_check.FREE_RUN_ALLOW.get(name)}
        summary[name] = rep
        print(name, json.dumps(rep), flush=True)
    json.dump(summary, open(os.path.join(OUT, "summary.json"), "w"), indent=1, sort_keys=True)


def main(names, check=False):
    from oracle.glref import glref as G
    from oracle import oracle as O
    from idkengine_amd import gputypes as T
    import configs
    import glref_cases
    import glref_check
    os.makedirs(OUT, exist_ok=True)
    B = O.OracleBuilder()
    summary_path = os.path.join(OUT, "summary.json")
    summary = json.load(open(summary_path)) if os.path.exists(summary_path) else {}
    failed = []
    for name in names or list(glref_cases.GLREF_CASES):
        fac, camf, w, h, ov = glref_cases.GLREF_CASES[name]
        sc = fac(B); cam = camf(w, h)
        st = configs.apply_settings(T.Settings.default(), ov)
        depth, spp = int(st.RayDepth), int(st.SamplesPerPixel)
        out = {"mesa": np.frombuffer(G.gl().glref_info(), np.uint8), "depth": depth, "spp": spp, "width": w, "height": h}
        # ---- free run: the reference's own frame(s)
        pt = G.ReferencePathTracer(sc, w, h, st); pt.set_camera(cam)
        for _ in range(spp):
            pt.render()
        out["free_image"] = pt.image(0)
        if st.OutputAOVs:
            out["free_albedo"] = pt.image(1); out["free_normal"] = pt.image(2)
        out["free_counts"] = np.array(pt.alive_counts, np.uint32)
        out["free_final_alive"] = pt.final_alive
        pt.close()
        # ---- stage vectors (sample 0)
        def oracle_state(d):
            o = O.OraclePathTracer(sc, w, h); o.set_camera(cam); configs.apply_settings(o.settings, ov)
            o.settings.RayDepth = d; o.settings.SamplesPerPixel = 1
            o.render()
            r, q = o.rays(), o.alive_queue(); o.close()
            return r, q
        st1 = configs.apply_settings(T.Settings.default(), ov); st1.RayDepth = 1; st1.SamplesPerPixel = 1
        pt = G.ReferencePathTracer(sc, w, h, st1); pt.set_camera(cam)
        pt.render()
        out["fh_rays"] = pt.rays(T.GpuWavefrontRay); out["fh_queue"] = pt.final_alive
        pt.accumulated = 0                                   # the stage vectors are all of sample 0
        if not st.Gpu.DoDebugBVHTraversal:
            prev_out = None
            for j in range(1, depth):
                rin, qin = oracle_state(j)
                if st.DoRaySorting and j > 1 and not np.array_equal(prev_out, qin):
                    print(f"  {name}: bounce {j} not comparable from forced inputs (the reference's bounce {j - 1} queue differs; no sort keys for it)")
                    break
                rout, qout = pt.run_nhit_from(rin, qin, j, sort_first=bool(st.DoRaySorting))
                prev_out = qout
                out[f"in_hash_{j}"] = state_hash(rin, qin)
                out[f"out_rays_{j}"] = rout[qin]; out[f"out_queue_{j}"] = qout
        pt.close()
        if check:
            fx = np.load(os.path.join(OUT, name + ".npz"))
            bad = [k for k in out if k != "mesa" and not (k in fx and np.asarray(out[k]).tobytes() == np.asarray(fx[k]).tobytes())]
            bad += [k for k in fx.files if k not in out]
            print(name, "reproduced" if not bad else f"DIFFERS in {bad}", flush=True)
            if bad:
                failed.append(name)
            continue
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        # ---- summary against the oracle of today
        fx = np.load(os.path.join(OUT, name + ".npz"))
        o = O.OraclePathTracer(sc, w, h); o.set_camera(cam); configs.apply_settings(o.settings, ov); o.render()
        rep = glref_check.check_case(fx, oracle_state, dict(image=o.image(0), counts=o.stats()["alive_counts"], albedo=o.image(1) if st.OutputAOVs else None,
                                                           normal=o.image(2) if st.OutputAOVs else None), strict=False)
        o.close()
        summary[name] = rep
        print(name, json.dumps(rep), flush=True)
    if check:
        sys.exit(1 if failed else 0)
    json.dump(summary, open(summary_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    args = sys.argv[1:]
    if "--defect-d1" in args:
        defect_d1()
    elif "--eight-bit" in args:
        eight_bit_report()
    elif "--extended" in args:
        extended()
    elif "--summary" in args:
        summary_only([a for a in args if not a.startswith("--")])
    elif "--queries" in args:
        os.makedirs(OUT, exist_ok=True)
        sys.exit(1 if make_query_and_shadow_vectors(check="--check" in args) else 0)
    else:
        main([a for a in args if not a.startswith("--")], check="--check" in args)
