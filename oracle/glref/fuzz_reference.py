"""Randomised differential test of the ORACLE against the REFERENCE's own shaders (Mesa llvmpipe): the cases tools/fuzz_parity.py draws — random scenes
(several BLASes, every material kind, textures, mesh biases, instances, lights, constant or per-face sky), cameras (inside / outside, lens), odd frame sizes,
RayDepth 1..7, sorting, Russian roulette, lights, AOVs, TLAS — run through FirstHit / NHit / CountingSort of /root/reference, stage by stage from identical
inputs (sort keys included), and compared with the oracle under the gate of tests/glref_check.py.  tools/fuzz_parity.py compares the HIP path with the oracle on
the same cases bit for bit on the MI355X; this closes the triangle for them.  Test infrastructure; build container only.

    python oracle/glref/fuzz_reference.py [cases=100] [first_seed=0] [out.json]"""
import json
import os
import sys
import time
import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))


def run_case(seed, builder, G, O, T, configs, glref_check, draw_case):
    os.environ["FUZZ_P_BIG"] = "0"                      # (the > 2^21-triangle draw has its own whole-frame case: tests/golden/glref_full)
    sc, cam, w, h, ov, st, _opts, _frames, _batch, nb = draw_case(seed, builder)
    depth = int(st.RayDepth)

    textured = bool(len(sc.textures))

    def oracle_state(d, uv_stage=-1, uv_override=None, uv_dump=None):
        o = O.OraclePathTracer(sc, w, h); o.set_camera(cam); configs.apply_settings(o.settings, ov); o.settings.RayDepth = d; o.settings.SamplesPerPixel = 1
        if uv_stage >= 0:
            o.set_uv_hooks(uv_stage, uv_override, uv_dump)
        o.render(); r, q, k = o.rays().copy(), o.alive_queue().copy(), o.alive_keys().copy(); o.close()
        return r, q, k
    st1 = configs.apply_settings(T.Settings.default(), ov); st1.RayDepth = 1; st1.SamplesPerPixel = 1
    sky = sc.sky_faces
    distinct_1x1 = sky is not None and sky.shape[1] == 1 and not (np.asarray(sky) == np.asarray(sky)[0]).all()      # the C-ABI's "constant colour per face": unfiltered by definition
    pt = G.ReferencePathTracer(sc, w, h, st1, sky_nearest=distinct_1x1, dump_uv=textured); pt.set_camera(cam); pt.render()
    ref_rays, ref_q = pt.rays(T.GpuWavefrontRay), np.asarray(pt.final_alive, np.uint32); pt.accumulated = 0
    cur = oracle_state(1)
    rep = {"seed": seed, "size": [w, h], "triangles": int(len(sc.blas_triangles)), "settings": ov, "textured": bool(len(sc.textures)), "stages": 0, "rays": 0, "flips": 0, "beyond_tol": 0, "key_diffs": 0, "max_rel": 0.0, "beyond_by_field": {},
           # textured cases, stage by stage from identical inputs INCLUDING the texture coordinate of every tap (A9: the reference's own interpolated uv fed to the checker's stage)
           "pinned": {"stages": 0, "rays": 0, "taps": 0, "flips": 0, "beyond_tol": 0, "beyond_tol_with_a_tap": 0, "max_rel": 0.0, "uv_max_diff": 0.0, "beyond_by_field": {}}}

    def add_pinned(stage, ids, ref, rq, depth_now):
        """The same stage with the checker reading the reference's texture coordinates: what is left is shading arithmetic on identical taps.  Also: how far the two
        interpolated coordinates were apart (relative to max(1, |uv|)) — the input difference that the noise textures' contrast multiplies in the unpinned comparison."""
        uv_ref = pt.take_uv_dump()
        own = np.full_like(uv_ref, np.nan)
        cand_r, cand_q, _k = oracle_state(depth_now, stage, uv_ref, own)
        both = ~np.isnan(uv_ref[:, 0]) & ~np.isnan(own[:, 0])
        P = rep["pinned"]
        if both.any():
            P["uv_max_diff"] = max(P["uv_max_diff"], float((np.abs(uv_ref[both].astype(np.float64) - own[both]) / np.maximum(1.0, np.abs(uv_ref[both]).max(axis=1, keepdims=True))).max()))
        flips = np.setxor1d(cand_q, rq)
        keep = ~np.isin(ids, flips)
        beyond, _eq, worst = glref_check._compare_records(cand_r[ids][keep], ref[keep])
        P["stages"] += 1; P["rays"] += int(len(ids)); P["taps"] += int(both.sum()); P["flips"] += int(len(flips)); P["beyond_tol"] += int(beyond.sum()); P["max_rel"] = max(P["max_rel"], worst)
        if beyond.any():      # which fields: what a texture feeds directly is Throughput / Radiance (base colour, emission, transmission tint); Origin / PackedDirection beyond the gate is geometry
            a, b = cand_r[ids][keep], ref[keep]
            tapped = both[ids][keep]
            for f in glref_check.FIELDS:
                e = glref_check.rel_err(np.asarray(a[f], np.float32).reshape(len(a), -1).astype(np.float64), np.asarray(b[f], np.float32).reshape(len(b), -1).astype(np.float64), f)
                e = np.where(np.isnan(e), np.inf, e)
                if e.size and e.max() > glref_check.REL_TOL:
                    P["beyond_by_field"][f] = max(P["beyond_by_field"].get(f, 0.0), float(e.max()))
            P["beyond_tol_with_a_tap"] += int((beyond & tapped).sum())

    def add(ids, cand, ref, cand_q, rq):
        flips = np.setxor1d(cand_q, rq)
        keep = ~np.isin(ids, flips)
        beyond, _eq, worst = glref_check._compare_records(cand[keep], ref[keep])
        rep["stages"] += 1; rep["rays"] += int(len(ids)); rep["flips"] += int(len(flips)); rep["beyond_tol"] += int(beyond.sum()); rep["max_rel"] = max(rep["max_rel"], worst)
        if beyond.any():      # how far beyond: the worst error of the rays that miss the gate, per field
            c, r = cand[keep][beyond], ref[keep][beyond]
            for f in glref_check.FIELDS:
                x = np.asarray(c[f], np.float64).reshape(len(c), -1); y = np.asarray(r[f], np.float64).reshape(len(c), -1)
                e = glref_check.rel_err(x, y, f); e = e[np.isfinite(e)]
                if e.size and e.max() > glref_check.REL_TOL:
                    rep["beyond_by_field"][f] = max(rep["beyond_by_field"].get(f, 0.0), float(e.max()))
    add(np.arange(w * h), cur[0], ref_rays, cur[1], ref_q)
    if textured:
        add_pinned(0, np.arange(w * h), ref_rays, ref_q, 1)
    for j in range(1, depth):
        rin, qin = cur[0], cur[1]
        if len(qin) == 0:
            break
        rout, qout = pt.run_nhit_from(rin, qin, j, sort_first=bool(st.DoRaySorting), keys=(cur[2] if (st.DoRaySorting and j > 1) else None))
        nxt = oracle_state(j + 1)
        if st.DoRaySorting and pt.last_out_keys is not None:
            kr = np.zeros(w * h, np.int64) - 1; kr[np.asarray(qout, np.int64)] = pt.last_out_keys
            ko = np.zeros(w * h, np.int64) - 1; ko[nxt[1].astype(np.int64)] = nxt[2]
            both = np.intersect1d(qout, nxt[1]); rep["key_diffs"] += int((kr[both] != ko[both]).sum())
        add(qin, nxt[0][qin], rout[qin], nxt[1], np.asarray(qout, np.uint32))
        if textured:
            add_pinned(j, qin, rout[qin], np.asarray(qout, np.uint32), j + 1)
        cur = nxt
    pt.close()
    return rep


def main():
    from oracle.glref import glref as G
    from oracle import oracle as O
    from idkengine_amd import gputypes as T
    from idkengine_amd.bvh import NativeBuilder
    import configs
    import glref_check
    from fuzz_parity import draw_case
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    out = sys.argv[3] if len(sys.argv) > 3 else None
    seeds = [int(x) for x in os.environ["FUZZ_SEEDS"].split(",")] if os.environ.get("FUZZ_SEEDS") else list(range(first, first + cases))
    if os.environ.get("FUZZ_SAMPLER") == "llvmpipe":        # the checker's sampler in llvmpipe's arithmetic (ref_pathtracer.cpp ref_set_sampler_mode): what part of the textured cases' residue is the sampler's freedom
        O.lib().ref_set_sampler_mode(1)
    builder = NativeBuilder()
    t0 = time.time(); reps = []
    for seed in seeds:
        r = run_case(seed, builder, G, O, T, configs, glref_check, draw_case)
        reps.append(r)
        print(f"seed {seed}: {r['size'][0]}x{r['size'][1]} tris {r['triangles']} {r['settings']} stages {r['stages']} rays {r['rays']}: flips {r['flips']} beyond {r['beyond_tol']} {r['beyond_by_field'] or ''} key diffs {r['key_diffs']} max rel {r['max_rel']:.2e}", flush=True)
    tot = {"cases": len(seeds), "first_seed": first, "stages": sum(r["stages"] for r in reps), "rays": sum(r["rays"] for r in reps), "flips": sum(r["flips"] for r in reps),
           "beyond_tol": sum(r["beyond_tol"] for r in reps), "key_diffs": sum(r["key_diffs"] for r in reps), "max_rel_within_tol": max(r["max_rel"] for r in reps),
           "worst_error_of_the_rays_beyond_tolerance_by_field": {f: max(r["beyond_by_field"].get(f, 0.0) for r in reps) for f in glref_check.FIELDS if any(f in r["beyond_by_field"] for r in reps)},
           "cases_with_a_flip_or_a_value_beyond_tolerance": [r["seed"] for r in reps if r["flips"] or r["beyond_tol"]], "seconds": round(time.time() - t0, 1),
           "textured_cases": sum(1 for r in reps if r["textured"]), "beyond_tol_in_textured_cases": sum(r["beyond_tol"] for r in reps if r["textured"]), "beyond_tol_in_untextured_cases": sum(r["beyond_tol"] for r in reps if not r["textured"]),
           "worst_throughput_or_radiance_error_beyond_tolerance": max([max(r["beyond_by_field"].get("Throughput", 0.0), r["beyond_by_field"].get("Radiance", 0.0)) for r in reps] + [0.0]),
           "textured_stages_from_identical_taps": dict({k: (max if k in ("max_rel", "uv_max_diff") else sum)(r["pinned"][k] for r in reps) for k in ("stages", "rays", "taps", "flips", "beyond_tol", "beyond_tol_with_a_tap", "max_rel", "uv_max_diff")},
                                                       worst_error_beyond_tolerance_by_field={f: max(r["pinned"]["beyond_by_field"].get(f, 0.0) for r in reps) for f in glref_check.FIELDS if any(f in r["pinned"]["beyond_by_field"] for r in reps)}),
           "gate": {"rel_tol": glref_check.REL_TOL, "abs_floor": glref_check.ABS_FLOOR}, "checker_sampler": os.environ.get("FUZZ_SAMPLER") or "gl-spec"}
    print(json.dumps(tot))
    if out:
        json.dump({"total": tot, "cases": reps}, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
