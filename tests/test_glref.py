"""The oracle PINNED against the reference itself: tests/golden/glref/*.npz hold outputs of the reference's own GLSL
path-tracer shaders (FirstHit / NHit / FinalDraw / CountingSort, read from /root/reference and executed by Mesa llvmpipe
through oracle/glref/).  Here the CPU oracle is compared with them — every bounce from identical inputs, plus the
reference's free-running frame; tests/test_gpu_glref.py does the same for the HIP path.

The fixtures travel; the generator needs /root/reference + Mesa swrast and is exercised by the `live` tests, which skip
elsewhere (e.g. on the GPU box)."""
import json
import os
import subprocess
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden")); sys.path.insert(0, HERE)
import configs  # noqa: E402
import glref_cases  # noqa: E402
import glref_check  # noqa: E402

FIXTURES = os.path.join(HERE, "golden", "glref")


def _live():
    return os.path.isdir("/root/reference/IDKEngine/Resource/Shaders") and os.path.exists("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so")


live = pytest.mark.skipif(not _live(), reason="needs /root/reference and Mesa llvmpipe (build container only)")


def test_every_case_has_a_fixture():
    have = {f[:-4] for f in os.listdir(FIXTURES) if f.endswith(".npz")} - {"queries", "shadows", "updates"}
    assert have == set(glref_cases.GLREF_CASES)
    assert os.path.exists(os.path.join(FIXTURES, "queries.npz")) and os.path.exists(os.path.join(FIXTURES, "shadows.npz"))
    summary = json.load(open(os.path.join(FIXTURES, "summary.json")))
    assert set(summary) == have


@pytest.mark.parametrize("name", list(glref_cases.GLREF_CASES))
def test_oracle_matches_reference_shaders(name, oracle_mod):
    from idkengine_amd import gputypes as T
    O = oracle_mod
    fac, camf, w, h, ov = glref_cases.GLREF_CASES[name]
    sc = fac(O.OracleBuilder()); cam = camf(w, h)
    fx = np.load(os.path.join(FIXTURES, name + ".npz"))
    assert (int(fx["width"]), int(fx["height"])) == (w, h)

    def state_at(d):
        o = O.OraclePathTracer(sc, w, h); o.set_camera(cam); configs.apply_settings(o.settings, ov)
        o.settings.RayDepth = d; o.settings.SamplesPerPixel = 1
        o.render()
        r, q = o.rays(), o.alive_queue(); o.close()
        return r, q
    o = O.OraclePathTracer(sc, w, h); o.set_camera(cam); configs.apply_settings(o.settings, ov); o.render()
    aov = bool(configs.apply_settings(T.Settings.default(), ov).OutputAOVs)
    final = dict(image=o.image(0), counts=o.stats()["alive_counts"], albedo=o.image(1) if aov else None, normal=o.image(2) if aov else None)
    o.close()
    rep = glref_check.check_case(fx, state_at, final, strict=True, name=name)
    # what the committed fixtures show today (tests/golden/glref/summary.json): not one flipped decision, not one value beyond tolerance
    assert all(s["flips"] == 0 and s["beyond_tol"] == 0 and s["queue_identical"] for s in rep["stages"]), rep


def _mv():
    sys.path.insert(0, os.path.join(ROOT, "oracle", "glref"))
    import make_vectors
    return make_vectors


@pytest.mark.parametrize("use_tlas", [0, 1])
@pytest.mark.parametrize("any_hit", [0, 1])
@pytest.mark.parametrize("lights", [0, 1])
def test_oracle_ray_queries_match_reference_functions(oracle_mod, use_tlas, any_hit, lights):
    """oracle.trace_rays (what idkptTraceRays is held to) vs the reference's own TraceRay / TraceRayAny (BVHIntersect.glsl:183-411)."""
    fx = np.load(os.path.join(FIXTURES, "queries.npz"))
    sc = _mv().query_scene(oracle_mod.OracleBuilder())
    got = oracle_mod.trace_rays(sc, fx["rays"], any_hit=bool(any_hit), trace_lights=bool(lights), use_tlas=bool(use_tlas))
    rep = glref_check.check_query_hits(got, fx[f"hits_tlas{use_tlas}_any{any_hit}_lights{lights}"])
    assert rep["triangle_hits"] > 1000 and (not lights or rep["light_hits"] > 20)


def test_oracle_rt_shadows_match_reference_shader(oracle_mod):
    """oracle.trace_shadows (what idkptTraceShadows is held to) vs Shaders/ShadowsRayTraced/compute.glsl run on llvmpipe."""
    from idkengine_amd import scenes as S, gputypes as T
    mv = _mv()
    fx = np.load(os.path.join(FIXTURES, "shadows.npz"))
    w, h = mv.SHADOW_SIZE
    cam = S.cornell_camera(w, h)
    for variant, tlas in mv.SHADOW_CONFIGS:
        sc = mv.shadow_scene(oracle_mod.OracleBuilder(), variant)
        depth, normal = fx[f"depth_{variant}_{tlas}"], fx[f"normal_{variant}_{tlas}"]
        for light, samples, noise in mv.SHADOW_PARAMS:
            p = T.ShadowParams.make(cam.inv_proj_view, w, h, light_index=light, samples=samples, noise_index=noise, jitter=(0.0005, -0.0003))
            got = oracle_mod.trace_shadows(sc, p, depth, normal, visibility=np.full((h, w), np.float32(-3.0)), use_tlas=bool(tlas))
            ref = fx[f"vis_{variant}_{tlas}_{light}_{samples}_{noise}"]
            glref_check.check_shadow_image(got, ref)
            assert (ref == 1.0).any() and (ref == 0.0).any() and (ref == -3.0).any()


def test_oracle_refit_and_skinned_positions_match_reference_shaders(oracle_mod):
    """BLAS.Refit restatement vs Shaders/BLASRefit/compute.glsl (node arrays bit for bit) and the binary32 restatement of
    Shaders/Skinning/compute.glsl's position path vs the shader (bit for bit)."""
    from test_gpu_scene_updates import _skin_numpy
    mv = _mv()
    fx = np.load(os.path.join(FIXTURES, "updates.npz"))
    B = oracle_mod.OracleBuilder()
    sc, moved, un, joints, sk = mv.update_inputs(B)
    assert B.refit(sc.blas_nodes, moved, sc.blas_triangles).tobytes() == fx["refit_nodes"].tobytes()
    n = sk["count"]
    want = _skin_numpy(un[sk["input_offset"]:sk["input_offset"] + n], joints[sk["joint_offset"]:])
    assert want.tobytes() == fx["skin_positions"].tobytes()
    assert fx["skin_prev_positions"].tobytes() == moved[sk["output_offset"]:sk["output_offset"] + n].tobytes() and bool(fx["skin_untouched_ok"][0])


@live
def test_live_query_and_shadow_fixtures_are_reproducible():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "glref", "make_vectors.py"), "--queries", "--check"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr


@live
def test_live_fixtures_are_reproducible():
    """Re-runs the reference's shaders on llvmpipe for three cases and demands the committed fixtures bit for bit
    (separate process: Mesa brings its own LLVM, keep it away from this one)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "glref", "make_vectors.py"), "--check", "cornell_mixed_d5", "helmet_sort_d4", "cornell_textured_aov_d5"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr


@live
def test_live_extended_frames_have_no_flipped_decision():
    """Three larger frames that are not stored as fixtures (interior soup 256x144 depth 5, atrium 192x108 depth 4, Lucy 256x320 depth 4 with ray
    sorting): about 210 000 rays compared bounce by bounce from identical inputs, reference shaders on llvmpipe against the oracle."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "glref", "make_vectors.py"), "--extended"], capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout + r.stderr
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert len(rep) == 3
    for name, t in rep.items():
        assert t["rays"] > 20000 and t["flips"] == 0 and t["beyond_tol"] == 0 and t["max_rel"] < glref_check.REL_TOL, (name, t)


@live
def test_live_reference_defect_d1_reorder_item_count():
    """Reference defect D1 (oracle/glref/glref.py ADAPTATIONS A7): in the reference's host order Reorder bounds-checks against the
    PREVIOUS bounce's count, so its tail invocations re-insert stale entries and the 'sorted' queue is not a permutation of the alive
    queue.  With PingPongIndex uploaded first (A7) the reference's shaders produce exactly the stable counting sort the oracle states."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "glref", "make_vectors.py"), "--defect-d1"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    bug, fixed = rep["reference_order"], rep["with_A7"]
    assert bug["reorder_item_count"] > bug["alive"] and bug["dispatched_invocations"] > bug["alive"] and not bug["is_permutation"]
    assert fixed["reorder_item_count"] == fixed["alive"] and fixed["is_permutation"] and fixed["equals_stable_sort"]


@live
def test_live_llvmpipe_8bit_textures_pinned_where_it_is_exact_bounded_where_it_is_not():
    """idkpt_texture's 8-bit formats (include/idkpt.h) decode per texel as GL 4.6 2.3.5.1 / 8.24 write it and filter in float.  llvmpipe agrees to an ulp where it evaluates at
    full precision — RGBA8 under GL_NEAREST, the configuration the fixture sampler_states_rgba8_d3 pins — filters RGBA8 under GL_LINEAR in 8-bit fixed point (about one 8-bit step off
    the float filter) and decodes sRGB8 with a polynomial (2.5 % off the transfer function): both are bounded here, and the decode itself is pinned by tests/test_oracle_kats.py."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "glref", "make_vectors.py"), "--eight-bit"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert all(v["textured_pixels"] > 2000 for v in rep.values()), rep
    assert rep["rgba8_nearest"]["pixels_beyond_1e-4"] == 0 and rep["rgba8_nearest"]["max_abs"] < 2e-7, rep
    assert 1e-3 < rep["rgba8_linear"]["max_abs"] < 1.2 / 255.0, rep                                   # 8-bit fixed-point filter weights and results
    assert 5e-4 < rep["srgb8_nearest"]["max_abs"] < 2e-3 and rep["srgb8_nearest"]["max_rel"] < 3e-2, rep   # polynomial sRGB decode
    assert rep["srgb8_linear"]["max_abs"] < 2e-3 and rep["srgb8_linear"]["max_rel"] < 3e-2, rep           # ... filtered in float after it


@live
def test_live_preprocessor_follows_the_reference():
    """AppInclude is include-once, AppInsert falls back to 0, unreferenced storage blocks are dropped (BBG/Source/Objects/Shader.cs:177-335)."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from oracle.glref import glref as G\n"
            "s = G.preprocess('PathTracing/NHit/compute.glsl', {'USE_TLAS': '1', 'BLAS_STACK_SIZE': '17'})\n"
            "assert s.count('struct GpuBlasNode') == 1 and 'AppInclude' not in s and 'AppInsert' not in s\n"
            "assert '#define USE_TLAS 1' in s and 'max(17, 1)' in s and '#define PATH_TRACER_DO_RAY_SORTING 0' in s\n"
            "assert 'drawElementsCmdSSBO' not in s and 'wavefrontPTSSBO' in s and 'tlasSSBO' in s\n"
            "assert s.startswith('#version 460 core') and '#define APP_SHADER_STAGE_COMPUTE 1' in s\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


@live
def test_live_reference_fuzz_first_seeds():
    """oracle/glref/fuzz_reference.py: the random cases of tools/fuzz_parity.py (scenes, materials, textures, instances, lights, sky, lens, settings) through the
    reference's own shaders, stage by stage from identical inputs.  The first 12 seeds (31 stages, 47 000 rays) here; 400 seeds are in profiles/r03_reference_fuzz.json."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "glref", "fuzz_reference.py"), "12", "0"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    tot = json.loads(r.stdout.strip().splitlines()[-1])
    assert tot["cases"] == 12 and tot["stages"] >= 30 and tot["rays"] > 40000
    assert tot["flips"] == 0 and tot["beyond_tol"] == 0 and tot["key_diffs"] == 0, tot


@live
def test_live_reference_fuzz_textured_cases_within_the_sampler_spread():
    """60 random cases (24 of them textured, 204 stages, 250 000 rays) through the reference's shaders: everything untextured inside the 1e-4 gate, no alive decision
    and no sort key differs, and what textured cases leave beyond the gate stays inside the named, bounded allowance for the spread of GL's bilinear filter weights
    (glref_check.SAMPLER_SPREAD_ALLOW: at most 1e-4 of the rays, at most 3e-3, textured cases only) — the finding of profiles/r03_reference_fuzz.json as a gate."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "glref", "fuzz_reference.py"), "60", "0"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    tot = json.loads(r.stdout.strip().splitlines()[-1])
    A = glref_check.SAMPLER_SPREAD_ALLOW
    assert tot["cases"] == 60 and tot["textured_cases"] >= 15 and tot["rays"] > 200000
    assert tot["flips"] <= A["alive_flips"] and tot["key_diffs"] <= A["key_diffs"], tot
    assert tot["beyond_tol_in_untextured_cases"] <= A["untextured_rays_beyond"], tot
    assert tot["beyond_tol_in_textured_cases"] <= A["max_fraction_of_rays"] * tot["rays"], tot
    assert tot["worst_throughput_or_radiance_error_beyond_tolerance"] <= A["max_throughput_or_radiance_error"], tot
    assert set(tot["worst_error_of_the_rays_beyond_tolerance_by_field"]) <= {"Throughput", "Radiance"}, tot       # (no origin, no direction: no other lobe, no other hit)
    # the per-stage gate of the textured cases is north_star's own number: every textured stage once more with the checker's taps at the texture coordinates the
    # REFERENCE interpolated (glref.py A9) — identical stage inputs down to the tap — leaves nothing beyond 1e-4 and no flip; the allowance above is for the
    # free-running comparison (each side's own interpolated coordinate) only
    P = tot["textured_stages_from_identical_taps"]
    assert P["stages"] >= 60 and P["taps"] > 20000, P
    assert P["beyond_tol"] == 0 and P["beyond_tol_with_a_tap"] == 0 and P["flips"] == 0 and P["max_rel"] <= glref_check.REL_TOL, P
    # (1 400 seeds, profiles/r06_reference_fuzz_1400_identical_taps.json: 1.25 M taps, 0 rays WITH a tap beyond 1e-4; 11 of 2.1 M rays of those stages beyond it, none of which sampled a
    #  texture — Origin / PackedDirection of grazing refractions, the untextured cases' rate of 4e-6 — and one lobe flip)


@live
def test_live_reference_fuzz_textured_residue_is_texture_contrast_not_the_sampler():
    """The same 60 cases twice more: (a) with the textures' contrast scaled to 0.02 about 0.5 — the textured cases then sit inside the plain 1e-4 gate like the untextured
    ones, no allowance; (b) at full contrast with the oracle's sampler switched to llvmpipe's own arithmetic (coordinate reduced to [0, 1) first, lerp as v0 + w (v1 - v0)) —
    the rays beyond the gate do not go away.  Together: what SAMPLER_SPREAD_ALLOW covers is (texture gradient) x (input differences inside the gate), not filter weights."""
    def run(env):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "glref", "fuzz_reference.py"), "60", "0"], capture_output=True, text=True, timeout=1500, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        return json.loads(r.stdout.strip().splitlines()[-1])
    flat = run({"FUZZ_TEX_CONTRAST": "0.02"})
    assert flat["textured_cases"] >= 15 and flat["beyond_tol"] == 0 and flat["flips"] == 0 and flat["key_diffs"] == 0, flat
    llvm = run({"FUZZ_SAMPLER": "llvmpipe"})
    assert llvm["checker_sampler"] == "llvmpipe" and llvm["beyond_tol_in_textured_cases"] >= 6 and llvm["beyond_tol_in_untextured_cases"] == 0, llvm
