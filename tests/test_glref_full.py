"""Whole frames at BASELINE size against the reference's own shaders (tests/golden/glref_full/, minted by oracle/glref/make_full_vectors.py from
/root/reference's GLSL on Mesa llvmpipe): 1920x1080 on the 1M-triangle scenes bench.py times.  At generation every ray of every stage was compared with
the oracle (tests/golden/glref_full/summary.json); the fixture carries the sha256 of the compared state, the reference's records on a sample of the rays,
and every ray on which the two differ.  Here: the oracle still produces the compared state (so the whole-frame comparison holds for today's oracle) and
agrees with the reference's sampled records; tests/test_gpu_glref_full.py does the same for the HIP path, stage by stage."""
import json
import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden")); sys.path.insert(0, HERE)
import configs  # noqa: E402
import glref_cases  # noqa: E402
import glref_check  # noqa: E402

FIXTURES = os.path.join(HERE, "golden", "glref_full")
ROOT = os.path.dirname(HERE)


def _live():
    return os.path.isdir("/root/reference/IDKEngine/Resource/Shaders") and os.path.exists("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so")


live = pytest.mark.skipif(not _live(), reason="needs /root/reference and Mesa llvmpipe (build container only)")
_SCENES = {}


def full_scene(key, builder):
    if key not in _SCENES:
        _SCENES[key] = glref_cases.FULL_SCENES[key](builder)
    return _SCENES[key]


def test_every_full_case_has_a_fixture_and_a_clean_summary():
    have = {f[:-4] for f in os.listdir(FIXTURES) if f.endswith(".npz")} - {"updates_1m"}
    assert have == set(glref_cases.FULL_CASES) and os.path.exists(os.path.join(FIXTURES, "updates_1m.npz"))
    summary = json.load(open(os.path.join(FIXTURES, "summary.json")))
    assert set(summary) == have
    for name, rep in summary.items():
        assert len(rep["stages"]) >= (1 if "debugcost" in name else 2), name
        for s in rep["stages"]:
            # whole-stage statistics of the generation run: every ray compared; what is not within the gate is listed as an exception (a few per million)
            assert s["exceptions"] <= glref_check.FULL_ALLOW[name], (name, s, glref_check.FULL_EXCEPTION_REASON)
            assert s["flips"] <= s["exceptions"] and s["beyond_tol"] <= s["exceptions"], (name, s)
            assert s["max_rel_within_tol"] <= glref_check.REL_TOL


@pytest.mark.parametrize("name", list(glref_cases.FULL_CASES))
def test_oracle_still_produces_the_state_compared_with_the_reference(name, oracle_mod, native_builder):
    O = oracle_mod
    skey, camf, w, h, ov = glref_cases.FULL_CASES[name]
    sc = full_scene(skey, native_builder); cam = camf(w, h)
    fx = np.load(os.path.join(FIXTURES, name + ".npz"))
    assert (int(fx["width"]), int(fx["height"])) == (w, h)

    def state_at(d):
        o = O.OraclePathTracer(sc, w, h); o.set_camera(cam); configs.apply_settings(o.settings, ov)
        o.settings.RayDepth = d; o.settings.SamplesPerPixel = 1
        o.render()
        r, q = o.rays().copy(), o.alive_queue().copy(); o.close()
        return r, q
    # the headline case stage by stage; the others on their last compared stage (one whole-frame oracle render each keeps the CPU suite short)
    rep = glref_check.check_full_case(fx, state_at, strict=True, only_last=(name != "full_headline_d2"), name=name)
    assert all(s["state_is_the_compared_state"] and s["beyond_tol_in_sample"] == 0 for s in rep["stages"]), rep
    if "free_image_hash" in fx:  # the reference's whole free-running frame (incl. FinalDraw, accumulated samples) against the oracle's Result image
        o = O.OraclePathTracer(sc, w, h); o.set_camera(cam); configs.apply_settings(o.settings, ov)
        for _ in range(int(fx["free_samples"])):
            o.render()
        glref_check.check_full_frame(fx, o.image(0), name=name); o.close()
    if "ref_cost_sum" in fx:     # the reference's own traversal-cost counter against the oracle's P / T counters (the numerator of the roofline)
        o = O.OraclePathTracer(sc, w, h); o.set_camera(cam); configs.apply_settings(o.settings, ov); o.enable_counters(True); o.render(); st = o.stats(); o.close()
        glref_check.check_traversal_cost(fx, st["node_pair_visits"], st["triangle_tests"])


@live
def test_live_headline_frame_regenerates_the_committed_fixture():
    """Re-runs the reference's shaders on the whole headline frame (llvmpipe, ~15 s) and demands the committed fixture bit for bit: hashes of the compared
    states, the sampled reference records, the listed exceptions and their brute-force verdicts."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "glref", "make_full_vectors.py"), "--check", "full_headline_d2"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "full_headline_d2 reproduced" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def _mfv():
    sys.path.insert(0, os.path.join(ROOT, "oracle", "glref"))
    import make_full_vectors
    return make_full_vectors


def _sha(a):
    import hashlib
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def test_refit_and_skinning_at_config5_size_match_reference_shaders(oracle_mod, native_builder):
    """BASELINE configs[5] stand-in at full size (refittable soup-1M, 3 M vertices): Shaders/BLASRefit/compute.glsl and Shaders/Skinning/compute.glsl were run
    on llvmpipe over the whole scene (tests/golden/glref_full/updates_1m.npz: sha256 of all 1.85 M refitted nodes and of all 3 M skinned positions, plus
    samples).  The oracle's and the product library's BLAS.Refit and the binary32 restatement of the skinning arithmetic reproduce them bit for bit."""
    from test_gpu_scene_updates import _skin_numpy
    fx = np.load(os.path.join(FIXTURES, "updates_1m.npz"))
    sc, moved, un, joints = _mfv().full_update_inputs(native_builder)
    assert len(sc.blas_nodes) == int(fx["node_count"])
    for builder in (native_builder, oracle_mod.OracleBuilder()):
        nodes = builder.refit(sc.blas_nodes, moved, sc.blas_triangles)
        assert np.array_equal(_sha(nodes), fx["refit_nodes_hash"])
        assert nodes[::509].tobytes() == fx["refit_nodes_sample"].tobytes()
    pos = _skin_numpy(un, joints)
    assert np.array_equal(_sha(pos), fx["skin_positions_hash"]) and bool(fx["skin_prev_is_input"][0])
