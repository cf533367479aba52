"""k_trace2s (csrc/kernels_trace_split.hpp): once a launch's work list is empty, idle lanes take over the bottom stack entry of a busy lane and traverse it as a
piece of the same ray; the root lane combines the pieces or — when two hits lie within rounding distance of each other — traces the ray again by itself.
The hit every ray reports must be the sequential traversal's (BVHIntersect.glsl:27-105): image, every ray record, alive queue and primary hits equal the
oracle's bit for bit with the kernel forced on every launch ("split" 2), with every split ray traced again ("split" 3), on scenes full of shared edges and
coincident box faces (the Cornell box, the atrium), inside a triangle soup, with sphere lights in front of and behind the geometry."""
import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden")); sys.path.insert(0, HERE)
import configs  # noqa: E402
from idkengine_amd import scenes as S  # noqa: E402
from gpu_helpers import bits, oracle_render  # noqa: E402

pytestmark = pytest.mark.gpu


def _render(sc, cam, w, h, opts, frames, batch, **ov):
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd import gputypes as T
    pt = PathTracer(w, h, settings=configs.apply_settings(T.Settings.default(), ov))
    for k, v in opts.items():
        pt.set_option(k, v)
    pt.UploadScene(sc); pt.SetCamera(cam); pt.enable_primary_hit_capture(True); pt.set_max_batch(batch)      # (no visit counters: the counting build never splits)
    for _ in range(frames):
        pt.Compute()
    return pt


def _same(pt, o):
    assert (bits(pt.Result) == bits(o.image(0))).all()
    gt, gtri, gb = pt.primary_hits(); ot, otri, ob = o.primary_hits()
    assert (gtri == otri).all() and (bits(gt) == bits(ot)).all() and (bits(gb) == bits(ob)).all()
    assert pt.rays().tobytes() == o.rays().tobytes()
    assert (pt.alive_queue() == o.alive_queue()).all()
    assert pt.stats()["rays_traced"] == o.stats()["rays_traced"]


CASES = ["cornell", "cornell_lights", "soup_inside", "soup_outside", "atrium", "lucy"]


@pytest.mark.parametrize("case", CASES)
def test_split_rays_report_the_sequential_hit(case, oracle_mod, native_builder):
    ov = dict(RayDepth=4)
    if case == "cornell":
        sc, w, h = S.cornell_scene(native_builder, variant="mixed"), 160, 120; cam = S.cornell_camera(w, h)
    elif case == "cornell_lights":
        sc, w, h = S.cornell_scene(native_builder, variant="mixed"), 128, 96; cam = S.cornell_camera(w, h)
        from idkengine_amd import gputypes as T
        lights = np.zeros(2, T.GpuLight); lights["Position"] = [(0.0, 0.6, 0.2), (0.3, -0.2, 1.5)]; lights["Radius"] = [0.15, 0.1]; lights["Color"] = [(6.0, 5.0, 4.0), (1.0, 2.0, 6.0)]
        sc.lights = lights; ov["DoTraceLights"] = 1
    elif case == "soup_inside":
        sc, w, h = S.soup_scene(40000, native_builder, seed=8), 200, 120; cam = S.Camera(w, h, position=(0.0, 0.0, 0.0), view_dir=(0.2, 0.1, -1.0))
    elif case == "soup_outside":
        sc, w, h = S.soup_scene(40000, native_builder, seed=9), 200, 120; cam = S.Camera(w, h)
    elif case == "atrium":
        sc, w, h = S.atrium_scene(30000, native_builder), 192, 108; cam = S.atrium_camera(w, h)
    else:
        m = np.load(os.path.join(HERE, "golden", "models", "lucy.npz"))
        p = m["positions"].astype(np.float32); i = m["indices"].astype(np.uint32).reshape(-1, 3)
        tp = p[i]; pp, ii, nrm, tan = S.flat_shaded(tp)
        sc = S.assemble([{"meshes": [S.MeshInput(pp, ii, S.make_material((0.8, 0.7, 0.6, 1.0)), nrm, tan)]}], native_builder); w, h = 120, 160
        c = 0.5 * (p.min(0) + p.max(0)); ext = float((p.max(0) - p.min(0)).max())
        cam = S.Camera(w, h, position=(float(c[0]), float(c[1]), float(c[2] + 1.6 * ext)), fovy_deg=45.0)
    o = oracle_render(oracle_mod, sc, cam, w, h, frames=3, **ov)
    # one wave per CU and leaves tested one at a time: lanes run idle early, many rays are split; the default grid; several samples per launch
    for opts, batch in (({"split": 2, "trace_waves": 1, "leaf_min": 1}, 1), ({"split": 3, "trace_waves": 2}, 1), ({"split": 2, "split_donor": 0}, 3), ({"split": 3, "grab_unit_log2": 6, "split_donor": 0}, 3), ({"split": 1}, 1)):
        pt = _render(sc, cam, w, h, opts, 3, batch, **ov)
        _same(pt, o)
        pt.Dispose()
    o.close()


def test_split_is_not_used_on_a_tree_that_is_not_nested(oracle_mod, native_builder):
    """A host may patch node boxes (idkptUpdateBuffer); a child box that pokes out of its parent's is still a valid tree for the traversal — and takes away what the
    exactness argument of the split kernel needs, so such a scene keeps the plain kernel.  Results equal the oracle's on the patched tree either way."""
    from idkengine_amd import gputypes as T
    sc = S.soup_scene(6000, native_builder, seed=3); w, h = 120, 80; cam = S.Camera(w, h, position=(0.0, 0.0, 0.0), view_dir=(0.1, 0.2, -1.0))
    nodes = sc.blas_nodes.copy()
    inner = [n for n in range(2, len(nodes)) if nodes["TriCount"][n] == 0 and nodes["TriStartOrChild"][n] != 0][:40]
    for n in inner:                                                  # grow the left child beyond its parent (conservative: nothing is lost, more is visited)
        c = int(nodes["TriStartOrChild"][n]); nodes["Min"][c] -= np.float32(0.05); nodes["Max"][c] += np.float32(0.05)
    import copy
    sc2 = copy.copy(sc); sc2.blas_nodes = nodes
    o = oracle_render(oracle_mod, sc2, cam, w, h, frames=2, RayDepth=3)
    for how in ("upload", "patch"):
        pt = _render(sc2 if how == "upload" else sc, cam, w, h, {"split": 2, "trace_waves": 1}, 0, 1, RayDepth=3)
        if how == "patch":
            pt.UpdateBuffer(T.IDKPT_BUF_BLAS_NODES, nodes)
        pt.Compute(); pt.Compute()
        _same(pt, o)
        pt.Dispose()
    o.close()
