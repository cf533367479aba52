"""The wide-node walk (idkengine_amd/csrc/wide_nodes.hpp) against the reference's BVH2 walk on the CPU, ray by ray (tools/wide_sim.cpp): every ray the wide walk
vouches for must carry the BVH2 walk's hit bit for bit — T, barycentrics, TriangleId — on scenes that provoke what it has to flag: axis-aligned flat geometry
(degenerate leaf boxes), triangles shared by a leaf pair and PreSplit fragments (the same triangle reachable through several leaves), coplanar duplicates."""
import os
import re
import subprocess
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sim(tmp_path_factory):
    exe = tmp_path_factory.mktemp("wide") / "wide_sim"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fopenmp", "-I", os.path.join(ROOT, "idkengine_amd", "csrc"), os.path.join(ROOT, "tools", "wide_sim.cpp"), "-o", str(exe)])
    return str(exe)


def _dump(sc, path):
    nodes = np.ascontiguousarray(sc.blas_nodes); t = sc.blas_triangles; p = sc.vertex_positions.reshape(-1, 3)
    tv = np.zeros((len(t), 3, 4), np.float32)
    tv[:, 0, :3] = p[t["X"]]; tv[:, 1, :3] = p[t["Y"]]; tv[:, 2, :3] = p[t["Z"]]
    with open(path, "wb") as f:
        f.write(np.int32([len(nodes), len(t)]).tobytes()); f.write(nodes.tobytes()); f.write(tv.tobytes())


def _run(sim, scene_file, view, w=240, h=135, policy=0, cap=255):
    r = subprocess.run([sim, scene_file, view, str(w), str(h), str(policy), str(cap)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    mism = [int(x) for x in re.findall(r"UNFLAGGED MISMATCHES (\d+)", r.stdout)]
    viol = [int(x) for x in re.findall(r"assumption violations (\d+)", r.stdout)]
    flagged = [float(x) for x in re.findall(r"flagged \d+ \(([\d.]+) %\)", r.stdout)]
    trips = [float(x) for x in re.findall(r"\(x([\d.]+)\)", r.stdout)]
    assert len(mism) >= 1 and sum(mism) == 0
    return dict(flagged=flagged, violations=viol, trips=trips, out=r.stdout)


def _cam(c):
    from idkengine_amd import scenes as S  # noqa: F401
    m = np.linalg.inv(c.view)
    fwd = -(m[2, :3])
    return "cam:%f,%f,%f,%f,%f,%f,%f" % (c.position[0], c.position[1], c.position[2], fwd[0], fwd[1], fwd[2], 2.0 * np.degrees(np.arctan(1.0 / c.proj[1, 1])))


@pytest.mark.parametrize("kind", ["cornell", "soup", "presplit", "atrium", "coplanar"])
def test_vouched_rays_equal_the_bvh2_walk(kind, sim, native_builder, tmp_path):
    from idkengine_amd import scenes as S
    if kind == "cornell":
        sc = S.cornell_scene(native_builder, variant="mixed"); view = _cam(S.cornell_camera(240, 135))
    elif kind == "soup":
        sc = S.soup_scene(20000, native_builder, seed=5); view = "interior"
    elif kind == "presplit":
        sc = S.presplit_scene(native_builder); view = _cam(S.presplit_camera(240, 135))
    elif kind == "atrium":
        sc = S.atrium_scene(40000, native_builder); view = "atrium"
    else:   # every triangle twice (two ids, the same plane, the same t): the reference keeps whichever its walk meets first -> the wide walk must not vouch for those rays
        tp = S.soup_triangles(3000, seed=9, extent=3.0, edge=0.4)
        p, i, nrm, tan = S.flat_shaded(np.concatenate([tp, tp]))
        sc = S.assemble([{"meshes": [S.MeshInput(p, i, S.make_material((0.7, 0.6, 0.5, 1.0)), nrm, tan)]}], native_builder); view = "cam:0,0,9,0,0,-1,60"
    f = str(tmp_path / "scene.bin"); _dump(sc, f)
    res = _run(sim, f, view)
    assert sum(res["violations"]) == 0
    if kind in ("coplanar", "presplit"):
        assert res["flagged"][0] > 5.0, res["out"]          # the duplicates / the PreSplit fragments (three scene-spanning triangles) really are met — and handed to the exact kernel
    else:
        assert max(res["flagged"]) < 2.0, res["out"]
    assert res["trips"][0] < 0.8                            # fewer dependent round trips than the BVH2 walk


def test_short_stack_flags_instead_of_guessing(sim, native_builder, tmp_path):
    from idkengine_amd import scenes as S
    sc = S.soup_scene(20000, native_builder, seed=5)
    f = str(tmp_path / "scene.bin"); _dump(sc, f)
    res = _run(sim, f, "interior", cap=6)
    assert res["flagged"][0] > 1.0                          # rays that need more rows are flagged, every other ray is still exact
