"""The C-ABI itself: a plain C host, error paths."""
import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, HERE)
import configs  # noqa: E402,F401
from idkengine_amd import scenes as S  # noqa: E402,F401
from gpu_helpers import bits, gpu_render, oracle_render, assert_equal, read_device_image  # noqa: E402,F401

pytestmark = pytest.mark.gpu


def test_plain_c_host_matches_python_host(native_builder, tmp_path):
    """The boundary is a C ABI: a plain C11 program (tests/c_driver/abi_driver.c; gcc, include/idkpt.h, -lidkpt; no Python, torch or
    C++ on its side) uploads the same arrays, renders, and must produce the same bits and counters as the Python host."""
    import subprocess
    root = os.path.dirname(HERE)
    exe = str(tmp_path / "abi_driver")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-I", os.path.join(root, "include"), os.path.join(HERE, "c_driver", "abi_driver.c"),
                           "-L", os.path.join(root, "idkengine_amd"), "-lidkpt", "-Wl,-rpath," + os.path.join(root, "idkengine_amd"), "-o", exe])
    sc = S.cornell_scene(native_builder, "mixed", True); w, h = 96, 64; cam = S.cornell_camera(w, h)
    for name in ("blas_nodes", "blas_triangles", "blas_descs", "blas_instances", "tlas_nodes", "vertex_positions", "vertices", "meshes", "materials", "mesh_transforms", "lights"):
        np.ascontiguousarray(getattr(sc, name)).tofile(str(tmp_path / (name + ".bin")))
    np.ascontiguousarray(sc.sky_faces, np.float32).tofile(str(tmp_path / "sky_faces.bin"))
    np.concatenate([cam.inv_projection, cam.inv_view, cam.position.astype(np.float32)]).astype(np.float32).tofile(str(tmp_path / "camera.bin"))
    for use_tlas in (0, 1):
        out = subprocess.run([exe, str(tmp_path), str(w), str(h), "4", "2", str(use_tlas)], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        assert out.stdout.startswith("ok ")
        got = np.fromfile(str(tmp_path / "result.bin"), np.float32).reshape(h, w, 4)
        rays, pairs, tris, acc = (int(x) for x in open(str(tmp_path / "stats.txt")).read().split())
        pt = gpu_render(sc, cam, w, h, RayDepth=4, SamplesPerPixel=2, UseTlas=use_tlas)
        st = pt.stats()
        assert (bits(got) == bits(pt.Result)).all()
        assert (rays, pairs, tris, acc) == (st["rays_traced"], st["node_pair_visits"], st["triangle_tests"], pt.AccumulatedSamples)
        pt.Dispose()
    # the same host, ONE context on two devices (ids wrap around the visible GPUs): the whole frame comes back through the same calls
    for use_tlas in (0, 1):
        out = subprocess.run([exe, str(tmp_path), str(w), str(h), "4", "2", str(use_tlas), "2"], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        got = np.fromfile(str(tmp_path / "result.bin"), np.float32).reshape(h, w, 4)
        rays, pairs, tris, acc = (int(x) for x in open(str(tmp_path / "stats.txt")).read().split())
        pt = gpu_render(sc, cam, w, h, RayDepth=4, SamplesPerPixel=2, UseTlas=use_tlas)
        st = pt.stats()
        assert (bits(got) == bits(pt.Result)).all()
        assert (rays, pairs, tris, acc) == (st["rays_traced"], st["node_pair_visits"], st["triangle_tests"], pt.AccumulatedSamples)
        pt.Dispose()


def test_error_paths_fail_loudly(native_builder):
    from idkengine_amd.pathtracer import PathTracer, IdkPtError
    pt = PathTracer(64, 64)
    with pytest.raises(IdkPtError):
        pt.Compute()                                  # no scene uploaded
    sc = S.cornell_scene(native_builder)
    bad = S.cornell_scene(native_builder); bad.blas_triangles = bad.blas_triangles.copy(); bad.blas_triangles["X"][0] = 10 ** 6
    with pytest.raises(IdkPtError):
        pt.UploadScene(bad)                           # out-of-range vertex index is rejected on the host, never reaches the GPU
    pt.UploadScene(sc)
    with pytest.raises(IdkPtError):
        pt.UseTlas = 1; pt.BuildTlas(np.zeros(0, sc.tlas_nodes.dtype))
    with pytest.raises(IdkPtError):
        pt.RefitBlas(0)                               # BLAS is not refittable
    with pytest.raises(IdkPtError):
        pt.SetSize(8192, 64)                          # FirstHit seeds pack x into 12 bits
    # the adjacent entry points report misuse the same way
    from idkengine_amd import gputypes as T
    pt.UseTlas = 0
    cam = S.cornell_camera(16, 16)
    with pytest.raises(IdkPtError):
        pt.TraceShadows(T.ShadowParams.make(cam.inv_proj_view, 16, 16, light_index=0), np.zeros((16, 16), np.float32), np.zeros((16, 16, 2), np.float32))   # the scene has no lights
    with pytest.raises(IdkPtError):
        pt.SetFrameRing(0)
    with pytest.raises(IdkPtError):
        pt.SetFrameRing(2); pt.FrameResult(5)         # slot outside the ring
    with pytest.raises(IdkPtError):
        pt.SetRowRange(60, 10)                        # strip exceeds the image
    assert len(pt.TraceRays(np.zeros(0, T.RayQuery))) == 0   # empty query is fine
    pt.Dispose()
    # more samples in flight than the device can hold: a clean error, the previous configuration stays usable
    big = PathTracer(4096, 16384); big.UploadScene(sc); big.SetCamera(S.cornell_camera(4096, 16384))
    with pytest.raises(IdkPtError, match="samples in flight"):
        big.set_max_batch(256)                        # 3 ray planes alone would need 3 x 275 GB
    big.SetSize(64, 64); big.SetCamera(S.cornell_camera(64, 64)); big.RayDepth = 2; big.Compute()
    ref = PathTracer(64, 64); ref.UploadScene(sc); ref.SetCamera(S.cornell_camera(64, 64)); ref.RayDepth = 2; ref.Compute()
    assert (bits(big.Result) == bits(ref.Result)).all()
    big.Dispose(); ref.Dispose()


def test_undersized_traversal_stacks_are_rejected(native_builder):
    """The reference compiles its shaders with BLAS_STACK_SIZE = max RequiredStackSize (Bvh/BVH.cs:559-567); a smaller stack would drop
    pushes.  The library computes what every BLAS / the TLAS needs at upload and refuses anything smaller instead of returning a
    wrong image."""
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd._lib import IdkPtError
    import copy
    sc = S.soup_scene(20000, native_builder, seed=5, extent=3.0); w, h = 96, 64; cam = S.Camera(w, h, position=(0.0, 0.0, 9.0))
    need = int(sc.blas_descs["RequiredStackSize"].max())
    assert need >= 4
    pt = PathTracer(w, h); pt.UploadScene(sc); pt.SetCamera(cam); pt.RayDepth = 3
    pt.Compute(); ref = pt.Result
    with pytest.raises(IdkPtError, match="BlasStackSize"):
        pt.BlasStackSize = need - 1                   # idkptSetSettings: smaller than the resident scene needs
    pt._settings.BlasStackSize = 0                    # (the failed push left the host mirror modified)
    pt.BlasStackSize = need + 3                       # larger is fine and changes nothing
    pt.ResetAccumulation(); pt.Compute()
    assert (bits(pt.Result) == bits(ref)).all()
    # a host that under-reports RequiredStackSize
    bad = copy.copy(sc); bad.blas_descs = sc.blas_descs.copy(); bad.blas_descs["RequiredStackSize"] = need - 2
    with pytest.raises(IdkPtError, match="RequiredStackSize"):
        pt.UploadScene(bad)
    # a scene that needs more than the BlasStackSize that is already set
    cornell = S.cornell_scene(native_builder)
    small = PathTracer(w, h); small.UploadScene(cornell); small.BlasStackSize = int(cornell.blas_descs["RequiredStackSize"].max())
    assert small.BlasStackSize < need
    with pytest.raises(IdkPtError, match="BlasStackSize"):
        small.UploadScene(sc)
    # child indices that point backwards (a cycle would hang the traversal) never reach the GPU
    cyc = copy.copy(sc); cyc.blas_nodes = sc.blas_nodes.copy()
    inner = [i for i in range(2, len(cyc.blas_nodes)) if cyc.blas_nodes["TriCount"][i] == 0 and cyc.blas_nodes["TriStartOrChild"][i] > 2][5]
    cyc.blas_nodes["TriStartOrChild"][inner] = 2
    with pytest.raises(IdkPtError, match="child index"):
        pt.UploadScene(cyc)
    pt.Dispose(); small.Dispose()


def test_invalid_tlas_nodes_are_rejected(native_builder):
    """TLAS nodes from the host are index- and depth-validated like the BLAS nodes (idkptUploadScene, idkptBuildTlas)."""
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd._lib import IdkPtError
    import copy
    sc = S.cornell_scene(native_builder, "mixed", True)
    pt = PathTracer(64, 64); pt.UploadScene(sc); pt.SetCamera(S.cornell_camera(64, 64)); pt.UseTlas = 1; pt.RayDepth = 3
    pt.Compute(); ref = pt.Result
    word = "IsLeafAndChildOrInstanceId"
    t = sc.tlas_nodes.copy(); leaf = [i for i in range(len(t)) if t[word][i] >> 31][0]
    t[word][leaf] = (1 << 31) | 77                    # instance id out of range
    with pytest.raises(IdkPtError, match="instance"):
        pt.BuildTlas(t)
    t = sc.tlas_nodes.copy(); t[word][0] = 0           # the root's children = the root itself: a cycle
    with pytest.raises(IdkPtError, match="child"):
        pt.BuildTlas(t)
    bad = copy.copy(sc); bad.tlas_nodes = t
    with pytest.raises(IdkPtError, match="child"):
        pt.UploadScene(bad)
    pt.BuildTlas(sc.tlas_nodes); pt.ResetAccumulation(); pt.Compute()     # the context is still usable
    assert (bits(pt.Result) == bits(ref)).all()
    pt.Dispose()


def test_strip_is_revalidated_on_resize_and_settings_repush_is_idempotent(native_builder):
    """idkptSetSize must re-validate a strip set earlier with idkptSetRowRange; pushing the same settings struct again
    (DoDebugBVHTraversal stores RayDepth 1 internally) must neither flush nor reset the accumulation."""
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd._lib import IdkPtError
    sc = S.cornell_scene(native_builder)
    pt = PathTracer(64, 64); pt.UploadScene(sc); pt.SetCamera(S.cornell_camera(64, 64))
    pt.SetRowRange(40, 24)
    with pytest.raises(IdkPtError, match="strip"):
        pt.SetSize(64, 48)                            # rows 40..63 do not exist in a 48-row image
    pt.SetRowRange(0, 48); pt.SetSize(64, 48); pt.SetCamera(S.cornell_camera(64, 48)); pt.Compute()
    assert pt.Result.shape == (48, 64, 4)
    pt.Dispose()
    pt = PathTracer(64, 64); pt.UploadScene(sc); pt.SetCamera(S.cornell_camera(64, 64))
    pt._settings.RayDepth = 5; pt._settings.Gpu.DoDebugBVHTraversal = 1; pt._push_settings()      # a host that pushes its whole struct, RayDepth != 1
    for _ in range(3):
        pt.Compute(); pt._push_settings()
    assert pt.AccumulatedSamples == 3
    pt.Dispose()


def test_device_pointer_getters_launch_deferred_samples(native_builder):
    """idkpt.h: idkptGetImageDevicePtr / idkptGetFrameDevicePtr launch what is still deferred, so a consumer ordered behind the
    context's stream reads the finished image (they used to hand out a stale image)."""
    from idkengine_amd.pathtracer import PathTracer
    sc = S.cornell_scene(native_builder); w, h = 64, 48; cam = S.cornell_camera(w, h)
    a = PathTracer(w, h); a.UploadScene(sc); a.SetCamera(cam); a.RayDepth = 3; a.set_max_batch(8)
    b = PathTracer(w, h); b.UploadScene(sc); b.SetCamera(cam); b.RayDepth = 3
    for p in (a, b):
        for _ in range(3):
            p.Compute()
    ptr, nbytes = a.image_device_ptr(0)               # 3 samples were pending: this call launches them
    out = read_device_image(a, ptr, nbytes, (h, w, 4))
    assert (bits(out) == bits(b.Result)).all()
    a.Dispose(); b.Dispose()
