"""Round 5 / 6 soak: the instance loop through the library's own TLAS / with the instance sieve / through the unified trees (round 6) against the loop inside k_trace2, GPU against GPU at full size — the frame, the whole ray
state, the alive queue and the primary hits of every sample must be equal bit for bit.  Scenes: the atrium as 87 BLASes (connected surfaces, PreSplit fragments), 64 clusters, an
instanced scene with tied copies.  Prints rays compared / rays traced again / mismatching samples."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd import gputypes as T  # noqa: E402
from idkengine_amd.bvh import NativeBuilder  # noqa: E402
from idkengine_amd.pathtracer import PathTracer  # noqa: E402

W, H = 1920, 1080
SAMPLES = int(sys.argv[1]) if len(sys.argv) > 1 else 24


def clusters(e, n=64, tris=300000):
    rng = np.random.default_rng(7); blases = []
    for k in range(n):
        p, i, nrm, tan = S.flat_shaded(S.soup_triangles(tris // n, 100 + k, e, 0.15))
        blases.append({"meshes": [S.MeshInput(p, i, S.make_material((0.8, 0.8, 0.8, 1.0)), nrm, tan)], "transform": S.rotation_y(float(rng.uniform(0, 360))) @ S.translation(tuple(rng.uniform(-(10 - e), 10 - e, 3)))})
    return S.assemble(blases, NativeBuilder(), build_tlas=False)


def instanced():
    sc = clusters(3.0, n=6, tris=60000)
    ids = [0, 1, 2, 3, 4, 5, 1, 3, 0]
    m = [np.eye(4)] * 6 + [np.eye(4), S.translation((1e-6, 0.0, 0.0)), S.rotation_y(33.0) @ S.translation((4.0, 1.0, -2.0))]
    xf = [sc.mesh_transforms[i:i + 1] for i in range(6)] + [sc.mesh_transforms[1:2], None, None]
    xf[7] = S.transform_from_matrix(np.linalg.inv(np.eye(4)) @ S.translation((1e-6, 0.0, 0.0)))       # instance 7: BLAS 3 again, a hair aside of nothing (its own place is instance 3's)
    xf[8] = S.transform_from_matrix(m[8])
    xf[7] = sc.mesh_transforms[3:4].copy(); xf[7]["Model"][0][0, 3] += 1e-6; xf[7]["InvModel"][0][0, 3] -= 1e-6
    inst = np.zeros(len(ids), T.GpuBlasInstance); inst["BlasId"] = ids; inst["MeshTransformId"] = np.arange(len(ids))
    sc.blas_instances = inst; sc.mesh_transforms = np.concatenate(xf)
    return sc


def run(pt, opts, depth):
    for k, v in opts.items():
        pt.set_option(k, v)
    pt.RayDepth = depth; pt.ResetAccumulation(); pt.reset_stats()
    out = []
    for _ in range(SAMPLES // 8):
        for _ in range(8):
            pt.Compute()
        pt.flush()
        out.append((np.ascontiguousarray(pt.Result).view(np.uint32).copy(), pt.rays().tobytes(), pt.alive_queue().copy()))
    return out, pt.stats()


def main():
    cases = [("atrium_87_blases", S.atrium_scene(1_000_000, NativeBuilder(), per_mesh_blas=True), S.atrium_camera(W, H)),
             ("clusters_64", clusters(2.24), bench.view_camera(S, "interior", W, H)),
             ("instanced_with_tied_copies", instanced(), S.Camera(W, H, position=(0.5, 0.5, 16.0), fovy_deg=70.0))]
    res = {}
    for name, sc, cam in cases:
        pt = PathTracer(W, H); pt.UploadScene(sc); pt.SetCamera(cam); pt.set_max_batch(8)
        row = {}
        for depth in (2, 5):
            same_space = False
            off = {"inst_unify": 0, "inst_general": 0, "inst_braid": 0, "packet": 1}
            ref, st0 = run(pt, dict(off, inst_tlas=0, inst_sieve=0), depth)
            # round 5: the own TLAS, the sieved loop; round 6: entries under the own TLAS, the unified tree of same-space scenes (lane walk; packets for the primary launch), the general array
            for label, opts in (("own_tlas", dict(off, inst_tlas=2, inst_tlas_overlap=100, inst_sieve=0)), ("sieved_loop", dict(off, inst_tlas=0, inst_sieve=2, inst_sieve_overlap=100)),
                                ("own_tlas_braid_512", dict(off, inst_tlas=2, inst_tlas_overlap=100, inst_sieve=0, inst_braid=512)),
                                ("unified_tree", dict(off, inst_tlas=8, inst_sieve=8, inst_unify=4096, packet=0)), ("unified_tree_packets", dict(off, inst_tlas=8, inst_sieve=8, inst_unify=4096, packet=2)),
                                ("general_array", dict(off, inst_tlas=8, inst_sieve=8, inst_unify=4096, inst_general=2))):
                got, st = run(pt, opts, depth)
                if label == "unified_tree": same_space = st["inst_unified_entries"] > 0
                if (label.startswith("unified") and not same_space) or (label == "general_array" and (same_space or st["inst_unified_launches"] == 0)):
                    continue                                     # (does not apply to this scene: not one space / one space)
                bad = sum(1 for a, b in zip(ref, got) if not ((a[0] == b[0]).all() and a[1] == b[1] and a[2].shape == b[2].shape and (a[2] == b[2]).all()))
                row[f"depth{depth}_{label}"] = {"rays": st["rays_traced"], "rays_traced_again": st["inst_tlas_flagged_rays"] + st["packet_flagged_rays"], "unified_launches": st["inst_unified_launches"], "packets": st["packet_packets"],
                                                "mismatching_checkpoints": bad, "checkpoints": len(ref)}
        pt.Dispose()
        res[name] = row
        print(json.dumps({name: row}), flush=True)


if __name__ == "__main__":
    main()
