"""Mints tests/golden/*.npz from the CPU oracle.  Run from the repo root:  python tests/golden/make_golden.py
(The reference has no golden vectors and cannot run here; see oracle/ref_math.h.)"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
import numpy as np  # noqa: E402
from oracle import oracle as O  # noqa: E402
import configs  # noqa: E402


def main():
    b = O.OracleBuilder()
    for name, (mk_scene, mk_cam, w, h, ov) in configs.CASES.items():
        sc = mk_scene(b); cam = mk_cam(w, h)
        pt = O.OraclePathTracer(sc, w, h); pt.set_camera(cam)
        configs.apply_settings(pt.settings, ov)
        pt.enable_counters(True)
        pt.render()
        t, tri, bary = pt.primary_hits(); st = pt.stats()
        out = dict(result=pt.image(0), rays=pt.rays(), prim_t=t, prim_tri=tri, prim_bary=bary, alive=pt.alive_queue(),
                   alive_counts=np.asarray(st["alive_counts"], np.uint32), rays_traced=np.uint64(st["rays_traced"]),
                   pairs=np.uint64(st["node_pair_visits"]), tris=np.uint64(st["triangle_tests"]))
        if ov.get("OutputAOVs"):
            out["albedo"] = pt.image(1); out["normal"] = pt.image(2)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "rays", st["rays_traced"], "mean", float(out["result"][..., :3].mean()))
        pt.close()
    bvh = {}
    for name, mk in configs.BVH_CASES.items():
        sc = mk(b)
        bvh[name] = {
            "nodes_sha256": hashlib.sha256(sc.blas_nodes.tobytes()).hexdigest(), "tris_sha256": hashlib.sha256(sc.blas_triangles.tobytes()).hexdigest(),
            "tlas_sha256": hashlib.sha256(sc.tlas_nodes.tobytes()).hexdigest(), "node_count": int(len(sc.blas_nodes)), "tri_count": int(len(sc.blas_triangles)),
            "stack": [int(x) for x in sc.blas_descs["RequiredStackSize"]], "sah": configs.sah_cost(sc),
        }
    json.dump(bvh, open(os.path.join(HERE, "bvh.json"), "w"), indent=1, sort_keys=True)
    print("bvh.json written")


if __name__ == "__main__":
    main()
