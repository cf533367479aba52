"""Writes a scene's BLAS nodes + triangle positions in the format tools/sim_layout.cpp reads (developer tool).
usage: python tools/dump_scene_for_sim.py soup|atrium <tris> <out.bin>"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from idkengine_amd import scenes as S
from idkengine_amd.bvh import NativeBuilder

kind, n, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
sc = S.soup_scene(n, NativeBuilder()) if kind == "soup" else S.atrium_scene(n, NativeBuilder())
nodes = np.ascontiguousarray(sc.blas_nodes)
t = sc.blas_triangles
p = sc.vertex_positions.reshape(-1, 3)
tv = np.zeros((len(t), 3, 4), np.float32)
tv[:, 0, :3] = p[t["X"]]; tv[:, 1, :3] = p[t["Y"]]; tv[:, 2, :3] = p[t["Z"]]
with open(out, "wb") as f:
    f.write(np.int32([len(nodes), len(t)]).tobytes()); f.write(nodes.tobytes()); f.write(tv.tobytes())
print(kind, n, "->", out, len(nodes), "nodes", len(t), "triangles")
