"""Extracts the two glTF meshes the reference ships into small test fixtures:
    /root/reference/IDKEngine/Resource/Models/LucyCompressed/Lucy.{gltf,bin}            ->  tests/golden/models/lucy.npz   (8 954 triangles, 26 862 vertices)
    /root/reference/IDKEngine/Resource/Models/HelmetCompressed/{Helmet.gltf,DamagedHelmet.bin} -> helmet.npz (15 452 triangles, 14 356 vertices)
Only the plain accessors are read (POSITION / NORMAL / TEXCOORD_0 float32, indices uint16 — what Utils/ModelLoader.cs:829-1111 feeds the
BVH); textures (KTX2/Basis) are not decoded, the test materials use factors only.  Geometry data, not source code; Lucy is the Stanford
scan, DamagedHelmet the Khronos sample model (CC BY).  Run in the build container (the GPU box has no /root/reference):
    python tests/golden/make_models.py"""
import json
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/IDKEngine/Resource/Models"
DT = {5126: np.float32, 5123: np.uint16, 5125: np.uint32}
NC = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4}


def accessor(g, blob, idx):
    a = g["accessors"][idx]; v = g["bufferViews"][a["bufferView"]]
    dt = np.dtype(DT[a["componentType"]]); nc = NC[a["type"]]
    off = v.get("byteOffset", 0) + a.get("byteOffset", 0)
    stride = v.get("byteStride", dt.itemsize * nc)
    assert stride == dt.itemsize * nc, "interleaved views are not used by these files"
    return np.frombuffer(blob, dt, a["count"] * nc, off).reshape(a["count"], nc).copy()


def extract(folder, gltf, out):
    g = json.load(open(os.path.join(REF, folder, gltf)))
    blob = open(os.path.join(REF, folder, g["buffers"][0]["uri"]), "rb").read()
    assert len(g["meshes"]) == 1 and len(g["meshes"][0]["primitives"]) == 1
    p = g["meshes"][0]["primitives"][0]
    pos = accessor(g, blob, p["attributes"]["POSITION"]); nrm = accessor(g, blob, p["attributes"]["NORMAL"]); uv = accessor(g, blob, p["attributes"]["TEXCOORD_0"])
    idx = accessor(g, blob, p["indices"]).reshape(-1, 3)
    np.savez_compressed(os.path.join(HERE, "models", out), positions=pos, normals=nrm, uvs=uv, indices=idx.astype(np.uint16))
    print(out, "vertices", len(pos), "triangles", len(idx), "bounds", pos.min(0), pos.max(0))


if __name__ == "__main__":
    extract("LucyCompressed", "Lucy.gltf", "lucy.npz")
    extract("HelmetCompressed", "Helmet.gltf", "helmet.npz")
