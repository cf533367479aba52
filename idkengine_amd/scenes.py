"""Scene assembly and synthetic scene generators (host side).

Plays the role of the reference's ModelManager.Add + BVH.Add/BlasesBuild/TlasBuild for generated geometry:
  IDKEngine/Source/ModelManager.cs:128-216, IDKEngine/Source/Bvh/BVH.cs:236-298,300-470.
The BLAS/TLAS builders themselves are pluggable (`builder` argument): the product's native SweepSAH builder
(idkengine_amd.bvh.NativeBuilder) or, in tests only, the oracle's restatement.
"""
import math
import numpy as np
from . import gputypes as T

# ----------------------------------------------------------------------------------------------- camera (OpenTK math)


def look_at(eye, target, up):
    """OpenTK Matrix4.LookAt (row-vector convention, rows = basis vectors, Row3 = translation)."""
    eye, target, up = (np.asarray(v, np.float64) for v in (eye, target, up))
    z = eye - target; z /= np.linalg.norm(z)
    x = np.cross(up, z); x /= np.linalg.norm(x)
    y = np.cross(z, x); y /= np.linalg.norm(y)
    m = np.zeros((4, 4), np.float64)
    m[0, :3] = (x[0], y[0], z[0]); m[1, :3] = (x[1], y[1], z[1]); m[2, :3] = (x[2], y[2], z[2])
    m[3] = (-np.dot(x, eye), -np.dot(y, eye), -np.dot(z, eye), 1.0)
    return m


def perspective_zero_to_one(fovy, aspect, near, far):
    """MyMath.CreatePerspectiveFieldOfViewDepthZeroToOne (Source/Utils/MyMath.cs:180-188)."""
    max_y = near * math.tan(0.5 * fovy); min_y = -max_y
    min_x = min_y * aspect; max_x = max_y * aspect
    m = np.zeros((4, 4), np.float64)
    m[0, 0] = 2.0 * near / (max_x - min_x)
    m[1, 1] = 2.0 * near / (max_y - min_y)
    m[2, 0] = (max_x + min_x) / (max_x - min_x)
    m[2, 1] = (max_y + min_y) / (max_y - min_y)
    m[2, 2] = far / (near - far)
    m[2, 3] = -1.0
    m[3, 2] = -(far * near) / (far - near)
    return m


class Camera:
    """Camera defaults of Source/Camera.cs:41-45: near 0.1, far 250, FovY 102 degrees."""

    def __init__(self, width, height, position=(0.0, 0.0, 25.0), view_dir=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0),
                 fovy_deg=102.0, near=0.1, far=250.0):
        self.position = np.asarray(position, np.float32)
        view = look_at(position, np.asarray(position, np.float64) + np.asarray(view_dir, np.float64), up)
        proj = perspective_zero_to_one(math.radians(fovy_deg), width / float(height), near, far)
        # GpuPerFrameData.InvView / InvProjection (Application.cs:148-150), OpenTK memory order (row-major rows)
        self.view, self.proj = view, proj
        # GpuPerFrameData.InvProjView = (View * Projection).Inverted(), same memory order
        self.inv_proj_view = np.ascontiguousarray(np.linalg.inv(view @ proj).astype(np.float32).reshape(16))
        self.inv_view = np.ascontiguousarray(np.linalg.inv(view).astype(np.float32).reshape(16))
        self.inv_projection = np.ascontiguousarray(np.linalg.inv(proj).astype(np.float32).reshape(16))

# ----------------------------------------------------------------------------------------------- vertex packing


def compress_sr11g11b10(v):
    """Utils/Compression.cs:26-40 (MathF.Round = round-half-even)."""
    v = np.asarray(v, np.float32) * np.float32(0.5) + np.float32(0.5)
    r = np.rint(v[..., 0] * np.float32(2047)).astype(np.uint32)
    g = np.rint(v[..., 1] * np.float32(2047)).astype(np.uint32)
    b = np.rint(v[..., 2] * np.float32(1023)).astype(np.uint32)
    return (b << 22) | (g << 11) | r


def pack_rgba8(rgba):
    c = np.rint(np.clip(np.asarray(rgba, np.float64), 0, 1) * 255.0).astype(np.uint32)
    return int(c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24))


def transform_from_matrix(model4x4):
    """GpuMeshTransform from an OpenTK-convention 4x4 (rows = basis, row 3 = translation): stores the transposed
    3x4 of Model, its inverse and Prev = Model (GpuMeshTransform.cs:17-41)."""
    m = np.asarray(model4x4, np.float64)
    out = np.zeros(1, T.GpuMeshTransform)
    out["Model"][0] = m[:, :3].T.astype(np.float32)
    out["InvModel"][0] = np.linalg.inv(m)[:, :3].T.astype(np.float32)
    out["PrevModel"][0] = out["Model"][0]
    return out


def rotation_y(deg):
    a = math.radians(deg); c, s = math.cos(a), math.sin(a)
    m = np.eye(4); m[0, 0] = c; m[0, 2] = -s; m[2, 0] = s; m[2, 2] = c
    return m


def translation(t):
    m = np.eye(4); m[3, :3] = t
    return m

# ----------------------------------------------------------------------------------------------- materials


def make_material(base_color=(0.8, 0.8, 0.8, 1.0), emissive=(0, 0, 0), metallic=0.0, roughness=1.0, transmission=0.0,
                  ior=1.5, absorbance=(0, 0, 0), alpha_cutoff=0.0, volumetric=False):
    m = np.zeros(1, T.GpuMaterial)
    m["EmissiveFactor"] = emissive; m["BaseColorFactor"] = pack_rgba8(base_color)
    m["Absorbance"] = absorbance; m["IOR"] = ior
    m["TransmissionFactor"] = transmission; m["RoughnessFactor"] = roughness; m["MetallicFactor"] = metallic
    m["AlphaCutoff"] = alpha_cutoff; m["IsVolumetric"] = 1 if volumetric else 0
    return m


def make_mesh(material_id, normal_map_strength=0.0, **bias):
    m = np.zeros(1, T.GpuMesh)
    m["MaterialId"] = material_id; m["NormalMapStrength"] = normal_map_strength
    m["InstanceCount"] = 1; m["TintOnTransmissive"] = 1
    for k, v in bias.items():
        m[k] = v
    return m

# ----------------------------------------------------------------------------------------------- assembly


class MeshInput:
    """One glTF-primitive-like mesh: per-vertex positions/normals/tangents/uvs + triangle indices (local)."""

    def __init__(self, positions, indices, material, normals=None, tangents=None, uvs=None, mesh_kwargs=None):
        self.positions = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        self.indices = np.ascontiguousarray(indices, np.uint32).reshape(-1, 3)
        self.material = material
        self.normals = normals; self.tangents = tangents; self.uvs = uvs
        self.mesh_kwargs = mesh_kwargs or {}


def flat_shaded(tri_positions):
    """(n,3,3) triangle soup -> unshared vertices with face normal and an in-plane tangent (always valid: the
    reference's TBN goes NaN on a zero tangent, FirstHit/compute.glsl:148-153)."""
    tp = np.ascontiguousarray(tri_positions, np.float32).reshape(-1, 3, 3)
    e1 = (tp[:, 1] - tp[:, 0]).astype(np.float64); e2 = (tp[:, 2] - tp[:, 0]).astype(np.float64)
    n = np.cross(e1, e2); nl = np.linalg.norm(n, axis=1, keepdims=True); nl[nl == 0] = 1.0; n /= nl
    t = e1.copy(); tl = np.linalg.norm(t, axis=1, keepdims=True); tl[tl == 0] = 1.0; t /= tl
    positions = tp.reshape(-1, 3)
    normals = np.repeat(n, 3, axis=0).astype(np.float32); tangents = np.repeat(t, 3, axis=0).astype(np.float32)
    indices = np.arange(len(positions), dtype=np.uint32).reshape(-1, 3)
    return positions, indices, normals, tangents


def assemble(blases, builder, lights=None, sky_color=(1.0, 1.0, 1.0), build_tlas=True):
    """blases: list of dicts {meshes: [MeshInput], transform: 4x4 (OpenTK convention) or None, refittable: bool}.
    Mirrors ModelManager.Add (global vertex/mesh arrays), BVH.Add (BlasTriangles with global vertex ids + MeshId,
    Bvh/BVH.cs:236-276), BVH.BlasesBuild (per-BLAS build + offset fix-up, :300-451) and BVH.TlasBuild (:278-298)."""
    sc = T.Scene()
    pos, verts, meshes, materials, xforms = [], [], [], [], []
    per_blas_tris = []
    vertex_offset = 0
    for b_id, b in enumerate(blases):
        tris = []
        for mi in b["meshes"]:
            mesh_id = len(meshes)
            materials.append(mi.material); meshes.append(make_mesh(len(materials) - 1, **mi.mesh_kwargs))
            nv = len(mi.positions)
            v = np.zeros(nv, T.GpuVertex)
            normals = mi.normals if mi.normals is not None else np.tile(np.float32([0, 1, 0]), (nv, 1))
            tangents = mi.tangents if mi.tangents is not None else np.tile(np.float32([1, 0, 0]), (nv, 1))
            v["Normal"] = compress_sr11g11b10(normals); v["Tangent"] = compress_sr11g11b10(tangents)
            if mi.uvs is not None:
                v["TexCoord"] = mi.uvs
            t = np.zeros(len(mi.indices), T.GpuBlasTriangle)
            t["X"] = mi.indices[:, 0] + vertex_offset; t["Y"] = mi.indices[:, 1] + vertex_offset; t["Z"] = mi.indices[:, 2] + vertex_offset
            t["MeshId"] = mesh_id
            tris.append(t); pos.append(mi.positions); verts.append(v)
            vertex_offset += nv
        per_blas_tris.append(np.concatenate(tris))
        xf = b.get("transform")
        xforms.append(transform_from_matrix(np.eye(4) if xf is None else xf))
    sc.vertex_positions = np.ascontiguousarray(np.concatenate(pos), np.float32)
    sc.vertices = np.concatenate(verts)
    sc.meshes = np.concatenate(meshes); sc.materials = np.concatenate(materials); sc.mesh_transforms = np.concatenate(xforms)
    # per-BLAS build
    nodes, tris_out, parents, leaves = [], [], [], []
    descs = np.zeros(len(blases), T.GpuBlasDesc)
    for i, b in enumerate(blases):
        r = builder.build_blas(sc.vertex_positions, per_blas_tris[i], bool(b.get("refittable", False)))
        d = descs[i]
        d["NodeOffset"] = sum(len(x) for x in nodes); d["NodeCount"] = len(r["nodes"])
        d["TriangleOffset"] = sum(len(x) for x in tris_out); d["TriangleCount"] = len(r["triangles"])
        d["LeafIndicesOffset"] = sum(len(x) for x in leaves); d["LeafIndicesCount"] = len(r["leaves"])
        d["ParentIndicesOffset"] = sum(len(x) for x in parents); d["ParentIndicesCount"] = len(r["parents"])
        d["RequiredStackSize"] = r["required_stack_size"]; d["IsRefittable"] = 1 if b.get("refittable", False) else 0
        nodes.append(r["nodes"]); tris_out.append(r["triangles"]); parents.append(r["parents"]); leaves.append(r["leaves"])
    sc.blas_nodes = np.concatenate(nodes); sc.blas_triangles = np.concatenate(tris_out)
    sc.blas_parent_indices = np.concatenate(parents).astype(np.int32); sc.blas_leaf_indices = np.concatenate(leaves).astype(np.int32)
    sc.blas_descs = descs
    inst = np.zeros(len(blases), T.GpuBlasInstance)
    inst["BlasId"] = np.arange(len(blases)); inst["MeshTransformId"] = np.arange(len(blases))
    sc.blas_instances = inst
    sc.blas_stack_size = int(descs["RequiredStackSize"].max())  # BVH.UpdateBlasStackSize (Bvh/BVH.cs:559-567)
    if build_tlas:
        rebuild_tlas(sc, builder)
    if lights is not None:
        sc.lights = lights
    if sky_color is not None:
        sky = np.zeros((6, 1, 1, 4), np.float32); sky[..., :3] = sky_color; sky[..., 3] = 1.0
        sc.sky_faces = sky
    return sc


def rebuild_tlas(sc, builder):
    """BVH.TlasBuild (Bvh/BVH.cs:278-298): world-space bounds of each instance's BLAS root -> PLOC."""
    bounds = np.zeros((len(sc.blas_instances), 6), np.float32)
    for i, inst in enumerate(sc.blas_instances):
        d = sc.blas_descs[inst["BlasId"]]
        root = sc.blas_nodes[d["NodeOffset"] + 1: d["NodeOffset"] + 2]
        bounds[i] = builder.instance_world_bounds(root, sc.mesh_transforms[inst["MeshTransformId"]: inst["MeshTransformId"] + 1])
    sc.tlas_nodes = builder.build_tlas(bounds)

# ----------------------------------------------------------------------------------------------- generators


def pcg32_stream(seed, n):
    """n uint32 outputs of the reference's PCG hash (Shaders/include/Random.glsl:20-27) chained from `seed`."""
    out = np.empty(n, np.uint32)
    s = np.uint64(seed)
    M = np.uint64(0xFFFFFFFF)
    for i in range(n):
        s = (s * np.uint64(747796405) + np.uint64(2891336453)) & M
        w = (((s >> ((s >> np.uint64(28)) + np.uint64(4))) ^ s) * np.uint64(277803737)) & M
        out[i] = (w >> np.uint64(22)) ^ w
    return out


def soup_triangles(n, seed=1, extent=10.0, edge=0.15):
    """centres uniform in [-extent,extent]^3, edge vectors uniform in [-edge,edge]^3.
    Uses numpy's PCG64 Generator(seed) (documented stand-in for a 9n-long scalar PCG32 stream: same distribution,
    deterministic for a given numpy version; the generated arrays, not the generator, are what both oracle and
    GPU path consume)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    c = rng.uniform(-extent, extent, (n, 3)).astype(np.float32)
    e1 = rng.uniform(-edge, edge, (n, 3)).astype(np.float32)
    e2 = rng.uniform(-edge, edge, (n, 3)).astype(np.float32)
    return np.stack([c, c + e1, c + e2], axis=1)


def soup_scene(n_tris, builder, seed=1, refittable=False, albedo=0.8, extent=10.0, edge=0.15, sky_color=(1.0, 1.0, 1.0)):
    """Headline workload (BASELINE.json configs[2]): one mesh/material (diffuse albedo 0.8, metallic 0, roughness 1,
    opaque), one BLAS, identity transform, constant white sky."""
    tp = soup_triangles(n_tris, seed, extent, edge)
    p, i, nrm, tan = flat_shaded(tp)
    mat = make_material(base_color=(albedo, albedo, albedo, 1.0), metallic=0.0, roughness=1.0)
    return assemble([{"meshes": [MeshInput(p, i, mat, nrm, tan)], "refittable": refittable}], builder, sky_color=sky_color)


def soup_scene_multi(n_tris, builder, parts=3, seed=1, albedo=0.8, extent=10.0, edge=0.15, sky_color=(1.0, 1.0, 1.0)):
    """The soup split into `parts` BLASes with their own (rotated) GpuMeshTransform and no hoisting: the reference's default
    multi-model layout (one BLAS per model, instance loop in BVHIntersect.glsl:275-287 unless UseTlas)."""
    blases = []
    per = n_tris // parts
    for k in range(parts):
        n = per if k < parts - 1 else n_tris - per * (parts - 1)
        p, i, nrm, tan = flat_shaded(soup_triangles(n, seed + 17 * k, extent, edge))
        mat = make_material(base_color=(albedo, albedo, albedo, 1.0), metallic=0.0, roughness=1.0)
        blases.append({"meshes": [MeshInput(p, i, mat, nrm, tan)], "transform": None if k == 0 else rotation_y(23.0 * k) @ translation((0.5 * k, -0.25 * k, 0.0))})
    return assemble(blases, builder, sky_color=sky_color)


def make_lights(specs):
    """specs: [(position, radius, color)] -> GpuLight array (Source/GpuTypes/GpuLight.cs:5-45)."""
    l = np.zeros(len(specs), T.GpuLight)
    for i, (pos, radius, color) in enumerate(specs):
        l[i]["Position"] = pos; l[i]["Radius"] = radius; l[i]["Color"] = color; l[i]["PointShadowIndex"] = -1; l[i]["PrevPosition"] = pos
    return l


def primary_ray_queries(cam, width, height, max_dist=3.4028235e+38):
    """One ray per pixel centre (gputypes.RayQuery), row-major; used to make ray-query test sets and stand-in G-buffers."""
    ys, xs = np.mgrid[0:height, 0:width]
    ndc = np.stack([(xs + 0.5) / width * 2.0 - 1.0, (ys + 0.5) / height * 2.0 - 1.0, np.ones_like(xs, np.float64), np.ones_like(xs, np.float64)], -1).reshape(-1, 4)
    w = ndc @ cam.inv_proj_view.reshape(4, 4).astype(np.float64)
    p = w[:, :3] / w[:, 3:4]
    d = p - cam.position.astype(np.float64); d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = np.zeros(width * height, T.RayQuery)
    r["Origin"] = cam.position; r["Direction"] = d.astype(np.float32); r["MaxDist"] = max_dist
    return r


def oct_encode(n):
    """Compression.glsl:54-60 (EncodeUnitVec) in numpy; used to fabricate G-buffer normals (inputs only)."""
    n = np.asarray(n, np.float64); n = n / np.abs(n).sum(-1, keepdims=True)
    xy = n[..., :2].copy()
    wrap = (1.0 - np.abs(xy[..., ::-1])) * np.where(xy >= 0.0, 1.0, -1.0)
    xy = np.where(n[..., 2:3] >= 0.0, xy, wrap)
    return (xy * 0.5 + 0.5).astype(np.float32)


def gbuffer_from_hits(sc, cam, width, height, rays, hits):
    """Stand-in for the rasterizer's G-buffer (gBufferDataUBO.Depth / .Normal): zero-to-one depth of the primary hit point and
    the oct-encoded world-space face normal (turned towards the camera); sky pixels get depth 1.0."""
    hit = hits["Hit"] != 0
    P = rays["Origin"].astype(np.float64) + rays["Direction"].astype(np.float64) * hits["T"].astype(np.float64)[:, None]
    clip = np.c_[P, np.ones(len(P))] @ (cam.view @ cam.proj)
    depth = np.where(hit, clip[:, 2] / clip[:, 3], 1.0).astype(np.float32)
    depth[hit & (depth >= 1.0)] = np.float32(0.99999)
    tri = sc.blas_triangles[np.where(hit, hits["TriangleId"], 0)]
    nrm = np.zeros((len(P), 3))
    for k in np.unique(hits["MeshTransformId"][hit]):
        sel = hit & (hits["MeshTransformId"] == k)
        x = sc.mesh_transforms[k]["Model"].astype(np.float64)          # 3x4 (transposed 4x3): p_world = M[:, :3] @ p + M[:, 3]
        def world(ids):
            p = sc.vertex_positions[ids].astype(np.float64)
            return p @ x[:, :3].T + x[:, 3]
        p0, p1, p2 = world(tri["X"][sel]), world(tri["Y"][sel]), world(tri["Z"][sel])
        n = np.cross(p1 - p0, p2 - p0); n /= np.linalg.norm(n, axis=1, keepdims=True)
        n *= np.where((n * rays["Direction"][sel]).sum(1, keepdims=True) > 0.0, -1.0, 1.0)
        nrm[sel] = n
    nrm[~hit] = (0.0, 0.0, 1.0)
    return depth.reshape(height, width), oct_encode(nrm).reshape(height, width, 2)


def _quad(a, b, c, d):
    return np.float32([[a, b, c], [a, c, d]])


def _box_faces(lo, hi, with_bottom=False):
    x0, y0, z0 = lo; x1, y1, z1 = hi
    f = [
        _quad((x0, y1, z0), (x0, y1, z1), (x1, y1, z1), (x1, y1, z0)),  # top
        _quad((x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)),  # front (+z)
        _quad((x1, y0, z0), (x0, y0, z0), (x0, y1, z0), (x1, y1, z0)),  # back
        _quad((x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0)),  # left
        _quad((x1, y0, z1), (x1, y0, z0), (x1, y1, z0), (x1, y1, z1)),  # right
    ]
    if with_bottom:
        f.append(_quad((x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1)))
    return np.concatenate(f)


def cornell_meshes(variant="diffuse"):
    """32 triangles: 5 walls (10) + ceiling light (2) + short box (10) + tall box (10).  The reference ships no
    Cornell box; this is the generated stand-in of SURVEY.md §8(d) config 1.  Box interior spans [-1,1]^3, open at +z."""
    white = make_material((0.73, 0.73, 0.73, 1.0)); red = make_material((0.65, 0.05, 0.05, 1.0)); green = make_material((0.12, 0.45, 0.15, 1.0))
    light = make_material((0.78, 0.78, 0.78, 1.0), emissive=(15.0, 15.0, 15.0))
    if variant == "diffuse":
        short_m = make_material((0.73, 0.73, 0.73, 1.0)); tall_m = make_material((0.73, 0.73, 0.73, 1.0))
    else:  # "mixed": exercises specular + volumetric transmission + alpha paths
        short_m = make_material((0.9, 0.95, 1.0, 1.0), transmission=1.0, roughness=0.05, ior=1.45, absorbance=(0.3, 0.1, 0.05), volumetric=True)
        tall_m = make_material((0.95, 0.8, 0.4, 1.0), metallic=1.0, roughness=0.2)
    walls = np.concatenate([
        _quad((-1, -1, -1), (-1, -1, 1), (1, -1, 1), (1, -1, -1)),   # floor
        _quad((-1, 1, -1), (1, 1, -1), (1, 1, 1), (-1, 1, 1)),       # ceiling
        _quad((-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1)),   # back
    ])
    left = _quad((-1, -1, -1), (-1, 1, -1), (-1, 1, 1), (-1, -1, 1))
    right = _quad((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1))
    lq = _quad((-0.25, 0.998, -0.25), (0.25, 0.998, -0.25), (0.25, 0.998, 0.25), (-0.25, 0.998, 0.25))
    short = _box_faces((-0.3, 0.0, -0.3), (0.3, 0.6, 0.3))
    tall = _box_faces((-0.3, 0.0, -0.3), (0.3, 1.2, 0.3))

    def mk(tp, m):
        p, i, n, t = flat_shaded(tp)
        return MeshInput(p, i, m, n, t)
    return {"walls": [mk(walls, white), mk(left, red), mk(right, green), mk(lq, light)], "short": mk(short, short_m), "tall": mk(tall, tall_m)}


def cornell_scene(builder, variant="diffuse", instanced=False, sky_color=(0.0, 0.0, 0.0)):
    """instanced=False: all 32 triangles hoisted into one BLAS (boxes pre-transformed) — the layout the reference
    prefers (Application.cs:481).  instanced=True: walls / short box / tall box are three BLASes with their own
    GpuMeshTransform, exercising RayTransform, the normal matrix and the TLAS."""
    m = cornell_meshes(variant)
    xs = rotation_y(-18.0) @ translation((0.33, -1.0, 0.4)); xt = rotation_y(17.0) @ translation((-0.35, -1.0, -0.3))
    if instanced:
        blases = [{"meshes": m["walls"]}, {"meshes": [m["short"]], "transform": xs}, {"meshes": [m["tall"]], "transform": xt}]
    else:
        def bake(mi, x):
            p = (np.c_[mi.positions.astype(np.float64), np.ones(len(mi.positions))] @ x)[:, :3].astype(np.float32)
            pp, ii, nn, tt = flat_shaded(p[mi.indices.reshape(-1)].reshape(-1, 3, 3))
            return MeshInput(pp, ii, mi.material, nn, tt)
        blases = [{"meshes": m["walls"] + [bake(m["short"], xs), bake(m["tall"], xt)]}]
    return assemble(blases, builder, sky_color=sky_color)


def cornell_camera(width, height):
    return Camera(width, height, position=(0.0, 0.0, 3.4), fovy_deg=40.0)


def presplit_scene(builder, n_small=5000, seed=3, refittable=False):
    """Small soup + three scene-spanning triangles: the size contrast makes PreSplitting (Bvh/PreSplitting.cs) split the
    big ones, so fragment count > triangle count and leaves share straddling triangles."""
    tp = soup_triangles(n_small, seed=seed, extent=2.0, edge=0.1)
    big = np.float32([[[-3, -3, -2.5], [3, -3, -2.4], [0, 3, 2.6]], [[-3, 0.1, -3], [3, 0.2, 3], [-3, 0.3, 3]], [[-2.5, -2.5, 2.0], [2.5, -2.0, -2.0], [2.0, 2.5, 0.0]]])
    p, i, nrm, tan = flat_shaded(np.concatenate([tp, big]))
    return assemble([{"meshes": [MeshInput(p, i, make_material((0.7, 0.6, 0.5, 1.0)), nrm, tan)], "refittable": refittable}], builder, sky_color=(0.9, 0.95, 1.0))


def presplit_camera(width, height):
    return Camera(width, height, position=(0.5, 0.8, 6.0), fovy_deg=60.0)


# ----------------------------------------------------------------------------------------------- glTF-accessor meshes (fixtures)


def mesh_scene(npz_path, builder, material=None, refittable=False, sky_color=(0.6, 0.7, 0.9), transform=None):
    """One indexed mesh (positions / normals / uvs / uint16 indices as extracted by tests/golden/make_models.py from the glTF files the
    reference ships) as one BLAS: shared vertices, sliver triangles and real PreSplit priorities that the generated scenes do not have.
    Tangents are not stored in those files; a unit vector orthogonal to the normal stands in (NormalMapStrength is 0)."""
    d = np.load(npz_path)
    pos, nrm, uv, idx = d["positions"], d["normals"].astype(np.float64), d["uvs"], d["indices"].astype(np.uint32)
    nl = np.linalg.norm(nrm, axis=1, keepdims=True); nl[nl == 0] = 1.0; nrm = nrm / nl
    axis = np.where(np.abs(nrm[:, :1]) > 0.9, np.float64([[0.0, 1.0, 0.0]]), np.float64([[1.0, 0.0, 0.0]]))
    tan = np.cross(nrm, axis); tan /= np.linalg.norm(tan, axis=1, keepdims=True)
    mat = material if material is not None else make_material((0.8, 0.78, 0.7, 1.0), roughness=0.55)
    return assemble([{"meshes": [MeshInput(pos, idx, mat, nrm.astype(np.float32), tan.astype(np.float32), uv)], "refittable": refittable, "transform": transform}], builder, sky_color=sky_color)


# ----------------------------------------------------------------------------------------------- procedural atrium ("Sponza-class" stand-in)


def _grid_mesh(fn, nu, nv, material, flip=False):
    """Indexed (nu+1) x (nv+1) grid of the parametric surface fn(u, v) -> (x, y, z), u, v in [0, 1]; per-vertex normals / tangents from finite
    differences; shared vertices like a real asset (Utils/ModelLoader.cs meshes)."""
    u, v = np.meshgrid(np.linspace(0.0, 1.0, nu + 1), np.linspace(0.0, 1.0, nv + 1), indexing="ij")
    P = np.stack(fn(u, v), -1).astype(np.float64)
    du = np.gradient(P, axis=0); dv = np.gradient(P, axis=1)
    n = np.cross(du, dv)
    if flip:
        n = -n
    nl = np.linalg.norm(n, axis=-1, keepdims=True); nl[nl == 0] = 1.0; n = n / nl
    t = du / np.maximum(np.linalg.norm(du, axis=-1, keepdims=True), 1e-12)
    idx = np.arange((nu + 1) * (nv + 1)).reshape(nu + 1, nv + 1)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    tri = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)]) if not flip else np.concatenate([np.stack([a, c, b], 1), np.stack([a, d, c], 1)])
    uv = np.stack([u, v], -1).reshape(-1, 2).astype(np.float32)
    return MeshInput(P.reshape(-1, 3).astype(np.float32), tri.astype(np.uint32), material, n.reshape(-1, 3).astype(np.float32), t.reshape(-1, 3).astype(np.float32), uv)


def atrium_scene(target_tris, builder, seed=7, sky_color=(1.0, 1.0, 1.0), per_mesh_blas=False):
    """A procedural two-storey colonnaded atrium (floor, outer walls with relief, two rows of columns on two levels, arches, gallery slabs,
    hanging drapes, vases) tessellated to about `target_tris` triangles in ONE BLAS: the layout of the Sponza atrium BASELINE.json's configs[1]
    and north_star's "Sponza-class scene" refer to (the reference checkout ships Sponza.gltf without its geometry buffer, so the real mesh is not
    available).  Unlike the random soup, surfaces are connected 2-manifolds with shared vertices, large empty spaces and occlusion.
    Hall: x in [-18, 18] (long axis), z in [-7, 7], y in [0, 12]; open to the sky.
    per_mesh_blas: one BLAS per mesh (87 of them) instead — the reference's default layout of a multi-mesh model without hoisting (Bvh/BVH.cs:156)."""
    rng = np.random.default_rng(seed)
    k = max(1.0, (target_tris / 42000.0) ** 0.5)                      # linear tessellation factor (the base model below has ~42 k triangles at k = 1)

    def q(n):
        return max(2, int(round(n * k)))
    stone = make_material((0.72, 0.68, 0.6, 1.0)); floor_m = make_material((0.55, 0.5, 0.45, 1.0), roughness=0.6, metallic=0.05)
    red = make_material((0.6, 0.12, 0.1, 1.0)); green = make_material((0.15, 0.45, 0.2, 1.0)); blue = make_material((0.15, 0.25, 0.6, 1.0))
    bronze = make_material((0.8, 0.55, 0.3, 1.0), metallic=0.9, roughness=0.35)
    meshes = []
    bump = lambda u, v, a, f: a * np.sin(f * 6.28318 * u) * np.sin(f * 6.28318 * v)          # noqa: E731  (relief so that the walls are not two giant coplanar sheets)
    # floor and gallery slabs
    meshes.append(_grid_mesh(lambda u, v: (-18 + 36 * u, bump(u, v, 0.03, 9.0), -7 + 14 * v), q(64), q(28), floor_m, flip=True))
    for zs in (-1, 1):
        meshes.append(_grid_mesh(lambda u, v, zs=zs: (-18 + 36 * u, 6.0 + bump(u, v, 0.02, 7.0), zs * (4.2 + 2.8 * v)), q(64), q(8), stone, flip=zs > 0))   # gallery floor (seen from below)
    # outer walls (long sides, two ends) with shallow relief and window-like recesses
    for zs in (-1, 1):
        meshes.append(_grid_mesh(lambda u, v, zs=zs: (-18 + 36 * u, 12 * v, zs * (7.0 - 0.25 * (np.sin(25.1327 * u) > 0.6) * (np.abs(v - 0.3) < 0.12) - bump(u, v, 0.04, 5.0))), q(72), q(30), stone, flip=zs < 0))
    for xs in (-1, 1):
        meshes.append(_grid_mesh(lambda u, v, xs=xs: (xs * (18.0 - bump(u, v, 0.05, 3.0)), 12 * v, -7 + 14 * u), q(30), q(30), stone, flip=xs > 0))
    # columns (two rows, two levels) and arches between them
    cols_x = np.linspace(-15.0, 15.0, 9)
    for level, (y0, hgt, rad) in enumerate(((0.0, 5.2, 0.45), (6.0, 4.6, 0.35))):
        for zs in (-1, 1):
            for cx in cols_x:
                meshes.append(_grid_mesh(lambda u, v, cx=cx, zs=zs, y0=y0, hgt=hgt, rad=rad: (cx + rad * (1 + 0.06 * np.cos(50.2655 * u)) * (1.0 - 0.12 * v) * np.cos(6.28318 * u), y0 + hgt * v,
                                                                                            zs * 4.2 + rad * (1 + 0.06 * np.cos(50.2655 * u)) * (1.0 - 0.12 * v) * np.sin(6.28318 * u)), q(20), q(10), stone))
            for cx0, cx1 in zip(cols_x[:-1], cols_x[1:]):
                mid, half = 0.5 * (cx0 + cx1), 0.5 * (cx1 - cx0)
                meshes.append(_grid_mesh(lambda u, v, mid=mid, half=half, zs=zs, y0=y0, hgt=hgt: (mid - half * np.cos(3.14159 * u), y0 + hgt + 0.8 * np.sin(3.14159 * u) * (0.55 + 0.45 * v),
                                                                                              zs * (4.2 - 0.3 + 0.6 * v)), q(14), q(3), stone, flip=zs > 0))
    # drapes hanging from the gallery (wavy sheets) and a row of vases (displaced spheres)
    for i, cx in enumerate(np.linspace(-12.0, 12.0, 5)):
        m = (red, green, blue)[i % 3]
        ph = rng.uniform(0, 6.28)
        meshes.append(_grid_mesh(lambda u, v, cx=cx, ph=ph: (cx - 1.6 + 3.2 * u, 5.8 - 4.2 * v, (-1) ** i * 3.6 + 0.25 * np.sin(18.85 * u + ph) * (0.3 + v)), q(26), q(22), m))
    for cx in np.linspace(-13.5, 13.5, 7):
        meshes.append(_grid_mesh(lambda u, v, cx=cx: (cx + (0.35 + 0.2 * np.sin(3.14159 * v) ** 2) * np.sin(3.14159 * v) ** 0.5 * np.cos(6.28318 * u), 0.05 + 1.3 * v,
                                                     (0.35 + 0.2 * np.sin(3.14159 * v) ** 2) * np.sin(3.14159 * v) ** 0.5 * np.sin(6.28318 * u)), q(16), q(10), bronze))
    if per_mesh_blas:
        return assemble([{"meshes": [m]} for m in meshes], builder, sky_color=sky_color)
    return assemble([{"meshes": meshes}], builder, sky_color=sky_color)


def atrium_camera(width, height):
    """Standing in the courtyard near one end, looking down the hall and slightly up (the classic Sponza view)."""
    return Camera(width, height, position=(-15.5, 2.2, 0.6), view_dir=(1.0, 0.12, -0.05), fovy_deg=70.0)
