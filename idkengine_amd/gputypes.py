"""numpy mirrors of include/idkpt_types.h (byte-exact; sizes asserted at import).

Reference: IDKEngine/Source/GpuTypes/*.cs <-> IDKEngine/Resource/Shaders/include/GpuTypes.glsl.
"""
import ctypes as C
import numpy as np

GpuBlasNode = np.dtype([("Min", "<f4", 3), ("TriStartOrChild", "<u4"), ("Max", "<f4", 3), ("TriCount", "<u4")])
GpuBlasTriangle = np.dtype([("X", "<u4"), ("Y", "<u4"), ("Z", "<u4"), ("MeshId", "<u4")])
GpuBlasDesc = np.dtype([
    ("NodeOffset", "<i4"), ("NodeCount", "<i4"), ("TriangleOffset", "<i4"), ("TriangleCount", "<i4"),
    ("LeafIndicesOffset", "<i4"), ("LeafIndicesCount", "<i4"), ("ParentIndicesOffset", "<i4"), ("ParentIndicesCount", "<i4"),
    ("RequiredStackSize", "<i4"), ("IsRefittable", "u1"), ("_pad", "u1", 3)])
GpuBlasInstance = np.dtype([("BlasId", "<u4"), ("MeshTransformId", "<u4")])
GpuTlasNode = np.dtype([("Min", "<f4", 3), ("IsLeafAndChildOrInstanceId", "<u4"), ("Max", "<f4", 3), ("_pad0", "<f4")])
GpuMeshTransform = np.dtype([("Model", "<f4", (3, 4)), ("InvModel", "<f4", (3, 4)), ("PrevModel", "<f4", (3, 4))])
GpuMesh = np.dtype([
    ("LocalBoundsMin", "<f4", 3), ("MaterialId", "<i4"), ("LocalBoundsMax", "<f4", 3), ("NormalMapStrength", "<f4"),
    ("AbsorbanceBias", "<f4", 3), ("MeshletsOffset", "<i4"), ("MeshletCount", "<i4"), ("EmissiveBias", "<f4"),
    ("SpecularBias", "<f4"), ("RoughnessBias", "<f4"), ("TransmissionBias", "<f4"), ("IORBias", "<f4"),
    ("InstanceCount", "<i4"), ("VertexCount", "<i4"), ("_pad0", "<f4", 3), ("TintOnTransmissive", "u1"), ("_pad1", "u1", 3)])
GpuMaterial = np.dtype([
    ("EmissiveFactor", "<f4", 3), ("BaseColorFactor", "<u4"), ("Absorbance", "<f4", 3), ("IOR", "<f4"),
    ("TransmissionFactor", "<f4"), ("RoughnessFactor", "<f4"), ("MetallicFactor", "<f4"), ("AlphaCutoff", "<f4"),
    ("BaseColorTexture", "<u8"), ("MetallicRoughnessTexture", "<u8"), ("NormalTexture", "<u8"), ("EmissiveTexture", "<u8"),
    ("TransmissionTexture", "<u8"), ("IsVolumetric", "u1"), ("_pad0", "u1", 3), ("IsDoubleSided", "u1"), ("_pad1", "u1", 3)])
GpuVertex = np.dtype([("TexCoord", "<f4", 2), ("Tangent", "<u4"), ("Normal", "<u4")])
GpuLight = np.dtype([("Position", "<f4", 3), ("Radius", "<f4"), ("Color", "<f4", 3), ("PointShadowIndex", "<i4"),
                     ("PrevPosition", "<f4", 3), ("_pad0", "<f4")])
GpuWavefrontRay = np.dtype([("Origin", "<f4", 3), ("PreviousIOROrTraverseCost", "<f4"), ("Throughput", "<f4", 3),
                            ("PackedDirectionX", "<f4"), ("Radiance", "<f4", 3), ("PackedDirectionY", "<f4")])
GpuUnskinnedVertex = np.dtype([("JointIndices", "<u4", 4), ("JointWeights", "<f4", 4), ("Position", "<f4", 3),
                               ("Tangent", "<u4"), ("Normal", "<u4")])

for _dt, _sz in ((GpuBlasNode, 32), (GpuBlasTriangle, 16), (GpuBlasDesc, 40), (GpuBlasInstance, 8), (GpuTlasNode, 32),
                 (GpuMeshTransform, 144), (GpuMesh, 96), (GpuMaterial, 96), (GpuVertex, 16), (GpuLight, 48),
                 (GpuWavefrontRay, 48), (GpuUnskinnedVertex, 52)):
    assert _dt.itemsize == _sz, (_dt, _dt.itemsize, _sz)


class GpuSettings(C.Structure):
    """PathTracer.GpuSettings (Source/Render/PathTracer.cs:127-138)."""
    _fields_ = [("FocalLength", C.c_float), ("LenseRadius", C.c_float), ("DoDebugBVHTraversal", C.c_int32),
                ("DoTraceLights", C.c_int32), ("DoRussianRoulette", C.c_int32)]


class Settings(C.Structure):
    """idkpt_settings (include/idkpt.h)."""
    _fields_ = [("Gpu", GpuSettings), ("RayDepth", C.c_int32), ("SamplesPerPixel", C.c_int32), ("DoRaySorting", C.c_int32),
                ("OutputAOVs", C.c_int32), ("UseTlas", C.c_int32), ("BlasStackSize", C.c_int32)]

    @staticmethod
    def default():
        s = Settings()
        s.Gpu.FocalLength = 8.0
        s.Gpu.LenseRadius = 0.0
        s.Gpu.DoDebugBVHTraversal = 0
        s.Gpu.DoTraceLights = 0
        s.Gpu.DoRussianRoulette = 1
        s.RayDepth = 7
        s.SamplesPerPixel = 1
        s.DoRaySorting = 0
        s.OutputAOVs = 0
        s.UseTlas = 0
        s.BlasStackSize = 0
        return s


class Texture(C.Structure):
    """idkpt_texture (include/idkpt.h), 32 bytes."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgba", C.c_void_p), ("wrapS", C.c_int32), ("wrapT", C.c_int32), ("magFilter", C.c_int32), ("format", C.c_int32)]


# enum idkpt_wrap / idkpt_filter / idkpt_texture_format
IDKPT_WRAP_REPEAT, IDKPT_WRAP_CLAMP_TO_EDGE, IDKPT_WRAP_MIRRORED_REPEAT = range(3)
IDKPT_FILTER_LINEAR, IDKPT_FILTER_NEAREST = range(2)
IDKPT_TEXFMT_RGBA32F, IDKPT_TEXFMT_RGBA8, IDKPT_TEXFMT_SRGB8_A8 = range(3)
GL_WRAP = {0x2901: IDKPT_WRAP_REPEAT, 0x812F: IDKPT_WRAP_CLAMP_TO_EDGE, 0x8370: IDKPT_WRAP_MIRRORED_REPEAT}      # GLSampler.WrapMode (glTF sampler.wrapS / wrapT) -> enum idkpt_wrap
GL_MAG_FILTER = {0x2601: IDKPT_FILTER_LINEAR, 0x2600: IDKPT_FILTER_NEAREST}                                       # GLSampler.MagFilter -> enum idkpt_filter


class TextureImage:
    """One image of the texture table with its sampler state: what a GpuMaterial handle stands for in the reference (a bindless texture + the GLSampler.SamplerState of
    Utils/ModelLoader.cs:1166-1197).  data: (h, w, 4) float32 (RGBA32F) or uint8 (RGBA8 / SRGB8_A8).  A bare float32 array in Scene.textures means REPEAT / REPEAT / LINEAR."""

    def __init__(self, data, wrap_s=IDKPT_WRAP_REPEAT, wrap_t=IDKPT_WRAP_REPEAT, mag_filter=IDKPT_FILTER_LINEAR, srgb=False):
        data = np.asarray(data)
        if data.dtype == np.uint8:
            self.data = np.ascontiguousarray(data); self.format = IDKPT_TEXFMT_SRGB8_A8 if srgb else IDKPT_TEXFMT_RGBA8
        else:
            assert not srgb, "sRGB storage is 8-bit"
            self.data = np.ascontiguousarray(data, np.float32); self.format = IDKPT_TEXFMT_RGBA32F
        assert self.data.ndim == 3 and self.data.shape[2] == 4
        self.wrap_s, self.wrap_t, self.mag_filter = int(wrap_s), int(wrap_t), int(mag_filter)

    @staticmethod
    def of(t):
        return t if isinstance(t, TextureImage) else TextureImage(t)

    def fill(self, rec):
        """Writes this image into a Texture record; the record borrows self.data."""
        rec.height, rec.width, rec.rgba = self.data.shape[0], self.data.shape[1], self.data.ctypes.data
        rec.wrapS, rec.wrapT, rec.magFilter, rec.format = self.wrap_s, self.wrap_t, self.mag_filter, self.format


class SceneDesc(C.Structure):
    """idkpt_scene_desc (include/idkpt.h)."""
    _fields_ = [
        ("BlasNodes", C.c_void_p), ("BlasNodeCount", C.c_int32),
        ("BlasTriangles", C.c_void_p), ("BlasTriangleCount", C.c_int32),
        ("BlasDescs", C.c_void_p), ("BlasDescCount", C.c_int32),
        ("BlasInstances", C.c_void_p), ("BlasInstanceCount", C.c_int32),
        ("TlasNodes", C.c_void_p), ("TlasNodeCount", C.c_int32),
        ("BlasParentIndices", C.c_void_p), ("BlasParentIndexCount", C.c_int32),
        ("BlasLeafIndices", C.c_void_p), ("BlasLeafIndexCount", C.c_int32),
        ("VertexPositions", C.c_void_p), ("VertexCount", C.c_int32),
        ("Vertices", C.c_void_p),
        ("Meshes", C.c_void_p), ("MeshCount", C.c_int32),
        ("Materials", C.c_void_p), ("MaterialCount", C.c_int32),
        ("MeshTransforms", C.c_void_p), ("MeshTransformCount", C.c_int32),
        ("Lights", C.c_void_p), ("LightCount", C.c_int32),
        ("SkyFaces", C.c_void_p), ("SkyFaceSize", C.c_int32),
        ("Textures", C.c_void_p), ("TextureCount", C.c_int32),
    ]


class Stats(C.Structure):
    _fields_ = [("RaysTraced", C.c_uint64), ("PrimaryRays", C.c_uint64), ("Frames", C.c_uint64),
                ("LastAliveCounts", C.c_uint32 * 16), ("LastTraceMs", C.c_float), ("LastFrameMs", C.c_float),
                ("NodePairVisits", C.c_uint64), ("TriangleTests", C.c_uint64),
                ("TraceMsTotal", C.c_double), ("TraceLaunches", C.c_uint64),
                ("WideFlaggedRays", C.c_uint64), ("WideNodeVisits", C.c_uint64), ("WideLeafRecords", C.c_uint64), ("WideTriangleTests", C.c_uint64), ("InstTlasFlaggedRays", C.c_uint64),
                ("PacketFlaggedRays", C.c_uint64), ("PacketPackets", C.c_uint64), ("PacketNodeSteps", C.c_uint64), ("PacketLiveLanes", C.c_uint64), ("PacketRaysEntered", C.c_uint64), ("PacketTriangleRounds", C.c_uint64), ("InstUnifiedLaunches", C.c_uint64), ("InstUnifiedEntries", C.c_uint32), ("InstUnifiedTopDepth", C.c_uint32)]


def _ptr(a):
    return None if a is None or len(a) == 0 else a.ctypes.data


class Scene:
    """Host-side owner of every array the path tracer consumes (the role of BVH + ModelManager + LightManager
    in the reference).  Arrays are C-contiguous numpy arrays of the dtypes above."""

    def __init__(self):
        self.blas_nodes = np.zeros(0, GpuBlasNode)
        self.blas_triangles = np.zeros(0, GpuBlasTriangle)
        self.blas_descs = np.zeros(0, GpuBlasDesc)
        self.blas_instances = np.zeros(0, GpuBlasInstance)
        self.tlas_nodes = np.zeros(0, GpuTlasNode)
        self.blas_parent_indices = np.zeros(0, np.int32)
        self.blas_leaf_indices = np.zeros(0, np.int32)
        self.vertex_positions = np.zeros((0, 3), np.float32)
        self.vertices = np.zeros(0, GpuVertex)
        self.meshes = np.zeros(0, GpuMesh)
        self.materials = np.zeros(0, GpuMaterial)
        self.mesh_transforms = np.zeros(0, GpuMeshTransform)
        self.lights = np.zeros(0, GpuLight)
        self.sky_faces = None  # (6, S, S, 4) float32
        self.textures = []     # list of (h, w, 4) float32 arrays (REPEAT, LINEAR) or TextureImage objects (sampler state, 8-bit formats)
        self.blas_stack_size = 0

    def desc(self):
        """Returns (SceneDesc, keepalive) — keepalive must outlive the call that consumes the desc."""
        keep = []

        def c(a, dt=None):
            a = np.ascontiguousarray(a, dtype=dt)
            keep.append(a)
            return a

        d = SceneDesc()
        a = c(self.blas_nodes); d.BlasNodes, d.BlasNodeCount = _ptr(a), len(a)
        a = c(self.blas_triangles); d.BlasTriangles, d.BlasTriangleCount = _ptr(a), len(a)
        a = c(self.blas_descs); d.BlasDescs, d.BlasDescCount = _ptr(a), len(a)
        a = c(self.blas_instances); d.BlasInstances, d.BlasInstanceCount = _ptr(a), len(a)
        a = c(self.tlas_nodes); d.TlasNodes, d.TlasNodeCount = _ptr(a), len(a)
        a = c(self.blas_parent_indices, np.int32); d.BlasParentIndices, d.BlasParentIndexCount = _ptr(a), len(a)
        a = c(self.blas_leaf_indices, np.int32); d.BlasLeafIndices, d.BlasLeafIndexCount = _ptr(a), len(a)
        a = c(self.vertex_positions, np.float32); d.VertexPositions, d.VertexCount = _ptr(a), len(a)
        a = c(self.vertices); d.Vertices = _ptr(a)
        assert len(a) == d.VertexCount
        a = c(self.meshes); d.Meshes, d.MeshCount = _ptr(a), len(a)
        a = c(self.materials); d.Materials, d.MaterialCount = _ptr(a), len(a)
        a = c(self.mesh_transforms); d.MeshTransforms, d.MeshTransformCount = _ptr(a), len(a)
        a = c(self.lights); d.Lights, d.LightCount = _ptr(a), len(a)
        if self.sky_faces is not None:
            a = c(self.sky_faces, np.float32)
            assert a.ndim == 4 and a.shape[0] == 6 and a.shape[1] == a.shape[2] and a.shape[3] == 4
            d.SkyFaces, d.SkyFaceSize = a.ctypes.data, a.shape[1]
        if self.textures:
            arr = (Texture * len(self.textures))()
            for i, t in enumerate(self.textures):
                t = TextureImage.of(t); keep.append(t)
                t.fill(arr[i])
            keep.append(arr)
            d.Textures, d.TextureCount = C.addressof(arr), len(self.textures)
        return d, keep

# enum idkpt_buffer (include/idkpt.h)
(IDKPT_BUF_MESH_TRANSFORMS, IDKPT_BUF_VERTEX_POSITIONS, IDKPT_BUF_VERTICES, IDKPT_BUF_MESHES, IDKPT_BUF_MATERIALS, IDKPT_BUF_LIGHTS,
 IDKPT_BUF_BLAS_NODES, IDKPT_BUF_TLAS_NODES, IDKPT_BUF_JOINT_MATRICES, IDKPT_BUF_WIDE_NODES, IDKPT_BUF_WIDE_LEAVES, IDKPT_BUF_WIDE_COUNTS) = range(12)


# ray queries / RT shadows (include/idkpt.h: idkpt_ray, idkpt_hit, idkpt_shadow_params, enum idkpt_trace_flags)
RayQuery = np.dtype([("Origin", "<f4", 3), ("MaxDist", "<f4"), ("Direction", "<f4", 3), ("_pad0", "<u4")])
RayHit = np.dtype([("T", "<f4"), ("BaryX", "<f4"), ("BaryY", "<f4"), ("TriangleId", "<u4"), ("MeshTransformId", "<u4"), ("Hit", "<u4"),
                   ("_pad0", "<u4"), ("_pad1", "<u4")])
assert RayQuery.itemsize == 32 and RayHit.itemsize == 32
IDKPT_TRACE_ANY_HIT, IDKPT_TRACE_LIGHTS = 1, 2


class ShadowParams(C.Structure):
    _fields_ = [("InvProjView", C.c_float * 16), ("TaaJitter", C.c_float * 2), ("Width", C.c_int32), ("Height", C.c_int32),
                ("LightIndex", C.c_int32), ("RayTracingSamples", C.c_int32), ("NoiseIndex", C.c_uint32), ("_pad0", C.c_uint32)]

    @staticmethod
    def make(inv_proj_view, width, height, light_index, samples=1, noise_index=0, jitter=(0.0, 0.0)):
        p = ShadowParams()
        m = np.ascontiguousarray(inv_proj_view, np.float32).reshape(16)
        for i in range(16):
            p.InvProjView[i] = float(m[i])
        p.TaaJitter[0], p.TaaJitter[1] = float(jitter[0]), float(jitter[1])
        p.Width, p.Height, p.LightIndex, p.RayTracingSamples, p.NoiseIndex = width, height, light_index, samples, noise_index
        return p
