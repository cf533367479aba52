/*
 * idkpt_types.h — byte-exact mirrors of the IDKEngine GPU structs that cross the
 * PathTracer boundary.  These ARE the ABI payload: the C# host uploads its
 * blittable `record struct`s verbatim, so every size/offset below is pinned by
 * a static_assert against the reference layout.
 *
 * Reference definitions (relative to /root/reference/IDKEngine):
 *   Source/GpuTypes/{Gpu...}.cs  <->  Resource/Shaders/include/GpuTypes.glsl
 *
 * Plain C (also valid C++ / HIP).  No torch, no HIP types.
 */
#ifndef IDKPT_TYPES_H
#define IDKPT_TYPES_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
#define IDKPT_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define IDKPT_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

/* Source/GpuTypes/GpuBlasNode.cs:7-37, GpuTypes.glsl:186-192.
 * Node 0 is padding, root = 1 (never a leaf), children are adjacent (pair is 64-B aligned). */
typedef struct GpuBlasNode {
    float    Min[3];
    uint32_t TriStartOrChild;
    float    Max[3];
    uint32_t TriCount; /* leaf <=> TriCount > 0 */
} GpuBlasNode;
IDKPT_STATIC_ASSERT(sizeof(GpuBlasNode) == 32, "GpuBlasNode");

/* Source/GpuTypes/GpuBlasTriangle.cs:3-9, GpuTypes.glsl:160-164. Global vertex ids, leaf order. */
typedef struct GpuBlasTriangle {
    uint32_t X, Y, Z;
    uint32_t MeshId;
} GpuBlasTriangle;
IDKPT_STATIC_ASSERT(sizeof(GpuBlasTriangle) == 16, "GpuBlasTriangle");

/* Source/GpuTypes/GpuBlasDesc.cs:3-20, GpuTypes.glsl:166-178. C# bool is 1 byte + 3 pad; GLSL reads 4. */
typedef struct GpuBlasDesc {
    int32_t NodeOffset;
    int32_t NodeCount;
    int32_t TriangleOffset;
    int32_t TriangleCount;
    int32_t LeafIndicesOffset;
    int32_t LeafIndicesCount;
    int32_t ParentIndicesOffset;
    int32_t ParentIndicesCount;
    int32_t RequiredStackSize;
    uint8_t IsRefittable;
    uint8_t _pad[3];
} GpuBlasDesc;
IDKPT_STATIC_ASSERT(sizeof(GpuBlasDesc) == 40, "GpuBlasDesc");

/* Source/GpuTypes/GpuBlasInstance.cs:3-7 */
typedef struct GpuBlasInstance {
    uint32_t BlasId;
    uint32_t MeshTransformId;
} GpuBlasInstance;
IDKPT_STATIC_ASSERT(sizeof(GpuBlasInstance) == 8, "GpuBlasInstance");

/* Source/GpuTypes/GpuTlasNode.cs:7-46; root = 0; bit31 = IsLeaf, low 31 bits child-or-instance id */
typedef struct GpuTlasNode {
    float    Min[3];
    uint32_t IsLeafAndChildOrInstanceId;
    float    Max[3];
    float    _pad0;
} GpuTlasNode;
IDKPT_STATIC_ASSERT(sizeof(GpuTlasNode) == 32, "GpuTlasNode");

/* Source/GpuTypes/GpuMeshTransform.cs:6-54: three Matrix3x4 (= transposed 4x3), 3 rows of 4 floats each.
 * Row r holds (M[0][r], M[1][r], M[2][r], M[3][r]) of the OpenTK row-vector Matrix4 M, so that
 * world_i = dot(Row_i.xyz, p) + Row_i.w  (GLSL: row_major mat4x3 * vec4(p,1)). */
typedef struct GpuMeshTransform {
    float Model[3][4];
    float InvModel[3][4];
    float PrevModel[3][4];
} GpuMeshTransform;
IDKPT_STATIC_ASSERT(sizeof(GpuMeshTransform) == 144, "GpuMeshTransform");

/* Source/GpuTypes/GpuMesh.cs:5-33, GpuTypes.glsl:122-145 */
typedef struct GpuMesh {
    float    LocalBoundsMin[3];
    int32_t  MaterialId;
    float    LocalBoundsMax[3];
    float    NormalMapStrength;
    float    AbsorbanceBias[3];
    int32_t  MeshletsOffset;
    int32_t  MeshletCount;
    float    EmissiveBias;
    float    SpecularBias;
    float    RoughnessBias;
    float    TransmissionBias;
    float    IORBias;
    int32_t  InstanceCount;
    int32_t  VertexCount;
    float    _pad0[3];
    uint8_t  TintOnTransmissive; /* C# bool (1 B); GLSL reads the dword */
    uint8_t  _pad1[3];
} GpuMesh;
IDKPT_STATIC_ASSERT(sizeof(GpuMesh) == 96, "GpuMesh");
IDKPT_STATIC_ASSERT(offsetof(GpuMesh, TintOnTransmissive) == 92, "GpuMesh.Tint");

/* Source/GpuTypes/GpuMaterial.cs:8-67, GpuTypes.glsl:226-248.
 * The five 64-bit fields are GL bindless sampler handles in the reference. Behind this ABI
 * they are interpreted as texture-table ids: 0 = the 1x1 white default
 * (Utils/ModelLoader.cs:1857-1877), k>0 = texture k-1 of idkpt_scene_desc.Textures. */
typedef struct GpuMaterial {
    float    EmissiveFactor[3];
    uint32_t BaseColorFactor; /* RGBA8 unorm, R in low byte */
    float    Absorbance[3];
    float    IOR;
    float    TransmissionFactor;
    float    RoughnessFactor;
    float    MetallicFactor;
    float    AlphaCutoff; /* 2.0 means alpha blending */
    uint64_t BaseColorTexture;
    uint64_t MetallicRoughnessTexture;
    uint64_t NormalTexture;
    uint64_t EmissiveTexture;
    uint64_t TransmissionTexture;
    uint8_t  IsVolumetric;
    uint8_t  _pad0[3];
    uint8_t  IsDoubleSided;
    uint8_t  _pad1[3];
} GpuMaterial;
IDKPT_STATIC_ASSERT(sizeof(GpuMaterial) == 96, "GpuMaterial");
IDKPT_STATIC_ASSERT(offsetof(GpuMaterial, BaseColorTexture) == 48, "GpuMaterial.tex");
IDKPT_STATIC_ASSERT(offsetof(GpuMaterial, IsVolumetric) == 88, "GpuMaterial.vol");

/* Source/GpuTypes/GpuVertex.cs:5-10; Tangent/Normal are SR11G11B10 (Utils/Compression.cs:26-40) */
typedef struct GpuVertex {
    float    TexCoord[2];
    uint32_t Tangent;
    uint32_t Normal;
} GpuVertex;
IDKPT_STATIC_ASSERT(sizeof(GpuVertex) == 16, "GpuVertex");

/* Source/GpuTypes/GpuLight.cs:5-45 */
typedef struct GpuLight {
    float   Position[3];
    float   Radius;
    float   Color[3];
    int32_t PointShadowIndex;
    float   PrevPosition[3];
    float   _pad0;
} GpuLight;
IDKPT_STATIC_ASSERT(sizeof(GpuLight) == 48, "GpuLight");
#define IDKPT_MAX_LIGHTS 256 /* StaticUniformBuffers.glsl:9 GPU_MAX_UBO_LIGHT_COUNT */

/* Source/GpuTypes/GpuPerFrameData.cs:5-21. Matrices are in OpenTK memory order (row-vector
 * convention, 16 consecutive floats); GLSL sees the transpose, so GLSL `M * v` == C# `v * M`.
 * The path tracer consumes only InvView, ViewPos (CameraPos) and InvProjection (FirstHit:58-60). */
typedef struct GpuPerFrameData {
    float    ProjView[16];
    float    View[16];
    float    InvView[16];
    float    PrevView[16];
    float    ViewPos[3];
    uint32_t Frame;
    float    Projection[16];
    float    InvProjection[16];
    float    InvProjView[16];
    float    PrevProjView[16];
    float    NearPlane;
    float    FarPlane;
    float    DeltaRenderTime;
    float    Time;
} GpuPerFrameData;
IDKPT_STATIC_ASSERT(sizeof(GpuPerFrameData) == 544, "GpuPerFrameData");
IDKPT_STATIC_ASSERT(offsetof(GpuPerFrameData, InvView) == 128, "PerFrame.InvView");
IDKPT_STATIC_ASSERT(offsetof(GpuPerFrameData, ViewPos) == 256, "PerFrame.ViewPos");
IDKPT_STATIC_ASSERT(offsetof(GpuPerFrameData, InvProjection) == 336, "PerFrame.InvProjection");

/* Source/Render/PathTracer.cs:127-138 `GpuSettings` (std140 UBO 0, FirstHit/compute.glsl:28-35) */
typedef struct GpuSettings {
    float   FocalLength;          /* default 8.0 */
    float   LenseRadius;          /* default 0.0 */
    int32_t DoDebugBVHTraversal;  /* default 0 */
    int32_t DoTraceLights;        /* default 0 */
    int32_t DoRussianRoulette;    /* default 1 */
} GpuSettings;
IDKPT_STATIC_ASSERT(sizeof(GpuSettings) == 20, "GpuSettings");

/* Source/GpuTypes/GpuWavefrontRay.cs:5-15 — the reference's internal ray state (SSBO 30).
 * Kept here because idkptDownloadRays() exposes it for parity checks; in HBM the library stores
 * the same fields as SoA planes (DESIGN.md "Data layout"). */
typedef struct GpuWavefrontRay {
    float Origin[3];
    float PreviousIOROrTraverseCost;
    float Throughput[3];
    float PackedDirectionX;
    float Radiance[3];
    float PackedDirectionY;
} GpuWavefrontRay;
IDKPT_STATIC_ASSERT(sizeof(GpuWavefrontRay) == 48, "GpuWavefrontRay");

/* Source/GpuTypes/GpuAovRay.cs:5-11 */
typedef struct GpuAovRay {
    float Albedo[3];
    float NewWeight;
    float Normal[3];
    float _pad0;
} GpuAovRay;
IDKPT_STATIC_ASSERT(sizeof(GpuAovRay) == 32, "GpuAovRay");

/* Shaders/include/GpuTypes.glsl (UnskinnedVertex) — Skinning/compute.glsl input */
typedef struct GpuUnskinnedVertex {
    uint32_t JointIndices[4];
    float    JointWeights[4];
    float    Position[3];
    uint32_t Tangent;
    uint32_t Normal;
} GpuUnskinnedVertex;
IDKPT_STATIC_ASSERT(sizeof(GpuUnskinnedVertex) == 52, "GpuUnskinnedVertex");

/* Sort key capacity: CountingSort/BlellochScan/include/Constants.glsl:1-3 (10 + 11 bits) */
#define IDKPT_SORT_KEY_BITS 21

#endif /* IDKPT_TYPES_H */
