/*
 * idkbvh.h — C-ABI of libidkbvh.so: native (C++, SSE, multi-threaded) SweepSAH BLAS builder with PreSplitting,
 * PLOC TLAS builder and CPU refit.  Host-side companion of libidkpt.so: the C# engine keeps its own builder
 * (north_star), so this library is what non-.NET hosts (the Python harness, bench.py, C++ hosts) use to produce the
 * arrays idkptUploadScene consumes.  Output is bit-identical to the reference builder's (node ids, bounds, triangle
 * order, RequiredStackSize): tests compare it against the oracle's independent restatement.
 *
 * Replaces (relative to /root/reference/IDKEngine/Source):
 *   Bvh/BLAS.cs:128-274,441-534,672-937 (GetBuildData, Build, TrySplit, OptimizeStackSize, RemoveEmptySubtrees,
 *   GetUnindexedTriangles, GetParentIndices, GetLeafIndices, Refit), Bvh/PreSplitting.cs:26-273, Bvh/TLAS.cs:28-141,
 *   Bvh/BVH.cs:285-296 (instance bounds).
 */
#ifndef IDKBVH_H
#define IDKBVH_H

#include "idkpt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define IDKBVH_API __attribute__((visibility("default")))

typedef struct idkbvh_blas idkbvh_blas;

typedef struct idkbvh_blas_info {
    int32_t NodeCount;          /* GpuBlasDesc.NodeCount */
    int32_t TriangleCount;      /* GpuBlasDesc.TriangleCount (after PreSplit de-duplication) */
    int32_t RequiredStackSize;  /* GpuBlasDesc.RequiredStackSize */
    int32_t ParentIndexCount;   /* 0 unless refittable */
    int32_t LeafIndexCount;     /* 0 unless refittable */
    int32_t FragmentCount;      /* boxes fed to the builder (== input triangles when PreSplit is off) */
    double  Sah;                /* BLAS.ComputeGlobalSAH of the final tree (Bvh/BLAS.cs:629-657) */
    double  BuildMs;
} idkbvh_blas_info;

/* BVH.BlasesBuild body for one BLAS (Bvh/BVH.cs:315-375).  `tris` is this BLAS' slice of BVH.BlasTriangles as filled
 * by BVH.Add (global vertex ids + MeshId, Bvh/BVH.cs:236-276); `positions` the global packed-float3 vertex array.
 * PreSplitting (factor `preSplitFactor`, reference default 0.3) runs iff !isRefittable (Bvh/BVH.cs:325).
 * threads <= 0: use all hardware threads (the reference spawns a thread per subtree >= 8192 triangles and a task per
 * sort axis >= 65536 fragments; results do not depend on the thread count). */
IDKBVH_API int32_t idkbvhBuildBlas(const float* positions, const GpuBlasTriangle* tris, int32_t triCount, int32_t isRefittable,
                                   float preSplitFactor, int32_t threads, idkbvh_blas** outBlas);
IDKBVH_API int32_t idkbvhBlasGetInfo(const idkbvh_blas* blas, idkbvh_blas_info* outInfo);
/* Copies the results into caller arrays sized from idkbvhBlasGetInfo (parents/leaves may be NULL). */
IDKBVH_API int32_t idkbvhBlasCopy(const idkbvh_blas* blas, GpuBlasNode* nodes, GpuBlasTriangle* triangles, int32_t* parentIndices, int32_t* leafIndices);
IDKBVH_API void    idkbvhBlasFree(idkbvh_blas* blas);

/* Box.Transformed(blas.Root bounds, ModelMatrix) (Shapes/Box.cs:177-187, Bvh/BVH.cs:285-296): out = min.xyz, max.xyz */
IDKBVH_API int32_t idkbvhInstanceWorldBounds(const GpuBlasNode* blasRoot, const GpuMeshTransform* transform, float outMinMax[6]);
/* TLAS.Build (Bvh/TLAS.cs:28-141): serial PLOC over `count` leaf boxes (6 floats each); outNodes holds 2*count-1 nodes, root = 0 */
IDKBVH_API int32_t idkbvhBuildTlas(const float* leafBounds, int32_t count, int32_t searchRadius, GpuTlasNode* outNodes);
/* BLAS.Refit (Bvh/BLAS.cs:276-293): bottom-up bounds update of one BLAS' node array in place */
IDKBVH_API int32_t idkbvhRefitBlas(GpuBlasNode* nodes, int32_t nodeCount, const float* positions, const GpuBlasTriangle* blasTriangles);

#ifdef __cplusplus
}
#endif
#endif /* IDKBVH_H */
