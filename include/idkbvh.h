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
 * threads <= 0: use the hardware threads this process may occupy (affinity mask, capped by the container's cgroup CPU quota) (the reference spawns a thread per subtree >= 8192 triangles and a task per
 * sort axis >= 65536 fragments; results do not depend on the thread count).
 * Returns 0, or: 2 null / empty argument, 3 PreSplit asks for more than 2^27 fragments (split factor / priorities), 4 a triangle's split recursion needs more
 * than the 64 stack entries the reference allocates (PreSplitting.cs:57: it throws there), 5 a referenced vertex position is not finite (the reference's
 * builder has no defined result for NaN / infinite boxes).  On an error no handle is returned. */
IDKBVH_API int32_t idkbvhBuildBlas(const float* positions, const GpuBlasTriangle* tris, int32_t triCount, int32_t isRefittable,
                                   float preSplitFactor, int32_t threads, idkbvh_blas** outBlas);
/* The same build in three steps, for hosts that run the middle one elsewhere (idkptBuildBlasCore of libidkpt.so: the SweepSAH recursion on
 * the GPU).  Begin = fragments (PreSplit, or one box per triangle when refittable).  The CORE consumes the fragment boxes (8 floats each:
 * min.xyz, pad, max.xyz, pad) and produces (a) the node array of 2 * fragmentCount entries in the builder's id scheme — a subtree's ids are
 * reserved from its fragment count (Bvh/BLAS.cs:221-241), unused entries zero, no compaction yet — and (b) the final order of the x-sorted id
 * array (leaves index into it).  Finish = single-leaf root, OptimizeStackSize, RemoveEmptySubtrees, GetUnindexedTriangles, parent / leaf
 * indices, SAH; afterwards idkbvhBlasGetInfo / idkbvhBlasCopy work as after idkbvhBuildBlas.  idkbvhBuildBlas == Begin + CoreCpu + Finish. */
IDKBVH_API int32_t idkbvhBlasBegin(const float* positions, const GpuBlasTriangle* tris, int32_t triCount, int32_t isRefittable,
                                   float preSplitFactor, int32_t threads, idkbvh_blas** outBlas);
IDKBVH_API int32_t idkbvhBlasFragments(const idkbvh_blas* blas, const float** outBoxes, int32_t* outCount);   /* borrowed until idkbvhBlasFree */
IDKBVH_API int32_t idkbvhBlasCoreCpu(idkbvh_blas* blas);
IDKBVH_API int32_t idkbvhBlasCoreGet(const idkbvh_blas* blas, GpuBlasNode* nodes /* 2 * fragments, may be NULL */, int32_t* sortedIdsX /* fragments, may be NULL */);
IDKBVH_API int32_t idkbvhBlasCoreSet(idkbvh_blas* blas, const GpuBlasNode* nodes, const int32_t* sortedIdsX);
/* the builder's own core arrays, sized for the current fragments, for an external core to write into directly (instead of idkbvhBlasCoreSet's copy) */
IDKBVH_API int32_t idkbvhBlasCoreBuffers(idkbvh_blas* blas, GpuBlasNode** outNodes, int32_t** outSortedIdsX);
IDKBVH_API int32_t idkbvhBlasFinish(idkbvh_blas* blas, const float* positions, const GpuBlasTriangle* tris);
IDKBVH_API int32_t idkbvhBlasGetInfo(const idkbvh_blas* blas, idkbvh_blas_info* outInfo);
/* Copies the results into caller arrays sized from idkbvhBlasGetInfo (parents/leaves may be NULL). */
IDKBVH_API int32_t idkbvhBlasCopy(const idkbvh_blas* blas, GpuBlasNode* nodes, GpuBlasTriangle* triangles, int32_t* parentIndices, int32_t* leafIndices);
IDKBVH_API void    idkbvhBlasFree(idkbvh_blas* blas);
/* Developer knob (process-wide, default off): builds print their phase times on stderr.  The library reads no environment variable. */
IDKBVH_API void    idkbvhSetPhaseTiming(int32_t enabled);

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
