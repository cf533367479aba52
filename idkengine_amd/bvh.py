"""NativeBuilder — ctypes front-end of libidkbvh.so (include/idkbvh.h): the product's SweepSAH(+PreSplit) BLAS
builder, PLOC TLAS builder and CPU refit.  Same builder interface scenes.assemble() expects."""
import ctypes as C
import os
import numpy as np
from . import gputypes as T

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libidkbvh.so")
SYMBOLS = ["idkbvhBuildBlas", "idkbvhBlasBegin", "idkbvhBlasFragments", "idkbvhBlasCoreCpu", "idkbvhBlasCoreGet", "idkbvhBlasCoreSet", "idkbvhBlasCoreBuffers", "idkbvhBlasFinish", "idkbvhBlasGetInfo", "idkbvhBlasCopy", "idkbvhBlasFree", "idkbvhSetPhaseTiming", "idkbvhInstanceWorldBounds", "idkbvhBuildTlas", "idkbvhRefitBlas"]
_lib = None


class BlasInfo(C.Structure):
    _fields_ = [("NodeCount", C.c_int32), ("TriangleCount", C.c_int32), ("RequiredStackSize", C.c_int32), ("ParentIndexCount", C.c_int32),
                ("LeafIndexCount", C.c_int32), ("FragmentCount", C.c_int32), ("Sah", C.c_double), ("BuildMs", C.c_double)]


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libidkbvh.so not built: run __graft_entry__.build()")
        L = C.CDLL(LIB_PATH)
        L.idkbvhBuildBlas.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.POINTER(C.c_void_p)]
        L.idkbvhBlasBegin.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.POINTER(C.c_void_p)]
        L.idkbvhBlasFragments.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]
        L.idkbvhBlasCoreCpu.argtypes = [C.c_void_p]
        L.idkbvhBlasCoreGet.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.idkbvhBlasCoreSet.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.idkbvhBlasCoreBuffers.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.idkbvhBlasFinish.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.idkbvhBlasGetInfo.argtypes = [C.c_void_p, C.c_void_p]
        L.idkbvhBlasCopy.argtypes = [C.c_void_p] * 5
        L.idkbvhBlasFree.argtypes = [C.c_void_p]; L.idkbvhBlasFree.restype = None
        L.idkbvhSetPhaseTiming.argtypes = [C.c_int32]; L.idkbvhSetPhaseTiming.restype = None
        L.idkbvhInstanceWorldBounds.argtypes = [C.c_void_p] * 3
        L.idkbvhBuildTlas.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        L.idkbvhRefitBlas.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


class NativeBuilder:
    def __init__(self, presplit_factor=0.3, threads=0):
        self.presplit_factor = presplit_factor
        self.threads = threads
        self.last_build_ms = 0.0

    def build_blas(self, positions, tris, refittable):
        L = load()
        positions = np.ascontiguousarray(positions, np.float32); tris = np.ascontiguousarray(tris)
        h = C.c_void_p()
        rc = L.idkbvhBuildBlas(positions.ctypes.data, tris.ctypes.data, len(tris), 1 if refittable else 0, self.presplit_factor, self.threads, C.byref(h))
        if rc != 0:
            raise RuntimeError(f"idkbvhBuildBlas failed: {rc}")
        try:
            info = BlasInfo(); L.idkbvhBlasGetInfo(h, C.addressof(info))
            nodes = np.zeros(info.NodeCount, T.GpuBlasNode); out_tris = np.zeros(info.TriangleCount, T.GpuBlasTriangle)
            parents = np.zeros(info.ParentIndexCount, np.int32); leaves = np.zeros(info.LeafIndexCount, np.int32)
            L.idkbvhBlasCopy(h, nodes.ctypes.data, out_tris.ctypes.data, parents.ctypes.data if len(parents) else None, leaves.ctypes.data if len(leaves) else None)
            self.last_build_ms = info.BuildMs
            return {"nodes": nodes, "triangles": out_tris, "parents": parents, "leaves": leaves, "required_stack_size": info.RequiredStackSize,
                    "sah": info.Sah, "fragments": info.FragmentCount, "build_ms": info.BuildMs}
        finally:
            L.idkbvhBlasFree(h)

    def core_arrays(self, positions, tris, refittable):
        """(fragment boxes, node array before compaction, final x-sorted id order) of the CPU core: what idkptBuildBlasCore must reproduce."""
        L = load()
        positions = np.ascontiguousarray(positions, np.float32); tris = np.ascontiguousarray(tris)
        h = C.c_void_p()
        if L.idkbvhBlasBegin(positions.ctypes.data, tris.ctypes.data, len(tris), 1 if refittable else 0, self.presplit_factor, self.threads, C.byref(h)) != 0:
            raise RuntimeError("idkbvhBlasBegin failed")
        try:
            bp = C.c_void_p(); n = C.c_int32()
            L.idkbvhBlasFragments(h, C.byref(bp), C.byref(n))
            boxes = np.ctypeslib.as_array(C.cast(bp, C.POINTER(C.c_float)), shape=(n.value, 8)).copy()
            L.idkbvhBlasCoreCpu(h)
            nodes = np.zeros(max(2 * n.value, 4), T.GpuBlasNode); order = np.zeros(n.value, np.int32)
            L.idkbvhBlasCoreGet(h, nodes.ctypes.data, order.ctypes.data)
            return boxes, nodes, order
        finally:
            L.idkbvhBlasFree(h)

    def instance_world_bounds(self, root_node, xform):
        out = np.zeros(6, np.float32)
        root_node = np.ascontiguousarray(root_node); xform = np.ascontiguousarray(xform)
        load().idkbvhInstanceWorldBounds(root_node.ctypes.data, xform.ctypes.data, out.ctypes.data)
        return out

    def build_tlas(self, leaf_bounds, search_radius=15):
        leaf_bounds = np.ascontiguousarray(leaf_bounds, np.float32)
        n = len(leaf_bounds)
        nodes = np.zeros(max(2 * n - 1, 0), T.GpuTlasNode)
        if n:
            load().idkbvhBuildTlas(leaf_bounds.ctypes.data, n, search_radius, nodes.ctypes.data)
        return nodes

    def refit(self, nodes, positions, tris):
        nodes = np.ascontiguousarray(nodes).copy(); positions = np.ascontiguousarray(positions, np.float32); tris = np.ascontiguousarray(tris)
        load().idkbvhRefitBlas(nodes.ctypes.data, len(nodes), positions.ctypes.data, tris.ctypes.data)
        return nodes


class GpuBuilder(NativeBuilder):
    """BLAS build with the SweepSAH core on the GPU: idkbvhBlasBegin (fragments / PreSplit, host) -> idkptBuildBlasCore (sort + recursion,
    libidkpt.so on the context's device) -> idkbvhBlasFinish (stack-size optimisation, compaction, un-indexing, host).  Same bytes as
    NativeBuilder.  `pt` is any idkengine_amd.pathtracer.PathTracer (only its context handle is used; no scene is needed)."""

    def __init__(self, pt, presplit_factor=0.3, threads=0):
        super().__init__(presplit_factor, threads)
        self._pt = pt
        self.last_levels = 0; self.last_core_ms = 0.0

    def core_on_gpu(self, boxes):
        import time
        boxes = np.ascontiguousarray(boxes, np.float32)
        n = len(boxes)
        nodes = np.zeros(max(2 * n, 4), T.GpuBlasNode); order = np.zeros(n, np.int32); lv = C.c_int32()
        t0 = time.perf_counter()
        self._pt._check(self._pt._L.idkptBuildBlasCore(self._pt._ctx, boxes.ctypes.data, n, nodes.ctypes.data, order.ctypes.data, C.byref(lv)))
        self.last_core_ms = (time.perf_counter() - t0) * 1e3; self.last_levels = lv.value
        return nodes, order

    def build_blas(self, positions, tris, refittable):
        L = load()
        positions = np.ascontiguousarray(positions, np.float32); tris = np.ascontiguousarray(tris)
        h = C.c_void_p()
        if L.idkbvhBlasBegin(positions.ctypes.data, tris.ctypes.data, len(tris), 1 if refittable else 0, self.presplit_factor, self.threads, C.byref(h)) != 0:
            raise RuntimeError("idkbvhBlasBegin failed")
        try:
            bp = C.c_void_p(); n = C.c_int32()
            L.idkbvhBlasFragments(h, C.byref(bp), C.byref(n))
            # the GPU core writes straight into the builder's own arrays
            pn = C.c_void_p(); po = C.c_void_p(); lv = C.c_int32()
            if L.idkbvhBlasCoreBuffers(h, C.byref(pn), C.byref(po)) != 0:
                raise RuntimeError("idkbvhBlasCoreBuffers failed")
            self._pt._check(self._pt._L.idkptBuildBlasCore(self._pt._ctx, bp, n.value, pn, po, C.byref(lv)))
            self.last_levels = lv.value
            if L.idkbvhBlasFinish(h, positions.ctypes.data, tris.ctypes.data) != 0:
                raise RuntimeError("idkbvhBlasFinish failed")
            info = BlasInfo(); L.idkbvhBlasGetInfo(h, C.addressof(info))
            out_nodes = np.zeros(info.NodeCount, T.GpuBlasNode); out_tris = np.zeros(info.TriangleCount, T.GpuBlasTriangle)
            parents = np.zeros(info.ParentIndexCount, np.int32); leaves = np.zeros(info.LeafIndexCount, np.int32)
            L.idkbvhBlasCopy(h, out_nodes.ctypes.data, out_tris.ctypes.data, parents.ctypes.data if len(parents) else None, leaves.ctypes.data if len(leaves) else None)
            self.last_build_ms = info.BuildMs
            return {"nodes": out_nodes, "triangles": out_tris, "parents": parents, "leaves": leaves, "required_stack_size": info.RequiredStackSize,
                    "sah": info.Sah, "fragments": info.FragmentCount, "build_ms": info.BuildMs}
        finally:
            L.idkbvhBlasFree(h)


class BlasBuildInfo(C.Structure):
    _fields_ = [("NodeCount", C.c_int32), ("TriangleCount", C.c_int32), ("RequiredStackSize", C.c_int32), ("ParentIndexCount", C.c_int32), ("LeafIndexCount", C.c_int32),
                ("FragmentCount", C.c_int32), ("Levels", C.c_int32), ("_pad", C.c_int32), ("Sah", C.c_double), ("BuildMs", C.c_double)]


class DeviceBuilder(NativeBuilder):
    """The whole BLAS build on the GPU (idkptBuildBlas / idkptBuildBlasFetch, csrc/bvh_gpu_full.hpp): PreSplit, SweepSAH, stack-size optimisation,
    compaction, un-indexing, parent / leaf indices.  Same bytes as NativeBuilder (tests/test_gpu_builder.py).  TLAS build and refit stay the
    inherited host routines.  `pt` is any idkengine_amd.pathtracer.PathTracer (only its context handle is used)."""

    def __init__(self, pt, presplit_factor=0.3, threads=0):
        super().__init__(presplit_factor, threads)
        self._pt = pt
        self.last_levels = 0

    def build_blas(self, positions, tris, refittable):
        import time
        positions = np.ascontiguousarray(positions, np.float32).reshape(-1, 3); tris = np.ascontiguousarray(tris)
        info = BlasBuildInfo()
        t0 = time.perf_counter()
        self._pt._check(self._pt._L.idkptBuildBlas(self._pt._ctx, positions.ctypes.data, len(positions), tris.ctypes.data, len(tris), 1 if refittable else 0,
                                                   C.c_float(self.presplit_factor), C.addressof(info)))
        out_nodes = np.empty(info.NodeCount, T.GpuBlasNode); out_tris = np.empty(info.TriangleCount, T.GpuBlasTriangle)
        parents = np.empty(info.ParentIndexCount, np.int32); leaves = np.empty(info.LeafIndexCount, np.int32)
        self._pt._check(self._pt._L.idkptBuildBlasFetch(self._pt._ctx, out_nodes.ctypes.data, out_tris.ctypes.data, parents.ctypes.data if len(parents) else None,
                                                        leaves.ctypes.data if len(leaves) else None))
        self.last_build_ms = (time.perf_counter() - t0) * 1e3; self.last_levels = info.Levels; self.last_device_ms = info.BuildMs
        return {"nodes": out_nodes, "triangles": out_tris, "parents": parents, "leaves": leaves, "required_stack_size": info.RequiredStackSize,
                "sah": info.Sah, "fragments": info.FragmentCount, "build_ms": self.last_build_ms}
