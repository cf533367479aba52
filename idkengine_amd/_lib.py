"""ctypes binding of libidkpt.so (include/idkpt.h).  There is NO fallback: if the HIP library is missing or no GPU is
visible, creating a PathTracer raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IDKPT_LIB_PATH") or os.path.join(_HERE, "libidkpt.so")   # (the override is a developer knob: A/B runs of two builds on one box)

# every symbol include/idkpt.h declares (tests/test_abi.py checks the header against this list and the .so)
SYMBOLS = [
    "idkptCreate", "idkptDestroy", "idkptGetLastError", "idkptSetErrorCallback", "idkptGetDeviceCount", "idkptGetVersionString", "idkptGetAbiVersion", "idkptGetContextDeviceCount", "idkptGetTransportInfo", "idkptTransportSelfTest", "idkptSetGroupSharding",
    "idkptSetSize", "idkptSetSceneVersions", "idkptSetRowSharding", "idkptSetRowBands", "idkptSetRowRange", "idkptSetBounceExchange", "idkptSetBandExchange", "idkptSetBandExchangeDevice", "idkptSetSettings", "idkptGetSettings",
    "idkptSetPerFrame", "idkptSetPerFrameData", "idkptUploadScene", "idkptUpdateBuffer", "idkptUpdateTexture", "idkptSetLightCount",
    "idkptBuildTlas", "idkptBuildTlasOnDevice", "idkptBuildBlasCore", "idkptBuildBlas", "idkptBuildBlasFetch", "idkptCbrtProbe", "idkptRefitBlas", "idkptUploadUnskinnedVertices", "idkptSkin", "idkptDownloadBuffer",
    "idkptResetAccumulation", "idkptGetAccumulatedSamples", "idkptSetSampleSequence", "idkptRender", "idkptSynchronize", "idkptDownload",
    "idkptDownloadRays", "idkptDownloadAliveQueue", "idkptEnablePrimaryHitCapture", "idkptDownloadPrimaryHits",
    "idkptGetStats", "idkptGetStatsSized", "idkptResetStats", "idkptEnableCounters", "idkptEnableTiming", "idkptGetImageDevicePtr",
    "idkptSetStream", "idkptGetStream", "idkptSetDeveloperOption", "idkptSetMaxBatch", "idkptFlush", "idkptTraceRays", "idkptTraceShadows", "idkptTraceRaysDevice", "idkptTraceShadowsDevice", "idkptSetFrameRing", "idkptBeginFrame", "idkptDownloadFrame", "idkptGetFrameDevicePtr",
]

_lib = None


class IdkPtError(RuntimeError):
    pass


def load():
    """Loads libidkpt.so; raises IdkPtError (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IdkPtError("libidkpt.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(hipcc --offload-arch=gfx950); there is no CPU fallback for the path tracer")
    L = C.CDLL(LIB_PATH)
    vp, i32, u32, sz = C.c_void_p, C.c_int32, C.c_uint32, C.c_size_t
    sig = {
        "idkptCreate": [i32, C.POINTER(i32), C.POINTER(vp)], "idkptDestroy": [vp], "idkptGetLastError": [vp, C.POINTER(C.c_char_p)], "idkptSetErrorCallback": [vp, vp, vp],
        "idkptGetDeviceCount": [C.POINTER(i32)], "idkptGetContextDeviceCount": [vp, C.POINTER(i32)], "idkptGetTransportInfo": [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(C.c_char_p)], "idkptTransportSelfTest": [i32, C.POINTER(i32), C.c_char_p, sz], "idkptSetGroupSharding": [vp, i32], "idkptSetSize": [vp, i32, i32], "idkptSetRowSharding": [vp, i32, i32], "idkptSetSceneVersions": [vp, i32], "idkptSetRowBands": [vp, i32, i32, i32],
        "idkptSetRowRange": [vp, i32, i32], "idkptSetBounceExchange": [vp, vp, vp], "idkptSetBandExchange": [vp, vp, vp], "idkptSetBandExchangeDevice": [vp, vp, vp], "idkptSetSettings": [vp, vp], "idkptGetSettings": [vp, vp],
        "idkptSetPerFrame": [vp, vp, vp, vp], "idkptSetPerFrameData": [vp, vp], "idkptUploadScene": [vp, vp],
        "idkptUpdateBuffer": [vp, i32, sz, sz, vp], "idkptUpdateTexture": [vp, i32, vp], "idkptSetLightCount": [vp, i32], "idkptBuildTlas": [vp, vp, i32], "idkptBuildTlasOnDevice": [vp, i32], "idkptBuildBlasCore": [vp, vp, i32, vp, vp, vp], "idkptBuildBlas": [vp, vp, i32, vp, i32, i32, C.c_float, vp], "idkptBuildBlasFetch": [vp, vp, vp, vp, vp], "idkptCbrtProbe": [vp, vp, vp, i32], "idkptTraceRays": [vp, vp, sz, u32, vp], "idkptTraceShadows": [vp, vp, vp, vp, vp], "idkptTraceRaysDevice": [vp, vp, sz, u32, vp], "idkptTraceShadowsDevice": [vp, vp, vp, vp, vp], "idkptSetFrameRing": [vp, i32], "idkptBeginFrame": [vp, C.POINTER(i32)],
        "idkptDownloadFrame": [vp, i32, i32, vp, sz], "idkptGetFrameDevicePtr": [vp, i32, i32, C.POINTER(vp), C.POINTER(sz)],
        "idkptRefitBlas": [vp, i32], "idkptUploadUnskinnedVertices": [vp, vp, i32], "idkptSkin": [vp, u32, u32, u32, u32],
        "idkptDownloadBuffer": [vp, i32, sz, sz, vp], "idkptResetAccumulation": [vp], "idkptSetSampleSequence": [vp, u32, u32], "idkptGetAccumulatedSamples": [vp, C.POINTER(u32)],
        "idkptRender": [vp], "idkptSynchronize": [vp], "idkptDownload": [vp, i32, vp, sz], "idkptDownloadRays": [vp, vp, sz],
        "idkptDownloadAliveQueue": [vp, vp, sz, C.POINTER(u32)], "idkptEnablePrimaryHitCapture": [vp, i32],
        "idkptDownloadPrimaryHits": [vp, vp, vp, vp, sz], "idkptGetStats": [vp, vp], "idkptGetStatsSized": [vp, vp, sz], "idkptResetStats": [vp],
        "idkptEnableCounters": [vp, i32], "idkptEnableTiming": [vp, i32], "idkptGetImageDevicePtr": [vp, i32, C.POINTER(vp), C.POINTER(sz)],
        "idkptSetStream": [vp, vp], "idkptGetStream": [vp, C.POINTER(vp)], "idkptSetDeveloperOption": [vp, C.c_char_p, i32], "idkptSetMaxBatch": [vp, i32], "idkptFlush": [vp],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = i32
    L.idkptGetAbiVersion.argtypes = []
    L.idkptGetAbiVersion.restype = i32
    L.idkptGetVersionString.argtypes = []
    L.idkptGetVersionString.restype = C.c_char_p
    _lib = L
    return L
