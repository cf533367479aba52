// transport_rccl.hpp — RCCL as the device-to-device transport of a multi-device context (idkptCreate(deviceCount = N)): ncclBroadcast replicates the scene that
// member 0 uploaded, grouped ncclSend / ncclRecv gather the members' rows of a frame on device 0 (RCCL, like NCCL, has no gather collective: a group of
// point-to-point calls is how it expresses one).  north_star: "RCCL broadcast of the BVH + gather of tiles over xGMI".
// The library does not link against RCCL: librccl.so is probed with dlopen at run time, in the style of the engine's own optional native library
// (NativeLibrary.TryLoad, Source/OIDN/OIDN.cs:11-20).  No RCCL, a communicator that cannot be formed (two members on one GPU: the one-GPU test boxes), or any
// failing call: the context keeps / falls back to its peer copies (hipMemcpyPeerAsync, host_scene.hpp member_copy) and says so (idkptGetTransportInfo).
// Part of the single translation unit idkpt.hip (included there, in this order).
#pragma once
#include <dlfcn.h>

struct RcclApi {
    void* lib = nullptr;
    // (ncclResult_t is an int-sized enum, 0 = success; ncclComm_t an opaque pointer; ncclDataType_t an int-sized enum with ncclChar = ncclInt8 = 0: rccl.h of ROCm 7)
    int (*GetVersion)(int*) = nullptr;
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string path;
};
// loads the library once per process; returns null (and why) when it is not there or lacks a symbol
static RcclApi* rccl_api(std::string* why)
{
    static RcclApi api; static bool tried = false; static std::string err;
    if (!tried) {
        tried = true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};   // (a process that already loaded RCCL — torch — gets that copy back by its soname)
        for (const char* n : names) { api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (api.lib) { api.path = n; break; } }
        if (!api.lib) err = std::string("librccl.so not found (") + (dlerror() ? dlerror() : "dlopen failed") + ")";
        else {
            bool ok = true;
            auto sym = [&](const char* s) { void* p = dlsym(api.lib, s); if (!p) { ok = false; err = std::string("librccl: missing symbol ") + s; } return p; };
            api.GetVersion = (int (*)(int*))sym("ncclGetVersion");
            api.CommInitAll = (int (*)(void**, int, const int*))sym("ncclCommInitAll");
            api.CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
            api.GroupStart = (int (*)())sym("ncclGroupStart");
            api.GroupEnd = (int (*)())sym("ncclGroupEnd");
            api.Broadcast = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))sym("ncclBroadcast");
            api.Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))sym("ncclSend");
            api.Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))sym("ncclRecv");
            api.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))sym("ncclAllGather");
            api.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
            if (!ok) { dlclose(api.lib); api.lib = nullptr; }
        }
    }
    if (!api.lib) { if (why) *why = err; return nullptr; }
    return &api;
}

// One communicator per member of a multi-device context (single process, N devices: ncclCommInitAll).
struct RcclTransport {
    RcclApi* api = nullptr;
    std::vector<void*> comms;            // rank d = member d
    int version = 0;
    bool ready = false, tried = false;
    std::string why;                     // why RCCL is not in use (empty while it is)
    std::string lastError;
    bool ok(int rc, const char* what) { if (rc == 0) return true; lastError = std::string(what) + ": " + (api && api->GetErrorString ? api->GetErrorString(rc) : "error") + " (" + std::to_string(rc) + ")"; return false; }
    // forms the communicators; false (with `why`) when RCCL cannot serve this set of devices
    bool init(const std::vector<int>& devices)
    {
        if (tried) return ready;
        tried = true;
        for (size_t a = 0; a < devices.size(); a++) for (size_t b = a + 1; b < devices.size(); b++)
            if (devices[a] == devices[b]) { why = "two members share GPU " + std::to_string(devices[a]) + " (RCCL wants one rank per device)"; return false; }
        api = rccl_api(&why);
        if (!api) return false;
        (void)api->GetVersion(&version);
        comms.assign(devices.size(), nullptr);
        int cur = 0; (void)hipGetDevice(&cur);
        const int rc = api->CommInitAll(comms.data(), (int)devices.size(), devices.data());
        (void)hipSetDevice(cur);
        if (rc != 0) { why = std::string("ncclCommInitAll: ") + (api->GetErrorString ? api->GetErrorString(rc) : "error"); comms.clear(); (void)hipGetLastError(); return false; }
        ready = true; why.clear();
        return true;
    }
    void shutdown() { if (api) for (void* cm : comms) if (cm) (void)api->CommDestroy(cm); comms.clear(); ready = false; }
    // every member's copy of one buffer from member 0's: dst[0] is the source itself (in place on the root), streams[d] the member's stream
    bool broadcast(const std::vector<void*>& dst, size_t bytes, const std::vector<int>& devices, const std::vector<hipStream_t>& streams)
    {
        if (bytes == 0) return true;
        if (!ok(api->GroupStart(), "ncclGroupStart")) return false;
        bool good = true;
        for (size_t d = 0; d < comms.size() && good; d++) {
            (void)hipSetDevice(devices[d]);
            good = ok(api->Broadcast(dst[0], dst[d], bytes, 0 /* ncclChar */, 0, comms[d], streams[d]), "ncclBroadcast");
        }
        const bool ended = ok(api->GroupEnd(), "ncclGroupEnd");
        return good && ended;
    }
    // member d's `bytes[d]` at src[d] -> dst[d] on member 0 (d >= 1); member 0's own part is the caller's (a local copy)
    bool gather(const std::vector<const void*>& src, const std::vector<void*>& dst, const std::vector<size_t>& bytes, const std::vector<int>& devices, const std::vector<hipStream_t>& streams)
    {
        if (!ok(api->GroupStart(), "ncclGroupStart")) return false;
        bool good = true;
        for (size_t d = 1; d < comms.size() && good; d++) {
            if (bytes[d] == 0) continue;
            (void)hipSetDevice(devices[d]);
            good = ok(api->Send(src[d], bytes[d], 0, 0, comms[d], streams[d]), "ncclSend");
            (void)hipSetDevice(devices[0]);
            good = good && ok(api->Recv(dst[d], bytes[d], 0, (int)d, comms[0], streams[0]), "ncclRecv");
        }
        const bool ended = ok(api->GroupEnd(), "ncclGroupEnd");
        return good && ended;
    }
};

// idkptTransportSelfTest: loads RCCL and runs a one-rank communicator on `device` through every call the transport uses — broadcast (in place), send + recv to itself,
// all-gather — checking the bytes.  What a host (or a test on a one-GPU box) can run to know that the RCCL path is usable before it creates an N-device context.
static int32_t rccl_self_test(int device, int32_t* outVersion, std::string* why)
{
    RcclApi* api = rccl_api(why);
    if (outVersion) *outVersion = 0;
    if (!api) return IDKPT_ERR_INVALID_OPERATION;
    int ver = 0; (void)api->GetVersion(&ver); if (outVersion) *outVersion = ver;
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); *why = "hipSetDevice failed"; return IDKPT_ERR_HIP; }
    void* comm = nullptr;
    int rc = api->CommInitAll(&comm, 1, &device);
    if (rc != 0) { *why = std::string("ncclCommInitAll: ") + api->GetErrorString(rc); return IDKPT_ERR_UNKNOWN; }
    const size_t n = 1 << 20;
    unsigned char *a = nullptr, *b = nullptr; hipStream_t st = nullptr;
    bool good = hipMalloc((void**)&a, n) == hipSuccess && hipMalloc((void**)&b, 2 * n) == hipSuccess && hipStreamCreate(&st) == hipSuccess;
    std::vector<unsigned char> h(n), g(2 * n, 0);
    for (size_t i = 0; i < n; i++) h[i] = (unsigned char)((i * 131u + 7u) >> 3);
    good = good && hipMemcpy(a, h.data(), n, hipMemcpyHostToDevice) == hipSuccess && hipMemset(b, 0, 2 * n) == hipSuccess;
    std::string step;
    if (good) {
        step = "ncclBroadcast"; good = api->Broadcast(a, a, n, 0, 0, comm, st) == 0;
        if (good) { step = "ncclSend/ncclRecv"; good = api->GroupStart() == 0 && api->Send(a, n, 0, 0, comm, st) == 0 && api->Recv(b, n, 0, 0, comm, st) == 0 && api->GroupEnd() == 0; }
        if (good) { step = "ncclAllGather"; good = api->AllGather(a, b + n, n, 0, comm, st) == 0; }
        if (good) { step = "synchronise"; good = hipStreamSynchronize(st) == hipSuccess && hipMemcpy(g.data(), b, 2 * n, hipMemcpyDeviceToHost) == hipSuccess; }
        if (good) { step = "compare"; good = memcmp(g.data(), h.data(), n) == 0 && memcmp(g.data() + n, h.data(), n) == 0; }
    } else step = "allocation";
    if (st) (void)hipStreamDestroy(st);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    (void)api->CommDestroy(comm);
    if (!good) { (void)hipGetLastError(); *why = "RCCL self test failed at " + step; return IDKPT_ERR_UNKNOWN; }
    return IDKPT_OK;
}
