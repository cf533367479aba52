// idkpt.hip — libidkpt.so: kernels' __global__ entry points and the C-ABI of include/idkpt.h.
// Host schedule mirrors PathTracer.Compute (Source/Render/PathTracer.cs:214-271); see DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <chrono>
#include <algorithm>
#include "../../include/idkpt.h"
#include "pt_kernels.hpp"
#include "node_layout.hpp"

using namespace ptd;

#define WAVE 64
#define MAX_DEPTH_SLOTS 64
#define WORK_WORDS (GRAB_SLICES * GRAB_STRIDE)   // work counters: launch j uses word j of every 512-B line (k_trace2: one line per work-list slice, kernels_trace.hpp)

// =================================================================================================== kernels (one translation unit)
#include "kernels_common.hpp"
#include "kernels_trace.hpp"
#include "kernels_wide.hpp"
#include "kernels_trace_split.hpp"
#include "kernels_trace_quad.hpp"
#include "kernels_trace_park.hpp"
#include "kernels_query.hpp"
#include "kernels_shade.hpp"
#include "kernels_trace_fused.hpp"
#include "kernels_queue.hpp"
#include "kernels_frame.hpp"
#include "kernels_scene.hpp"
#include "bvh_gpu.hpp"
#include "bvh_gpu_full.hpp"

// =================================================================================================== host side

struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    hipError_t ensure(size_t n) { if (n <= bytes && p) return hipSuccess; if (p) (void)hipFree(p); p = nullptr; bytes = 0; if (n == 0) return hipSuccess; hipError_t e = hipMalloc(&p, n); if (e == hipSuccess) bytes = n; return e; }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <class T> T* as() const { return (T*)p; }
};

// Tuning / test options (idkptSetDeveloperOption; none is part of the reference's interface and results are bit-identical under all of them:
// tests/test_gpu_worklist.py, test_gpu_layout.py).  The library itself never reads the environment; the Python host mirror forwards IDKPT_<NAME>.
struct DevOptions {
    int forceGeneric = 0;        // thread-per-ray kernels instead of k_trace2 (the cross-check of the tests)
    int noTileCull = 0;          // per-pixel root-box cull only
    int noLeanPrimary = 0;       // k_gen_primary stores the full ray state of surviving rays
    int leafMin = 0;             // k_trace2: lanes parked on a leaf before the node phase is left (0: 16 for batches of >= 4 samples, 12 below)
    int grabUnitLog2 = 10, grabFixed = 0;   // work-list hand-out: run length of a slice, entries reserved per atomic (0: what the refill needs)
    int ldsPad = 0;              // bytes of LDS added per workgroup of k_trace2 (caps the resident waves)
    int traceWaves = 0;          // one-wave workgroups per CU in the persistent grid (0: what LDS allows, at most 32)
    int gridHint = 2;            // bounce launches: grid = gridHint x the queue length the same bounce had in the previous batch (0: full grid)
    int deferLast = 1;           // the last bounce's continuation (state, queue) on demand instead of every frame, where only radiance of it is visible (kernels_shade.hpp k_shade_last)
    int graphProbe = 0;          // developer build only: capture the next batch into a hipGraph and time this many replays (tools/graph_probe.py)
    int gridMidWaves = 20;       // launches below GRID_MID_RAYS rays: one-wave workgroups per CU (0: off) — see small_launch_grid
    int gridRaysX4 = 6;          // small launches: quarter-rays per lane the persistent grid is sized for (from the previous batch's counts; 0: gridHint's rule alone)
    int nodeLayout = 0;          // derived node order (node_layout.hpp): 0 = reference order (default: the derived orders raise the L2 hit rate, not the speed — profiles/r03_layout_order_pmc.json), 1 = line couples depth-first, 2 = line couples in treelets
    int treeletDepth = 3;
    int traceOrder = 0;          // bounce launches handed out in spatial order (kernels_queue.hpp k_order_*): 0 = queue order (default: L2 hit rate 0.62 -> 0.88, launch -7 %, the permutation costs more), 1 = batches of >= 4 samples, 2 = always
    int traceVariant = 0;        // IDKPT_DEVELOPER builds only: instrumented / probe instantiations of k_trace2
    int bvhTiming = 0, bvhSmall = 32;   // idkptBuildBlasCore: phase times on stderr; subtrees of at most this many fragments are finished by one thread
    int advMin = 0;              // k_trace2 MODE 1-4: instance entries / TLAS steps are taken by at least this many lanes together (or when no other lane has work); 0 = 8 for batches of >= 4 samples, 1 below (measured, 3-BLAS soup-1M: 1 / 8 / 16 / 24 / 32 = 3 365 / 3 408 / 3 226 / 2 925 / 2 568 Mray/s through the instance loop, 2 671 / 2 771 / 2 671 / 2 466 / 2 180 through the TLAS; one frame at a time 8 costs 1-6 %)
    int poolMin = 12;            // pooled leaf phase: pairs a wave must have parked (below: every lane walks its own triangles as before; 0-20 measure the same, 32+ lose the gain)
    int leafPool = -1;           // k_trace2<.., DBG = 16> (MODE 0): the leaf phase tests the wave's pooled (ray, triangle) pairs with all lanes in one round trip (kernels_trace.hpp).
                                 // Mask of launch kinds: 1 = primary launches, 2 = the first bounce, 4 = later bounces.  -1 (default) = by measurement (profiles/r04_leaf_pool.md): later
                                 // bounces always lose (few pairs per phase: -4 %), the first bounce gains 3-5 % where most pixels traverse the scene and loses 1 % on sparse views,
                                 // the primary launch gains 4 % on sparse views and nothing elsewhere -> 1 on sparse views (fewer than half of the pixels enter the traversal), 3 otherwise
    int instanceRecords = 0;     // (default 0: measured slower as a whole — the records cost the producers more than they save the traversal, profiles/r04_multi_blas.md) scenes of 2..MAX_REC_INSTANCES instances (and USE_TLAS scenes of up to that many): one trace-ready record per (ray, instance), written by the producers (MODE 3 / 4 of k_trace2); 0: the instance entry is computed inside the traversal kernel (MODE 1 / 2)
    int spec = 0;                // developer build only: k_trace2<.., DBG = 8> (speculative touch of both children and the stack top before the box tests): 0 = never (default: measured slower at every launch size, profiles/r04_small_launch_experiments.md), 1 = launches below SPEC_MAX_RAYS rays, 2 = every launch
    int splitScatter = 6;        // k_trace2s: log2 of the entries that stay together when the work list is handed out scattered (6 = list order)
    int queryScheduler = 1;      // idkptTraceRays (closest hit) through k_trace2's scheduler instead of the thread-per-ray kernel (kernels_query.hpp)
    int groupThreads = -1;       // multi-device contexts (idkpt_api.hpp group_flush): members' batches enqueued by one host thread each also where no exchange needs it (-1: from 4 members on)
    int park = 0;                // k_trace2p (kernels_trace_park.hpp): a lane may carry two parked leaves (the second found with a stale T, re-validated before it is tested).  Bit mask like leaf_pool: 1 primary launches, 2 first bounce, 4 later bounces
    int quad = 0;                // k_trace2q (kernels_trace_quad.hpp): two binary levels per round trip on a derived 192-B record.  0 off, 1 small launches (want_quad), 2 wherever it applies
    int fused = 1;               // k_trace_fused (kernels_trace_fused.hpp): FirstHit + shading + the last NHit's traversal in one persistent launch at RayDepth 2.  0 off, 1 small launches on sparse views (want_fused), 2 wherever it is exact
    int fusedShadeMin = 16;      // ... lanes that wait for the shading phase before it runs (or as many as are still tracing)
    int splitPeek = 64;          // k_trace2s: iterations between two looks at the work-list heads (a wave with < 32 idle lanes never refills, so it has to ask whether the list is empty).  Measured: 64 -> 16 -> 8 -> 2 = 2 147 -> 1 750 -> 1 743 -> 1 697 Mray/s on the headline view one frame at a time: splitting EARLY multiplies pieces (and their bookkeeping) while most lanes still have rays of their own; it pays in the real tail only
    int splitDonor = 1;          // k_trace2s: 1 = only rays that have not hit anything yet donate subtrees, 0 = every busy lane does
    int split = 1;               // k_trace2s (long rays split across the idle lanes of their wave once the work list is empty): 0 = never, 1 = small launches of sparse views (default, want_split), 2 = every launch, 3 = every launch + every split ray traced again (test hook for the re-trace path)
    int wide = 0;                // k_trace_wide (kernels_wide.hpp): closest-hit launches of one-BLAS scenes walk the derived 4-wide nodes; rays it cannot vouch for are re-traced by k_trace2.
                                 // 0 (default) = k_trace2 only.  Measured in round 5 (profiles/r05_wide_nodes.md): 0.57x the dependent round trips, 0.61x the vector-memory requests, half the
                                 // memory-wait cycles — and 1.34x the VALU instructions, with the SIMDs' VALU issue already 80 % busy under k_trace2: 0.93-1.00x with 32 samples in flight,
                                 // 0.5-0.8x one frame at a time (the re-trace launch has its own latency floor).  Bit-identical results either way (tests/test_gpu_wide.py).
    int wideCap = 0;             // ... rows of its per-lane stack (0: 24; a ray that needs more is re-traced by k_trace2)
    int wideCount = 0;           // ... count node visits / leaf records / triangle tests (idkpt_stats.Wide*)
    int bvhStackOptHost = 0;            // idkptBuildBlas: take OptimizeStackSize's decisions from the reference's own walk on a host copy (the fallback path, forced: for its test)
};

// Scene versions (idkptSetSceneVersions): the buffers the render kernels read of the geometry an animated frame rewrites.  Each may hold several states
// ("slots" of one arena); a queued sample remembers the slots that were current when it was queued (PendingSample::vs), so frames with different geometry can
// be traced by one batch while later updates already write other slots (ver_writable).
enum { VB_NODES = 0, VB_TNODES, VB_TRIVERTS, VB_VERTICES, VB_TLAS, VB_XFORMS, VB_COUNT };
struct PendingSample { uint32_t accum; int slot; float cam[36]; uint8_t vs[VB_COUNT]; };   // cam = invProj[16] invView[16] viewPos[3] pad; vs = scene-version slots

struct dev_ctx {
    int device = 0;
    hipStream_t stream = nullptr; bool ownStream = true;
    std::string lastError;
    idkpt_error_fn errFn = nullptr; void* errUser = nullptr;   // idkptSetErrorCallback
    int numCUs = 256;
    // config
    idkpt_settings st;          // effective settings
    idkpt_settings stCaller;    // the struct the host passed last (idkptSetSettings compares against this one)
    int W = 0, H = 0, rowMod = 1, rowRem = 0, rows = 0, rowBandLog2 = 0;   // rows dealt in bands of 2^rowBandLog2 rows (idkptSetRowBands)
    float invProj[16], invView[16], viewPos[3];
    // frame ring (idkptSetFrameRing): ringSize result-image sets; every queued sample remembers its slot, its camera and its
    // AccumulatedSamples index, so several frames (different cameras) can be in flight in one batch
    int ringSize = 1, curSlot = 0; bool ringStarted = false; std::vector<uint32_t> accum = std::vector<uint32_t>(1, 0u);
    bool counters = false, timing = false, capturePrimary = false;
    DevOptions opt;
    uint32_t seqFirst = 0, seqStride = 1;                         // idkptSetSampleSequence
    // scene
    bool haveScene = false, frameOk = false;
    DevBuf nodes, tnodes, nodeSlot, tris, triVerts, descs, instances, tlas, parents, leaves, positions, prevPositions, vertices, meshes, materials, xforms, lights, sky, texDescs, unskinned, joints, levelNodes, tlasScratch, queryIn, queryOut, queryRec, queryList, quads, tileClass, gbases;   // (+ camTab below)
    std::vector<DevBuf> texData; std::vector<std::pair<int, int>> texDims;
    std::vector<GpuBlasDesc> hDescs;
    std::vector<std::vector<uint32_t>> levelOffsets; // per BLAS: offsets into levelNodes (level l occupies [off[l], off[l+1]))
    std::vector<uint32_t> levelBase;                 // per BLAS base into levelNodes
    std::vector<char> refitCoversAll;                // per BLAS: leaves + internal nodes of the refit schedule are every node but node 0 (a refit into a fresh slot needs no copy of the old nodes)
    int nodeCount = 0, triCount = 0, instanceCount = 0, tlasCount = 0, vertexCount = 0, meshCount = 0, materialCount = 0, xformCount = 0, lightCount = 0, skySize = 0, textureCount = 0, unskinnedCount = 0;
    int sceneStack = 1;
    int hInst0Blas = 0;                              // BlasId of instance 0 (MODE 0 traverses that BLAS)
    bool quadValid = false;                          // `quads` (kernels_trace_quad.hpp) matches the current nodes
    // wide nodes (kernels_wide.hpp): derived per BLAS from nodes + triVerts.  wideTopoValid: the children lists match the node topology; wideFillValid: boxes / leaf records match the current boxes and positions
    DevBuf wnodes, wleaf, wids, wpair, wcounts, wtotals; std::vector<uint32_t> wNodeOff, wLeafOff; bool wideTopoValid = false, wideFillValid = false;
    bool layoutActive = false;                       // tnodes holds the derived order (else the traversal reads `nodes`)
    // scene versions: slot count a versioned buffer may grow to, per buffer the bytes of one state / the slot pitch / the slots its arena holds / the current slot
    int verSlots = 1; size_t vbytes[VB_COUNT] = {0}, vstride[VB_COUNT] = {0}; int valloc[VB_COUNT] = {1, 1, 1, 1, 1, 1}, vcur[VB_COUNT] = {0};
    uint64_t lastMask[VB_COUNT] = {0};               // slots the last launched batch reads (a deferred last bounce still does: finish_deferred)
    uint8_t lastSlots[VB_COUNT] = {0}; bool lastMulti = false;   // ... the one set of slots of a single-version batch, or "per sample: verTab"
    DevBuf verTab; uint32_t* hVerTab = nullptr; hipEvent_t evVer[2] = {nullptr, nullptr}; int verHalf = 0;   // per-sample version table of the batch being launched (pinned, double-buffered staging)
    char* hStage = nullptr; hipEvent_t evStage[4] = {nullptr, nullptr, nullptr, nullptr}; int stageNext = 0;   // pinned ring for small host -> device updates (joint matrices, transforms): no stream synchronisation per call
    int (*groupFlushAll)(void* user) = nullptr;      // member of a multi-device context: launches what ALL members have queued (a member never flushes on its own)
    // wavefront state
    DevBuf trRec, contFlag, blockSums, rayO, rayT, rayR, aovA, aovN, hit, hitCost, primHit, queue[2], keys[2], keysTmp, sortKeys, sortVals, ordKeys[2], ordVals[2], ordIdx, contMask, waveCounts, counts, work, sortHist, counters64;
    DevBuf img[3];
    DevBuf camTab;                                       // per-sample cameras of the batch being launched (ring mode)
    int rowLimit = 0x7fffffff;                           // idkptSetRowRange: at most this many local rows
    idkpt_bounce_exchange_fn exchangeFn = nullptr; void* exchangeUser = nullptr;   // exact multi-GPU deep paths (idkptSetBounceExchange)
    idkpt_band_exchange_fn bandExchangeFn = nullptr; void* bandExchangeUser = nullptr; DevBuf bandTab;
    idkpt_band_exchange_device_fn bandExchangeDevFn = nullptr; void* bandExchangeDevUser = nullptr;   // ... enqueued on the stream, no host synchronisation (idkptSetBandExchangeDevice)   // ... for interleaved rows / bands (idkptSetBandExchange)
    // stats
    idkpt_stats stats;
    uint32_t* hCounts = nullptr; uint32_t* dCountsMirror = nullptr;   // host-mapped mirror of the queue lengths (written by k_scan_blocks, read by the host after a sync)
    bool sceneNested = false;       // every child box of every BLAS lies inside its parent's box (what k_trace2s's exactness argument needs; refits keep it)
    bool sceneNoEmission = false;   // no material / mesh of the uploaded scene emits and every texel is finite: a hit of the last bounce cannot change the radiance (k_shade_last)
    struct { bool valid = false, allHits = false; int j = 0, side = 0, B = 0; uint32_t total = 0, Npad = 0; } defer;   // the last bounce of the last batch still owes its continuation (finish_deferred)
    DevBuf radSave, deferCount;
    bool countersDirty = true;   // the batch counters were not reset by the last k_final_draw (first batch, or a batch that failed half way)
    uint32_t* hOverflow = nullptr; uint32_t* dOverflow = nullptr;   // host-mapped word the kernels set when a traversal-stack push is dropped (checked after every sync)
    int tlasNeed = 1;            // rows the TLAS walk needs (validated for host-built TLAS nodes; min(instances, TLAS_STACK_SIZE) for a device build)
    hipEvent_t evFrame[2] = {nullptr, nullptr};
    // trace-kernel timing (idkptEnableTiming): one event pair per trace launch, resolved lazily in idkptGetStats
    std::vector<hipEvent_t> evPool; size_t evUsed = 0;
    double traceMsAcc = 0.0; uint64_t traceLaunchesAcc = 0;
    int lastQueueSide = 0; int lastQueueCountSlot = 0; bool lastFast = false, lastNeedsRegen = false; int lastBatch = 1; Frame lastFrame;
    int maxBatch = 1; uint32_t Npad = 0; std::vector<PendingSample> pending; DevBuf bases, qwork; uint32_t* hBases = nullptr; uint32_t* dBasesMirror = nullptr;
    float* hCams = nullptr; hipEvent_t evCams[2] = {nullptr, nullptr}; int camHalf = 0;   // pinned, double-buffered staging of the per-sample cameras (frame ring)    // member of a multi-device context (idkpt_api.hpp): samples are only queued (the group launches all members together), per-bounce events tell
    // the members that own later rows when this member's alive counts of a bounce are final, and the group supplies the slot bases
    bool grouped = false, inGroupFlush = false; int groupIndex = 0;
    hipEvent_t* evBounce = nullptr;                              // [MAX_DEPTH_SLOTS]; evBounce[j] = bases[j] (alive counts entering bounce j) written
    int (*groupExchange)(void* user, dev_ctx* member, int bounce, int samples, const uint32_t** outBases) = nullptr; void* groupUser = nullptr;
    struct BuilderScratch* bscratch = nullptr;                    // device buffers of idkptBuildBlas / idkptBuildBlasCore, kept between calls (grow only)
    struct PeerPolicy* peer = nullptr;                           // multi-device contexts: how device-to-device copies are made (member_copy)
};

static hipEvent_t next_event(dev_ctx* ctx)
{
    if (ctx->evUsed == ctx->evPool.size()) { hipEvent_t e = nullptr; if (hipEventCreate(&e) != hipSuccess) return nullptr; ctx->evPool.push_back(e); }
    return ctx->evPool[ctx->evUsed++];
}
// folds all recorded (start, stop) pairs into the accumulators; requires the stream to be idle
static void resolve_trace_events(dev_ctx* ctx)
{
    for (size_t i = 0; i + 1 < ctx->evUsed; i += 2) { float ms = 0.0f; if (hipEventElapsedTime(&ms, ctx->evPool[i], ctx->evPool[i + 1]) == hipSuccess) { ctx->traceMsAcc += ms; ctx->traceLaunchesAcc++; } }
    ctx->evUsed = 0;
}
#define TRACE_T0() do { if (ctx->timing) { hipEvent_t _e = next_event(ctx); if (_e) (void)hipEventRecord(_e, st); } } while (0)
#define TRACE_T1() do { if (ctx->timing) { hipEvent_t _e = next_event(ctx); if (_e) (void)hipEventRecord(_e, st); } } while (0)

// every failing entry point ends here: the message is kept for idkptGetLastError and handed to the host's error callback (idkptSetErrorCallback; oidnSetDeviceErrorFunction's
// pattern, Source/OIDN/OIDN.cs:108-109) on the thread that detected the error
static int fail(dev_ctx* c, int code, const std::string& msg) { if (c) { c->lastError = msg; if (c->errFn) c->errFn(c->errUser, (int32_t)code, c->lastError.c_str()); } return code; }
// (a failed runtime call leaves its code in the thread's last-error slot: reset it, or the next hipGetLastError() check would report it again)
#define HIPC(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { (void)hipGetLastError(); return fail(ctx, IDKPT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); } } while (0)
#define REQUIRE(cond, msg) do { if (!(cond)) return fail(ctx, IDKPT_ERR_INVALID_ARGUMENT, msg); } while (0)

// rows y of the image with (y >> bandLog2) % mod == rem (bandLog2 = 0: y % mod == rem; mod = 1: the rows from rem on)
static int local_rows(int H, int mod, int rem, int bandLog2 = 0)
{
    if (bandLog2 == 0) { int n = 0; for (int y = rem; y < H; y += mod) n++; return n; }
    int n = 0; const int band = 1 << bandLog2;
    for (int b = rem; (b << bandLog2) < H; b += mod) n += std::min(band, H - (b << bandLog2));
    return n;
}

// ---- wide nodes (kernels_wide.hpp): which launches use them, and their derivation on the device ----------------------------------------------------------------
// One-BLAS scenes, closest hit, the reference's counters not asked for, one scene version: everything else keeps k_trace2.
static bool wide_wanted(const dev_ctx* ctx) { return ctx->opt.wide != 0 && ctx->instanceCount == 1 && !ctx->st.UseTlas && ctx->verSlots == 1 && !ctx->counters && (ctx->opt.traceVariant == 0 || ctx->opt.traceVariant == 100 || ctx->opt.traceVariant == 213); }
static int wide_stack_rows(const dev_ctx* ctx) { return std::min(96, std::max(4, ctx->opt.wideCap > 0 ? ctx->opt.wideCap : 24)); }
// (re-)derives what is stale: the children lists after an upload / a node patch (k_wide_topo, one workgroup per BLAS), box bytes and leaf records after anything that
// moved boxes or positions (k_wide_fill).  Stream-ordered in front of the batch that is about to be launched; with one scene version every update launches the queued samples first.
static char* vb_ptr(dev_ctx* ctx, int b, int slot);
static int wide_prepare(dev_ctx* ctx)
{
    if (ctx->wideTopoValid && ctx->wideFillValid) return IDKPT_OK;
    if (!ctx->wtotals.p) { HIPC(ctx->wtotals.ensure(64)); HIPC(hipMemsetAsync(ctx->wtotals.p, 0, 64, ctx->stream)); }
    const size_t nb = ctx->hDescs.size();
    hipStream_t st = ctx->stream;
    const float4* nodes = (const float4*)vb_ptr(ctx, VB_NODES, ctx->vcur[VB_NODES]);
    const float4* triVerts = (const float4*)vb_ptr(ctx, VB_TRIVERTS, ctx->vcur[VB_TRIVERTS]);
    if (!ctx->wideTopoValid) {
        ctx->wNodeOff.assign(nb + 1, 0u); ctx->wLeafOff.assign(nb + 1, 0u);
        for (size_t b = 0; b < nb; b++) {
            const GpuBlasDesc& d = ctx->hDescs[b];
            const uint32_t pairs = (uint32_t)std::max(0, d.NodeCount) / 2u + 1u, leavesMax = pairs + 1u;                       // a wide node stands for at least one pair; a tree of L leaves has L - 1 internal nodes
            ctx->wNodeOff[b + 1] = ctx->wNodeOff[b] + pairs;
            ctx->wLeafOff[b + 1] = ctx->wLeafOff[b] + 5u * leavesMax + 3u * (uint32_t)std::max(0, d.TriangleCount) + 4u;   // 2 + 3 per leaf-range triangle (a leaf pair may share one triangle)
        }
        HIPC(ctx->wnodes.ensure((size_t)ctx->wNodeOff[nb] * 64 + 64)); HIPC(ctx->wids.ensure((size_t)ctx->wNodeOff[nb] * 16 + 16)); HIPC(ctx->wpair.ensure((size_t)ctx->wNodeOff[nb] * 4 + 16));
        HIPC(ctx->wleaf.ensure((size_t)ctx->wLeafOff[nb] * 16 + 80)); HIPC(ctx->wcounts.ensure(nb * 8 + 8));
        for (size_t b = 0; b < nb; b++) {
            const GpuBlasDesc& d = ctx->hDescs[b];
            hipLaunchKernelGGL(k_wide_topo, dim3(1), dim3(WIDE_TOPO_THREADS), 0, st, nodes + 2 * (size_t)d.NodeOffset, (uint32_t)d.NodeCount, ctx->wpair.as<uint32_t>() + ctx->wNodeOff[b],
                               ctx->wids.as<uint4>() + ctx->wNodeOff[b], ctx->wnodes.as<uint4>() + 4 * (size_t)ctx->wNodeOff[b], ctx->wcounts.as<uint32_t>() + 2 * b);
        }
        HIPC(hipGetLastError());
        ctx->wideTopoValid = true; ctx->wideFillValid = false;
    }
    for (size_t b = 0; b < nb; b++) {
        const GpuBlasDesc& d = ctx->hDescs[b];
        const uint32_t pairs = ctx->wNodeOff[b + 1] - ctx->wNodeOff[b];
        hipLaunchKernelGGL(k_wide_fill, dim3((pairs + 255) / 256), dim3(256), 0, st, nodes + 2 * (size_t)d.NodeOffset, triVerts + 3 * (size_t)d.TriangleOffset, (const uint4*)(ctx->wids.as<uint4>() + ctx->wNodeOff[b]),
                           ctx->wnodes.as<uint4>() + 4 * (size_t)ctx->wNodeOff[b], ctx->wleaf.as<float4>() + ctx->wLeafOff[b], (const uint32_t*)(ctx->wcounts.as<uint32_t>() + 2 * b));
    }
    HIPC(hipGetLastError());
    ctx->wideFillValid = true;
    return IDKPT_OK;
}

template <bool PRIMARY>
static void launch_trace2(dev_ctx* ctx, uint32_t grid, size_t lds, hipStream_t st, const DScene& s, const Frame& f, const RayBufs& rays, const TraceBufs& tr, const HitBufs& hits,
                          const uint32_t* list, const uint32_t* cnt, uint32_t* work, uint64_t* counters, bool split = false, bool spec = false, int bounce = 0, bool anyHit = false, bool quad = false)
{
    if (anyHit) {   // idkptTraceRays with IDKPT_TRACE_ANY_HIT (kernels_query.hpp): TraceRayAny's walk on the same scheduler
#define T2A(M) hipLaunchKernelGGL((k_trace2<true, false, 32, 1, false, 24, M, 0, false, true>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters)
        if (f.useTlas) T2A(2); else if (ctx->instanceCount > 1) T2A(1); else T2A(0);
#undef T2A
        return;
    }
    if (ctx->wideFillValid && ctx->wideTopoValid && wide_wanted(ctx) && !s.ver && !f.useTlas && !f.queryMode && !f.hitsByRid) {
        // wide-node walk (kernels_wide.hpp), then — on the launch's own list of the rays it does not vouch for, almost always empty — the exact kernel
        const int b0 = ctx->hInst0Blas;
        WideBufs wb;
        wb.nodes = (const uint4*)(ctx->wnodes.as<uint4>() + 4 * (size_t)ctx->wNodeOff[b0]); wb.leaves = (const float4*)(ctx->wleaf.as<float4>() + ctx->wLeafOff[b0]);
        wb.flagCount = work + 128; wb.flagA = ctx->sortKeys.as<uint32_t>(); wb.flagB = ctx->sortVals.as<uint32_t>(); wb.totals = ctx->wtotals.as<unsigned long long>(); wb.cap = wide_stack_rows(ctx);
        const size_t ldsW = (size_t)(wb.cap + 2) * WAVE * 4 + (size_t)std::max(0, ctx->opt.ldsPad);     // + the dummy and the spare row
#ifdef IDKPT_DEVELOPER
        if (ctx->opt.traceVariant == 213) hipLaunchKernelGGL((k_trace_wide<PRIMARY, false, 32, true>), dim3(grid), dim3(WAVE), ldsW, st, s, f, rays, tr, hits, list, cnt, work, wb, counters);   // s_memtime-instrumented
        else
#endif
        if (ctx->opt.wideCount) hipLaunchKernelGGL((k_trace_wide<PRIMARY, true>), dim3(grid), dim3(WAVE), ldsW, st, s, f, rays, tr, hits, list, cnt, work, wb, counters);
        else hipLaunchKernelGGL((k_trace_wide<PRIMARY, false>), dim3(grid), dim3(WAVE), ldsW, st, s, f, rays, tr, hits, list, cnt, work, wb, counters);
        TraceBufs trf = tr; trf.order = nullptr; trf.orderIdx = nullptr;
        if (!PRIMARY) { trf.order = wb.flagA; trf.orderIdx = wb.flagB; }        // position -> queue slot and ray id (the hit is stored at the slot, as always)
        Frame ff = f; ff.gridRaysX4 = 6u; ff.gridMid = 0u;                       // (the device sizes the launch from its actual count: k_trace2's own rule)
        hipLaunchKernelGGL((k_trace2<PRIMARY, false>), dim3(std::min<uint32_t>(grid, 2048u)), dim3(WAVE), lds, st, s, ff, rays, trf, hits, (const uint32_t*)wb.flagA, (const uint32_t*)wb.flagCount, work + 64, counters);
        return;
    }
#ifdef IDKPT_DEVELOPER
    if (spec && !split && !s.ver && !f.useTlas && ctx->instanceCount == 1 && (ctx->opt.traceVariant == 0 || ctx->opt.traceVariant == 100)) {   // the candidates of the next step are requested before this step's box tests (kernels_trace.hpp, DBG 8)
        if (ctx->counters) hipLaunchKernelGGL((k_trace2<PRIMARY, true, 32, 1, false, 24, 0, 8>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters);
        else hipLaunchKernelGGL((k_trace2<PRIMARY, false, 32, 1, false, 24, 0, 8>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters);
        return;
    }
#endif
    (void)spec;
    if (quad && !s.ver && !f.useTlas && ctx->instanceCount == 1 && !ctx->counters && ctx->verSlots == 1 && !f.queryMode && (ctx->opt.traceVariant == 0 || ctx->opt.traceVariant == 100)) {
        // two binary levels per round trip (kernels_trace_quad.hpp); the 192-B records are (re-)derived when the nodes have changed since
        bool ok = true;
        if (!ctx->quadValid) {
            ok = ctx->quads.ensure((size_t)ctx->nodeCount * 96 + 256) == hipSuccess;
            if (ok) {
                const float4* tn = (const float4*)s.tnodes;
                for (const GpuBlasDesc& d : ctx->hDescs)
                    if (d.NodeCount >= 4) hipLaunchKernelGGL(k_derive_quads, dim3(((uint32_t)d.NodeCount / 2 + 256) / 256), dim3(256), 0, st, tn + 2 * (size_t)d.NodeOffset, ctx->quads.as<float4>() + 6 * (size_t)d.NodeOffset, (uint32_t)d.NodeCount);
                ctx->quadValid = true;
            }
        }
        if (ok) {
            const GpuBlasDesc& d0 = ctx->hDescs[ctx->hInst0Blas];
            hipLaunchKernelGGL((k_trace2q<PRIMARY>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, (const float4*)(ctx->quads.as<float4>() + 6 * (size_t)d0.NodeOffset));
            return;
        }
    }
    if (split && !s.ver && !f.useTlas && ctx->instanceCount == 1 && !ctx->counters && ctx->sceneNested && (ctx->opt.traceVariant == 0 || ctx->opt.traceVariant == 100)) {   // small launch: long rays are split across idle lanes (kernels_trace_split.hpp)
        hipLaunchKernelGGL((k_trace2s<PRIMARY>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work);
        return;
    }
    if (((ctx->opt.park >> std::min(bounce, 2)) & 1) && !s.ver && !f.useTlas && ctx->instanceCount == 1 && !ctx->counters && ctx->sceneNested && !f.queryMode && (ctx->opt.traceVariant == 0 || ctx->opt.traceVariant == 100)) {
        hipLaunchKernelGGL((k_trace2p<PRIMARY>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work);   // two parked leaves per lane (kernels_trace_park.hpp)
        return;
    }
    // pooled leaf phase (kernels_trace.hpp, DBG 16): per launch kind — option leaf_pool, a mask: 1 = primary launches, 2 = the first bounce, 4 = later bounces; -1 = automatic
    int poolMask = ctx->opt.leafPool;
    if (poolMask < 0) {   // automatic: by view class, known from the previous batch of the same shape (unknown: the dense-view choice)
        const uint64_t pixels = (uint64_t)ctx->W * ctx->rows * (uint64_t)std::max(1, f.batch);
        const bool sparse = ctx->lastFast && ctx->lastBatch == f.batch && (uint64_t)ctx->hCounts[MAX_DEPTH_SLOTS - 1] * 2u < pixels;
        poolMask = sparse ? 1 : 3;
    }
    const bool pool = ((poolMask >> std::min(bounce, 2)) & 1) && (ctx->opt.traceVariant == 0 || ctx->opt.traceVariant == 100);
#define T2X(C, M, D, V) hipLaunchKernelGGL((k_trace2<PRIMARY, C, 32, 1, false, 24, M, D, V>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters)
#define T2P(C, M, V) do { if (pool) T2X(C, M, 16, V); else T2X(C, M, 0, V); } while (0)
#define T2C(M, V) do { if (ctx->counters) T2P(true, M, V); else T2P(false, M, V); } while (0)
    const bool rec = f.recPerRay > 1;      // per-instance trace-ready records (scenes of up to MAX_REC_INSTANCES instances, option instance_records): MODE 3 / 4 instead of 1 / 2 (never pooled)
    if (s.ver) {                    // scene versions: the samples of this batch see different states of the geometry (VER instantiations, kernels_trace.hpp)
// the instance-loop / TLAS kernels: never pooled (they gain nothing from it), and idle lanes are refilled from 16 on instead of 32 — their rays live two to three BLAS walks,
// so a refill is rarer per step than in MODE 0 and lanes are what these modes lack (profiles/r04_multi_blas.md: 8 / 12 / 16 / 24 / 32 / 48 = 3 524 / 3 572 / 3 592-3 654 / 3 616 /
// 3 446-3 474 / 2 684 Mray/s through the instance loop, TLAS alike; MODE 0 keeps 32: 24 measured -4.5 % there in round 1)
#define T2M(C, M, V) hipLaunchKernelGGL((k_trace2<PRIMARY, C, 16, 1, false, 24, M, 0, V>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters)
#define T2N(M, V) do { if (ctx->counters) T2M(true, M, V); else T2M(false, M, V); } while (0)
        if (f.useTlas) { if (rec) T2N(4, true); else T2N(2, true); }
        else if (ctx->instanceCount > 1) { if (rec) T2N(3, true); else T2N(1, true); }
        else T2C(0, true);
        return;
    }
#ifdef IDKPT_DEVELOPER
    // developer probe (trace_variant 24 / 48 / 16): the refill threshold of the instance-loop / TLAS kernels (their rays live two to three BLAS walks: fewer refills per step than MODE 0's)
    if (!rec && !ctx->counters && (f.useTlas || ctx->instanceCount > 1) && (ctx->opt.traceVariant == 24 || ctx->opt.traceVariant == 48 || ctx->opt.traceVariant == 16 || ctx->opt.traceVariant == 8 || ctx->opt.traceVariant == 12)) {
#define T2R(R, M) hipLaunchKernelGGL((k_trace2<PRIMARY, false, R, 1, false, 24, M, 0, false>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters)
        const int m = f.useTlas ? 2 : 1;
        if (ctx->opt.traceVariant == 24) { if (m == 2) T2R(24, 2); else T2R(24, 1); }
        else if (ctx->opt.traceVariant == 48) { if (m == 2) T2R(48, 2); else T2R(48, 1); }
        else if (ctx->opt.traceVariant == 16) { if (m == 2) T2R(16, 2); else T2R(16, 1); }
        else if (ctx->opt.traceVariant == 12) { if (m == 2) T2R(12, 2); else T2R(12, 1); }
        else { if (m == 2) T2R(8, 2); else T2R(8, 1); }
#undef T2R
        return;
    }
#endif
    if (f.useTlas) {                // TLAS walk inside the kernel (MODE 4: the leaves' instance entries come from per-instance records, MODE 2: computed in the kernel)
        if (rec) T2N(4, false); else T2N(2, false);
        return;
    }
    if (ctx->instanceCount > 1) {   // instance loop inside the kernel (MODE 3: per-instance records, MODE 1: computed in the kernel)
        if (rec) T2N(3, false); else T2N(1, false);
        return;
    }
    if (pool) { if (ctx->counters) T2X(true, 0, 16, false); else T2X(false, 0, 16, false); return; }
#undef T2N
#undef T2M
#undef T2C
#undef T2P
#undef T2X
    if (ctx->counters) { hipLaunchKernelGGL((k_trace2<PRIMARY, true>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters); return; }
#ifdef IDKPT_DEVELOPER
    switch (ctx->opt.traceVariant) {   // developer builds (libidkpt_dev.so, option "trace_variant"): s_memtime-instrumented and probe instantiations; results are bit-identical
        case 107: hipLaunchKernelGGL((k_trace2<PRIMARY, false, 16, 1, true, 65>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters); return;   // instrumented, old policy
        case 113: hipLaunchKernelGGL((k_trace2<PRIMARY, false, 32, 1, true, 24>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters); return;  // instrumented, default policy
        case 121: hipLaunchKernelGGL((k_trace2<PRIMARY, false, 32, 1, false, 24, 0, 32>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters); return;  // the leaf's first triangle requested in the node step
        case 122: hipLaunchKernelGGL((k_trace2<PRIMARY, false, 32, 1, true, 24, 0, 32>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters); return;  // ... instrumented
        case 116: hipLaunchKernelGGL((k_trace2<PRIMARY, false, 32, 1, true, 24, 0, 16>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters); return;  // instrumented, pooled leaf phase
#define T2V(R, L) hipLaunchKernelGGL((k_trace2<PRIMARY, false, R, 1, false, L>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters)
        case 901: T2V(32, 20); return; case 902: T2V(40, 16); return; case 903: T2V(24, 24); return; case 904: T2V(16, 24); return;   // (903 / 904: MODE 0's refill threshold re-measured in round 4)
        case 961: hipLaunchKernelGGL((k_trace2<PRIMARY, false, 32, 7, false, 24>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters); return;   // occupancy probes: 7 / 8 waves per SIMD forced (launch bounds)
        case 962: hipLaunchKernelGGL((k_trace2<PRIMARY, false, 32, 8, false, 24>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters); return;
#define T2D(D) hipLaunchKernelGGL((k_trace2<PRIMARY, false, 32, 1, false, 24, 0, D>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters)
        case 951: T2D(1); return; case 952: T2D(2); return; case 953: T2D(3); return; case 954: T2D(4); return;   // bottleneck probes: +16 VALU / +16 SALU / +48 SALU / +48 VALU instructions per node step
#undef T2D
#undef T2V
        default: break;
    }
#endif
    hipLaunchKernelGGL((k_trace2<PRIMARY, false>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters);
}

static int alloc_frame_impl(dev_ctx* ctx)
{
    const size_t N = (size_t)ctx->W * ctx->rows;
    ctx->Npad = (uint32_t)((N + 63) / 64 * 64);
    const size_t cap = (size_t)ctx->maxBatch * ctx->Npad;   // ray ids of one batch
    ctx->pending.clear();
    ctx->lastFast = false; ctx->lastNeedsRegen = false; ctx->lastBatch = 1;   // nothing rendered into the new buffers yet
    HIPC(ctx->rayO.ensure(cap * 16)); HIPC(ctx->rayT.ensure(cap * 16)); HIPC(ctx->rayR.ensure(cap * 16));
    HIPC(ctx->aovA.ensure(cap * 16)); HIPC(ctx->aovN.ensure(cap * 16));
    HIPC(ctx->trRec.ensure(cap * 64)); HIPC(ctx->contFlag.ensure(cap));
    HIPC(ctx->blockSums.ensure(((cap + 63) / 64 + SCAN_WAVES_PER_BLOCK - 1) / SCAN_WAVES_PER_BLOCK * 4 + 16));
    HIPC(ctx->hit.ensure(cap * 32)); HIPC(ctx->hitCost.ensure(cap * 4));
    for (int i = 0; i < 2; i++) { HIPC(ctx->queue[i].ensure(cap * 4)); HIPC(ctx->keys[i].ensure(cap * 4)); }
    HIPC(ctx->keysTmp.ensure(cap * 4)); HIPC(ctx->sortKeys.ensure(cap * 4)); HIPC(ctx->sortVals.ensure(cap * 4));
    if (ctx->opt.traceOrder) { for (int i = 0; i < 2; i++) { HIPC(ctx->ordKeys[i].ensure(cap * 4)); HIPC(ctx->ordVals[i].ensure(cap * 4)); } HIPC(ctx->ordIdx.ensure(cap * 4)); }
    size_t nW = (cap + 63) / 64;
    HIPC(ctx->contMask.ensure(nW * 8)); HIPC(ctx->waveCounts.ensure(nW * 4));
    HIPC(ctx->counts.ensure(MAX_DEPTH_SLOTS * 4)); HIPC(ctx->work.ensure(WORK_WORDS * 4)); HIPC(ctx->counters64.ensure(128));
    HIPC(ctx->bases.ensure((size_t)MAX_DEPTH_SLOTS * (MAX_BATCH + 1) * 4));
    size_t nTiles = (cap + SORT_TILE - 1) / SORT_TILE;
    HIPC(ctx->sortHist.ensure((SORT_RADIX * nTiles + SORT_RADIX) * 4));
    for (int i = 0; i < 3; i++) { HIPC(ctx->img[i].ensure(N * 16 * ctx->ringSize)); HIPC(hipMemsetAsync(ctx->img[i].p, 0, N * 16 * ctx->ringSize, ctx->stream)); }   // slot s at offset s*N
    HIPC(hipMemsetAsync(ctx->counters64.p, 0, 128, ctx->stream));
    ctx->defer.valid = false;      // (callers complete a deferred last bounce before they get here; whatever is left refers to buffers that are gone)
    ctx->countersDirty = true;   // (the first batch resets its counters itself)
    HIPC(hipMemsetAsync(ctx->aovA.p, 0, cap * 16, ctx->stream)); HIPC(hipMemsetAsync(ctx->aovN.p, 0, cap * 16, ctx->stream));
    HIPC(hipMemsetAsync(ctx->contFlag.p, 0, cap, ctx->stream));   // per-batch values are written by k_gen_primary; the pad ids [N, Npad) must read 0
    ctx->accum.assign(ctx->ringSize, 0u); ctx->curSlot = 0; ctx->ringStarted = false;
    return IDKPT_OK;
}

// a failed allocation leaves the context without a usable frame (idkptRender refuses) until a later idkptSetSize / idkptSetMaxBatch succeeds
static int alloc_frame(dev_ctx* ctx) { int rc = alloc_frame_impl(ctx); ctx->frameOk = rc == IDKPT_OK; return rc; }

// maxBatch changed: the wavefront buffers grow, the accumulation images (and their contents) stay
static int alloc_frame_keep_images(dev_ctx* ctx)
{
    const size_t N = (size_t)ctx->W * ctx->rows;
    DevBuf saved[3];
    for (int i = 0; i < 3; i++) { saved[i] = ctx->img[i]; ctx->img[i] = DevBuf(); }
    int rc = alloc_frame(ctx);
    // The restore is ordered on the context's stream, behind the zero-fill alloc_frame_impl queued there: the stream is non-blocking, so
    // a null-stream copy would be unordered against that fill (the fill could land after the restore and wipe the accumulation).
    hipError_t e = hipSuccess;
    for (int i = 0; i < 3 && rc == IDKPT_OK && e == hipSuccess; i++)
        if (saved[i].p) e = hipMemcpyAsync(ctx->img[i].p, saved[i].p, std::min(saved[i].bytes, N * 16 * ctx->ringSize), hipMemcpyDeviceToDevice, ctx->stream);
    if (rc == IDKPT_OK && e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // the saved buffers are released right below
    for (int i = 0; i < 3; i++) saved[i].release();
    if (rc == IDKPT_OK && e != hipSuccess) { (void)hipGetLastError(); ctx->frameOk = false; return fail(ctx, IDKPT_ERR_HIP, std::string("alloc_frame_keep_images: ") + hipGetErrorString(e)); }
    return rc;
}

static int flush_batch(dev_ctx* ctx);
// After a stream synchronisation: did any traversal drop a stack push?  (Cannot happen for scenes that passed idkptUploadScene's
// validation with BlasStackSize >= the computed need; the flag is the safety net for buffers patched later with idkptUpdateBuffer
// and for device-built TLASes deeper than TLAS_STACK_SIZE.)  The results of the affected batch are invalid: report, never return them silently.
static int check_overflow(dev_ctx* ctx)
{
    if (!ctx->hOverflow || *(volatile uint32_t*)ctx->hOverflow == 0u) return IDKPT_OK;
    *(volatile uint32_t*)ctx->hOverflow = 0u;
    return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "traversal stack overflow: a BLAS/TLAS is deeper than the traversal stack (BlasStackSize / TLAS_STACK_SIZE); results since the last synchronisation are invalid");
}
#define SYNC_CHECKED() do { HIPC(hipStreamSynchronize(ctx->stream)); int _rc = check_overflow(ctx); if (_rc) return _rc; } while (0)
static int finish_deferred(dev_ctx* ctx);
// FLUSH: issue the samples still queued, and complete a deferred last bounce (readers of ray state / queues, and everything that changes what its kernels would read).
// FLUSH_KEEP: issue only (images, synchronisation, camera, statistics, batching knobs: nothing that looks at or invalidates the deferred part).
// (a member of a multi-device context never launches on its own: the group launches what ALL members have queued, idkpt_api.hpp group_flush_all)
static int flush_any(dev_ctx* ctx);
#define FLUSH() do { int _rc = flush_any(ctx); if (_rc) return _rc; _rc = finish_deferred(ctx); if (_rc) return _rc; } while (0)
#define FLUSH_KEEP() do { int _rc = flush_any(ctx); if (_rc) return _rc; } while (0)

// ---- scene versions -------------------------------------------------------------------------------------------------------------------------------
// A scene update (skinning, refit, TLAS rebuild, a patched transform) used to launch every queued sample first: the kernels of a batch read "the" scene, so
// an animated host rendered one frame at a time (a quarter of the batched rate).  With idkptSetSceneVersions(n > 1) the buffers those updates write — and the
// render kernels read — are arenas of up to n slots.  An update that would overwrite a slot which queued samples (or the deferred last bounce of the launched
// batch) still read moves the CURRENT state of that buffer to a free slot first (copying it unless the update rewrites all of it) and writes there; updates run
// eagerly in stream order, rendering stays deferred, and a batch whose samples saw different slots gets a per-sample table (DScene::ver, VER kernels).
// Everything is on the context's one stream, so kernels already launched are ordered before any later write: only unlaunched samples pin a slot.
// With one slot (the default) the first branch below never finds room and falls through to "launch what is queued": exactly the old behaviour.
static int flush_any(dev_ctx* ctx)
{
    if (ctx->pending.empty()) return IDKPT_OK;
    if (!(ctx->grouped && !ctx->inGroupFlush && ctx->groupFlushAll)) return flush_batch(ctx);
    // the group launches every member's batch and leaves some other member's device current: the caller (an update of THIS member: hipMalloc,
    // event creation, kernel launches follow) continues on its own device
    const int rc = ctx->groupFlushAll(ctx->groupUser);
    const hipError_t e = hipSetDevice(ctx->device);
    if (rc) return rc;
    if (e != hipSuccess) return fail(ctx, IDKPT_ERR_HIP, std::string("hipSetDevice after the group flush: ") + hipGetErrorString(e));
    return IDKPT_OK;
}
static DevBuf& vb_buf(dev_ctx* ctx, int b)
{
    switch (b) { case VB_NODES: return ctx->nodes; case VB_TNODES: return ctx->tnodes; case VB_TRIVERTS: return ctx->triVerts; case VB_VERTICES: return ctx->vertices; case VB_TLAS: return ctx->tlas; default: return ctx->xforms; }
}
static char* vb_ptr(dev_ctx* ctx, int b, int slot) { return (char*)vb_buf(ctx, b).p + (size_t)slot * ctx->vstride[b]; }
template <class T> static T* vb_cur(dev_ctx* ctx, int b) { return (T*)vb_ptr(ctx, b, ctx->vcur[b]); }
// after idkptUploadScene / a clone / a re-derived node order: one state per buffer, in slot 0 of whatever allocation the buffer has
static void ver_reset(dev_ctx* ctx)
{
    ctx->quadValid = false; ctx->wideTopoValid = false; ctx->wideFillValid = false;
    const size_t one[VB_COUNT] = {(size_t)ctx->nodeCount * 32, ctx->layoutActive ? (size_t)ctx->nodeCount * 32 : 0, (size_t)ctx->triCount * 48, (size_t)ctx->vertexCount * 16,
                                  (size_t)std::max(ctx->tlasCount, 2 * ctx->instanceCount - 1) * 32, (size_t)ctx->xformCount * sizeof(GpuMeshTransform)};
    for (int b = 0; b < VB_COUNT; b++) { ctx->vbytes[b] = one[b]; ctx->vstride[b] = (one[b] + 255) / 256 * 256; ctx->valloc[b] = 1; ctx->vcur[b] = 0; ctx->lastMask[b] = 0; ctx->lastSlots[b] = 0; }
    ctx->lastMulti = false;
}
// ... and every buffer really holds one whole slot: a scene uploaded without TLAS nodes (legal when !UseTlas) leaves `tlas` a 16-byte allocation while its slot
// is sized for the 2n - 1 nodes idkptBuildTlas / idkptBuildTlasOnDevice write in place later (ver_writable hands out slot 0 without looking at the allocation)
static int ver_reserve(dev_ctx* ctx)
{
    for (int b = 0; b < VB_COUNT; b++) {
        DevBuf& buf = vb_buf(ctx, b);
        if (ctx->vbytes[b] == 0 || buf.bytes >= ctx->vbytes[b]) continue;
        DevBuf nb; HIPC(nb.ensure(ctx->vstride[b]));
        if (buf.p && buf.bytes) HIPC(hipMemcpyAsync(nb.p, buf.p, std::min(buf.bytes, ctx->vbytes[b]), hipMemcpyDeviceToDevice, ctx->stream));
        HIPC(hipStreamSynchronize(ctx->stream));            // (the old allocation is released right below; once per upload, only for a buffer that was short)
        buf.release(); buf = nb;
    }
    return IDKPT_OK;
}
// the arena of buffer b gets room for every slot (first use of a second slot): only slot 0 is in use at that moment
static int ver_grow(dev_ctx* ctx, int b)
{
    if (ctx->valloc[b] >= ctx->verSlots) return IDKPT_OK;
    DevBuf& buf = vb_buf(ctx, b);
    const size_t need = ctx->vstride[b] * (size_t)ctx->verSlots;
    if (buf.bytes < need) {
        DevBuf nb; HIPC(nb.ensure(need));
        if (buf.p && ctx->vbytes[b]) HIPC(hipMemcpyAsync(nb.p, vb_ptr(ctx, b, ctx->vcur[b]), ctx->vbytes[b], hipMemcpyDeviceToDevice, ctx->stream));
        HIPC(hipStreamSynchronize(ctx->stream));            // (the old allocation is released right below; one-time cost per scene)
        buf.release(); buf = nb; ctx->vcur[b] = 0;
    }
    ctx->valloc[b] = ctx->verSlots;
    return IDKPT_OK;
}
// Where an update may write buffer b: *dst = the slot to write, *src = the slot that holds the current state (== *dst when the update can go in place).
// `full`: the update rewrites every byte of the buffer's state (nothing to carry over).  May launch queued samples / complete a deferred bounce when no slot is free.
static int ver_writable(dev_ctx* ctx, int b, bool full, char** src, char** dst)
{
    if (b == VB_NODES || b == VB_TRIVERTS) ctx->wideFillValid = false;   // (node boxes or triangle positions are about to change: boxes and leaf records of the wide nodes are re-derived before their next use)
    if (b == VB_NODES || b == VB_TNODES) ctx->quadValid = false;   // (somebody is about to rewrite node boxes: the quad records are re-derived before their next use)
    const int p = ctx->vcur[b];
    auto free_slot = [&](uint64_t busy) { if (ctx->verSlots > 1 && ctx->vbytes[b] > 0) for (int k = 0; k < ctx->verSlots; k++) if (!((busy >> k) & 1ull)) return k; return -1; };
    uint64_t pend = 0; for (const PendingSample& ps : ctx->pending) pend |= 1ull << ps.vs[b];
    const uint64_t held = ctx->defer.valid ? ctx->lastMask[b] : 0ull;                 // slots the deferred last bounce of the launched batch still reads
    if (!(((pend | held) >> p) & 1ull)) { *src = *dst = vb_ptr(ctx, b, p); return IDKPT_OK; }
    int q = free_slot(pend | held);
    if (q < 0 && held && (!((pend >> p) & 1ull) || free_slot(pend) >= 0)) {
        // no room, and completing the deferred bounce makes some (the current slot itself, or another one): cheaper than launching a short batch
        int rc = finish_deferred(ctx); if (rc) return rc;
        if (!((pend >> p) & 1ull)) { *src = *dst = vb_ptr(ctx, b, p); return IDKPT_OK; }
        q = free_slot(pend);
    }
    if (q < 0) {
        // every slot is pinned by queued samples: they are launched now (and the continuation they defer completed); the write is ordered behind them on the stream
        int rc = flush_any(ctx); if (rc) return rc;
        rc = finish_deferred(ctx); if (rc) return rc;
        *src = *dst = vb_ptr(ctx, b, p);
        return IDKPT_OK;
    }
    { int rc = ver_grow(ctx, b); if (rc) return rc; }
    *src = vb_ptr(ctx, b, ctx->vcur[b]); *dst = vb_ptr(ctx, b, q);
    if (!full) { HIPC(hipMemcpyAsync(*dst, *src, ctx->vbytes[b], hipMemcpyDeviceToDevice, ctx->stream)); *src = *dst; }
    ctx->vcur[b] = q;
    return IDKPT_OK;
}
// An update of something the render kernels read that is NOT versioned (materials, meshes, lights, settings ...): queued samples are launched first.
// Small host -> device update without a stream synchronisation: the bytes are staged in a pinned ring (4 x 256 KB) the copy engine reads later.
#define STAGE_BYTES (256u * 1024u)
static int staged_upload(dev_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (bytes == 0) return IDKPT_OK;
    if (bytes > STAGE_BYTES) { HIPC(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream)); HIPC(hipStreamSynchronize(ctx->stream)); return IDKPT_OK; }   // (host arrays are only borrowed for the duration of the call)
    if (!ctx->hStage) { HIPC(hipHostMalloc((void**)&ctx->hStage, (size_t)4 * STAGE_BYTES, hipHostMallocDefault)); for (int i = 0; i < 4; i++) HIPC(hipEventCreateWithFlags(&ctx->evStage[i], hipEventDisableTiming)); ctx->stageNext = 0; for (int i = 0; i < 4; i++) HIPC(hipEventRecord(ctx->evStage[i], ctx->stream)); }
    const int k = ctx->stageNext; ctx->stageNext = (k + 1) & 3;
    HIPC(hipEventSynchronize(ctx->evStage[k]));                // the copy that last read this quarter has finished
    memcpy(ctx->hStage + (size_t)k * STAGE_BYTES, src, bytes);
    HIPC(hipMemcpyAsync(dst, ctx->hStage + (size_t)k * STAGE_BYTES, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPC(hipEventRecord(ctx->evStage[k], ctx->stream));
    return IDKPT_OK;
}

// Rows a traversal stack needs for one BLAS: BLAS.ComputeRequiredStackSize (Bvh/BLAS.cs:672-702) evaluated bottom-up.  Requires what the
// validation established first: every child pair lies behind its parent (acyclic), so one reverse sweep over the node array suffices.
static int blas_stack_need(const GpuBlasNode* nodes, int nodeCount)
{
    std::vector<int> need((size_t)nodeCount, 0);                       // need[p] = stack rows while traversing the pair (p, p+1)
    for (int p = nodeCount - 2; p >= 2; p--) {
        const GpuBlasNode& l = nodes[p]; const GpuBlasNode& r = nodes[p + 1];
        const bool tl = l.TriCount == 0 && l.TriStartOrChild != 0, tr = r.TriCount == 0 && r.TriStartOrChild != 0;
        if (tl && tr) need[p] = std::max(need[l.TriStartOrChild], need[r.TriStartOrChild]) + 1;
        else if (tl || tr) need[p] = need[tl ? l.TriStartOrChild : r.TriStartOrChild];
    }
    return nodeCount > 2 ? need[2] : 0;
}

// Host-built TLAS nodes (BVH.TlasBuild / TLAS.Build, Bvh/TLAS.cs:28-141: parents are placed in front of their children): index
// validation + the depth the per-lane TLAS stack must hold.  Returns < 0 with `why` set when the array is not a valid tree.
static int tlas_validate(const GpuTlasNode* nodes, int nodeCount, int instanceCount, const char** why)
{
    if (nodeCount <= 0) return 0;
    const uint32_t* w = (const uint32_t*)nodes;                         // 8 dwords per node: Min.xyz, IsLeaf:1|ChildOrInstanceID:31, Max.xyz, pad
    std::vector<int> need((size_t)nodeCount, 0);
    for (int i = nodeCount - 1; i >= 0; i--) {
        const uint32_t packed = w[8 * (size_t)i + 3], id = packed & 0x7fffffffu;
        if (packed >> 31) { if (id >= (uint32_t)instanceCount) { *why = "TLAS leaf references an instance out of range"; return -1; } continue; }
        if (id <= (uint32_t)i || (uint64_t)id + 1 >= (uint64_t)nodeCount) { *why = "TLAS child index out of range (children must lie behind their parent)"; return -1; }
        need[i] = std::max(need[id], need[id + 1]) + 1;
    }
    return need[0];
}

// BLAS node arrays a host hands over (idkptUploadScene, idkptUpdateBuffer on IDKPT_BUF_BLAS_NODES): index validation so that a bad array cannot
// fault the GPU, plus the traversal stack the trees need.  Returns null when valid, else the reason.
static const char* validate_blas_nodes(const GpuBlasNode* nodes, int nodeCount, const GpuBlasDesc* descs, int descCount, int triangleCount, bool checkClaim, int* outMaxStack)
{
    int maxStack = 1;
    for (int i = 0; i < descCount; i++) {
        const GpuBlasDesc& d = descs[i];
        if (!(d.NodeOffset >= 0 && d.NodeCount >= 4 && d.NodeOffset + d.NodeCount <= nodeCount && d.TriangleOffset >= 0 && d.TriangleOffset + d.TriangleCount <= triangleCount)) return "BlasDesc range out of bounds";
        for (int n = 1; n < d.NodeCount; n++) {
            const GpuBlasNode& nd = nodes[d.NodeOffset + n];
            if (nd.TriCount > 0) { if (!((uint64_t)nd.TriStartOrChild + nd.TriCount <= (uint64_t)d.TriangleCount)) return "leaf triangle range out of bounds"; }
            else if (n == 1 || nd.TriStartOrChild != 0) { if (!(nd.TriStartOrChild >= 2 && nd.TriStartOrChild > (uint32_t)n && nd.TriStartOrChild + 1 < (uint32_t)d.NodeCount)) return "child index out of bounds (children must lie behind their parent)"; }
        }
        // the traversal stack is sized from what the tree really needs (BLAS.ComputeRequiredStackSize, Bvh/BLAS.cs:672-702); a host that
        // claims less in RequiredStackSize would have compiled the reference's shaders with too small a BLAS_STACK_SIZE (Bvh/BVH.cs:559-567)
        const int need = blas_stack_need(nodes + d.NodeOffset, d.NodeCount);
        if (checkClaim && d.RequiredStackSize < need) return "BlasDesc.RequiredStackSize is smaller than the stack the BLAS needs";
        maxStack = std::max(maxStack, need);
    }
    *outMaxStack = maxStack;
    return nullptr;
}

// Are all BLASes nested — every child box inside its parent's box?  (What the builder produces, and what a refit keeps: a parent is the union of its children.
// k_trace2s's exactness argument needs it; a host-patched tree that is not nested simply never gets that kernel.)  NaN coordinates compare false: not nested.
static bool blas_nested(const GpuBlasNode* nodes, const GpuBlasDesc* descs, int descCount)
{
    for (int i = 0; i < descCount; i++) {
        const GpuBlasDesc& d = descs[i];
        for (int n = 1; n < d.NodeCount; n++) {
            const GpuBlasNode& p = nodes[d.NodeOffset + n];
            if (p.TriCount > 0 || p.TriStartOrChild == 0) continue;
            for (int k = 0; k < 2; k++) {
                const GpuBlasNode& c = nodes[d.NodeOffset + p.TriStartOrChild + k];
                if (c.TriCount == 0 && c.TriStartOrChild == 0 && n != 0) continue;          // (an empty node is never entered)
                for (int a = 0; a < 3; a++) if (!(c.Min[a] >= p.Min[a] && c.Max[a] <= p.Max[a])) return false;
            }
        }
    }
    return true;
}

static int derive_nodes(dev_ctx* ctx, int blasId)
{
    if (!ctx->layoutActive) return IDKPT_OK;
    const GpuBlasDesc& d = ctx->hDescs[blasId];
    const uint32_t pairs = (uint32_t)d.NodeCount / 2;
    char *src, *dst; int rc = ver_writable(ctx, VB_TNODES, ctx->hDescs.size() == 1 && d.NodeOffset == 0 && d.NodeCount == ctx->nodeCount, &src, &dst); if (rc) return rc;
    hipLaunchKernelGGL(k_derive_nodes, dim3((pairs + 255) / 256), dim3(256), 0, ctx->stream, (const float4*)vb_cur<float4>(ctx, VB_NODES), (const uint32_t*)ctx->nodeSlot.as<uint32_t>(), (float4*)dst, (uint32_t)d.NodeOffset, pairs);
    HIPC(hipGetLastError());
    return IDKPT_OK;
}

// The derived node order of the whole scene (node_layout.hpp), from a host copy of the (validated) reference nodes: one permutation per BLAS, the
// slots on the device, then k_derive_nodes per BLAS.  Scenes whose BLAS ranges are not disjoint, pair-aligned pieces of the node array keep the
// reference order (the traversal then reads `nodes`): never produced by the reference's builder, legal for the traversal.
static int rebuild_node_layout(dev_ctx* ctx, const GpuBlasNode* hostNodes)
{
    ctx->layoutActive = false;
    ctx->vbytes[VB_TNODES] = 0; ctx->vstride[VB_TNODES] = 0; ctx->valloc[VB_TNODES] = 1; ctx->vcur[VB_TNODES] = 0;   // (callers have launched / completed everything that read the old order)
    if (ctx->opt.nodeLayout == 0) { ctx->tnodes.release(); ctx->nodeSlot.release(); return IDKPT_OK; }
    const size_t pairsTotal = (size_t)ctx->nodeCount / 2;
    std::vector<std::pair<int, int>> ranges;
    for (const GpuBlasDesc& d : ctx->hDescs) { if ((d.NodeOffset & 1) || (d.NodeCount & 1)) return IDKPT_OK; ranges.push_back({d.NodeOffset, d.NodeCount}); }
    std::sort(ranges.begin(), ranges.end());
    for (size_t i = 1; i < ranges.size(); i++) if (ranges[i].first < ranges[i - 1].first + ranges[i - 1].second) return IDKPT_OK;   // shared / overlapping node ranges
    std::vector<uint32_t> slots(pairsTotal + 1, 0u), one;
    for (const GpuBlasDesc& d : ctx->hDescs) {
        nodelayout::compute((const nodelayout::Node*)(hostNodes + d.NodeOffset), d.NodeCount, (uint32_t)d.NodeOffset / 2, ctx->opt.nodeLayout, ctx->opt.treeletDepth, one);
        std::copy(one.begin(), one.end(), slots.begin() + d.NodeOffset / 2);
    }
    HIPC(ctx->nodeSlot.ensure(slots.size() * 4)); HIPC(ctx->tnodes.ensure(std::max<size_t>((size_t)ctx->nodeCount * 32, 64)));
    HIPC(hipMemcpyAsync(ctx->nodeSlot.p, slots.data(), slots.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    ctx->layoutActive = true;
    ctx->vbytes[VB_TNODES] = (size_t)ctx->nodeCount * 32; ctx->vstride[VB_TNODES] = (ctx->vbytes[VB_TNODES] + 255) / 256 * 256;
    for (int b = 0; b < (int)ctx->hDescs.size(); b++) { int rc = derive_nodes(ctx, b); if (rc) { ctx->layoutActive = false; return rc; } }
    HIPC(hipStreamSynchronize(ctx->stream));            // `slots` is a stack vector
    return IDKPT_OK;
}

// ---- single-device implementation of the C-ABI (dev_*); the exported entry points and the multi-device group layer are in idkpt_api.hpp

static const char* dev_GetVersionString(void) { return "idkpt 0.1 (gfx950)"; }

static int32_t dev_GetDeviceCount(int32_t* outCount)
{
    int n = 0; hipError_t e = hipGetDeviceCount(&n);
    if (outCount) *outCount = (e == hipSuccess) ? n : 0;
    return e == hipSuccess ? IDKPT_OK : IDKPT_ERR_NO_DEVICE;
}

static int32_t dev_Create(int32_t deviceCount, const int32_t* deviceIds, dev_ctx** outCtx)
{
    if (!outCtx) return IDKPT_ERR_INVALID_ARGUMENT;
    *outCtx = nullptr;
    if (deviceCount != 1) return IDKPT_ERR_INVALID_ARGUMENT; // one member per device (several devices: the group layer, idkpt_api.hpp)
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return IDKPT_ERR_NO_DEVICE;
    int dev = deviceIds ? deviceIds[0] : 0;
    if (dev < 0 || dev >= n) return IDKPT_ERR_INVALID_ARGUMENT;
    if (hipSetDevice(dev) != hipSuccess) return IDKPT_ERR_HIP;
    dev_ctx* ctx = new dev_ctx();
    ctx->device = dev;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) ctx->numCUs = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return IDKPT_ERR_HIP; }
    memset(&ctx->st, 0, sizeof(ctx->st));
    ctx->st.Gpu.FocalLength = 8.0f; ctx->st.Gpu.DoRussianRoulette = 1; ctx->st.RayDepth = 7; ctx->st.SamplesPerPixel = 1;
    ctx->stCaller = ctx->st;
    memset(&ctx->stats, 0, sizeof(ctx->stats));
    memset(ctx->invProj, 0, 64); memset(ctx->invView, 0, 64); memset(ctx->viewPos, 0, 12);
    if (hipHostMalloc((void**)&ctx->hCounts, MAX_DEPTH_SLOTS * 4 + 16, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer((void**)&ctx->dCountsMirror, ctx->hCounts, 0) != hipSuccess) { delete ctx; return IDKPT_ERR_OUT_OF_MEMORY; }
    memset(ctx->hCounts, 0, MAX_DEPTH_SLOTS * 4 + 16);
    if (hipHostMalloc((void**)&ctx->hOverflow, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer((void**)&ctx->dOverflow, ctx->hOverflow, 0) != hipSuccess) { delete ctx; return IDKPT_ERR_OUT_OF_MEMORY; }
    *ctx->hOverflow = 0;
    if (hipHostMalloc((void**)&ctx->hBases, MAX_DEPTH_SLOTS * (MAX_BATCH + 1) * 4, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer((void**)&ctx->dBasesMirror, ctx->hBases, 0) != hipSuccess) { delete ctx; return IDKPT_ERR_OUT_OF_MEMORY; }
    memset(ctx->hBases, 0, MAX_DEPTH_SLOTS * (MAX_BATCH + 1) * 4);
    (void)hipEventCreate(&ctx->evFrame[0]); (void)hipEventCreate(&ctx->evFrame[1]);
    *outCtx = ctx;
    return IDKPT_OK;
}

static void builder_scratch_free(dev_ctx* ctx);
static int32_t dev_Destroy(dev_ctx* ctx)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(ctx->device);
    ctx->pending.clear();
    (void)hipStreamSynchronize(ctx->stream);
    DevBuf* all[] = {&ctx->wnodes, &ctx->wleaf, &ctx->wids, &ctx->wpair, &ctx->wcounts, &ctx->wtotals, &ctx->nodes, &ctx->tnodes, &ctx->nodeSlot, &ctx->ordKeys[0], &ctx->ordKeys[1], &ctx->ordVals[0], &ctx->ordVals[1], &ctx->ordIdx, &ctx->tris, &ctx->triVerts, &ctx->descs, &ctx->instances, &ctx->tlas, &ctx->parents, &ctx->leaves, &ctx->positions, &ctx->prevPositions, &ctx->vertices, &ctx->meshes,
                     &ctx->materials, &ctx->xforms, &ctx->lights, &ctx->sky, &ctx->texDescs, &ctx->unskinned, &ctx->joints, &ctx->levelNodes, &ctx->tlasScratch, &ctx->queryIn, &ctx->queryOut, &ctx->queryRec, &ctx->queryList, &ctx->quads, &ctx->bandTab, &ctx->tileClass, &ctx->gbases, &ctx->camTab, &ctx->verTab, &ctx->trRec, &ctx->contFlag, &ctx->blockSums, &ctx->rayO, &ctx->rayT, &ctx->rayR, &ctx->aovA, &ctx->aovN, &ctx->hit,
                     &ctx->hitCost, &ctx->primHit, &ctx->queue[0], &ctx->queue[1], &ctx->keys[0], &ctx->keys[1], &ctx->keysTmp, &ctx->sortKeys, &ctx->sortVals, &ctx->contMask, &ctx->waveCounts,
                     &ctx->counts, &ctx->work, &ctx->qwork, &ctx->radSave, &ctx->deferCount, &ctx->sortHist, &ctx->counters64, &ctx->bases, &ctx->img[0], &ctx->img[1], &ctx->img[2]};
    for (DevBuf* b : all) b->release();
    for (auto& t : ctx->texData) t.release();
    builder_scratch_free(ctx);
    if (ctx->hCounts) (void)hipHostFree(ctx->hCounts);
    if (ctx->hOverflow) (void)hipHostFree(ctx->hOverflow);
    if (ctx->hBases) (void)hipHostFree(ctx->hBases);
    if (ctx->hCams) (void)hipHostFree(ctx->hCams);
    for (int i = 0; i < 2; i++) if (ctx->evCams[i]) (void)hipEventDestroy(ctx->evCams[i]);
    if (ctx->hVerTab) (void)hipHostFree(ctx->hVerTab);
    for (int i = 0; i < 2; i++) if (ctx->evVer[i]) (void)hipEventDestroy(ctx->evVer[i]);
    if (ctx->hStage) (void)hipHostFree(ctx->hStage);
    for (int i = 0; i < 4; i++) if (ctx->evStage[i]) (void)hipEventDestroy(ctx->evStage[i]);
    if (ctx->evFrame[0]) (void)hipEventDestroy(ctx->evFrame[0]);
    if (ctx->evFrame[1]) (void)hipEventDestroy(ctx->evFrame[1]);
    for (hipEvent_t e : ctx->evPool) (void)hipEventDestroy(e);
    if (ctx->ownStream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return IDKPT_OK;
}

static int32_t dev_GetLastError(dev_ctx* ctx, const char** outMessage)
{
    if (!ctx || !outMessage) return IDKPT_ERR_INVALID_ARGUMENT;
    *outMessage = ctx->lastError.c_str();
    return IDKPT_OK;
}

static int32_t dev_SetSize(dev_ctx* ctx, int32_t width, int32_t height)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(width > 0 && height > 0 && width <= 4096 && height <= 65536, "idkptSetSize: bad size (FirstHit seeds pack x into 12 bits: width <= 4096)");
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    REQUIRE(ctx->rowLimit == 0x7fffffff || ctx->rowRem + ctx->rowLimit <= height, "idkptSetSize: the strip set by idkptSetRowRange exceeds the new image height (set a new range first)");
    REQUIRE((ctx->rowRem << ctx->rowBandLog2) < height, "idkptSetSize: this context's row remainder (idkptSetRowSharding / idkptSetRowBands) is outside the new image height");
    ctx->W = width; ctx->H = height; ctx->rows = std::min(ctx->rowLimit, local_rows(height, ctx->rowMod, ctx->rowRem, ctx->rowBandLog2));
    return alloc_frame(ctx);
}

// size and row layout in one step (group layer): bands (y >> bandLog2) % rowMod == rowRem (rowLimit = 0x7fffffff) or the strip [rowRem, rowRem + rowLimit) (rowMod = 1, bandLog2 = 0)
static int32_t dev_SetLayout(dev_ctx* ctx, int32_t width, int32_t height, int32_t rowMod, int32_t rowRem, int32_t rowLimit, int32_t bandLog2 = 0)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(width > 0 && height > 0 && width <= 4096 && height <= 65536, "idkptSetSize: bad size (FirstHit seeds pack x into 12 bits: width <= 4096)");
    REQUIRE(rowMod >= 1 && rowRem >= 0 && bandLog2 >= 0 && bandLog2 <= 6 && (rowRem << bandLog2) < height && rowLimit >= 1 && (rowMod == 1 ? bandLog2 == 0 : rowRem < rowMod), "internal: bad row layout");
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    ctx->W = width; ctx->H = height; ctx->rowMod = rowMod; ctx->rowRem = rowRem; ctx->rowLimit = rowLimit; ctx->rowBandLog2 = bandLog2;
    ctx->rows = std::min(rowLimit, local_rows(height, rowMod, rowRem, bandLog2));
    return alloc_frame(ctx);
}

// rows dealt in bands of bandRows rows: band k of the image (rows [k * bandRows, (k + 1) * bandRows)) belongs to the context with k % rowModulo == rowRemainder
// (bandRows = 1: idkptSetRowSharding)
static int32_t dev_SetRowBands(dev_ctx* ctx, int32_t bandRows, int32_t rowModulo, int32_t rowRemainder)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(bandRows >= 1 && bandRows <= 64 && (bandRows & (bandRows - 1)) == 0, "idkptSetRowBands: bandRows must be a power of two in 1..64");
    REQUIRE(rowModulo >= 1 && rowRemainder >= 0 && rowRemainder < rowModulo, "idkptSetRowSharding / idkptSetRowBands: need 0 <= remainder < modulo");
    int bandLog2 = 0; while ((1 << bandLog2) < bandRows) bandLog2++;
    if (rowModulo == 1) bandLog2 = 0;                                   // the whole frame: bands mean nothing
    REQUIRE(ctx->W <= 0 || (rowRemainder << bandLog2) < ctx->H, "idkptSetRowSharding / idkptSetRowBands: no row of the image has this remainder");
    FLUSH();
    ctx->rowMod = rowModulo; ctx->rowRem = rowRemainder; ctx->rowLimit = 0x7fffffff; ctx->rowBandLog2 = bandLog2;
    if (ctx->W > 0) { HIPC(hipSetDevice(ctx->device)); ctx->rows = local_rows(ctx->H, rowModulo, rowRemainder, bandLog2); return alloc_frame(ctx); }
    return IDKPT_OK;
}
static int32_t dev_SetRowSharding(dev_ctx* ctx, int32_t rowModulo, int32_t rowRemainder) { return dev_SetRowBands(ctx, 1, rowModulo, rowRemainder); }

static int32_t dev_SetRowRange(dev_ctx* ctx, int32_t firstRow, int32_t rowCount)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(firstRow >= 0 && rowCount >= 1, "idkptSetRowRange: need firstRow >= 0 and rowCount >= 1");
    REQUIRE(ctx->W <= 0 || firstRow + rowCount <= ctx->H, "idkptSetRowRange: strip exceeds the image height");
    FLUSH();
    ctx->rowMod = 1; ctx->rowRem = firstRow; ctx->rowLimit = rowCount; ctx->rowBandLog2 = 0;
    if (ctx->W > 0) { HIPC(hipSetDevice(ctx->device)); ctx->rows = std::min(ctx->rowLimit, local_rows(ctx->H, 1, firstRow)); return alloc_frame(ctx); }
    return IDKPT_OK;
}

static int32_t dev_SetBounceExchange(dev_ctx* ctx, idkpt_bounce_exchange_fn fn, void* user)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    FLUSH();
    ctx->exchangeFn = fn; ctx->exchangeUser = user;
    return IDKPT_OK;
}

static int32_t dev_SetBandExchange(dev_ctx* ctx, idkpt_band_exchange_fn fn, void* user)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    FLUSH();
    ctx->bandExchangeFn = fn; ctx->bandExchangeUser = user;
    return IDKPT_OK;
}

static int32_t dev_SetBandExchangeDevice(dev_ctx* ctx, idkpt_band_exchange_device_fn fn, void* user)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    FLUSH();
    ctx->bandExchangeDevFn = fn; ctx->bandExchangeDevUser = user;
    return IDKPT_OK;
}

static int32_t dev_SetSettings(dev_ctx* ctx, const idkpt_settings* s)
{
    if (!ctx || !s) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(s->RayDepth >= 1 && s->RayDepth < MAX_DEPTH_SLOTS - 1, "idkptSetSettings: RayDepth out of range");
    REQUIRE(s->SamplesPerPixel >= 1, "idkptSetSettings: SamplesPerPixel must be >= 1");
    REQUIRE(s->BlasStackSize >= 0, "idkptSetSettings: BlasStackSize must be >= 0 (0 = derive from BlasDescs)");
    // BVH.BlasStackSize is the maximum RequiredStackSize of all BLASes (Bvh/BVH.cs:559-567): a smaller stack cannot hold the traversal
    REQUIRE(s->BlasStackSize == 0 || !ctx->haveScene || s->BlasStackSize >= ctx->sceneStack, "idkptSetSettings: BlasStackSize is smaller than the scene's maximum RequiredStackSize");
    if (memcmp(&ctx->stCaller, s, sizeof(*s)) == 0) return IDKPT_OK;   // the struct the host pushed last time: nothing changed
    FLUSH();                                                            // pending samples were submitted under the old settings
    ctx->stCaller = *s;
    const idkpt_settings o = ctx->st;
    // PathTracer setters that call ResetAccumulation (PathTracer.cs:17-98): RayDepth, FocalLength, LenseRadius, DoDebugBVHTraversal, DoTraceLights
    bool reset = o.RayDepth != s->RayDepth || o.Gpu.FocalLength != s->Gpu.FocalLength || o.Gpu.LenseRadius != s->Gpu.LenseRadius ||
                 o.Gpu.DoDebugBVHTraversal != s->Gpu.DoDebugBVHTraversal || o.Gpu.DoTraceLights != s->Gpu.DoTraceLights || o.UseTlas != s->UseTlas;
    ctx->st = *s;
    if (ctx->st.Gpu.DoDebugBVHTraversal) ctx->st.RayDepth = 1; // PathTracer.cs:67-71
    if (reset) std::fill(ctx->accum.begin(), ctx->accum.end(), 0u);
    return IDKPT_OK;
}
static int32_t dev_GetSettings(dev_ctx* ctx, idkpt_settings* out) { if (!ctx || !out) return IDKPT_ERR_INVALID_ARGUMENT; *out = ctx->st; return IDKPT_OK; }

static int32_t dev_SetPerFrame(dev_ctx* ctx, const float invProjection[16], const float invView[16], const float viewPos[3])
{
    if (!ctx || !invProjection || !invView || !viewPos) return IDKPT_ERR_INVALID_ARGUMENT;
    // one camera per batch unless a frame ring is active (then every queued sample carries its own camera)
    if (ctx->ringSize == 1 && (memcmp(ctx->invProj, invProjection, 64) || memcmp(ctx->invView, invView, 64) || memcmp(ctx->viewPos, viewPos, 12))) FLUSH_KEEP();
    memcpy(ctx->invProj, invProjection, 64); memcpy(ctx->invView, invView, 64); memcpy(ctx->viewPos, viewPos, 12);
    return IDKPT_OK;
}
static int32_t dev_SetPerFrameData(dev_ctx* ctx, const GpuPerFrameData* p) { if (!ctx || !p) return IDKPT_ERR_INVALID_ARGUMENT; return dev_SetPerFrame(ctx, p->InvProjection, p->InvView, p->ViewPos); }

// triVerts[first, first + count) from the current positions, into a slot queued samples do not read (ver_writable)
static int regather_triverts(dev_ctx* ctx, uint32_t first, uint32_t count)
{
    if (count == 0) return IDKPT_OK;
    char *src, *dst; int rc = ver_writable(ctx, VB_TRIVERTS, first == 0 && count == (uint32_t)ctx->triCount, &src, &dst); if (rc) return rc;
    hipLaunchKernelGGL(k_gather_triverts, dim3((count + 255) / 256), dim3(256), 0, ctx->stream, ctx->tris.as<uint4>(), ctx->positions.as<float>(), (float4*)dst, first, count);
    HIPC(hipGetLastError());
    return IDKPT_OK;
}

static int upload(dev_ctx* ctx, DevBuf& b, const void* src, size_t bytes)
{
    HIPC(b.ensure(std::max<size_t>(bytes, 16)));
    if (bytes) HIPC(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return IDKPT_OK;
}

// The fast path stores nothing but a flag for pre-culled pixels of the most recent sample; this completes their ray state (origin,
// direction, miss radiance) from the frame constants of that batch.  Must run while the scene the batch was rendered with is still
// resident (the sky decides the miss radiance): called by idkptDownloadRays and before a new scene replaces the old one.
static DScene make_dscene_last(dev_ctx* ctx);
static int materialize_culled_rays(dev_ctx* ctx)
{
    if (!ctx->lastNeedsRegen) return IDKPT_OK;
    const size_t N = (size_t)ctx->W * ctx->rows;
    RayBufs rays = {ctx->rayO.as<float4>(), ctx->rayT.as<float4>(), ctx->rayR.as<float4>(), ctx->aovA.as<float4>(), ctx->aovN.as<float4>()};
    hipLaunchKernelGGL(k_regen_culled, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, make_dscene_last(ctx), ctx->lastFrame, rays, (const uint8_t*)ctx->contFlag.as<uint8_t>(), (uint32_t)(ctx->lastBatch - 1), (uint32_t)N);
    HIPC(hipGetLastError());
    ctx->lastNeedsRegen = false;
    return IDKPT_OK;
}

static int32_t dev_UploadScene(dev_ctx* ctx, const idkpt_scene_desc* sc)
{
    if (!ctx || !sc) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(sc->BlasNodes && sc->BlasNodeCount >= 4, "idkptUploadScene: BlasNodes missing");
    REQUIRE(sc->BlasTriangles && sc->BlasTriangleCount > 0, "idkptUploadScene: BlasTriangles missing");
    REQUIRE(sc->BlasDescs && sc->BlasDescCount > 0 && sc->BlasInstances && sc->BlasInstanceCount > 0, "idkptUploadScene: BlasDescs/BlasInstances missing");
    REQUIRE(sc->VertexPositions && sc->Vertices && sc->VertexCount > 0, "idkptUploadScene: vertices missing");
    REQUIRE(sc->Meshes && sc->MeshCount > 0 && sc->Materials && sc->MaterialCount > 0 && sc->MeshTransforms && sc->MeshTransformCount > 0, "idkptUploadScene: meshes/materials/transforms missing");
    REQUIRE(sc->LightCount >= 0 && sc->LightCount <= IDKPT_MAX_LIGHTS, "idkptUploadScene: more than 256 lights");
    // validate indices so that a bad host array cannot fault the GPU
    for (int i = 0; i < sc->BlasTriangleCount; i++) { const GpuBlasTriangle& t = sc->BlasTriangles[i]; REQUIRE(t.X < (uint32_t)sc->VertexCount && t.Y < (uint32_t)sc->VertexCount && t.Z < (uint32_t)sc->VertexCount && t.MeshId < (uint32_t)sc->MeshCount, "idkptUploadScene: BlasTriangle index out of range"); }
    for (int i = 0; i < sc->MeshCount; i++) REQUIRE(sc->Meshes[i].MaterialId >= 0 && sc->Meshes[i].MaterialId < sc->MaterialCount, "idkptUploadScene: Mesh.MaterialId out of range");
    for (int i = 0; i < sc->BlasInstanceCount; i++) REQUIRE(sc->BlasInstances[i].BlasId < (uint32_t)sc->BlasDescCount && sc->BlasInstances[i].MeshTransformId < (uint32_t)sc->MeshTransformCount, "idkptUploadScene: BlasInstance out of range");
    int maxStack = 1;
    { const char* why = validate_blas_nodes(sc->BlasNodes, sc->BlasNodeCount, sc->BlasDescs, sc->BlasDescCount, sc->BlasTriangleCount, true, &maxStack); REQUIRE(why == nullptr, std::string("idkptUploadScene: ") + (why ? why : "")); }
    REQUIRE(ctx->st.BlasStackSize == 0 || ctx->st.BlasStackSize >= maxStack, "idkptUploadScene: the BlasStackSize set with idkptSetSettings is smaller than this scene's maximum RequiredStackSize");
    int tlasNeed = 1;
    if (sc->TlasNodes && sc->TlasNodeCount > 0) {
        const char* why = nullptr;
        tlasNeed = tlas_validate(sc->TlasNodes, sc->TlasNodeCount, sc->BlasInstanceCount, &why);
        REQUIRE(tlasNeed >= 0, std::string("idkptUploadScene: ") + (why ? why : "bad TLAS"));
        REQUIRE(tlasNeed <= TLAS_STACK_SIZE, "idkptUploadScene: TLAS deeper than TLAS_STACK_SIZE (32)");
    }
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    { int rc = materialize_culled_rays(ctx); if (rc) return rc; }   // while the old sky is still resident
    int rc;
    if ((rc = upload(ctx, ctx->nodes, sc->BlasNodes, (size_t)sc->BlasNodeCount * 32))) return rc;
    if ((rc = upload(ctx, ctx->tris, sc->BlasTriangles, (size_t)sc->BlasTriangleCount * 16))) return rc;
    if ((rc = upload(ctx, ctx->descs, sc->BlasDescs, (size_t)sc->BlasDescCount * sizeof(GpuBlasDesc)))) return rc;
    if ((rc = upload(ctx, ctx->instances, sc->BlasInstances, (size_t)sc->BlasInstanceCount * 8))) return rc;
    if ((rc = upload(ctx, ctx->tlas, sc->TlasNodes, (size_t)(sc->TlasNodes ? sc->TlasNodeCount : 0) * 32))) return rc;
    if ((rc = upload(ctx, ctx->parents, sc->BlasParentIndices, (size_t)(sc->BlasParentIndices ? sc->BlasParentIndexCount : 0) * 4))) return rc;
    if ((rc = upload(ctx, ctx->leaves, sc->BlasLeafIndices, (size_t)(sc->BlasLeafIndices ? sc->BlasLeafIndexCount : 0) * 4))) return rc;
    if ((rc = upload(ctx, ctx->positions, sc->VertexPositions, (size_t)sc->VertexCount * 12))) return rc;
    if ((rc = upload(ctx, ctx->vertices, sc->Vertices, (size_t)sc->VertexCount * 16))) return rc;
    if ((rc = upload(ctx, ctx->meshes, sc->Meshes, (size_t)sc->MeshCount * sizeof(GpuMesh)))) return rc;
    if ((rc = upload(ctx, ctx->materials, sc->Materials, (size_t)sc->MaterialCount * sizeof(GpuMaterial)))) return rc;
    if ((rc = upload(ctx, ctx->xforms, sc->MeshTransforms, (size_t)sc->MeshTransformCount * sizeof(GpuMeshTransform)))) return rc;
    HIPC(ctx->lights.ensure(IDKPT_MAX_LIGHTS * sizeof(GpuLight)));
    if (sc->Lights && sc->LightCount) HIPC(hipMemcpyAsync(ctx->lights.p, sc->Lights, (size_t)sc->LightCount * sizeof(GpuLight), hipMemcpyHostToDevice, ctx->stream));
    {   // can a surface of this scene add radiance?  (k_shade_last: without emission the last bounce's hits leave the radiance alone)
        bool none = true;
        // (... and can the throughput a hit is shaded with be relied on to be finite when it came in finite?  Volumetric absorption multiplies it by exp(-absorbance x T)
        // BEFORE the emission is added (ShadeHit): an infinite Absorbance / AbsorbanceBias with T == 0 makes it NaN, and 0 x NaN is not 0 — such scenes shade every hit)
        auto fin = [](float v) { return v - v == 0.0f; };
        for (int i = 0; i < sc->MaterialCount && none; i++) { const GpuMaterial& m = sc->Materials[i]; none = m.EmissiveFactor[0] == 0.0f && m.EmissiveFactor[1] == 0.0f && m.EmissiveFactor[2] == 0.0f && fin(m.Absorbance[0]) && fin(m.Absorbance[1]) && fin(m.Absorbance[2]); }
        for (int i = 0; i < sc->MeshCount && none; i++) none = sc->Meshes[i].EmissiveBias == 0.0f && fin(sc->Meshes[i].AbsorbanceBias[0]) && fin(sc->Meshes[i].AbsorbanceBias[1]) && fin(sc->Meshes[i].AbsorbanceBias[2]);
        for (int i = 0; i < sc->TextureCount && none; i++) {      // (0 x a texel is only 0 for a finite texel)
            const idkpt_texture& t = sc->Textures[i];
            if (!(t.width > 0 && t.height > 0 && t.rgba)) { none = false; break; }
            const uint32_t* w = reinterpret_cast<const uint32_t*>(t.rgba); uint32_t bad = 0;
            for (size_t k = 0, e = (size_t)t.width * t.height * 4; k < e; k++) bad |= (uint32_t)((w[k] & 0x7f800000u) == 0x7f800000u);
            none = !bad;
        }
        ctx->sceneNoEmission = none;
    }
    ctx->skySize = (sc->SkyFaces && sc->SkyFaceSize > 0) ? sc->SkyFaceSize : 0;
    if ((rc = upload(ctx, ctx->sky, sc->SkyFaces, (size_t)6 * ctx->skySize * ctx->skySize * 16))) return rc;
    for (auto& t : ctx->texData) t.release();
    ctx->texData.clear(); ctx->texDims.clear();
    std::vector<TexDesc> td;
    for (int i = 0; i < sc->TextureCount; i++) {
        const idkpt_texture& t = sc->Textures[i];
        REQUIRE(t.width > 0 && t.height > 0 && t.rgba, "idkptUploadScene: bad texture");
        ctx->texData.emplace_back();
        if ((rc = upload(ctx, ctx->texData.back(), t.rgba, (size_t)t.width * t.height * 16))) return rc;
        td.push_back({ctx->texData.back().as<float4>(), t.width, t.height});
        ctx->texDims.push_back({t.width, t.height});
    }
    if ((rc = upload(ctx, ctx->texDescs, td.data(), td.size() * sizeof(TexDesc)))) return rc;
    HIPC(ctx->triVerts.ensure((size_t)sc->BlasTriangleCount * 48));
    ctx->nodeCount = sc->BlasNodeCount; ctx->triCount = sc->BlasTriangleCount; ctx->instanceCount = sc->BlasInstanceCount; ctx->tlasCount = sc->TlasNodes ? sc->TlasNodeCount : 0;
    ctx->vertexCount = sc->VertexCount; ctx->meshCount = sc->MeshCount; ctx->materialCount = sc->MaterialCount; ctx->xformCount = sc->MeshTransformCount;
    ctx->lightCount = sc->Lights ? sc->LightCount : 0; ctx->textureCount = sc->TextureCount;
    ctx->hDescs.assign(sc->BlasDescs, sc->BlasDescs + sc->BlasDescCount); ctx->hInst0Blas = (int)sc->BlasInstances[0].BlasId;
    ctx->sceneStack = maxStack; ctx->tlasNeed = std::max(1, tlasNeed);
    ctx->sceneNested = blas_nested(sc->BlasNodes, sc->BlasDescs, sc->BlasDescCount);
    ver_reset(ctx);                                   // one state per versioned buffer, in slot 0 (everything that read the old scene was launched by FLUSH above)
    if ((rc = ver_reserve(ctx))) return rc;
    // refit schedule: internal nodes of every refittable BLAS grouped by depth (children have larger ids than parents)
    ctx->levelOffsets.assign(sc->BlasDescCount, {}); ctx->levelBase.assign(sc->BlasDescCount, 0); ctx->refitCoversAll.assign(sc->BlasDescCount, 0);
    std::vector<int32_t> allLevels;
    for (int bi = 0; bi < sc->BlasDescCount; bi++) {
        const GpuBlasDesc& d = sc->BlasDescs[bi];
        if (!d.IsRefittable) continue;
        std::vector<int> depth(d.NodeCount, 0); int maxD = 0;
        for (int n = 1; n < d.NodeCount; n++) { const GpuBlasNode& nd = sc->BlasNodes[d.NodeOffset + n]; if (nd.TriCount == 0) { int c = (int)nd.TriStartOrChild; depth[c] = depth[c + 1] = depth[n] + 1; maxD = std::max(maxD, depth[n]); } }
        std::vector<std::vector<int32_t>> lv(maxD + 1);
        for (int n = 1; n < d.NodeCount; n++) if (sc->BlasNodes[d.NodeOffset + n].TriCount == 0 && (n == 1 || n >= 2)) lv[depth[n]].push_back(n);
        ctx->levelBase[bi] = (uint32_t)allLevels.size();
        uint32_t off = 0;
        for (auto& l : lv) { ctx->levelOffsets[bi].push_back(off); off += (uint32_t)l.size(); allLevels.insert(allLevels.end(), l.begin(), l.end()); }
        ctx->levelOffsets[bi].push_back(off);
        ctx->refitCoversAll[bi] = (int64_t)off + (int64_t)d.LeafIndicesCount == (int64_t)d.NodeCount - 1;
    }
    if ((rc = upload(ctx, ctx->levelNodes, allLevels.data(), allLevels.size() * 4))) return rc;
    if ((rc = regather_triverts(ctx, 0, (uint32_t)sc->BlasTriangleCount))) return rc;
    if ((rc = rebuild_node_layout(ctx, sc->BlasNodes))) return rc;
    HIPC(hipStreamSynchronize(ctx->stream)); // host arrays are only borrowed for the duration of the call
    ctx->haveScene = true;
    std::fill(ctx->accum.begin(), ctx->accum.end(), 0u);
    return IDKPT_OK;
}

// Device-to-device copies of a multi-device context.  xGMI peer copies (hipMemcpyPeerAsync, ordered on `st`) where the runtime grants them; on a
// node whose GPUs refuse peer access — or under the option "force_no_peer" — every copy is staged through pinned host memory instead: wait for
// `st` (so that what the stream order promised about the source holds), D2H on the source device, H2D on the destination device, both blocking.
// Slower (two PCIe crossings and a host synchronisation), same results; the first refusal is logged once.
struct PeerPolicy { bool forceStaged = false, warned = false; void* stage = nullptr; size_t stageBytes = 0; };
static hipError_t member_copy(PeerPolicy* pol, void* dst, int dstDev, const void* src, int srcDev, size_t bytes, hipStream_t st)
{
    if (bytes == 0) return hipSuccess;
    if (!pol || !pol->forceStaged) {
        hipError_t e = hipMemcpyPeerAsync(dst, dstDev, src, srcDev, bytes, st);
        if (e == hipSuccess || !pol) return e;
        (void)hipGetLastError();
        if (!pol->warned) { fprintf(stderr, "[idkpt] warning: peer copy GPU %d -> GPU %d refused (%s); staging device-to-device copies through host memory from now on\n", srcDev, dstDev, hipGetErrorString(e)); pol->warned = true; }
        pol->forceStaged = true;
    }
    hipError_t e = hipStreamSynchronize(st); if (e != hipSuccess) return e;
    if (pol->stageBytes < bytes) {
        if (pol->stage) (void)hipHostFree(pol->stage);
        pol->stage = nullptr; pol->stageBytes = 0;
        e = hipHostMalloc(&pol->stage, bytes, hipHostMallocDefault); if (e != hipSuccess) return e;
        pol->stageBytes = bytes;
    }
    int cur = 0; (void)hipGetDevice(&cur);
    e = hipSetDevice(srcDev); if (e == hipSuccess) e = hipMemcpy(pol->stage, src, bytes, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipSetDevice(dstDev);
    if (e == hipSuccess) e = hipMemcpy(dst, pol->stage, bytes, hipMemcpyHostToDevice);
    (void)hipSetDevice(cur);
    return e;
}

// Multi-device contexts: the scene one member uploaded (validated, derived layouts built) is replicated to another member device-to-device
// (hipMemcpyPeerAsync: xGMI between MI355X GPUs) instead of crossing PCIe once per GPU — the "broadcast of the BVH" of the group layer.
static int32_t dev_CloneSceneFrom(dev_ctx* ctx, dev_ctx* src)
{
    if (!ctx || !src || !src->haveScene) return IDKPT_ERR_INVALID_ARGUMENT;
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    { int rc = materialize_culled_rays(ctx); if (rc) return rc; }   // while the old sky is still resident
    DevBuf* d[] = {&ctx->nodes, &ctx->tnodes, &ctx->nodeSlot, &ctx->tris, &ctx->triVerts, &ctx->descs, &ctx->instances, &ctx->tlas, &ctx->parents, &ctx->leaves, &ctx->positions, &ctx->vertices, &ctx->meshes,
                   &ctx->materials, &ctx->xforms, &ctx->lights, &ctx->sky, &ctx->levelNodes};
    DevBuf* f[] = {&src->nodes, &src->tnodes, &src->nodeSlot, &src->tris, &src->triVerts, &src->descs, &src->instances, &src->tlas, &src->parents, &src->leaves, &src->positions, &src->vertices, &src->meshes,
                   &src->materials, &src->xforms, &src->lights, &src->sky, &src->levelNodes};
    for (size_t i = 0; i < sizeof(d) / sizeof(d[0]); i++) {
        if (!f[i]->p || f[i]->bytes == 0) continue;
        // a versioned buffer of the source may be an arena of several states: its current one goes to slot 0 here
        int vb = -1; for (int b = 0; b < VB_COUNT; b++) if (f[i] == &vb_buf(src, b)) vb = b;
        const size_t bytes = (vb >= 0 && src->vbytes[vb] > 0) ? src->vbytes[vb] : f[i]->bytes;
        const char* from = (vb >= 0 && src->vbytes[vb] > 0) ? vb_ptr(src, vb, src->vcur[vb]) : (const char*)f[i]->p;
        HIPC(d[i]->ensure(bytes));
        HIPC(member_copy(ctx->peer, d[i]->p, ctx->device, from, src->device, bytes, ctx->stream));
    }
    for (auto& t : ctx->texData) t.release();
    ctx->texData.clear(); ctx->texDims = src->texDims;
    std::vector<TexDesc> td;
    for (size_t i = 0; i < src->texData.size(); i++) {
        ctx->texData.emplace_back();
        HIPC(ctx->texData.back().ensure(src->texData[i].bytes));
        HIPC(member_copy(ctx->peer, ctx->texData.back().p, ctx->device, src->texData[i].p, src->device, src->texData[i].bytes, ctx->stream));
        td.push_back({ctx->texData.back().as<float4>(), src->texDims[i].first, src->texDims[i].second});
    }
    { int rc = upload(ctx, ctx->texDescs, td.data(), td.size() * sizeof(TexDesc)); if (rc) return rc; }
    ctx->nodeCount = src->nodeCount; ctx->triCount = src->triCount; ctx->instanceCount = src->instanceCount; ctx->tlasCount = src->tlasCount; ctx->vertexCount = src->vertexCount;
    ctx->meshCount = src->meshCount; ctx->materialCount = src->materialCount; ctx->xformCount = src->xformCount; ctx->lightCount = src->lightCount; ctx->skySize = src->skySize;
    ctx->textureCount = src->textureCount; ctx->hDescs = src->hDescs; ctx->hInst0Blas = src->hInst0Blas; ctx->quadValid = false; ctx->sceneNoEmission = src->sceneNoEmission; ctx->sceneNested = src->sceneNested; ctx->sceneStack = src->sceneStack; ctx->tlasNeed = src->tlasNeed; ctx->layoutActive = src->layoutActive;
    ctx->levelOffsets = src->levelOffsets; ctx->levelBase = src->levelBase; ctx->refitCoversAll = src->refitCoversAll;
    ver_reset(ctx);
    { int rc = ver_reserve(ctx); if (rc) return rc; }
    HIPC(hipStreamSynchronize(ctx->stream));           // td is a stack vector
    ctx->haveScene = true;
    std::fill(ctx->accum.begin(), ctx->accum.end(), 0u);
    return IDKPT_OK;
}

static int32_t dev_SetLightCount(dev_ctx* ctx, int32_t count)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(count >= 0 && count <= IDKPT_MAX_LIGHTS, "idkptSetLightCount: out of range");
    FLUSH();
    ctx->lightCount = count; return IDKPT_OK;
}

// buffer id -> allocation, size of its state, and (versioned buffers) which arena it is (-1: a plain buffer)
static DevBuf* which_buffer(dev_ctx* ctx, int which, size_t* cap, int* vb)
{
    *vb = -1;
    switch (which) {
        case IDKPT_BUF_MESH_TRANSFORMS: *cap = (size_t)ctx->xformCount * sizeof(GpuMeshTransform); *vb = VB_XFORMS; return &ctx->xforms;
        case IDKPT_BUF_VERTEX_POSITIONS: *cap = (size_t)ctx->vertexCount * 12; return &ctx->positions;
        case IDKPT_BUF_VERTICES: *cap = (size_t)ctx->vertexCount * 16; *vb = VB_VERTICES; return &ctx->vertices;
        case IDKPT_BUF_MESHES: *cap = (size_t)ctx->meshCount * sizeof(GpuMesh); return &ctx->meshes;
        case IDKPT_BUF_MATERIALS: *cap = (size_t)ctx->materialCount * sizeof(GpuMaterial); return &ctx->materials;
        case IDKPT_BUF_LIGHTS: *cap = (size_t)IDKPT_MAX_LIGHTS * sizeof(GpuLight); return &ctx->lights;
        case IDKPT_BUF_BLAS_NODES: *cap = (size_t)ctx->nodeCount * 32; *vb = VB_NODES; return &ctx->nodes;
        case IDKPT_BUF_TLAS_NODES: *cap = (size_t)ctx->tlasCount * 32; *vb = VB_TLAS; return &ctx->tlas;
        case IDKPT_BUF_JOINT_MATRICES: *cap = ctx->joints.bytes; return &ctx->joints;
        default: return nullptr;
    }
}

static int32_t dev_UpdateBuffer(dev_ctx* ctx, int32_t which, size_t offsetBytes, size_t bytes, const void* data)
{
    if (!ctx || !data) return IDKPT_ERR_INVALID_ARGUMENT;
    if (!ctx->haveScene) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptUpdateBuffer: no scene uploaded");
    HIPC(hipSetDevice(ctx->device));
    // What queued samples still have to read decides whether they are launched first: joint matrices and vertex positions are read by the update kernels only
    // (k_skin, k_gather_triverts: they run in stream order, right now); transforms, vertices and tree nodes are versioned (ver_writable finds or makes room);
    // meshes, materials and lights are read by the shading kernels of every queued sample.
    if (which == IDKPT_BUF_MESHES || which == IDKPT_BUF_MATERIALS || which == IDKPT_BUF_LIGHTS || which == IDKPT_BUF_BLAS_NODES || which == IDKPT_BUF_TLAS_NODES) FLUSH();
    if (which == IDKPT_BUF_JOINT_MATRICES) { size_t need = offsetBytes + bytes; if (need > ctx->joints.bytes) { DevBuf nb; HIPC(nb.ensure(need)); if (ctx->joints.p) { HIPC(hipMemcpyAsync(nb.p, ctx->joints.p, ctx->joints.bytes, hipMemcpyDeviceToDevice, ctx->stream)); } HIPC(hipStreamSynchronize(ctx->stream)); ctx->joints.release(); ctx->joints = nb; } }
    size_t cap = 0; int vb = -1; DevBuf* b = which_buffer(ctx, which, &cap, &vb);
    REQUIRE(b != nullptr, "idkptUpdateBuffer: unknown buffer");
    REQUIRE(offsetBytes + bytes <= cap, "idkptUpdateBuffer: range exceeds buffer");
    if (which == IDKPT_BUF_BLAS_NODES || which == IDKPT_BUF_TLAS_NODES) {
        // Patched tree nodes are validated like uploaded ones BEFORE they reach the device (a bad child index must not fault the GPU, a deeper tree must
        // not overflow the traversal stack), on a host copy of the array with the patch applied; BLAS nodes: the derived order is rebuilt from that copy.
        // The patch must leave a valid tree after EVERY call (idkpt.h): a host that streams a rebuilt tree in pieces uses idkptUploadScene / idkptBuildTlas.
        char* cur = vb_ptr(ctx, vb, ctx->vcur[vb]);           // (nothing is queued or deferred any more: the patch goes in place)
        std::vector<char> h(cap);
        HIPC(hipMemcpyAsync(h.data(), cur, cap, hipMemcpyDeviceToHost, ctx->stream)); HIPC(hipStreamSynchronize(ctx->stream));
        memcpy(h.data() + offsetBytes, data, bytes);
        if (which == IDKPT_BUF_BLAS_NODES) {
            int maxStack = 1;
            const char* why = validate_blas_nodes((const GpuBlasNode*)h.data(), ctx->nodeCount, ctx->hDescs.data(), (int)ctx->hDescs.size(), ctx->triCount, false, &maxStack);
            REQUIRE(why == nullptr, std::string("idkptUpdateBuffer: ") + (why ? why : ""));
            REQUIRE(ctx->st.BlasStackSize == 0 || ctx->st.BlasStackSize >= maxStack, "idkptUpdateBuffer: the patched BLAS needs a deeper traversal stack than the BlasStackSize set with idkptSetSettings");
            HIPC(hipMemcpyAsync(cur + offsetBytes, data, bytes, hipMemcpyHostToDevice, ctx->stream));
            ctx->sceneStack = maxStack; ctx->wideTopoValid = false; ctx->wideFillValid = false;   // (the tree itself may have changed: the wide nodes are derived anew)
            ctx->sceneNested = blas_nested((const GpuBlasNode*)h.data(), ctx->hDescs.data(), (int)ctx->hDescs.size());
            int rc = rebuild_node_layout(ctx, (const GpuBlasNode*)h.data()); if (rc) return rc;
        } else {
            const char* why = nullptr;
            const int need = tlas_validate((const GpuTlasNode*)h.data(), ctx->tlasCount, ctx->instanceCount, &why);
            REQUIRE(need >= 0, std::string("idkptUpdateBuffer: ") + (why ? why : "bad TLAS"));
            REQUIRE(need <= TLAS_STACK_SIZE, "idkptUpdateBuffer: TLAS deeper than TLAS_STACK_SIZE (32)");
            HIPC(hipMemcpyAsync(cur + offsetBytes, data, bytes, hipMemcpyHostToDevice, ctx->stream));
            ctx->tlasNeed = std::max(1, need);
        }
        HIPC(hipStreamSynchronize(ctx->stream));
        return IDKPT_OK;
    }
    if (which == IDKPT_BUF_MESHES || which == IDKPT_BUF_MATERIALS) ctx->sceneNoEmission = false;   // (a patched material may emit: decided again at the next idkptUploadScene)
    char* dst = (char*)b->p;
    if (vb >= 0) { char* src; int rc = ver_writable(ctx, vb, offsetBytes == 0 && bytes == cap, &src, &dst); if (rc) return rc; }
    { int rc = staged_upload(ctx, dst + offsetBytes, data, bytes); if (rc) return rc; }   // (small updates — joints, transforms — do not wait for the stream)
    if (which == IDKPT_BUF_VERTEX_POSITIONS) { int rc = regather_triverts(ctx, 0, (uint32_t)ctx->triCount); if (rc) return rc; }
    return IDKPT_OK;
}

// idkptSetDeveloperOption: tuning / test hooks (DevOptions above).  Unknown names are an error, so that a typo cannot silently test nothing.
static int32_t dev_SetOption(dev_ctx* ctx, const char* name, int32_t value)
{
    if (!ctx || !name) return IDKPT_ERR_INVALID_ARGUMENT;
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    DevOptions& o = ctx->opt;
    const std::string n(name);
    if (n == "force_generic") o.forceGeneric = value != 0;
    else if (n == "no_tile_cull") o.noTileCull = value != 0;
    else if (n == "no_lean_primary") o.noLeanPrimary = value != 0;
    else if (n == "leaf_min") o.leafMin = std::max(0, value);
    else if (n == "grab_unit_log2") o.grabUnitLog2 = value;
    else if (n == "grab_fixed") o.grabFixed = value;
    else if (n == "lds_pad") o.ldsPad = value;
    else if (n == "trace_waves") o.traceWaves = std::max(0, value);
    else if (n == "grid_hint") o.gridHint = std::max(0, value);
    else if (n == "grid_rays_x4") o.gridRaysX4 = std::max(0, value);
    else if (n == "grid_mid_waves") o.gridMidWaves = std::max(0, value);
    else if (n == "defer_last") o.deferLast = value != 0;
    else if (n == "split") { REQUIRE(value >= 0 && value <= 3, "idkptSetDeveloperOption: split is 0..3"); o.split = value; }
    else if (n == "split_donor") o.splitDonor = value != 0;
    else if (n == "split_peek") o.splitPeek = std::max(1, value);
    else if (n == "split_scatter") o.splitScatter = std::min(6, std::max(0, value));
    else if (n == "query_scheduler") o.queryScheduler = value != 0;
    else if (n == "park") o.park = value & 7;
    else if (n == "group_threads") o.groupThreads = value;
    else if (n == "quad") { REQUIRE(value >= 0 && value <= 2, "idkptSetDeveloperOption: quad is 0..2"); o.quad = value; }
    else if (n == "fused") { REQUIRE(value >= 0 && value <= 2, "idkptSetDeveloperOption: fused is 0..2"); o.fused = value; }
    else if (n == "fused_shade_min") o.fusedShadeMin = std::min(64, std::max(1, value));
    else if (n == "instance_records") o.instanceRecords = value != 0;
    else if (n == "leaf_pool") { REQUIRE(value >= -1 && value <= 7, "idkptSetDeveloperOption: leaf_pool is -1 (automatic) or a mask 0..7 (1: primary launches, 2: the first bounce, 4: later bounces)"); o.leafPool = value; }
    else if (n == "pool_min") o.poolMin = std::max(0, value);
    else if (n == "adv_min") o.advMin = std::min(64, std::max(0, value));
    else if (n == "spec") {
#ifdef IDKPT_DEVELOPER
        REQUIRE(value >= 0 && value <= 2, "idkptSetDeveloperOption: spec is 0..2"); o.spec = value;
#else
        REQUIRE(value == 0, "idkptSetDeveloperOption: spec needs the developer build of the library (libidkpt_dev.so)");
#endif
    }
#ifdef IDKPT_DEVELOPER
    else if (n == "graph_probe") o.graphProbe = std::max(0, value);
#endif
    else if (n == "wide") { REQUIRE(value >= 0 && value <= 1, "idkptSetDeveloperOption: wide is 0 or 1"); FLUSH(); o.wide = value; }
    else if (n == "wide_cap") { REQUIRE(value >= 0 && value <= 96, "idkptSetDeveloperOption: wide_cap is 0 (default) or 4..96 rows"); o.wideCap = value; }
    else if (n == "wide_count") o.wideCount = value != 0;
    else if (n == "bvh_timing") o.bvhTiming = value != 0;
    else if (n == "bvh_small") o.bvhSmall = value;
    else if (n == "bvh_stackopt_host") o.bvhStackOptHost = value != 0;
    else if (n == "force_no_peer") { }                 // (multi-device contexts: idkpt_api.hpp; nothing to stage on one device)
    else if (n == "trace_variant") {
#ifdef IDKPT_DEVELOPER
        o.traceVariant = value;
#else
        REQUIRE(value == 0 || value == 100, "idkptSetDeveloperOption: trace_variant needs the developer build of the library (libidkpt_dev.so)");
#endif
    }
    else if (n == "trace_order") {
        REQUIRE(value >= 0 && value <= 2, "idkptSetDeveloperOption: trace_order is 0, 1 or 2");
        o.traceOrder = value;
        if (value && ctx->W > 0 && !ctx->ordIdx.p) {       // the permutation buffers of the current frame size
            const size_t cap = (size_t)ctx->maxBatch * ctx->Npad;
            for (int i = 0; i < 2; i++) { HIPC(ctx->ordKeys[i].ensure(cap * 4)); HIPC(ctx->ordVals[i].ensure(cap * 4)); } HIPC(ctx->ordIdx.ensure(cap * 4));
        }
    }
    else if (n == "node_layout" || n == "treelet_depth") {
        if (n == "node_layout") { REQUIRE(value >= 0 && value <= 2, "idkptSetDeveloperOption: node_layout is 0, 1 or 2"); o.nodeLayout = value; }
        else { REQUIRE(value >= 1 && value <= 16, "idkptSetDeveloperOption: treelet_depth is 1..16"); o.treeletDepth = value; }
        if (ctx->haveScene) {                               // re-derive the resident scene
            std::vector<GpuBlasNode> h((size_t)ctx->nodeCount);
            HIPC(hipMemcpyAsync(h.data(), vb_cur<char>(ctx, VB_NODES), h.size() * 32, hipMemcpyDeviceToHost, ctx->stream)); HIPC(hipStreamSynchronize(ctx->stream));
            int rc = rebuild_node_layout(ctx, h.data()); if (rc) return rc;
        }
    }
    else return fail(ctx, IDKPT_ERR_INVALID_ARGUMENT, "idkptSetDeveloperOption: unknown option '" + n + "'");
    return IDKPT_OK;
}

static int32_t dev_DownloadBuffer(dev_ctx* ctx, int32_t which, size_t offsetBytes, size_t bytes, void* dst)
{
    if (!ctx || !dst) return IDKPT_ERR_INVALID_ARGUMENT;
    if (which == IDKPT_BUF_WIDE_NODES || which == IDKPT_BUF_WIDE_LEAVES || which == IDKPT_BUF_WIDE_COUNTS) {   // read-only views of the derived traversal structure (tests, tools)
        if (!ctx->haveScene) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptDownloadBuffer: no scene uploaded");
        HIPC(hipSetDevice(ctx->device));
        FLUSH();
        { int rc = wide_prepare(ctx); if (rc) return rc; }
        DevBuf& wbuf = which == IDKPT_BUF_WIDE_NODES ? ctx->wnodes : (which == IDKPT_BUF_WIDE_LEAVES ? ctx->wleaf : ctx->wcounts);
        REQUIRE(offsetBytes + bytes <= (which == IDKPT_BUF_WIDE_COUNTS ? ctx->hDescs.size() * 8 : wbuf.bytes), "idkptDownloadBuffer: bad buffer/range");
        HIPC(hipMemcpyAsync(dst, (char*)wbuf.p + offsetBytes, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIPC(hipStreamSynchronize(ctx->stream));
        return IDKPT_OK;
    }
    size_t cap = 0; int vb = -1; DevBuf* b = which_buffer(ctx, which, &cap, &vb);
    REQUIRE(b != nullptr && offsetBytes + bytes <= cap, "idkptDownloadBuffer: bad buffer/range");
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    HIPC(hipMemcpyAsync(dst, (vb >= 0 ? vb_ptr(ctx, vb, ctx->vcur[vb]) : (char*)b->p) + offsetBytes, bytes, hipMemcpyDeviceToHost, ctx->stream));   // the current state
    HIPC(hipStreamSynchronize(ctx->stream));
    return IDKPT_OK;
}

static int32_t dev_BuildTlas(dev_ctx* ctx, const GpuTlasNode* nodes, int32_t nodeCount)
{
    if (!ctx || !nodes || nodeCount <= 0) return IDKPT_ERR_INVALID_ARGUMENT;
    if (!ctx->haveScene) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptBuildTlas: no scene uploaded");
    const char* why = nullptr;
    const int need = tlas_validate(nodes, nodeCount, ctx->instanceCount, &why);
    REQUIRE(need >= 0, std::string("idkptBuildTlas: ") + (why ? why : "bad TLAS"));
    REQUIRE(need <= TLAS_STACK_SIZE, "idkptBuildTlas: TLAS deeper than TLAS_STACK_SIZE (32)");
    HIPC(hipSetDevice(ctx->device));
    if ((size_t)nodeCount * 32 > ctx->vbytes[VB_TLAS] || nodeCount != ctx->tlasCount) {
        // another node count (or more nodes than a slot holds): every queued sample is launched first, then the TLAS buffer is laid out anew
        FLUSH();
        if ((size_t)nodeCount * 32 > ctx->vbytes[VB_TLAS]) {
            HIPC(hipStreamSynchronize(ctx->stream));
            ctx->tlas.release();
            ctx->vbytes[VB_TLAS] = (size_t)nodeCount * 32; ctx->vstride[VB_TLAS] = (ctx->vbytes[VB_TLAS] + 255) / 256 * 256; ctx->valloc[VB_TLAS] = 1; ctx->vcur[VB_TLAS] = 0;
            HIPC(ctx->tlas.ensure(ctx->vstride[VB_TLAS]));
        }
    }
    char *src, *dst; int rc = ver_writable(ctx, VB_TLAS, true, &src, &dst); if (rc) return rc;
    HIPC(hipMemcpyAsync(dst, nodes, (size_t)nodeCount * 32, hipMemcpyHostToDevice, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
    ctx->tlasCount = nodeCount; ctx->tlasNeed = std::max(1, need);
    return IDKPT_OK;
}

static int32_t dev_BuildTlasOnDevice(dev_ctx* ctx, int32_t searchRadius)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    if (!ctx->haveScene) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptBuildTlasOnDevice: no scene uploaded");
    REQUIRE(searchRadius >= 1, "idkptBuildTlasOnDevice: searchRadius must be >= 1 (reference: 15)");
    HIPC(hipSetDevice(ctx->device));
    const int n = ctx->instanceCount, nodeCount = 2 * n - 1;
    if (nodeCount != ctx->tlasCount) FLUSH();            // (queued samples were queued with another node count; a slot always has room for 2n - 1 nodes: ver_reset)
    char *tsrc, *tdst; { int rc = ver_writable(ctx, VB_TLAS, true, &tsrc, &tdst); if (rc) return rc; }
    // scratch: temp nodes (2n-1) + leaves (n) as float4 pairs, keys (n), pref (n)
    const size_t tempOff = 0, leafOff = (size_t)nodeCount * 32, keyOff = leafOff + (size_t)n * 32, prefOff = keyOff + (size_t)n * 4;
    HIPC(ctx->tlasScratch.ensure(prefOff + (size_t)n * 4));
    char* sc = ctx->tlasScratch.as<char>();
    hipLaunchKernelGGL(k_tlas_build, dim3(1), dim3(TLAS_BUILD_THREADS), 0, ctx->stream, (const float4*)vb_cur<float4>(ctx, VB_NODES), ctx->descs.as<GpuBlasDesc>(), ctx->instances.as<GpuBlasInstance>(),
                       (const float4*)vb_cur<float4>(ctx, VB_XFORMS), n, (int)searchRadius, (float4*)tdst, (float4*)(sc + tempOff), (float4*)(sc + leafOff), (uint32_t*)(sc + keyOff), (int*)(sc + prefOff));
    HIPC(hipGetLastError());
    ctx->tlasCount = nodeCount; ctx->tlasNeed = std::min(TLAS_STACK_SIZE, std::max(1, n));   // depth unknown on the host: all rows a tree over n leaves can need, up to the limit (beyond it: overflow flag)
    return IDKPT_OK;
}

// ---- the BLAS builder on the device: idkptBuildBlasCore (bvh_gpu.hpp) and idkptBuildBlas (bvh_gpu_full.hpp)
struct BuilderScratch {
    // core (per fragment / per node / per chunk)
    DevBuf fb, ids[3][2], keys[2], vals[2], hist, nodes, act[2], cnt, nodeChunk0, chunkNode, chunkBegin, cboxL, cboxR, carryL, carryR, rc, cbestCost, cbestPos, dec, sideL, sideR, freshOf,
           swapOf, leftCountOf, startOf, countOf, leftTable, pcnt, poff, smallList, aux;
    // whole build
    DevBuf pos, tris, prio, splitCnt, first, scanTmp[3], gboxPart, gbox, origTri, parent, jump[2], dist[2], arrived, need, aggStart, aggCount, bins, maxDepth, used, rank, outNodes, leafCnt, at, uniq, ucount,
           outTris, parents, leafFlag, leaves, sahPart, status;
    // results of the last idkptBuildBlas, for idkptBuildBlasFetch
    int outNodeCount = 0, outTriCount = 0, outParentCount = 0, outLeafCount = 0; bool haveResult = false;
    void release()
    {
        DevBuf* all[] = {&fb, &ids[0][0], &ids[0][1], &ids[1][0], &ids[1][1], &ids[2][0], &ids[2][1], &keys[0], &keys[1], &vals[0], &vals[1], &hist, &nodes, &act[0], &act[1], &cnt, &nodeChunk0, &chunkNode, &chunkBegin,
                         &cboxL, &cboxR, &carryL, &carryR, &rc, &cbestCost, &cbestPos, &dec, &sideL, &sideR, &freshOf, &swapOf, &leftCountOf, &startOf, &countOf, &leftTable, &pcnt, &poff, &smallList, &aux,
                         &pos, &tris, &prio, &splitCnt, &first, &scanTmp[0], &scanTmp[1], &scanTmp[2], &gboxPart, &gbox, &origTri, &parent, &jump[0], &jump[1], &dist[0], &dist[1], &arrived, &need, &aggStart, &aggCount,
                         &bins, &maxDepth, &used, &rank, &outNodes, &leafCnt, &at, &uniq, &ucount, &outTris, &parents, &leafFlag, &leaves, &sahPart, &status};
        for (DevBuf* b : all) b->release();
    }
};
static void builder_scratch_free(dev_ctx* ctx) { if (ctx->bscratch) { ctx->bscratch->release(); delete ctx->bscratch; ctx->bscratch = nullptr; } }
static BuilderScratch& builder_scratch(dev_ctx* ctx) { if (!ctx->bscratch) ctx->bscratch = new BuilderScratch(); return *ctx->bscratch; }

// BLAS.GetBuildData + the SweepSAH recursion over the n fragment boxes in B.fb (device): leaves the node array (2 * n entries, reference id scheme,
// not compacted) in B.nodes and the final x-sorted id order in B.ids[0][*outPP].  Issues on the context's stream; one host sync per level.
static int bvh_core(dev_ctx* ctx, BuilderScratch& B, int n, int* outPP, int* outLevels)
{
    using namespace bvhgpu;
    hipStream_t st = ctx->stream;
    const bool timing = ctx->opt.bvhTiming != 0;   // option "bvh_timing": host-side phase times on stderr
    auto tq = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (!timing) return; (void)hipStreamSynchronize(st); auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[idkpt bvh] %-12s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(t - tq).count()); tq = t; };
    const size_t nodeCount = (size_t)std::max(2 * n, 4);
    const int Cmax = n / CH + n + 2;                             // chunks of a level: at most one per CH positions plus one per active node
    for (int a = 0; a < 3; a++) for (int k = 0; k < 2; k++) HIPC(B.ids[a][k].ensure((size_t)n * 4));
    for (int k = 0; k < 2; k++) { HIPC(B.keys[k].ensure((size_t)n * 4)); HIPC(B.vals[k].ensure((size_t)n * 4)); HIPC(B.act[k].ensure(nodeCount * 4)); }
    const uint32_t nTiles = ((uint32_t)n + SORT_TILE - 1) / SORT_TILE;
    HIPC(B.hist.ensure(((size_t)SORT_RADIX * nTiles + SORT_RADIX) * 4));
    HIPC(B.nodes.ensure(nodeCount * 32)); HIPC(B.cnt.ensure(64));
    HIPC(B.nodeChunk0.ensure((nodeCount + 1) * 4)); HIPC(B.chunkNode.ensure((size_t)Cmax * 4)); HIPC(B.chunkBegin.ensure((size_t)Cmax * 4));
    HIPC(B.cboxL.ensure((size_t)3 * Cmax * sizeof(BBox))); HIPC(B.cboxR.ensure((size_t)3 * Cmax * sizeof(BBox))); HIPC(B.carryL.ensure((size_t)3 * Cmax * sizeof(BBox))); HIPC(B.carryR.ensure((size_t)3 * Cmax * sizeof(BBox)));
    HIPC(B.rc.ensure((size_t)3 * n * 4)); HIPC(B.cbestCost.ensure((size_t)3 * Cmax * 4)); HIPC(B.cbestPos.ensure((size_t)3 * Cmax * 4));
    HIPC(B.dec.ensure(nodeCount * sizeof(Decision))); HIPC(B.sideL.ensure((size_t)Cmax * sizeof(BBox))); HIPC(B.sideR.ensure((size_t)Cmax * sizeof(BBox)));
    HIPC(B.freshOf.ensure(nodeCount * 4)); HIPC(B.swapOf.ensure(nodeCount * 4)); HIPC(B.leftCountOf.ensure(nodeCount * 4)); HIPC(B.startOf.ensure(nodeCount * 4)); HIPC(B.countOf.ensure(nodeCount * 4));
    HIPC(B.leftTable.ensure((size_t)n)); HIPC(B.pcnt.ensure((size_t)3 * Cmax * 4)); HIPC(B.poff.ensure((size_t)3 * Cmax * 4));
    HIPC(B.smallList.ensure(nodeCount * 4)); HIPC(B.aux.ensure((size_t)n * 4));
    const int smallMax = std::max(0, std::min(128, ctx->opt.bvhSmall));   // subtrees of at most this many fragments are finished by one thread each (option "bvh_small")
    lap("alloc");
    // device words: [0] = n (count for the sort kernels), [1] = chunk count of the level, [2] = next level's node count, [3] = small subtrees
    uint32_t* cnt = B.cnt.as<uint32_t>();
    { uint32_t h[4] = {(uint32_t)n, 0u, 0u, 0u}; HIPC(hipMemcpyAsync(cnt, h, 16, hipMemcpyHostToDevice, st)); HIPC(hipStreamSynchronize(st)); }
    // ---- BLAS.GetBuildData: per axis a stable sort of the ids by FloatToKey(min + max) (five 7-bit LSD passes over the 32-bit key)
    uint32_t* digitTotals = B.hist.as<uint32_t>() + (size_t)SORT_RADIX * nTiles;
    for (int axis = 0; axis < 3; axis++) {
        hipLaunchKernelGGL(k_keys, dim3((n + 255) / 256), dim3(256), 0, st, (const float4*)B.fb.as<float4>(), n, axis, B.keys[0].as<uint32_t>(), B.vals[0].as<uint32_t>());
        int cur = 0;
        for (int pass = 0; pass < 5; pass++) {
            hipLaunchKernelGGL(k_sort_hist, dim3(nTiles), dim3(SORT_BLOCK), 0, st, (const uint32_t*)B.keys[cur].as<uint32_t>(), (const uint32_t*)cnt, (uint32_t)(7 * pass), B.hist.as<uint32_t>(), nTiles);
            hipLaunchKernelGGL(k_sort_scan, dim3(SORT_RADIX), dim3(1024), 0, st, (const uint32_t*)cnt, B.hist.as<uint32_t>(), nTiles, digitTotals);
            hipLaunchKernelGGL(k_sort_scatter, dim3(nTiles), dim3(SORT_BLOCK), 0, st, (const uint32_t*)B.keys[cur].as<uint32_t>(), (const uint32_t*)B.vals[cur].as<uint32_t>(), (const uint32_t*)cnt, (uint32_t)(7 * pass),
                               (const uint32_t*)B.hist.as<uint32_t>(), nTiles, (const uint32_t*)digitTotals, B.keys[1 - cur].as<uint32_t>(), B.vals[1 - cur].as<uint32_t>());
            cur = 1 - cur;
        }
        HIPC(hipMemcpyAsync(B.ids[axis][0].p, B.vals[cur].p, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
    }
    lap("sort");
    // ---- the recursion, one level at a time
    HIPC(hipMemsetAsync(B.nodes.p, 0, nodeCount * 32, st));
    { HNodeG root = {}; root.startOrChild = 0; root.count = n; HIPC(hipMemcpyAsync(B.nodes.as<HNodeG>() + 1, &root, 32, hipMemcpyHostToDevice, st));
      int one = 1, two = 2; HIPC(hipMemcpyAsync(B.act[0].p, &one, 4, hipMemcpyHostToDevice, st)); HIPC(hipMemcpyAsync(B.freshOf.as<int>() + 1, &two, 4, hipMemcpyHostToDevice, st)); HIPC(hipStreamSynchronize(st)); }
    HNodeG* nodes = B.nodes.as<HNodeG>();
    int A = 1, curAct = 0, pp = 0, levels = 0;
    if (n <= smallMax) { A = 0; int one = 1; HIPC(hipMemcpyAsync(B.smallList.p, &one, 4, hipMemcpyHostToDevice, st)); HIPC(hipMemcpyAsync(cnt + 3, &one, 4, hipMemcpyHostToDevice, st)); HIPC(hipStreamSynchronize(st)); }
    while (A > 0) {
        Level L; L.act = B.act[curAct].as<int>(); L.A = A; L.nodeChunk0 = B.nodeChunk0.as<int>(); L.chunkNode = B.chunkNode.as<int>(); L.chunkBegin = B.chunkBegin.as<int>(); L.chunkCount = (int*)(cnt + 1);
        const int Cl = std::min(Cmax, n / CH + A + 1);            // grid bound for this level's chunks (workgroups beyond the real count exit)
        const int gA = (A + 255) / 256;
        const int* i0 = B.ids[0][pp].as<int>(); const int* i1 = B.ids[1][pp].as<int>(); const int* i2 = B.ids[2][pp].as<int>();
        int* o0 = B.ids[0][1 - pp].as<int>(); int* o1 = B.ids[1][1 - pp].as<int>(); int* o2 = B.ids[2][1 - pp].as<int>();
        const float4* fb = B.fb.as<float4>();
        HIPC(hipMemsetAsync(cnt + 2, 0, 4, st));
        hipLaunchKernelGGL(k_chunks, dim3(1), dim3(1024), 0, st, (const HNodeG*)nodes, L);
        hipLaunchKernelGGL(k_snapshot_ranges, dim3(gA), dim3(256), 0, st, (const HNodeG*)nodes, L, B.startOf.as<int>(), B.countOf.as<int>());
        hipLaunchKernelGGL(k_chunk_box, dim3(Cl, 3), dim3(CH), 0, st, (const HNodeG*)nodes, L, fb, i0, i1, i2, B.cboxL.as<BBox>(), B.cboxR.as<BBox>(), Cmax);
        hipLaunchKernelGGL(k_node_carry, dim3(A), dim3(CH), 0, st, nodes, L, (const BBox*)B.cboxL.as<BBox>(), (const BBox*)B.cboxR.as<BBox>(), B.carryL.as<BBox>(), B.carryR.as<BBox>(), Cmax);
        hipLaunchKernelGGL(k_chunk_rc, dim3(Cl, 3), dim3(CH), 0, st, (const HNodeG*)nodes, L, fb, i0, i1, i2, (const BBox*)B.carryR.as<BBox>(), B.rc.as<float>(), n, Cmax);
        hipLaunchKernelGGL(k_chunk_cost, dim3(Cl, 3), dim3(CH), 0, st, (const HNodeG*)nodes, L, fb, i0, i1, i2, (const BBox*)B.carryL.as<BBox>(), (const float*)B.rc.as<float>(), B.cbestCost.as<float>(), B.cbestPos.as<int>(), n, Cmax);
        hipLaunchKernelGGL(k_node_decide, dim3(A), dim3(CH), 0, st, (const HNodeG*)nodes, L, (const float*)B.cbestCost.as<float>(), (const int*)B.cbestPos.as<int>(), B.dec.as<Decision>(), Cmax);
        hipLaunchKernelGGL(k_chunk_sides, dim3(Cl), dim3(CH), 0, st, (const HNodeG*)nodes, L, (const Decision*)B.dec.as<Decision>(), fb, i0, i1, i2, B.sideL.as<BBox>(), B.sideR.as<BBox>());
        hipLaunchKernelGGL(k_node_finalize, dim3(A), dim3(CH), 0, st, nodes, L, B.dec.as<Decision>(), (const BBox*)B.sideL.as<BBox>(), (const BBox*)B.sideR.as<BBox>(), B.freshOf.as<int>(), B.swapOf.as<int>(), B.leftCountOf.as<int>(),
                           B.act[1 - curAct].as<int>(), (int*)(cnt + 2), B.smallList.as<int>(), (int*)(cnt + 3), smallMax);
        hipLaunchKernelGGL(k_mark, dim3(Cl), dim3(CH), 0, st, (const HNodeG*)nodes, L, (const Decision*)B.dec.as<Decision>(), (const int*)B.swapOf.as<int>(), (const int*)B.startOf.as<int>(), (const int*)B.countOf.as<int>(), i0, i1, i2, B.leftTable.as<uint8_t>());
        for (int a = 0; a < 3; a++) HIPC(hipMemcpyAsync(B.ids[a][1 - pp].p, B.ids[a][pp].p, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_part_count, dim3(Cl, 3), dim3(CH), 0, st, L, (const Decision*)B.dec.as<Decision>(), (const int*)B.startOf.as<int>(), (const int*)B.countOf.as<int>(), i0, i1, i2, (const uint8_t*)B.leftTable.as<uint8_t>(), B.pcnt.as<int>(), Cmax);
        hipLaunchKernelGGL(k_part_offsets, dim3(A), dim3(CH), 0, st, L, (const Decision*)B.dec.as<Decision>(), (const int*)B.pcnt.as<int>(), B.poff.as<int>(), Cmax);
        hipLaunchKernelGGL(k_part_scatter, dim3(Cl, 3), dim3(CH), 0, st, L, (const Decision*)B.dec.as<Decision>(), (const int*)B.startOf.as<int>(), (const int*)B.countOf.as<int>(), (const int*)B.leftCountOf.as<int>(), i0, i1, i2, o0, o1, o2,
                           (const uint8_t*)B.leftTable.as<uint8_t>(), (const int*)B.poff.as<int>(), Cmax);
        HIPC(hipGetLastError());
        uint32_t next = 0;
        HIPC(hipMemcpyAsync(&next, cnt + 2, 4, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st));
        A = (int)next; curAct = 1 - curAct; pp = 1 - pp; levels++;
        if (levels > 4096) return fail(ctx, IDKPT_ERR_UNKNOWN, "idkptBuildBlasCore: recursion does not terminate");
    }
    lap("levels");
    hipLaunchKernelGGL(k_small_subtrees, dim3((unsigned)((nodeCount + 63) / 64)), dim3(64), 0, st, nodes, (const int*)B.smallList.as<int>(), (const int*)(cnt + 3), (const float4*)B.fb.as<float4>(), B.ids[0][pp].as<int>(), B.ids[1][pp].as<int>(), B.ids[2][pp].as<int>(),
                       B.rc.as<float>(), B.aux.as<int>(), B.leftTable.as<uint8_t>(), (const int*)B.freshOf.as<int>());
    HIPC(hipGetLastError());
    lap("subtrees");
    *outPP = pp; if (outLevels) *outLevels = levels;
    return IDKPT_OK;
}

// idkptBuildBlasCore: fragment boxes in (host), node array + final x-sorted id order out (host).  Stateless apart from the device, the stream and
// the cached scratch buffers of the context: no scene is needed and none is touched.
static int32_t dev_BuildBlasCore(dev_ctx* ctx, const float* fragBoxes, int32_t n, GpuBlasNode* outNodes, int32_t* outSortedX, int32_t* outLevels)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(fragBoxes && outNodes && outSortedX && n >= 1, "idkptBuildBlasCore: null argument or no fragments");
    REQUIRE(n <= (1 << 27), "idkptBuildBlasCore: too many fragments");
    HIPC(hipSetDevice(ctx->device));
    BuilderScratch& B = builder_scratch(ctx);
    B.haveResult = false;
    hipStream_t st = ctx->stream;
    HIPC(B.fb.ensure((size_t)n * 32));
    HIPC(hipMemcpyAsync(B.fb.p, fragBoxes, (size_t)n * 32, hipMemcpyHostToDevice, st));
    int pp = 0;
    int rc = bvh_core(ctx, B, n, &pp, outLevels); if (rc) return rc;
    HIPC(hipMemcpyAsync(outNodes, B.nodes.p, (size_t)std::max(2 * n, 4) * 32, hipMemcpyDeviceToHost, st));
    HIPC(hipMemcpyAsync(outSortedX, B.ids[0][pp].p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    return IDKPT_OK;
}

// exclusive scan of n uint32 on the device (bvh_gpu_full.hpp k_scan_*); in and out may alias; tmp[level] hold the block totals
static int scan_u32(dev_ctx* ctx, BuilderScratch& B, const uint32_t* in, uint32_t* out, uint32_t n, int level = 0)
{
    using namespace bvhgpu;
    if (n == 0) return IDKPT_OK;
    if (level > 2) return fail(ctx, IDKPT_ERR_UNKNOWN, "scan_u32: too many elements");
    const uint32_t per = SCAN_BLOCK * SCAN_ITEMS, blocks = (n + per - 1) / per;
    HIPC(B.scanTmp[level].ensure((size_t)blocks * 4 + 16));
    uint32_t* tot = B.scanTmp[level].as<uint32_t>();
    hipLaunchKernelGGL(k_scan_block, dim3(blocks), dim3(SCAN_BLOCK), 0, ctx->stream, in, out, n, tot);
    if (blocks > 1) {
        int rc = scan_u32(ctx, B, tot, tot, blocks, level + 1); if (rc) return rc;
        hipLaunchKernelGGL(k_scan_add, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, out, n, (const uint32_t*)tot);
    }
    HIPC(hipGetLastError());
    return IDKPT_OK;
}

// OptimizeStackSize in the reference's own order (BLAS.cs:875-936) on a host copy of the uncompacted tree: the fallback of dev_BuildBlas for a
// decision its error bound cannot settle (and for trees deeper than its per-depth table).  Serial, recursive, binary64 like the reference.
namespace stackopt_host {
struct HN { float mn[3]; int32_t startOrChild; float mx[3]; int32_t count; };
static inline bool leaf(const HN& n) { return n.count > 0; }
static inline float half_area(const HN& n) { float x = n.mx[0] - n.mn[0], y = n.mx[1] - n.mn[1], z = n.mx[2] - n.mn[2]; return fmaf(x + y, z, x * y); }
static double global_sah(const std::vector<HN>& nodes)
{
    double cost = 0.0; const double rootArea = 1.0 / (double)half_area(nodes[1]);
    std::vector<int> st; st.push_back(1);
    while (!st.empty()) {
        const HN& n = nodes[st.back()]; st.pop_back();
        const double prob = (double)half_area(n) * rootArea;
        if (leaf(n)) cost += (double)(1.1f * (float)n.count) * prob;
        else { cost += (double)1.0f * prob; st.push_back(n.startOrChild + 1); st.push_back(n.startOrChild); }
    }
    return cost;
}
static void collapse(std::vector<HN>& nodes, int newStackSize, bool firstPass, double& nextCost, double rootHalfArea)
{
    // post-order without recursion: (node, depth, state)
    struct F { int p, depth, state; };
    std::vector<F> st; st.push_back({1, 0, 0});
    while (!st.empty()) {
        F& f = st.back();
        const int c = nodes[f.p].startOrChild;
        if (f.state == 0) { f.state = 1; if (!leaf(nodes[c])) { st.push_back({c, f.depth + 1, 0}); continue; } }
        if (f.state == 1) { f.state = 2; if (!leaf(nodes[c + 1])) { st.push_back({c + 1, f.depth + 1, 0}); continue; } }
        HN& p = nodes[f.p]; const HN l = nodes[c], r = nodes[c + 1]; const int depth = f.depth;
        if (leaf(l) && leaf(r)) {
            if (depth > newStackSize && !firstPass) { p.startOrChild = l.startOrChild; p.count = l.count + r.count; }
            if ((depth == newStackSize && !firstPass) || (depth > newStackSize && firstPass)) {
                const double leavesCost = (double)1.1f * ((double)l.count * (double)half_area(l) + (double)r.count * (double)half_area(r));
                const double newParentLeafCost = (double)1.1f * (double)(l.count + r.count);
                nextCost += ((double)half_area(nodes[f.p]) * (newParentLeafCost - (double)1.0f) - leavesCost) / rootHalfArea;
            }
        }
        st.pop_back();
    }
}
// returns the final RequiredStackSize; `nodes` is modified like the reference modifies its array
static int optimize(std::vector<HN>& nodes, int requiredStack)
{
    if (requiredStack < 16) return requiredStack;
    const double current = global_sah(nodes); double added = 0.0;
    const double rootHalfArea = (double)half_area(nodes[1]);
    collapse(nodes, requiredStack - 1, true, added, rootHalfArea);
    double inc = added / current;
    while (inc <= (double)0.0009745f && requiredStack > 0) { collapse(nodes, --requiredStack, false, added, rootHalfArea); inc = added / current; }
    return requiredStack;
}
}

// idkptBuildBlas: the whole BLAS build of one geometry on the device (bvh_gpu_full.hpp); results stay on the device until idkptBuildBlasFetch.
// Byte-identical to idkbvhBuildBlas (libidkbvh.so): nodes, triangles, parent / leaf indices, RequiredStackSize (tests/test_gpu_builder.py).
static int32_t dev_BuildBlas(dev_ctx* ctx, const float* positions, int32_t vertexCount, const GpuBlasTriangle* tris, int32_t triCount, int32_t isRefittable, float preSplitFactor, idkpt_blas_build_info* info)
{
    using namespace bvhgpu;
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(positions && tris && info && vertexCount > 0 && triCount > 0, "idkptBuildBlas: null argument or empty geometry");
    REQUIRE(triCount <= (1 << 26), "idkptBuildBlas: too many triangles");
    // One walk over the triangles: index range, the vertex range they reference, and — like idkbvhBuildBlas (bvh_builder.cpp), which this call mirrors — finiteness
    // of the REFERENCED positions only (the reference's builder has no defined result for NaN boxes / integer conversions of NaN; vertices no triangle of this BLAS
    // uses belong to other meshes of the host's global vertex array and are none of this call's business).
    uint32_t vmin = 0xffffffffu, vmax = 0u;
    {
        const uint32_t* pb = reinterpret_cast<const uint32_t*>(positions); uint32_t bad = 0;
        for (int i = 0; i < triCount; i++) {
            const uint32_t v[3] = {tris[i].X, tris[i].Y, tris[i].Z};
            REQUIRE(v[0] < (uint32_t)vertexCount && v[1] < (uint32_t)vertexCount && v[2] < (uint32_t)vertexCount, "idkptBuildBlas: triangle index out of range");
            for (int k = 0; k < 3; k++) {
                vmin = std::min(vmin, v[k]); vmax = std::max(vmax, v[k]);
                const uint32_t* q = pb + 3 * (size_t)v[k];
                bad |= (uint32_t)((q[0] & 0x7f800000u) == 0x7f800000u) | (uint32_t)((q[1] & 0x7f800000u) == 0x7f800000u) | (uint32_t)((q[2] & 0x7f800000u) == 0x7f800000u);
            }
        }
        REQUIRE(!bad, "idkptBuildBlas: a vertex position is not finite");
    }
    HIPC(hipSetDevice(ctx->device));
    BuilderScratch& B = builder_scratch(ctx);
    B.haveResult = false;
    hipStream_t st = ctx->stream;
    const auto t0 = std::chrono::steady_clock::now();
    const bool timing = ctx->opt.bvhTiming != 0;
    auto tq = t0;
    auto lap = [&](const char* what) { if (!timing) return; (void)hipStreamSynchronize(st); auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[idkpt blas] %-14s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(t - tq).count()); tq = t; };
    const int nT = triCount;
    const unsigned gT = (unsigned)((nT + 255) / 256);
    // only the vertex range this BLAS references crosses PCIe (a host that passes its GLOBAL vertex array for every BLAS pays for its own vertices, not for
    // O(#BLAS x V)); the kernels keep indexing with the host's ids through a pointer moved back by the range's first vertex
    const size_t vRange = (size_t)vmax - vmin + 1;
    HIPC(B.pos.ensure(vRange * 12)); HIPC(B.tris.ensure((size_t)nT * 16));
    HIPC(hipMemcpyAsync(B.pos.p, positions + 3 * (size_t)vmin, vRange * 12, hipMemcpyHostToDevice, st));
    HIPC(hipMemcpyAsync(B.tris.p, tris, (size_t)nT * 16, hipMemcpyHostToDevice, st));
    const float* dPos = B.pos.as<float>() - 3 * (ptrdiff_t)vmin; const uint4* dTris = B.tris.as<uint4>();
    lap("upload");
    // ---- fragments
    int F = nT;
    if (isRefittable) {
        HIPC(B.fb.ensure((size_t)nT * 32));
        hipLaunchKernelGGL(k_tri_boxes, dim3(gT), dim3(256), 0, st, dPos, dTris, nT, B.fb.as<float4>());
    } else {
        HIPC(B.prio.ensure((size_t)nT * 4)); HIPC(B.splitCnt.ensure(((size_t)nT + 1) * 4)); HIPC(B.first.ensure(((size_t)nT + 1) * 4));
        hipLaunchKernelGGL(k_tri_prio, dim3(gT), dim3(256), 0, st, dPos, dTris, nT, B.prio.as<float>());
        // the binary32 running sum of the priorities in index order (PreSplitting.cs:32-37): one dependent chain -> on the host
        std::vector<float> hp((size_t)nT);
        HIPC(hipMemcpyAsync(hp.data(), B.prio.p, (size_t)nT * 4, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st));
        float total = 0.0f;
        for (int i = 0; i < nT; i++) total += hp[i];
        HIPC(B.status.ensure(16)); HIPC(hipMemsetAsync(B.status.p, 0, 16, st));      // [0..7] exact fragment count, [8..11] PreSplit stack flag
        hipLaunchKernelGGL(k_split_count, dim3(gT), dim3(256), 0, st, (const float*)B.prio.as<float>(), total, nT, preSplitFactor, B.splitCnt.as<uint32_t>(), B.status.as<unsigned long long>());
        HIPC(hipMemsetAsync(B.splitCnt.as<uint32_t>() + nT, 0, 4, st));
        { int rc = scan_u32(ctx, B, B.splitCnt.as<uint32_t>(), B.first.as<uint32_t>(), (uint32_t)nT + 1u); if (rc) return rc; }
        uint32_t hF = 0; unsigned long long hSum = 0;
        HIPC(hipMemcpyAsync(&hF, B.first.as<uint32_t>() + nT, 4, hipMemcpyDeviceToHost, st)); HIPC(hipMemcpyAsync(&hSum, B.status.p, 8, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st));
        REQUIRE(hSum <= (1ull << 27), "idkptBuildBlas: PreSplit asks for more than 2^27 fragments (split factor / priorities)");
        REQUIRE(hF == (uint32_t)hSum && hF >= (uint32_t)nT, "idkptBuildBlas: PreSplit produced an implausible fragment count");
        F = (int)hF;
        const int parts = (nT + CH - 1) / CH;
        HIPC(B.gboxPart.ensure((size_t)parts * sizeof(BBox))); HIPC(B.gbox.ensure(sizeof(BBox)));
        hipLaunchKernelGGL(k_global_box_partial, dim3(parts), dim3(CH), 0, st, dPos, dTris, nT, B.gboxPart.as<BBox>());
        hipLaunchKernelGGL(k_global_box_final, dim3(1), dim3(CH), 0, st, (const BBox*)B.gboxPart.as<BBox>(), parts, B.gbox.as<BBox>());
        HIPC(B.fb.ensure((size_t)F * 32)); HIPC(B.origTri.ensure((size_t)F * 4));
        hipLaunchKernelGGL(k_presplit, dim3((unsigned)((nT + 63) / 64)), dim3(64), 0, st, dPos, dTris, nT, (const uint32_t*)B.splitCnt.as<uint32_t>(), (const uint32_t*)B.first.as<uint32_t>(), (const BBox*)B.gbox.as<BBox>(), B.fb.as<float4>(), B.origTri.as<int>(), B.status.as<uint32_t>() + 2);
        uint32_t hOvf = 0;
        HIPC(hipMemcpyAsync(&hOvf, B.status.as<uint32_t>() + 2, 4, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st));
        REQUIRE(!hOvf, "idkptBuildBlas: a triangle's PreSplit recursion needs more than 64 stack entries (the reference throws here, PreSplitting.cs:57)");
    }
    HIPC(hipGetLastError());
    lap("fragments");
    // ---- SweepSAH core
    int pp = 0, levels = 0;
    { int rc = bvh_core(ctx, B, F, &pp, &levels); if (rc) return rc; }
    lap("core");
    // ---- tail
    const int nodeCount = std::max(2 * F, 4);
    const unsigned gN = (unsigned)((nodeCount + 255) / 256);
    HNodeG* nodes = B.nodes.as<HNodeG>();
    const int* sorted0 = B.ids[0][pp].as<int>();
    HIPC(B.parent.ensure((size_t)nodeCount * 4)); HIPC(B.arrived.ensure((size_t)nodeCount * 4)); HIPC(B.need.ensure((size_t)nodeCount * 4)); HIPC(B.aggStart.ensure((size_t)nodeCount * 4)); HIPC(B.aggCount.ensure((size_t)nodeCount * 4));
    for (int k = 0; k < 2; k++) { HIPC(B.jump[k].ensure((size_t)nodeCount * 4)); HIPC(B.dist[k].ensure((size_t)nodeCount * 4)); }
    hipLaunchKernelGGL(k_fix_root, dim3(1), dim3(64), 0, st, nodes);
    hipLaunchKernelGGL(k_tree_init, dim3(gN), dim3(256), 0, st, (const HNodeG*)nodes, nodeCount, B.parent.as<int>(), B.jump[0].as<int>(), B.dist[0].as<int>(), B.arrived.as<int>());
    // depths by pointer jumping (after r rounds dist = min(depth, 2^r)), then the bottom-up pass: required stack rows + fragment range of every subtree
    int curJ = 0;
    for (int span = 1; span < nodeCount; span <<= 1) {
        hipLaunchKernelGGL(k_depth_jump, dim3(gN), dim3(256), 0, st, (const HNodeG*)nodes, nodeCount, (const int*)B.jump[curJ].as<int>(), (const int*)B.dist[curJ].as<int>(), B.jump[1 - curJ].as<int>(), B.dist[1 - curJ].as<int>());
        curJ = 1 - curJ;
    }
    const int* depthAll = B.dist[curJ].as<int>();
    HIPC(B.maxDepth.ensure(16)); HIPC(hipMemsetAsync(B.maxDepth.p, 0, 16, st));
    hipLaunchKernelGGL(k_leaf_init, dim3(gN), dim3(256), 0, st, (const HNodeG*)nodes, nodeCount, depthAll, B.need.as<int>(), B.aggStart.as<int>(), B.aggCount.as<int>(), B.maxDepth.as<int>());
    int treeDepth = 0;
    HIPC(hipMemcpyAsync(&treeDepth, B.maxDepth.p, 4, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st));
    if (treeDepth <= 512) {
        for (int d = treeDepth - 1; d >= 0; d--)
            hipLaunchKernelGGL(k_level_up, dim3(gN), dim3(256), 0, st, (const HNodeG*)nodes, nodeCount, depthAll, d, B.need.as<int>(), B.aggStart.as<int>(), B.aggCount.as<int>());
    } else   // a chain-like tree (identical fragments): one launch per level would be one per fragment -> the fenced climb
        hipLaunchKernelGGL(k_climb, dim3(gN), dim3(256), 0, st, (const HNodeG*)nodes, nodeCount, (const int*)B.parent.as<int>(), B.arrived.as<int>(), B.need.as<int>(), B.aggStart.as<int>(), B.aggCount.as<int>());
    int hNeed = 0;
    HIPC(hipMemcpyAsync(&hNeed, B.need.as<int>() + 1, 4, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st));
    int requiredStack = hNeed;
    lap("bottom-up");
    int S = 0x7fffffff;                                              // depth threshold of the final collapse (none)
    const int* depth = nullptr;
    if (requiredStack >= 16) {                                       // StackOptThreshold (BLAS.cs:41, :880)
        depth = depthAll;
        const size_t nBins = 4 + 3 * (size_t)OPT_MAX_DEPTH;
        HIPC(B.bins.ensure(nBins * 8));
        HIPC(hipMemsetAsync(B.bins.p, 0, nBins * 8, st)); HIPC(hipMemsetAsync(B.maxDepth.p, 0, 16, st));
        hipLaunchKernelGGL(k_opt_sums, dim3(gN), dim3(256), 0, st, (const HNodeG*)nodes, nodeCount, depth, (const int*)B.aggCount.as<int>(), requiredStack, B.bins.as<double>(), B.maxDepth.as<int>());
        std::vector<double> bins(nBins); int maxDepth = 0;
        HIPC(hipMemcpyAsync(bins.data(), B.bins.p, nBins * 8, hipMemcpyDeviceToHost, st)); HIPC(hipMemcpyAsync(&maxDepth, B.maxDepth.p, 4, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st));
        lap("opt sums");
        // replay of the loop of BLAS.cs:882-894 on the sums.  The reference adds the same terms one by one in tree order; two binary64 summation
        // orders of N terms differ by at most ~2 N u sum|t| (u = 2^-53), so `inc <= acceptance` is decided here only outside that margin (x 8).
        const double accept = (double)0.0009745f, u = 1.1102230246251565e-16;
        bool certain = maxDepth < OPT_MAX_DEPTH && requiredStack - 1 < OPT_MAX_DEPTH && !ctx->opt.bvhStackOptHost;
        const double current = bins[0]; double added = bins[1], addedAbs = bins[2], nTerms = (double)nodeCount;
        int rs = requiredStack, sLast = 0x7fffffff;
        auto decide = [&](bool& le) {
            const double inc = added / current;
            const double margin = 8.0 * (2.0 * nTerms * u * addedAbs / current + 2.0 * nTerms * u * std::fabs(inc) + 4.0 * u * std::fabs(inc));
            if (std::fabs(inc - accept) <= margin || !(current > 0.0)) return false;
            le = inc <= accept; return true;
        };
        while (certain) {
            bool le = false;
            if (!decide(le)) { certain = false; break; }
            if (!(le && rs > 0)) break;
            rs--; sLast = rs;
            added += bins[4 + 3 * (size_t)rs]; addedAbs += bins[4 + 3 * (size_t)rs + 1];
        }
        if (certain) {
            requiredStack = rs; S = sLast;
            if (S != 0x7fffffff) hipLaunchKernelGGL(k_collapse, dim3(gN), dim3(256), 0, st, nodes, nodeCount, depth, (const int*)B.aggStart.as<int>(), (const int*)B.aggCount.as<int>(), S);
        } else {
            // never observed: a decision inside the rounding margin (or a tree deeper than the table) -> the reference's own walk on a host copy
            std::vector<stackopt_host::HN> hn((size_t)nodeCount);
            HIPC(hipMemcpyAsync(hn.data(), nodes, (size_t)nodeCount * 32, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st));
            requiredStack = stackopt_host::optimize(hn, requiredStack);
            HIPC(hipMemcpyAsync(nodes, hn.data(), (size_t)nodeCount * 32, hipMemcpyHostToDevice, st)); HIPC(hipStreamSynchronize(st));
            depth = nullptr; S = 0x7fffffff;                         // the host walk turned every collapsed node into a leaf: "internal" means live again
        }
    }
    lap("stack-opt");
    // RemoveEmptySubtrees: used child pairs in id order = the reference's pre-order numbering
    const int pairCount = nodeCount / 2;
    HIPC(B.used.ensure(((size_t)pairCount + 1) * 4)); HIPC(B.rank.ensure(((size_t)pairCount + 1) * 4));
    HIPC(hipMemsetAsync(B.used.p, 0, ((size_t)pairCount + 1) * 4, st));
    hipLaunchKernelGGL(k_mark_pairs, dim3(gN), dim3(256), 0, st, (const HNodeG*)nodes, nodeCount, depthAll, depth ? S : 0x7fffffff, B.used.as<uint32_t>());   // (no collapse on the device: every internal node is live)
    { int rc = scan_u32(ctx, B, B.used.as<uint32_t>(), B.rank.as<uint32_t>(), (uint32_t)pairCount + 1u); if (rc) return rc; }
    uint32_t usedPairs = 0;
    HIPC(hipMemcpyAsync(&usedPairs, B.rank.as<uint32_t>() + pairCount, 4, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st));
    const int outNodeCount = 2 + 2 * (int)usedPairs;
    HIPC(B.outNodes.ensure((size_t)outNodeCount * 32));
    hipLaunchKernelGGL(k_compact_nodes, dim3((unsigned)((pairCount + 255) / 256)), dim3(256), 0, st, (const HNodeG*)nodes, pairCount, (const uint32_t*)B.used.as<uint32_t>(), (const uint32_t*)B.rank.as<uint32_t>(), B.outNodes.as<HNodeG>());
    HNodeG* on = B.outNodes.as<HNodeG>();
    const unsigned gO = (unsigned)((outNodeCount + 255) / 256);
    lap("compact");
    // un-indexing
    int outTriCount = 0;
    if (isRefittable) {
        HIPC(B.leafCnt.ensure(((size_t)outNodeCount + 1) * 4)); HIPC(B.at.ensure(((size_t)outNodeCount + 1) * 4));
        hipLaunchKernelGGL(k_leaf_counts, dim3((unsigned)((outNodeCount + 1 + 255) / 256)), dim3(256), 0, st, (const HNodeG*)on, outNodeCount, B.leafCnt.as<uint32_t>());
        HIPC(hipMemsetAsync(B.leafCnt.as<uint32_t>() + outNodeCount, 0, 4, st));
        { int rc = scan_u32(ctx, B, B.leafCnt.as<uint32_t>(), B.at.as<uint32_t>(), (uint32_t)outNodeCount + 1u); if (rc) return rc; }
        uint32_t tot = 0; HIPC(hipMemcpyAsync(&tot, B.at.as<uint32_t>() + outNodeCount, 4, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st));
        outTriCount = (int)tot;
        HIPC(B.outTris.ensure(std::max<size_t>((size_t)outTriCount * 16, 16)));
        hipLaunchKernelGGL(k_unindex_plain, dim3(gO), dim3(256), 0, st, on, outNodeCount, (const uint32_t*)B.at.as<uint32_t>(), sorted0, dTris, B.outTris.as<uint4>());
    } else {
        const int pairs = (outNodeCount - 2) / 2;
        HIPC(B.uniq.ensure((size_t)2 * F * 4)); HIPC(B.ucount.ensure((size_t)outNodeCount * 4)); HIPC(B.leafCnt.ensure(((size_t)pairs + 1) * 4)); HIPC(B.at.ensure(((size_t)pairs + 1) * 4));
        HIPC(hipMemsetAsync(B.ucount.p, 0, (size_t)outNodeCount * 4, st));
        const unsigned gP = (unsigned)((pairs + 63) / 64);
        hipLaunchKernelGGL(k_unindex_ps_count, dim3(gP), dim3(64), 0, st, (const HNodeG*)on, pairs, sorted0, (const int*)B.origTri.as<int>(), F, B.uniq.as<int>(), B.ucount.as<int>(), B.leafCnt.as<uint32_t>());
        HIPC(hipMemsetAsync(B.leafCnt.as<uint32_t>() + pairs, 0, 4, st));
        { int rc = scan_u32(ctx, B, B.leafCnt.as<uint32_t>(), B.at.as<uint32_t>(), (uint32_t)pairs + 1u); if (rc) return rc; }
        uint32_t tot = 0; HIPC(hipMemcpyAsync(&tot, B.at.as<uint32_t>() + pairs, 4, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st));
        outTriCount = (int)tot;
        HIPC(B.outTris.ensure(std::max<size_t>((size_t)outTriCount * 16, 16)));
        hipLaunchKernelGGL(k_unindex_ps_write, dim3(gP), dim3(64), 0, st, on, pairs, F, (const int*)B.uniq.as<int>(), (const int*)B.ucount.as<int>(), (const uint32_t*)B.at.as<uint32_t>(), dTris, B.outTris.as<uint4>());
    }
    lap("unindex");
    // parent / leaf indices (refittable BLASes only, BVH.cs:357-358)
    int leafCount = 0, parentCount = 0;
    if (isRefittable) {
        parentCount = outNodeCount;
        HIPC(B.parents.ensure((size_t)outNodeCount * 4)); HIPC(B.leafFlag.ensure(((size_t)outNodeCount + 1) * 4)); HIPC(B.rank.ensure(((size_t)outNodeCount + 1) * 4));
        hipLaunchKernelGGL(k_parents, dim3(gO), dim3(256), 0, st, (const HNodeG*)on, outNodeCount, B.parents.as<int>());
        hipLaunchKernelGGL(k_leaf_flags, dim3(gO), dim3(256), 0, st, (const HNodeG*)on, outNodeCount, B.leafFlag.as<uint32_t>());
        HIPC(hipMemsetAsync(B.leafFlag.as<uint32_t>() + outNodeCount, 0, 4, st));
        { int rc = scan_u32(ctx, B, B.leafFlag.as<uint32_t>(), B.rank.as<uint32_t>(), (uint32_t)outNodeCount + 1u); if (rc) return rc; }
        uint32_t tot = 0; HIPC(hipMemcpyAsync(&tot, B.rank.as<uint32_t>() + outNodeCount, 4, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st));
        leafCount = (int)tot;
        HIPC(B.leaves.ensure(std::max<size_t>((size_t)leafCount * 4, 16)));
        hipLaunchKernelGGL(k_leaf_list, dim3(gO), dim3(256), 0, st, (const uint32_t*)B.leafFlag.as<uint32_t>(), (const uint32_t*)B.rank.as<uint32_t>(), outNodeCount, B.leaves.as<int>());
    }
    // ComputeGlobalSAH of the finished tree
    const int sahBlocks = (int)gO;
    HIPC(B.sahPart.ensure((size_t)sahBlocks * 8));
    hipLaunchKernelGGL(k_sah_partial, dim3(gO), dim3(256), 0, st, (const HNodeG*)on, outNodeCount, B.sahPart.as<double>());
    HIPC(hipGetLastError());
    std::vector<double> sp((size_t)sahBlocks);
    HIPC(hipMemcpyAsync(sp.data(), B.sahPart.p, (size_t)sahBlocks * 8, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st));
    double sah = 0.0; for (double v : sp) sah += v;
    lap("indices+sah");
    B.outNodeCount = outNodeCount; B.outTriCount = outTriCount; B.outParentCount = parentCount; B.outLeafCount = leafCount; B.haveResult = true;
    info->NodeCount = outNodeCount; info->TriangleCount = outTriCount; info->RequiredStackSize = requiredStack; info->ParentIndexCount = parentCount; info->LeafIndexCount = leafCount;
    info->FragmentCount = F; info->Levels = levels; info->Sah = sah;
    info->BuildMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return IDKPT_OK;
}

static int32_t dev_BuildBlasFetch(dev_ctx* ctx, GpuBlasNode* nodes, GpuBlasTriangle* triangles, int32_t* parents, int32_t* leaves)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    if (!ctx->bscratch || !ctx->bscratch->haveResult) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptBuildBlasFetch: no finished idkptBuildBlas on this context");
    HIPC(hipSetDevice(ctx->device));
    BuilderScratch& B = *ctx->bscratch;
    hipStream_t st = ctx->stream;
    if (nodes) HIPC(hipMemcpyAsync(nodes, B.outNodes.p, (size_t)B.outNodeCount * 32, hipMemcpyDeviceToHost, st));
    if (triangles && B.outTriCount) HIPC(hipMemcpyAsync(triangles, B.outTris.p, (size_t)B.outTriCount * 16, hipMemcpyDeviceToHost, st));
    if (parents && B.outParentCount) HIPC(hipMemcpyAsync(parents, B.parents.p, (size_t)B.outParentCount * 4, hipMemcpyDeviceToHost, st));
    if (leaves && B.outLeafCount) HIPC(hipMemcpyAsync(leaves, B.leaves.p, (size_t)B.outLeafCount * 4, hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    return IDKPT_OK;
}

// developer / test hook: the device cbrtf of the PreSplit priorities on an array of inputs (compared with the host's cbrtf by the tests)
static int32_t dev_CbrtProbe(dev_ctx* ctx, const float* in, float* out, int32_t n)
{
    if (!ctx || !in || !out || n <= 0) return IDKPT_ERR_INVALID_ARGUMENT;
    HIPC(hipSetDevice(ctx->device));
    BuilderScratch& B = builder_scratch(ctx);
    HIPC(B.prio.ensure((size_t)n * 4)); HIPC(B.splitCnt.ensure((size_t)n * 4));
    HIPC(hipMemcpyAsync(B.prio.p, in, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(bvhgpu::k_cbrt_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const float*)B.prio.as<float>(), B.splitCnt.as<float>(), n);
    HIPC(hipGetLastError());
    HIPC(hipMemcpyAsync(out, B.splitCnt.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
    return IDKPT_OK;
}

static DScene make_dscene(dev_ctx* ctx);
// frame constants for the ray-query / shadow kernels: only the traversal-related fields are read
static int query_frame(dev_ctx* ctx, Frame& f, size_t& ldsBytes, uint32_t& grid)
{
    memset(&f, 0, sizeof(f));
    f.g = ctx->st.Gpu; f.useTlas = ctx->st.UseTlas;
    f.stackCap = std::max(1, ctx->st.BlasStackSize > 0 ? ctx->st.BlasStackSize : ctx->sceneStack);
    f.tlasCap = std::min(TLAS_STACK_SIZE, std::max(1, ctx->tlasNeed));
    ldsBytes = (size_t)(f.stackCap + 2 + (f.useTlas ? f.tlasCap : 0)) * WAVE * 4;   // + the dummy and the spare row of k_trace2's stack (kernels_trace.hpp)
    if (ldsBytes > 64 * 1024) return fail(ctx, IDKPT_ERR_INVALID_ARGUMENT, "BlasStackSize too large for the LDS traversal stack");
    int wavesPerCU = (int)std::min<size_t>(32, (160 * 1024) / std::max<size_t>(ldsBytes, 1));
    grid = (uint32_t)(ctx->numCUs * std::max(1, wavesPerCU));
    return IDKPT_OK;
}

// issue only (H2D, kernel, D2H on the context's stream); the caller synchronises.  hits must stay valid until then.
static int32_t dev_TraceRaysIssue(dev_ctx* ctx, const idkpt_ray* rays, size_t count, uint32_t flags, idkpt_hit* hits, bool devicePtrs = false /* rays / hits live on this context's device: no copies, nothing to wait for */)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    if (!ctx->haveScene) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptTraceRays: no scene uploaded");
    REQUIRE(count == 0 || (rays && hits), "idkptTraceRays: null rays/hits");
    REQUIRE(count < (1ull << 31), "idkptTraceRays: too many rays in one call");
    REQUIRE((flags & ~3u) == 0, "idkptTraceRays: unknown flags");
    if (ctx->st.UseTlas && ctx->tlasCount == 0) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptTraceRays: UseTlas set but no TLAS nodes uploaded");
    if (count == 0) return IDKPT_OK;
    HIPC(hipSetDevice(ctx->device));
    FLUSH_KEEP();
    Frame f; size_t ldsBytes; uint32_t grid;
    int rc = query_frame(ctx, f, ldsBytes, grid); if (rc) return rc;
    DScene s = make_dscene(ctx);
    hipStream_t st = ctx->stream;
    const idkpt_ray* dIn = rays; idkpt_hit* dOut = hits;
    if (!devicePtrs) {
        HIPC(ctx->queryIn.ensure(count * sizeof(idkpt_ray))); HIPC(ctx->queryOut.ensure(count * sizeof(idkpt_hit)));
        HIPC(hipMemcpyAsync(ctx->queryIn.p, rays, count * sizeof(idkpt_ray), hipMemcpyHostToDevice, st));
        dIn = ctx->queryIn.as<idkpt_ray>(); dOut = ctx->queryOut.as<idkpt_hit>();
    }
    HIPC(ctx->qwork.ensure((WORK_WORDS + 128) * 4));                       // its own work-list counters: the frame's are reset by the frame's last kernel, not per batch
    uint32_t* work = ctx->qwork.as<uint32_t>();
    const int lights = (flags & IDKPT_TRACE_LIGHTS) ? 1 : 0;
    if (ctx->opt.queryScheduler && !f.g.DoDebugBVHTraversal) {
        const bool anyHit = (flags & IDKPT_TRACE_ANY_HIT) != 0;
        // closest hit / any hit: k_trace2's persistent-wave scheduler (kernels_query.hpp): prepare (lights, root test, trace-ready records) -> k_trace2 -> Hit flags
        HIPC(ctx->queryRec.ensure(count * 64)); HIPC(ctx->queryList.ensure(count * 4));
        HIPC(hipMemsetAsync(work, 0, (WORK_WORDS + 128) * 4, st));
        uint32_t* listCount = work + WORK_WORDS;
        f.queryMode = 1; f.g.DoTraceLights = 0;                            // (the lights are folded into the records)
        f.grabUnitLog2 = std::min(24, std::max(6, ctx->opt.grabUnitLog2)); f.grabFixed = std::max(0, ctx->opt.grabFixed); f.leafMin = ctx->opt.leafMin > 0 ? ctx->opt.leafMin : 16;
        f.poolMin = ctx->opt.poolMin; f.advMin = ctx->opt.advMin > 0 ? ctx->opt.advMin : 8; f.recPerRay = 1; f.batch = 1; f.Npad = (uint32_t)count;
        TraceBufs tr = {ctx->queryRec.as<float4>(), nullptr, nullptr};
        const uint32_t blocks = (uint32_t)((count + 255) / 256);
        hipLaunchKernelGGL(k_query_prepare, dim3(blocks), dim3(256), 0, st, s, f, dIn, dOut, (uint32_t)count, lights, anyHit ? 1 : 0, tr, ctx->queryList.as<uint32_t>(), listCount);
        RayBufs noRays = {nullptr, nullptr, nullptr, nullptr, nullptr};
        HitBufs qhits = {(float4*)dOut, ctx->hitCost.as<float>()};
        const uint32_t g2 = std::min<uint32_t>(grid, std::max<uint32_t>(1u, (uint32_t)((count + 63) / 64)));
        launch_trace2<true>(ctx, g2, ldsBytes, st, s, f, noRays, tr, qhits, (const uint32_t*)ctx->queryList.as<uint32_t>(), (const uint32_t*)listCount, work, (uint64_t*)(work + WORK_WORDS + 64) /* visit counters of queries do not count as the frame's */, false, false, 0, anyHit);
        hipLaunchKernelGGL(k_query_finish, dim3(blocks), dim3(256), 0, st, dIn, dOut, (const uint32_t*)ctx->queryList.as<uint32_t>(), (const uint32_t*)listCount);
        HIPC(hipGetLastError());
        if (!devicePtrs) HIPC(hipMemcpyAsync(hits, ctx->queryOut.p, count * sizeof(idkpt_hit), hipMemcpyDeviceToHost, st));
        return IDKPT_OK;
    }
    HIPC(hipMemsetAsync(work, 0, 4, st));
    if (flags & IDKPT_TRACE_ANY_HIT) hipLaunchKernelGGL((k_trace_query<true>), dim3(grid), dim3(WAVE), ldsBytes, st, s, f, dIn, dOut, (uint32_t)count, lights, work);
    else hipLaunchKernelGGL((k_trace_query<false>), dim3(grid), dim3(WAVE), ldsBytes, st, s, f, dIn, dOut, (uint32_t)count, lights, work);
    HIPC(hipGetLastError());
    if (!devicePtrs) HIPC(hipMemcpyAsync(hits, ctx->queryOut.p, count * sizeof(idkpt_hit), hipMemcpyDeviceToHost, st));
    return IDKPT_OK;
}
static int32_t dev_TraceRaysDevice(dev_ctx* ctx, const idkpt_ray* dRays, size_t count, uint32_t flags, idkpt_hit* dHits) { return dev_TraceRaysIssue(ctx, dRays, count, flags, dHits, true); }
static int32_t dev_TraceRays(dev_ctx* ctx, const idkpt_ray* rays, size_t count, uint32_t flags, idkpt_hit* hits)
{
    int rc = dev_TraceRaysIssue(ctx, rays, count, flags, hits); if (rc) return rc;
    if (count == 0) return IDKPT_OK;
    SYNC_CHECKED();
    return IDKPT_OK;
}

static int32_t dev_TraceShadows(dev_ctx* ctx, const idkpt_shadow_params* p, const float* depth, const float* normalOct, float* visibility, bool devicePtrs = false)
{
    if (!ctx || !p) return IDKPT_ERR_INVALID_ARGUMENT;
    if (!ctx->haveScene) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptTraceShadows: no scene uploaded");
    REQUIRE(depth && normalOct && visibility, "idkptTraceShadows: null image");
    REQUIRE(p->Width > 0 && p->Height > 0 && (size_t)p->Width * p->Height < (1ull << 30), "idkptTraceShadows: bad image size");
    REQUIRE(p->RayTracingSamples >= 1, "idkptTraceShadows: RayTracingSamples must be >= 1");
    REQUIRE(p->LightIndex >= 0 && p->LightIndex < ctx->lightCount, "idkptTraceShadows: LightIndex out of range");
    if (ctx->st.UseTlas && ctx->tlasCount == 0) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptTraceShadows: UseTlas set but no TLAS nodes uploaded");
    HIPC(hipSetDevice(ctx->device));
    FLUSH_KEEP();
    Frame f; size_t ldsBytes; uint32_t grid;
    int rc = query_frame(ctx, f, ldsBytes, grid); if (rc) return rc;
    DScene s = make_dscene(ctx);
    const size_t N = (size_t)p->Width * p->Height;
    hipStream_t st = ctx->stream;
    const float* dDepth = depth; const float2* dNormal = (const float2*)normalOct; float* dVis = visibility;
    if (!devicePtrs) {
        HIPC(ctx->queryIn.ensure(N * 12)); HIPC(ctx->queryOut.ensure(N * 4));
        float* in = ctx->queryIn.as<float>();
        HIPC(hipMemcpyAsync(in, depth, N * 4, hipMemcpyHostToDevice, st));
        HIPC(hipMemcpyAsync(in + N, normalOct, N * 8, hipMemcpyHostToDevice, st));
        HIPC(hipMemcpyAsync(ctx->queryOut.p, visibility, N * 4, hipMemcpyHostToDevice, st));
        dDepth = in; dNormal = (const float2*)(in + N); dVis = ctx->queryOut.as<float>();
    }
    const uint32_t tiles = (uint32_t)(((p->Width + 7) / 8) * ((p->Height + 7) / 8));
    hipLaunchKernelGGL(k_shadows, dim3(tiles), dim3(WAVE), ldsBytes, st, s, f, *p, (const float*)dDepth, (const float2*)dNormal, dVis);
    HIPC(hipGetLastError());
    if (devicePtrs) return IDKPT_OK;                                       // (asynchronous, in stream order: idkptSynchronize or the host's own stream wait completes it)
    HIPC(hipMemcpyAsync(visibility, dVis, N * 4, hipMemcpyDeviceToHost, st));
    SYNC_CHECKED();
    return IDKPT_OK;
}
static int32_t dev_TraceShadowsDevice(dev_ctx* ctx, const idkpt_shadow_params* p, const float* dDepth, const float* dNormalOct, float* dVisibility) { return dev_TraceShadows(ctx, p, dDepth, dNormalOct, dVisibility, true); }

static int32_t dev_RefitBlas(dev_ctx* ctx, int32_t blasId)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    if (!ctx->haveScene) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptRefitBlas: no scene uploaded");
    REQUIRE(blasId >= 0 && blasId < (int)ctx->hDescs.size(), "idkptRefitBlas: blasId out of range");
    const GpuBlasDesc& d = ctx->hDescs[blasId];
    if (!d.IsRefittable || d.LeafIndicesCount == 0) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptRefitBlas: BLAS is not refittable (no leaf/parent indices)");
    HIPC(hipSetDevice(ctx->device));
    // (no launch of the queued samples: the triangle records and the nodes are rewritten in slots they do not read, ver_writable)
    int rc = regather_triverts(ctx, (uint32_t)d.TriangleOffset, (uint32_t)d.TriangleCount); if (rc) return rc;
    // The refit writes every node of this BLAS but its unused node 0 (leaves, then the internal nodes level by level); topology words (.w) are read from the
    // state being replaced, child boxes from the state being written.  A scene that is this one BLAS needs no copy of the old state (node 0 aside).
    const bool whole = ctx->hDescs.size() == 1 && d.NodeOffset == 0 && d.NodeCount == ctx->nodeCount && ctx->refitCoversAll[blasId];
    char *nsrc, *ndst; rc = ver_writable(ctx, VB_NODES, whole, &nsrc, &ndst); if (rc) return rc;
    if (whole && nsrc != ndst) HIPC(hipMemcpyAsync(ndst, nsrc, 32, hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(k_refit_leaves, dim3((d.LeafIndicesCount + 63) / 64), dim3(64), 0, ctx->stream, (const float4*)nsrc, (float4*)ndst, ctx->tris.as<uint4>(), (const float4*)vb_cur<float4>(ctx, VB_TRIVERTS),
                       ctx->leaves.as<int32_t>() + d.LeafIndicesOffset, (uint32_t)d.LeafIndicesCount, (uint32_t)d.NodeOffset, (uint32_t)d.TriangleOffset);
    const std::vector<uint32_t>& off = ctx->levelOffsets[blasId];
    for (int l = (int)off.size() - 2; l >= 0; l--) {
        uint32_t cnt = off[l + 1] - off[l];
        if (!cnt) continue;
        hipLaunchKernelGGL(k_refit_level, dim3((cnt + 63) / 64), dim3(64), 0, ctx->stream, (const float4*)nsrc, (float4*)ndst, ctx->levelNodes.as<int32_t>() + ctx->levelBase[blasId] + off[l], cnt, (uint32_t)d.NodeOffset);
    }
    HIPC(hipGetLastError());
    return derive_nodes(ctx, blasId);          // the refitted boxes, in the order the traversal fetches them
}

static int32_t dev_UploadUnskinnedVertices(dev_ctx* ctx, const GpuUnskinnedVertex* verts, int32_t count)
{
    if (!ctx || !verts || count <= 0) return IDKPT_ERR_INVALID_ARGUMENT;
    HIPC(hipSetDevice(ctx->device));
    int rc = upload(ctx, ctx->unskinned, verts, (size_t)count * sizeof(GpuUnskinnedVertex)); if (rc) return rc;
    HIPC(hipStreamSynchronize(ctx->stream));
    ctx->unskinnedCount = count;
    return IDKPT_OK;
}

static int32_t dev_Skin(dev_ctx* ctx, uint32_t inOff, uint32_t outOff, uint32_t jointOff, uint32_t count)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    if (!ctx->haveScene || ctx->unskinnedCount == 0 || ctx->joints.bytes == 0) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptSkin: needs scene, unskinned vertices and joint matrices");
    REQUIRE((uint64_t)inOff + count <= (uint64_t)ctx->unskinnedCount && (uint64_t)outOff + count <= (uint64_t)ctx->vertexCount, "idkptSkin: range out of bounds");
    HIPC(hipSetDevice(ctx->device));
    HIPC(ctx->prevPositions.ensure((size_t)ctx->vertexCount * 12));
    // positions are read by update kernels only (stream order); the re-compressed normals / tangents go to a vertex slot no queued sample reads (ver_writable)
    char *vsrc, *vdst; { int rc = ver_writable(ctx, VB_VERTICES, outOff == 0 && count == (uint32_t)ctx->vertexCount, &vsrc, &vdst); if (rc) return rc; }
    if (count) hipLaunchKernelGGL(k_skin, dim3((count + 63) / 64), dim3(64), 0, ctx->stream, ctx->unskinned.as<GpuUnskinnedVertex>(), ctx->joints.as<float4>(), ctx->positions.as<float>(),
                                  ctx->prevPositions.as<float>(), (const uint4*)vsrc, (uint4*)vdst, inOff, outOff, jointOff, count);
    HIPC(hipGetLastError());
    return IDKPT_OK;
}

static int32_t dev_ResetAccumulation(dev_ctx* ctx) { if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT; ctx->accum[ctx->curSlot] = 0; return IDKPT_OK; }
// Sample-parallel rendering: context r of N renders the reference's samples r, r + N, r + 2N, ... (their RNG streams), each context accumulating
// its own running mean; the mean of the N accumulations is an accumulation over N * K distinct reference samples.
static int32_t dev_SetSampleSequence(dev_ctx* ctx, uint32_t first, uint32_t stride)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(stride >= 1, "idkptSetSampleSequence: stride must be >= 1");
    HIPC(hipSetDevice(ctx->device));
    FLUSH_KEEP();
    if (first == ctx->seqFirst && stride == ctx->seqStride) return IDKPT_OK;
    ctx->seqFirst = first; ctx->seqStride = stride;
    std::fill(ctx->accum.begin(), ctx->accum.end(), 0u);           // other RNG streams: the accumulation starts over
    return IDKPT_OK;
}
static int32_t dev_GetAccumulatedSamples(dev_ctx* ctx, uint32_t* out) { if (!ctx || !out) return IDKPT_ERR_INVALID_ARGUMENT; *out = ctx->accum[ctx->curSlot]; return IDKPT_OK; }

// slots: which state of every versioned buffer the kernels read (null: the current one); multi: the batch's samples saw different states -> the pointers are the
// arena bases and DScene::ver holds every sample's offsets (VER kernels)
static DScene make_dscene(dev_ctx* ctx, const uint8_t* slots, bool multi)
{
    DScene s;
    auto at = [&](int b) -> char* { return multi ? (char*)vb_buf(ctx, b).p : vb_ptr(ctx, b, slots ? slots[b] : ctx->vcur[b]); };
    s.nodes = (const float4*)at(VB_NODES); s.tnodes = ctx->layoutActive ? (const float4*)at(VB_TNODES) : s.nodes; s.tris = ctx->tris.as<uint4>(); s.triVerts = (const float4*)at(VB_TRIVERTS);
    s.descs = ctx->descs.as<GpuBlasDesc>(); s.instances = ctx->instances.as<GpuBlasInstance>(); s.instanceCount = ctx->instanceCount;
    s.tlas = (const float4*)at(VB_TLAS); s.tlasCount = ctx->tlasCount; s.vertices = (const uint4*)at(VB_VERTICES);
    s.meshes = ctx->meshes.as<GpuMesh>(); s.materials = ctx->materials.as<GpuMaterial>(); s.xforms = (const float4*)at(VB_XFORMS);
    s.lights = ctx->lights.as<GpuLight>(); s.lightCount = ctx->lightCount; s.sky = ctx->sky.as<float4>(); s.skySize = ctx->skySize;
    s.textures = ctx->texDescs.as<TexDesc>(); s.textureCount = ctx->textureCount;
    s.overflow = ctx->dOverflow;
    s.ver = multi ? ctx->verTab.as<uint32_t>() : nullptr;
    return s;
}
static DScene make_dscene(dev_ctx* ctx) { return make_dscene(ctx, nullptr, false); }
static DScene make_dscene_last(dev_ctx* ctx) { return make_dscene(ctx, ctx->lastSlots, ctx->lastMulti); }   // what the last launched batch read (finish_deferred, regeneration of culled rays)

static float4* image_ptr(dev_ctx* ctx, int i, int slot) { return ctx->img[i].as<float4>() + (size_t)slot * ((size_t)ctx->W * ctx->rows); }

// fast path = persistent while-while traversal (one BLAS, instance list or TLAS); only the debug traversal-cost view uses the general kernel
static bool fast_path(dev_ctx* ctx) { return ctx->instanceCount >= 1 && !ctx->st.Gpu.DoDebugBVHTraversal && !ctx->opt.forceGeneric; }

// One batch of B deferred samples: FirstHit -> [sort ->] NHit x (RayDepth-1) -> FinalDraw (PathTracer.cs:218-270), every
// stage launched once for all B samples.  Sample k owns ray ids [k*Npad, k*Npad+N); alive queues are batch-wide but stay
// grouped by sample (stable compaction / sort with the sample index above the key), and every ray's NHit slot is its
// position inside its own sample's queue, so each sample gets exactly the RNG streams of a stand-alone frame.
// Grid of a traversal launch whose ray count is only known on the device: `prev` = the count the same launch had in the previous batch (host-mapped mirror,
// possibly one batch stale; 0 = unknown -> full grid).  Any grid >= 1 is correct (the waves are persistent); the size only costs or saves time:
//   * never more than hintMul x the waves that hold all rays at once (tiny frames would otherwise spend their time dispatching idle workgroups), at least 256;
//   * launches below ~1.5 rays per lane of the full grid run faster on FEWER, fuller waves — every wave instruction costs the same whatever its exec mask, and a
//     launch this small lasts as long as its longest rays, whose steps get faster when fewer waves share a SIMD: raysX4 / 4 rays per lane, but not below
//     1024 waves where the first rule allows them (round 3, same box: headline one frame at a time +6 %, Cornell 1080p RayDepth 5 +12 %; profiles/r03_trace_experiments.md 7).
//   * launches of up to GRID_MID_RAYS rays (the headline frame with up to ~24 samples in flight, one rank's share of an N-GPU frame) run 2-4 % faster on 20 than on 24
//     waves per CU for the same reason, and views whose launches are that small only with a few samples in flight (every pixel traversing) lose nothing measurable;
//     above it 24 is never worse (profiles/r03_trace_experiments.md 9).
#define GRID_MID_RAYS 14000000u
// Which traversal kernel a launch gets: k_trace2s (kernels_trace_split.hpp: long rays split across the idle lanes of their wave once the work list is empty) pays
// where a launch ends with a few long rays on an otherwise idle chip — launches of up to SPLIT_MAX_RAYS rays (a frame traced alone, the bounce launches of small
// batches, one rank's share of an N-GPU frame); its extra registers (one wave per SIMD less) cost large launches more than their tails are worth.
// Measured (profiles/r04_small_launch_experiments.md): headline view one frame at a time (0.36 M + 0.28 M rays per launch) +8 %, three samples in flight +4 %, one
// rank's share of an 8 / 4-GPU frame +12 % / +5 %; the atrium and the interior view one frame at a time (1.9-2.1 M rays per launch, every pixel traverses) -10 % / -1.5 %.
// So: launches of fewer than SPLIT_MAX_RAYS rays, and only on views where most pixels miss the scene's root box (fewer than half of the primary rays entered the
// traversal in the previous batch) — there the launch time is the dependent chain of the rays that cross the whole scene without hitting anything.
#define SPLIT_MAX_RAYS 1500000u
static bool want_split(const dev_ctx* ctx, uint32_t prev, bool known, int samples)
{
    if (ctx->opt.split == 0) return false;
    if (ctx->opt.split >= 2) return true;
    const uint64_t pixels = (uint64_t)ctx->W * ctx->rows * (uint64_t)std::max(1, samples);
    const bool sparse = ctx->lastFast && ctx->lastBatch == samples && (uint64_t)ctx->hCounts[MAX_DEPTH_SLOTS - 1] * 2u < pixels;
    return known && sparse && prev > 0u && prev < SPLIT_MAX_RAYS;
}
// k_trace_fused: where a batch's two traversal launches are bound by their longest rays, not by their ray count (the same regime as the split)
// (measured: it saves launches, not chain length — +5 % where one sparse frame is traced alone, a loss everywhere else: kernels_trace_fused.hpp)
#define FUSED_MAX_RAYS 600000u
static bool want_fused(const dev_ctx* ctx, uint32_t prev, bool known, int samples)
{
    if (ctx->opt.fused == 0) return false;
    if (ctx->opt.fused >= 2) return true;
    const uint64_t pixels = (uint64_t)ctx->W * ctx->rows * (uint64_t)std::max(1, samples);
    return known && prev > 0u && prev < FUSED_MAX_RAYS && (uint64_t)prev * 2u < pixels;
}
// k_trace2q (kernels_trace_quad.hpp): two binary levels per round trip; 3x the bytes, so only where the chain of the longest rays bounds the launch
#define QUAD_MAX_RAYS 1500000u
static bool want_quad(const dev_ctx* ctx, uint32_t prev, bool known)
{
    if (ctx->opt.quad == 0) return false;
    if (ctx->opt.quad >= 2) return true;
    return known && prev > 0u && prev < QUAD_MAX_RAYS;
}
#define SPEC_MAX_RAYS 1500000u
static bool want_spec(const dev_ctx* ctx, uint32_t prev, bool known)
{
    if (ctx->opt.spec == 0) return false;
    if (ctx->opt.spec >= 2) return true;
    return known && prev > 0u && prev < SPEC_MAX_RAYS;
}
static uint32_t small_launch_grid(uint32_t fullGrid, uint32_t prev, int hintMul, int raysX4, uint32_t midGrid)
{
    if (prev == 0u || hintMul <= 0) return fullGrid;
    const uint32_t cap = std::max<uint32_t>(256u, (uint32_t)(((uint64_t)hintMul * prev + 63) / 64));
    uint32_t g = cap;
    if (raysX4 > 0) g = std::max<uint32_t>((uint32_t)(((uint64_t)prev * 4u / (uint32_t)raysX4 + 63) / 64), std::min<uint32_t>(cap, 1024u));
    if (midGrid > 0u && prev < GRID_MID_RAYS) g = std::min(g, midGrid);
    return std::min(fullGrid, std::min(g, cap));
}

// The continuation of a deferred last bounce (k_shade_last): the radiance k_shade_last replaced goes back, then the ordinary kernels of the bounce run — shading,
// scan, scatter — on the inputs the batch left untouched (hit records, ray state, the queue entering the bounce, its per-sample bases).  Afterwards ray state, alive
// queue and counts are what the eager path leaves, bit for bit; the frame was complete before.
static int finish_deferred(dev_ctx* ctx)
{
    if (!ctx->defer.valid) return IDKPT_OK;
    ctx->defer.valid = false;
    using namespace ptd;
    hipStream_t st = ctx->stream;
    const int j = ctx->defer.j, side = ctx->defer.side, B = ctx->defer.B, BS = MAX_BATCH + 1;
    const uint32_t total = ctx->defer.total, Npad = ctx->defer.Npad, gridTotal = (total + 255) / 256;
    const Frame f = ctx->lastFrame;
    DScene s = make_dscene_last(ctx);                                // the scene states the deferred batch was traced with (its slots are pinned until now: ver_writable)
    const bool multiVer = ctx->lastMulti;
    RayBufs rays = {ctx->rayO.as<float4>(), ctx->rayT.as<float4>(), ctx->rayR.as<float4>(), ctx->aovA.as<float4>(), ctx->aovN.as<float4>()};
    HitBufs hits = {ctx->hit.as<float4>(), ctx->hitCost.as<float>()};
    TraceBufs tr = {ctx->trRec.as<float4>(), nullptr, nullptr};
    uint32_t* counts = ctx->counts.as<uint32_t>(); uint32_t* bases = ctx->bases.as<uint32_t>();
    unsigned long long* contMask = ctx->contMask.as<unsigned long long>();
    uint32_t* waveLocal = ctx->waveCounts.as<uint32_t>(); uint32_t* blockSums = ctx->blockSums.as<uint32_t>(); uint32_t* keysTmp = ctx->keysTmp.as<uint32_t>();
    const uint32_t scanBlocks = ((total + 63) / 64 + SCAN_WAVES_PER_BLOCK - 1) / SCAN_WAVES_PER_BLOCK;
    const uint32_t* q = ctx->queue[side].as<uint32_t>();
    const uint32_t* cnt = ctx->deferCount.as<uint32_t>();            // (counts[j] itself was reset by the batch's last kernel)
    if (ctx->defer.allHits) hipLaunchKernelGGL((k_restore_last<true>), dim3(gridTotal), dim3(256), 0, st, rays, hits, q, cnt, (const float4*)ctx->radSave.as<float4>(), f.hitsByRid);
    else hipLaunchKernelGGL((k_restore_last<false>), dim3(gridTotal), dim3(256), 0, st, rays, hits, q, cnt, (const float4*)ctx->radSave.as<float4>(), f.hitsByRid);
    if (multiVer) hipLaunchKernelGGL((k_shade<false, true>), dim3(gridTotal), dim3(256), 0, st, s, f, rays, tr, hits, q, cnt, 0u, (const uint32_t*)(bases + j * BS), (const uint32_t*)nullptr, contMask, waveLocal, keysTmp);
    else hipLaunchKernelGGL((k_shade<false, false>), dim3(gridTotal), dim3(256), 0, st, s, f, rays, tr, hits, q, cnt, 0u, (const uint32_t*)(bases + j * BS), (const uint32_t*)nullptr, contMask, waveLocal, keysTmp);
    hipLaunchKernelGGL((k_scan_local<false>), dim3(scanBlocks), dim3(SCAN_WAVES_PER_BLOCK), 0, st, cnt, 0u, (const uint8_t*)nullptr, contMask, waveLocal, blockSums);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, st, cnt, 0u, blockSums, (const uint32_t*)waveLocal, counts + j + 1, (unsigned long long*)nullptr,
                       (const unsigned long long*)contMask, (const uint32_t*)(bases + j * BS), Npad, B, bases + (j + 1) * BS,
                       ctx->dCountsMirror + j + 1, ctx->dBasesMirror + (size_t)(j + 1) * BS, (const uint32_t*)nullptr, (uint32_t*)nullptr);
    hipLaunchKernelGGL((k_compact<false>), dim3(gridTotal), dim3(256), 0, st, q, cnt, 0u, (const unsigned long long*)contMask, (const uint32_t*)waveLocal, (const uint32_t*)blockSums,
                       (const uint32_t*)keysTmp, ctx->queue[1 - side].as<uint32_t>(), ctx->keys[1 - side].as<uint32_t>());
    HIPC(hipGetLastError());
    ctx->lastQueueSide = 1 - side;
    ctx->countersDirty = true;                                       // (counts[j + 1] was written after the batch's reset: the next batch clears its counters itself)
    return IDKPT_OK;
}

static int flush_batch(dev_ctx* ctx)
{
    const int B = (int)ctx->pending.size();
    if (B == 0) return IDKPT_OK;
    if (ctx->grouped && !ctx->inGroupFlush) return fail(ctx, IDKPT_ERR_UNKNOWN, "internal: a member of a multi-device context was flushed on its own");
    const uint32_t N = (uint32_t)((size_t)ctx->W * ctx->rows);
    const uint32_t Npad = ctx->Npad;
    const uint32_t total = (uint32_t)B * Npad;
    // which state of the geometry every sample sees (scene versions): one for all -> plain pointers; else a per-sample table of offsets into the arenas (VER kernels)
    bool multiVer = false;
    for (int k = 1; k < B && !multiVer; k++) multiVer = memcmp(ctx->pending[k].vs, ctx->pending[0].vs, VB_COUNT) != 0;
    for (int b = 0; b < VB_COUNT; b++) { uint64_t m = 0; for (int k = 0; k < B; k++) m |= 1ull << ctx->pending[k].vs[b]; ctx->lastMask[b] = m; ctx->lastSlots[b] = ctx->pending[0].vs[b]; }
    ctx->lastMulti = multiVer;
    if (multiVer) {
        if (!ctx->hVerTab) {
            HIPC(hipHostMalloc((void**)&ctx->hVerTab, (size_t)2 * MAX_BATCH * SCENE_VER_WORDS * 4, hipHostMallocDefault));
            for (int i = 0; i < 2; i++) HIPC(hipEventCreateWithFlags(&ctx->evVer[i], hipEventDisableTiming));
            ctx->verHalf = 0;
        } else HIPC(hipEventSynchronize(ctx->evVer[ctx->verHalf]));          // the copy that last read this half has finished
        uint32_t* stage = ctx->hVerTab + (size_t)ctx->verHalf * MAX_BATCH * SCENE_VER_WORDS;
        for (int k = 0; k < B; k++) {
            const uint8_t* vs = ctx->pending[k].vs;
            uint32_t* row = stage + (size_t)k * SCENE_VER_WORDS;
            for (int b = 0; b < VB_COUNT; b++) row[b] = (uint32_t)(((size_t)vs[b] * ctx->vstride[b]) / 16);   // 16-byte units
            if (!ctx->layoutActive) row[VB_TNODES] = row[VB_NODES];
            row[6] = row[7] = 0u;
        }
        HIPC(ctx->verTab.ensure((size_t)MAX_BATCH * SCENE_VER_WORDS * 4));
        HIPC(hipMemcpyAsync(ctx->verTab.p, stage, (size_t)B * SCENE_VER_WORDS * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPC(hipEventRecord(ctx->evVer[ctx->verHalf], ctx->stream));
        ctx->verHalf ^= 1;
    }
    DScene s = make_dscene_last(ctx);
    Frame f; memset(&f, 0, sizeof(f));                                 // (every field a kernel variant may look at has a defined value: queryMode, hitsByRid, ...)
    memcpy(f.invProj, ctx->pending[0].cam, 64); memcpy(f.invView, ctx->pending[0].cam + 16, 64); memcpy(f.viewPos, ctx->pending[0].cam + 32, 12);   // the camera the samples were queued with
    f.W = ctx->W; f.H = ctx->H; f.rowMod = ctx->rowMod; f.rowRem = ctx->rowRem; f.rows = ctx->rows; f.rowBandLog2 = ctx->rowBandLog2;
    f.g = ctx->st.Gpu; f.useTlas = ctx->st.UseTlas;
    f.stackCap = std::max(1, ctx->st.BlasStackSize > 0 ? ctx->st.BlasStackSize : ctx->sceneStack);
    f.outputAovs = ctx->st.OutputAOVs;
    f.batch = B; f.Npad = Npad;
    // trace-ready records per ray id (pt_kernels.hpp Frame::recPerRay): one, or one per instance (+ the world ray under USE_TLAS) on scenes of few instances
    f.recPerRay = 1;
    if (ctx->opt.instanceRecords && ctx->instanceCount <= MAX_REC_INSTANCES && (ctx->instanceCount > 1 || ctx->st.UseTlas)) f.recPerRay = ctx->instanceCount + (ctx->st.UseTlas ? 1 : 0);
    HIPC(ctx->trRec.ensure((size_t)ctx->maxBatch * Npad * 64 * (size_t)f.recPerRay));   // (grows on the first batch of such a scene; nothing of an earlier batch is read from it: a deferred last bounce is dropped below)
    for (int k = 0; k < MAX_BATCH; k++) { f.accum[k] = k < B ? ctx->pending[k].accum : 0u; f.slotOf[k] = (uint32_t)(k < B ? ctx->pending[k].slot : 0); }
    f.seqFirst = ctx->seqFirst; f.seqStride = ctx->seqStride;
    f.accumulated = f.seqFirst + f.accum[0] * f.seqStride;
    f.cams = nullptr;
    if (ctx->ringSize > 1) {   // frame ring: every sample renders with the camera it was queued with
        // pinned double-buffered staging: no stream synchronisation per batch (the host may run ahead of the GPU)
        if (!ctx->hCams) {
            HIPC(hipHostMalloc((void**)&ctx->hCams, (size_t)2 * MAX_BATCH * 36 * 4, hipHostMallocDefault));
            for (int i = 0; i < 2; i++) HIPC(hipEventCreateWithFlags(&ctx->evCams[i], hipEventDisableTiming));
            ctx->camHalf = 0;
        } else HIPC(hipEventSynchronize(ctx->evCams[ctx->camHalf]));          // the copy that last read this half has finished
        float* stage = ctx->hCams + (size_t)ctx->camHalf * MAX_BATCH * 36;
        for (int k = 0; k < B; k++) memcpy(stage + (size_t)k * 36, ctx->pending[k].cam, 36 * 4);
        HIPC(ctx->camTab.ensure((size_t)MAX_BATCH * 36 * 4));
        HIPC(hipMemcpyAsync(ctx->camTab.p, stage, (size_t)B * 36 * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPC(hipEventRecord(ctx->evCams[ctx->camHalf], ctx->stream));
        ctx->camHalf ^= 1;
        f.cams = ctx->camTab.as<float>();
    }
    f.tilePerSample = (f.cams != nullptr || multiVer) ? 1 : 0;
    RayBufs rays = {ctx->rayO.as<float4>(), ctx->rayT.as<float4>(), ctx->rayR.as<float4>(), ctx->aovA.as<float4>(), ctx->aovN.as<float4>()};
    HitBufs hits = {ctx->hit.as<float4>(), ctx->hitCost.as<float>()};
    uint32_t* counts = ctx->counts.as<uint32_t>();
    uint32_t* bases = ctx->bases.as<uint32_t>();             // [MAX_DEPTH_SLOTS][MAX_BATCH+1]
    uint32_t* work = ctx->work.as<uint32_t>();
    uint64_t* counters = ctx->counters64.as<uint64_t>();
    const int depth = ctx->st.RayDepth;
    hipStream_t st = ctx->stream;
#ifdef IDKPT_DEVELOPER
    // "graph_probe" (developer build): is a hipGraph of a batch's launches faster than the launches?  The batch is captured instead of executed, then
    // executed once as a graph and replayed graph_probe times between two events (the replays re-accumulate the same sample: timing only).
    bool capturing = false;
    if (ctx->opt.graphProbe > 0 && !ctx->timing) capturing = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess;
#endif
    if (ctx->timing) HIPC(hipEventRecord(ctx->evFrame[0], st));
    ctx->defer.valid = false;                                      // (a deferred last bounce of the previous batch that nobody asked for: its buffers are reused now)
    if (ctx->countersDirty) { HIPC(hipMemsetAsync(work, 0, WORK_WORDS * 4, st)); HIPC(hipMemsetAsync(counts, 0, MAX_DEPTH_SLOTS * 4, st)); }   // (otherwise the previous batch's k_final_draw has reset them)
    ctx->countersDirty = true;
    uint32_t* hostCounts = ctx->dCountsMirror; uint32_t* hostBases = ctx->dBasesMirror;

    f.tlasCap = std::min(TLAS_STACK_SIZE, std::max(1, ctx->tlasNeed));
    f.grabUnitLog2 = std::min(24, std::max(6, ctx->opt.grabUnitLog2)); f.grabFixed = std::max(0, ctx->opt.grabFixed);   // work-list hand-out (kernels_trace.hpp)
    f.leafMin = ctx->opt.leafMin > 0 ? ctx->opt.leafMin : (B >= 4 ? 16 : 12);        // (measured: 16-20 with many samples in flight, 12 for a frame traced alone; tools/sweep_sched.py)
    size_t ldsBytes = (size_t)(f.stackCap + 2 + (f.useTlas ? f.tlasCap : 0)) * WAVE * 4;   // + the dummy and the spare row of k_trace2's stack (kernels_trace.hpp)
    ldsBytes += (size_t)std::max(0, ctx->opt.ldsPad);   // option "lds_pad": caps the waves per CU (occupancy experiments)
    if (ldsBytes > 64 * 1024) { ctx->pending.clear(); return fail(ctx, IDKPT_ERR_INVALID_ARGUMENT, "BlasStackSize too large for the LDS traversal stack"); }
    // persistent trace grid: as many 1-wave workgroups as the chip holds (32 waves/CU, limited by LDS)
    int wavesPerCU = (int)std::min<size_t>(32, (160 * 1024) / std::max<size_t>(ldsBytes, 1));
    wavesPerCU = std::max(1, wavesPerCU);
    if (ctx->opt.traceWaves > 0) wavesPerCU = ctx->opt.traceWaves;   // option "trace_waves": one-wave workgroups per CU in the persistent grid
    // (a launch never needs more waves than it can have rays: small frames would otherwise spend their time dispatching idle workgroups)
    const uint32_t traceGrid = std::min<uint32_t>((uint32_t)(ctx->numCUs * wavesPerCU), std::max<uint32_t>(1u, (uint32_t)(((size_t)B * N + 63) / 64)));
    const uint32_t midGrid = (ctx->opt.gridMidWaves > 0 && ctx->opt.traceWaves == 0) ? (uint32_t)(ctx->numCUs * std::min(wavesPerCU, ctx->opt.gridMidWaves)) : 0u;   // (an explicit trace_waves wins)
    f.gridRaysX4 = (uint32_t)std::max(0, ctx->opt.gridRaysX4); f.gridMid = midGrid; f.gridMidRays = GRID_MID_RAYS; f.splitMode = (ctx->opt.split == 3 ? 2 : 1) | (ctx->opt.splitDonor ? 4 : 0); f.poolMin = ctx->opt.poolMin; f.advMin = ctx->opt.advMin > 0 ? ctx->opt.advMin : (B >= 4 ? 8 : 1); f.splitPeek = ctx->opt.splitPeek;   // the same rules inside k_trace2, on the launch's actual ray count
    const bool debug = f.g.DoDebugBVHTraversal != 0;
    const uint32_t gridTotal = (total + 255) / 256;
    const bool fast = fast_path(ctx);
    if (fast && wide_wanted(ctx)) { int rc = wide_prepare(ctx); if (rc) { ctx->pending.clear(); return rc; } }
    // one launch for FirstHit + the last NHit (kernels_trace_fused.hpp): RayDepth 2, one BLAS instance, the last bounce deferred (no AOVs, no debug view), nothing that looks at
    // the primary hits or the visit counters, no per-bounce exchange with other contexts — and a launch small enough to be bound by its longest rays
    const bool fused = fast && ctx->st.RayDepth == 2 && ctx->opt.deferLast != 0 && !f.outputAovs && !f.g.DoDebugBVHTraversal && !f.useTlas && ctx->instanceCount == 1 && !multiVer && !ctx->counters
                       && !ctx->capturePrimary && !ctx->groupExchange && !ctx->exchangeFn && !ctx->bandExchangeFn && !ctx->bandExchangeDevFn && f.recPerRay == 1 && (ctx->opt.traceVariant == 0 || ctx->opt.traceVariant == 100)
                       && !(wide_wanted(ctx) && ctx->opt.fused < 2)    // (the wide-node walk shortens the dependent chains the fused launch only stops paying launches for)
                       && want_fused(ctx, ctx->hCounts[MAX_DEPTH_SLOTS - 1], ctx->lastFast && ctx->lastBatch == B, B);
    f.hitsByRid = fused ? 1 : 0; f.shadeMin = ctx->opt.fusedShadeMin; f.scatterLog2 = ctx->opt.splitScatter;
    if (!fast && (B != 1 || multiVer)) { ctx->pending.clear(); return fail(ctx, IDKPT_ERR_UNKNOWN, "internal: generic path is never batched"); }
    unsigned long long* contMask = ctx->contMask.as<unsigned long long>();
    uint32_t* waveCounts = ctx->waveCounts.as<uint32_t>();
    const int BS = MAX_BATCH + 1;

    const uint8_t* tileClass = nullptr;                   // per-tile pre-classification of this batch (fast path with pre-cull only)
    // ---- FirstHit
    uint32_t* activeList = ctx->sortVals.as<uint32_t>(); // scratch (capacity uints), free until the first sort
    uint32_t* activeCount = counts + (MAX_DEPTH_SLOTS - 1);
    TraceBufs tr = {ctx->trRec.as<float4>(), nullptr, nullptr};
    TraceBufs trNone = {nullptr, nullptr, nullptr};
    uint32_t* waveLocal = waveCounts;                     // per-wave exclusive offset inside its 256-wave scan block
    uint32_t* blockSums = ctx->blockSums.as<uint32_t>();
    const uint32_t scanBlocks = ((total + 63) / 64 + SCAN_WAVES_PER_BLOCK - 1) / SCAN_WAVES_PER_BLOCK;
    uint32_t* keysTmp = ctx->keysTmp.as<uint32_t>();
    {
        if (fast) {
            const uint32_t tilesX = ((uint32_t)f.W + 7) / 8, tilesY = ((uint32_t)f.rows + 7) / 8;
            const uint32_t genWaves = tilesX * tilesY;
            const int cull = f.g.DoTraceLights ? 0 : 1;
            // single instance without lights: k_trace2 reads nothing but the trace-ready record, so the planes of a surviving primary ray need not exist before k_shade_first
            const bool noLean = ctx->opt.noLeanPrimary != 0;
            const int lean = (cull && !f.useTlas && s.instanceCount == 1 && !noLean) ? 1 : 0;
            if (ctx->capturePrimary) hipLaunchKernelGGL(k_fill_miss, dim3((N + 255) / 256), dim3(256), 0, st, hits, (size_t)(B - 1) * Npad, N);
            tileClass = nullptr;
            if (cull && !ctx->opt.noTileCull) {   // sample-independent pre-classification of the 8x8 tiles (conservative whole-tile miss test)
                const uint32_t classSets = f.tilePerSample ? (uint32_t)B : 1u;       // one classification per camera / scene version
                HIPC(ctx->tileClass.ensure((size_t)genWaves * classSets));
                if (multiVer) hipLaunchKernelGGL((k_classify_tiles<true>), dim3((genWaves + 255) / 256, classSets), dim3(256), 0, st, s, f, ctx->tileClass.as<uint8_t>(), tilesX, tilesY);
                else hipLaunchKernelGGL((k_classify_tiles<false>), dim3((genWaves + 255) / 256, classSets), dim3(256), 0, st, s, f, ctx->tileClass.as<uint8_t>(), tilesX, tilesY);
                tileClass = ctx->tileClass.as<uint8_t>();
            }
            if (multiVer) hipLaunchKernelGGL((k_gen_primary<true>), dim3(B, (genWaves + 15) / 16), dim3(1024), 0, st, s, f, rays, tr, cull, activeList, activeCount, keysTmp, ctx->contFlag.as<uint8_t>(), tileClass, lean);
            else hipLaunchKernelGGL((k_gen_primary<false>), dim3(B, (genWaves + 15) / 16), dim3(1024), 0, st, s, f, rays, tr, cull, activeList, activeCount, keysTmp, ctx->contFlag.as<uint8_t>(), tileClass, lean);
            TRACE_T0();
            uint32_t grid0 = traceGrid;
            if (ctx->opt.gridRaysX4 > 0 && ctx->lastFast && ctx->lastBatch == B) grid0 = small_launch_grid(traceGrid, ctx->hCounts[MAX_DEPTH_SLOTS - 1], 2, ctx->opt.gridRaysX4, midGrid);
            if (fused) {
                // FirstHit's traversal, its shading and the bounce's traversal in one persistent launch (kernels_trace_fused.hpp); the bounce's hits are stored per ray id
                hipLaunchKernelGGL((k_trace_fused<32>), dim3(grid0), dim3(WAVE), ldsBytes, st, s, f, rays, tr, hits, (const uint32_t*)activeList, (const uint32_t*)activeCount, work + 0, ctx->contFlag.as<uint8_t>(), keysTmp, lean);
            } else
            launch_trace2<true>(ctx, grid0, ldsBytes, st, s, f, rays, tr, hits, (const uint32_t*)activeList, (const uint32_t*)activeCount, work + 0, counters,
                                want_split(ctx, ctx->hCounts[MAX_DEPTH_SLOTS - 1], ctx->lastFast && ctx->lastBatch == B, B), want_spec(ctx, ctx->hCounts[MAX_DEPTH_SLOTS - 1], ctx->lastFast && ctx->lastBatch == B), 0, false,
                                want_quad(ctx, ctx->hCounts[MAX_DEPTH_SLOTS - 1], ctx->lastFast && ctx->lastBatch == B));
            TRACE_T1();
            if (ctx->capturePrimary) { HIPC(ctx->primHit.ensure((size_t)N * 16)); hipLaunchKernelGGL(k_capture_primary, dim3((N + 255) / 256), dim3(256), 0, st, hits, (size_t)(B - 1) * Npad, N, ctx->primHit.as<float4>()); }
            if (fused) {}
            else if (multiVer) hipLaunchKernelGGL((k_shade_first<true>), dim3(gridTotal), dim3(256), 0, st, s, f, rays, tr, hits, (const uint32_t*)activeList, (const uint32_t*)activeCount, ctx->contFlag.as<uint8_t>(), keysTmp, lean);
            else hipLaunchKernelGGL((k_shade_first<false>), dim3(gridTotal), dim3(256), 0, st, s, f, rays, tr, hits, (const uint32_t*)activeList, (const uint32_t*)activeCount, ctx->contFlag.as<uint8_t>(), keysTmp, lean);
            hipLaunchKernelGGL((k_scan_local<true>), dim3(scanBlocks), dim3(SCAN_WAVES_PER_BLOCK), 0, st, (const uint32_t*)nullptr, total, (const uint8_t*)ctx->contFlag.as<uint8_t>(), contMask, waveLocal, blockSums);
        } else {
            TRACE_T0();
            uint32_t g = std::min<uint32_t>(traceGrid, (N + 63) / 64);
            if (ctx->counters) { if (debug) hipLaunchKernelGGL((k_trace_primary<true, true>), dim3(g), dim3(WAVE), ldsBytes, st, s, f, rays, hits, N, work + 0, counters); else hipLaunchKernelGGL((k_trace_primary<true, false>), dim3(g), dim3(WAVE), ldsBytes, st, s, f, rays, hits, N, work + 0, counters); }
            else { if (debug) hipLaunchKernelGGL((k_trace_primary<false, true>), dim3(g), dim3(WAVE), ldsBytes, st, s, f, rays, hits, N, work + 0, counters); else hipLaunchKernelGGL((k_trace_primary<false, false>), dim3(g), dim3(WAVE), ldsBytes, st, s, f, rays, hits, N, work + 0, counters); }
            TRACE_T1();
            if (ctx->capturePrimary) { HIPC(ctx->primHit.ensure((size_t)N * 16)); hipLaunchKernelGGL(k_capture_primary, dim3((N + 255) / 256), dim3(256), 0, st, hits, (size_t)0, N, ctx->primHit.as<float4>()); }
            hipLaunchKernelGGL((k_shade<true, false>), dim3(gridTotal), dim3(256), 0, st, s, f, rays, trNone, hits, (const uint32_t*)nullptr, (const uint32_t*)nullptr, total, (const uint32_t*)nullptr, (const uint32_t*)nullptr,
                               contMask, waveCounts, keysTmp);
            hipLaunchKernelGGL((k_scan_local<false>), dim3(scanBlocks), dim3(SCAN_WAVES_PER_BLOCK), 0, st, (const uint32_t*)nullptr, total, (const uint8_t*)nullptr, contMask, waveLocal, blockSums);
        }
        hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, st, (const uint32_t*)nullptr, total, blockSums, (const uint32_t*)waveLocal, counts + 1, (unsigned long long*)(1 < depth ? counters + 2 : nullptr),
                           (const unsigned long long*)contMask, (const uint32_t*)nullptr, Npad, B, bases + 1 * BS,
                           hostCounts + 1, hostBases + 1 * BS, (const uint32_t*)(counts + MAX_DEPTH_SLOTS - 1), hostCounts + MAX_DEPTH_SLOTS - 1);
        if (ctx->evBounce) HIPC(hipEventRecord(ctx->evBounce[1], st));      // bases[1] (alive counts entering bounce 1) are final
        hipLaunchKernelGGL((k_compact<true>), dim3(gridTotal), dim3(256), 0, st, (const uint32_t*)nullptr, (const uint32_t*)nullptr, total, (const unsigned long long*)contMask, (const uint32_t*)waveLocal, (const uint32_t*)blockSums,
                           (const uint32_t*)keysTmp, ctx->queue[1].as<uint32_t>(), ctx->keys[1].as<uint32_t>());
    }
    int side = 1; // queue[side] holds the rays entering bounce j, its length is counts[j], sample k starts at bases[j][k]
    // may the last bounce's continuation wait until somebody asks for it?  (k_shade_last: only where a hit of that bounce cannot change the radiance and nothing else of it reaches the frame)
    const bool deferLast = fast && depth >= 2 && ctx->opt.deferLast != 0 && !f.outputAovs && !debug;
    const bool deferAllHits = !(ctx->sceneNoEmission && !(f.g.DoTraceLights && s.lightCount > 0));   // a hit of the last bounce may add radiance: emission somewhere in the scene, or light hits
    for (int j = 1; j < depth; j++) {
        uint32_t* q = ctx->queue[side].as<uint32_t>(); uint32_t* k = ctx->keys[side].as<uint32_t>();
        const uint32_t* cnt = counts + j;
        const uint32_t* kq = k;                                           // keys of the queue entries, position by position
        // exact multi-GPU deep paths: the host tells every sample how many alive rays the contexts above this strip hold (idkpt.h)
        const uint32_t* gbase = nullptr;
        const bool bandExchange = (ctx->bandExchangeFn || ctx->bandExchangeDevFn) && ctx->rowMod > 1 && !(ctx->st.DoRaySorting && j > 1);
        if (bandExchange) {}                                               // (below; a member of a multi-device context with interleaved rows takes this route as well)
        else if (ctx->groupExchange) {   // member of a multi-device context: the group sums the counts of the members that own earlier rows, on the device (idkpt_api.hpp)
            int rc = ctx->groupExchange(ctx->groupUser, ctx, j, B, &gbase); if (rc) { ctx->pending.clear(); return rc; }
        } else if (ctx->exchangeFn) {
            std::vector<uint32_t> hb(B + 1), local(B), outBases(B, 0u);
            HIPC(hipMemcpyAsync(hb.data(), bases + j * BS, (size_t)(B + 1) * 4, hipMemcpyDeviceToHost, st));
            HIPC(hipStreamSynchronize(st));
            for (int b2 = 0; b2 < B; b2++) local[b2] = hb[b2 + 1] - hb[b2];
            ctx->exchangeFn(ctx->exchangeUser, j, B, local.data(), outBases.data());
            HIPC(ctx->gbases.ensure((size_t)MAX_BATCH * 4));
            HIPC(hipMemcpyAsync(ctx->gbases.p, outBases.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
            HIPC(hipStreamSynchronize(st));                      // outBases is a stack vector
            gbase = ctx->gbases.as<uint32_t>();
        }
        if (bandExchange) {
            // interleaved rows / bands (idkpt.h idkptSetBandExchange): the rays of one local band are a contiguous run of a sample's queue segment (ordered compaction);
            // the host returns, per (sample, band), the alive rays of all contexts in the image bands before it; k_shade adds the position inside the run
            const int bandRows = 1 << ctx->rowBandLog2, LB = (ctx->rows + bandRows - 1) / bandRows;
            HIPC(ctx->bandTab.ensure((size_t)4 * MAX_BATCH * ((size_t)LB + 1) * 4));
            uint32_t* dStarts = ctx->bandTab.as<uint32_t>(); uint32_t* dTab = dStarts + (size_t)MAX_BATCH * (LB + 1);
            hipLaunchKernelGGL(k_band_starts, dim3((uint32_t)(((size_t)B * (LB + 1) + 255) / 256)), dim3(256), 0, st, (const uint32_t*)q, (const uint32_t*)(bases + j * BS), B, LB, (uint32_t)ctx->W * (uint32_t)bandRows, Npad, dStarts);
            if (ctx->bandExchangeDevFn) {
                // device-side variant: counts -> (the host enqueues its exchange on this stream) -> bases -> table; nothing waits on the host
                uint32_t* dCounts = dTab + (size_t)MAX_BATCH * (LB + 1); uint32_t* dBases = dCounts + (size_t)MAX_BATCH * (LB + 1);
                const uint32_t nb = (uint32_t)(((size_t)B * LB + 255) / 256);
                hipLaunchKernelGGL(k_band_counts, dim3(nb), dim3(256), 0, st, (const uint32_t*)dStarts, B, LB, dCounts);
                HIPC(hipGetLastError());
                ctx->bandExchangeDevFn(ctx->bandExchangeDevUser, j, B, LB, dCounts, dBases, (void*)st);
                hipLaunchKernelGGL(k_band_tab, dim3(nb), dim3(256), 0, st, (const uint32_t*)dStarts, (const uint32_t*)dBases, B, LB, dTab);
                gbase = dTab; f.gbStride = LB; f.gbBands = 1;
            } else {
            std::vector<uint32_t> starts((size_t)B * (LB + 1)), local((size_t)B * LB), outBases((size_t)B * LB, 0u), tab((size_t)B * LB);
            HIPC(hipMemcpyAsync(starts.data(), dStarts, starts.size() * 4, hipMemcpyDeviceToHost, st));
            HIPC(hipStreamSynchronize(st));
            for (int k2 = 0; k2 < B; k2++) for (int b2 = 0; b2 < LB; b2++) local[(size_t)k2 * LB + b2] = starts[(size_t)k2 * (LB + 1) + b2 + 1] - starts[(size_t)k2 * (LB + 1) + b2];
            ctx->bandExchangeFn(ctx->bandExchangeUser, j, B, LB, local.data(), outBases.data());
            for (int k2 = 0; k2 < B; k2++) for (int b2 = 0; b2 < LB; b2++) tab[(size_t)k2 * LB + b2] = outBases[(size_t)k2 * LB + b2] - starts[(size_t)k2 * (LB + 1) + b2];   // (mod 2^32: + position inside the sample's segment = global slot)
            HIPC(hipMemcpyAsync(dTab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, st));
            HIPC(hipStreamSynchronize(st));                      // tab is a stack vector
            gbase = dTab; f.gbStride = LB; f.gbBands = 1;
            }
        }
        if (!bandExchange) { f.gbStride = 1; f.gbBands = 0; }
        if (ctx->st.DoRaySorting && j > 1) {
            // RaySorting() (PathTracer.cs:232-237): stable sort of (key, rayIndex); key = 21-bit triangle id with the batch's
            // sample index above it, so one sort orders every sample's queue exactly like a stand-alone counting sort
            const uint32_t nTiles = (total + SORT_TILE - 1) / SORT_TILE;
            int sampleBits = 0; while ((1 << sampleBits) < B) sampleBits++;
            const int passes = (IDKPT_SORT_KEY_BITS + sampleBits + 6) / 7;    // 7-bit digits over key + sample index: 3 passes alone, 4 up to 128 samples, 5 up to 256
            uint32_t* digitTotals = ctx->sortHist.as<uint32_t>() + (size_t)SORT_RADIX * nTiles;   // 128 words behind the [digit][tile] table
            uint32_t* ka = k; uint32_t* va = q; uint32_t* kb = ctx->sortKeys.as<uint32_t>(); uint32_t* vb = ctx->sortVals.as<uint32_t>();
            for (int pass = 0; pass < passes; pass++) {
                hipLaunchKernelGGL(k_sort_hist, dim3(nTiles), dim3(SORT_BLOCK), 0, st, (const uint32_t*)ka, cnt, (uint32_t)(7 * pass), ctx->sortHist.as<uint32_t>(), nTiles);
                hipLaunchKernelGGL(k_sort_scan, dim3(SORT_RADIX), dim3(1024), 0, st, cnt, ctx->sortHist.as<uint32_t>(), nTiles, digitTotals);
                hipLaunchKernelGGL(k_sort_scatter, dim3(nTiles), dim3(SORT_BLOCK), 0, st, (const uint32_t*)ka, (const uint32_t*)va, cnt, (uint32_t)(7 * pass), (const uint32_t*)ctx->sortHist.as<uint32_t>(), nTiles, (const uint32_t*)digitTotals, kb, vb);
                std::swap(ka, kb); std::swap(va, vb);
            }
            // odd pass count: the sorted data sits in (sortKeys, sortVals) -> copy the indices back (the reference copies W*H*4 B too, PathTracer.cs:296)
            if (va != q) HIPC(hipMemcpyAsync(q, va, (size_t)total * 4, hipMemcpyDeviceToDevice, st));
            kq = ka;                                                      // the keys that line up with the sorted queue
        }
        // trace order (kernels_queue.hpp k_order_*): the bounce launch is handed out by the triangle its rays start on, all samples of the batch together
        TraceBufs trj = tr;
        if (fast && ctx->opt.traceOrder && (ctx->opt.traceOrder >= 2 || B >= 4) && ctx->ordIdx.p) {
            int bits = 1; while (bits < 31 && (1u << bits) < (uint32_t)std::max(2, ctx->triCount)) bits++;
            bits = std::min(bits, IDKPT_SORT_KEY_BITS);                  // (the key holds the low 21 bits of the triangle id, NHit/compute.glsl:81)
            const int shiftLo = std::max(0, bits - 14), nPass = bits > 7 ? 2 : 1;
            const uint32_t nTiles = (total + SORT_TILE - 1) / SORT_TILE;
            uint32_t* digitTotals = ctx->sortHist.as<uint32_t>() + (size_t)SORT_RADIX * nTiles;
            const uint32_t* kin = kq; const uint32_t* vin = nullptr;     // (first pass: value = the item's own index = its slot)
            for (int pass = 0; pass < nPass; pass++) {
                uint32_t* kout = ctx->ordKeys[pass].as<uint32_t>(); uint32_t* vout = ctx->ordVals[pass].as<uint32_t>();
                const uint32_t shift = (uint32_t)(shiftLo + 7 * pass);
                hipLaunchKernelGGL(k_sort_hist, dim3(nTiles), dim3(SORT_BLOCK), 0, st, kin, cnt, shift, ctx->sortHist.as<uint32_t>(), nTiles);
                hipLaunchKernelGGL(k_sort_scan, dim3(SORT_RADIX), dim3(1024), 0, st, cnt, ctx->sortHist.as<uint32_t>(), nTiles, digitTotals);
                hipLaunchKernelGGL(k_sort_scatter, dim3(nTiles), dim3(SORT_BLOCK), 0, st, kin, vin, cnt, shift, (const uint32_t*)ctx->sortHist.as<uint32_t>(), nTiles, (const uint32_t*)digitTotals, kout, vout);
                kin = kout; vin = vout;
            }
            hipLaunchKernelGGL(k_order_gather, dim3(gridTotal), dim3(256), 0, st, vin, (const uint32_t*)q, cnt, ctx->ordIdx.as<uint32_t>());
            trj.order = vin; trj.orderIdx = ctx->ordIdx.as<uint32_t>();
        }
        if (!fused) {
        TRACE_T0();
        // grid of the bounce launch: its queue length is only known on the device; the length the same bounce had in the previous batch (pinned copy,
        // possibly one batch stale) is a good predictor, and a grid that is too small or too large only costs time (the waves are persistent)
        uint32_t gridj = traceGrid;
        const int hintMul = ctx->opt.gridHint;
        if (hintMul > 0 && ctx->lastBatch == B && ctx->hBases) gridj = small_launch_grid(traceGrid, ctx->hBases[(size_t)j * BS + B], hintMul, ctx->opt.gridRaysX4, midGrid);
        if (fast) launch_trace2<false>(ctx, gridj, ldsBytes, st, s, f, rays, trj, hits, (const uint32_t*)q, cnt, work + j, counters,
                                       want_split(ctx, ctx->hBases ? ctx->hBases[(size_t)j * BS + B] : 0u, ctx->lastFast && ctx->lastBatch == B && ctx->hBases != nullptr, B),
                                       want_spec(ctx, ctx->hBases ? ctx->hBases[(size_t)j * BS + B] : 0u, ctx->lastFast && ctx->lastBatch == B && ctx->hBases != nullptr), j, false,
                                       want_quad(ctx, ctx->hBases ? ctx->hBases[(size_t)j * BS + B] : 0u, ctx->lastFast && ctx->lastBatch == B && ctx->hBases != nullptr));
        else {
            if (ctx->counters) hipLaunchKernelGGL((k_trace_queue<true>), dim3(traceGrid), dim3(WAVE), ldsBytes, st, s, f, rays, hits, (const uint32_t*)q, cnt, work + j, counters);
            else hipLaunchKernelGGL((k_trace_queue<false>), dim3(traceGrid), dim3(WAVE), ldsBytes, st, s, f, rays, hits, (const uint32_t*)q, cnt, work + j, counters);
        }
        TRACE_T1();
        }
        if (deferLast && j == depth - 1 && gbase == nullptr) {
            // the last bounce: only its radiance is visible in the frame (kernels_shade.hpp k_shade_last); state, queue and counts follow on demand (finish_deferred)
            HIPC(ctx->radSave.ensure((size_t)ctx->maxBatch * ctx->Npad * 16)); HIPC(ctx->deferCount.ensure(64));
#define SHADE_LAST(A, V) hipLaunchKernelGGL((k_shade_last<A, V>), dim3(gridTotal), dim3(256), 0, st, s, f, rays, hits, (const uint32_t*)q, cnt, (const uint32_t*)(bases + j * BS), ctx->radSave.as<float4>(), ctx->deferCount.as<uint32_t>())
            if (deferAllHits) { if (multiVer) SHADE_LAST(true, true); else SHADE_LAST(true, false); }
            else SHADE_LAST(false, false);                              // (misses only: the sky is not versioned)
#undef SHADE_LAST
            ctx->defer.allHits = deferAllHits; ctx->defer.valid = true; ctx->defer.j = j; ctx->defer.side = side; ctx->defer.B = B; ctx->defer.total = total; ctx->defer.Npad = Npad;
            break;
        }
        if (multiVer) hipLaunchKernelGGL((k_shade<false, true>), dim3(gridTotal), dim3(256), 0, st, s, f, rays, fast ? tr : trNone, hits, (const uint32_t*)q, cnt, 0u, (const uint32_t*)(bases + j * BS), gbase, contMask, waveCounts, keysTmp);
        else hipLaunchKernelGGL((k_shade<false, false>), dim3(gridTotal), dim3(256), 0, st, s, f, rays, fast ? tr : trNone, hits, (const uint32_t*)q, cnt, 0u, (const uint32_t*)(bases + j * BS), gbase, contMask, waveCounts, keysTmp);
        hipLaunchKernelGGL((k_scan_local<false>), dim3(scanBlocks), dim3(SCAN_WAVES_PER_BLOCK), 0, st, cnt, 0u, (const uint8_t*)nullptr, contMask, waveLocal, blockSums);
        hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, st, cnt, 0u, blockSums, (const uint32_t*)waveLocal, counts + j + 1, (unsigned long long*)(j + 1 < depth ? counters + 2 : nullptr),
                           (const unsigned long long*)contMask, (const uint32_t*)(bases + j * BS), Npad, B, bases + (j + 1) * BS,
                           hostCounts + j + 1, hostBases + (size_t)(j + 1) * BS, (const uint32_t*)nullptr, (uint32_t*)nullptr);
        if (ctx->evBounce) HIPC(hipEventRecord(ctx->evBounce[j + 1], st));
        hipLaunchKernelGGL((k_compact<false>), dim3(gridTotal), dim3(256), 0, st, (const uint32_t*)q, cnt, 0u, (const unsigned long long*)contMask, (const uint32_t*)waveLocal, (const uint32_t*)blockSums,
                           (const uint32_t*)keysTmp, ctx->queue[1 - side].as<uint32_t>(), ctx->keys[1 - side].as<uint32_t>());
        side = 1 - side;
    }
    ctx->lastQueueSide = side; ctx->lastQueueCountSlot = depth; ctx->lastFast = fast; ctx->lastNeedsRegen = fast; ctx->lastBatch = B; ctx->lastFrame = f;
    hipLaunchKernelGGL(k_final_draw, dim3((N + 255) / 256), dim3(256), 0, st, s, f, rays, image_ptr(ctx, 0, 0), image_ptr(ctx, 1, 0), image_ptr(ctx, 2, 0), N, tileClass,
                       work, (uint32_t)WORK_WORDS, counts, (uint32_t)MAX_DEPTH_SLOTS);
    HIPC(hipGetLastError());
    ctx->countersDirty = false;
#ifdef IDKPT_DEVELOPER
    if (capturing) {
        hipGraph_t g = nullptr; hipGraphExec_t ex = nullptr;
        const int K = ctx->opt.graphProbe; ctx->opt.graphProbe = 0;
        if (hipStreamEndCapture(st, &g) == hipSuccess && g && hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) == hipSuccess) {
            hipEvent_t e0 = nullptr, e1 = nullptr; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipGraphLaunch(ex, st); (void)hipStreamSynchronize(st);                       // the batch itself (a capture does not execute)
            (void)hipEventRecord(e0, st);
            for (int k = 0; k < K; k++) (void)hipGraphLaunch(ex, st);
            (void)hipEventRecord(e1, st); (void)hipStreamSynchronize(st);
            float ms = 0.0f; (void)hipEventElapsedTime(&ms, e0, e1);
            size_t nodes = 0; (void)hipGraphGetNodes(g, nullptr, &nodes);
            fprintf(stderr, "[idkpt graph] batch of %d sample(s) captured: %zu graph nodes; %d replays: %.1f us per batch\n", B, nodes, K, ms * 1000.0f / (float)K);
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipGraphExecDestroy(ex);
        } else fprintf(stderr, "[idkpt graph] capture or instantiation failed: %s\n", hipGetErrorString(hipGetLastError()));
        if (g) (void)hipGraphDestroy(g);
    }
#endif
    // queue lengths stay on the GPU during the batch; k_scan_blocks mirrors them into host-mapped memory for GetStats and the queue downloads (no copy, no sync here)
    if (ctx->timing) HIPC(hipEventRecord(ctx->evFrame[1], st));
    ctx->stats.Frames += (uint64_t)B;
    ctx->stats.PrimaryRays += (uint64_t)N * (uint64_t)B;
    ctx->pending.clear();
    return IDKPT_OK;
}

static int32_t dev_Render(dev_ctx* ctx)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    if (!ctx->haveScene) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptRender: no scene uploaded");
    if (ctx->W <= 0 || !ctx->frameOk) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptRender: no frame buffers (idkptSetSize not called, or its allocation failed)");
    if (ctx->st.UseTlas && ctx->tlasCount == 0) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptRender: UseTlas set but no TLAS nodes uploaded");
    HIPC(hipSetDevice(ctx->device));
    if (ctx->timing && ctx->evUsed > 4096) { HIPC(hipStreamSynchronize(ctx->stream)); resolve_trace_events(ctx); }
    // every sample is deferred; a batch is launched as soon as maxBatch samples are pending (or on any call that needs
    // results).  The general path (multi-instance / TLAS / debug cost) is launched sample by sample.
    const int limit = fast_path(ctx) ? ctx->maxBatch : 1;
    for (int i = 0; i < ctx->st.SamplesPerPixel; i++) {
        PendingSample ps; ps.accum = ctx->accum[ctx->curSlot]++; ps.slot = ctx->curSlot;
        memcpy(ps.cam, ctx->invProj, 64); memcpy(ps.cam + 16, ctx->invView, 64); memcpy(ps.cam + 32, ctx->viewPos, 12); ps.cam[35] = 0.0f;
        for (int b = 0; b < VB_COUNT; b++) ps.vs[b] = (uint8_t)ctx->vcur[b];           // the state of the geometry this sample sees
        ctx->pending.push_back(ps);
        if (!ctx->grouped && (int)ctx->pending.size() >= limit) { int rc = flush_batch(ctx); if (rc) return rc; }   // (members of a multi-device context: the group launches)
    }
    return IDKPT_OK;
}

static int32_t dev_Synchronize(dev_ctx* ctx) { if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT; HIPC(hipSetDevice(ctx->device)); FLUSH_KEEP(); SYNC_CHECKED(); return IDKPT_OK; }

// Launches whatever is pending without waiting for it (lets a host overlap its own work with the GPU).
static int32_t dev_Flush(dev_ctx* ctx) { if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT; HIPC(hipSetDevice(ctx->device)); FLUSH_KEEP(); return IDKPT_OK; }

static int32_t dev_SetMaxBatch(dev_ctx* ctx, int32_t maxBatch)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(maxBatch >= 1 && maxBatch <= MAX_BATCH, "idkptSetMaxBatch: 1..256");
    HIPC(hipSetDevice(ctx->device));
    FLUSH();   // (the wavefront buffers are about to be reallocated: a deferred last bounce is completed first)
    HIPC(hipStreamSynchronize(ctx->stream));
    if (maxBatch == ctx->maxBatch) return IDKPT_OK;
    const int previous = ctx->maxBatch;
    ctx->maxBatch = maxBatch;
    if (ctx->W > 0) {
        std::vector<uint32_t> acc = ctx->accum; int slot = ctx->curSlot; const bool started = ctx->ringStarted;
        int rc = alloc_frame_keep_images(ctx);
        if (rc) {   // e.g. out of device memory: fall back to the previous (smaller) buffer set; the accumulation restarts
            const std::string why = ctx->lastError;
            ctx->maxBatch = previous;
            (void)alloc_frame(ctx);
            return fail(ctx, rc, "idkptSetMaxBatch: could not allocate the wavefront buffers for " + std::to_string(maxBatch) + " samples in flight (" + why + "); kept " + std::to_string(previous));
        }
        ctx->accum = acc; ctx->curSlot = slot; ctx->ringStarted = started;
    }
    return IDKPT_OK;
}

// idkptSetSceneVersions: how many states of the geometry may be in flight (1: a scene update launches every queued sample first, as the reference's frame loop does)
static int32_t dev_SetSceneVersions(dev_ctx* ctx, int32_t versions)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(versions >= 1 && versions <= 64, "idkptSetSceneVersions: 1..64 versions");
    HIPC(hipSetDevice(ctx->device));
    if (versions == ctx->verSlots) return IDKPT_OK;
    FLUSH();                                                           // nothing queued or deferred: every buffer has exactly one live state, its current one
    if (versions < ctx->verSlots) {
        for (int b = 0; b < VB_COUNT; b++) {
            if (ctx->vcur[b] >= versions && ctx->vbytes[b] > 0) { HIPC(hipMemcpyAsync(vb_ptr(ctx, b, 0), vb_ptr(ctx, b, ctx->vcur[b]), ctx->vbytes[b], hipMemcpyDeviceToDevice, ctx->stream)); ctx->vcur[b] = 0; }
            ctx->valloc[b] = std::min(ctx->valloc[b], versions);
        }
        HIPC(hipStreamSynchronize(ctx->stream));
    }
    ctx->verSlots = versions;
    return IDKPT_OK;
}

static int32_t dev_SetFrameRing(dev_ctx* ctx, int32_t frames)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(frames >= 1 && frames <= 128, "idkptSetFrameRing: 1..128 frames");
    HIPC(hipSetDevice(ctx->device));
    FLUSH();   // (the wavefront buffers are about to be reallocated: a deferred last bounce is completed first)
    HIPC(hipStreamSynchronize(ctx->stream));
    if (frames == ctx->ringSize) return IDKPT_OK;
    ctx->ringSize = frames;
    if (ctx->W > 0) return alloc_frame(ctx);       // images are re-created (cleared); accumulation restarts in slot 0
    ctx->accum.assign(frames, 0u); ctx->curSlot = 0; ctx->ringStarted = false;
    return IDKPT_OK;
}

static int32_t dev_BeginFrame(dev_ctx* ctx, int32_t* outSlot)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    if (ctx->ringStarted) ctx->curSlot = (ctx->curSlot + 1) % ctx->ringSize;   // the first frame after idkptSetFrameRing / idkptSetSize uses slot 0
    ctx->ringStarted = true;
    ctx->accum[ctx->curSlot] = 0;                   // a new frame: its first sample overwrites whatever the slot held
    if (outSlot) *outSlot = ctx->curSlot;
    return IDKPT_OK;
}

static int32_t dev_DownloadFrame(dev_ctx* ctx, int32_t slot, int32_t image, float* rgba, size_t bytes)
{
    if (!ctx || !rgba) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(image >= 0 && image < 3, "idkptDownloadFrame: bad image id");
    REQUIRE(slot >= 0 && slot < ctx->ringSize, "idkptDownloadFrame: slot outside the frame ring");
    size_t need = (size_t)ctx->W * ctx->rows * 16;
    REQUIRE(bytes == need && need > 0, "idkptDownloadFrame: bytes must equal localRows*width*16");
    HIPC(hipSetDevice(ctx->device));
    FLUSH_KEEP();
    HIPC(hipMemcpyAsync(rgba, image_ptr(ctx, image, slot), need, hipMemcpyDeviceToHost, ctx->stream));
    SYNC_CHECKED();
    return IDKPT_OK;
}

static int32_t dev_GetFrameDevicePtr(dev_ctx* ctx, int32_t slot, int32_t image, void** outPtr, size_t* outBytes)
{
    if (!ctx || !outPtr) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(image >= 0 && image < 3, "idkptGetFrameDevicePtr: bad image id");
    REQUIRE(slot >= 0 && slot < ctx->ringSize, "idkptGetFrameDevicePtr: slot outside the frame ring");
    HIPC(hipSetDevice(ctx->device));
    FLUSH_KEEP();                                        // launches what is still deferred (stream-ordered: a consumer on the context's stream sees the finished image)
    { int rc = check_overflow(ctx); if (rc) return rc; }   // (no wait: reports an overflow of batches that have already finished; a zero-copy consumer sees the rest at its next idkptSynchronize)
    *outPtr = image_ptr(ctx, image, slot);
    if (outBytes) *outBytes = (size_t)ctx->W * ctx->rows * 16;
    return IDKPT_OK;
}

static int32_t dev_Download(dev_ctx* ctx, int32_t image, float* rgba, size_t bytes)
{
    if (!ctx || !rgba) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(image >= 0 && image < 3, "idkptDownload: bad image id");
    size_t need = (size_t)ctx->W * ctx->rows * 16;
    REQUIRE(bytes == need && need > 0, "idkptDownload: bytes must equal localRows*width*16");
    HIPC(hipSetDevice(ctx->device));
    FLUSH_KEEP();
    HIPC(hipMemcpyAsync(rgba, image_ptr(ctx, image, ctx->curSlot), need, hipMemcpyDeviceToHost, ctx->stream));
    SYNC_CHECKED();
    return IDKPT_OK;
}

static int32_t dev_DownloadRays(dev_ctx* ctx, GpuWavefrontRay* out, size_t bytes)
{
    if (!ctx || !out) return IDKPT_ERR_INVALID_ARGUMENT;
    size_t N = (size_t)ctx->W * ctx->rows;
    REQUIRE(bytes == N * sizeof(GpuWavefrontRay) && N > 0, "idkptDownloadRays: bytes must equal pixelCount*48");
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    const size_t off = (size_t)(ctx->lastBatch - 1) * ctx->Npad * 16; // the most recent sample of the last batch
    { int rc = materialize_culled_rays(ctx); if (rc) return rc; }   // complete what the ray generation left out for pre-culled pixels
    std::vector<float4> a(N), b(N), c(N);
    HIPC(hipMemcpyAsync(a.data(), (char*)ctx->rayO.p + off, N * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipMemcpyAsync(b.data(), (char*)ctx->rayT.p + off, N * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipMemcpyAsync(c.data(), (char*)ctx->rayR.p + off, N * 16, hipMemcpyDeviceToHost, ctx->stream));
    SYNC_CHECKED();
    for (size_t i = 0; i < N; i++) {
        GpuWavefrontRay& r = out[i];
        r.Origin[0] = a[i].x; r.Origin[1] = a[i].y; r.Origin[2] = a[i].z; r.PreviousIOROrTraverseCost = a[i].w;
        r.Throughput[0] = b[i].x; r.Throughput[1] = b[i].y; r.Throughput[2] = b[i].z; r.PackedDirectionX = b[i].w;
        r.Radiance[0] = c[i].x; r.Radiance[1] = c[i].y; r.Radiance[2] = c[i].z; r.PackedDirectionY = c[i].w;
    }
    return IDKPT_OK;
}

static int32_t dev_DownloadAliveQueue(dev_ctx* ctx, uint32_t* indices, size_t capacity, uint32_t* outCount)
{
    if (!ctx || !outCount) return IDKPT_ERR_INVALID_ARGUMENT;
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    HIPC(hipStreamSynchronize(ctx->stream));
    // the most recent sample's segment of the batch-wide queue; entries are ray ids -> subtract the sample's id offset
    const uint32_t* hb = ctx->hBases + (size_t)ctx->lastQueueCountSlot * (MAX_BATCH + 1);
    const uint32_t first = hb[ctx->lastBatch - 1], n = hb[ctx->lastBatch] - first;
    *outCount = n;
    if (indices && n) {
        REQUIRE(capacity >= n, "idkptDownloadAliveQueue: capacity too small");
        HIPC(hipMemcpyAsync(indices, ctx->queue[ctx->lastQueueSide].as<uint32_t>() + first, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream)); HIPC(hipStreamSynchronize(ctx->stream));
        const uint32_t sub = (uint32_t)(ctx->lastBatch - 1) * ctx->Npad;
        for (uint32_t i = 0; i < n; i++) indices[i] -= sub;
    }
    return IDKPT_OK;
}

static int32_t dev_EnablePrimaryHitCapture(dev_ctx* ctx, int32_t enable) { if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT; ctx->capturePrimary = enable != 0; return IDKPT_OK; }

static int32_t dev_DownloadPrimaryHits(dev_ctx* ctx, float* t, uint32_t* triangleId, float* baryXY, size_t pixelCount)
{
    if (!ctx || !t || !triangleId || !baryXY) return IDKPT_ERR_INVALID_ARGUMENT;
    size_t N = (size_t)ctx->W * ctx->rows;
    REQUIRE(pixelCount == N, "idkptDownloadPrimaryHits: pixelCount mismatch");
    HIPC(hipSetDevice(ctx->device));
    FLUSH_KEEP();
    if (!ctx->capturePrimary || ctx->primHit.bytes < N * 16) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptDownloadPrimaryHits: call idkptEnablePrimaryHitCapture(ctx,1) before idkptRender");
    std::vector<float4> h(N);
    HIPC(hipMemcpyAsync(h.data(), ctx->primHit.p, N * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < N; i++) { t[i] = h[i].x; baryXY[2 * i] = h[i].y; baryXY[2 * i + 1] = h[i].z; memcpy(&triangleId[i], &h[i].w, 4); }
    return IDKPT_OK;
}

static int32_t dev_GetStats(dev_ctx* ctx, idkpt_stats* out)
{
    if (!ctx || !out) return IDKPT_ERR_INVALID_ARGUMENT;
    HIPC(hipSetDevice(ctx->device));
    FLUSH_KEEP();
    SYNC_CHECKED();
    idkpt_stats s = ctx->stats;
    for (int j = 0; j < 16; j++) { const uint32_t* hb = ctx->hBases + (size_t)j * (MAX_BATCH + 1); s.LastAliveCounts[j] = (j >= 1 && j < ctx->st.RayDepth) ? hb[ctx->lastBatch] - hb[ctx->lastBatch - 1] : 0; }
    // [0]: primary rays that entered the traversal kernel (all pixels, or the survivors of the root-box pre-cull on the fast path)
    s.LastAliveCounts[0] = s.Frames ? (ctx->lastFast ? ctx->hCounts[MAX_DEPTH_SLOTS - 1] : (uint32_t)((size_t)ctx->W * ctx->rows)) : 0;
    s.LastFrameMs = 0.0f; s.LastTraceMs = 0.0f;
    if (ctx->timing && s.Frames > 0) { float ms = 0.0f; if (hipEventElapsedTime(&ms, ctx->evFrame[0], ctx->evFrame[1]) == hipSuccess) s.LastFrameMs = ms; }
    resolve_trace_events(ctx);
    s.TraceMsTotal = ctx->traceMsAcc; s.TraceLaunches = ctx->traceLaunchesAcc;
    s.LastTraceMs = s.TraceLaunches ? (float)(s.TraceMsTotal / (double)s.TraceLaunches) : 0.0f;
    uint64_t c[4] = {0, 0, 0, 0};
    HIPC(hipMemcpyAsync(c, ctx->counters64.p, 32, hipMemcpyDeviceToHost, ctx->stream)); HIPC(hipStreamSynchronize(ctx->stream));
    s.NodePairVisits = c[0]; s.TriangleTests = c[1];
    if (ctx->opt.traceVariant == 107 || ctx->opt.traceVariant == 113 || ctx->opt.traceVariant == 116 || ctx->opt.traceVariant == 213) { uint64_t d[16]; HIPC(hipMemcpyAsync(d, ctx->counters64.p, 128, hipMemcpyDeviceToHost, ctx->stream)); HIPC(hipStreamSynchronize(ctx->stream)); fprintf(stderr, "[idkpt prof] cycles refill %llu node %llu leaf %llu other %llu | refills %llu lanes %llu | nodeSteps %llu lanes %llu | leafPhases %llu lanes %llu | leafTests %llu leafTrips %llu\n", (unsigned long long)d[4], (unsigned long long)d[5], (unsigned long long)d[6], (unsigned long long)d[7], (unsigned long long)d[8], (unsigned long long)d[9], (unsigned long long)d[10], (unsigned long long)d[11], (unsigned long long)d[12], (unsigned long long)d[13], (unsigned long long)d[14], (unsigned long long)d[15]); }
    s.RaysTraced = s.PrimaryRays + c[2]; // N per sample + every alive-queue entry that entered a bounce
    if (ctx->wtotals.p) { uint64_t w[4] = {0, 0, 0, 0}; HIPC(hipMemcpyAsync(w, ctx->wtotals.p, 32, hipMemcpyDeviceToHost, ctx->stream)); HIPC(hipStreamSynchronize(ctx->stream)); s.WideFlaggedRays = w[0]; s.WideNodeVisits = w[1]; s.WideLeafRecords = w[2]; s.WideTriangleTests = w[3]; }
    *out = s;
    return IDKPT_OK;
}

static int32_t dev_ResetStats(dev_ctx* ctx)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    HIPC(hipSetDevice(ctx->device));
    FLUSH_KEEP();
    HIPC(hipStreamSynchronize(ctx->stream));
    memset(&ctx->stats, 0, sizeof(ctx->stats));
    ctx->evUsed = 0; ctx->traceMsAcc = 0.0; ctx->traceLaunchesAcc = 0;
    memset(ctx->hCounts, 0, (MAX_DEPTH_SLOTS - 1) * 4);     // (the last word, the length of the primary active list, is also the grid hint of the next batch: idkptGetStats reports it only once frames were rendered)
    HIPC(hipMemsetAsync(ctx->counters64.p, 0, 128, ctx->stream)); if (ctx->wtotals.p) HIPC(hipMemsetAsync(ctx->wtotals.p, 0, 64, ctx->stream)); HIPC(hipStreamSynchronize(ctx->stream));
    return IDKPT_OK;
}

static int32_t dev_EnableCounters(dev_ctx* ctx, int32_t enable) { if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT; ctx->counters = enable != 0; return IDKPT_OK; }
static int32_t dev_EnableTiming(dev_ctx* ctx, int32_t enable) { if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT; ctx->timing = enable != 0; return IDKPT_OK; }

static int32_t dev_GetImageDevicePtr(dev_ctx* ctx, int32_t image, void** outPtr, size_t* outBytes)
{
    if (!ctx || !outPtr) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(image >= 0 && image < 3 && ctx->W > 0, "idkptGetImageDevicePtr: bad image / no size");
    HIPC(hipSetDevice(ctx->device));
    FLUSH_KEEP();                                        // launches what is still deferred
    { int rc = check_overflow(ctx); if (rc) return rc; }   // (no wait: see idkptGetFrameDevicePtr)
    *outPtr = image_ptr(ctx, image, ctx->curSlot);
    if (outBytes) *outBytes = (size_t)ctx->W * ctx->rows * 16;
    return IDKPT_OK;
}

static int32_t dev_SetStream(dev_ctx* ctx, void* hipStream)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    HIPC(hipStreamSynchronize(ctx->stream));
    if (hipStream) { if (ctx->ownStream && ctx->stream) (void)hipStreamDestroy(ctx->stream); ctx->stream = (hipStream_t)hipStream; ctx->ownStream = false; }
    else if (!ctx->ownStream) { HIPC(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)); ctx->ownStream = true; }
    return IDKPT_OK;
}
static int32_t dev_GetStream(dev_ctx* ctx, void** out) { if (!ctx || !out) return IDKPT_ERR_INVALID_ARGUMENT; *out = (void*)ctx->stream; return IDKPT_OK; }


#include "idkpt_api.hpp"
