// host_context.hpp — buffers, options and state of one device's context (dev_ctx); error plumbing.
// Part of the single translation unit idkpt.hip (included there, in this order).
#pragma once


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
    int traceVariant = 0;        // IDKPT_DEVELOPER builds only: instrumented / probe instantiations of k_trace2
    int bvhTiming = 0, bvhSmall = 32;   // idkptBuildBlasCore: phase times on stderr; subtrees of at most this many fragments are finished by one thread
    int advMin = 0;              // k_trace2 MODE 1-4: instance entries / TLAS steps are taken by at least this many lanes together (or when no other lane has work); 0 = 8 for batches of >= 4 samples, 1 below (measured, 3-BLAS soup-1M: 1 / 8 / 16 / 24 / 32 = 3 365 / 3 408 / 3 226 / 2 925 / 2 568 Mray/s through the instance loop, 2 671 / 2 771 / 2 671 / 2 466 / 2 180 through the TLAS; one frame at a time 8 costs 1-6 %)
    int poolMin = 12;            // pooled leaf phase: pairs a wave must have parked (below: every lane walks its own triangles as before; 0-20 measure the same, 32+ lose the gain)
    int leafPool = -1;           // k_trace2<.., DBG = 16> (MODE 0): the leaf phase tests the wave's pooled (ray, triangle) pairs with all lanes in one round trip (kernels_trace.hpp).
                                 // Mask of launch kinds: 1 = primary launches, 2 = the first bounce, 4 = later bounces.  -1 (default) = by measurement (profiles/r04_leaf_pool.md): later
                                 // bounces always lose (few pairs per phase: -4 %), the first bounce gains 3-5 % where most pixels traverse the scene and loses 1 % on sparse views,
                                 // the primary launch gains 4 % on sparse views and nothing elsewhere -> 1 on sparse views (fewer than half of the pixels enter the traversal), 3 otherwise
    int splitScatter = 6;        // k_trace2s: log2 of the entries that stay together when the work list is handed out scattered (6 = list order)
    int queryScheduler = 1;      // idkptTraceRays (closest hit) through k_trace2's scheduler instead of the thread-per-ray kernel (kernels_query.hpp)
    int groupThreads = -1;       // multi-device contexts (idkpt_api.hpp group_flush): members' batches enqueued by one host thread each also where no exchange needs it (-1: from 4 members on)
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
    int genPixelMajor = 8;       // k_gen_primary: batches of at least this many samples append their primary rays pixel by pixel (a traversal wave = 4 pixels x 16 samples — rays that differ by their sub-pixel
                                 // jitter — instead of one 8x8 tile of one sample); 0 = never.  Measured (profiles/r05_pixel_major.md): interior view +5 %, RayDepth 5 +2 %, atrium +1 %, headline +-0
    int bouncePixelMajor = 1;    // ... and the first bounce is traced in that order too (k_shade_first appends the continuing rays to a list of its own, their hits are stored per ray id):
                                 // 0 never, 1 on sparse views (measured: +1.1-2.2 % there, -2 % where every pixel traverses), 2 always
    int genGroupMax = 16;        // ... samples per group (= waves per workgroup of k_gen_primary), at most 16.  Mean trace launch, interior view: 2 / 4 / 8 / 16 samples 33.5 / 32.8 / 32.4 / 31.8 ms (tile-major 33.9);
                                 // 32 (two passes per wave) 31.5 ms, but the two-pass kernel itself cost the headline view 1 %: not kept
    int instSieve = 8;           // k_trace_inst<P, EXACT>: scenes of at least this many instances (up to 1024) that keep the instance loop run it with the instances a ray cannot meet sieved out up front
                                 // (the kernel that serves the own-TLAS walk's flagged rays, as the main kernel): exact by construction, visit for visit.  0 = k_trace2 MODE 1.
    int instSieveOverlap = 50;   // ... while a random line meets at most this many percent of the instances' boxes.  Measured against k_trace2 MODE 1 (profiles/r05_instance_tlas.md): the atrium's 87
                                 // meshes 2.2x; 64 clusters at 10 / 20 / 30 % overlap 1.68x / 1.44x / 1.19x from outside, 1.30x / 0.88x / 0.85x from inside; soup-1M in 60 parts (37 %) 1.06x / 1.12x,
                                 // in 12 parts (72 %) 1.03x / 0.98x
    int instTlas = 8;            // k_trace_inst (kernels_trace_inst.hpp): scenes of at least this many BLAS instances rendered WITHOUT UseTlas (the reference's instance loop, its default) are walked
                                 // through a TLAS the library builds for itself; rays whose result could depend on the loop's order are traced again by the exact loop.  0 = the loop only.
                                 // Measured (round 5, profiles/r05_instance_tlas.md): the atrium as 87 BLASes 453 -> 1 641 Mray/s (3.6x; one frame at a time 363 -> 1 029), 0.3 % of the
                                 // rays traced again.  Bit-identical hits either way (tests/test_gpu_inst_tlas.py, tools/fuzz_parity.py).
    int instTlasOverlap = 10;    // ... only where the instances' boxes overlap little: a random line through the scene meets at most this many PERCENT of them (k_tlas_build measures it).
                                 // 64 clusters at 5 / 10 / 20 / 30 %: the tree is 2.2x / 1.7x / 1.1x / 0.8x the loop seen from outside and 1.6x / 0.77x / 0.7x / 0.7x seen from inside;
                                 // soup-1M in 12 / 60 interleaved parts (72 / 37 %): 0.73-0.89x / 0.87-1.03x; the atrium's 87 meshes: 2.7 %.  100 = whatever the overlap
    int instBraid = 0;           // ... and the tree's leaves are SUBTREES of the instances' BLASes (partial re-braiding, k_braid in kernels_scene.hpp): the entries with the largest boxes are opened
                                 // into their node's children until the list has this many entries (0 = whole instances only).  Scenes whose BLAS boxes nest (dev_ctx::sceneNested).
                                 // Measured (profiles/r06_braid.md): the walk's TLAS phase pays two dependent fetches per step and a RayTransform per entry, so more and smaller entries LOSE
                                 // (atrium-87 2 018 -> 1 746 Mray/s at 2 048 entries, 3-part soup 2 721 -> 1 467): default 0; the idea pays where an entry costs nothing, inst_unify below
    int instUnify = 4096;        // k_trace_inst<.., UNI>: scenes of >= 2 instances that all carry the SAME InvModel (bit for bit) and use every BLAS at most once — the reference's usual static scene —
                                 // walk ONE tree in their common BLAS space: a PLOC top over this many subtrees of the BLASes at most (k_braid), the BLASes' own nodes below (k_unify_*,
                                 // kernels_scene.hpp); flagged rays go to the exact loop as with the own TLAS.  0 = off.  Measured: profiles/r06_braid.md
    int instGeneral = 0;         // k_trace_inst<.., TREE 2>: scenes of at least this many instances that are NOT one space (different transforms, or a BLAS used several times) walk the
                                 // general array — a world-space top over subtrees, entries take the ray into their instance's space (kernels_scene.hpp k_unify_top, general).  0 = off
                                 // (default).  Measured (profiles/r06_braid.md §5): with whole instances as entries it equals the own TLAS (2 754 vs 2 675 Mray/s on the 3 rotated
                                 // soups, the loop: 3 628); with subtrees as entries it LOSES (4 096 entries: 2 182 | interior 428 vs 722): the world box of a rotated subtree is loose
                                 // and every entry met costs a RayTransform, a stub step and a RESTORE step.  The idea pays only where an entry is free: one space (inst_unify)
    int uniRefill = 32;          // ... idle lanes at which a wave of the unified walk takes new rays (16 or 32; measured: 32 is +0.6 % batched, +2 % one frame at a time on the atrium as 87 BLASes)
    int instUnifyRadius = 15;    // ... PLOC search radius of that top (TLAS.cs's SearchRadius is 15: the own TLAS keeps the reference's)
    int pairNodes = 1;           // k_trace2 FAST: one-BLAS closest-hit launches (one scene version, no counters) step on DScene::pairNodes, the sibling pairs regrouped for 2-wide
                                 // arithmetic (+ 64 bytes per pair of device memory, re-derived after node updates): 62 -> 53 vector instructions per node step.  0 = the reference's layout
    int packet = 1;              // k_trace_packet (kernels_packet.hpp): primary launches of one-BLAS scenes whose work list is pixel-major (batches of >= gen_pixel_major samples) walk the BVH2 as
                                 // packets — one shared walk per wave, node pairs through the scalar cache; rays it cannot vouch for are re-traced by k_trace2.  0 = never, 1 (default) = where the
                                 // kernel's own counters say the wave's rays want the same nodes (packet_decide: live lanes per node step), 2 = every primary launch of a one-BLAS scene
                                 // (whatever the list's order: the tests' setting).  Bit-identical results either way (tests/test_gpu_packet.py).
    int packetMinLive = 60;      // ... percent of a wave's lanes that must be live in an average node step for the packet walk to stay on (tools/packet_sim.cpp: break-even against k_trace2's
                                 // 39 live lanes of 64 is 61 %; measured: profiles/r06_packet.md)
    int packetWaves = 0;         // ... one-wave workgroups per CU of its persistent grid (0: 28 — 7 waves per SIMD at its 68 VGPRs)
    int bvhStackOptHost = 0;            // idkptBuildBlas: take OptimizeStackSize's decisions from the reference's own walk on a host copy (the fallback path, forced: for its test)
};

// Scene versions (idkptSetSceneVersions): the buffers the render kernels read of the geometry an animated frame rewrites.  Each may hold several states
// ("slots" of one arena); a queued sample remembers the slots that were current when it was queued (PendingSample::vs), so frames with different geometry can
// be traced by one batch while later updates already write other slots (ver_writable).
enum { VB_NODES = 0, VB_TRIVERTS, VB_VERTICES, VB_TLAS, VB_XFORMS, VB_COUNT };
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
    DevBuf nodes, tris, triVerts, descs, instances, tlas, parents, leaves, positions, prevPositions, vertices, meshes, materials, xforms, lights, sky, texDescs, unskinned, joints, levelNodes, tlasScratch, queryIn, queryOut, queryRec, queryList, tileClass, gbases;   // (+ camTab below)
    std::vector<DevBuf> texData; std::vector<std::pair<int, int>> texDims; std::vector<uint32_t> texState;   // per image: device copy, (width, height), TexDesc::state
    DevBuf srgbLut;                                      // DScene::srgbLut (256 floats), made with the first scene
    std::vector<GpuBlasDesc> hDescs;
    std::vector<std::vector<uint32_t>> levelOffsets; // per BLAS: offsets into levelNodes (level l occupies [off[l], off[l+1]))
    std::vector<uint32_t> levelBase;                 // per BLAS base into levelNodes
    std::vector<char> refitCoversAll;                // per BLAS: leaves + internal nodes of the refit schedule are every node but node 0 (a refit into a fresh slot needs no copy of the old nodes)
    int nodeCount = 0, triCount = 0, instanceCount = 0, tlasCount = 0, vertexCount = 0, meshCount = 0, materialCount = 0, xformCount = 0, lightCount = 0, skySize = 0, textureCount = 0, unskinnedCount = 0;
    int sceneStack = 1;
    int hInst0Blas = 0;                              // BlasId of instance 0 (MODE 0 traverses that BLAS)
    // wide nodes (kernels_wide.hpp): derived per BLAS from nodes + triVerts.  wideTopoValid: the children lists match the node topology; wideFillValid: boxes / leaf records match the current boxes and positions
    DevBuf wnodes, wleaf, wids, wpair, wcounts, wtotals; std::vector<uint32_t> wNodeOff, wLeafOff; bool wideTopoValid = false, wideFillValid = false;
    // the library's own TLAS for the instance loop (kernels_trace_inst.hpp): padded PLOC tree over the instances, and the per-triangle "not contained in its leaf box" marks;
    // both derived on the device before the first batch that wants them and after everything that moves boxes, positions or transforms
    DevBuf pairNodes; bool pairValid = false;             // DScene::pairNodes (k_pair_nodes): re-derived after everything that changes node boxes or topology
    DevBuf instRec; bool instRecValid = false;            // DScene::instRec (k_inst_records): one scene version only; re-derived with the own TLAS's triggers
    DevBuf itlas, imarks, ichunks; int itlasNeed = 1; uint32_t ichunkCount = 0; bool itlasValid = false, imarksValid = false;
    int uniMode = 0; uint32_t uniBaseB = 0, uniRestoreIdx = 0;   // uniMode: 1 = the unified tree of a same-space scene (TREE 1), 2 = the general array (TREE 2: world-space top, entries switch spaces)
    DevBuf unodes, utlas, uTabs, uniBuf, uniEntRec; bool uniValid = false, uniEligible = false, uniTabsValid = false; int uniCap = 0, uniEntries = 0, uniDepth = 0; float* hUni = nullptr; float* dUni = nullptr;   // the unified tree (k_unify_*): nodes, its PLOC top, the per-BLAS tables; host-mapped: [1] entries, [2] depth of the top
    std::vector<GpuBlasInstance> hInstances; std::vector<char> hXforms; uint64_t uniLaunches = 0;   // host copies of the instance list and of the GpuMeshTransforms as last uploaded / patched
    DevBuf entRec, braidBuf; bool itlasBraided = false; int itlasDepth = 0, itlasEntries = 0;   // k_braid's entry records / (entries, areas, leaf boxes, count); the built tree's depth and leaf count as last read from the device (0: not known)
    // the packet walk's decision (host_launch.hpp packet_decide): 0 = probing (packets on, counters awaited), 1 = on, 2 = off; the kernel's counters arrive through host-mapped memory
    // one or two batches late (k_packet_mirror): [1] packets, [2] node steps, [3] live lanes, [4] rays entered, [5] triangle rounds — totals since the last reset
    int pkState = 0, pkBatchesSinceProbe = 0; unsigned long long* hPkStats = nullptr; unsigned long long* dPkStats = nullptr; unsigned long long pkSeen[6] = {0, 0, 0, 0, 0, 0}; float pkCam[36] = {0}; int pkW = 0, pkRows = 0, pkBatch = 0; float pkLastLive = -1.0f;
    float* hInstOverlap = nullptr; float* dInstOverlap = nullptr; bool instOverlapKnown = false, itlasBuilt = false, isieveWorth = false;   // host-mapped: instance boxes a random line meets (k_tlas_build); known = read at least once since the upload
    // scene versions: slot count a versioned buffer may grow to, per buffer the bytes of one state / the slot pitch / the slots its arena holds / the current slot
    int verSlots = 1; size_t vbytes[VB_COUNT] = {0}, vstride[VB_COUNT] = {0}; int valloc[VB_COUNT] = {1, 1, 1, 1, 1}, vcur[VB_COUNT] = {0};
    uint64_t lastMask[VB_COUNT] = {0};               // slots the last launched batch reads (a deferred last bounce still does: finish_deferred)
    uint8_t lastSlots[VB_COUNT] = {0}; bool lastMulti = false;   // ... the one set of slots of a single-version batch, or "per sample: verTab"
    DevBuf verTab; uint32_t* hVerTab = nullptr; hipEvent_t evVer[2] = {nullptr, nullptr}; int verHalf = 0;   // per-sample version table of the batch being launched (pinned, double-buffered staging)
    char* hStage = nullptr; hipEvent_t evStage[4] = {nullptr, nullptr, nullptr, nullptr}; int stageNext = 0;   // pinned ring for small host -> device updates (joint matrices, transforms): no stream synchronisation per call
    int (*groupFlushAll)(void* user) = nullptr;      // member of a multi-device context: launches what ALL members have queued (a member never flushes on its own)
    // wavefront state
    DevBuf pmList;                                       // the first bounce's work list in pixel-major order (k_shade_first)
    DevBuf trRec, contFlag, blockSums, rayO, rayT, rayR, aovA, aovN, hit, hitCost, primHit, queue[2], keys[2], keysTmp, sortKeys, sortVals, contMask, waveCounts, counts, work, sortHist, counters64;
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
