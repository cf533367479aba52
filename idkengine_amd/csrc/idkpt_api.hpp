// idkpt_api.hpp — the exported C-ABI of include/idkpt.h and the multi-device ("group") layer behind idkptCreate(deviceCount = N).
// Included at the end of idkpt.hip (one translation unit); the single-device implementation is the dev_* functions above.
//
// One handle, N GPUs (north_star: "tiles of the framebuffer optionally sharded across the 8 GPUs of one node ... broadcast of the BVH +
// gather of tiles over xGMI"; the reference itself is single-GPU, Source/EntryPoint.cs:10-33):
//   * every device gets a member context (dev_ctx: own stream, own wavefront buffers, its rows of the frame);
//   * the scene crosses PCIe once (member 0) and is replicated device-to-device: one ncclBroadcast per scene buffer where RCCL can form the
//     communicators (transport_rccl.hpp, dlopen), otherwise hipMemcpyPeerAsync — point-to-point copies are what xGMI is;
//   * rows: interleaved (y % N == d: balances sky rows against geometry rows; N-device output == 1-device output bit for bit at RayDepth <= 2,
//     where radiance does not depend on the queue slot) or contiguous strips with a device-side exchange of the per-sample alive counts
//     at every bounce (exact at any depth with sorting off: NHit seeds its RNG from the queue slot, NHit/compute.glsl:54) — "auto" picks
//     by RayDepth; no host synchronisation in either mode: members wait on each other's per-bounce events, counts travel by peer copies.
//     Interleaved rows / bands chosen explicitly stay exact beyond RayDepth 2 too: one host thread per member, meeting at every bounce (GroupBarrier below);
//   * results: idkptDownload writes every member's rows straight into the host image (N PCIe links in parallel); idkptGetImageDevicePtr
//     gathers the rows into a full frame on device 0 (grouped ncclSend / ncclRecv or peer copies + one interleave kernel).
// With deviceCount == 1 every entry point forwards to the single member: no behavioural change, no overhead.
#pragma once
#include <array>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>

// Interleaved rows / bands beyond RayDepth 2 (round 4): a band's slot base counts the alive rays of ALL members in the image bands before it — also of members that are
// enqueued later — so the members' batches cannot be enqueued one after the other as the strips' are.  They are enqueued by one host thread per member, and the
// per-bounce exchange is the single-context one (idkptSetBandExchange: dev_ctx::bandExchangeFn) with the members meeting at this barrier.
struct GroupBarrier {
    std::mutex m; std::condition_variable cv; int n = 0, arrived = 0; unsigned gen = 0; bool aborted = false;
    void reset(int members) { std::lock_guard<std::mutex> l(m); n = members; arrived = 0; aborted = false; }
    bool wait()
    {
        std::unique_lock<std::mutex> l(m);
        if (aborted) return false;
        const unsigned g = gen;
        if (++arrived == n) { arrived = 0; gen++; cv.notify_all(); return true; }
        cv.wait(l, [&] { return gen != g || aborted; });
        return !aborted;
    }
    void abort() { std::lock_guard<std::mutex> l(m); aborted = true; cv.notify_all(); }
};
// One host thread per member, kept for the lifetime of the context (created at the first threaded flush): a flush hands every worker its member's batch and waits
// for all of them — no thread creation per flush (a deep-path frame flushes once per batch of samples; round 4 spawned and joined N threads each time).
struct GroupWorkers {
    std::vector<std::thread> th;
    std::mutex m; std::condition_variable cvGo, cvDone;
    unsigned epoch = 0; int remaining = 0; bool quit = false;
    std::function<void(size_t)> job;
    void start(size_t n)
    {
        if (!th.empty()) return;
        for (size_t d = 0; d < n; d++)
            th.emplace_back([this, d]() {
                unsigned seen = 0;
                for (;;) {
                    std::function<void(size_t)> j;
                    { std::unique_lock<std::mutex> l(m); cvGo.wait(l, [&] { return quit || epoch != seen; }); if (quit) return; seen = epoch; j = job; }
                    j(d);
                    { std::lock_guard<std::mutex> l(m); if (--remaining == 0) cvDone.notify_all(); }
                }
            });
    }
    void run(const std::function<void(size_t)>& j)       // every worker runs j(its index); returns when all are done
    {
        std::unique_lock<std::mutex> l(m);
        job = j; remaining = (int)th.size(); epoch++;
        cvGo.notify_all();
        cvDone.wait(l, [&] { return remaining == 0; });
    }
    void stop()
    {
        { std::lock_guard<std::mutex> l(m); quit = true; }
        cvGo.notify_all();
        for (std::thread& t : th) if (t.joinable()) t.join();
        th.clear();
    }
};
struct idkpt_ctx;
struct GroupBandUser { idkpt_ctx* c; int d; };

struct idkpt_ctx {
    std::vector<dev_ctx*> dev;
    std::string lastError;
    idkpt_error_fn errFn = nullptr; void* errUser = nullptr;
    PeerPolicy peer;                                        // how the members copy to each other (xGMI peer copies, or staged through the host)
    RcclTransport rccl; int transportOpt = 0;               // option "transport": 0 = RCCL where it can be formed, else peer copies; 1 = peer copies; 2 = RCCL or fail (transport_rccl.hpp)
    hipEvent_t evGatherStart = nullptr;                     // device 0: what was queued on its stream before a gather (the consumer of the previous frame)
    bool frameOk = true;                                    // false after a failed re-layout: idkptRender refuses until the next successful idkptSetSize
    int W = 0, H = 0;
    int shardMode = IDKPT_SHARD_AUTO; bool strips = false;
    int bandLog2 = 0;                                       // interleaved layouts: rows are dealt in bands of 2^bandLog2 rows (IDKPT_SHARD_BANDS: 8; IDKPT_SHARD_ROWS: 1)
    idkpt_settings st;
    int maxBatch = 1, pending = 0;
    std::vector<int> firstRow, rowCount;                    // strips: member d renders rows [firstRow[d], firstRow[d] + rowCount[d])
    std::vector<std::array<hipEvent_t, MAX_DEPTH_SLOTS>> evBounce;
    std::vector<hipEvent_t> evFlushDone, evGather; std::vector<char> flushDoneValid;
    std::vector<DevBuf> peerStage, gbase;                   // on member d: the lower members' per-sample bases of the current bounce; the summed slot bases
    GroupWorkers workers;                                   // one enqueuing thread per member (threaded flushes)
    GroupBarrier bar; std::vector<GroupBandUser> bandUser; std::vector<std::vector<uint32_t>> bandCounts; std::vector<int> bandLB;   // interleaved layouts beyond RayDepth 2 (group_band_exchange)
    DevBuf full[3], gatherStage, rowOffDev;                 // on device 0: gathered full-frame images; landing zone of the interleaved rows; first landing row of every member
    size_t n() const { return dev.size(); }
};

// gbase[k] = sum over the members that own earlier rows of their alive count of sample k (count = bases[k+1] - bases[k])
__global__ void k_group_bases(const uint32_t* stage, int lower, int stride, int samples, uint32_t* gbase)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= samples) return;
    uint32_t sum = 0;
    for (int d = 0; d < lower; d++) sum += stage[(size_t)d * stride + k + 1] - stage[(size_t)d * stride + k];
    gbase[k] = sum;
}
// full[y][x] = rows of member (y >> bandLog2) % n, landed contiguously per member in `stage` (member d at rowOffset[d] rows; its local row of image row y:
// band (y >> bandLog2) / n of the member, row y & (band - 1) inside it — every band but the image's last is complete)
__global__ void k_interleave_rows(const float4* stage, float4* full, int W, int H, int n, const int* rowOffset, int bandLog2)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)W * H) return;
    const int y = (int)(i / W), x = (int)(i % W), band = y >> bandLog2, d = band % n, ly = ((band / n) << bandLog2) | (y & ((1 << bandLog2) - 1));
    full[i] = stage[((size_t)rowOffset[d] + ly) * W + x];
}

// one place for the last error: the context's string; a one-device context mirrors it into its member (idkptGetLastError reads the member there)
// (group-level errors reach the host's error callback here; a member's own error already did in fail(), so mfail only copies the message)
static int gfail(idkpt_ctx* c, int code, const std::string& msg) { c->lastError = msg; if (c->n() == 1) c->dev[0]->lastError = msg; if (c->errFn) c->errFn(c->errUser, (int32_t)code, c->lastError.c_str()); return code; }
static int mfail(idkpt_ctx* c, dev_ctx* m, int rc) { c->lastError = m->lastError; return rc; }
// RCCL for the bulk device-to-device traffic of this context (scene replication, frame gather), or not: decided at the first use, reported by idkptGetTransportInfo
static bool group_rccl(idkpt_ctx* c)
{
    if (c->n() < 2 || c->transportOpt == 1) return false;
    if (!c->rccl.tried) { std::vector<int> devs; for (dev_ctx* m : c->dev) devs.push_back(m->device); (void)c->rccl.init(devs); }
    return c->rccl.ready;
}
// a failing RCCL call: say so once, stop using RCCL on this context (the caller repeats the operation with peer copies)
static void group_rccl_failed(idkpt_ctx* c) { fprintf(stderr, "[idkpt] warning: %s; this context uses peer copies from now on\n", c->rccl.lastError.c_str()); c->rccl.why = c->rccl.lastError; c->rccl.shutdown(); }
#define GREQ(cond, msg) do { if (!(cond)) return gfail(c, IDKPT_ERR_INVALID_ARGUMENT, msg); } while (0)
#define GHIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { (void)hipGetLastError(); return gfail(c, IDKPT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); } } while (0)
#define ONE(call) do { if (c->n() == 1) { dev_ctx* m = c->dev[0]; return call; } } while (0)
#define ALL(call) do { for (dev_ctx* m : c->dev) { int _rc = call; if (_rc) return mfail(c, m, _rc); } } while (0)

static void strip_of(int H, int n, int d, int* first, int* count) { const int base = H / n, extra = H % n; *first = d * base + std::min(d, extra); *count = base + (d < extra ? 1 : 0); }

// the hook every member calls at the start of a bounce (flush_batch): slot bases of its samples = alive rays of the members above it
static int group_exchange(void* user, dev_ctx* m, int bounce, int samples, const uint32_t** outBases)
{
    idkpt_ctx* c = (idkpt_ctx*)user;
    *outBases = nullptr;
    if (!c->strips || m->groupIndex == 0) return IDKPT_OK;               // interleaved rows: no exchange (exact at RayDepth <= 2); member 0 starts at slot 0
    const int d = m->groupIndex, BS = MAX_BATCH + 1;
    dev_ctx* ctx = m;                                                      // (HIPC reports into the member)
    HIPC(c->peerStage[d].ensure((size_t)d * BS * 4)); HIPC(c->gbase[d].ensure((size_t)MAX_BATCH * 4));
    for (int d2 = 0; d2 < d; d2++) {
        dev_ctx* lo = c->dev[d2];
        HIPC(hipStreamWaitEvent(m->stream, c->evBounce[d2][bounce], 0));  // the lower member's counts of this bounce are final (it was enqueued first)
        HIPC(member_copy(&c->peer, c->peerStage[d].as<uint32_t>() + (size_t)d2 * BS, m->device, lo->bases.as<uint32_t>() + (size_t)bounce * BS, lo->device, (size_t)(samples + 1) * 4, m->stream));
    }
    hipLaunchKernelGGL(k_group_bases, dim3((samples + 63) / 64), dim3(64), 0, m->stream, (const uint32_t*)c->peerStage[d].as<uint32_t>(), d, BS, samples, c->gbase[d].as<uint32_t>());
    HIPC(hipGetLastError());
    *outBases = c->gbase[d].as<uint32_t>();
    return IDKPT_OK;
}

// the host side of idkptSetBandExchange inside one multi-device context: every member posts its per-(sample, band) alive counts, all meet, every member sums the
// counts of the image bands before each of its own bands (image band g belongs to member g % n as its (g / n)-th band)
static void group_band_exchange(void* user, int32_t bounce, int32_t samples, int32_t bands, const uint32_t* localCounts, uint32_t* outBases)
{
    (void)bounce;
    GroupBandUser* u = (GroupBandUser*)user; idkpt_ctx* c = u->c; const int d = u->d, n = (int)c->n();
    c->bandCounts[d].assign(localCounts, localCounts + (size_t)samples * bands); c->bandLB[d] = bands;
    // (a member failed and aborted the barrier: this batch is reported as failed by group_flush; the bases are zeroed so that nothing reads what the buffer held)
    if (!c->bar.wait()) { memset(outBases, 0, (size_t)samples * bands * sizeof(uint32_t)); return; }   // everybody has posted
    int total = 0; for (int r = 0; r < n; r++) total += c->bandLB[r];
    for (int k = 0; k < samples; k++) {
        uint32_t cum = 0;
        for (int g = 0; g < total; g++) {
            const int r = g % n, b = g / n;
            if (b >= c->bandLB[r]) continue;                                   // (cannot happen with a round-robin deal: kept for safety)
            if (r == d) outBases[(size_t)k * bands + b] = cum;
            cum += c->bandCounts[r][(size_t)k * c->bandLB[r] + b];
        }
    }
    (void)c->bar.wait();                                                    // nobody posts the next bounce before everybody has read this one
}

static int group_flush(idkpt_ctx* c);
// what a member calls when a scene update finds no free scene-version slot (ver_writable, idkpt.hip): the whole group launches what it has queued
static int group_flush_all(void* user) { return group_flush((idkpt_ctx*)user); }

// launches what the members have queued.  Members are enqueued in row order, so a member only ever waits for members enqueued before it.
static int group_flush(idkpt_ctx* c)
{
    if (c->pending == 0) return IDKPT_OK;
    // interleaved rows / bands beyond RayDepth 2 with sorting off: exact slot numbers need every member's band counts at every bounce -> one host thread per member,
    // meeting in group_band_exchange (strips keep the pipelined, device-side exchange below; RayDepth <= 2 needs none)
    // (option group_threads: the interleaved layouts at RayDepth <= 2 need no exchange and can be enqueued by one thread or by one per member — N x ~15 launches of host
    // time in a row or side by side; -1 = one per member from 4 members on)
    const bool exchange = !c->strips && c->n() > 1 && c->st.RayDepth > 2;
    const int threadsOpt = c->dev[0]->opt.groupThreads;
    const bool threaded = exchange || (!c->strips && c->n() > 1 && (threadsOpt < 0 ? c->n() >= 4 : threadsOpt != 0));
    for (size_t d = 0; d < c->n(); d++) { dev_ctx* m = c->dev[d]; m->bandExchangeFn = exchange ? group_band_exchange : nullptr; m->bandExchangeUser = exchange ? (void*)&c->bandUser[d] : nullptr; }
    if (threaded) {
        c->bar.reset((int)c->n());
        std::vector<int> rcs(c->n(), IDKPT_OK);
        c->workers.start(c->n());
        c->workers.run([c, &rcs](size_t d) {
            dev_ctx* m = c->dev[d];
            int rc = hipSetDevice(m->device) == hipSuccess ? IDKPT_OK : IDKPT_ERR_HIP;      // (the current device is per host thread: set on every job, the member list may be any ids)
            if (rc == IDKPT_OK) { m->inGroupFlush = true; rc = flush_batch(m); m->inGroupFlush = false; }
            rcs[d] = rc;
            if (rc) c->bar.abort();                                          // (the others leave their exchanges and finish; the group reports this member's error)
        });
        (void)hipSetDevice(c->dev[0]->device);
        for (size_t d = 0; d < c->n(); d++) if (rcs[d]) { for (dev_ctx* o : c->dev) o->pending.clear(); c->pending = 0; return mfail(c, c->dev[d], rcs[d]); }
        c->pending = 0;
        return IDKPT_OK;
    }
    for (size_t d = 0; d < c->n(); d++) {
        dev_ctx* m = c->dev[d];
        GHIP(hipSetDevice(m->device));
        // the next batch of this member must not overwrite the per-bounce tables that members below it may still be copying from the last one
        if (c->strips) for (size_t d2 = d + 1; d2 < c->n(); d2++) if (c->flushDoneValid[d2]) GHIP(hipStreamWaitEvent(m->stream, c->evFlushDone[d2], 0));
        m->inGroupFlush = true;
        const int rc = flush_batch(m);
        m->inGroupFlush = false;
        if (rc) { for (dev_ctx* o : c->dev) o->pending.clear(); c->pending = 0; return mfail(c, m, rc); }
        if (c->strips) { GHIP(hipEventRecord(c->evFlushDone[d], m->stream)); c->flushDoneValid[d] = 1; }
    }
    c->pending = 0;
    return IDKPT_OK;
}
#define GFLUSH() do { int _rc = group_flush(c); if (_rc) return _rc; } while (0)

static int group_sync(idkpt_ctx* c)
{
    for (dev_ctx* m : c->dev) {
        GHIP(hipSetDevice(m->device)); GHIP(hipStreamSynchronize(m->stream));
        int rc = check_overflow(m); if (rc) return mfail(c, m, rc);
    }
    return IDKPT_OK;
}

// (round 4: AUTO deals bands at every RayDepth — beyond 2 with the per-band count exchange — because strips balance badly on views with empty rows: BASELINE configs[3]
// on 8 GPUs projects 6.0x with bands against 3.8x with strips, profiles/r04_raw/shard_config4_batch32.txt; strips only when asked for)
static bool mode_wants_strips(int mode, int rayDepth) { (void)rayDepth; return mode == IDKPT_SHARD_STRIPS; }
// interleaved layouts: AUTO and BANDS deal bands of 8 rows (a wave's 8x8 pixel tile stays one 8x8 block of the image on every device: single rows cost
// 1.5-4.5 % of traversal coherence at N = 2..8, profiles/r02_shard_coherence.txt), ROWS deals single rows.  Images too small for one band per device fall back to rows.
static int mode_band_log2(int mode, int H, int n) { return (mode == IDKPT_SHARD_ROWS || (H + 7) / 8 < n) ? 0 : 3; }

// (re)applies size + row layout to every member.  Strips for deep paths (exact slot numbering), interleaved bands / rows otherwise.
static int group_layout(idkpt_ctx* c, int W, int H)
{
    const int n = (int)c->n();
    const bool strips = mode_wants_strips(c->shardMode, c->st.RayDepth);
    c->strips = strips;
    if (W <= 0) return IDKPT_OK;
    GREQ(H >= n, "idkptSetSize: a multi-device context needs at least one image row per device");     // (checked before any member is touched)
    const int bandLog2 = strips ? 0 : mode_band_log2(c->shardMode, H, n);
    // A member's allocation can fail half way through the loop; the group then has no usable frame (idkptRender refuses) until a later
    // idkptSetSize / idkptSetSettings / idkptSetGroupSharding re-layout succeeds — never a mix of members on the old and the new size.
    c->frameOk = false;
    std::vector<int> first(n), count(n);
    for (int d = 0; d < n; d++) {
        int rc;
        if (strips) { strip_of(H, n, d, &first[d], &count[d]); rc = dev_SetLayout(c->dev[d], W, H, 1, first[d], count[d]); }
        else { first[d] = d << bandLog2; count[d] = local_rows(H, n, d, bandLog2); rc = dev_SetLayout(c->dev[d], W, H, n, d, 0x7fffffff, bandLog2); }
        if (rc) { for (dev_ctx* o : c->dev) o->frameOk = false; return mfail(c, c->dev[d], rc); }
        c->flushDoneValid[d] = 0;
    }
    // where member d's rows land in the gather staging area (interleaved layout), kept on device 0 for k_interleave_rows
    std::vector<int> rowOff(n, 0);
    for (int d = 1; d < n; d++) rowOff[d] = rowOff[d - 1] + c->dev[d - 1]->rows;
    dev_ctx* m0 = c->dev[0];
    GHIP(hipSetDevice(m0->device));
    GHIP(c->rowOffDev.ensure((size_t)n * 4));
    GHIP(hipMemcpyAsync(c->rowOffDev.p, rowOff.data(), (size_t)n * 4, hipMemcpyHostToDevice, m0->stream));
    GHIP(hipStreamSynchronize(m0->stream));
    c->W = W; c->H = H; c->firstRow = first; c->rowCount = count; c->bandLog2 = bandLog2; c->frameOk = true;      // commit
    return IDKPT_OK;
}

// member rows -> host image (every member copies its rows itself: one PCIe link per GPU)
static int group_download_image(idkpt_ctx* c, int image, int slot, float* rgba, size_t bytes)
{
    GREQ(image >= 0 && image < 3, "idkptDownload: bad image id");
    const size_t rowBytes = (size_t)c->W * 16;
    GREQ(c->W > 0 && bytes == rowBytes * c->H, "idkptDownload: bytes must equal height*width*16 (the whole frame of a multi-device context)");
    GREQ(slot >= 0 && slot < c->dev[0]->ringSize, "idkptDownloadFrame: slot outside the frame ring");
    GFLUSH();
    const int n = (int)c->n();
    for (int d = 0; d < n; d++) {
        dev_ctx* m = c->dev[d];
        GHIP(hipSetDevice(m->device));
        const float4* src = image_ptr(m, image, slot);
        if (c->strips) GHIP(hipMemcpyAsync((char*)rgba + (size_t)c->firstRow[d] * rowBytes, src, (size_t)m->rows * rowBytes, hipMemcpyDeviceToHost, m->stream));
        else {
            // band k of this member = band k * n + d of the image: one 2D copy whose "rows" are whole bands, plus the member's last band when the image cuts it short
            const size_t band = (size_t)1 << c->bandLog2, bandBytes = band * rowBytes, fullBands = (size_t)m->rows >> c->bandLog2, tail = (size_t)m->rows - (fullBands << c->bandLog2);
            if (fullBands) GHIP(hipMemcpy2DAsync((char*)rgba + (size_t)d * bandBytes, (size_t)n * bandBytes, src, bandBytes, bandBytes, fullBands, hipMemcpyDeviceToHost, m->stream));
            if (tail) GHIP(hipMemcpyAsync((char*)rgba + ((size_t)d + fullBands * n) * bandBytes, (const char*)src + fullBands * bandBytes, tail * rowBytes, hipMemcpyDeviceToHost, m->stream));
        }
    }
    return group_sync(c);
}

// member rows -> full frame on device 0 (RCCL send / recv, or peer copies; interleaved rows land in a staging area and are woven together by one kernel)
static int group_gather_device(idkpt_ctx* c, int image, int slot, void** outPtr, size_t* outBytes)
{
    GREQ(image >= 0 && image < 3 && c->W > 0, "idkptGetImageDevicePtr: bad image / no size");
    GREQ(slot >= 0 && slot < c->dev[0]->ringSize, "idkptGetFrameDevicePtr: slot outside the frame ring");
    GFLUSH();
    const int n = (int)c->n();
    const size_t rowBytes = (size_t)c->W * 16, frameBytes = rowBytes * c->H;
    dev_ctx* m0 = c->dev[0];
    GHIP(hipSetDevice(m0->device));
    GHIP(c->full[image].ensure(frameBytes));
    std::vector<int> rowOff(n + 1, 0);
    for (int d = 0; d < n; d++) rowOff[d + 1] = rowOff[d] + c->dev[d]->rows;
    if (!c->strips) GHIP(c->gatherStage.ensure(frameBytes));
    // The landing buffers on device 0 may still be read by what is queued on device 0's stream (the previous gather's k_interleave_rows, a consumer
    // the host ordered behind idkptGetStream): no member may write them before that work is done.
    GHIP(hipEventRecord(c->evGatherStart, m0->stream));
    if (c->peer.forceStaged) GHIP(hipEventSynchronize(c->evGatherStart));          // (staged copies land from the host, outside any stream order)
    bool viaRccl = group_rccl(c) && !c->peer.forceStaged;
    if (viaRccl) {
        // grouped ncclSend (member d, behind its FinalDraw) / ncclRecv (device 0, behind evGatherStart by stream order); member 0's own rows are a local copy
        std::vector<const void*> src(n); std::vector<void*> dst(n); std::vector<size_t> bytes(n); std::vector<int> devs(n); std::vector<hipStream_t> streams(n);
        for (int d = 0; d < n; d++) {
            dev_ctx* m = c->dev[d];
            src[d] = image_ptr(m, image, slot); bytes[d] = (size_t)m->rows * rowBytes; devs[d] = m->device; streams[d] = m->stream;
            dst[d] = c->strips ? (char*)c->full[image].p + (size_t)c->firstRow[d] * rowBytes : (char*)c->gatherStage.p + (size_t)rowOff[d] * rowBytes;
            if (d > 0) { GHIP(hipSetDevice(m->device)); GHIP(hipStreamWaitEvent(m->stream, c->evGatherStart, 0)); }
        }
        GHIP(hipSetDevice(m0->device));
        if (bytes[0]) GHIP(hipMemcpyAsync(dst[0], src[0], bytes[0], hipMemcpyDeviceToDevice, m0->stream));
        if (!c->rccl.gather(src, dst, bytes, devs, streams)) { group_rccl_failed(c); viaRccl = false; }
        else for (int d = 0; d < n; d++) { GHIP(hipSetDevice(c->dev[d]->device)); GHIP(hipEventRecord(c->evGather[d], c->dev[d]->stream)); }
    }
    if (!viaRccl)
    for (int d = 0; d < n; d++) {
        dev_ctx* m = c->dev[d];
        GHIP(hipSetDevice(m->device));
        GHIP(hipStreamWaitEvent(m->stream, c->evGatherStart, 0));
        char* dst = c->strips ? (char*)c->full[image].p + (size_t)c->firstRow[d] * rowBytes : (char*)c->gatherStage.p + (size_t)rowOff[d] * rowBytes;
        GHIP(member_copy(&c->peer, dst, m0->device, image_ptr(m, image, slot), m->device, (size_t)m->rows * rowBytes, m->stream));   // ordered behind the member's FinalDraw
        GHIP(hipEventRecord(c->evGather[d], m->stream));
    }
    GHIP(hipSetDevice(m0->device));
    for (int d = 1; d < n; d++) GHIP(hipStreamWaitEvent(m0->stream, c->evGather[d], 0));
    if (!c->strips) {
        const size_t px = (size_t)c->W * c->H;
        hipLaunchKernelGGL(k_interleave_rows, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, m0->stream, (const float4*)c->gatherStage.p, (float4*)c->full[image].p, c->W, c->H, n, (const int*)c->rowOffDev.p, c->bandLog2);
        GHIP(hipGetLastError());
    }
    *outPtr = c->full[image].p;
    if (outBytes) *outBytes = frameBytes;
    return IDKPT_OK;
}

// local pixel index of member d -> pixel index of the whole frame
static inline uint32_t global_pixel(const idkpt_ctx* c, int d, uint32_t local)
{
    const uint32_t W = (uint32_t)c->W, ly = local / W, x = local % W;
    const uint32_t b = (uint32_t)c->bandLog2;
    const uint32_t y = c->strips ? (uint32_t)c->firstRow[d] + ly : ((((ly >> b) * (uint32_t)c->n() + (uint32_t)d) << b) | (ly & ((1u << b) - 1u)));
    return y * W + x;
}

extern "C" {

const char* idkptGetVersionString(void) { return "idkpt 0.3 (gfx950)"; }
int32_t idkptGetAbiVersion(void) { return IDKPT_ABI_VERSION; }
int32_t idkptGetDeviceCount(int32_t* outCount) { return dev_GetDeviceCount(outCount); }

int32_t idkptCreate(int32_t deviceCount, const int32_t* deviceIds, idkpt_ctx** outCtx)
{
    if (!outCtx) return IDKPT_ERR_INVALID_ARGUMENT;
    *outCtx = nullptr;
    if (deviceCount < 1 || deviceCount > 64) return IDKPT_ERR_INVALID_ARGUMENT;
    idkpt_ctx* c = new idkpt_ctx();
    memset(&c->st, 0, sizeof(c->st));
    for (int d = 0; d < deviceCount; d++) {
        int32_t id = deviceIds ? deviceIds[d] : d;
        dev_ctx* m = nullptr;
        int rc = dev_Create(1, &id, &m);
        if (rc) { for (dev_ctx* o : c->dev) dev_Destroy(o); delete c; return rc; }
        c->dev.push_back(m);
    }
    c->st = c->dev[0]->st;
    const int n = deviceCount;
    if (n > 1) {
        c->firstRow.assign(n, 0); c->rowCount.assign(n, 0); c->evBounce.resize(n); c->evFlushDone.assign(n, nullptr); c->evGather.assign(n, nullptr);
        c->flushDoneValid.assign(n, 0); c->peerStage.resize(n); c->gbase.resize(n);
        c->bandUser.resize(n); c->bandCounts.resize(n); c->bandLB.assign(n, 0); for (int d = 0; d < n; d++) c->bandUser[d] = GroupBandUser{c, d};
        bool ok = true;
        for (int d = 0; d < n && ok; d++) {
            dev_ctx* m = c->dev[d];
            ok = hipSetDevice(m->device) == hipSuccess;
            for (int j = 0; j < MAX_DEPTH_SLOTS && ok; j++) ok = hipEventCreateWithFlags(&c->evBounce[d][j], hipEventDisableTiming) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&c->evFlushDone[d], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c->evGather[d], hipEventDisableTiming) == hipSuccess;
            m->grouped = true; m->groupIndex = d; m->evBounce = c->evBounce[d].data(); m->groupExchange = group_exchange; m->groupUser = c; m->peer = &c->peer; m->groupFlushAll = group_flush_all;
            // direct xGMI access between the members' devices (copies work without it through staging; a refusal is not an error)
            for (int d2 = 0; d2 < n; d2++) if (c->dev[d2]->device != m->device) { int can = 0; if (hipDeviceCanAccessPeer(&can, m->device, c->dev[d2]->device) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(c->dev[d2]->device, 0); (void)hipGetLastError(); }
        }
        ok = ok && hipSetDevice(c->dev[0]->device) == hipSuccess && hipEventCreateWithFlags(&c->evGatherStart, hipEventDisableTiming) == hipSuccess;
        if (!ok) { (void)hipGetLastError(); for (dev_ctx* o : c->dev) dev_Destroy(o); delete c; return IDKPT_ERR_HIP; }
    }
    *outCtx = c;
    return IDKPT_OK;
}

int32_t idkptDestroy(idkpt_ctx* c)
{
    if (!c) return IDKPT_ERR_INVALID_ARGUMENT;
    for (dev_ctx* m : c->dev) { (void)hipSetDevice(m->device); (void)hipStreamSynchronize(m->stream); }
    c->workers.stop();
    c->rccl.shutdown();
    if (c->n() > 1) {
        for (size_t d = 0; d < c->n(); d++) {
            (void)hipSetDevice(c->dev[d]->device);
            for (hipEvent_t e : c->evBounce[d]) if (e) (void)hipEventDestroy(e);
            if (c->evFlushDone[d]) (void)hipEventDestroy(c->evFlushDone[d]);
            if (c->evGather[d]) (void)hipEventDestroy(c->evGather[d]);
            c->peerStage[d].release(); c->gbase[d].release();
        }
        (void)hipSetDevice(c->dev[0]->device);
        for (int i = 0; i < 3; i++) c->full[i].release();
        c->gatherStage.release(); c->rowOffDev.release();
        if (c->evGatherStart) (void)hipEventDestroy(c->evGatherStart);
        if (c->peer.stage) (void)hipHostFree(c->peer.stage);
    }
    for (dev_ctx* m : c->dev) dev_Destroy(m);
    delete c;
    return IDKPT_OK;
}

int32_t idkptGetLastError(idkpt_ctx* c, const char** outMessage)
{
    if (!c || !outMessage) return IDKPT_ERR_INVALID_ARGUMENT;
    if (c->n() == 1) return dev_GetLastError(c->dev[0], outMessage);      // (gfail mirrors group-level messages into the member)
    *outMessage = c->lastError.c_str();
    return IDKPT_OK;
}

int32_t idkptSetErrorCallback(idkpt_ctx* c, idkpt_error_fn fn, void* user)
{
    if (!c) return IDKPT_ERR_INVALID_ARGUMENT;
    c->errFn = fn; c->errUser = user;
    for (dev_ctx* m : c->dev) { m->errFn = fn; m->errUser = user; }
    return IDKPT_OK;
}

int32_t idkptGetContextDeviceCount(idkpt_ctx* c, int32_t* outCount) { if (!c || !outCount) return IDKPT_ERR_INVALID_ARGUMENT; *outCount = (int32_t)c->n(); return IDKPT_OK; }

int32_t idkptGetTransportInfo(idkpt_ctx* c, int32_t* outKind, int32_t* outRanks, int32_t* outRcclVersion, const char** outDetail)
{
    if (!c) return IDKPT_ERR_INVALID_ARGUMENT;
    const bool rccl = group_rccl(c);               // (forms the communicators if that has not been tried yet: the answer is what the next upload / gather will use)
    if (outKind) *outKind = c->n() < 2 ? IDKPT_TRANSPORT_NONE : (rccl ? IDKPT_TRANSPORT_RCCL : IDKPT_TRANSPORT_PEER_COPY);
    if (outRanks) *outRanks = rccl ? (int32_t)c->rccl.comms.size() : 0;
    if (outRcclVersion) *outRcclVersion = c->rccl.version;
    if (outDetail) *outDetail = c->n() < 2 ? "one device: nothing travels" : (rccl ? c->rccl.api->path.c_str() : (c->transportOpt == 1 ? "peer copies by option" : c->rccl.why.c_str()));
    return IDKPT_OK;
}
int32_t idkptTransportSelfTest(int32_t device, int32_t* outRcclVersion, char* outDetail, size_t detailBytes)
{
    std::string why;
    const int32_t rc = rccl_self_test((int)device, outRcclVersion, &why);
    if (outDetail && detailBytes) { snprintf(outDetail, detailBytes, "%s", rc == IDKPT_OK ? "ok" : why.c_str()); }
    return rc;
}

int32_t idkptSetGroupSharding(idkpt_ctx* c, int32_t mode)
{
    if (!c) return IDKPT_ERR_INVALID_ARGUMENT;
    GREQ(mode == IDKPT_SHARD_AUTO || mode == IDKPT_SHARD_ROWS || mode == IDKPT_SHARD_STRIPS || mode == IDKPT_SHARD_BANDS, "idkptSetGroupSharding: unknown mode");
    if (c->n() == 1) { c->shardMode = mode; return IDKPT_OK; }
    GFLUSH();
    const bool was = c->strips; const int wasBand = c->bandLog2;
    c->shardMode = mode;
    const bool now = mode_wants_strips(mode, c->st.RayDepth);
    const int nowBand = now ? 0 : mode_band_log2(mode, c->H, (int)c->n());
    return (now != was || (c->W > 0 && nowBand != wasBand)) ? group_layout(c, c->W, c->H) : IDKPT_OK;
}

int32_t idkptSetSize(idkpt_ctx* c, int32_t width, int32_t height)
{
    if (!c) return IDKPT_ERR_INVALID_ARGUMENT;
    ONE(dev_SetSize(m, width, height));
    GREQ(width > 0 && height > 0 && width <= 4096 && height <= 65536, "idkptSetSize: bad size (FirstHit seeds pack x into 12 bits: width <= 4096)");
    GFLUSH();
    return group_layout(c, width, height);
}
int32_t idkptSetRowSharding(idkpt_ctx* c, int32_t rowModulo, int32_t rowRemainder)
{
    if (!c) return IDKPT_ERR_INVALID_ARGUMENT;
    ONE(dev_SetRowSharding(m, rowModulo, rowRemainder));
    return gfail(c, IDKPT_ERR_INVALID_OPERATION, "idkptSetRowSharding: a multi-device context deals its rows itself (idkptSetGroupSharding)");
}
int32_t idkptSetRowBands(idkpt_ctx* c, int32_t bandRows, int32_t rowModulo, int32_t rowRemainder)
{
    if (!c) return IDKPT_ERR_INVALID_ARGUMENT;
    ONE(dev_SetRowBands(m, bandRows, rowModulo, rowRemainder));
    return gfail(c, IDKPT_ERR_INVALID_OPERATION, "idkptSetRowBands: a multi-device context deals its rows itself (idkptSetGroupSharding)");
}
int32_t idkptSetRowRange(idkpt_ctx* c, int32_t firstRow, int32_t rowCount)
{
    if (!c) return IDKPT_ERR_INVALID_ARGUMENT;
    ONE(dev_SetRowRange(m, firstRow, rowCount));
    return gfail(c, IDKPT_ERR_INVALID_OPERATION, "idkptSetRowRange: a multi-device context deals its rows itself (idkptSetGroupSharding)");
}
int32_t idkptSetBandExchangeDevice(idkpt_ctx* c, idkpt_band_exchange_device_fn fn, void* user)
{
    if (!c) return IDKPT_ERR_INVALID_ARGUMENT;
    ONE(dev_SetBandExchangeDevice(m, fn, user));
    return gfail(c, IDKPT_ERR_INVALID_OPERATION, "idkptSetBandExchangeDevice: a multi-device context deals and numbers its rows itself");
}
int32_t idkptSetBandExchange(idkpt_ctx* c, idkpt_band_exchange_fn fn, void* user)
{
    if (!c) return IDKPT_ERR_INVALID_ARGUMENT;
    ONE(dev_SetBandExchange(m, fn, user));
    return gfail(c, IDKPT_ERR_INVALID_OPERATION, "idkptSetBandExchange: a multi-device context deals and numbers its rows itself");
}
int32_t idkptSetBounceExchange(idkpt_ctx* c, idkpt_bounce_exchange_fn fn, void* user)
{
    if (!c) return IDKPT_ERR_INVALID_ARGUMENT;
    ONE(dev_SetBounceExchange(m, fn, user));
    return gfail(c, IDKPT_ERR_INVALID_OPERATION, "idkptSetBounceExchange: a multi-device context exchanges its alive counts on the devices");
}

int32_t idkptSetSettings(idkpt_ctx* c, const idkpt_settings* s)
{
    if (!c || !s) return IDKPT_ERR_INVALID_ARGUMENT;
    ONE(dev_SetSettings(m, s));
    if (memcmp(&c->st, s, sizeof(*s)) == 0) return IDKPT_OK;
    GFLUSH();
    const bool wantStrips = mode_wants_strips(c->shardMode, s->RayDepth);
    ALL(dev_SetSettings(m, s));                       // (validates; nothing is changed when it fails on the first member)
    c->st = *s;
    if (wantStrips != c->strips) { int rc = group_layout(c, c->W, c->H); if (rc) return rc; }   // the RayDepth setter resets the accumulation anyway (PathTracer.cs:16-25)
    return IDKPT_OK;
}
int32_t idkptGetSettings(idkpt_ctx* c, idkpt_settings* out) { if (!c || !out) return IDKPT_ERR_INVALID_ARGUMENT; return dev_GetSettings(c->dev[0], out); }

int32_t idkptSetPerFrame(idkpt_ctx* c, const float invProjection[16], const float invView[16], const float viewPos[3])
{
    if (!c || !invProjection || !invView || !viewPos) return IDKPT_ERR_INVALID_ARGUMENT;
    ONE(dev_SetPerFrame(m, invProjection, invView, viewPos));
    dev_ctx* m0 = c->dev[0];
    if (m0->ringSize == 1 && (memcmp(m0->invProj, invProjection, 64) || memcmp(m0->invView, invView, 64) || memcmp(m0->viewPos, viewPos, 12))) GFLUSH();   // one camera per batch
    ALL(dev_SetPerFrame(m, invProjection, invView, viewPos));
    return IDKPT_OK;
}
int32_t idkptSetPerFrameData(idkpt_ctx* c, const GpuPerFrameData* p) { if (!c || !p) return IDKPT_ERR_INVALID_ARGUMENT; return idkptSetPerFrame(c, p->InvProjection, p->InvView, p->ViewPos); }

int32_t idkptUploadScene(idkpt_ctx* c, const idkpt_scene_desc* scene)
{
    if (!c || !scene) return IDKPT_ERR_INVALID_ARGUMENT;
    ONE(dev_UploadScene(m, scene));
    GFLUSH();
    { int rc = dev_UploadScene(c->dev[0], scene); if (rc) return mfail(c, c->dev[0], rc); }          // host -> device 0 (validated there)
    // device 0 -> every other device: one ncclBroadcast per buffer over all members (RCCL), or member-by-member peer copies
    if (c->transportOpt == 2 && !group_rccl(c)) return gfail(c, IDKPT_ERR_INVALID_OPERATION, "idkptUploadScene: option transport = 2 asks for RCCL, which is not usable here: " + c->rccl.why);
    if (group_rccl(c)) {
        const size_t n = c->n();
        std::vector<std::vector<CloneItem>> items(n);
        for (size_t d = 1; d < n; d++) { int rc = clone_prepare(c->dev[d], c->dev[0], items[d]); if (rc) return mfail(c, c->dev[d], rc); }
        std::vector<int> devs; std::vector<hipStream_t> streams;
        for (dev_ctx* m : c->dev) { devs.push_back(m->device); streams.push_back(m->stream); }
        bool good = true;
        for (size_t i = 0; i < items[1].size() && good; i++) {
            std::vector<void*> dst(n); dst[0] = (void*)items[1][i].src;
            for (size_t d = 1; d < n; d++) dst[d] = items[d][i].dst;
            good = c->rccl.broadcast(dst, items[1][i].bytes, devs, streams);
        }
        if (!good) {   // (nothing of the scene is in use yet: the same buffers are filled again by peer copies)
            group_rccl_failed(c);
            for (size_t d = 1; d < n; d++) { dev_ctx* ctx = c->dev[d]; for (const CloneItem& it : items[d]) { if (member_copy(ctx->peer, it.dst, ctx->device, it.src, c->dev[0]->device, it.bytes, ctx->stream) != hipSuccess) return gfail(c, IDKPT_ERR_HIP, "idkptUploadScene: device-to-device copy failed"); } }
        }
        for (size_t d = 1; d < n; d++) { int rc = clone_finish(c->dev[d], c->dev[0]); if (rc) return mfail(c, c->dev[d], rc); }
        (void)hipSetDevice(c->dev[0]->device);
        return IDKPT_OK;
    }
    for (size_t d = 1; d < c->n(); d++) { int rc = dev_CloneSceneFrom(c->dev[d], c->dev[0]); if (rc) return mfail(c, c->dev[d], rc); }   // device 0 -> device d
    return IDKPT_OK;
}
#define REPLICATE(name, ...) do { if (!c) return IDKPT_ERR_INVALID_ARGUMENT; ONE(dev_##name(m, ##__VA_ARGS__)); GFLUSH(); ALL(dev_##name(m, ##__VA_ARGS__)); return IDKPT_OK; } while (0)
// Scene updates that the members apply through their scene-version slots (idkptSetSceneVersions): the group does NOT launch what is queued first — every member
// finds or makes room itself (ver_writable) and, when it cannot, launches the whole group through group_flush_all.  All members hold the same queue and the same
// version state (every call is replicated), so they take the same decisions.
#define REPLICATE_VERSIONED(name, ...) do { if (!c) return IDKPT_ERR_INVALID_ARGUMENT; ONE(dev_##name(m, ##__VA_ARGS__)); ALL(dev_##name(m, ##__VA_ARGS__)); return IDKPT_OK; } while (0)
int32_t idkptUpdateBuffer(idkpt_ctx* c, int32_t which, size_t offsetBytes, size_t bytes, const void* data) { REPLICATE_VERSIONED(UpdateBuffer, which, offsetBytes, bytes, data); }
int32_t idkptSetSceneVersions(idkpt_ctx* c, int32_t versions) { REPLICATE(SetSceneVersions, versions); }
int32_t idkptSetLightCount(idkpt_ctx* c, int32_t count) { REPLICATE(SetLightCount, count); }
int32_t idkptUpdateTexture(idkpt_ctx* c, int32_t index, const idkpt_texture* texture) { REPLICATE(UpdateTexture, index, texture); }
int32_t idkptSetDeveloperOption(idkpt_ctx* c, const char* name, int32_t value)
{
    if (!c || !name) return IDKPT_ERR_INVALID_ARGUMENT;
    if (c->n() > 1 && std::string(name) == "force_no_peer") { GFLUSH(); c->peer.forceStaged = value != 0; return IDKPT_OK; }   // device-to-device copies through pinned host memory
    if (std::string(name) == "transport") {                                                                                   // how the members' bulk traffic travels (transport_rccl.hpp)
        GREQ(value >= 0 && value <= 2, "idkptSetDeveloperOption: transport is 0 (RCCL where usable), 1 (peer copies) or 2 (RCCL required)");
        if (c->n() > 1) GFLUSH();
        c->transportOpt = value;
        return IDKPT_OK;
    }
    REPLICATE(SetOption, name, value);
}
int32_t idkptBuildTlas(idkpt_ctx* c, const GpuTlasNode* nodes, int32_t nodeCount) { REPLICATE_VERSIONED(BuildTlas, nodes, nodeCount); }
int32_t idkptBuildTlasOnDevice(idkpt_ctx* c, int32_t searchRadius) { REPLICATE_VERSIONED(BuildTlasOnDevice, searchRadius); }      // every member rebuilds its own copy (0.1 ms; cheaper than shipping it)
int32_t idkptRefitBlas(idkpt_ctx* c, int32_t blasId) { REPLICATE_VERSIONED(RefitBlas, blasId); }
int32_t idkptBuildBlasCore(idkpt_ctx* c, const float* fragmentBoxes, int32_t fragmentCount, GpuBlasNode* outNodes, int32_t* outSortedIdsX, int32_t* outLevels)
{
    if (!c) return IDKPT_ERR_INVALID_ARGUMENT;
    dev_ctx* m = c->dev[0];                                     // a host-side service: runs on the first device, touches no scene state
    const int rc = dev_BuildBlasCore(m, fragmentBoxes, fragmentCount, outNodes, outSortedIdsX, outLevels);
    return rc ? mfail(c, m, rc) : IDKPT_OK;
}
int32_t idkptBuildBlas(idkpt_ctx* c, const float* positions, int32_t vertexCount, const GpuBlasTriangle* triangles, int32_t triangleCount, int32_t isRefittable, float preSplitFactor, idkpt_blas_build_info* outInfo)
{
    if (!c) return IDKPT_ERR_INVALID_ARGUMENT;
    dev_ctx* m = c->dev[0];                                     // a host-side service like idkptBuildBlasCore: first device, no scene state
    const int rc = dev_BuildBlas(m, positions, vertexCount, triangles, triangleCount, isRefittable, preSplitFactor, outInfo);
    return rc ? mfail(c, m, rc) : IDKPT_OK;
}
int32_t idkptBuildBlasFetch(idkpt_ctx* c, GpuBlasNode* outNodes, GpuBlasTriangle* outTriangles, int32_t* outParentIndices, int32_t* outLeafIndices)
{
    if (!c) return IDKPT_ERR_INVALID_ARGUMENT;
    dev_ctx* m = c->dev[0];
    const int rc = dev_BuildBlasFetch(m, outNodes, outTriangles, outParentIndices, outLeafIndices);
    return rc ? mfail(c, m, rc) : IDKPT_OK;
}
int32_t idkptCbrtProbe(idkpt_ctx* c, const float* in, float* out, int32_t n) { if (!c) return IDKPT_ERR_INVALID_ARGUMENT; dev_ctx* m = c->dev[0]; const int rc = dev_CbrtProbe(m, in, out, n); return rc ? mfail(c, m, rc) : IDKPT_OK; }
int32_t idkptUploadUnskinnedVertices(idkpt_ctx* c, const GpuUnskinnedVertex* verts, int32_t count) { REPLICATE(UploadUnskinnedVertices, verts, count); }
int32_t idkptSkin(idkpt_ctx* c, uint32_t inOff, uint32_t outOff, uint32_t jointOff, uint32_t count) { REPLICATE_VERSIONED(Skin, inOff, outOff, jointOff, count); }
int32_t idkptDownloadBuffer(idkpt_ctx* c, int32_t which, size_t offsetBytes, size_t bytes, void* dst)
{
    if (!c) return IDKPT_ERR_INVALID_ARGUMENT;
    ONE(dev_DownloadBuffer(m, which, offsetBytes, bytes, dst));
    GFLUSH();
    int rc = dev_DownloadBuffer(c->dev[0], which, offsetBytes, bytes, dst);       // the scene is replicated: any member's copy
    return rc ? mfail(c, c->dev[0], rc) : IDKPT_OK;
}

// ray queries are independent: the array is cut into one contiguous piece per device
int32_t idkptTraceRays(idkpt_ctx* c, const idkpt_ray* rays, size_t count, uint32_t flags, idkpt_hit* hits)
{
    if (!c) return IDKPT_ERR_INVALID_ARGUMENT;
    ONE(dev_TraceRays(m, rays, count, flags, hits));
    GREQ(count == 0 || (rays && hits), "idkptTraceRays: null rays/hits");
    GFLUSH();
    const size_t n = c->n(), per = (count + n - 1) / n;
    for (size_t d = 0; d < n; d++) {
        const size_t first = std::min(count, d * per), cnt = std::min(count, first + per) - first;
        if (cnt == 0) continue;
        int rc = dev_TraceRaysIssue(c->dev[d], rays + first, cnt, flags, hits + first); if (rc) { (void)group_sync(c); return mfail(c, c->dev[d], rc); }
    }
    return group_sync(c);
}
int32_t idkptTraceShadows(idkpt_ctx* c, const idkpt_shadow_params* p, const float* depth, const float* normalOct, float* visibility)
{
    if (!c) return IDKPT_ERR_INVALID_ARGUMENT;
    ONE(dev_TraceShadows(m, p, depth, normalOct, visibility));
    GFLUSH();
    int rc = dev_TraceShadows(c->dev[0], p, depth, normalOct, visibility);          // one shadow map: runs on the first device
    return rc ? mfail(c, c->dev[0], rc) : IDKPT_OK;
}

// the same queries on buffers that already live on the context's device (a host's own G-buffer / ray buffers): no copies, asynchronous in stream order
int32_t idkptTraceRaysDevice(idkpt_ctx* c, const idkpt_ray* dRays, size_t count, uint32_t flags, idkpt_hit* dHits)
{
    if (!c) return IDKPT_ERR_INVALID_ARGUMENT;
    ONE(dev_TraceRaysDevice(m, dRays, count, flags, dHits));
    return gfail(c, IDKPT_ERR_INVALID_OPERATION, "idkptTraceRaysDevice: device pointers belong to one device; use a single-device context (or idkptTraceRays)");
}
int32_t idkptTraceShadowsDevice(idkpt_ctx* c, const idkpt_shadow_params* p, const float* dDepth, const float* dNormalOct, float* dVisibility)
{
    if (!c) return IDKPT_ERR_INVALID_ARGUMENT;
    ONE(dev_TraceShadowsDevice(m, p, dDepth, dNormalOct, dVisibility));
    return gfail(c, IDKPT_ERR_INVALID_OPERATION, "idkptTraceShadowsDevice: device pointers belong to one device; use a single-device context (or idkptTraceShadows)");
}

int32_t idkptSetFrameRing(idkpt_ctx* c, int32_t frames) { REPLICATE(SetFrameRing, frames); }
int32_t idkptBeginFrame(idkpt_ctx* c, int32_t* outSlot)
{
    if (!c) return IDKPT_ERR_INVALID_ARGUMENT;
    ONE(dev_BeginFrame(m, outSlot));
    int32_t slot = 0;
    for (dev_ctx* m : c->dev) { int rc = dev_BeginFrame(m, &slot); if (rc) return mfail(c, m, rc); }
    if (outSlot) *outSlot = slot;
    return IDKPT_OK;
}
int32_t idkptDownloadFrame(idkpt_ctx* c, int32_t slot, int32_t image, float* rgba, size_t bytes)
{
    if (!c || !rgba) return IDKPT_ERR_INVALID_ARGUMENT;
    ONE(dev_DownloadFrame(m, slot, image, rgba, bytes));
    return group_download_image(c, image, slot, rgba, bytes);
}
int32_t idkptGetFrameDevicePtr(idkpt_ctx* c, int32_t slot, int32_t image, void** outPtr, size_t* outBytes)
{
    if (!c || !outPtr) return IDKPT_ERR_INVALID_ARGUMENT;
    ONE(dev_GetFrameDevicePtr(m, slot, image, outPtr, outBytes));
    return group_gather_device(c, image, slot, outPtr, outBytes);
}

int32_t idkptResetAccumulation(idkpt_ctx* c) { if (!c) return IDKPT_ERR_INVALID_ARGUMENT; for (dev_ctx* m : c->dev) dev_ResetAccumulation(m); return IDKPT_OK; }
int32_t idkptSetSampleSequence(idkpt_ctx* c, uint32_t first, uint32_t stride) { REPLICATE(SetSampleSequence, first, stride); }
int32_t idkptGetAccumulatedSamples(idkpt_ctx* c, uint32_t* out) { if (!c || !out) return IDKPT_ERR_INVALID_ARGUMENT; return dev_GetAccumulatedSamples(c->dev[0], out); }

int32_t idkptRender(idkpt_ctx* c)
{
    if (!c) return IDKPT_ERR_INVALID_ARGUMENT;
    ONE(dev_Render(m));
    if (c->W <= 0 || !c->frameOk) return gfail(c, IDKPT_ERR_INVALID_OPERATION, "idkptRender: no frame buffers (idkptSetSize not called, or the last re-layout of the devices failed)");
    // every member queues the same samples for its rows; the batch is launched on all of them together.  The general path (debug
    // traversal-cost view) is never batched.
    const int limit = fast_path(c->dev[0]) ? c->maxBatch : 1;
    for (int i = 0; i < c->st.SamplesPerPixel; i++) {
        for (dev_ctx* m : c->dev) {                   // one sample per member and round, so that the batch limit is honoured between the samples of one call
            const int spp = m->st.SamplesPerPixel; m->st.SamplesPerPixel = 1;
            const int rc = dev_Render(m);
            m->st.SamplesPerPixel = spp;
            if (rc) { for (dev_ctx* o : c->dev) o->pending.clear(); c->pending = 0; return mfail(c, m, rc); }
        }
        if (++c->pending >= limit) GFLUSH();
    }
    return IDKPT_OK;
}
int32_t idkptFlush(idkpt_ctx* c) { if (!c) return IDKPT_ERR_INVALID_ARGUMENT; ONE(dev_Flush(m)); return group_flush(c); }
int32_t idkptSynchronize(idkpt_ctx* c) { if (!c) return IDKPT_ERR_INVALID_ARGUMENT; ONE(dev_Synchronize(m)); GFLUSH(); return group_sync(c); }
int32_t idkptSetMaxBatch(idkpt_ctx* c, int32_t maxBatch)
{
    if (!c) return IDKPT_ERR_INVALID_ARGUMENT;
    ONE(dev_SetMaxBatch(m, maxBatch));
    GREQ(maxBatch >= 1 && maxBatch <= MAX_BATCH, "idkptSetMaxBatch: 1..256");
    GFLUSH();
    ALL(dev_SetMaxBatch(m, maxBatch));
    c->maxBatch = maxBatch;
    return IDKPT_OK;
}

int32_t idkptDownload(idkpt_ctx* c, int32_t image, float* rgba, size_t bytes)
{
    if (!c || !rgba) return IDKPT_ERR_INVALID_ARGUMENT;
    ONE(dev_Download(m, image, rgba, bytes));
    return group_download_image(c, image, c->dev[0]->curSlot, rgba, bytes);
}
int32_t idkptGetImageDevicePtr(idkpt_ctx* c, int32_t image, void** outPtr, size_t* outBytes)
{
    if (!c || !outPtr) return IDKPT_ERR_INVALID_ARGUMENT;
    ONE(dev_GetImageDevicePtr(m, image, outPtr, outBytes));
    return group_gather_device(c, image, c->dev[0]->curSlot, outPtr, outBytes);
}

// parity accessors: per-pixel state of the whole frame, reassembled from the members' rows
int32_t idkptDownloadRays(idkpt_ctx* c, GpuWavefrontRay* out, size_t bytes)
{
    if (!c || !out) return IDKPT_ERR_INVALID_ARGUMENT;
    ONE(dev_DownloadRays(m, out, bytes));
    GREQ(bytes == (size_t)c->W * c->H * sizeof(GpuWavefrontRay) && bytes > 0, "idkptDownloadRays: bytes must equal pixelCount*48");
    GFLUSH();
    for (size_t d = 0; d < c->n(); d++) {
        dev_ctx* m = c->dev[d];
        const size_t N = (size_t)c->W * m->rows;
        std::vector<GpuWavefrontRay> tmp(N);
        int rc = dev_DownloadRays(m, tmp.data(), N * sizeof(GpuWavefrontRay)); if (rc) return mfail(c, m, rc);
        for (size_t i = 0; i < N; i++) out[global_pixel(c, (int)d, (uint32_t)i)] = tmp[i];
    }
    return IDKPT_OK;
}
int32_t idkptDownloadAliveQueue(idkpt_ctx* c, uint32_t* indices, size_t capacityElems, uint32_t* outCount)
{
    if (!c || !outCount) return IDKPT_ERR_INVALID_ARGUMENT;
    ONE(dev_DownloadAliveQueue(m, indices, capacityElems, outCount));
    GFLUSH();
    // the one-device queue is in increasing pixel order (sorting off): strips concatenate, interleaved rows merge
    if (c->st.DoRaySorting && c->st.RayDepth > 2) return gfail(c, IDKPT_ERR_INVALID_OPERATION, "idkptDownloadAliveQueue: with DoRaySorting every device sorts its own queue; there is no single queue to return");
    std::vector<uint32_t> all;
    for (size_t d = 0; d < c->n(); d++) {
        uint32_t cnt = 0;
        int rc = dev_DownloadAliveQueue(c->dev[d], nullptr, 0, &cnt); if (rc) return mfail(c, c->dev[d], rc);
        std::vector<uint32_t> q(cnt);
        if (cnt) { rc = dev_DownloadAliveQueue(c->dev[d], q.data(), cnt, &cnt); if (rc) return mfail(c, c->dev[d], rc); }
        for (uint32_t v : q) all.push_back(global_pixel(c, (int)d, v));
    }
    if (!c->strips) std::sort(all.begin(), all.end());
    *outCount = (uint32_t)all.size();
    if (indices && !all.empty()) { GREQ(capacityElems >= all.size(), "idkptDownloadAliveQueue: capacity too small"); memcpy(indices, all.data(), all.size() * 4); }
    return IDKPT_OK;
}
int32_t idkptEnablePrimaryHitCapture(idkpt_ctx* c, int32_t enable) { if (!c) return IDKPT_ERR_INVALID_ARGUMENT; for (dev_ctx* m : c->dev) dev_EnablePrimaryHitCapture(m, enable); return IDKPT_OK; }
int32_t idkptDownloadPrimaryHits(idkpt_ctx* c, float* t, uint32_t* triangleId, float* baryXY, size_t pixelCount)
{
    if (!c || !t || !triangleId || !baryXY) return IDKPT_ERR_INVALID_ARGUMENT;
    ONE(dev_DownloadPrimaryHits(m, t, triangleId, baryXY, pixelCount));
    GREQ(pixelCount == (size_t)c->W * c->H, "idkptDownloadPrimaryHits: pixelCount mismatch");
    GFLUSH();
    for (size_t d = 0; d < c->n(); d++) {
        dev_ctx* m = c->dev[d];
        const size_t N = (size_t)c->W * m->rows;
        std::vector<float> tt(N), bb(2 * N); std::vector<uint32_t> ii(N);
        int rc = dev_DownloadPrimaryHits(m, tt.data(), ii.data(), bb.data(), N); if (rc) return mfail(c, m, rc);
        for (size_t i = 0; i < N; i++) { const uint32_t g = global_pixel(c, (int)d, (uint32_t)i); t[g] = tt[i]; triangleId[g] = ii[i]; baryXY[2 * (size_t)g] = bb[2 * i]; baryXY[2 * (size_t)g + 1] = bb[2 * i + 1]; }
    }
    return IDKPT_OK;
}

int32_t idkptGetStats(idkpt_ctx* c, idkpt_stats* out)
{
    if (!c || !out) return IDKPT_ERR_INVALID_ARGUMENT;
    ONE(dev_GetStats(m, out));
    GFLUSH();
    idkpt_stats sum; memset(&sum, 0, sizeof(sum));
    for (size_t d = 0; d < c->n(); d++) {
        idkpt_stats s; int rc = dev_GetStats(c->dev[d], &s); if (rc) return mfail(c, c->dev[d], rc);
        sum.RaysTraced += s.RaysTraced; sum.PrimaryRays += s.PrimaryRays; sum.NodePairVisits += s.NodePairVisits; sum.TriangleTests += s.TriangleTests;
        sum.WideFlaggedRays += s.WideFlaggedRays; sum.WideNodeVisits += s.WideNodeVisits; sum.WideLeafRecords += s.WideLeafRecords; sum.WideTriangleTests += s.WideTriangleTests; sum.InstTlasFlaggedRays += s.InstTlasFlaggedRays;
        sum.PacketFlaggedRays += s.PacketFlaggedRays; sum.PacketPackets += s.PacketPackets; sum.PacketNodeSteps += s.PacketNodeSteps; sum.PacketLiveLanes += s.PacketLiveLanes; sum.PacketRaysEntered += s.PacketRaysEntered; sum.PacketTriangleRounds += s.PacketTriangleRounds;
        sum.InstUnifiedLaunches += s.InstUnifiedLaunches; if (d == 0) { sum.InstUnifiedEntries = s.InstUnifiedEntries; sum.InstUnifiedTopDepth = s.InstUnifiedTopDepth; }
        for (int j = 0; j < 16; j++) sum.LastAliveCounts[j] += s.LastAliveCounts[j];
        sum.LastFrameMs = std::max(sum.LastFrameMs, s.LastFrameMs); sum.LastTraceMs = std::max(sum.LastTraceMs, s.LastTraceMs);       // the devices run side by side
        sum.TraceMsTotal = std::max(sum.TraceMsTotal, s.TraceMsTotal);
        if (d == 0) { sum.Frames = s.Frames; sum.TraceLaunches = s.TraceLaunches; }
    }
    *out = sum;
    return IDKPT_OK;
}
int32_t idkptGetStatsSized(idkpt_ctx* c, void* out, size_t bytes)
{
    if (!c || !out) return IDKPT_ERR_INVALID_ARGUMENT;
    idkpt_stats s; const int32_t rc = idkptGetStats(c, &s); if (rc) return rc;
    memcpy(out, &s, std::min(bytes, sizeof(s)));
    return IDKPT_OK;
}
int32_t idkptResetStats(idkpt_ctx* c) { REPLICATE(ResetStats); }
int32_t idkptEnableCounters(idkpt_ctx* c, int32_t enable) { if (!c) return IDKPT_ERR_INVALID_ARGUMENT; ONE(dev_EnableCounters(m, enable)); GFLUSH(); for (dev_ctx* m : c->dev) dev_EnableCounters(m, enable); return IDKPT_OK; }
int32_t idkptEnableTiming(idkpt_ctx* c, int32_t enable) { if (!c) return IDKPT_ERR_INVALID_ARGUMENT; for (dev_ctx* m : c->dev) dev_EnableTiming(m, enable); return IDKPT_OK; }

int32_t idkptSetStream(idkpt_ctx* c, void* hipStream)
{
    if (!c) return IDKPT_ERR_INVALID_ARGUMENT;
    ONE(dev_SetStream(m, hipStream));
    return gfail(c, IDKPT_ERR_INVALID_OPERATION, "idkptSetStream: a multi-device context owns one stream per device");
}
int32_t idkptGetStream(idkpt_ctx* c, void** out) { if (!c || !out) return IDKPT_ERR_INVALID_ARGUMENT; return dev_GetStream(c->dev[0], out); }   // device 0's stream: work on it is ordered behind idkptGetImageDevicePtr's gather

} // extern "C"
