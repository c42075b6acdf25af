// Context lifecycle, device memory helpers, measurement hooks and the KZG commit / open entry points.
#include <algorithm>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include "common.cuh"
#include "host_math.hpp"
#include "kernels.h"
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace plk {

int32_t set_err(plk_ctx* ctx, int32_t code, const char* what, hipError_t e, int line) {
    if (ctx) {
        char buf[512];
        snprintf(buf, sizeof(buf), "%s failed: %s (line %d)", what, hipGetErrorString(e), line);
        ctx->err = buf;
    }
    return code;
}

static bool prof_event(plk_ctx* ctx, hipEvent_t* e) {
    if (!ctx->prof_free.empty()) { *e = ctx->prof_free.back(); ctx->prof_free.pop_back(); return true; }
    return hipEventCreate(e) == hipSuccess;
}
void prof_begin(plk_ctx* ctx, hipStream_t s, const char* name) {
    ctx->prof_open = false;
    if (!ctx->profiling || (ctx->profiling == 2 && strcmp(name, "msm_accumulate") != 0)) return;
    PendingEvent p;
    p.name = name;
    if (!prof_event(ctx, &p.e0)) return;
    if (!prof_event(ctx, &p.e1)) { ctx->prof_free.push_back(p.e0); return; }
    hipEventRecord(p.e0, s);
    ctx->pending.push_back(p);
    ctx->prof_open = true;
}
void prof_end(plk_ctx* ctx, hipStream_t s) {
    if (!ctx->prof_open) return;
    hipEventRecord(ctx->pending.back().e1, s);
    ctx->prof_open = false;
}
// Books the brackets whose closing event has completed; with wait = true it blocks for all of them.  Start and end are also
// kept relative to the event recorded at the last reset, so that the time during which at least one bracket of a name was
// open (kernels of one name overlap when they run on different streams) can be told from the sum of the durations.
void prof_collect(plk_ctx* ctx, bool wait) {
    size_t kept = 0;
    for (size_t i = 0; i < ctx->pending.size(); ++i) {
        PendingEvent& p = ctx->pending[i];
        const hipError_t st = wait ? hipEventSynchronize(p.e1) : hipEventQuery(p.e1);
        if (st == hipErrorNotReady) { ctx->pending[kept++] = p; continue; }
        float ms = 0, t0 = 0;
        if (st == hipSuccess && hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            ProfEntry& e = ctx->prof[p.name];
            e.ms += ms;
            e.launches += 1;
            if (ctx->prof_base && hipEventElapsedTime(&t0, ctx->prof_base, p.e0) == hipSuccess) e.spans.push_back({(double)t0, (double)t0 + ms});
        }
        ctx->prof_free.push_back(p.e0);
        ctx->prof_free.push_back(p.e1);
    }
    ctx->pending.resize(kept);
}

int32_t ensure_scratch(plk_ctx* ctx, uint64_t bytes) {
    if (bytes <= ctx->scratch_cap) return PLK_OK;
    PLK_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->scratch) PLK_HIP(hipFree(ctx->scratch));
    ctx->scratch = nullptr; ctx->scratch_cap = 0;
    PLK_HIP(hipMalloc(&ctx->scratch, bytes));
    ctx->scratch_cap = bytes;
    return PLK_OK;
}

static int32_t grow(plk_ctx* ctx, hipStream_t st, uint32_t** buf, uint64_t* cap, uint64_t words) {
    if (words <= *cap) return PLK_OK;
    PLK_HIP(hipStreamSynchronize(st));
    if (*buf) PLK_HIP(hipFree(*buf));
    *buf = nullptr; *cap = 0;
    PLK_HIP(hipMalloc(buf, words * 4));
    *cap = words;
    return PLK_OK;
}

// One stream after another: everything enqueued on `first` so far is ordered before whatever is enqueued on `then` from now on.
// The events come from a ring kept by the context (creating and destroying one per call cost a proof about forty pairs of
// runtime calls on its critical path).
int32_t order_streams(plk_ctx* ctx, hipStream_t first, hipStream_t then) {
    if (first == then) return PLK_OK;
    hipEvent_t& ev = ctx->order_ev[ctx->order_turn++ % plk_ctx::ORDER_EVENTS];
    if (!ev) PLK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    PLK_HIP(hipEventRecord(ev, first));
    PLK_HIP(hipStreamWaitEvent(then, ev, 0));
    return PLK_OK;
}
// The front work of every second job runs on the reduction stream and reads the job's polynomial there, while the calls that may
// overwrite that polynomial afterwards (transforms in place, copies, uploads, pointwise kernels) run on the front stream.  The job's
// own `sorted` event (recorded behind its division and sort) serves as the mark of its last read, and the next compute call on the
// front stream waits for it: the ordering costs nothing until somebody needs it (making the front stream wait at once would hold up
// the sort of the job after next; an event of its own between the sort's kernels cost 0.08 ms per proof of 2^20 gates).
int32_t lane_read_mark(plk_ctx* ctx, hipStream_t lane, hipEvent_t recorded) {
    if (ctx->detached || lane != ctx->st_red) return PLK_OK;
    ctx->lane_read = recorded;
    ctx->lane_read_pending = true;
    return PLK_OK;
}
int32_t compute_guard(plk_ctx* ctx) {
    if (!ctx->lane_read_pending || ctx->stream != ctx->st_sort) return PLK_OK;   // detached compute work shares the stream of those reads
    PLK_HIP(hipStreamWaitEvent(ctx->st_sort, ctx->lane_read, 0));
    ctx->lane_read_pending = false;
    return PLK_OK;
}

// orders the sort stream (the entry of the MSM pipeline) after everything already enqueued on the compute stream
static int32_t sort_after_compute(plk_ctx* ctx, hipStream_t lane) {
    if (ctx->detached) return PLK_OK;   // the caller promised that this job's inputs do not come from the detached compute work
    // The event also covers the divisions and sorts that earlier *_begin calls have put on the compute stream, which this job does
    // not need.  Waiting for them all the same is the better schedule: with an event that marks the end of the compute work only,
    // the two front streams run freely side by side, more sorting kernels share the chip with the accumulation kernels at a time,
    // and a proof of 2^20 gates takes 0.5 ms longer (measured on one box: 31.5 against 31.0 ms).
    return order_streams(ctx, ctx->stream, lane);
}

}  // namespace plk

using namespace plk;

// Every context of one device shares one set of four streams.  A process that opens many contexts would otherwise hold
// many more HIP streams than the device has hardware queues (four per priority level here), and dependent work of
// different contexts multiplexed onto one hardware queue was seen to stop making progress.
namespace {
struct DeviceStreams {
    hipStream_t stream = nullptr, st_sort = nullptr, st_acc[2] = {nullptr, nullptr}, st_red = nullptr;
    int users = 0;
};
constexpr size_t MAX_DEVICE_STREAMS = 4;   // hardware queues a process gets per priority level (GPU_MAX_HW_QUEUES)
std::mutex g_streams_mutex;
std::map<int, DeviceStreams> g_streams;

bool streams_acquire(plk_ctx* ctx) {
    std::lock_guard<std::mutex> lock(g_streams_mutex);
    DeviceStreams& d = g_streams[ctx->device];
    if (d.users == 0) {
        // All streams have the same priority on purpose.  A higher-priority queue whose head is a barrier packet waiting
        // for an event was seen to starve a normal-priority queue on the same hardware pipe; when that queue carries the
        // work the barrier waits for, the process hangs.
        // Four streams, because a process gets four hardware queues per priority level here and streams that share one are
        // serialised: the front stream (NTT, pointwise and scan kernels, MSM sorts), two accumulation streams, one for reductions.
        if (hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&d.st_acc[0], hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&d.st_acc[1], hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&d.st_red, hipStreamNonBlocking) != hipSuccess)
            return false;
        d.st_sort = d.stream;
        // The two rules above are enforced, not only written down: whoever adds a stream or a priority gets a refused plk_create
        // instead of a process that hangs under load.
        const hipStream_t all[] = {d.stream, d.st_sort, d.st_acc[0], d.st_acc[1], d.st_red};
        std::vector<hipStream_t> distinct;
        int prio0 = 0;
        for (hipStream_t st : all) {
            if (std::find(distinct.begin(), distinct.end(), st) != distinct.end()) continue;
            int prio = 0;
            if (hipStreamGetPriority(st, &prio) != hipSuccess) return false;
            if (distinct.empty()) prio0 = prio;
            if (prio != prio0) { fprintf(stderr, "libplonk_hip: streams of different priorities (%d, %d) are not allowed\n", prio0, prio); return false; }
            distinct.push_back(st);
        }
        if (distinct.size() > MAX_DEVICE_STREAMS) { fprintf(stderr, "libplonk_hip: %zu streams per device, at most %zu hardware queues\n", distinct.size(), MAX_DEVICE_STREAMS); return false; }
    }
    ++d.users;
    ctx->stream = d.stream; ctx->st_sort = d.st_sort; ctx->st_acc[0] = d.st_acc[0]; ctx->st_acc[1] = d.st_acc[1]; ctx->st_red = d.st_red;
    return true;
}
void streams_release(plk_ctx* ctx) {
    std::lock_guard<std::mutex> lock(g_streams_mutex);
    auto it = g_streams.find(ctx->device);
    if (it == g_streams.end() || !ctx->stream) return;
    if (--it->second.users == 0) {
        DeviceStreams& d = it->second;
        hipStreamDestroy(d.stream); hipStreamDestroy(d.st_acc[0]); hipStreamDestroy(d.st_red);
        if (d.st_sort != d.stream) hipStreamDestroy(d.st_sort);
        if (d.st_acc[1] != d.st_acc[0]) hipStreamDestroy(d.st_acc[1]);
        g_streams.erase(it);
    }
}
}  // namespace

// ---- ordinary (pageable) host arrays -----------------------------------------------------------
// hipMemcpyAsync from pageable memory is staged by the runtime on the calling thread: 26 GB/s on the MI355X host against 52 GB/s from
// page-locked memory, and the call blocks meanwhile.  A caller that hands over ordinary vectors (the reference's &DensePolynomial /
// &[Fr]) gets the same link rate here: the array goes through a ring of page-locked buffers, copied into them by a few helper
// threads and the calling thread together (one thread moves 24-27 GB/s, four 75 GB/s), each buffer's transfer enqueued as soon
// as it is full.  hipHostRegister, which would avoid the copy, is refused on this system (invalid argument).
namespace {
struct StagePool {
    static constexpr int SLOTS = 4, HELPERS = 3;
    static constexpr size_t CHUNK = 16u << 20;
    char* buf[SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t copied[SLOTS] = {nullptr, nullptr, nullptr, nullptr};   // the transfer out of the buffer has finished
    bool used[SLOTS] = {false, false, false, false};
    unsigned next = 0;
    struct Task { char* dst; const char* src; size_t n; };
    std::vector<std::thread> helpers;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    std::deque<Task> queue;
    int pending = 0;
    bool stop = false;

    void helper() {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lock(m);
                cv_work.wait(lock, [&] { return stop || !queue.empty(); });
                if (queue.empty()) return;   // stop
                t = queue.front();
                queue.pop_front();
            }
            memcpy(t.dst, t.src, t.n);
            {
                std::lock_guard<std::mutex> lock(m);
                if (--pending == 0) cv_done.notify_one();
            }
        }
    }
    // dst[0, n) = src[0, n), the helpers and the caller each taking a share; returns when all of it is in place
    void copy(char* dst, const char* src, size_t n) {
        const size_t parts = HELPERS + 1;
        const size_t per = ((n + parts - 1) / parts + 4095) & ~(size_t)4095;
        size_t own_off = 0, own_n = std::min(per, n);
        {
            std::lock_guard<std::mutex> lock(m);
            for (size_t off = per; off < n; off += per) {
                queue.push_back({dst + off, src + off, std::min(per, n - off)});
                ++pending;
            }
        }
        cv_work.notify_all();
        memcpy(dst + own_off, src + own_off, own_n);
        std::unique_lock<std::mutex> lock(m);
        cv_done.wait(lock, [&] { return pending == 0; });
    }
    bool init() {
        for (int i = 0; i < SLOTS; ++i) {
            if (hipHostMalloc((void**)&buf[i], CHUNK, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return false; }
            if (hipEventCreateWithFlags(&copied[i], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
        }
        for (int i = 0; i < HELPERS; ++i) helpers.emplace_back([this] { helper(); });
        return true;
    }
    ~StagePool() {
        {
            std::lock_guard<std::mutex> lock(m);
            stop = true;
        }
        cv_work.notify_all();
        for (auto& t : helpers) t.join();
        for (int i = 0; i < SLOTS; ++i) {
            if (copied[i]) (void)hipEventDestroy(copied[i]);
            if (buf[i]) (void)hipHostFree(buf[i]);
        }
    }
};
StagePool* stage_pool(plk_ctx* ctx) {
    if (!ctx->stage) {
        StagePool* sp = new StagePool();
        if (!sp->init()) { delete sp; return nullptr; }
        ctx->stage = sp;
    }
    return static_cast<StagePool*>(ctx->stage);
}
// page-locked (hipHostMalloc / registered) host memory: the runtime transfers from it directly
bool host_memory_is_pinned(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}
}  // namespace

extern "C" int32_t plk_create(int32_t device, plk_ctx** out) {
    if (!out) return PLK_ERR_INVALID_ARG;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return PLK_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return PLK_ERR_INVALID_ARG;
    plk_ctx* ctx = new plk_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || !streams_acquire(ctx)) {
        delete ctx;
        return PLK_ERR_HIP;
    }
    if (hipMalloc(&ctx->flag_dev, 64) != hipSuccess || hipHostMalloc(&ctx->flag_host, 64, hipHostMallocDefault) != hipSuccess) {
        streams_release(ctx);
        delete ctx;
        return PLK_ERR_HIP;
    }
    if (static_scratch_init(ctx) != PLK_OK) {
        hipFree(ctx->flag_dev); hipHostFree(ctx->flag_host);
        streams_release(ctx);
        delete ctx;
        return PLK_ERR_HIP;
    }
    *out = ctx;
    return PLK_OK;
}
extern "C" int32_t plk_destroy(plk_ctx* ctx) {
    if (!ctx) return PLK_ERR_INVALID_ARG;
    hipSetDevice(ctx->device);
    hipDeviceSynchronize();
    if (ctx->detached) { ctx->stream = ctx->st_sort; ctx->detached = false; }
    prof_collect(ctx, true);
    if (ctx->prof_base) hipEventDestroy(ctx->prof_base);
    for (hipEvent_t e : ctx->prof_free) hipEventDestroy(e);
    for (hipEvent_t e : ctx->order_ev) if (e) hipEventDestroy(e);
    hipFree(ctx->batch_dev); hipHostFree(ctx->batch_host);
    if (ctx->comm) plk_comm_destroy(ctx);
    for (auto& s : ctx->slot) slot_destroy(s);
    delete static_cast<StagePool*>(ctx->stage);
    ctx->stage = nullptr;
    ntt_release(ctx);
    hipFree(ctx->tables); hipFree(ctx->tables_mid); hipFree(ctx->tables_short); hipFree(ctx->marg_all_dev); hipHostFree(ctx->marg_all_host); hipFree(ctx->scratch); hipFree(ctx->flag_dev); hipHostFree(ctx->flag_host);
    streams_release(ctx);
    delete ctx;
    return PLK_OK;
}
extern "C" const char* plk_last_error(plk_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" int32_t plk_malloc(plk_ctx* ctx, uint64_t bytes, void** dev) {
    if (!ctx || !dev) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    if (hipMalloc(dev, bytes ? bytes : 32) != hipSuccess) { ctx->err = "device allocation failed"; return PLK_ERR_ALLOC; }
    return PLK_OK;
}
extern "C" int32_t plk_free(plk_ctx* ctx, void* dev) {
    if (!ctx) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    PLK_HIP(hipFree(dev));
    return PLK_OK;
}
extern "C" int32_t plk_upload(plk_ctx* ctx, void* dev_dst, const void* host_src, uint64_t bytes) {
    if (!ctx || (bytes && (!dev_dst || !host_src))) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    PLK_TRY(compute_guard(ctx));
    PLK_HIP(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    PLK_HIP(hipStreamSynchronize(ctx->stream));
    return PLK_OK;
}
// Upload that does not wait for the transfer: the copy runs on the detached stream (st_red, idle at the start of a round), and the stream
// that carries the compute work and the MSM sorts is ordered after it from this call on.  Later calls that read dev_dst therefore see the
// data, earlier ones are not held up, and the transfers of several such calls follow each other without gaps.  host_src may be reused
// when the call returns if it is ordinary memory (it has been copied out), page-locked memory must stay untouched until the transfer
// has run (plk_sync or any later result that depends on it).
extern "C" int32_t plk_upload_enqueue(plk_ctx* ctx, void* dev_dst, const void* host_src, uint64_t bytes) {
    if (!ctx || (bytes && (!dev_dst || !host_src))) return PLK_ERR_INVALID_ARG;
    if (!bytes) return PLK_OK;
    PLK_HIP(hipSetDevice(ctx->device));
    // The transfers use the reduction stream as their copy queue.  The first one of a batch is ordered behind everything the front
    // stream holds (sorts, divisions and compute work of earlier calls that may still read or write dev_dst); the later ones follow it
    // without waiting for the sorts begun in between, which read other buffers (see include/plonk_hip.h).
    if (!ctx->uploads_pending) PLK_TRY(order_streams(ctx, ctx->st_sort, ctx->st_red));
    ctx->uploads_pending = true;
    StagePool* sp = bytes >= (1u << 20) && !host_memory_is_pinned(host_src) ? stage_pool(ctx) : nullptr;
    if (sp) {
        // ordinary memory: through the page-locked ring, one buffer's worth at a time; the call returns when the last piece has been
        // copied into the ring and its transfer enqueued (the transfers themselves are not waited for)
        const char* src = static_cast<const char*>(host_src);
        char* dst = static_cast<char*>(dev_dst);
        for (uint64_t off = 0; off < bytes; off += StagePool::CHUNK) {
            const size_t n = (size_t)std::min<uint64_t>(StagePool::CHUNK, bytes - off);
            const int slot = (int)(sp->next++ % StagePool::SLOTS);
            if (sp->used[slot]) PLK_HIP(hipEventSynchronize(sp->copied[slot]));   // the buffer's previous transfer
            sp->copy(sp->buf[slot], src + off, n);
            PLK_HIP(hipMemcpyAsync(dst + off, sp->buf[slot], n, hipMemcpyHostToDevice, ctx->st_red));
            PLK_HIP(hipEventRecord(sp->copied[slot], ctx->st_red));
            sp->used[slot] = true;
        }
    } else {
        PLK_HIP(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->st_red));
    }
    return order_streams(ctx, ctx->st_red, ctx->st_sort);
}
extern "C" int32_t plk_host_malloc(plk_ctx* ctx, uint64_t bytes, void** host) {
    if (!ctx || !host) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    if (hipHostMalloc(host, bytes ? bytes : 32, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); ctx->err = "pinned host allocation failed"; return PLK_ERR_ALLOC; }
    return PLK_OK;
}
extern "C" int32_t plk_host_free(plk_ctx* ctx, void* host) {
    if (!ctx) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    PLK_HIP(hipHostFree(host));
    return PLK_OK;
}
extern "C" int32_t plk_download(plk_ctx* ctx, void* host_dst, const void* dev_src, uint64_t bytes) {
    if (!ctx || (bytes && (!host_dst || !dev_src))) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    PLK_HIP(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PLK_HIP(hipStreamSynchronize(ctx->stream));
    return PLK_OK;
}
extern "C" int32_t plk_copy_dev(plk_ctx* ctx, void* dev_dst, const void* dev_src, uint64_t bytes, uint64_t total_bytes) {
    if (!ctx || !dev_dst || (bytes && !dev_src) || total_bytes < bytes) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    PLK_TRY(compute_guard(ctx));
    if (bytes) PLK_HIP(hipMemcpyAsync(dev_dst, dev_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    if (total_bytes > bytes) PLK_HIP(hipMemsetAsync((char*)dev_dst + bytes, 0, total_bytes - bytes, ctx->stream));
    return PLK_OK;
}
extern "C" int32_t plk_compute_detach(plk_ctx* ctx, int32_t on) {
    if (!ctx) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    if (on < 0 || on > 3) return PLK_ERR_INVALID_ARG;
    const bool detach = on == 1 || on == 3;
    if (detach == ctx->detached) return PLK_OK;
    if (detach) {
        if (on == 1) PLK_TRY(order_streams(ctx, ctx->st_sort, ctx->st_red));   // detached work may read what earlier calls produced
        ctx->stream = ctx->st_red;
        ctx->detached = true;
    } else {
        if (on == 0) { PLK_TRY(order_streams(ctx, ctx->st_red, ctx->st_sort)); ctx->lane_read_pending = false; }   // later calls may read what the detached work produced
        ctx->stream = ctx->st_sort;
        ctx->detached = false;
    }
    return PLK_OK;
}
extern "C" int32_t plk_sync(plk_ctx* ctx) {
    if (!ctx) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    PLK_HIP(hipDeviceSynchronize());
    prof_collect(ctx, false);
    return PLK_OK;
}

extern "C" int32_t plk_profile_enable(plk_ctx* ctx, int32_t on) {
    if (!ctx) return PLK_ERR_INVALID_ARG;
    ctx->profiling = on == 2 ? 2 : (on != 0 ? 1 : 0);
    return PLK_OK;
}
extern "C" int32_t plk_profile_reset(plk_ctx* ctx) {
    if (!ctx) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    prof_collect(ctx, true);
    ctx->prof.clear();
    if (!ctx->prof_base) PLK_HIP(hipEventCreate(&ctx->prof_base));
    PLK_HIP(hipEventRecord(ctx->prof_base, ctx->stream));
    PLK_HIP(hipEventSynchronize(ctx->prof_base));
    return PLK_OK;
}
extern "C" int32_t plk_profile_get(plk_ctx* ctx, const char* name, double* total_ms, uint64_t* launches) {
    if (!ctx || !name || !total_ms || !launches) return PLK_ERR_INVALID_ARG;
    prof_collect(ctx, true);
    auto it = ctx->prof.find(name);
    *total_ms = it == ctx->prof.end() ? 0.0 : it->second.ms;
    *launches = it == ctx->prof.end() ? 0 : it->second.launches;
    return PLK_OK;
}
extern "C" int32_t plk_profile_get_busy(plk_ctx* ctx, const char* name, double* busy_ms) {
    if (!ctx || !name || !busy_ms) return PLK_ERR_INVALID_ARG;
    prof_collect(ctx, true);
    *busy_ms = 0.0;
    auto it = ctx->prof.find(name);
    if (it == ctx->prof.end()) return PLK_OK;
    std::vector<std::pair<double, double>> spans = it->second.spans;
    std::sort(spans.begin(), spans.end());
    double lo = 0, hi = -1;
    for (const auto& sp : spans) {
        if (sp.first > hi) { if (hi > lo) *busy_ms += hi - lo; lo = sp.first; hi = sp.second; }
        else if (sp.second > hi) hi = sp.second;
    }
    if (hi > lo) *busy_ms += hi - lo;
    return PLK_OK;
}

// ---- RCCL communicator of the ranks that share one proof ---------------------------------------------------------------
// One process per GPU; every rank holds a share of the SRS, runs the same prover and contributes a partial sum (an unnormalised
// point of 128 bytes) to every commitment and opening.  Point addition is not a reduction RCCL offers, so the exchange is an
// all-gather and every rank adds the partial sums itself.  RCCL is opened at run time: single-GPU users need no librccl.
namespace {
struct RcclId { char internal[128]; };
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(RcclId*) = nullptr;
    int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;       // the four point-to-point symbols are optional:
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;             // without them the sharded transform is refused,
    int (*GroupStart)() = nullptr;                                                  // the exchange of partial sums still works
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mutex;
bool rccl_open(std::string* why) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.lib) return true;
    // a copy the process already holds (PyTorch brings its own) comes first: one RCCL per process is enough
    const char* names[] = {"librccl.so.1", "librccl.so"};
    void* h = nullptr;
    for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h) { *why = std::string("librccl not found: ") + dlerror(); return false; }
    Rccl r;
    r.lib = h;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(h, "ncclAllGather"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(h, "ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(h, "ncclRecv"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(h, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy || !r.GetErrorString) { *why = "librccl lacks an expected symbol"; return false; }
    g_rccl = r;
    return true;
}
int32_t rccl_fail(plk_ctx* ctx, const char* what, int rc) {
    ctx->err = std::string(what) + " failed: " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
    return PLK_ERR_HIP;
}
constexpr uint64_t COMM_MAX_POINTS = 64;    // partial sums per exchange (a round of the prover has at most nine)
constexpr int RCCL_UINT64 = 5;              // ncclUint64
}  // namespace

extern "C" int32_t plk_comm_get_unique_id(uint8_t id[128]) {
    if (!id) return PLK_ERR_INVALID_ARG;
    std::string why;
    if (!rccl_open(&why)) return PLK_ERR_HIP;
    RcclId u;
    if (g_rccl.GetUniqueId(&u) != 0) return PLK_ERR_HIP;
    memcpy(id, u.internal, 128);
    return PLK_OK;
}
extern "C" int32_t plk_comm_init(plk_ctx* ctx, const uint8_t id[128], uint32_t rank, uint32_t world) {
    if (!ctx || !id || world == 0 || rank >= world) return PLK_ERR_INVALID_ARG;
    if (ctx->comm) { ctx->err = "plk_comm_init: this context already has a communicator"; return PLK_ERR_INVALID_ARG; }
    std::string why;
    if (!rccl_open(&why)) { ctx->err = why; return PLK_ERR_HIP; }
    PLK_HIP(hipSetDevice(ctx->device));
    RcclId u;
    memcpy(u.internal, id, 128);
    void* comm = nullptr;
    const int rc = g_rccl.CommInitRank(&comm, (int)world, u, (int)rank);
    if (rc != 0) return rccl_fail(ctx, "ncclCommInitRank", rc);
    // the staging buffers first: a context whose communicator exists always has them
    void *send = nullptr, *recv = nullptr, *host = nullptr;
    if (hipMalloc(&send, COMM_MAX_POINTS * 128) != hipSuccess || hipMalloc(&recv, COMM_MAX_POINTS * 128 * world) != hipSuccess ||
        hipHostMalloc(&host, COMM_MAX_POINTS * 128 * (world + 1), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        hipFree(send); hipFree(recv); hipHostFree(host);
        g_rccl.CommDestroy(comm);
        ctx->err = "plk_comm_init: no memory for the staging buffers";
        return PLK_ERR_ALLOC;
    }
    ctx->comm = comm; ctx->comm_rank = rank; ctx->comm_world = world;
    ctx->comm_send = send; ctx->comm_recv = recv; ctx->comm_host = host;
    return PLK_OK;
}
extern "C" int32_t plk_comm_destroy(plk_ctx* ctx) {
    if (!ctx) return PLK_ERR_INVALID_ARG;
    if (!ctx->comm) return PLK_OK;
    PLK_HIP(hipSetDevice(ctx->device));
    PLK_HIP(hipDeviceSynchronize());
    g_rccl.CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    hipFree(ctx->comm_send); hipFree(ctx->comm_recv); hipHostFree(ctx->comm_host);
    ctx->comm_send = ctx->comm_recv = nullptr; ctx->comm_host = nullptr;
    return PLK_OK;
}
// mine[count][16] -> all[world][count][16] (host arrays): staged through pinned memory, ncclAllGather on the compute stream
extern "C" int32_t plk_comm_all_gather_partials(plk_ctx* ctx, const uint64_t* mine, uint64_t count, uint64_t* all) {
    if (!ctx || !mine || !all) return PLK_ERR_INVALID_ARG;
    if (!ctx->comm) { ctx->err = "plk_comm_all_gather_partials: no communicator (plk_comm_init)"; return PLK_ERR_INVALID_ARG; }
    if (count == 0) return PLK_OK;
    if (count > COMM_MAX_POINTS) { ctx->err = "plk_comm_all_gather_partials: more than 64 points in one exchange"; return PLK_ERR_INVALID_ARG; }
    PLK_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint64_t bytes = count * 128, world = ctx->comm_world;
    uint8_t* host_send = (uint8_t*)ctx->comm_host;
    uint8_t* host_recv = host_send + COMM_MAX_POINTS * 128;
    memcpy(host_send, mine, bytes);
    PLK_HIP(hipMemcpyAsync(ctx->comm_send, host_send, bytes, hipMemcpyHostToDevice, st));
    const int rc = g_rccl.AllGather(ctx->comm_send, ctx->comm_recv, count * 16, RCCL_UINT64, ctx->comm, st);
    if (rc != 0) return rccl_fail(ctx, "ncclAllGather", rc);
    PLK_HIP(hipMemcpyAsync(host_recv, ctx->comm_recv, bytes * world, hipMemcpyDeviceToHost, st));
    PLK_HIP(hipStreamSynchronize(st));
    memcpy(all, host_recv, bytes * world);
    return PLK_OK;
}

// ---- four-step NTT across the ranks of the communicator --------------------------------------------------------------
// all-to-all of this rank's M elements: block r of `send` goes to rank r, block g of rank r's buffer arrives as block r of `recv`.
// Grouped point-to-point transfers on the compute stream: xGMI links are point to point, so every pair of ranks uses its own link
// and no ring is involved; nothing waits on the host.
static int32_t exchange_blocks(plk_ctx* ctx, hipStream_t st, const uint32_t* send, uint32_t* recv, uint64_t block_words) {
    if (!g_rccl.Send || !g_rccl.Recv || !g_rccl.GroupStart || !g_rccl.GroupEnd) { ctx->err = "librccl lacks ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd"; return PLK_ERR_HIP; }
    const uint32_t world = ctx->comm_world;
    int rc = g_rccl.GroupStart();
    if (rc != 0) return rccl_fail(ctx, "ncclGroupStart", rc);
    for (uint32_t r = 0; r < world && rc == 0; ++r) {
        rc = g_rccl.Send(send + (size_t)r * block_words, block_words / 2, RCCL_UINT64, (int)r, ctx->comm, st);
        if (rc == 0) rc = g_rccl.Recv(recv + (size_t)r * block_words, block_words / 2, RCCL_UINT64, (int)r, ctx->comm, st);
    }
    const int rc_end = g_rccl.GroupEnd();
    if (rc != 0) return rccl_fail(ctx, "ncclSend / ncclRecv", rc);
    if (rc_end != 0) return rccl_fail(ctx, "ncclGroupEnd", rc_end);
    return PLK_OK;
}
extern "C" int32_t plk_ntt_sharded_dev(plk_ctx* ctx, void* data_dev, void* tmp_dev, uint32_t log_n, int32_t inverse, const uint64_t* coset_gen, uint32_t phase,
                                       uint32_t rank, uint32_t world, int32_t* result_in_tmp) {
    if (!ctx || !data_dev || phase > 3) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    PLK_TRY(compute_guard(ctx));
    hipStream_t st = ctx->stream;
    uint32_t* data = (uint32_t*)data_dev;
    if (phase == 1 || phase == 2) {       // the halves on their own: the caller moves the blocks (logical ranks on one device, another transport)
        if (result_in_tmp) *result_in_tmp = 0;
        return ntt_sharded_half(ctx, st, data, log_n, inverse != 0, coset_gen, rank, world, (int)phase);
    }
    if (!ctx->comm) { ctx->err = "plk_ntt_sharded_dev: no communicator (plk_comm_init)"; return PLK_ERR_INVALID_ARG; }
    if (!tmp_dev || !result_in_tmp) return PLK_ERR_INVALID_ARG;
    rank = ctx->comm_rank; world = ctx->comm_world;
    PLK_TRY(ntt_sharded_half(ctx, st, data, log_n, inverse != 0, coset_gen, rank, world, 1));
    uint32_t* out = data;
    if (world > 1 || phase == 3) {        // phase 3: the exchange also with a single rank (it sends to itself), so that one GPU exercises it
        uint32_t log_g = 0;
        while ((1u << log_g) < world) ++log_g;
        const uint64_t m = 1ull << (log_n - log_g);
        PLK_TRY(exchange_blocks(ctx, st, data, (uint32_t*)tmp_dev, m / world * 8));
        out = (uint32_t*)tmp_dev;
    }
    PLK_TRY(ntt_sharded_half(ctx, st, out, log_n, inverse != 0, coset_gen, rank, world, 2));
    *result_in_tmp = out != data ? 1 : 0;
    return PLK_OK;
}

// ---- KZG commit ------------------------------------------------------------------------------------
static void slot_release(plk_ctx* ctx, MsmSlot& s);
// Lanes for the members of a round that is begun with one call (0 = first front stream, 1 = second), in the order given: the
// alternation of single begins.  PLK_LANES_COMMIT / PLK_LANES_OPEN (a string of 0 and 1, one per member) override it for experiments.
// Measured at 2^20 gates on one box against the alternation (30.6-30.8 ms): the 4n-coefficient s_sigma alone on the second stream
// and everybody else on the first 31.3-31.5 ms; the same with the last members alternating again 31.4-31.5 ms; leader and s_sigma
// on the second stream 31.0 ms.  Front work that gets ahead of the accumulation kernels only adds sorting kernels beside them.
static void plan_lanes(plk_ctx* ctx, uint32_t count, const char* env_name, uint8_t* lane) {
    for (uint32_t k = 0; k < count; ++k) lane[k] = 0;
    if (ctx->detached || ctx->uploads_pending) return;
    const char* ov = getenv(env_name);
    if (ov && strlen(ov) >= count) { for (uint32_t k = 0; k < count; ++k) lane[k] = ov[k] == '1' ? 1 : 0; return; }
    for (uint32_t k = 0; k < count; ++k) lane[k] = (ctx->lane_turn++ & 1) ? 1 : 0;
}
// Staging of a batch's small per-member data, so that it crosses PCIe in one transfer per direction: [member][16] hiding scalars of
// openings, [member][8] remainders of their divisions, [member][24] blinding scalars of commitments.
constexpr size_t BATCH_HW = 0, BATCH_RES = (size_t)MSM_SLOTS * 16, BATCH_BLIND = (size_t)MSM_SLOTS * 24, BATCH_WORDS = (size_t)MSM_SLOTS * 48;
static int32_t batch_buffers(plk_ctx* ctx) {
    if (ctx->batch_dev) return PLK_OK;
    PLK_HIP(hipMalloc(&ctx->batch_dev, BATCH_WORDS * 4));
    PLK_HIP(hipHostMalloc(&ctx->batch_host, BATCH_WORDS * 4, hipHostMallocDefault));
    return PLK_OK;
}
// The MSMs of a round that is begun with one call, as launches: members whose share on this rank is short are put together, so
// that a round of a sharded or small proof is a few launches instead of three per member, and its reductions one or two instead of
// one per member (each costs about the same whatever the number of members, and they run one at a time).  Members on the wide plan
// are taken in the order given and share a launch as long as it stays below WIDE_GROUP_ENTRIES sorted entries (about 0.6 ms of
// accumulation: the reduction of one group hides beside the accumulation of the next); a long MSM stays alone.  The members on
// the narrow plan form one group: where its first member stands when it is a few terms only (the two-coefficient s_id: its result
// is there long before the round ends; enqueued last, its reduction would end the round behind everybody else's), behind the
// others otherwise (they are the longer ones; measured at 2^16 gates: 4.02-4.05 ms per proof against 4.40 ms with the group in front).
constexpr uint64_t WIDE_GROUP_ENTRIES = 9ull << 20;
struct LaunchUnits {
    uint32_t n = 0;
    uint32_t size[MSM_SLOTS];
    uint32_t member[MSM_SLOTS][MSM_SLOTS];
    int plan[MSM_SLOTS];
};
static void plan_units(uint32_t count, const uint64_t* share, const int* plan, LaunchUnits* out) {
    static const uint64_t group_entries = getenv("PLK_WIDE_GROUP_ENTRIES") ? strtoull(getenv("PLK_WIDE_GROUP_ENTRIES"), nullptr, 0) : WIDE_GROUP_ENTRIES;
    uint32_t narrow[MSM_SLOTS], nn = 0, first_narrow = count;
    uint64_t narrow_scalars = 0;
    for (uint32_t k = 0; k < count; ++k) if (plan[k] == MSM_C_SHORT) { if (!nn) first_narrow = k; narrow[nn++] = k; narrow_scalars += share[k]; }
    const uint32_t narrow_pos = narrow_scalars <= 4096 ? first_narrow : count;
    auto put_narrow = [&]() {
        if (!nn) return;
        out->plan[out->n] = MSM_C_SHORT; out->size[out->n] = nn;
        for (uint32_t m = 0; m < nn; ++m) out->member[out->n][m] = narrow[m];
        ++out->n;
    };
    out->n = 0;
    int open = -1;                      // the unit that still takes members (of its own plan)
    uint64_t open_entries = 0;
    for (uint32_t k = 0; k < count; ++k) {
        if (k == narrow_pos) put_narrow();   // an open unit goes on taking members: it was begun before the narrow group and stays in front of it
        if (plan[k] == MSM_C_SHORT) continue;
        const uint64_t e = share[k] * (uint64_t)msm_plan_windows(plan[k]);
        if (open < 0 || out->plan[open] != plan[k] || open_entries + e > group_entries) { open = (int)out->n++; out->plan[open] = plan[k]; out->size[open] = 0; open_entries = 0; }
        out->member[open][out->size[open]++] = k;
        open_entries += e;
    }
    if (narrow_pos == count) put_narrow();
}
static int32_t commit_begin_on(plk_ctx* ctx, int si, hipStream_t lane, const void* coeffs_dev, uint64_t len, const uint64_t* blind, int idle_hint) {
    MsmSlot& s = ctx->slot[si];
    PLK_TRY(sort_after_compute(ctx, lane));
    memcpy(s.blind_host, blind, 96);
    PLK_HIP(hipMemcpyAsync(s.blind_dev, s.blind_host, 96, hipMemcpyHostToDevice, lane));
    PLK_TRY(msm_enqueue(ctx, s, (const uint32_t*)coeffs_dev, len, 0, 3, lane, idle_hint));
    s.busy = true; s.kind = 1; s.in_batch = false;
    return PLK_OK;
}
extern "C" int32_t plk_kzg_commit_begin(plk_ctx* ctx, const void* coeffs_dev, uint64_t len, const uint64_t* blind, int32_t* ticket) {
    if (!ctx || !blind || !ticket || (len && !coeffs_dev)) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    int si;
    PLK_TRY(msm_acquire_slot(ctx, &si));
    PLK_TRY(commit_begin_on(ctx, si, front_lane(ctx), coeffs_dev, len, blind, -1));
    *ticket = si;
    return PLK_OK;
}
// The commitments of a round begun with one call (src/prover.rs:150-156: seven polynomials).  Same results as `count` calls of
// plk_kzg_commit_begin in the same order; knowing the whole round, the library chooses the streams of the members' sorts itself.
extern "C" int32_t plk_kzg_commit_begin_batch(plk_ctx* ctx, uint32_t count, const void* const* coeffs_dev, const uint64_t* len, const uint64_t* blind,
                                              int32_t* tickets) {
    if (!ctx || !count || !coeffs_dev || !len || !blind || !tickets) return PLK_ERR_INVALID_ARG;
    if (count > (uint32_t)MSM_SLOTS) { ctx->err = "plk_kzg_commit_begin_batch: more polynomials than MSM slots"; return PLK_ERR_INVALID_ARG; }
    for (uint32_t k = 0; k < count; ++k) if (len[k] && !coeffs_dev[k]) return PLK_ERR_INVALID_ARG;
    if (ctx->batch_open) { ctx->err = "plk_kzg_commit_begin_batch: tickets of the previous batch are still open"; return PLK_ERR_INVALID_ARG; }
    PLK_HIP(hipSetDevice(ctx->device));
    uint32_t free_slots = 0;
    for (const auto& s : ctx->slot) free_slots += s.busy ? 0 : 1;
    if (free_slots < count) { ctx->err = "all MSM slots are busy: end an open ticket first"; return PLK_ERR_INVALID_ARG; }
    // Members that are short on this rank share their launches (one sort, one accumulation, one reduction per group); the blinding
    // scalars of a group cross PCIe in one transfer.
    PLK_TRY(batch_buffers(ctx));
    bool idle = ctx->reduce_pending.empty();
    for (const auto& s : ctx->slot) idle = idle && !s.busy;
    int32_t rc = PLK_OK;
    int si[MSM_SLOTS];
    uint64_t share[MSM_SLOTS];
    int plan[MSM_SLOTS];
    for (uint32_t k = 0; k < count && rc == PLK_OK; ++k) {
        rc = msm_acquire_slot(ctx, &si[k]);
        if (rc != PLK_OK) break;
        tickets[k] = si[k];
        MsmSlot& s = ctx->slot[si[k]];
        s.busy = true; s.kind = 1; s.in_batch = true;   // its blinding scalars sit in the batch's staging area until its sort has run
        ++ctx->batch_open;
        share[k] = msm_share(ctx, len[k], 0);
        plan[k] = msm_plan_for(ctx, len[k], 0);
    }
    LaunchUnits units;
    if (rc == PLK_OK) plan_units(count, share, plan, &units);
    uint8_t lanes[MSM_SLOTS];
    plan_lanes(ctx, units.n, "PLK_LANES_COMMIT", lanes);
    uint32_t staged = 0;
    for (uint32_t u = 0; u < units.n && rc == PLK_OK; ++u) {
        MsmJob jobs[MSM_SLOTS];
        const uint32_t first_staged = staged;
        for (uint32_t m = 0; m < units.size[u]; ++m, ++staged) {
            const uint32_t k = units.member[u][m];
            memcpy(ctx->batch_host + BATCH_BLIND + 24 * staged, blind + 12 * k, 96);
            jobs[m].slot = si[k]; jobs[m].scalars = (const uint32_t*)coeffs_dev[k]; jobs[m].k = len[k]; jobs[m].base_offset = 0;
            jobs[m].n_blind = 3; jobs[m].blind_dev = ctx->batch_dev + BATCH_BLIND + 24 * staged;
        }
        hipStream_t lane = lanes[u] ? ctx->st_red : ctx->st_sort;
        rc = sort_after_compute(ctx, lane);
        if (rc == PLK_OK && hipMemcpyAsync(ctx->batch_dev + BATCH_BLIND + 24 * first_staged, ctx->batch_host + BATCH_BLIND + 24 * first_staged,
                                           (size_t)units.size[u] * 96, hipMemcpyHostToDevice, lane) != hipSuccess) rc = PLK_ERR_HIP;
        if (rc == PLK_OK) rc = msm_enqueue_group(ctx, jobs, units.size[u], lane, (idle && u == 0) ? 1 : 0, units.plan[u]);
    }
    if (rc != PLK_OK) { (void)plk_kzg_cancel(ctx); return rc; }   // the members begun so far are given up with everything else that is open
    return PLK_OK;
}
extern "C" int32_t plk_kzg_submit(plk_ctx* ctx) {
    if (!ctx) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    ctx->uploads_pending = false;
    return msm_flush_reductions(ctx);
}
extern "C" int32_t plk_kzg_ready(plk_ctx* ctx, int32_t ticket, int32_t* ready) {
    if (!ctx || !ready || ticket < 0 || ticket >= MSM_SLOTS || !ctx->slot[ticket].busy) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    PLK_TRY(msm_flush_reductions(ctx));
    const hipError_t e = hipEventQuery(ctx->slot[ctx->slot[ticket].done_owner].done);
    if (e != hipSuccess && e != hipErrorNotReady) { ctx->err = hipGetErrorString(e); return PLK_ERR_HIP; }
    *ready = e == hipSuccess ? 1 : 0;
    return PLK_OK;
}
extern "C" int32_t plk_kzg_cancel(plk_ctx* ctx) {
    if (!ctx) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    // whatever the open tickets enqueued runs to completion (their buffers stay valid), then every slot is handed back
    PLK_HIP(hipStreamSynchronize(ctx->st_sort));
    PLK_HIP(hipStreamSynchronize(ctx->st_acc[0]));
    PLK_HIP(hipStreamSynchronize(ctx->st_acc[1]));
    PLK_HIP(hipStreamSynchronize(ctx->st_red));
    ctx->reduce_pending.clear();
    for (auto& s : ctx->slot) slot_release(ctx, s);
    ctx->batch_open = 0;
    if (ctx->detached) { ctx->stream = ctx->st_sort; ctx->detached = false; }   // everything was waited for above
    ctx->quotient_stream = nullptr;
    return PLK_OK;
}
static void slot_release(plk_ctx* ctx, MsmSlot& s) {
    if (s.busy && s.in_batch && ctx->batch_open > 0) --ctx->batch_open;
    s.busy = false; s.in_batch = false;
}
static int32_t ticket_slot(plk_ctx* ctx, int32_t ticket, int kind, MsmSlot** out) {
    if (!ctx || ticket < 0 || ticket >= MSM_SLOTS || !ctx->slot[ticket].busy || ctx->slot[ticket].kind != kind) {
        if (ctx) ctx->err = "bad ticket";
        return PLK_ERR_INVALID_ARG;
    }
    *out = &ctx->slot[ticket];
    return PLK_OK;
}
extern "C" int32_t plk_kzg_commit_end(plk_ctx* ctx, int32_t ticket, uint64_t out_xy[8], int32_t* out_is_inf) {
    MsmSlot* s;
    PLK_TRY(ticket_slot(ctx, ticket, 1, &s));
    if (!out_xy || !out_is_inf) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    host::pt r;
    const int32_t rc = msm_wait(ctx, *s, &r);
    slot_release(ctx, *s);
    prof_collect(ctx, false);
    PLK_TRY(rc);
    host::pt_to_affine(r, out_xy, out_is_inf);
    return PLK_OK;
}
extern "C" int32_t plk_kzg_commit_dev(plk_ctx* ctx, const void* coeffs_dev, uint64_t len, const uint64_t* blind, uint64_t out_xy[8], int32_t* out_is_inf) {
    int32_t t;
    PLK_TRY(plk_kzg_commit_begin(ctx, coeffs_dev, len, blind, &t));
    return plk_kzg_commit_end(ctx, t, out_xy, out_is_inf);
}
extern "C" int32_t plk_kzg_commit(plk_ctx* ctx, const uint64_t* coeffs_host, uint64_t len, const uint64_t* blind, uint64_t out_xy[8], int32_t* out_is_inf) {
    if (!ctx || (len && !coeffs_host)) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    DevBuf d;
    if (len) PLK_HIP(d.alloc_copy(coeffs_host, len * 32));
    return plk_kzg_commit_dev(ctx, d.p, len, blind, out_xy, out_is_inf);
}

// ---- KZG open --------------------------------------------------------------------------------------
// hiding part of an opening, on the host: r' = challenge * r (3 coefficients), r'/(X - zeta) has 2 (-> hw, 16 words), random_v = r'(zeta)
constexpr uint64_t BATCH_DIVISION_BELOW = (1ull << 19) + 1;   // rounds whose longest MSM (on this rank) is shorter get their divisions as one batch
static void hiding_part(const uint64_t* blind, const uint64_t zeta[4], const uint64_t challenge[4], uint64_t random_v[4], uint32_t* hw) {
    using namespace host;
    const el z = el_load(zeta), ch = el_load(challenge);
    const el r0 = mul(FR, ch, el_load(blind)), r1 = mul(FR, ch, el_load(blind + 4)), r2 = mul(FR, ch, el_load(blind + 8));
    const el hw1 = r2, hw0 = add(FR, r1, mul(FR, z, r2));
    const el rv = add(FR, r0, mul(FR, z, hw0));
    memcpy(random_v, rv.v, 32);
    memcpy(hw, hw0.v, 32);
    memcpy(hw + 8, hw1.v, 32);
}
static int32_t open_begin_on(plk_ctx* ctx, int si, hipStream_t lane, const void* coeffs_dev, uint64_t len, const uint64_t* blind, const uint64_t zeta[4],
                             const uint64_t challenge[4], int idle_hint) {
    MsmSlot& s = ctx->slot[si];
    PLK_TRY(grow(ctx, ctx->st_sort, &s.scal_dev, &s.scal_cap, (len + 1) * 8));
    PLK_TRY(grow(ctx, ctx->st_sort, &s.chain_dev, &s.chain_cap, division_work_words(len)));
    PLK_TRY(sort_after_compute(ctx, lane));
    // witness polynomial of challenge*f: scalars challenge * h_i, i >= 1; h_0 = f(zeta)
    // a contiguous shard needs its own range of quotient coefficients only, a block-cyclic one has a part in every stretch
    const bool cyclic = ctx->cyc_world > 1;
    PLK_TRY(division_enqueue(ctx, lane, (const uint32_t*)coeffs_dev, len, zeta, challenge, s.scal_dev, s.res_dev, s.chain_dev,
                             cyclic ? 0 : ctx->shard_first, cyclic ? len : ctx->shard_len));
    PLK_HIP(hipMemcpyAsync(s.res_host, s.res_dev, 32, hipMemcpyDeviceToHost, lane));
    s.eval_host = s.res_host; s.in_batch = false;
    hiding_part(blind, zeta, challenge, s.host_rv, s.blind_host);
    PLK_HIP(hipMemcpyAsync(s.blind_dev, s.blind_host, 64, hipMemcpyHostToDevice, lane));
    PLK_TRY(msm_enqueue(ctx, s, s.scal_dev, len >= 2 ? len - 1 : 0, 0, 2, lane, idle_hint));
    s.busy = true; s.kind = 2;
    return PLK_OK;
}
extern "C" int32_t plk_kzg_open_begin(plk_ctx* ctx, const void* coeffs_dev, uint64_t len, const uint64_t* blind, const uint64_t zeta[4],
                                      const uint64_t challenge[4], int32_t* ticket) {
    if (!ctx || !blind || !zeta || !challenge || !ticket || (len && !coeffs_dev)) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    int si;
    PLK_TRY(msm_acquire_slot(ctx, &si));
    PLK_TRY(open_begin_on(ctx, si, front_lane(ctx), coeffs_dev, len, blind, zeta, challenge, -1));
    *ticket = si;
    return PLK_OK;
}
// All openings of a round at once (src/prover.rs:295-303: nine polynomials, one point, one challenge).  Short polynomials (all of
// them of at most 2^19 coefficients): the divisions run as one batch of launches at the head of the round instead of three dependent
// launches per polynomial threaded between the sorts and accumulations of the others (nine divisions of 2^16 coefficients are
// latency and nothing else); the sorts and accumulations then follow as for plk_kzg_open_begin, in the order given.  Long
// polynomials: each member's division stays in front of its own sort (batched, the divisions delay the round's first accumulation
// kernel, and without divisions between them the sorts of the two front streams bunch up beside the accumulation kernels: measured
// 0.3-0.7 ms slower at 2^20 gates).
extern "C" int32_t plk_kzg_open_begin_batch(plk_ctx* ctx, uint32_t count, const void* const* coeffs_dev, const uint64_t* len, const uint64_t* blind,
                                            const uint64_t zeta[4], const uint64_t challenge[4], int32_t* tickets) {
    if (!ctx || !count || !coeffs_dev || !len || !blind || !zeta || !challenge || !tickets) return PLK_ERR_INVALID_ARG;
    if (count > (uint32_t)MSM_SLOTS) { ctx->err = "plk_kzg_open_begin_batch: more polynomials than MSM slots"; return PLK_ERR_INVALID_ARG; }
    for (uint32_t k = 0; k < count; ++k) if (len[k] && !coeffs_dev[k]) return PLK_ERR_INVALID_ARG;
    if (ctx->batch_open) { ctx->err = "plk_kzg_open_begin_batch: tickets of the previous batch are still open"; return PLK_ERR_INVALID_ARG; }
    PLK_HIP(hipSetDevice(ctx->device));
    PLK_TRY(batch_buffers(ctx));
    uint32_t free_slots = 0;
    bool idle = ctx->reduce_pending.empty();
    for (const auto& s : ctx->slot) { free_slots += s.busy ? 0 : 1; idle = idle && !s.busy; }
    if (free_slots < count) { ctx->err = "all MSM slots are busy: end an open ticket first"; return PLK_ERR_INVALID_ARG; }
    // what decides is the length of this rank's MSMs: a rank of a sharded proof divides whole polynomials and accumulates its share
    uint64_t longest = 0;
    for (uint32_t k = 0; k < count; ++k) longest = std::max(longest, msm_share(ctx, len[k] >= 2 ? len[k] - 1 : 0, 0));
    static const uint64_t batch_below = getenv("PLK_BATCH_DIV_BELOW") ? strtoull(getenv("PLK_BATCH_DIV_BELOW"), nullptr, 0) : BATCH_DIVISION_BELOW;
    if (longest >= batch_below) {
        uint8_t lanes[MSM_SLOTS];
        plan_lanes(ctx, count, "PLK_LANES_OPEN", lanes);
        for (uint32_t k = 0; k < count; ++k) {
            int si;
            int32_t rc = msm_acquire_slot(ctx, &si);
            if (rc == PLK_OK) rc = open_begin_on(ctx, si, lanes[k] ? ctx->st_red : ctx->st_sort, coeffs_dev[k], len[k], blind + 12 * k, zeta, challenge, -1);
            if (rc != PLK_OK) { (void)plk_kzg_cancel(ctx); return rc; }
            tickets[k] = si;
        }
        return PLK_OK;
    }
    uint32_t* hw_dev = ctx->batch_dev + BATCH_HW;
    uint32_t* res_dev = ctx->batch_dev + BATCH_RES;
    uint32_t* hw_host = ctx->batch_host + BATCH_HW;
    uint32_t* res_host = ctx->batch_host + BATCH_RES;
    const bool cyclic = ctx->cyc_world > 1;
    int si[MSM_SLOTS];
    DivJob jobs[MSM_SLOTS];
    int32_t rc = PLK_OK;
    uint32_t taken = 0;
    for (uint32_t k = 0; k < count && rc == PLK_OK; ++k) {
        rc = msm_acquire_slot(ctx, &si[k]);
        if (rc != PLK_OK) break;
        MsmSlot& s = ctx->slot[si[k]];
        s.busy = true; s.kind = 2; s.in_batch = true;
        ++taken; ++ctx->batch_open;
        rc = grow(ctx, ctx->st_sort, &s.scal_dev, &s.scal_cap, (len[k] + 1) * 8);
        if (rc == PLK_OK) rc = grow(ctx, ctx->st_sort, &s.chain_dev, &s.chain_cap, division_work_words(len[k]));
        jobs[k].coeffs = (const uint32_t*)coeffs_dev[k]; jobs[k].len = len[k];
        jobs[k].q_out = s.scal_dev; jobs[k].h0_out = res_dev + 8 * k; jobs[k].work = s.chain_dev;
        jobs[k].q_first = cyclic ? 0 : ctx->shard_first; jobs[k].q_count = cyclic ? len[k] : ctx->shard_len;
        hiding_part(blind + 12 * k, zeta, challenge, s.host_rv, hw_host + 16 * k);
        s.eval_host = res_host + 8 * k;
    }
    // All divisions as one batch on the first front stream (polynomials this short: latency and nothing else), then the sorts,
    // in groups that share their launches (plan_units), on alternating front streams.
    hipStream_t st0 = ctx->st_sort;
    if (rc == PLK_OK && !ctx->detached) rc = order_streams(ctx, ctx->stream, st0);
    if (rc == PLK_OK && hipMemcpyAsync(hw_dev, hw_host, (size_t)count * 64, hipMemcpyHostToDevice, st0) != hipSuccess) rc = PLK_ERR_HIP;
    if (rc == PLK_OK) rc = division_enqueue_batch(ctx, st0, count, jobs, zeta, challenge);
    if (rc == PLK_OK && hipMemcpyAsync(res_host, res_dev, (size_t)count * 32, hipMemcpyDeviceToHost, st0) != hipSuccess) rc = PLK_ERR_HIP;
    uint64_t share[MSM_SLOTS];
    int plan[MSM_SLOTS];
    for (uint32_t k = 0; k < count && rc == PLK_OK; ++k) {
        const uint64_t scalars = len[k] >= 2 ? len[k] - 1 : 0;
        tickets[k] = si[k];
        share[k] = msm_share(ctx, scalars, 0);
        plan[k] = msm_plan_for(ctx, scalars, 0);
    }
    LaunchUnits units;
    if (rc == PLK_OK) plan_units(count, share, plan, &units);
    for (uint32_t u = 0; u < units.n && rc == PLK_OK; ++u) {
        MsmJob jobs[MSM_SLOTS];
        for (uint32_t m = 0; m < units.size[u]; ++m) {
            const uint32_t k = units.member[u][m];
            jobs[m].slot = si[k]; jobs[m].scalars = ctx->slot[si[k]].scal_dev; jobs[m].k = len[k] >= 2 ? len[k] - 1 : 0; jobs[m].base_offset = 0;
            jobs[m].n_blind = 2; jobs[m].blind_dev = hw_dev + 16 * k;
        }
        hipStream_t lane = front_lane(ctx);
        rc = sort_after_compute(ctx, lane);   // the second front stream waits for the divisions (and the sorts before this one)
        if (rc == PLK_OK) rc = msm_enqueue_group(ctx, jobs, units.size[u], lane, (idle && u == 0) ? 1 : 0, units.plan[u]);
    }
    if (rc != PLK_OK) {   // what was enqueued runs to completion on buffers that stay valid; the slots are handed back
        (void)hipStreamSynchronize(ctx->st_sort); (void)hipStreamSynchronize(ctx->st_red);
        (void)hipStreamSynchronize(ctx->st_acc[0]); (void)hipStreamSynchronize(ctx->st_acc[1]);
        for (uint32_t k = 0; k < taken; ++k) {
            auto& rp = ctx->reduce_pending;
            rp.erase(std::remove(rp.begin(), rp.end(), si[k]), rp.end());
            slot_release(ctx, ctx->slot[si[k]]);
        }
    }
    return rc;
}
extern "C" int32_t plk_kzg_open_end(plk_ctx* ctx, int32_t ticket, uint64_t w_xy[8], int32_t* w_is_inf, uint64_t random_v[4], uint64_t eval[4]) {
    MsmSlot* s;
    PLK_TRY(ticket_slot(ctx, ticket, 2, &s));
    if (!w_xy || !w_is_inf || !random_v || !eval) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    host::pt r;
    const int32_t rc = msm_wait(ctx, *s, &r);
    slot_release(ctx, *s);
    prof_collect(ctx, false);
    PLK_TRY(rc);
    host::pt_to_affine(r, w_xy, w_is_inf);
    memcpy(random_v, s->host_rv, 32);
    memcpy(eval, s->eval_host, 32);
    return PLK_OK;
}
extern "C" int32_t plk_kzg_end_partial(plk_ctx* ctx, int32_t ticket, uint64_t out_xyzz[16], uint64_t random_v[4], uint64_t eval[4]) {
    if (!ctx || ticket < 0 || ticket >= MSM_SLOTS || !ctx->slot[ticket].busy || !out_xyzz) return PLK_ERR_INVALID_ARG;
    MsmSlot* s = &ctx->slot[ticket];
    PLK_HIP(hipSetDevice(ctx->device));
    host::pt r;
    const int32_t rc = msm_wait(ctx, *s, &r);
    slot_release(ctx, *s);
    prof_collect(ctx, false);
    PLK_TRY(rc);
    host::pt_store(r, out_xyzz);
    if (s->kind == 2) {
        if (random_v) memcpy(random_v, s->host_rv, 32);
        if (eval) memcpy(eval, s->eval_host, 32);
    }
    return PLK_OK;
}
// ---- ending several tickets at once ------------------------------------------------------------------------------------
// Every MSM ends on the host with about 140 point additions and, for an affine result, one inversion: 60-100 microseconds of one
// core.  A round ends with several results arriving together (the short MSMs of a small proof share their reduction launch; a
// rank of a sharded proof has nine short MSMs), and their tails, done one after the other, were a tenth of such a proof.  A few
// helper threads share them: the tail is cut into five independent pieces per MSM (the sum of all buckets and one weighted sum per
// bit-field), then one combination and inversion per MSM.
namespace {
struct WorkPool {
    static constexpr int HELPERS = 3;
    std::vector<std::thread> helpers;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    const std::function<void(int)>* job = nullptr;
    int next = 0, total = 0, running = 0;
    uint64_t generation = 0;
    bool stop = false;
    void helper() {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lock(m);
            cv_work.wait(lock, [&] { return stop || (generation != seen && next < total); });
            if (stop) return;
            seen = generation;
            ++running;
            while (next < total) {
                const int i = next++;
                const std::function<void(int)>* f = job;
                lock.unlock();
                (*f)(i);
                lock.lock();
            }
            if (--running == 0) cv_done.notify_all();
        }
    }
    // f(0) .. f(n - 1) on the helpers and the calling thread; returns when all have run
    void run(int n, const std::function<void(int)>& f) {
        if (n <= 1 || helpers.empty()) { for (int i = 0; i < n; ++i) f(i); return; }
        {
            std::lock_guard<std::mutex> lock(m);
            job = &f; next = 0; total = n; ++generation;
        }
        cv_work.notify_all();
        std::unique_lock<std::mutex> lock(m);
        ++running;
        while (next < total) {
            const int i = next++;
            lock.unlock();
            f(i);
            lock.lock();
        }
        --running;
        cv_done.wait(lock, [&] { return running == 0; });
        total = 0; job = nullptr;
    }
    WorkPool() {
        const char* e = getenv("PLK_HOST_HELPERS");          // experiment switch
        const int n = e && *e ? std::max(0, std::min(31, atoi(e))) : HELPERS;
        for (int i = 0; i < n; ++i) helpers.emplace_back([this] { helper(); });
    }
    ~WorkPool() {
        { std::lock_guard<std::mutex> lock(m); stop = true; }
        cv_work.notify_all();
        for (auto& t : helpers) t.join();
    }
};
WorkPool& work_pool() { static WorkPool pool; return pool; }
std::mutex g_pool_use;   // one caller at a time (contexts of different host threads share the pool)
}  // namespace

// Ends `count` tickets: out receives 16 limbs per ticket, the affine point x || y in the first eight when affine != 0 (is_inf[k]
// set), the unnormalised partial sum X, Y, ZZ, ZZZ otherwise.  random_v / eval (4 limbs per ticket) are filled for openings and may
// be null.  Tickets whose results have arrived are finished while the device still works on the others.
extern "C" int32_t plk_kzg_end_many(plk_ctx* ctx, uint32_t count, const int32_t* tickets, int32_t affine, uint64_t* out, int32_t* is_inf, uint64_t* random_v,
                                    uint64_t* eval) {
    if (!ctx || !count || !tickets || !out || (affine && !is_inf) || count > (uint32_t)MSM_SLOTS) return PLK_ERR_INVALID_ARG;
    for (uint32_t k = 0; k < count; ++k) {
        if (tickets[k] < 0 || tickets[k] >= MSM_SLOTS || !ctx->slot[tickets[k]].busy) { ctx->err = "bad ticket"; return PLK_ERR_INVALID_ARG; }
        for (uint32_t j = 0; j < k; ++j) if (tickets[j] == tickets[k]) { ctx->err = "ticket listed twice"; return PLK_ERR_INVALID_ARG; }
    }
    PLK_HIP(hipSetDevice(ctx->device));
    PLK_TRY(msm_flush_reductions(ctx));
    std::vector<uint32_t> pending(count);
    for (uint32_t k = 0; k < count; ++k) pending[k] = k;
    int32_t rc = PLK_OK;
    while (!pending.empty() && rc == PLK_OK) {
        // everything that has arrived; when nothing has, the ticket listed first is waited for
        std::vector<uint32_t> ready, rest;
        for (uint32_t k : pending) {
            const MsmSlot& s = ctx->slot[tickets[k]];
            const hipError_t e = hipEventQuery(ctx->slot[s.done_owner].done);
            if (e == hipSuccess) ready.push_back(k);
            else if (e == hipErrorNotReady) rest.push_back(k);
            else { ctx->err = hipGetErrorString(e); rc = PLK_ERR_HIP; break; }
        }
        if (rc != PLK_OK) break;
        if (ready.empty()) {
            const MsmSlot& s = ctx->slot[tickets[pending[0]]];
            if (hipEventSynchronize(ctx->slot[s.done_owner].done) != hipSuccess) { ctx->err = "waiting for a result failed"; rc = PLK_ERR_HIP; break; }
            continue;
        }
        const int n = (int)ready.size();
        std::vector<host::pt> pieces((size_t)n * MSM_TAIL_PIECES);
        const std::function<void(int)> piece = [&](int i) {
            const MsmSlot& s = ctx->slot[tickets[ready[i / MSM_TAIL_PIECES]]];
            msm_tail_piece(s.marg_host, s.plan_c, i % MSM_TAIL_PIECES, &pieces[i]);
        };
        const std::function<void(int)> finish = [&](int i) {
            const uint32_t k = ready[i];
            const MsmSlot& s = ctx->slot[tickets[k]];
            host::pt r;
            msm_tail_combine(&pieces[(size_t)i * MSM_TAIL_PIECES], s.plan_c, &r);
            if (affine) { host::pt_to_affine(r, out + 16 * k, &is_inf[k]); memset(out + 16 * k + 8, 0, 64); }
            else host::pt_store(r, out + 16 * k);
        };
        {
            std::lock_guard<std::mutex> lock(g_pool_use);
            work_pool().run(n * MSM_TAIL_PIECES, piece);
            work_pool().run(n, finish);
        }
        for (uint32_t k : ready) {
            MsmSlot& s = ctx->slot[tickets[k]];
            if (s.kind == 2) {
                if (random_v) memcpy(random_v + 4 * k, s.host_rv, 32);
                if (eval) memcpy(eval + 4 * k, s.eval_host, 32);
            }
            slot_release(ctx, s);
        }
        pending.swap(rest);
    }
    if (rc != PLK_OK) for (uint32_t k : pending) slot_release(ctx, ctx->slot[tickets[k]]);
    prof_collect(ctx, false);
    return rc;
}
// plk_g1_sum_partials for `count` points at once: xyzz holds [world][count][16] limbs (what an all-gather of the ranks' partial sums
// returns), out_xy [count][8], out_is_inf [count].
extern "C" int32_t plk_g1_sum_partials_many(const uint64_t* xyzz, uint64_t world, uint64_t count, uint64_t* out_xy, int32_t* out_is_inf) {
    if (!xyzz || !out_xy || !out_is_inf || !world) return PLK_ERR_INVALID_ARG;
    const std::function<void(int)> one = [&](int i) {
        host::pt acc = host::pt_inf();
        for (uint64_t r = 0; r < world; ++r) acc = host::pt_add(acc, host::pt_load(xyzz + (r * count + (uint64_t)i) * 16));
        host::pt_to_affine(acc, out_xy + 8 * (uint64_t)i, &out_is_inf[i]);
    };
    std::lock_guard<std::mutex> lock(g_pool_use);
    work_pool().run((int)count, one);
    return PLK_OK;
}

extern "C" int32_t plk_kzg_open_dev(plk_ctx* ctx, const void* coeffs_dev, uint64_t len, const uint64_t* blind, const uint64_t zeta[4],
                                    const uint64_t challenge[4], uint64_t w_xy[8], int32_t* w_is_inf, uint64_t random_v[4], uint64_t eval[4]) {
    int32_t t;
    PLK_TRY(plk_kzg_open_begin(ctx, coeffs_dev, len, blind, zeta, challenge, &t));
    return plk_kzg_open_end(ctx, t, w_xy, w_is_inf, random_v, eval);
}
extern "C" int32_t plk_kzg_open(plk_ctx* ctx, const uint64_t* coeffs_host, uint64_t len, const uint64_t* blind, const uint64_t zeta[4],
                                const uint64_t challenge[4], uint64_t w_xy[8], int32_t* w_is_inf, uint64_t random_v[4], uint64_t eval[4]) {
    if (!ctx || (len && !coeffs_host)) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    DevBuf d;
    if (len) PLK_HIP(d.alloc_copy(coeffs_host, len * 32));
    return plk_kzg_open_dev(ctx, d.p, len, blind, zeta, challenge, w_xy, w_is_inf, random_v, eval);
}
