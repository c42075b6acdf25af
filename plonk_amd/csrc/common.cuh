// Shared declarations of libplonk_hip: the context object, error plumbing and small device helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>
#include "../../include/plonk_hip.h"
#include "g1.cuh"

namespace plk {

// Window layouts of the MSM.  Every window has its own table of 2^(C j) * P, so all windows of an MSM share one set of 2^(C-1)
// buckets (digit magnitude 1..2^(C-1) -> bucket 0..2^(C-1)-1).  Three widths, chosen by the number of scalars an MSM has on this
// rank: 17-bit windows (15 signed digits per scalar, 2^16 buckets) for long MSMs, where a sixteenth fewer additions outweigh the
// doubled reduction of the buckets; 16-bit windows (16 digits, 2^15 buckets) in between; 13-bit windows (20 digits, 2^12 buckets) for
// short ones, where the reduction, whose cost does not depend on the MSM's length, would otherwise be a fifth of the work and most
// of the latency.  The tables of the long plan cover all points of the rank, those of the other two a prefix (a short MSM of the
// prover starts at power 0).
constexpr int MSM_C_LONG = 17, MSM_C = 16, MSM_C_SHORT = 13;
template <int C>
struct MsmPlan {
    static constexpr int WINDOWS = (254 + C - 1) / C;                 // 15 | 16 | 20: the top window's digit is small, its carry never leaves the scalar
    static constexpr int BUCKETS = 1 << (C - 1);                      // 65536 | 32768 | 4096
    static constexpr int FIELDS = (C - 1 + 3) / 4;                    // bit-fields of the bucket index in the marginal sums: 4 | 4 | 3
    static constexpr int LAST_BITS = C - 1 - 4 * (FIELDS - 1);        // width of the top field: 4 | 3 | 4 (the others have 4 bits)
    static constexpr int MARGINALS = 16 * (FIELDS - 1) + (1 << LAST_BITS);   // 64 | 56 | 48 partial sums handed to the host
    static_assert(WINDOWS * C >= 255 && C >= 10 && C <= 17, "window layout");
};
constexpr int MSM_MAX_WINDOWS = 20;
constexpr int MSM_BUCKETS = MsmPlan<MSM_C_LONG>::BUCKETS;   // the largest of the bucket counts: sizes the per-slot scratch
constexpr int MSM_MARGINALS = 64;       // the largest of the plans' outputs
constexpr int MSM_SLOTS = 10;           // MSMs in flight (a round of the prover enqueues up to nine at once)
constexpr int XYZZ_WORDS = 36;          // unnormalised point in device scratch: 4 coordinates x 9 limbs

struct MsmSeg {
    const uint32_t* scalars;   // Fr elements, 8 words each, x*2^256 form: entry idx of the segment reads scalars[8 * scalar_index(idx)]
    uint32_t count;
    uint32_t table_off;        // index of the first base inside a window table; entry idx uses base table_off + idx
    // Block-cyclic shards (world > 1): the context holds the blocks of 2^log_block consecutive SRS points whose number is
    // congruent to rank modulo world, packed in order; `scalars` then points at the scalar of global point `origin`, and entry idx
    // belongs to global point ((t >> log_block) * world + rank) << log_block | (t & (2^log_block - 1)),  t = table_off + idx.
    uint32_t world, rank, log_block;
    uint64_t origin;
};
struct MsmSegs {
    MsmSeg s[2];
    uint32_t nseg;
    uint32_t total;
};

struct ProfEntry {
    double ms = 0;
    uint64_t launches = 0;
    std::vector<std::pair<double, double>> spans;   // start, end in ms since the last reset
};
struct PendingEvent {
    hipEvent_t e0, e1;
    std::string name;
};

struct MsmSlot {
    bool ready = false;
    hipEvent_t sorted = nullptr, accumulated = nullptr, done = nullptr;   // stage hand-offs between the three MSM streams
    uint32_t* counts = nullptr;        // [workgroups of the sort][partitions]
    uint32_t* totals = nullptr;        // [MSM_BUCKETS] (the sort keeps its partition starts here)
    uint32_t* bucket_start = nullptr;  // [MSM_BUCKETS + 1]
    uint32_t* entries = nullptr;       // sorted (table index | sign << 31)
    uint64_t entries_cap = 0;
    uint32_t* bucket_sums = nullptr;   // [MSM_BUCKETS][XYZZ_WORDS]
    uint32_t* flushed = nullptr;       // [MSM_BUCKETS][XYZZ_WORDS] part of a bucket summed by the lane whose run crossed its end
    uint32_t* tails = nullptr;         // [XYZZ_WORDS][tail_cap] what each accumulation lane held at the end of its run
    uint64_t tail_cap = 0;
    uint32_t* marg_dev = nullptr;      // [MSM_MARGINALS][32]
    uint32_t* marg_part = nullptr;     // [32768][XYZZ_WORDS] partial marginal sums (every lane of k_msm_marginals_part stores one)
    uint32_t* marg_host = nullptr;     // pinned
    uint32_t* blind_dev = nullptr;     // 3 Fr (+ room for opening scalars)
    uint32_t* blind_host = nullptr;    // pinned staging
    uint32_t* scal_dev = nullptr;      // opening: quotient scalars
    uint64_t scal_cap = 0;
    uint32_t* chain_dev = nullptr;     // opening: per-chunk carries of the division scan
    uint64_t chain_cap = 0;
    uint32_t* res_dev = nullptr;       // opening: remainder (f(zeta) * challenge) and friends
    uint32_t* res_host = nullptr;      // pinned
    bool busy = false;
    int kind = 0;                      // 0 msm, 1 commit, 2 open
    uint64_t max_entries = 0;          // of the job in flight: its accumulation time is proportional to this
    int acc_stream = 0;                // which of the two accumulation streams runs it
    int plan_c = MSM_C_LONG;           // window width of the job in flight
    int ev_owner = 0;                  // slot whose `accumulated` event covers this job (members of a batch share their launches)
    int done_owner = 0;                // slot whose `done` event covers this job's read-back
    uint32_t acc_blocks = 0;
    uint64_t host_rv[4], host_chal[4]; // opening: values finished on the host
    const uint32_t* eval_host = nullptr;   // opening: where the remainder f(zeta) arrives (res_host, or the batch's pinned buffer)
    bool in_batch = false;             // begun by plk_kzg_open_begin_batch
};

}  // namespace plk

struct plk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;      // NTT / pointwise work
    // MSMs are software pipelined by stage: every MSM's sort runs on st_sort, its bucket accumulation on one of the two
    // st_acc streams (alternating, so the next kernel's waves fill the chip while the previous one drains) and its
    // reduction + read-back on st_red: the accumulation kernels (most of a proof) run back to back while the
    // latency-bound stages of their neighbours fill in beside them.
    hipStream_t st_sort = nullptr, st_acc[2] = {nullptr, nullptr}, st_red = nullptr;
    unsigned lane_turn = 0;            // front_lane(): whose turn it is
    // Events that order one stream behind another (order_streams): a ring, created on first use and re-recorded in turn.  A wait
    // refers to the record that was current when it was enqueued, so a slot may be recorded again while earlier waits are pending.
    static constexpr int ORDER_EVENTS = 32;
    hipEvent_t order_ev[ORDER_EVENTS] = {};
    unsigned order_turn = 0;
    // A *_begin whose front work went to the second front stream (st_red) reads its polynomial there.  lane_read marks the end of
    // the last such read (the `sorted` event of that job's slot, not owned); compute work on the first front stream that may overwrite
    // the polynomial waits for it (compute_guard).
    hipEvent_t lane_read = nullptr;
    bool lane_read_pending = false;
    int batch_open = 0;                // tickets of the last plk_kzg_open_begin_batch that have not ended yet
    uint32_t* batch_dev = nullptr;     // per member of a batch of openings: 16 words of hiding scalars, 8 words of remainder
    uint32_t* batch_host = nullptr;    // pinned, same layout
    bool uploads_pending = false;      // plk_upload_enqueue since the last plk_kzg_submit: transfers queue on st_red, sorts keep off it
    bool detached = false;             // plk_compute_detach: `stream` is st_red for the time being, so compute work runs beside the sorts on st_sort
    uint64_t acc_load[2] = {0, 0};     // entries queued on each accumulation stream since the device was last idle
    plk::MsmSlot slot[plk::MSM_SLOTS];
    int next_slot = 0;
    std::vector<int> reduce_pending;   // slots whose sort and accumulation are enqueued but whose reduction is not yet (see msm_flush_reductions)
    // SRS
    uint64_t srs_total = 0, shard_first = 0, shard_len = 0;
    uint32_t cyc_world = 1, cyc_rank = 0, cyc_log_block = 0;   // cyc_world > 1: block-cyclic shard (shard_first unused, shard_len = points held)
    bool has_gamma = false;
    uint64_t table_stride = 0;         // points per window table: shard_len + 3 (gamma powers at the end)
    uint32_t* tables = nullptr;        // [15 windows][table_stride] affine, 16 words each, x*2^261 form: 2^(17 j) P
    uint32_t* tables_mid = nullptr;    // [16 windows][mid_stride]: 2^(16 j) P over a prefix, for the MSMs of middling length
    uint64_t mid_max = 0;              // MSMs of at most this many scalars (on this rank) take the 16-bit plan (0: never)
    uint64_t mid_len = 0, mid_stride = 0;       // points those tables cover (a prefix of this rank's points), points per table (prefix + 3 gamma powers)
    uint32_t* tables_short = nullptr;  // [20 windows][short_stride]: 2^(13 j) P, for the MSMs short enough for the narrow plan
    uint64_t short_max = 0;            // MSMs of at most this many scalars take the narrow plan (0: never)
    uint64_t short_len = 0, short_stride = 0;   // as mid_len, mid_stride
    uint32_t* marg_all_dev = nullptr;  // [MSM_SLOTS][MSM_MARGINALS][32]: the slots' outputs side by side, so that a batch of reductions is read back with one copy
    uint32_t* marg_all_host = nullptr; // pinned
    // NTT
    std::map<int, uint32_t*> twiddles; // log_n -> w^i for i < n/2, x*2^261 form
    std::map<std::vector<uint64_t>, uint32_t*> coset_factors;   // per-element factor tables of the large transforms (x*2^261 form): (log_n, inverse, generator) -> g^i or n^-1 g^-i; (~0, log_n, inverse, pass, ...) -> inter-pass twiddles
    uint32_t* ntt_tmp[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // ping-pong buffers of multi-pass transforms: set 0 for the attached compute stream, set 1 for detached work (the two may run at once)
    uint64_t ntt_tmp_cap[2] = {0, 0};
    // scratch for pointwise / scan kernels
    uint32_t* scratch = nullptr;
    uint64_t scratch_cap = 0;
    hipStream_t quotient_stream = nullptr;   // stream of a quotient enqueued with plk_quotient_dev_enqueue and not yet finished
    uint32_t* flag_dev = nullptr;
    uint32_t* flag_host = nullptr;
    void* stage = nullptr;             // staging of ordinary (pageable) host arrays for plk_upload_enqueue: page-locked ring + copy threads, made on first use
    // RCCL communicator of the ranks that share one proof (plk_comm_init), with its staging buffers
    void* comm = nullptr;
    uint32_t comm_rank = 0, comm_world = 1;
    void *comm_send = nullptr, *comm_recv = nullptr, *comm_host = nullptr;
    // profiling
    int profiling = 0;                 // 0 off, 1 every kernel class, 2 "msm_accumulate" only (cheap enough to stay on while a run is timed)
    bool prof_open = false;            // the last prof_begin recorded an event and waits for its prof_end
    std::vector<hipEvent_t> prof_free; // events handed back by finished brackets
    hipEvent_t prof_base = nullptr;    // recorded at the last profile reset
    std::map<std::string, plk::ProfEntry> prof;
    std::vector<plk::PendingEvent> pending;
    std::string err;
};

namespace plk {

int32_t set_err(plk_ctx* ctx, int32_t code, const char* what, hipError_t e, int line);
#define PLK_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) return plk::set_err(ctx, PLK_ERR_HIP, #call, e_, __LINE__); \
    } while (0)
#define PLK_TRY(call)                      \
    do {                                   \
        int32_t rc_ = (call);              \
        if (rc_ != PLK_OK) return rc_;     \
    } while (0)

// Device allocation that is released when the scope ends, for the entry points that stage host buffers and may return early.
struct DevBuf {
    void* p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 32); }
    hipError_t alloc_copy(const void* host, size_t bytes) {
        const hipError_t e = alloc(bytes);
        if (e != hipSuccess || bytes == 0) return e;
        return hipMemcpy(p, host, bytes, hipMemcpyHostToDevice);
    }
};

// profiling brackets (no-ops unless enabled)
void prof_begin(plk_ctx* ctx, hipStream_t s, const char* name);
void prof_end(plk_ctx* ctx, hipStream_t s);
void prof_collect(plk_ctx* ctx, bool wait);

int32_t ensure_scratch(plk_ctx* ctx, uint64_t bytes);
// everything enqueued on `first` so far is ordered before whatever is enqueued on `then` from now on
int32_t order_streams(plk_ctx* ctx, hipStream_t first, hipStream_t then);
// records the end of a polynomial read on the second front stream / orders the compute stream behind the last such read
int32_t lane_read_mark(plk_ctx* ctx, hipStream_t lane, hipEvent_t recorded);
int32_t compute_guard(plk_ctx* ctx);

// Issue priority of the short kernels that run beside the accumulation kernel.
#ifndef PLK_SIDE_PRIO_LEVEL
#define PLK_SIDE_PRIO_LEVEL 3
#endif
#define PLK_SIDE_PRIO() __builtin_amdgcn_s_setprio(PLK_SIDE_PRIO_LEVEL)

// ---- device helpers ----------------------------------------------------------------------------
// f(integral_constant<0>) ... f(integral_constant<N-1>): a loop the compiler cannot leave rolled.  Loops whose body contains
// several inlined modular products are sometimes kept as loops despite "#pragma unroll", and then a register array indexed by
// the loop variable moves to scratch memory.
template <int I, int N, class F>
__device__ __forceinline__ void static_for_impl(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_impl<I + 1, N>(f);
    }
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl<0, N>(f); }

__device__ __forceinline__ void load8(const uint32_t* p, uint32_t* w) {
    const uint4 a = *reinterpret_cast<const uint4*>(p), b = *reinterpret_cast<const uint4*>(p + 4);
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}
__device__ __forceinline__ void store8(uint32_t* p, const uint32_t* w) {
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    *reinterpret_cast<uint4*>(p + 4) = make_uint4(w[4], w[5], w[6], w[7]);
}
// 32-byte element in memory -> limbs (value unchanged)
__device__ __forceinline__ fe ld_fe(const uint32_t* p) {
    uint32_t w[8];
    load8(p, w);
    return unpack(w);
}
// normalized, bound 2 -> canonical 32 bytes
template <class F>
__device__ __forceinline__ void st_fe(uint32_t* p, const fe& a) {
    uint32_t w[8];
    pack(reduce<F>(a), w);
    store8(p, w);
}
__device__ __forceinline__ xyzz ld_xyzz(const uint32_t* p) {
    xyzz r;
#pragma unroll
    for (int i = 0; i < 9; ++i) { r.x.v[i] = p[i]; r.y.v[i] = p[9 + i]; r.zz.v[i] = p[18 + i]; r.zzz.v[i] = p[27 + i]; }
    return r;
}
__device__ __forceinline__ void st_xyzz(uint32_t* p, const xyzz& r) {
#pragma unroll
    for (int i = 0; i < 9; ++i) { p[i] = r.x.v[i]; p[9 + i] = r.y.v[i]; p[18 + i] = r.zz.v[i]; p[27 + i] = r.zzz.v[i]; }
}
__device__ __forceinline__ xyzz shfl_down_xyzz(const xyzz& a, int delta) {
    xyzz r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        r.x.v[i] = __shfl_down(a.x.v[i], delta);
        r.y.v[i] = __shfl_down(a.y.v[i], delta);
        r.zz.v[i] = __shfl_down(a.zz.v[i], delta);
        r.zzz.v[i] = __shfl_down(a.zzz.v[i], delta);
    }
    return r;
}

// ---- point addition shared by four lanes ------------------------------------------------------------------------------------
// A lone wave issues one VALU instruction per four cycles, so a point addition (fourteen field products one after the other) costs
// it 6-7 microseconds, and the reductions that end a round are chains of about twenty of them on a chip that is otherwise idle.
// The products of an addition are not a chain, though: four of them open it, and the ones that follow come in twos and threes.
// xyzz_add_quad spreads them over the four lanes of a quad (lanes 4k .. 4k+3): every lane holds the same operands, computes one
// product per step (four steps instead of fourteen) and hands it to the other three through DPP quad permutes.  More instructions
// in total (about 1.5 times), a third of the latency: for kernels that leave most SIMDs empty.  Control flow around it must be
// uniform within a quad.
template <int CTRL>
__device__ __forceinline__ fe fe_quad_perm(const fe& x) {
    fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = (uint32_t)__builtin_amdgcn_update_dpp((int)x.v[i], (int)x.v[i], CTRL, 0xf, 0xf, false);
    return r;
}
template <int K>
__device__ __forceinline__ fe fe_quad_bcast(const fe& x) { return fe_quad_perm<K * 0x55>(x); }   // quad_perm(K, K, K, K)
// lane q of the quad takes x_q
__device__ __forceinline__ fe fe_sel4(uint32_t q, const fe& x0, const fe& x1, const fe& x2, const fe& x3) {
    const bool b0 = (q & 1) != 0, b1 = (q & 2) != 0;
    fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const uint32_t lo = b0 ? x1.v[i] : x0.v[i], hi = b0 ? x3.v[i] : x2.v[i];
        r.v[i] = b1 ? hi : lo;
    }
    return r;
}
// a + b for all four lanes of a quad, which hold the same a and b; q = lane & 3.  Same formulas as xyzz_add; Y3 comes out as the sum
// of two reduced products (bound 4, the bound a stored Y may have) instead of one merged reduction.
__device__ __forceinline__ xyzz xyzz_add_quad(const xyzz& a, const xyzz& b, uint32_t q) {
    if (xyzz_is_inf(a)) return b;
    if (xyzz_is_inf(b)) return a;
    const fe m1 = mul<FqP>(fe_sel4(q, a.x, b.x, a.y, b.y), fe_sel4(q, b.zz, a.zz, b.zzz, a.zzz));
    const fe u1 = fe_quad_bcast<0>(m1), u2 = fe_quad_bcast<1>(m1), s1 = fe_quad_bcast<2>(m1), s2 = fe_quad_bcast<3>(m1);
    fe p = sub2<FqP>(u2, u1); norm(p);            // bound 4
    fe r = sub2<FqP>(s2, s1); norm(r);            // bound 4
    if (is_zero_mod<FqP>(p)) {
        if (is_zero_mod<FqP>(r)) return xyzz_dbl(a);
        return xyzz_inf();
    }
    const fe m2 = mul<FqP>(fe_sel4(q, a.zz, a.zzz, p, r), fe_sel4(q, b.zz, b.zzz, p, r));
    const fe zzab = fe_quad_bcast<0>(m2), zzzab = fe_quad_bcast<1>(m2), pp = fe_quad_bcast<2>(m2), rr = fe_quad_bcast<3>(m2);
    const fe m3 = mul<FqP>(fe_sel4(q, p, u1, zzab, zzab), pp);
    const fe ppp = fe_quad_bcast<0>(m3), qv = fe_quad_bcast<1>(m3);
    xyzz o;
    o.zz = fe_quad_bcast<2>(m3);
#pragma unroll
    for (int i = 0; i < 9; ++i) o.x.v[i] = rr.v[i] + FqP::SUB6X3[i] - ppp.v[i] - 2 * qv.v[i];
    norm(o.x);                                    // bound 8
    const fe t = sub8<FqP>(qv, o.x);              // bound 10, limbs < 3 * 2^29
    const fe ns1 = sub2<FqP>(fe_zero(), s1);      // 2p - S1: bound 2, limbs < 2^30
    const fe m4 = mul<FqP>(fe_sel4(q, r, ns1, zzzab, zzzab), fe_sel4(q, t, ppp, ppp, ppp));
    o.y = add(fe_quad_bcast<0>(m4), fe_quad_bcast<1>(m4)); norm(o.y);   // r t - S1 ppp: bound 4
    o.zzz = fe_quad_bcast<2>(m4);
    return o;
}

}  // namespace plk
