// Pippenger MSM over BN254 G1 for gfx950, and the SRS tables it runs on.
//
// Layout.  The SRS is resident as tables T[j][i] = 2^(C j) * P_i, one per window j (affine, 64 B per point, coordinates in the
// 2^261 Montgomery form of fp29.cuh).  With one table per window all windows share a single set of 2^(C-1) buckets:
// sum_i s_i P_i = sum_k k * B_k,  B_k = sum of +-T[j][i] over all (i, j) whose signed C-bit digit has magnitude k.
// Three plans (MsmPlan in common.cuh): C = 17 over the whole SRS, C = 16 and C = 13 over prefixes of it for shorter MSMs.
//
// Pipeline of one MSM (sort on a front stream, accumulation and reduction on streams of their own, several MSMs in flight):
//   count          : signed digits of every scalar, histogram of 256 partitions per workgroup
//   partition      : the digits again (cheaper than keeping them: a product and some shifts against 32 bytes written and read per
//                    scalar); every workgroup appends its 8-byte records to its own piece of each partition (upper 8 bucket bits)
//   buckets        : one workgroup per partition spreads the records over the partition's buckets -> entries sorted by bucket
//   accumulate: every lane adds one run of consecutive sorted entries (mixed XYZZ additions); sums of runs that end inside a
//               bucket are joined by the fold kernel                              -> bucket_sums
//   marginals : sum_k k B_k is linear in the bits of k; the bucket index is cut into 4-bit fields and the sums "all buckets whose
//               field d equals v" come from serial partial sums and one tree each
//   host      : ~130 point additions and one inversion finish the job (a GPU lane needs ~6 us per point
//               addition, a host core ~0.3 us, so the serial tail belongs on the host).
// Every kernel takes a batch (blockIdx.y or a block table selects the member): the short MSMs of a round share their launches.
#include "common.cuh"
#include "host_math.hpp"
#include "kernels.h"
#include <stdlib.h>
#include <algorithm>

namespace plk {

// ------------------------------------------------------------------------------------------ scalars -> digits
// scalar in memory: x * 2^256, canonical.  Returns WINDOWS signed digits d_j in [-2^(C-1)+1, 2^(C-1)] with x = sum d_j 2^(C j).
template <int C>
__device__ __forceinline__ void scalar_digits(const uint32_t* sw, int32_t* d) {   // sw: the scalar's eight words, already loaded
    const fe s = reduce<FrP>(mul<FrP>(unpack(sw), fe_const(FrP::RAW32)));
    uint32_t w[8];
    pack(s, w);
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < MsmPlan<C>::WINDOWS; ++j) {
        const int bit = C * j, wi = bit >> 5, off = bit & 31;
        uint32_t raw = w[wi] >> off;
        if (off + C > 32 && wi + 1 < 8) raw |= w[wi + 1] << (32 - off);
        raw = (raw & ((1u << C) - 1u)) + carry;
        carry = raw > (1u << (C - 1)) ? 1u : 0u;
        d[j] = (int32_t)raw - (int32_t)(carry << C);
    }
}
__device__ __forceinline__ const uint32_t* seg_scalar(const MsmSegs& segs, uint32_t idx, uint32_t* table_index) {
    if (idx < segs.s[0].count) {
        const uint32_t t = segs.s[0].table_off + idx;
        *table_index = t;
        if (segs.s[0].world > 1) {
            const uint32_t lb = segs.s[0].log_block;
            const uint64_t global = ((((uint64_t)(t >> lb)) * segs.s[0].world + segs.s[0].rank) << lb) | (t & ((1u << lb) - 1));
            return segs.s[0].scalars + 8ull * (global - segs.s[0].origin);
        }
        return segs.s[0].scalars + 8ull * idx;
    }
    idx -= segs.s[0].count;
    *table_index = segs.s[1].table_off + idx;
    return segs.s[1].scalars + 8ull * idx;
}

constexpr int SORT_THREADS = 256;   // four waves per block: a block fits beside the accumulation kernel's waves on a busy CU

// ------------------------------------------------------------------------------------------ two-level sort
// A one-level scatter (every workgroup writes each entry straight to its bucket's slot, cursors per workgroup and bucket) hands
// every (chunk, bucket) pair its own few slots, so each 4-byte store lands in a line that no other store of the same workgroup
// completes soon: the line leaves the L2 with four valid bytes and memory sees one partial write per entry (measured: 7.5 times the
// entry bytes written, 2.35 ms per proof of 2^20 gates lost by the accumulation kernels beside it).  Here the entries first go to
// SORT_PARTS = 256 partitions by the upper eight bits of their bucket (8-byte records: entry word + the lower bucket bits), each
// workgroup appending to its own reserved piece of a partition, so the 256 lines it has open fill within an iteration or two; then
// one workgroup per partition spreads its records over the partition's buckets (256, 128 or 16 by window width), all inside a
// region that stays in the L2 until its lines are full.  Three launches; 1 KB of LDS per workgroup.
constexpr int SORT_PARTS = 256;
constexpr int SORT_MAX_BLOCKS = 512;                                            // workgroups of the first two kernels (rows of the count matrix)
template <int C>
struct SortPlan {
    static constexpr int PART_BITS = C - 1 - 8;                                 // bucket bits below the partition: 8 | 7 | 4
    static constexpr int PART_BUCKETS = 1 << PART_BITS;                         // buckets per partition: 256 | 128 | 16
    static_assert(PART_BUCKETS <= SORT_THREADS, "one counter per thread in k_sort_buckets");
};
// What the three kernels need to know about every member of a batch.
struct SortBatch {
    MsmSegs segs[MSM_SLOTS];
    uint32_t chunk[MSM_SLOTS];         // scalars per workgroup of the first two kernels
    uint32_t nblocks[MSM_SLOTS];       // workgroups of the member (the launch has the largest count; the others leave)
    uint32_t* counts[MSM_SLOTS];       // [workgroup][SORT_PARTS]
    uint32_t* part_start[MSM_SLOTS];   // SORT_PARTS + 1 words
    uint64_t* records[MSM_SLOTS];
    uint32_t* bucket_start[MSM_SLOTS]; // BUCKETS + 1 words
    uint32_t* entries[MSM_SLOTS];
    uint32_t table_stride;
};
// this workgroup's histogram of partitions -> counts[workgroup][SORT_PARTS]
template <int C>
__global__ void __launch_bounds__(SORT_THREADS) k_sort_count(SortBatch sb) {
    PLK_SIDE_PRIO();
    constexpr int W = MsmPlan<C>::WINDOWS;
    const uint32_t mb = blockIdx.y;
    if (blockIdx.x >= sb.nblocks[mb]) return;
    const MsmSegs& segs = sb.segs[mb];
    const uint32_t chunk = sb.chunk[mb];
    __shared__ uint32_t hist[SORT_PARTS];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t lo = blockIdx.x * chunk, hi = min(lo + chunk, segs.total);
    uint32_t sw_next[8], ti;
    if (lo + threadIdx.x < hi) load8(seg_scalar(segs, lo + threadIdx.x, &ti), sw_next);
    for (uint32_t idx = lo + threadIdx.x; idx < hi; idx += SORT_THREADS) {
        uint32_t sw[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) sw[i] = sw_next[i];
        if (idx + SORT_THREADS < hi) load8(seg_scalar(segs, idx + SORT_THREADS, &ti), sw_next);   // in flight while this scalar is converted
        int32_t d[W];
        scalar_digits<C>(sw, d);
#pragma unroll
        for (int j = 0; j < W; ++j)
            if (d[j] != 0) atomicAdd(&hist[(uint32_t)((d[j] < 0 ? -d[j] : d[j]) - 1) >> SortPlan<C>::PART_BITS], 1u);
    }
    __syncthreads();
    sb.counts[mb][(size_t)blockIdx.x * SORT_PARTS + threadIdx.x] = hist[threadIdx.x];
}
// records[slot] = entry word | (lower bucket bits << 32), appended to this workgroup's piece of the record's partition
// Every workgroup works out where its pieces begin from the whole count matrix (nblocks x 256 words, read from the L2): thread p sums
// partition p over all workgroups (the partition's size) and over the workgroups before its own, a scan over p gives the
// partition's start.  A kernel of its own for this took one more launch and, as a single workgroup beside the accumulation waves,
// as long as the partitioning itself.  Workgroup 0 also publishes the partition starts for k_sort_buckets.
template <int C>
__global__ void __launch_bounds__(SORT_THREADS) k_sort_partition(SortBatch sb) {
    PLK_SIDE_PRIO();
    constexpr int W = MsmPlan<C>::WINDOWS;
    const uint32_t mb = blockIdx.y, nblocks = sb.nblocks[mb];
    if (blockIdx.x >= nblocks) return;
    const MsmSegs& segs = sb.segs[mb];
    const uint32_t* __restrict__ part_counts = sb.counts[mb];
    uint32_t* __restrict__ part_start = sb.part_start[mb];
    uint64_t* __restrict__ records = sb.records[mb];
    const uint32_t chunk = sb.chunk[mb], table_stride = sb.table_stride;
    __shared__ uint32_t cursor[SORT_PARTS];
    __shared__ uint32_t wave_sum[SORT_THREADS / 64];
    {
        const uint32_t p = threadIdx.x, lane = p & 63, wave = p >> 6;
        uint32_t total = 0, before = 0;
        uint32_t k = 0;
        for (; k + 16 <= nblocks; k += 16) {   // sixteen loads in flight instead of one
            uint32_t c[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) c[u] = part_counts[(size_t)(k + u) * SORT_PARTS + p];
#pragma unroll
            for (int u = 0; u < 16; ++u) { total += c[u]; before += k + u < blockIdx.x ? c[u] : 0; }
        }
        for (; k < nblocks; ++k) {
            const uint32_t c = part_counts[(size_t)k * SORT_PARTS + p];
            total += c;
            before += k < blockIdx.x ? c : 0;
        }
        uint32_t incl = total;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t x = __shfl_up(incl, o); if (lane >= (uint32_t)o) incl += x; }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        uint32_t start = incl - total;
        for (uint32_t k2 = 0; k2 < wave; ++k2) start += wave_sum[k2];
        cursor[p] = start + before;
        if (blockIdx.x == 0) {
            part_start[p] = start;
            if (p == SORT_PARTS - 1) part_start[SORT_PARTS] = start + total;
        }
    }
    __syncthreads();
    const uint32_t lo = blockIdx.x * chunk, hi = min(lo + chunk, segs.total);
    uint32_t sw_next[8], ti_next = 0;
    if (lo + threadIdx.x < hi) load8(seg_scalar(segs, lo + threadIdx.x, &ti_next), sw_next);
    for (uint32_t idx = lo + threadIdx.x; idx < hi; idx += SORT_THREADS) {
        const uint32_t ti = ti_next;
        uint32_t sw[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) sw[i] = sw_next[i];
        if (idx + SORT_THREADS < hi) load8(seg_scalar(segs, idx + SORT_THREADS, &ti_next), sw_next);   // in flight while this scalar is placed
        int32_t d[W];
        scalar_digits<C>(sw, d);
#pragma unroll
        for (int j = 0; j < W; ++j) {
            if (d[j] == 0) continue;
            const uint32_t b = (uint32_t)((d[j] < 0 ? -d[j] : d[j]) - 1);
            const uint32_t slot = atomicAdd(&cursor[b >> SortPlan<C>::PART_BITS], 1u);
            const uint32_t entry = (d[j] < 0 ? 0x80000000u : 0u) | ((uint32_t)j * table_stride + ti);
            records[slot] = (uint64_t)entry | ((uint64_t)(b & (SortPlan<C>::PART_BUCKETS - 1)) << 32);
        }
    }
}
// one workgroup per partition: bucket sizes, bucket starts, then the entries in bucket order.  T threads: 256 beside accumulation
// kernels (a four-wave workgroup finds room on a busy CU), 1024 when the chip is idle (the first sort of a round: every CU has one
// partition to itself and nothing else to do, so sixteen waves go through its records four times as fast).
template <int T, int C>
__global__ void __launch_bounds__(T) k_sort_buckets(SortBatch sb) {
    PLK_SIDE_PRIO();
    constexpr int PB = SortPlan<C>::PART_BUCKETS, BUCKETS = MsmPlan<C>::BUCKETS;
    const uint32_t mb = blockIdx.y;
    const uint64_t* __restrict__ records = sb.records[mb];
    const uint32_t* __restrict__ part_start = sb.part_start[mb];
    uint32_t* __restrict__ bucket_start = sb.bucket_start[mb];
    uint32_t* __restrict__ entries = sb.entries[mb];
    __shared__ uint32_t cnt[PB];
    __shared__ uint32_t wave_sum[4];
    const uint32_t p = blockIdx.x, t = threadIdx.x;
    const uint32_t first = part_start[p], last = part_start[p + 1];
    if (t < PB) cnt[t] = 0;
    __syncthreads();
    // four records per thread and step, so that four loads are in flight instead of one
    for (uint32_t i = first + t; i < last; i += 4 * T) {
        uint64_t r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = i + k * T < last ? records[i + k * T] : ~0ull;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (r[k] != ~0ull) atomicAdd(&cnt[(uint32_t)(r[k] >> 32)], 1u);
    }
    __syncthreads();
    {                                   // exclusive scan of the counters (the first four waves hold them)
        const uint32_t lane = t & 63, wave = t >> 6, c = t < PB ? cnt[t] : 0;
        uint32_t incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t x = __shfl_up(incl, o); if (lane >= (uint32_t)o) incl += x; }
        if (lane == 63 && wave < 4) wave_sum[wave] = incl;
        __syncthreads();
        if (t < PB) {
            uint32_t start = first + incl - c;
            for (uint32_t k2 = 0; k2 < wave; ++k2) start += wave_sum[k2];
            bucket_start[p * PB + t] = start;
            cnt[t] = start;             // from here on: the next free slot of the bucket
        }
    }
    if (p == SORT_PARTS - 1 && t == 0) bucket_start[BUCKETS] = last;
    __syncthreads();
    for (uint32_t i = first + t; i < last; i += 4 * T) {
        uint64_t r[4];
        uint32_t at[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = i + k * T < last ? records[i + k * T] : ~0ull;
#pragma unroll
        for (int k = 0; k < 4; ++k) at[k] = r[k] != ~0ull ? atomicAdd(&cnt[(uint32_t)(r[k] >> 32)], 1u) : 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (r[k] != ~0ull) entries[at[k]] = (uint32_t)r[k];
    }
}

// ------------------------------------------------------------------------------------------ buckets
// The accumulation kernel keeps two waves of up to 176 registers on every SIMD (352 of 512).  Kernels of neighbouring MSMs
// that must run beside it are held to the remaining 160, otherwise their waves wait for an accumulation wave to retire and
// then occupy its place.
#define SIDE_KERNEL_VGPRS __attribute__((amdgpu_waves_per_eu(4, 4)))
constexpr int ACC_THREADS = 64;   // one wavefront per workgroup: waves retire and are replaced independently, no round quantisation
// Every lane adds one run of `run` consecutive sorted entries, whatever buckets they belong to, so all lanes do the same
// amount of work.  When a run crosses the end of bucket b the partial sum goes to flushed[b] (a bucket ends only once);
// the sum a lane holds at the end of its run goes to tails[lane] (structure of arrays, word i of lane l at i * tail_stride + l).
// k_msm_fold then rebuilds each bucket from its flushed part and the tails of the lanes whose run ended inside it.
// The loop is software pipelined: the table point of entry k+1 and the index of entry k+2 are in flight while entry k is added.
// The run length follows from the number of entries the sort actually produced (zero digits produce none, so a polynomial
// of zeros with three blinding terms has 48 entries whatever its declared length) and the lanes launched for the worst case.
__device__ __forceinline__ uint32_t acc_run_length(uint32_t n_entries, uint32_t lanes_launched) {
    const uint32_t run = (n_entries + lanes_launched - 1) / lanes_launched;
    return run < 1 ? 1 : run;
}
// Members of a batch of accumulations: the workgroups [block_first[m], block_first[m + 1]) of the launch belong to member m.
struct AccBatch {
    const uint32_t* entries[MSM_SLOTS];
    const uint32_t* bucket_start[MSM_SLOTS];
    uint32_t* tails[MSM_SLOTS];
    uint32_t* flushed[MSM_SLOTS];
    uint32_t tail_stride[MSM_SLOTS];
    uint32_t block_first[MSM_SLOTS + 1];
    uint32_t n;
};
// Two waves per SIMD: the loop needs about 170 registers, and more resident waves were measured to add nothing (the kernel
// is bound by VALU issue, not by latency).
template <int BUCKETS>
__global__ void __launch_bounds__(ACC_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_msm_accumulate(AccBatch ab, const uint32_t* __restrict__ tables) {
    uint32_t mb = 0;
    while (mb + 1 < ab.n && blockIdx.x >= ab.block_first[mb + 1]) ++mb;
    const uint32_t* __restrict__ entries = ab.entries[mb];
    const uint32_t* __restrict__ bucket_start = ab.bucket_start[mb];
    uint32_t* __restrict__ tails = ab.tails[mb];
    uint32_t* __restrict__ flushed = ab.flushed[mb];
    const uint32_t tail_stride = ab.tail_stride[mb];
    const uint32_t lane = (blockIdx.x - ab.block_first[mb]) * ACC_THREADS + threadIdx.x;
    const uint32_t n_entries = bucket_start[BUCKETS];
    const uint32_t run = acc_run_length(n_entries, (ab.block_first[mb + 1] - ab.block_first[mb]) * ACC_THREADS);
    uint32_t e = lane * run;
    if (e >= n_entries) return;
    const uint32_t e_end = min(e + run, n_entries);
    // bucket of the first entry: the largest b with bucket_start[b] <= e (empty buckets share their start with the next one)
    uint32_t b = 0;
    {
        uint32_t hi = BUCKETS;
        while (hi - b > 1) {
            const uint32_t mid = (b + hi) >> 1;
            if (bucket_start[mid] <= e) b = mid; else hi = mid;
        }
    }
    uint32_t next = bucket_start[b + 1];
    xyzz acc = xyzz_inf();
    uint32_t enc1 = e + 1 < e_end ? entries[e + 1] : 0;
    aff q;   // the point of entry e, unpacked and with its sign applied
    {
        const uint32_t enc0 = entries[e];
        uint32_t wx[8], wy[8];
        const uint32_t* p = tables + (size_t)(enc0 & 0x7fffffffu) * 16;
        load8(p, wx);
        load8(p + 8, wy);
        q.x = unpack(wx);
        q.y = unpack(wy);
        if (enc0 >> 31) q.y = fq_neg_y(q.y);
    }
    // Order inside an iteration: (1) the bucket change, whose loads are waited for at once, while nothing else is in flight;
    // (2) the loads of the next point; (3) the addition, which touches no loaded register; (4) the next point is unpacked, and only
    // here the wave waits for (2).  With the loaded words consumed at the top of the iteration the compiler put a wait for ALL
    // outstanding loads there (it cannot count the loads issued under a divergent condition), and the wave sat out the full latency
    // of the gather it had just issued, in every iteration.
    while (true) {
        if (e == next) {   // the run leaves bucket b here
            st_xyzz(flushed + (size_t)b * XYZZ_WORDS, acc);
            acc = xyzz_inf();
            do { ++b; next = bucket_start[b + 1]; } while (next <= e);
        }
        const bool more = e + 1 < e_end;
        uint32_t nx[8], ny[8], enc2 = 0;
        if (more) {
            const uint32_t* p = tables + (size_t)(enc1 & 0x7fffffffu) * 16;
            load8(p, nx);
            load8(p + 8, ny);
            if (e + 2 < e_end) enc2 = entries[e + 2];
        }
        xyzz_madd(acc, q);
        if (!more) break;
        q.x = unpack(nx);
        q.y = unpack(ny);
        if (enc1 >> 31) q.y = fq_neg_y(q.y);
        enc1 = enc2; ++e;
    }
    uint32_t* dst = tails + lane;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        dst[(size_t)i * tail_stride] = acc.x.v[i];
        dst[(size_t)(9 + i) * tail_stride] = acc.y.v[i];
        dst[(size_t)(18 + i) * tail_stride] = acc.zz.v[i];
        dst[(size_t)(27 + i) * tail_stride] = acc.zzz.v[i];
    }
}

__device__ __forceinline__ xyzz ld_tail(const uint32_t* tails, uint32_t tail_stride, uint32_t lane) {
    const uint32_t* src = tails + lane;
    xyzz r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        r.x.v[i] = src[(size_t)i * tail_stride];
        r.y.v[i] = src[(size_t)(9 + i) * tail_stride];
        r.zz.v[i] = src[(size_t)(18 + i) * tail_stride];
        r.zzz.v[i] = src[(size_t)(27 + i) * tail_stride];
    }
    return r;
}
// The reductions of several MSMs that finish at about the same time run as one launch each (blockIdx.y = member): a reduction
// costs about the same whatever the MSM's length, so for short MSMs (the per-rank shares of a sharded proof) one launch per MSM
// would make the reduction stream the longest chain of a round.
struct ReduceBatch {
    int32_t n;
    const uint32_t* bucket_start[MSM_SLOTS];
    const uint32_t* tails[MSM_SLOTS];
    const uint32_t* flushed[MSM_SLOTS];
    uint32_t tail_stride[MSM_SLOTS];
    uint32_t lanes_launched[MSM_SLOTS];
    uint32_t* bucket_sums[MSM_SLOTS];
    uint32_t* marg_part[MSM_SLOTS];
    uint32_t* marg_dev[MSM_SLOTS];
};
// bucket_sums[b] = flushed[b] (if the bucket ended inside a run) + the tails of all lanes whose run ended inside bucket b.
// FOLD_LANES lanes per bucket share the tails, a shuffle tree joins them.
template <int FOLD_LANES, int C, int G = 1>
__device__ __forceinline__ void fold_body(const ReduceBatch& rb);
// a + b by one lane (G = 1) or by the four lanes of a quad together (G = 4, xyzz_add_quad)
template <int G>
__device__ __forceinline__ xyzz add_g(const xyzz& a, const xyzz& b, uint32_t q) {
    if constexpr (G == 4) return xyzz_add_quad(a, b, q);
    else return xyzz_add(a, b);
}
// Two builds of every reduction kernel.  "side": held to 128 registers so that its waves fit beside two accumulation waves per SIMD
// (they spill, which costs nothing that matters: these reductions run beside the accumulation kernels of later MSMs).
// "tail": no register limit, for the reduction that ends a batch: nothing else is running then, and the round waits for it.
template <int FOLD_LANES, int C>
__global__ void __launch_bounds__(256) SIDE_KERNEL_VGPRS k_msm_fold(ReduceBatch rb) {
    PLK_SIDE_PRIO();
    fold_body<FOLD_LANES, C>(rb);
}
template <int FOLD_LANES, int C>
__global__ void __launch_bounds__(256) k_msm_fold_tail(ReduceBatch rb) { fold_body<FOLD_LANES, C>(rb); }
// "quad": a tail build in which four lanes share every addition (a third of the latency, one and a half times the instructions), for
// the launches that leave most of the chip empty
template <int FOLD_LANES, int C>
__global__ void __launch_bounds__(256) k_msm_fold_quad(ReduceBatch rb) { fold_body<FOLD_LANES, C, 4>(rb); }
template <int FOLD_LANES, int C, int G>
__device__ __forceinline__ void fold_body(const ReduceBatch& rb) {
    constexpr int BUCKETS = MsmPlan<C>::BUCKETS;
    const uint32_t* __restrict__ bucket_start = rb.bucket_start[blockIdx.y];
    const uint32_t* __restrict__ tails = rb.tails[blockIdx.y];
    const uint32_t* __restrict__ flushed = rb.flushed[blockIdx.y];
    const uint32_t tail_stride = rb.tail_stride[blockIdx.y], lanes_launched = rb.lanes_launched[blockIdx.y];
    uint32_t* __restrict__ bucket_sums = rb.bucket_sums[blockIdx.y];
    const uint32_t t = (blockIdx.x * 256 + threadIdx.x) / G, q = threadIdx.x & 3;   // G lanes share the work of one
    const uint32_t b = t / FOLD_LANES, j = t % FOLD_LANES;
    const uint32_t first = bucket_start[b], last = bucket_start[b + 1], n_entries = bucket_start[BUCKETS];
    const uint32_t run = acc_run_length(n_entries, lanes_launched);
    xyzz acc = xyzz_inf();
    if (last > first) {
        // lane l ends its run at entry (l + 1) * run - 1 (the final lane at n_entries - 1): it belongs here when that entry is in [first, last)
        const uint32_t m_lo = first / run + 1, m_hi = last / run;
        for (uint32_t m = m_lo + j; m <= m_hi; m += FOLD_LANES) acc = add_g<G>(acc, ld_tail(tails, tail_stride, m - 1), q);
        if (j == FOLD_LANES - 1 && last % run != 0) {
            if (last < n_entries) acc = add_g<G>(acc, ld_xyzz(flushed + (size_t)b * XYZZ_WORDS), q);
            else acc = add_g<G>(acc, ld_tail(tails, tail_stride, m_hi), q);   // the final, shorter run
        }
    }
    // only the lanes whose partner belongs to the same bucket add: a lane past the end of the wave would receive its own value back,
    // take the doubling branch, and the whole wave would execute that branch beside the addition at every level
    for (int s = FOLD_LANES / 2; s > 0; s >>= 1) {
        const xyzz other = shfl_down_xyzz(acc, s * G);
        if (j < (uint32_t)s) acc = add_g<G>(acc, other, q);
    }
    if (j == 0 && (G == 1 || q == 0)) st_xyzz(bucket_sums + (size_t)b * XYZZ_WORDS, acc);
}

// Marginal sums: output o = the sum of all buckets whose index has field d (bits [4d, 4d+4); the top field may be narrower) equal
// to v.  Two kernels.  In the first every lane adds MARG_K buckets of one output one after the other and stores its sum: no lane
// idles in a shuffle tree, so a batch of nine MSMs costs the chip little more than one does.  The second joins the partial sums of
// an output: a few per lane, then one shuffle tree.
constexpr int MARG_K = 8;
template <int C>
struct MargPlan {
    using P = MsmPlan<C>;
    static constexpr int NA = 16 * (P::FIELDS - 1), NB = 1 << P::LAST_BITS;                         // outputs of the 4-bit fields, of the top field
    static constexpr int PART_A = (P::BUCKETS >> 4) / MARG_K, PART_B = (P::BUCKETS >> P::LAST_BITS) / MARG_K;   // partial sums per output
    static constexpr int PARTIALS = NA * PART_A + NB * PART_B;                                       // 32768 | 16384 | 1536 per MSM
    static_assert(PARTIALS % 64 == 0, "whole waves");
};
constexpr int MARG_PARTIALS_MAX = MargPlan<MSM_C_LONG>::PARTIALS;
template <int C, int G = 1> __device__ __forceinline__ void marginals_part_body(const ReduceBatch& rb);
template <int C> __device__ __forceinline__ void marginals_join_body(const ReduceBatch& rb);
template <int C> __device__ __forceinline__ void marginals_join_quad_body(const ReduceBatch& rb);
template <int C>
__global__ void __launch_bounds__(64) SIDE_KERNEL_VGPRS k_msm_marginals_part(ReduceBatch rb) {
    PLK_SIDE_PRIO();   // short kernel on the critical path of the pipeline: issue ahead of co-resident accumulation waves
    marginals_part_body<C>(rb);
}
template <int C>
__global__ void __launch_bounds__(64) k_msm_marginals_part_tail(ReduceBatch rb) { marginals_part_body<C>(rb); }
template <int C>
__global__ void __launch_bounds__(64) SIDE_KERNEL_VGPRS k_msm_marginals_join(ReduceBatch rb) {
    PLK_SIDE_PRIO();
    marginals_join_body<C>(rb);
}
template <int C>
__global__ void __launch_bounds__(64) k_msm_marginals_join_tail(ReduceBatch rb) { marginals_join_body<C>(rb); }
template <int C>
__global__ void __launch_bounds__(64) k_msm_marginals_part_quad(ReduceBatch rb) { marginals_part_body<C, 4>(rb); }
template <int C>
__global__ void __launch_bounds__(256) k_msm_marginals_join_quad(ReduceBatch rb) { marginals_join_quad_body<C>(rb); }
template <int C, int G>
__device__ __forceinline__ void marginals_part_body(const ReduceBatch& rb) {
    using M = MargPlan<C>;
    const uint32_t* __restrict__ bucket_sums = rb.bucket_sums[blockIdx.y];
    uint32_t* __restrict__ part = rb.marg_part[blockIdx.y];
    const uint32_t lane_id = (blockIdx.x * 64 + threadIdx.x) / G, q = threadIdx.x & 3;     // < PARTIALS
    uint32_t o, p, stride;
    if (lane_id < M::NA * M::PART_A) { o = lane_id / M::PART_A; p = lane_id % M::PART_A; stride = M::PART_A; }
    else { const uint32_t r = lane_id - M::NA * M::PART_A; o = M::NA + r / M::PART_B; p = r % M::PART_B; stride = M::PART_B; }
    const uint32_t d = o < M::NA ? o >> 4 : MsmPlan<C>::FIELDS - 1, v = o < M::NA ? (o & 15) : o - M::NA;
    const uint32_t shift = 4 * d, bits = o < M::NA ? 4 : MsmPlan<C>::LAST_BITS;
    xyzz acc = xyzz_inf();
#pragma unroll 1
    for (int j = 0; j < MARG_K; ++j) {
        const uint32_t m = p + (uint32_t)j * stride;            // neighbouring lanes read neighbouring members
        const uint32_t idx = ((m >> shift) << (shift + bits)) | (v << shift) | (m & ((1u << shift) - 1));
        acc = add_g<G>(acc, ld_xyzz(bucket_sums + (size_t)idx * XYZZ_WORDS), q);
    }
    if (G == 1 || q == 0) st_xyzz(part + (size_t)lane_id * XYZZ_WORDS, acc);
}
template <int C>
__device__ __forceinline__ void marginals_join_body(const ReduceBatch& rb) {
    using M = MargPlan<C>;
    const uint32_t* __restrict__ part = rb.marg_part[blockIdx.y];
    uint32_t* __restrict__ out = rb.marg_dev[blockIdx.y];
    const uint32_t o = blockIdx.x, lane = threadIdx.x;
    const uint32_t first = o < M::NA ? o * M::PART_A : M::NA * M::PART_A + (o - M::NA) * M::PART_B, count = o < M::NA ? M::PART_A : M::PART_B;
    xyzz acc = xyzz_inf();
#pragma unroll 1
    for (uint32_t k = lane; k < count; k += 64) acc = xyzz_add(acc, ld_xyzz(part + (size_t)(first + k) * XYZZ_WORDS));
    for (int s = 32; s > 0; s >>= 1) {   // as in the fold: lanes without a partner must not add their own value
        const xyzz other = shfl_down_xyzz(acc, s);
        if (lane < (uint32_t)s) acc = xyzz_add(acc, other);
    }
    if (lane == 0) {
        uint32_t* dst = out + (size_t)o * 32;
        const fe to_r = fe_const(FqP::TO_R);
        st_fe<FqP>(dst, mul<FqP>(acc.x, to_r));
        st_fe<FqP>(dst + 8, mul<FqP>(acc.y, to_r));
        st_fe<FqP>(dst + 16, mul<FqP>(acc.zz, to_r));
        st_fe<FqP>(dst + 24, mul<FqP>(acc.zzz, to_r));
    }
}

// The join with four lanes per addition: a workgroup of four waves per output, 64 quads that take the partial sums in turn, a
// shuffle tree over the sixteen quads of a wave, the four waves' sums through LDS and two more levels in the first wave.
template <int C>
__device__ __forceinline__ void marginals_join_quad_body(const ReduceBatch& rb) {
    using M = MargPlan<C>;
    __shared__ uint32_t wave_sums[4 * XYZZ_WORDS];
    const uint32_t* __restrict__ part = rb.marg_part[blockIdx.y];
    uint32_t* __restrict__ out = rb.marg_dev[blockIdx.y];
    const uint32_t o = blockIdx.x, lane = threadIdx.x >> 2, q = threadIdx.x & 3, wave = threadIdx.x >> 6, lane_in_wave = lane & 15;
    const uint32_t first = o < M::NA ? o * M::PART_A : M::NA * M::PART_A + (o - M::NA) * M::PART_B, count = o < M::NA ? M::PART_A : M::PART_B;
    xyzz acc = xyzz_inf();
#pragma unroll 1
    for (uint32_t k = lane; k < count; k += 64) acc = xyzz_add_quad(acc, ld_xyzz(part + (size_t)(first + k) * XYZZ_WORDS), q);
    for (int s = 8; s > 0; s >>= 1) {   // lanes without a partner must not add their own value
        const xyzz other = shfl_down_xyzz(acc, 4 * s);
        if (lane_in_wave < (uint32_t)s) acc = xyzz_add_quad(acc, other, q);
    }
    if (lane_in_wave == 0 && q == 0) st_xyzz(wave_sums + wave * XYZZ_WORDS, acc);
    __syncthreads();
    if (wave != 0) return;
    acc = lane < 4 ? ld_xyzz(wave_sums + lane * XYZZ_WORDS) : xyzz_inf();
    for (int s = 2; s > 0; s >>= 1) {
        const xyzz other = shfl_down_xyzz(acc, 4 * s);
        if (lane < (uint32_t)s) acc = xyzz_add_quad(acc, other, q);
    }
    if (threadIdx.x == 0) {
        uint32_t* dst = out + (size_t)o * 32;
        const fe to_r = fe_const(FqP::TO_R);
        st_fe<FqP>(dst, mul<FqP>(acc.x, to_r));
        st_fe<FqP>(dst + 8, mul<FqP>(acc.y, to_r));
        st_fe<FqP>(dst + 16, mul<FqP>(acc.zz, to_r));
        st_fe<FqP>(dst + 24, mul<FqP>(acc.zzz, to_r));
    }
}

// ------------------------------------------------------------------------------------------ SRS tables
// x*2^256 words -> canonical x*2^261 words
__global__ void __launch_bounds__(256) k_fq_to_rp(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t n_elems) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_elems) return;
    st_fe<FqP>(out + 8 * i, mul<FqP>(ld_fe(in + 8 * i), fe_const(FqP::TO_RP)));
}
__global__ void __launch_bounds__(256) k_fq_to_r(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t n_elems) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_elems) return;
    st_fe<FqP>(out + 8 * i, mul<FqP>(ld_fe(in + 8 * i), fe_const(FqP::TO_R)));
}
__device__ __forceinline__ void store_affine_of(const xyzz& a, uint32_t* dst) {
    const fe i = inv<FqP>(mul<FqP>(a.zz, a.zzz));
    st_fe<FqP>(dst, mul<FqP>(a.x, mul<FqP>(i, a.zzz)));
    st_fe<FqP>(dst + 8, mul<FqP>(a.y, mul<FqP>(i, a.zz)));
}
// next[i] = 2^doublings * prev[i]
__global__ void __launch_bounds__(256) k_table_next(const uint32_t* __restrict__ prev, uint32_t* __restrict__ next, uint64_t count, int doublings) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    xyzz a = xyzz_dbl_affine(load_aff(prev + 16 * i));
    for (int k = 1; k < doublings; ++k) a = xyzz_dbl(a);
    store_affine_of(a, next + 16 * i);
}
// dst[i] = scale * tau^(first + i) * G through 8-bit fixed-base windows of G: gt[w][d] = d * 2^(8 w) * G.
// tau_pow2[k] = tau^(2^k), scale: x*2^256 words.
__global__ void __launch_bounds__(256) k_srs_generate(const uint32_t* __restrict__ gt, const uint32_t* __restrict__ tau_pow2, const uint32_t* __restrict__ scale,
                                                      uint64_t first, uint64_t count, uint32_t* __restrict__ dst, uint32_t world, uint32_t rank, uint32_t log_block) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const fe to_rp = fe_const(FrP::TO_RP);
    fe s = ld_fe(scale);
    // power of tau of local point i: contiguous shards start at `first`, block-cyclic ones own every world-th block
    const uint64_t e = world > 1 ? (((i >> log_block) * world + rank) << log_block | (i & ((1ull << log_block) - 1))) : first + i;
    for (int k = 0; k < 40; ++k)
        if ((e >> k) & 1) s = mul<FrP>(s, mul<FrP>(ld_fe(tau_pow2 + 8 * k), to_rp));
    uint32_t w[8];
    pack(reduce<FrP>(mul<FrP>(s, fe_const(FrP::RAW32))), w);
    xyzz acc = xyzz_inf();
    for (int j = 0; j < 32; ++j) {
        const uint32_t d = (w[j >> 2] >> (8 * (j & 3))) & 0xffu;
        if (d) xyzz_madd(acc, load_aff(gt + ((size_t)j * 256 + d) * 16));
    }
    store_affine_of(acc, dst + 16 * i);
}

// ------------------------------------------------------------------------------------------ host side
// What the host needs to know about a window plan on this context: its layout and the tables it runs on.
struct PlanView {
    int c, windows, buckets;
    const uint32_t* tables;
    uint64_t len;      // points of this rank the tables cover (a prefix; all of them for the long plan); the gamma powers follow
    uint64_t stride;   // points per window table: len + 3
};
static PlanView plan_view(const plk_ctx* ctx, int plan_c) {
    if (plan_c == MSM_C_SHORT) return {MSM_C_SHORT, MsmPlan<MSM_C_SHORT>::WINDOWS, MsmPlan<MSM_C_SHORT>::BUCKETS, ctx->tables_short, ctx->short_len, ctx->short_stride};
    if (plan_c == MSM_C) return {MSM_C, MsmPlan<MSM_C>::WINDOWS, MsmPlan<MSM_C>::BUCKETS, ctx->tables_mid, ctx->mid_len, ctx->mid_stride};
    return {MSM_C_LONG, MsmPlan<MSM_C_LONG>::WINDOWS, MsmPlan<MSM_C_LONG>::BUCKETS, ctx->tables, ctx->shard_len, ctx->table_stride};
}
int msm_plan_windows(int plan_c) { return plan_c == MSM_C_SHORT ? MsmPlan<MSM_C_SHORT>::WINDOWS : plan_c == MSM_C ? MsmPlan<MSM_C>::WINDOWS : MsmPlan<MSM_C_LONG>::WINDOWS; }
// f.template operator()<C>() for the plan's window width
#define PLK_FOR_PLAN(plan_c, ...)                                                         \
    do {                                                                                  \
        if ((plan_c) == MSM_C_SHORT) { constexpr int C = MSM_C_SHORT; __VA_ARGS__; }      \
        else if ((plan_c) == MSM_C) { constexpr int C = MSM_C; __VA_ARGS__; }             \
        else { constexpr int C = MSM_C_LONG; __VA_ARGS__; }                               \
    } while (0)
static int32_t slot_init(plk_ctx* ctx, MsmSlot& s) {
    if (s.ready) return PLK_OK;
    if (!ctx->marg_all_dev) {
        PLK_HIP(hipMalloc(&ctx->marg_all_dev, (size_t)MSM_SLOTS * MSM_MARGINALS * 32 * 4));
        PLK_HIP(hipHostMalloc(&ctx->marg_all_host, (size_t)MSM_SLOTS * MSM_MARGINALS * 32 * 4, hipHostMallocDefault));
    }
    const size_t idx = (size_t)(&s - ctx->slot);
    PLK_HIP(hipEventCreateWithFlags(&s.sorted, hipEventDisableTiming));
    PLK_HIP(hipEventCreateWithFlags(&s.accumulated, hipEventDisableTiming));
    PLK_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    PLK_HIP(hipMalloc(&s.counts, (size_t)SORT_MAX_BLOCKS * SORT_PARTS * 4));
    PLK_HIP(hipMalloc(&s.totals, (size_t)MSM_BUCKETS * 4));
    PLK_HIP(hipMalloc(&s.bucket_start, (size_t)(MSM_BUCKETS + 1) * 4));
    PLK_HIP(hipMalloc(&s.bucket_sums, (size_t)MSM_BUCKETS * XYZZ_WORDS * 4));
    PLK_HIP(hipMalloc(&s.flushed, (size_t)MSM_BUCKETS * XYZZ_WORDS * 4));
    s.marg_dev = ctx->marg_all_dev + idx * MSM_MARGINALS * 32;       // slices of the context's arrays: not freed with the slot
    s.marg_host = ctx->marg_all_host + idx * MSM_MARGINALS * 32;
    PLK_HIP(hipMalloc(&s.marg_part, (size_t)MARG_PARTIALS_MAX * XYZZ_WORDS * 4));
    PLK_HIP(hipMalloc(&s.blind_dev, 64 * 4));
    PLK_HIP(hipHostMalloc(&s.blind_host, 64 * 4, hipHostMallocDefault));
    PLK_HIP(hipMalloc(&s.res_dev, 64 * 4));
    PLK_HIP(hipHostMalloc(&s.res_host, 64 * 4, hipHostMallocDefault));
    s.ready = true;
    return PLK_OK;
}
// Entries per lane of the accumulation kernel: close to 64, chosen so that the launch is a whole number of chip-wide rounds
// (256 CUs x 4 SIMDs x 2 waves x 64 lanes), never below 16.
static uint32_t acc_run(uint64_t max_entries) {
    const uint64_t target = 64;   // measured inside a proof of 2^20 gates: 48 +0.25 ms, 96 +0.4 ms, 128 +0.55 ms
    const uint64_t chip = 256ull * 4 * 2 * 64;
    uint64_t rounds = (max_entries + chip * target / 2) / (chip * target);
    if (rounds < 1) rounds = 1;
    uint64_t run = (max_entries + rounds * chip - 1) / (rounds * chip);
    return (uint32_t)(run < 16 ? 16 : run);
}
static uint64_t acc_lanes(uint64_t max_entries, uint32_t run) {
    const uint64_t lanes = (max_entries + run - 1) / run;
    return lanes < 1 ? 1 : lanes;
}
static int32_t tails_reserve(plk_ctx* ctx, MsmSlot& s, uint64_t lanes) {
    lanes = (lanes + 63) & ~63ull;
    if (lanes <= s.tail_cap) return PLK_OK;
    if (s.tails) PLK_HIP(hipFree(s.tails));
    s.tails = nullptr; s.tail_cap = 0;
    PLK_HIP(hipMalloc(&s.tails, lanes * XYZZ_WORDS * 4));
    s.tail_cap = lanes;
    return PLK_OK;
}
// Sizes the per-MSM scratch of every slot for the largest job this SRS admits, so that no allocation (an implicit device
// synchronisation) happens while proofs are running.
static int32_t slots_reserve(plk_ctx* ctx) {
    uint64_t entries = 0;
    for (int c : {MSM_C_LONG, MSM_C, MSM_C_SHORT}) {
        const PlanView v = plan_view(ctx, c);
        if (v.tables) entries = std::max<uint64_t>(entries, (v.len + 3) * v.windows);
    }
    entries = (entries + 1024 + 15) & ~15ull;
    const uint64_t scal_words = (ctx->srs_total + 1) * 8, chain_words = division_work_words(ctx->srs_total);
    for (auto& s : ctx->slot) {
        PLK_TRY(slot_init(ctx, s));
        if (s.entries_cap < entries) { if (s.entries) PLK_HIP(hipFree(s.entries)); s.entries = nullptr; s.entries_cap = 0; PLK_HIP(hipMalloc(&s.entries, entries * 12)); s.entries_cap = entries; }   // per entry: 4 bytes sorted entry + 8 record of the two-level sort
        PLK_TRY(tails_reserve(ctx, s, acc_lanes(entries, 16)));
        if (s.scal_cap < scal_words) { if (s.scal_dev) PLK_HIP(hipFree(s.scal_dev)); s.scal_dev = nullptr; s.scal_cap = 0; PLK_HIP(hipMalloc(&s.scal_dev, scal_words * 4)); s.scal_cap = scal_words; }
        if (s.chain_cap < chain_words) { if (s.chain_dev) PLK_HIP(hipFree(s.chain_dev)); s.chain_dev = nullptr; s.chain_cap = 0; PLK_HIP(hipMalloc(&s.chain_dev, chain_words * 4)); s.chain_cap = chain_words; }
    }
    return PLK_OK;
}
void slot_destroy(MsmSlot& s) {
    if (!s.ready) return;
    hipFree(s.counts); hipFree(s.totals); hipFree(s.bucket_start); hipFree(s.entries); hipFree(s.bucket_sums); hipFree(s.flushed); hipFree(s.tails); hipFree(s.marg_part);
    hipFree(s.blind_dev); hipHostFree(s.blind_host); hipFree(s.scal_dev); hipFree(s.chain_dev);
    hipFree(s.res_dev); hipHostFree(s.res_host);
    hipEventDestroy(s.sorted); hipEventDestroy(s.accumulated); hipEventDestroy(s.done);
    s = MsmSlot();
}

int32_t msm_acquire_slot(plk_ctx* ctx, int* out) {
    for (int k = 0; k < MSM_SLOTS; ++k) {
        const int i = (ctx->next_slot + k) % MSM_SLOTS;
        if (!ctx->slot[i].busy) {
            PLK_TRY(slot_init(ctx, ctx->slot[i]));
            ctx->next_slot = (i + 1) % MSM_SLOTS;
            *out = i;
            return PLK_OK;
        }
    }
    ctx->err = "all MSM slots are busy: end an open ticket first";
    return PLK_ERR_INVALID_ARG;
}

// Number of SRS points with global index < x that a block-cyclic shard holds (they are packed in index order).
static uint64_t cyclic_owned_below(uint32_t world, uint32_t rank, uint32_t log_block, uint64_t x) {
    const uint64_t B = 1ull << log_block, blk = x >> log_block, rem = x & (B - 1);
    const uint64_t cycles = blk / world, pos = blk % world;
    return cycles * B + (pos > rank ? B : (pos == rank ? rem : 0));
}
// Restricts the global range [base_offset, base_offset + k) to this context's shard.
static void clip_to_shard(const plk_ctx* ctx, const uint32_t* scalars, uint64_t k, uint64_t base_offset, MsmSeg* seg) {
    seg->world = 1; seg->rank = 0; seg->log_block = 0; seg->origin = 0;
    if (ctx->cyc_world > 1) {
        const uint64_t t_lo = cyclic_owned_below(ctx->cyc_world, ctx->cyc_rank, ctx->cyc_log_block, base_offset);
        const uint64_t t_hi = cyclic_owned_below(ctx->cyc_world, ctx->cyc_rank, ctx->cyc_log_block, base_offset + k);
        seg->scalars = scalars; seg->origin = base_offset;
        seg->count = (uint32_t)(t_hi - t_lo); seg->table_off = (uint32_t)t_lo;
        seg->world = ctx->cyc_world; seg->rank = ctx->cyc_rank; seg->log_block = ctx->cyc_log_block;
        return;
    }
    uint64_t lo = base_offset, hi = base_offset + k;
    const uint64_t slo = ctx->shard_first, shi = ctx->shard_first + ctx->shard_len;
    if (lo < slo) lo = slo;
    if (hi > shi) hi = shi;
    if (hi <= lo) { seg->scalars = scalars; seg->count = 0; seg->table_off = 0; return; }
    seg->scalars = scalars + 8 * (lo - base_offset);
    seg->count = (uint32_t)(hi - lo);
    seg->table_off = (uint32_t)(lo - slo);
}
// Scalars of the global range [base_offset, base_offset + k) that fall to this context's shard.
uint64_t msm_share(const plk_ctx* ctx, uint64_t k, uint64_t base_offset) {
    MsmSeg seg;
    clip_to_shard(ctx, nullptr, k, base_offset, &seg);
    return seg.count;
}
// Window width for the MSM over the global range [base_offset, base_offset + k): by the number of scalars on this rank; the two
// narrower plans only when the range lies inside the prefix of the SRS that their tables cover.
int msm_plan_for(const plk_ctx* ctx, uint64_t k, uint64_t base_offset) {
    if (base_offset + k > ctx->srs_total) return MSM_C_LONG;
    MsmSeg seg;
    clip_to_shard(ctx, nullptr, k, base_offset, &seg);
    const uint64_t end = (uint64_t)seg.table_off + seg.count;
    if (ctx->tables_short && seg.count <= ctx->short_max && end <= ctx->short_len) return MSM_C_SHORT;
    if (ctx->tables_mid && seg.count <= ctx->mid_max && end <= ctx->mid_len) return MSM_C;
    return MSM_C_LONG;
}

// The stream for the front work (division, sort) of the next job: consecutive jobs alternate between the compute stream and the
// reduction stream, so that the sort of one runs beside the division or sort of the next (0.2-0.3 ms per proof of 2^20 gates
// against a single front stream).  Detached compute work and pending uploads own the reduction stream: then every job stays on the
// compute stream.
hipStream_t front_lane(plk_ctx* ctx) {
    if (ctx->detached || ctx->uploads_pending) return ctx->st_sort;
    return (ctx->lane_turn++ & 1) ? ctx->st_red : ctx->st_sort;
}

template <int C>
static void launch_sort(hipStream_t st, const SortBatch& sb, uint32_t max_blocks, uint32_t n, bool fast) {
    hipLaunchKernelGGL(k_sort_count<C>, dim3(max_blocks, n), dim3(SORT_THREADS), 0, st, sb);
    hipLaunchKernelGGL(k_sort_partition<C>, dim3(max_blocks, n), dim3(SORT_THREADS), 0, st, sb);
    if (fast) hipLaunchKernelGGL((k_sort_buckets<1024, C>), dim3(SORT_PARTS, n), dim3(1024), 0, st, sb);
    else hipLaunchKernelGGL((k_sort_buckets<SORT_THREADS, C>), dim3(SORT_PARTS, n), dim3(SORT_THREADS), 0, st, sb);
}
// Enqueues the MSMs of `n` jobs that share their launches (one sort, one accumulation; their reductions are batched by
// msm_flush_reductions anyway) and one window plan.  Per job: the main segment (device scalars over powers_of_g) plus, when
// n_blind > 0 and this shard holds the gamma powers, n_blind scalars over powers_of_gamma_g.  The sort goes to `lane`.
int32_t msm_enqueue_group(plk_ctx* ctx, const MsmJob* jobs, uint32_t n, hipStream_t lane, int idle_hint, int plan_c) {
    if (!ctx->tables) { ctx->err = "no SRS loaded"; return PLK_ERR_INVALID_ARG; }
    if (n == 0 || n > (uint32_t)MSM_SLOTS) return PLK_ERR_INVALID_ARG;
    const PlanView plan = plan_view(ctx, plan_c);
    if (!plan.tables) { ctx->err = "window plan without its tables"; return PLK_ERR_INVALID_ARG; }
    const uint32_t windows = (uint32_t)plan.windows;
    hipStream_t st = lane ? lane : ctx->st_sort;
    bool idle = ctx->reduce_pending.empty();   // nothing of this context is running or queued on the device
    for (const auto& other : ctx->slot) {
        bool mine = false;
        for (uint32_t m = 0; m < n; ++m) mine = mine || &other == &ctx->slot[jobs[m].slot];
        idle = idle && (!other.busy || mine);
    }
    if (idle_hint >= 0) idle = idle_hint != 0;   // a batch marks all its slots busy before it enqueues the first
    SortBatch sb;
    memset(&sb, 0, sizeof(sb));
    sb.table_stride = (uint32_t)plan.stride;
    uint64_t max_entries[MSM_SLOTS], total_entries = 0;
    uint32_t max_blocks = 1;
    // Raised issue priority for every sort: given to the first four sorts of a round only, the later ones take twice as long, the
    // accumulation kernels 3 % less, and the proof the same or 0.1 ms more; to the first two only, 1.7 ms more.
    // 8192 scalars per workgroup (pieces of 512 records per partition), down to 1024 for short MSMs, never more than 512 workgroups
    // (every workgroup of the partitioning kernel reads the whole count matrix).  Measured inside a proof of 2^20 gates: 4096 scalars
    // per workgroup the same, 16384 0.6 ms slower; four workgroups per partition in the second level (1024 workgroups, two launches)
    // halve the isolated sort and make the proof 0.7 ms slower; without the raised issue priority 0.9 ms slower.
    // The first sort of a round finds the chip idle and the round waiting for it: four times as many workgroups in the first level
    // (two per CU at 2^20 scalars; 0.1 ms per proof better than twice as many, workgroups of 512 or 1024 threads instead: no better),
    // sixteen waves per partition in the second (0.2 ms per proof).
    const bool fast = idle;
    for (uint32_t m = 0; m < n; ++m) {
        const MsmJob& j = jobs[m];
        MsmSlot& s = ctx->slot[j.slot];
        if (j.base_offset + j.k > ctx->srs_total) { ctx->err = "MSM longer than the SRS"; return PLK_ERR_SRS_TOO_SHORT; }
        MsmSegs& segs = sb.segs[m];
        clip_to_shard(ctx, j.scalars, j.k, j.base_offset, &segs.s[0]);
        if ((uint64_t)segs.s[0].table_off + segs.s[0].count > plan.len) { ctx->err = "MSM outside the prefix that the plan's tables cover"; return PLK_ERR_INVALID_ARG; }
        segs.s[1].scalars = j.blind_dev ? j.blind_dev : s.blind_dev;
        segs.s[1].count = ctx->has_gamma ? j.n_blind : 0;
        segs.s[1].table_off = (uint32_t)plan.len;
        segs.s[1].world = 1; segs.s[1].rank = 0; segs.s[1].log_block = 0; segs.s[1].origin = 0;
        segs.nseg = 2;
        segs.total = segs.s[0].count + segs.s[1].count;
        max_entries[m] = (uint64_t)segs.total * windows;
        total_entries += max_entries[m];
        if (max_entries[m] > s.entries_cap) {   // the slot is idle here: its previous job was waited for before it was handed out again
            if (s.entries) PLK_HIP(hipFree(s.entries));
            s.entries = nullptr; s.entries_cap = 0;
            const uint64_t cap = (max_entries[m] + max_entries[m] / 8 + 1024 + 15) & ~15ull;
            PLK_HIP(hipMalloc(&s.entries, cap * 12));   // 4 bytes per entry + 8 per record
            s.entries_cap = cap;
        }
        uint32_t per = fast ? 2048 : 8192;
        while (per > 1024 && (segs.total + per - 1) / per < 128) per >>= 1;
        uint32_t nb = (segs.total + per - 1) / per;
        if (nb < 1) nb = 1;
        if (nb > (uint32_t)SORT_MAX_BLOCKS) nb = SORT_MAX_BLOCKS;
        sb.nblocks[m] = nb;
        sb.chunk[m] = (segs.total + nb - 1) / nb;
        sb.records[m] = reinterpret_cast<uint64_t*>(s.entries + s.entries_cap);
        sb.counts[m] = s.counts;
        sb.part_start[m] = s.totals;                   // SORT_PARTS + 1 words (the buffer holds MSM_BUCKETS)
        sb.bucket_start[m] = s.bucket_start;
        sb.entries[m] = s.entries;
        max_blocks = std::max(max_blocks, nb);
    }
    // run length of the accumulation from the whole group: together the members should fill the chip a whole number of times
    uint32_t run = acc_run(total_entries);
    if (plan_c == MSM_C_SHORT) { static const uint32_t min_run = getenv("PLK_NARROW_MIN_RUN") ? (uint32_t)atoi(getenv("PLK_NARROW_MIN_RUN")) : 16; run = std::max(run, min_run); }
    AccBatch ab;
    memset(&ab, 0, sizeof(ab));
    ab.n = n;
    uint32_t blocks_so_far = 0;
    for (uint32_t m = 0; m < n; ++m) {
        MsmSlot& s = ctx->slot[jobs[m].slot];
        const uint64_t lanes = acc_lanes(max_entries[m], run);
        PLK_TRY(tails_reserve(ctx, s, lanes));
        const uint32_t blocks = (uint32_t)((lanes + ACC_THREADS - 1) / ACC_THREADS);
        ab.entries[m] = s.entries; ab.bucket_start[m] = s.bucket_start; ab.tails[m] = s.tails; ab.flushed[m] = s.flushed;
        ab.tail_stride[m] = (uint32_t)s.tail_cap;
        ab.block_first[m] = blocks_so_far;
        blocks_so_far += blocks;
        s.acc_blocks = blocks;
    }
    for (uint32_t m = n; m <= (uint32_t)MSM_SLOTS; ++m) ab.block_first[m] = blocks_so_far;
    MsmSlot& leader = ctx->slot[jobs[0].slot];
    prof_begin(ctx, st, "msm_sort");
    PLK_FOR_PLAN(plan_c, launch_sort<C>(st, sb, max_blocks, n, fast));
    prof_end(ctx, st);
    PLK_HIP(hipEventRecord(leader.sorted, st));
    PLK_TRY(lane_read_mark(ctx, st, leader.sorted));   // behind the last reader of the callers' scalars (and of the division in front of the sort)
    // The accumulation goes to the stream with less work queued (entries of the jobs enqueued since the device was last idle): the kernel
    // behind a long one would otherwise wait for it to drain completely while the other stream has nothing left to run.
    if (idle) ctx->acc_load[0] = ctx->acc_load[1] = 0;
    const int acc_stream = ctx->acc_load[1] < ctx->acc_load[0] ? 1 : 0;
    ctx->acc_load[acc_stream] += total_entries + 65536;   // a launch costs something even when it is empty
    st = ctx->st_acc[acc_stream];
    PLK_HIP(hipStreamWaitEvent(st, leader.sorted, 0));
    prof_begin(ctx, st, "msm_accumulate");
    PLK_FOR_PLAN(plan_c, hipLaunchKernelGGL(k_msm_accumulate<MsmPlan<C>::BUCKETS>, dim3(blocks_so_far), dim3(ACC_THREADS), 0, st, ab, plan.tables));
    prof_end(ctx, st);
    PLK_HIP(hipEventRecord(leader.accumulated, st));
    for (uint32_t m = 0; m < n; ++m) {
        MsmSlot& s = ctx->slot[jobs[m].slot];
        s.max_entries = max_entries[m];
        s.acc_stream = acc_stream;
        s.plan_c = plan_c;
        s.ev_owner = jobs[0].slot;
        s.done_owner = jobs[m].slot;
        ctx->reduce_pending.push_back(jobs[m].slot);
    }
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}
int32_t msm_enqueue(plk_ctx* ctx, MsmSlot& s, const uint32_t* scalars_dev, uint64_t k, uint64_t base_offset, uint32_t n_blind, hipStream_t lane, int idle_hint,
                    const uint32_t* blind_dev) {
    if (!ctx->tables) { ctx->err = "no SRS loaded"; return PLK_ERR_INVALID_ARG; }
    if (base_offset + k > ctx->srs_total) { ctx->err = "MSM longer than the SRS"; return PLK_ERR_SRS_TOO_SHORT; }
    MsmJob j;
    j.slot = (int)(&s - ctx->slot); j.scalars = scalars_dev; j.k = k; j.base_offset = base_offset; j.n_blind = n_blind; j.blind_dev = blind_dev;
    return msm_enqueue_group(ctx, &j, 1, lane, idle_hint, msm_plan_for(ctx, k, base_offset));
}

// The reduction stream is a queue: a reduction that waits for a long accumulation holds up every reduction behind it, also those
// whose accumulation finished long ago.  Reductions are therefore not enqueued with their MSM but when the first result of a
// batch is asked for, in the order in which the accumulations are expected to finish (two streams, taken alternately, each
// kernel's time proportional to its entries).
template <int C>
static void launch_reduction(hipStream_t st, const ReduceBatch& rb, uint64_t tails_per_bucket, bool tail) {
    constexpr int BUCKETS = MsmPlan<C>::BUCKETS, PARTIALS = MargPlan<C>::PARTIALS, OUTPUTS = MsmPlan<C>::MARGINALS;
    // runs that end inside an average bucket; two lanes per bucket keep the fold cheap, more lanes keep its serial chain short
    if (tail) {
        // the batch ends with this reduction: all registers, no spills; and where a launch would leave most SIMDs without a wave
        // (1024 of them), four lanes share every addition
        static const bool quads = !(getenv("PLK_NO_QUADS") && atoi(getenv("PLK_NO_QUADS")));
        const int fl = tails_per_bucket <= 12 ? 2 : tails_per_bucket <= 48 ? 4 : 8;
        const uint64_t fold_waves = (uint64_t)BUCKETS * fl / 64 * rb.n;
        if (quads && fold_waves * 4 <= 2048) {
            if (fl == 2) hipLaunchKernelGGL((k_msm_fold_quad<2, C>), dim3(BUCKETS * 2 * 4 / 256, rb.n), dim3(256), 0, st, rb);
            else if (fl == 4) hipLaunchKernelGGL((k_msm_fold_quad<4, C>), dim3(BUCKETS * 4 * 4 / 256, rb.n), dim3(256), 0, st, rb);
            else hipLaunchKernelGGL((k_msm_fold_quad<8, C>), dim3(BUCKETS * 8 * 4 / 256, rb.n), dim3(256), 0, st, rb);
        } else {
            if (fl == 2) hipLaunchKernelGGL((k_msm_fold_tail<2, C>), dim3(BUCKETS * 2 / 256, rb.n), dim3(256), 0, st, rb);
            else if (fl == 4) hipLaunchKernelGGL((k_msm_fold_tail<4, C>), dim3(BUCKETS * 4 / 256, rb.n), dim3(256), 0, st, rb);
            else hipLaunchKernelGGL((k_msm_fold_tail<8, C>), dim3(BUCKETS * 8 / 256, rb.n), dim3(256), 0, st, rb);
        }
        if (quads && (uint64_t)PARTIALS / 64 * rb.n * 4 <= 1536) hipLaunchKernelGGL(k_msm_marginals_part_quad<C>, dim3(PARTIALS * 4 / 64, rb.n), dim3(64), 0, st, rb);
        else hipLaunchKernelGGL(k_msm_marginals_part_tail<C>, dim3(PARTIALS / 64, rb.n), dim3(64), 0, st, rb);
        if (quads && (uint64_t)OUTPUTS * rb.n * 4 <= 2048) hipLaunchKernelGGL(k_msm_marginals_join_quad<C>, dim3(OUTPUTS, rb.n), dim3(256), 0, st, rb);
        else hipLaunchKernelGGL(k_msm_marginals_join_tail<C>, dim3(OUTPUTS, rb.n), dim3(64), 0, st, rb);
    } else {
        if (tails_per_bucket <= 12) hipLaunchKernelGGL((k_msm_fold<2, C>), dim3(BUCKETS * 2 / 256, rb.n), dim3(256), 0, st, rb);
        else if (tails_per_bucket <= 48) hipLaunchKernelGGL((k_msm_fold<4, C>), dim3(BUCKETS * 4 / 256, rb.n), dim3(256), 0, st, rb);
        else hipLaunchKernelGGL((k_msm_fold<8, C>), dim3(BUCKETS * 8 / 256, rb.n), dim3(256), 0, st, rb);
        hipLaunchKernelGGL(k_msm_marginals_part<C>, dim3(PARTIALS / 64, rb.n), dim3(64), 0, st, rb);
        hipLaunchKernelGGL(k_msm_marginals_join<C>, dim3(OUTPUTS, rb.n), dim3(64), 0, st, rb);
    }
}
static int32_t enqueue_reduction(plk_ctx* ctx, const std::vector<int>& group, bool tail) {
    hipStream_t st = ctx->st_red;
    ReduceBatch rb;
    memset(&rb, 0, sizeof(rb));
    rb.n = (int32_t)group.size();
    const int plan_c = ctx->slot[group[0]].plan_c;
    const uint64_t buckets = (uint64_t)plan_view(ctx, plan_c).buckets;
    uint64_t tails_per_bucket = 0;
    int waited = -1, lo = MSM_SLOTS, hi = -1;
    for (int i = 0; i < rb.n; ++i) {
        MsmSlot& s = ctx->slot[group[i]];
        if (s.ev_owner != waited) { PLK_HIP(hipStreamWaitEvent(st, ctx->slot[s.ev_owner].accumulated, 0)); waited = s.ev_owner; }   // members of a batch share one event
        rb.bucket_start[i] = s.bucket_start; rb.tails[i] = s.tails; rb.flushed[i] = s.flushed;
        rb.tail_stride[i] = (uint32_t)s.tail_cap; rb.lanes_launched[i] = s.acc_blocks * ACC_THREADS;
        rb.bucket_sums[i] = s.bucket_sums; rb.marg_part[i] = s.marg_part; rb.marg_dev[i] = s.marg_dev;
        tails_per_bucket = std::max<uint64_t>(tails_per_bucket, (uint64_t)s.acc_blocks * ACC_THREADS / buckets);
        lo = std::min(lo, group[i]); hi = std::max(hi, group[i]);
    }
    prof_begin(ctx, st, "msm_reduce");
    PLK_FOR_PLAN(plan_c, launch_reduction<C>(st, rb, tails_per_bucket, tail));
    prof_end(ctx, st);
    PLK_HIP(hipGetLastError());
    // One copy brings back the outputs of the whole launch: the slots' output arrays lie side by side.  Slots between the first and
    // the last of the group that belong to other jobs are copied as well: their device and host contents are equal at this point
    // (a slot's outputs change only in its own reduction, which is followed by its own copy on this same stream).
    const size_t slice = (size_t)MSM_MARGINALS * 32;
    PLK_HIP(hipMemcpyAsync(ctx->marg_all_host + lo * slice, ctx->marg_all_dev + lo * slice, (size_t)(hi - lo + 1) * slice * 4, hipMemcpyDeviceToHost, st));
    PLK_HIP(hipEventRecord(ctx->slot[group[0]].done, st));
    for (int i = 0; i < rb.n; ++i) ctx->slot[group[i]].done_owner = group[0];
    return PLK_OK;
}
// Which MSMs share a reduction launch.  All results of a round are asked for together, so what counts is when the last reduction
// ends: a launch costs about the same for one member as for ten, and the reduction stream runs one launch at a time.  The
// accumulation launches are replayed on a model of the device (two streams that share the chip, a kernel's time proportional to its
// entries), then the reduction stream: whenever it is free, everything whose accumulation is over by then goes into one launch
// (per window plan: a launch is compiled for one bucket count).  Long MSMs (a proof of 2^20 gates on one GPU) still get a launch
// each, beside the accumulations that follow; the short MSMs of a sharded or small proof, which finish while the previous
// reduction is running, share one (measured on one rank of eight at 2^20 gates: three reductions one after the other at the end of
// every round before, one or two now).
int32_t msm_flush_reductions(plk_ctx* ctx) {
    if (ctx->reduce_pending.empty()) return PLK_OK;
    // accumulation launches in the order they were enqueued (the members of a batch share one)
    struct Launch { int owner; int stream; double work; double finish; };
    std::vector<Launch> launches;
    for (int si : ctx->reduce_pending) {
        const MsmSlot& s = ctx->slot[si];
        if (launches.empty() || launches.back().owner != s.ev_owner) launches.push_back({s.ev_owner, s.acc_stream, 65536.0, 0.0});   // a launch costs something even when it is empty
        launches.back().work += (double)s.max_entries;
    }
    {
        std::vector<size_t> queue[2];
        for (size_t i = 0; i < launches.size(); ++i) queue[launches[i].stream].push_back(i);
        size_t head[2] = {0, 0};
        double rem[2] = {queue[0].empty() ? 0 : launches[queue[0][0]].work, queue[1].empty() ? 0 : launches[queue[1][0]].work}, t = 0;
        for (;;) {
            const bool a0 = head[0] < queue[0].size(), a1 = head[1] < queue[1].size();
            if (!a0 && !a1) break;
            const int active = (a0 ? 1 : 0) + (a1 ? 1 : 0);
            const double step = a0 && a1 ? std::min(rem[0], rem[1]) : (a0 ? rem[0] : rem[1]);   // progress of every running kernel
            t += step * active;
            for (int q = 0; q < 2; ++q) {
                if (head[q] >= queue[q].size()) continue;
                rem[q] -= step;
                if (rem[q] <= 1e-6) {
                    launches[queue[q][head[q]]].finish = t;
                    if (++head[q] < queue[q].size()) rem[q] = launches[queue[q][head[q]]].work;
                }
            }
        }
    }
    struct Item { double finish; int slot; };
    std::vector<Item> pending;
    double last_finish = 0;
    for (int si : ctx->reduce_pending) {
        for (const auto& l : launches) if (l.owner == ctx->slot[si].ev_owner) { pending.push_back({l.finish, si}); last_finish = std::max(last_finish, l.finish); break; }
    }
    ctx->reduce_pending.clear();
    std::stable_sort(pending.begin(), pending.end(), [](const Item& x, const Item& y) { return x.finish < y.finish; });
    // in units of accumulated entries (about 14 per nanosecond): a reduction beside accumulation kernels 0.35 ms on the wide plan,
    // 0.1 ms on the narrow one; an MSM that finishes within 0.05 ms of the launch is waited for
    constexpr double REDUCE_WIDE = 5.0e6, REDUCE_NARROW = 1.5e6, SLACK = 0.7e6;
    double red_free = 0;
    while (!pending.empty()) {
        const double start = std::max(red_free, pending[0].finish);
        const int plan = ctx->slot[pending[0].slot].plan_c;
        const double cost = plan == MSM_C_SHORT ? REDUCE_NARROW : REDUCE_WIDE;
        // when everything still outstanding is over before a launch begun now would be, one launch for all of it ends sooner than two
        const double reach = last_finish <= start + cost ? cost : SLACK;
        std::vector<int> group;
        std::vector<Item> rest;
        double ready = start;
        for (const auto& it : pending) {
            if (it.finish <= start + reach && ctx->slot[it.slot].plan_c == plan) { group.push_back(it.slot); ready = std::max(ready, it.finish); }
            else rest.push_back(it);
        }
        // nothing is accumulating any more when this launch starts: the builds without a register limit
        const bool tail = last_finish <= ready + SLACK;
        PLK_TRY(enqueue_reduction(ctx, group, tail));
        red_free = ready + cost;
        pending.swap(rest);
    }
    return PLK_OK;
}

// Host tail: sum_k k*B_k from the marginal sums.  weight of bucket index idx is idx + 1.
// In pieces, so that several host threads can share one tail: piece 0 = the sum of all buckets (the "+1": every bucket has exactly
// one value in field 0), piece 1 + d = sum_v v M(d, v) of field d by running sums; msm_tail_combine puts them together.
constexpr int TAIL_PIECES = 1 + 4;
static int plan_fields(int plan_c) { int f; PLK_FOR_PLAN(plan_c, f = MsmPlan<C>::FIELDS); return f; }
static int plan_last_bits(int plan_c) { int b; PLK_FOR_PLAN(plan_c, b = MsmPlan<C>::LAST_BITS); return b; }
void msm_tail_piece(const uint32_t* marg, int plan_c, int piece, host::pt* out) {
    using namespace host;
    const uint64_t* m = reinterpret_cast<const uint64_t*>(marg);
    auto M = [&](int o) { return pt_load(m + 16 * o); };
    const int fields = plan_fields(plan_c);
    if (piece == 0) {
        pt total = pt_inf();
        for (int v = 0; v < 16; ++v) total = pt_add(total, M(v));
        *out = total;
        return;
    }
    const int d = piece - 1;
    if (d >= fields) { *out = pt_inf(); return; }
    const int nv = d < fields - 1 ? 16 : (1 << plan_last_bits(plan_c)), base = 16 * d;
    pt run = pt_inf(), s = pt_inf();
    for (int v = nv - 1; v >= 1; --v) { run = pt_add(run, M(base + v)); s = pt_add(s, run); }
    *out = s;
}
void msm_tail_combine(const host::pt* pieces, int plan_c, host::pt* out) {
    using namespace host;
    const int fields = plan_fields(plan_c);
    pt acc = pt_inf();
    for (int d = fields - 1; d >= 0; --d) {
        if (d < fields - 1) for (int k = 0; k < 4; ++k) acc = pt_dbl(acc);    // the field above sits 4 bits higher
        acc = pt_add(acc, pieces[1 + d]);
    }
    *out = pt_add(acc, pieces[0]);
}
void msm_host_tail(const uint32_t* marg, int plan_c, host::pt* out) {
    host::pt pieces[TAIL_PIECES];
    for (int i = 0; i < TAIL_PIECES; ++i) msm_tail_piece(marg, plan_c, i, &pieces[i]);
    msm_tail_combine(pieces, plan_c, out);
}

int32_t msm_wait(plk_ctx* ctx, MsmSlot& s, host::pt* out) {
    PLK_TRY(msm_flush_reductions(ctx));
    PLK_HIP(hipEventSynchronize(ctx->slot[s.done_owner].done));
    msm_host_tail(s.marg_host, s.plan_c, out);
    return PLK_OK;
}

// ---- SRS -------------------------------------------------------------------------------------------
static int32_t srs_alloc(plk_ctx* ctx, uint64_t shard_first, uint64_t shard_len, uint64_t total_len) {
    if (shard_first + shard_len > total_len || shard_len == 0 || shard_len + 3 >= (1ull << 27)) { ctx->err = "bad SRS shard"; return PLK_ERR_INVALID_ARG; }
    if (ctx->tables) { PLK_HIP(hipDeviceSynchronize()); PLK_HIP(hipFree(ctx->tables)); ctx->tables = nullptr; }
    if (ctx->tables_short) { PLK_HIP(hipFree(ctx->tables_short)); ctx->tables_short = nullptr; }
    if (ctx->tables_mid) { PLK_HIP(hipFree(ctx->tables_mid)); ctx->tables_mid = nullptr; }
    ctx->short_len = 0; ctx->short_stride = 0; ctx->mid_len = 0; ctx->mid_stride = 0;
    ctx->srs_total = total_len; ctx->shard_first = shard_first; ctx->shard_len = shard_len;
    ctx->cyc_world = 1; ctx->cyc_rank = 0; ctx->cyc_log_block = 0;
    ctx->has_gamma = shard_first == 0;
    ctx->table_stride = shard_len + 3;
    PLK_HIP(hipMalloc(&ctx->tables, (size_t)MsmPlan<MSM_C_LONG>::WINDOWS * ctx->table_stride * 64));
    if (!ctx->has_gamma) PLK_HIP(hipMemsetAsync(ctx->tables + 16 * shard_len, 0, 3 * 64, ctx->stream));
    return PLK_OK;
}
// Block-cyclic shard: blocks of 2^log_block points, block b belongs to rank b mod world.
static int32_t srs_alloc_cyclic(plk_ctx* ctx, uint32_t rank, uint32_t world, uint32_t log_block, uint64_t total_len) {
    if (world == 0 || rank >= world || log_block > 24 || total_len == 0) { ctx->err = "bad cyclic shard"; return PLK_ERR_INVALID_ARG; }
    if (world == 1) return srs_alloc(ctx, 0, total_len, total_len);
    const uint64_t held = cyclic_owned_below(world, rank, log_block, total_len);
    // a rank may hold nothing of a very short SRS; it still needs a (one-point) table so that every launch has valid operands
    PLK_TRY(srs_alloc(ctx, 0, held ? held : 1, total_len + (held ? 0 : 1)));
    ctx->srs_total = total_len;
    ctx->shard_len = held;
    ctx->cyc_world = world; ctx->cyc_rank = rank; ctx->cyc_log_block = log_block;
    ctx->has_gamma = rank == 0;
    if (!ctx->has_gamma) PLK_HIP(hipMemsetAsync(ctx->tables + 16 * ctx->shard_len, 0, 3 * 64, ctx->stream));
    return PLK_OK;
}
// MSMs of at most this many scalars (on this rank) take the 13-bit plan.  Chosen where the narrow plan stops paying: it adds a
// quarter to the additions of the accumulation and takes seven eighths off the reduction of the buckets.  Measured: a proof of 2^16
// gates (MSMs of 2^16 scalars) 4.02-4.04 ms against 4.14-4.2 ms with 16-bit windows throughout; one rank of eight at 2^20 gates
// (MSMs of 2^17 scalars) 7.3 against 6.8 ms.  PLK_MSM_SHORT_MAX overrides it (0: no MSM on 13-bit windows).
static uint64_t short_max_setting() {
    const char* e = getenv("PLK_MSM_SHORT_MAX");
    if (e && *e) return strtoull(e, nullptr, 0);
    return (1ull << 16) + 4096;
}
// MSMs of at most this many scalars (and more than the above) take the 16-bit plan, longer ones the 17-bit plan.  Measured on one
// rank of eight at 2^22 gates (MSMs of 2^19 scalars): 19.7-19.9 ms with 17-bit windows against 20.3 ms with 16-bit ones; one rank
// of four at 2^20 gates (2^18 scalars): 9.89-10.06 against 9.82-10.0 ms; of eight (2^17 scalars): 6.21-6.24 against 6.16-6.21 ms;
// a proof of 2^20 gates on one GPU (2^20 scalars): 29.2 against 29.7 ms.  PLK_MSM_MID_MAX overrides it (0: no MSM on 16-bit windows).
static uint64_t mid_max_setting() {
    const char* e = getenv("PLK_MSM_MID_MAX");
    if (e && *e) return strtoull(e, nullptr, 0);
    return (1ull << 18) + 8192;
}
// Tables of a narrower plan over a prefix of this rank's points (every MSM of the prover starts at power 0, so a short MSM lies in
// the prefix) and the gamma powers: windows x (prefix + 3) points.
static int32_t build_prefix_tables(plk_ctx* ctx, int c, int windows, uint64_t max_share, uint32_t** tables, uint64_t* len, uint64_t* stride) {
    *len = std::min<uint64_t>(ctx->shard_len, max_share);
    *stride = *len + 3;
    PLK_HIP(hipMalloc(tables, (size_t)windows * *stride * 64));
    PLK_HIP(hipMemcpyAsync(*tables, ctx->tables, *len * 64, hipMemcpyDeviceToDevice, ctx->stream));
    PLK_HIP(hipMemcpyAsync(*tables + 16 * *len, ctx->tables + 16 * ctx->shard_len, 3 * 64, hipMemcpyDeviceToDevice, ctx->stream));   // zeros on the ranks without gamma powers
    const uint64_t cnt = ctx->has_gamma ? *stride : *len;
    for (int j = 1; j < windows; ++j) {
        const uint32_t* prev = *tables + (size_t)(j - 1) * *stride * 16;
        uint32_t* next = *tables + (size_t)j * *stride * 16;
        hipLaunchKernelGGL(k_table_next, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, prev, next, cnt, c);
        if (!ctx->has_gamma) PLK_HIP(hipMemsetAsync(next + 16 * *len, 0, 3 * 64, ctx->stream));
    }
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}
static int32_t srs_build_tables(plk_ctx* ctx) {
    const uint64_t cnt = ctx->has_gamma ? ctx->shard_len + 3 : ctx->shard_len;   // a cyclic shard of a short SRS may hold no point at all
    for (int j = 1; j < MsmPlan<MSM_C_LONG>::WINDOWS && cnt > 0; ++j) {
        const uint32_t* prev = ctx->tables + (size_t)(j - 1) * ctx->table_stride * 16;
        uint32_t* next = ctx->tables + (size_t)j * ctx->table_stride * 16;
        hipLaunchKernelGGL(k_table_next, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, prev, next, cnt, MSM_C_LONG);
    }
    PLK_HIP(hipGetLastError());
    ctx->short_max = short_max_setting();
    ctx->mid_max = mid_max_setting();
    if (ctx->short_max > 0 && ctx->shard_len > 0)
        PLK_TRY(build_prefix_tables(ctx, MSM_C_SHORT, MsmPlan<MSM_C_SHORT>::WINDOWS, ctx->short_max, &ctx->tables_short, &ctx->short_len, &ctx->short_stride));
    if (ctx->mid_max > ctx->short_max && ctx->shard_len > ctx->short_len)   // no point of this rank beyond the narrow prefix: nothing for the 16-bit plan to do
        PLK_TRY(build_prefix_tables(ctx, MSM_C, MsmPlan<MSM_C>::WINDOWS, ctx->mid_max, &ctx->tables_mid, &ctx->mid_len, &ctx->mid_stride));
    PLK_HIP(hipStreamSynchronize(ctx->stream));
    return slots_reserve(ctx);
}

}  // namespace plk

using namespace plk;

extern "C" int32_t plk_srs_load_shard(plk_ctx* ctx, const uint64_t* g_xy_host, uint64_t shard_first, uint64_t shard_len, uint64_t total_len,
                                      const uint64_t* gamma_g_xy_host) {
    if (!ctx || !g_xy_host) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    PLK_TRY(srs_alloc(ctx, shard_first, shard_len, total_len));
    if (ctx->has_gamma && !gamma_g_xy_host) { ctx->err = "the shard holding index 0 needs the gamma powers"; return PLK_ERR_INVALID_ARG; }
    const uint64_t cnt = ctx->has_gamma ? shard_len + 3 : shard_len;
    DevBuf staging_buf;
    PLK_HIP(staging_buf.alloc(cnt * 64));
    uint32_t* staging = (uint32_t*)staging_buf.p;
    PLK_HIP(hipMemcpyAsync(staging, g_xy_host, shard_len * 64, hipMemcpyHostToDevice, ctx->stream));
    if (ctx->has_gamma) PLK_HIP(hipMemcpyAsync(staging + 16 * shard_len, gamma_g_xy_host, 3 * 64, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_fq_to_rp, dim3((unsigned)((cnt * 2 + 255) / 256)), dim3(256), 0, ctx->stream, staging, ctx->tables, cnt * 2);
    PLK_HIP(hipStreamSynchronize(ctx->stream));
    return srs_build_tables(ctx);
}
extern "C" int32_t plk_srs_load(plk_ctx* ctx, const uint64_t* g_xy_host, uint64_t len, const uint64_t* gamma_g_xy_host) {
    return plk_srs_load_shard(ctx, g_xy_host, 0, len, len, gamma_g_xy_host);
}

// gt[w][d] = d * 2^(8w) * base for 32 windows of 8 bits: affine, x*2^256 form, on the host
static void fixed_base_table(const uint64_t base_xy[8], std::vector<uint64_t>& gt) {
    using namespace host;
    gt.assign((size_t)32 * 256 * 8, 0);
    pt base = pt_from_affine(base_xy);
    for (int w = 0; w < 32; ++w) {
        pt acc = pt_inf();
        for (int d = 1; d < 256; ++d) {
            acc = pt_add(acc, base);
            int32_t inf;
            pt_to_affine(acc, &gt[((size_t)w * 256 + d) * 8], &inf);
        }
        for (int k = 0; k < 8; ++k) base = pt_dbl(base);
    }
}
static bool g1_on_curve(const uint64_t xy[8]) {
    using namespace host;
    const el x = el_load(xy), y = el_load(xy + 4);
    if (geq(x.v, FQ.p) || geq(y.v, FQ.p)) return false;
    return eq(mul(FQ, y, y), add(FQ, mul(FQ, mul(FQ, x, x), x), from_u64(FQ, 3)));
}
// fills the tables of the shard that srs_alloc / srs_alloc_cyclic described: powers_of_g[i] = tau^i g and
// powers_of_gamma_g[j] = tau^j gamma_g.  g_xy == nullptr: g is the generator (1, 2); gamma_g_xy == nullptr: gamma_g = gscal * g.
static int32_t srs_generate_points(plk_ctx* ctx, const uint64_t tau[4], const uint64_t gscal[4], const uint64_t* g_xy = nullptr,
                                   const uint64_t* gamma_g_xy = nullptr) {
    const uint64_t shard_first = ctx->shard_first, shard_len = ctx->shard_len;
    using namespace host;
    uint64_t gen[8];
    if (g_xy) memcpy(gen, g_xy, 64);
    else {
        const uint64_t gx[4] = {1, 0, 0, 0}, gy[4] = {2, 0, 0, 0};
        el x = from_canonical(FQ, gx), y = from_canonical(FQ, gy);
        memcpy(gen, x.v, 32); memcpy(gen + 4, y.v, 32);
    }
    std::vector<uint64_t> gt;
    fixed_base_table(gen, gt);
    std::vector<uint64_t> tp(40 * 4);
    {
        el t = el_load(tau);
        for (int k = 0; k < 40; ++k) { memcpy(&tp[4 * k], t.v, 32); t = mul(FR, t, t); }
    }
    DevBuf gt_buf, gt_rp_buf, tp_buf, sc_buf;   // released on every path out of this function
    PLK_HIP(gt_buf.alloc(gt.size() * 8));
    PLK_HIP(gt_rp_buf.alloc(gt.size() * 8));
    PLK_HIP(tp_buf.alloc(tp.size() * 8));
    PLK_HIP(sc_buf.alloc(64));
    uint32_t *gt_dev = (uint32_t*)gt_buf.p, *gt_rp = (uint32_t*)gt_rp_buf.p, *tp_dev = (uint32_t*)tp_buf.p, *sc_dev = (uint32_t*)sc_buf.p;
    PLK_HIP(hipMemcpy(gt_dev, gt.data(), gt.size() * 8, hipMemcpyHostToDevice));
    PLK_HIP(hipMemcpy(tp_dev, tp.data(), tp.size() * 8, hipMemcpyHostToDevice));
    uint64_t sc[8];
    memcpy(sc, FR.one, 32); memcpy(sc + 4, gamma_g_xy ? FR.one : gscal, 32);
    PLK_HIP(hipMemcpy(sc_dev, sc, 64, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_fq_to_rp, dim3(32 * 256 * 2 / 256), dim3(256), 0, ctx->stream, gt_dev, gt_rp, (uint64_t)32 * 256 * 2);
    if (shard_len)
        hipLaunchKernelGGL(k_srs_generate, dim3((unsigned)((shard_len + 255) / 256)), dim3(256), 0, ctx->stream, gt_rp, tp_dev, sc_dev, shard_first, shard_len, ctx->tables,
                           ctx->cyc_world, ctx->cyc_rank, ctx->cyc_log_block);
    if (ctx->has_gamma) {
        if (gamma_g_xy) {   // the gamma powers have a base of their own: swap its table in behind the launch above
            fixed_base_table(gamma_g_xy, gt);
            PLK_HIP(hipMemcpyAsync(gt_dev, gt.data(), gt.size() * 8, hipMemcpyHostToDevice, ctx->stream));
            hipLaunchKernelGGL(k_fq_to_rp, dim3(32 * 256 * 2 / 256), dim3(256), 0, ctx->stream, gt_dev, gt_rp, (uint64_t)32 * 256 * 2);
        }
        hipLaunchKernelGGL(k_srs_generate, dim3(1), dim3(256), 0, ctx->stream, gt_rp, tp_dev, sc_dev + 8, (uint64_t)0, (uint64_t)3, ctx->tables + 16 * shard_len, 1u, 0u, 0u);
    }
    PLK_HIP(hipGetLastError());
    PLK_HIP(hipStreamSynchronize(ctx->stream));
    return srs_build_tables(ctx);
}
extern "C" int32_t plk_srs_generate_shard(plk_ctx* ctx, const uint64_t tau[4], const uint64_t gscal[4], uint64_t shard_first, uint64_t shard_len,
                                          uint64_t total_len) {
    if (!ctx || !tau || !gscal) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    PLK_TRY(srs_alloc(ctx, shard_first, shard_len, total_len));
    return srs_generate_points(ctx, tau, gscal);
}
extern "C" int32_t plk_srs_generate_cyclic(plk_ctx* ctx, const uint64_t tau[4], const uint64_t gscal[4], uint32_t rank, uint32_t world, uint32_t log_block,
                                           uint64_t total_len) {
    if (!ctx || !tau || !gscal) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    PLK_TRY(srs_alloc_cyclic(ctx, rank, world, log_block, total_len));
    return srs_generate_points(ctx, tau, gscal);
}
extern "C" int32_t plk_srs_setup_cyclic(plk_ctx* ctx, const uint64_t beta[4], const uint64_t g_xy[8], const uint64_t gamma_g_xy[8], uint32_t rank, uint32_t world,
                                        uint32_t log_block, uint64_t total_len) {
    if (!ctx || !beta || !g_xy || !gamma_g_xy) return PLK_ERR_INVALID_ARG;
    if (!g1_on_curve(g_xy) || !g1_on_curve(gamma_g_xy)) { ctx->err = "plk_srs_setup: g or gamma_g is not a point of the curve"; return PLK_ERR_INVALID_ARG; }
    PLK_HIP(hipSetDevice(ctx->device));
    if (world <= 1) PLK_TRY(srs_alloc(ctx, 0, total_len, total_len));
    else PLK_TRY(srs_alloc_cyclic(ctx, rank, world, log_block, total_len));
    return srs_generate_points(ctx, beta, nullptr, g_xy, gamma_g_xy);
}
extern "C" int32_t plk_srs_setup(plk_ctx* ctx, const uint64_t beta[4], const uint64_t g_xy[8], const uint64_t gamma_g_xy[8], uint64_t len) {
    return plk_srs_setup_cyclic(ctx, beta, g_xy, gamma_g_xy, 0, 1, 0, len);
}
extern "C" int32_t plk_srs_load_cyclic(plk_ctx* ctx, const uint64_t* g_xy_host, uint64_t total_len, const uint64_t* gamma_g_xy_host, uint32_t rank, uint32_t world,
                                       uint32_t log_block) {
    if (!ctx || !g_xy_host) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    if (world <= 1) return plk_srs_load_shard(ctx, g_xy_host, 0, total_len, total_len, gamma_g_xy_host);
    PLK_TRY(srs_alloc_cyclic(ctx, rank, world, log_block, total_len));
    if (ctx->has_gamma && !gamma_g_xy_host) { ctx->err = "rank 0 needs the gamma powers"; return PLK_ERR_INVALID_ARG; }
    const uint64_t cnt = ctx->shard_len + (ctx->has_gamma ? 3 : 0);
    DevBuf staging_buf;
    PLK_HIP(staging_buf.alloc(cnt * 64));
    uint32_t* staging = (uint32_t*)staging_buf.p;
    const uint64_t B = 1ull << log_block;
    uint64_t local = 0;
    for (uint64_t blk = rank; (blk << log_block) < total_len; blk += world) {   // this rank's blocks, in index order
        const uint64_t first = blk << log_block, len = std::min(B, total_len - first);
        PLK_HIP(hipMemcpyAsync(staging + 16 * local, g_xy_host + 8 * first, len * 64, hipMemcpyHostToDevice, ctx->stream));
        local += len;
    }
    if (ctx->has_gamma) PLK_HIP(hipMemcpyAsync(staging + 16 * ctx->shard_len, gamma_g_xy_host, 3 * 64, hipMemcpyHostToDevice, ctx->stream));
    if (cnt) hipLaunchKernelGGL(k_fq_to_rp, dim3((unsigned)((cnt * 2 + 255) / 256)), dim3(256), 0, ctx->stream, staging, ctx->tables, cnt * 2);
    PLK_HIP(hipStreamSynchronize(ctx->stream));
    return srs_build_tables(ctx);
}
extern "C" int32_t plk_srs_generate(plk_ctx* ctx, const uint64_t tau[4], const uint64_t gscal[4], uint64_t len) {
    return plk_srs_generate_shard(ctx, tau, gscal, 0, len, len);
}

extern "C" int32_t plk_srs_read(plk_ctx* ctx, int32_t which, uint64_t first, uint64_t count, uint64_t* out_xy_host) {
    if (!ctx || !ctx->tables || !out_xy_host) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    uint64_t off;
    if (which == 0 && ctx->cyc_world > 1) {
        // a range inside one block that this rank holds
        const uint64_t B = 1ull << ctx->cyc_log_block;
        if (count == 0 || (first >> ctx->cyc_log_block) % ctx->cyc_world != ctx->cyc_rank || (first & (B - 1)) + count > B || first + count > ctx->srs_total) {
            ctx->err = "range outside one block of this cyclic shard"; return PLK_ERR_INVALID_ARG;
        }
        off = cyclic_owned_below(ctx->cyc_world, ctx->cyc_rank, ctx->cyc_log_block, first);
    } else if (which == 0) {
        if (first < ctx->shard_first || first + count > ctx->shard_first + ctx->shard_len) { ctx->err = "range outside this shard"; return PLK_ERR_INVALID_ARG; }
        off = first - ctx->shard_first;
    } else {
        if (!ctx->has_gamma || first + count > 3) { ctx->err = "gamma powers not on this shard"; return PLK_ERR_INVALID_ARG; }
        off = ctx->shard_len + first;
    }
    if (count == 0) return PLK_OK;
    DevBuf tmp_buf;
    PLK_HIP(tmp_buf.alloc(count * 64));
    uint32_t* tmp = (uint32_t*)tmp_buf.p;
    hipLaunchKernelGGL(k_fq_to_r, dim3((unsigned)((count * 2 + 255) / 256)), dim3(256), 0, ctx->stream, ctx->tables + 16 * off, tmp, count * 2);
    PLK_HIP(hipGetLastError());
    PLK_HIP(hipMemcpyAsync(out_xy_host, tmp, count * 64, hipMemcpyDeviceToHost, ctx->stream));
    PLK_HIP(hipStreamSynchronize(ctx->stream));
    return PLK_OK;
}

// ---- MSM entry points ------------------------------------------------------------------------------
extern "C" int32_t plk_msm_partial_dev(plk_ctx* ctx, const void* scalars_dev, uint64_t k, uint64_t base_offset, uint64_t out_xyzz[16]) {
    if (!ctx || !out_xyzz || (k && !scalars_dev)) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    PLK_HIP(hipStreamSynchronize(ctx->stream));   // scalars may have been produced on the compute stream
    // A long MSM is cut into point ranges that go through the pipeline like independent MSMs: the sort of range i + 1 and the
    // reduction of range i - 1 run beside the accumulation of range i, instead of three stages one after the other.
    constexpr uint64_t RANGE = 1ull << 20;
    const uint64_t nranges = k <= RANGE + RANGE / 2 ? 1 : std::min<uint64_t>((k + RANGE - 1) / RANGE, MSM_SLOTS);
    const uint64_t per = (k + nranges - 1) / nranges;
    int slots[MSM_SLOTS];
    uint64_t used = 0;
    int32_t rc = PLK_OK;
    for (uint64_t i = 0; i < nranges && rc == PLK_OK; ++i) {
        const uint64_t first = i * per, len = std::min(per, k - first);
        rc = msm_acquire_slot(ctx, &slots[used]);
        if (rc != PLK_OK) break;
        MsmSlot& s = ctx->slot[slots[used]];
        s.busy = true;
        ++used;
        rc = msm_enqueue(ctx, s, (const uint32_t*)scalars_dev + 8 * first, len, base_offset + first, 0);
    }
    host::pt total = host::pt_inf();
    for (uint64_t i = 0; i < used; ++i) {   // every slot that was handed out is waited for, also after an error
        MsmSlot& s = ctx->slot[slots[i]];
        host::pt r;
        const int32_t rc_wait = msm_wait(ctx, s, &r);
        s.busy = false;
        if (rc == PLK_OK) rc = rc_wait;
        if (rc == PLK_OK) total = host::pt_add(total, r);
    }
    PLK_TRY(rc);
    prof_collect(ctx, false);
    host::pt_store(total, out_xyzz);
    return PLK_OK;
}
extern "C" int32_t plk_msm_plan_query(plk_ctx* ctx, uint64_t k, uint64_t base_offset, int32_t* window_bits, int32_t* windows, int32_t* buckets) {
    if (!ctx || !ctx->tables || !window_bits || !windows || !buckets) return PLK_ERR_INVALID_ARG;
    const PlanView v = plan_view(ctx, msm_plan_for(ctx, k, base_offset));
    *window_bits = v.c; *windows = v.windows; *buckets = v.buckets;
    return PLK_OK;
}
extern "C" int32_t plk_g1_sum_partials(const uint64_t* xyzz, uint64_t count, uint64_t out_xy[8], int32_t* out_is_inf) {
    if (!out_xy || !out_is_inf || (count && !xyzz)) return PLK_ERR_INVALID_ARG;
    host::pt acc = host::pt_inf();
    for (uint64_t i = 0; i < count; ++i) acc = host::pt_add(acc, host::pt_load(xyzz + 16 * i));
    host::pt_to_affine(acc, out_xy, out_is_inf);
    return PLK_OK;
}
extern "C" int32_t plk_msm_dev(plk_ctx* ctx, const void* scalars_dev, uint64_t k, uint64_t base_offset, uint64_t out_xy[8], int32_t* out_is_inf) {
    uint64_t part[16];
    PLK_TRY(plk_msm_partial_dev(ctx, scalars_dev, k, base_offset, part));
    return plk_g1_sum_partials(part, 1, out_xy, out_is_inf);
}
extern "C" int32_t plk_msm(plk_ctx* ctx, const uint64_t* scalars_host, uint64_t k, uint64_t base_offset, uint64_t out_xy[8], int32_t* out_is_inf) {
    if (!ctx || (k && !scalars_host)) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    DevBuf d;
    if (k) PLK_HIP(d.alloc_copy(scalars_host, k * 32));
    return plk_msm_dev(ctx, d.p, k, base_offset, out_xy, out_is_inf);
}
