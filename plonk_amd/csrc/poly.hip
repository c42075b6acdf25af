// Pointwise, scan and reduction kernels of the prover rounds (all over BN254 Fr):
//   grand product z (src/circuit.rs:491-544), gate + copy-constraint values for t (src/circuit.rs:343-379, 446-453),
//   suffix-Horner scan = evaluation at zeta and synthetic division by (X - zeta) in one sweep
//   (DensePolynomial::evaluate, src/prover.rs:266-274, and the witness polynomial of kzg10::open).
// Elements in memory are x*2^256 (8 words); products against library constants in the 2^261 form keep that form.
#include "common.cuh"
#include "host_math.hpp"
#include "kernels.h"
#include <algorithm>
#include <mutex>

namespace plk {

struct W8 {
    uint32_t w[8];
};
static W8 w8_of(const uint64_t* limbs) { W8 r; memcpy(r.w, limbs, 32); return r; }
static W8 w8_of(const host::el& e) { return w8_of(e.v); }
__device__ __forceinline__ fe fe_of(const W8& a) { return unpack(a.w); }
// x*2^256 -> x*2^261
__device__ __forceinline__ fe to_rp(const fe& a) { return mul<FrP>(a, fe_const(FrP::TO_RP)); }
// normalized, bound 2 -> packed without the final subtraction (value < 2^255)
__device__ __forceinline__ void st_lazy(uint32_t* p, const fe& a) {
    uint32_t w[8];
    pack(a, w);
    store8(p, w);
}

// ------------------------------------------------------------------------------------------ element-wise (tests, microbenchmarks)
template <class F>
__global__ void __launch_bounds__(256) k_field_op(int op, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint32_t* __restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const fe x = ld_fe(a + 8 * i), y = ld_fe(b + 8 * i);
    fe r;
    if (op == 0) { r = add(x, y); norm(r); r = mul<F>(r, fe_const(F::ONE)); }
    else if (op == 1) { r = sub2<F>(x, y); norm(r); r = mul<F>(r, fe_const(F::ONE)); }
    else r = mul<F>(mul<F>(x, fe_const(F::TO_RP)), y);
    st_fe<F>(out + 8 * i, r);
}

// ------------------------------------------------------------------------------------------ suffix Horner scan
// Loads groups of four consecutive elements (one 128-byte line each) per lane, so that the per-lane walk over
// consecutive coefficients still moves whole lines.
__device__ __forceinline__ void load_quad(const uint32_t* __restrict__ c, uint64_t i0, uint64_t len, fe out[4]) {
    if (i0 + 4 <= len) {
        uint4 r[8];
        const uint4* p = reinterpret_cast<const uint4*>(c + 8 * i0);
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = p[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t w[8] = {r[2 * k].x, r[2 * k].y, r[2 * k].z, r[2 * k].w, r[2 * k + 1].x, r[2 * k + 1].y, r[2 * k + 1].z, r[2 * k + 1].w};
            out[k] = unpack(w);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k] = i0 + k < len ? ld_fe(c + 8 * (i0 + k)) : fe_zero();
    }
}

// Three kernels whatever the length.  A workgroup of 256 threads owns E = 256 L consecutive coefficients, a thread L = 2^LOG_L:
//   k_div_sums    A_b = sum_{i in block b} c_i z^(i - b E)                      (thread-local Horner, weighted tree over the block)
//   k_div_carries C_b = A_b + z^E C_{b+1}: suffix scan over the blocks by one workgroup -> carry into every block, and h_0
//   k_div_apply   thread-local Horner again, weighted suffix scan inside the block seeded with the block's carry, then the walk
//                 down the thread's L coefficients that writes scale * h_i
// (the coefficients are read three times, twice of them from L2; a scan with one kernel per level of a 32-ary tree took nine
// launches at 2^20 coefficients, most of them a few microseconds of work behind 25 microseconds of launch latency).
// L = 8 below 2^21 coefficients, 32 from there on.  A thread's Horner chain is L dependent products and a polynomial of n
// coefficients gives n / (256 L) workgroups: with L = 32 a polynomial of 2^16 coefficients is eight workgroups on 256 CUs whose
// every thread walks through 32 dependent products, and the division takes as long as at 2^20 coefficients (0.18 ms); measured
// alone, 2^20 coefficients take 0.17 ms with L = 8 and 0.22 ms with L = 32, 2^22 coefficients 0.41 against 0.35 ms.
constexpr int DIV_THREADS = 256;
constexpr int DIV_LOG_L_SHORT = 3, DIV_LOG_L_LONG = 5;
#ifndef PLK_DIV_LONG_FROM_LOG
#define PLK_DIV_LONG_FROM_LOG 21
#endif
constexpr uint64_t DIV_LONG_FROM = 1ull << PLK_DIV_LONG_FROM_LOG;
static int div_log_l(uint64_t len) { return len < DIV_LONG_FROM ? DIV_LOG_L_SHORT : DIV_LOG_L_LONG; }
static uint64_t div_block(uint64_t len) { return (uint64_t)DIV_THREADS << div_log_l(len); }

// S = sum_{k < L} c_{lo + k} z^k; coefficients past len count as zero.  Result bound 3, limbs < 2^30.
template <int LOG_L>
__device__ __forceinline__ fe div_local_sum(const uint32_t* __restrict__ c, uint64_t lo, uint64_t len, const fe& z) {
    fe h = fe_zero();
    if (lo >= len) return h;
#pragma unroll 1
    for (int q = (1 << LOG_L) / 4 - 1; q >= 0; --q) {
        fe v[4];
        load_quad(c, lo + 4 * q, len, v);
#pragma unroll
        for (int k = 3; k >= 0; --k) h = add(v[k], mul<FrP>(h, z));
    }
    return h;
}
// V_t <- sum_{u >= t} V_u w^(u - t) over the 256 threads of the workgroup (Kogge-Stone, eight steps); w in the 2^261 form.
// Every thread passes a normalized value; the result is normalized with bound 2.  The weight of step d is w^(2^d): squared on the
// way (weight_of(d) == nullptr) or read from a table the host prepared (the powers z^(2^k) of the evaluation point: every thread of
// every workgroup would otherwise square the same eight values, half of the products of a scan).
struct DivPowers {
    W8 p[14];              // z^(2^k), k = 0 .. 13, in the 2^261 form (canonical words)
};
template <bool TABLE>
__device__ __forceinline__ fe div_block_suffix_scan(fe v, fe w, const DivPowers* table, int first_power, uint32_t* lds /* 2 x 256 x 9 words */) {
    const uint32_t t = threadIdx.x;
    uint32_t* cur = lds;
    uint32_t* nxt = lds + DIV_THREADS * 9;
#pragma unroll
    for (int i = 0; i < 9; ++i) cur[t * 9 + i] = v.v[i];
    __syncthreads();
    int step = 0;
    for (int d = 1; d < DIV_THREADS; d <<= 1, ++step) {
        if (TABLE) w = fe_of(table->p[first_power + step]);
        if (t + d < DIV_THREADS) {
            fe o;
#pragma unroll
            for (int i = 0; i < 9; ++i) o.v[i] = cur[(t + d) * 9 + i];
            v = add(v, mul<FrP>(o, w));
            norm(v);
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) nxt[t * 9 + i] = v.v[i];
        __syncthreads();
        uint32_t* sw = cur; cur = nxt; nxt = sw;
        if (!TABLE) w = mul<FrP>(w, w);
    }
    return mul<FrP>(v, fe_const(FrP::ONE));   // bound <= 3 + 2 * 8 back to 2
}
__device__ __forceinline__ fe div_pow2(fe x, int squarings) {
    for (int i = 0; i < squarings; ++i) x = mul<FrP>(x, x);
    return x;
}

// The three kernels take a batch: blockIdx.y (plus member0) selects the polynomial, so that the nine openings of a round
// (src/prover.rs:295-303: all nine polynomials are known once zeta is) cost three to five launches instead of twenty-seven.
struct DivBatch {
    const uint32_t* c[DIV_BATCH_MAX];
    uint64_t len[DIV_BATCH_MAX];
    uint32_t* A[DIV_BATCH_MAX];            // block sums
    uint32_t* carry[DIV_BATCH_MAX];
    uint32_t* q_out[DIV_BATCH_MAX];
    uint32_t* h0_out[DIV_BATCH_MAX];
    uint32_t nb[DIV_BATCH_MAX];            // blocks of the member
    uint32_t first_block[DIV_BATCH_MAX];   // k_div_apply walks blocks first_block .. first_block + apply_blocks - 1
    uint32_t apply_blocks[DIV_BATCH_MAX];
    uint32_t log_block[DIV_BATCH_MAX];     // log2 of the coefficients per block
    W8 zeta, scale;
    DivPowers zp;
    uint32_t cyc_world, cyc_rank, cyc_log_block;
};
template <int LOG_L>
__global__ void __launch_bounds__(DIV_THREADS) k_div_sums(DivBatch db, uint32_t member0) {
    PLK_SIDE_PRIO();
    __shared__ uint32_t lds[2 * DIV_THREADS * 9];
    const uint32_t mb = member0 + blockIdx.y;
    if (blockIdx.x >= db.nb[mb]) return;
    const uint32_t* __restrict__ c = db.c[mb];
    const uint64_t len = db.len[mb];
    const fe z = fe_of(db.zp.p[0]);
    fe s = div_local_sum<LOG_L>(c, ((uint64_t)blockIdx.x * DIV_THREADS + threadIdx.x) << LOG_L, len, z);
    norm(s);
    const fe tot = div_block_suffix_scan<true>(s, z, &db.zp, LOG_L, lds);   // thread 0 ends up with the whole block
    if (threadIdx.x == 0) st_lazy(db.A[mb] + 8ull * blockIdx.x, tot);
}
// carry[b] = h at the first coefficient of block b + 1 (zero for the last block); h0 = h_0, canonical.  One workgroup per polynomial.
__global__ void __launch_bounds__(DIV_THREADS) k_div_carries(DivBatch db) {
    PLK_SIDE_PRIO();
    __shared__ uint32_t lds[2 * DIV_THREADS * 9];
    __shared__ uint32_t tops[DIV_THREADS * 9];
    const uint32_t mb = blockIdx.x;
    const uint32_t* __restrict__ A = db.A[mb];
    uint32_t* __restrict__ carry = db.carry[mb];
    uint32_t* __restrict__ h0_out = db.h0_out[mb];
    const uint32_t nb = db.nb[mb];
    const uint32_t t = threadIdx.x;
    const uint32_t m = (nb + DIV_THREADS - 1) / DIV_THREADS;            // blocks per thread
    const fe ze = fe_of(db.zp.p[db.log_block[mb]]);                      // z^E, E = 2^11 or 2^13
    // weight between the chunks of neighbouring threads: ze^m
    fe wm = fe_const(FrP::ONE);
    for (uint32_t k = 0; k < m; ++k) wm = mul<FrP>(wm, ze);
    const uint32_t lo = t * m;
    fe s = fe_zero();
    for (uint32_t j = m; j-- > 0;) {
        const uint32_t b = lo + j;
        const fe a = b < nb ? ld_fe(A + 8ull * b) : fe_zero();
        s = add(a, mul<FrP>(s, ze));
    }
    norm(s);
    const fe head = div_block_suffix_scan<false>(s, wm, nullptr, 0, lds);   // C at the first block of this thread's chunk
#pragma unroll
    for (int i = 0; i < 9; ++i) tops[t * 9 + i] = head.v[i];
    __syncthreads();
    fe h = fe_zero();                                                   // C at the first block of the next thread's chunk
    if (t + 1 < DIV_THREADS) {
#pragma unroll
        for (int i = 0; i < 9; ++i) h.v[i] = tops[(t + 1) * 9 + i];
    }
    for (uint32_t j = m; j-- > 0;) {
        const uint32_t b = lo + j;
        if (b >= nb) continue;
        st_lazy(carry + 8ull * b, mul<FrP>(h, fe_const(FrP::ONE)));
        h = add(ld_fe(A + 8ull * b), mul<FrP>(h, ze));
        norm(h);
    }
    if (t == 0 && h0_out) st_fe<FrP>(h0_out, mul<FrP>(h, fe_const(FrP::ONE)));
}
// q_out[i - 1] = scale * h_i for i >= 1
template <int LOG_L>
__global__ void __launch_bounds__(DIV_THREADS) k_div_apply(DivBatch db, uint32_t member0) {
    PLK_SIDE_PRIO();
    constexpr uint64_t BLOCK = (uint64_t)DIV_THREADS << LOG_L;
    __shared__ uint32_t lds[2 * DIV_THREADS * 9];
    __shared__ uint32_t heads[DIV_THREADS * 9];
    const uint32_t mb = member0 + blockIdx.y;
    if (blockIdx.x >= ((db.apply_blocks[mb] + 7) & ~7u)) return;
    const uint32_t* __restrict__ c = db.c[mb];
    const uint32_t* __restrict__ carry = db.carry[mb];
    uint32_t* __restrict__ q_out = db.q_out[mb];
    const uint64_t len = db.len[mb];
    const uint32_t cyc_world = db.cyc_world, cyc_rank = db.cyc_rank, cyc_log_block = db.cyc_log_block;
    // Consecutive workgroups go to different XCDs (eight of them), and on a block-cyclic shard only every world-th stretch of blocks
    // has work: taken in order, the working blocks of an eight-rank shard all land on one or two XCDs.  The grid is therefore walked
    // column by column: workgroup i takes block (i mod 8) * ceil(blocks / 8) + i / 8.
    // (Only there: with work in every block the plain order is the better one, neighbouring workgroups read neighbouring memory.)
    const uint32_t per_xcd = (db.apply_blocks[mb] + 7) >> 3;
    const uint32_t rel = db.cyc_world > 1 ? (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3) : blockIdx.x;
    if (rel >= db.apply_blocks[mb]) return;
    const uint32_t t = threadIdx.x, blk = db.first_block[mb] + rel;
    // A workgroup writes the quotient coefficients k in [blk BLOCK, (blk + 1) BLOCK): q_k = scale h_(k+1), so the last one is the
    // block's carry (h at the first coefficient of the next block) and the first coefficient's own h belongs to the block before.
    // Aligned like this, the division blocks coincide with the blocks of a block-cyclic SRS shard of the same or a larger size.
    if (cyc_world > 1) {
        // block-cyclic shard: q_out[k] is only used for the SRS points k this rank holds (block k >> log_block congruent to rank);
        // a workgroup whose coefficients touch none of them has nothing to write
        const uint64_t k_lo = (uint64_t)blk * BLOCK, k_hi = (uint64_t)blk * BLOCK + BLOCK - 1;
        const uint64_t b_lo = k_lo >> cyc_log_block, b_hi = k_hi >> cyc_log_block;
        bool need = b_hi - b_lo + 1 >= cyc_world;
        for (uint64_t b = b_lo; !need && b <= b_hi; ++b) need = b % cyc_world == cyc_rank;
        if (!need) return;
    }
    const fe z = fe_of(db.zp.p[0]);
    const fe zl = fe_of(db.zp.p[LOG_L]);
    const uint64_t lo = (uint64_t)blk * BLOCK + ((uint64_t)t << LOG_L);
    const fe cin = ld_fe(carry + 8ull * blk);
    fe s = div_local_sum<LOG_L>(c, lo, len, z);
    if (t == DIV_THREADS - 1) s = add(s, mul<FrP>(cin, zl));            // h at this thread's first coefficient, given the block's carry
    norm(s);
    const fe head = div_block_suffix_scan<true>(s, zl, &db.zp, LOG_L, lds);   // h at every thread's first coefficient
#pragma unroll
    for (int i = 0; i < 9; ++i) heads[t * 9 + i] = head.v[i];
    __syncthreads();
    const fe sc = to_rp(fe_of(db.scale));
    if (t == DIV_THREADS - 1 && (uint64_t)(blk + 1) * BLOCK < len) st_fe<FrP>(q_out + 8 * ((uint64_t)(blk + 1) * BLOCK - 1), mul<FrP>(cin, sc));
    if (lo >= len) return;
    fe h = cin;                                                         // h just above this thread's chunk
    if (t + 1 < DIV_THREADS) {
#pragma unroll
        for (int i = 0; i < 9; ++i) h.v[i] = heads[(t + 1) * 9 + i];
    }
#pragma unroll 1
    for (int q = (1 << LOG_L) / 4 - 1; q >= 0; --q) {
        fe v[4];
        load_quad(c, lo + 4 * q, len, v);
#pragma unroll
        for (int k = 3; k >= 0; --k) {
            const uint64_t i = lo + 4 * q + k;
            if (i >= len) continue;
            h = add(v[k], mul<FrP>(h, z));      // bound 3, limbs < 2^30
            if (i > (uint64_t)blk * BLOCK) st_fe<FrP>(q_out + 8 * (i - 1), mul<FrP>(h, sc));   // the block's first h belongs to the block before
        }
    }
}

uint64_t division_work_words(uint64_t len) {
    const uint64_t block = div_block(len), nb = (len + block - 1) / block;
    return 2 * (nb + 1) * 8 + 64;
}
// Members are handed over in any order; the launches take the short-block ones first (blockIdx.y of the two templated kernels
// counts inside its class).  A member of length zero gets h_0 = 0 from the carry kernel and nothing else.
int32_t division_enqueue_batch(plk_ctx* ctx, hipStream_t st, uint32_t count, const DivJob* jobs, const uint64_t zeta[4], const uint64_t scale[4]) {
    if (count == 0) return PLK_OK;
    if (count > (uint32_t)DIV_BATCH_MAX) { ctx->err = "division batch too long"; return PLK_ERR_INVALID_ARG; }
    DivBatch db;
    memset(&db, 0, sizeof(db));
    db.zeta = w8_of(zeta);
    db.scale = w8_of(scale ? scale : host::FR.one);
    {
        // z^(2^k) in the 2^261 form: the host's elements carry 2^256, a product with 32 supplies the rest
        using namespace host;
        const el f32 = from_u64(FR, 32);
        el zk = el_load(zeta);
        for (int k = 0; k < 14; ++k) { db.zp.p[k] = w8_of(mul(FR, zk, f32)); zk = mul(FR, zk, zk); }
    }
    db.cyc_world = ctx->cyc_world; db.cyc_rank = ctx->cyc_rank; db.cyc_log_block = ctx->cyc_log_block;
    uint32_t n_short = 0, n = 0, max_nb[2] = {0, 0}, max_apply[2] = {0, 0};
    // A batch of several long polynomials fills the chip whatever the block size, and then the work per coefficient decides (the
    // scan over a block costs sixteen products per thread, whatever L): the long form from 2^19 coefficients on.
    static const int batch_long_log = getenv("PLK_DIV_BATCH_LONG_FROM_LOG") ? atoi(getenv("PLK_DIV_BATCH_LONG_FROM_LOG")) : 19;
    const uint64_t long_from = count >= 4 ? std::min<uint64_t>(DIV_LONG_FROM, 1ull << batch_long_log) : DIV_LONG_FROM;
    for (int cls = 0; cls < 2; ++cls) {
        for (uint32_t k = 0; k < count; ++k) {
            const DivJob& j = jobs[k];
            const bool is_short = j.len < long_from;
            if (is_short != (cls == 0)) continue;
            const int log_l = is_short ? DIV_LOG_L_SHORT : DIV_LOG_L_LONG;
            const uint64_t block = (uint64_t)DIV_THREADS << log_l;
            const uint32_t nb = (uint32_t)((j.len + block - 1) / block);
            db.c[n] = j.coeffs; db.len[n] = j.len; db.nb[n] = nb;
            db.A[n] = j.work; db.carry[n] = j.work + (size_t)(nb + 1) * 8;
            db.q_out[n] = j.q_out; db.h0_out[n] = j.h0_out;
            db.log_block[n] = 8 + (uint32_t)log_l;
            db.first_block[n] = 0; db.apply_blocks[n] = 0;
            if (j.q_out && j.len > 1 && j.q_count > 0) {
                // block b writes q_out[k] for k in [b block, (b + 1) block): only the blocks that hold k = q_first .. q_first + q_count - 1 are walked
                const uint64_t hi = std::min<uint64_t>(j.q_first + j.q_count, j.len - 1);
                if (j.q_first < hi) {
                    const uint32_t b0 = (uint32_t)(j.q_first / block), b1 = (uint32_t)((hi - 1) / block);
                    db.first_block[n] = b0; db.apply_blocks[n] = b1 - b0 + 1;
                }
            }
            max_nb[cls] = std::max(max_nb[cls], nb);
            max_apply[cls] = std::max(max_apply[cls], db.apply_blocks[n]);
            ++n;
        }
        if (cls == 0) n_short = n;
    }
    const uint32_t n_long = n - n_short;
    prof_begin(ctx, st, "division");
    if (n_short && max_nb[0]) hipLaunchKernelGGL(k_div_sums<DIV_LOG_L_SHORT>, dim3(max_nb[0], n_short), dim3(DIV_THREADS), 0, st, db, 0u);
    if (n_long && max_nb[1]) hipLaunchKernelGGL(k_div_sums<DIV_LOG_L_LONG>, dim3(max_nb[1], n_long), dim3(DIV_THREADS), 0, st, db, n_short);
    hipLaunchKernelGGL(k_div_carries, dim3(n), dim3(DIV_THREADS), 0, st, db);
    if (n_short && max_apply[0]) hipLaunchKernelGGL(k_div_apply<DIV_LOG_L_SHORT>, dim3((max_apply[0] + 7) & ~7u, n_short), dim3(DIV_THREADS), 0, st, db, 0u);
    if (n_long && max_apply[1]) hipLaunchKernelGGL(k_div_apply<DIV_LOG_L_LONG>, dim3((max_apply[1] + 7) & ~7u, n_long), dim3(DIV_THREADS), 0, st, db, n_short);
    prof_end(ctx, st);
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}
int32_t division_enqueue(plk_ctx* ctx, hipStream_t st, const uint32_t* coeffs, uint64_t len, const uint64_t zeta[4], const uint64_t scale[4],
                         uint32_t* q_out, uint32_t* h0_out, uint32_t* work, uint64_t q_first, uint64_t q_count) {
    DivJob j;
    j.coeffs = coeffs; j.len = len; j.q_out = q_out; j.h0_out = h0_out; j.work = work; j.q_first = q_first; j.q_count = q_count;
    return division_enqueue_batch(ctx, st, 1, &j, zeta, scale);
}

// ------------------------------------------------------------------------------------------ grand product
// term = w + beta * idx + gamma in the 2^261 form (bound 6, normalized).  beta2 = beta * 2^522, gamma1 = gamma * 2^261.
__device__ __forceinline__ fe perm_term(const fe& w_r, uint32_t idx, const fe& beta2, const fe& gamma1) {
    fe j = fe_zero();
    j.v[0] = idx;                                  // idx < 2^29
    fe t = add(add(to_rp(w_r), mul<FrP>(beta2, j)), gamma1);
    norm(t);
    return t;
}
// Inclusive multiplicative scan of 2^261-form values, 2048 per block (8 per thread); reverse scans run from the top index down.
// blockIdx.y selects one of several arrays scanned by the same launch (the grand product scans numerators and denominators together).
constexpr int SCAN_PER_THREAD = 8;
constexpr int SCAN_BLOCK = 256 * SCAN_PER_THREAD;
// element(i) yields the value at scan position i < n; the inclusive prefix of position i goes to out + 8 * (reverse ? n - 1 - i : i)
template <typename Element>
__device__ __forceinline__ void scan_block(Element element, uint32_t* __restrict__ out, uint64_t n, int reverse, uint32_t* __restrict__ block_tot) {
    __shared__ uint32_t wave_tot[4][9];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_BLOCK + (uint64_t)threadIdx.x * SCAN_PER_THREAD;
    fe p[SCAN_PER_THREAD];
    fe run = fe_const(FrP::ONE);
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; ++k) {
        const uint64_t i = base + k;
        if (i < n) run = mul<FrP>(run, element(i));
        p[k] = run;
    }
    fe incl = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        fe other;
#pragma unroll
        for (int q = 0; q < 9; ++q) other.v[q] = __shfl_up(incl.v[q], o);
        const fe prod = mul<FrP>(other, incl);
        if (lane >= (uint32_t)o) incl = prod;
    }
    if (lane == 63) {
#pragma unroll
        for (int q = 0; q < 9; ++q) wave_tot[wave][q] = incl.v[q];
    }
    fe excl;
#pragma unroll
    for (int q = 0; q < 9; ++q) excl.v[q] = __shfl_up(incl.v[q], 1);
    if (lane == 0) excl = fe_const(FrP::ONE);
    __syncthreads();
    for (uint32_t v = 0; v < wave; ++v) {
        fe wt;
#pragma unroll
        for (int q = 0; q < 9; ++q) wt.v[q] = wave_tot[v][q];
        excl = mul<FrP>(wt, excl);
    }
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; ++k) {
        const uint64_t i = base + k;
        if (i < n) st_lazy(out + 8 * (reverse ? n - 1 - i : i), mul<FrP>(excl, p[k]));
    }
    if (threadIdx.x == 255) st_lazy(block_tot + 8ull * blockIdx.x, mul<FrP>(excl, p[SCAN_PER_THREAD - 1]));
}
__global__ void __launch_bounds__(256) k_scan_local(uint32_t* __restrict__ data, uint64_t data_stride, uint64_t n, int reverse, uint32_t* __restrict__ block_tot,
                                                    uint64_t tot_stride) {
    uint32_t* d = data + blockIdx.y * data_stride;
    scan_block([&](uint64_t i) { return ld_fe(d + 8 * (reverse ? n - 1 - i : i)); }, d, n, reverse, block_tot + blockIdx.y * tot_stride);
}
__global__ void __launch_bounds__(256) k_scan_apply(uint32_t* __restrict__ data, uint64_t data_stride, uint64_t n, int reverse, const uint32_t* __restrict__ block_incl,
                                                    uint64_t tot_stride) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t b = i / SCAN_BLOCK;
    if (b == 0) return;
    uint32_t* p = data + blockIdx.y * data_stride + 8 * (reverse ? n - 1 - i : i);
    st_lazy(p, mul<FrP>(ld_fe(block_incl + blockIdx.y * tot_stride + 8 * (b - 1)), ld_fe(p)));
}
// in-place inclusive scans of `arrays` arrays of n elements, data_stride words apart; work must hold the block totals of every
// level (arrays * (n / 2048 + n / 2048^2 + ...) elements, rounded up per level)
static int32_t scan_mul(plk_ctx* ctx, hipStream_t st, uint32_t* data, uint64_t data_stride, uint32_t arrays, uint64_t n, bool reverse, uint32_t* work) {
    const uint64_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    hipLaunchKernelGGL(k_scan_local, dim3((unsigned)nb, arrays), dim3(256), 0, st, data, data_stride, n, reverse ? 1 : 0, work, 8 * nb);
    if (nb > 1) {
        PLK_TRY(scan_mul(ctx, st, work, 8 * nb, arrays, nb, false, work + (size_t)arrays * 8 * nb));
        hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)((n + 255) / 256), arrays), dim3(256), 0, st, data, data_stride, n, reverse ? 1 : 0, work, 8 * nb);
    }
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}

// Grand product, first pass.  blockIdx.y = 0: numerators of the rows in index order; 1: denominators from the top row down.  The
// terms are computed where they are needed (three per row) and go straight into the block's scan: prefix products inside the block
// -> num / den (indexed by row), the block's product -> tot.
__global__ void __launch_bounds__(256) k_gp_local(const uint32_t* __restrict__ wflat, const uint64_t* __restrict__ sigma, uint64_t n, W8 beta, W8 gamma,
                                                  uint32_t* __restrict__ num, uint32_t* __restrict__ den, uint32_t* __restrict__ tot, uint64_t tot_stride,
                                                  uint32_t* __restrict__ flag) {
    const fe beta2 = mul<FrP>(to_rp(fe_of(beta)), fe_const(FrP::R2)), gamma1 = to_rp(fe_of(gamma));
    if (blockIdx.y == 0) {
        scan_block([&](uint64_t i) {
            fe nu;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint64_t j = 3 * i + k;
                const fe tn = perm_term(ld_fe(wflat + 8 * j), (uint32_t)j, beta2, gamma1);
                nu = k == 0 ? tn : mul<FrP>(nu, tn);
            }
            return nu;
        }, num, n, 0, tot);
    } else {
        scan_block([&](uint64_t i) {
            const uint64_t row = n - 1 - i;
            fe de;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                uint64_t sj = sigma[3 * row + k];
                if (sj >= 3 * n) { atomicOr(flag, 2u); sj = 0; }
                const fe td = perm_term(ld_fe(wflat + 8 * sj), (uint32_t)sj, beta2, gamma1);
                de = k == 0 ? td : mul<FrP>(de, td);
            }
            return de;
        }, den, n, 1, tot + tot_stride);
    }
}
// z[0] = 1, z[i] = N_{i-1} * S_i / D for 0 < i < n, with N the prefix products of the numerators, S the suffix products of the
// denominators (each the block-local product from k_gp_local times the scanned product of the blocks before), dinv = D^-1 * 2^256.
__global__ void __launch_bounds__(256) k_gp_finish(const uint32_t* __restrict__ num_loc, const uint32_t* __restrict__ den_loc, const uint32_t* __restrict__ tot_incl,
                                                   uint64_t tot_stride, uint64_t n, W8 dinv, uint32_t* __restrict__ z) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (i == 0) { st_fe<FrP>(z, mul<FrP>(fe_const(FrP::ONE), fe_const(FrP::TO_R))); return; }
    const uint64_t bn = (i - 1) / SCAN_BLOCK, bd = (n - 1 - i) / SCAN_BLOCK;     // scan blocks of N_{i-1} and of S_i
    fe nu = ld_fe(num_loc + 8 * (i - 1)), de = ld_fe(den_loc + 8 * i);
    if (bn) nu = mul<FrP>(ld_fe(tot_incl + 8 * (bn - 1)), nu);
    if (bd) de = mul<FrP>(ld_fe(tot_incl + tot_stride + 8 * (bd - 1)), de);
    st_fe<FrP>(z + 8 * i, mul<FrP>(mul<FrP>(nu, de), fe_of(dinv)));
}

// ------------------------------------------------------------------------------------------ quotient numerator on H
// vals[i] = q_add (a + b - c) + q_mul (a b - c) + [ z_i prod(w + beta (3i+k) + gamma) - z_{i+1} prod(w + beta sigma(3i+k) + gamma) ]
// and in the last row  z_i prod / prod' - 1  (src/circuit.rs:370-379).
__global__ void __launch_bounds__(256) k_quotient_vals(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, const uint32_t* __restrict__ c,
                                                       const uint32_t* __restrict__ qa, const uint32_t* __restrict__ qm, const uint32_t* __restrict__ zv,
                                                       const uint64_t* __restrict__ sigma, uint64_t n, W8 beta, W8 gamma, uint32_t* __restrict__ vals,
                                                       uint32_t* __restrict__ flag) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const fe beta2 = mul<FrP>(to_rp(fe_of(beta)), fe_const(FrP::R2)), gamma1 = to_rp(fe_of(gamma));
    const fe w[3] = {ld_fe(a + 8 * i), ld_fe(b + 8 * i), ld_fe(c + 8 * i)};
    // gate
    fe s = add(w[0], w[1]);
    s = sub2<FrP>(s, w[2]); norm(s);                                   // a + b - c, bound 4
    fe ab = mul<FrP>(to_rp(w[0]), w[1]);
    fe m = sub2<FrP>(ab, w[2]); norm(m);                               // a b - c, bound 4
    fe gate = add(mul<FrP>(to_rp(ld_fe(qa + 8 * i)), s), mul<FrP>(to_rp(ld_fe(qm + 8 * i)), m));   // bound 4, limbs < 2^30
    // copy constraint
    fe lhs, rhs;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        uint64_t sj = sigma[3 * i + k];
        if (sj >= 3 * n) { atomicOr(flag, 2u); sj = 0; }
        const fe tl = perm_term(w[k], (uint32_t)(3 * i + k), beta2, gamma1);
        const fe tr = perm_term(w[k], (uint32_t)sj, beta2, gamma1);
        lhs = k == 0 ? tl : mul<FrP>(lhs, tl);
        rhs = k == 0 ? tr : mul<FrP>(rhs, tr);
    }
    const fe zi = ld_fe(zv + 8 * i);
    fe perm;
    if (i + 1 < n) {
        const fe l = mul<FrP>(lhs, zi), r = mul<FrP>(rhs, ld_fe(zv + 8 * (i + 1)));   // x*2^256 form
        perm = sub2<FrP>(l, r);                                        // bound 4
    } else {
        // z lhs / rhs - 1 = (z lhs - rhs) / rhs: zero exactly when the numerator is, so the inversion is only paid on bad witnesses
        const fe l = mul<FrP>(lhs, zi), r = mul<FrP>(rhs, fe_const(FrP::TO_R));
        fe d = sub2<FrP>(l, r); norm(d);
        // rhs = 0 is the reference's inverse().unwrap() panic (src/circuit.rs:377), whatever the numerator is
        if (is_zero_mod<FrP>(rhs)) { atomicOr(flag, 4u); perm = fe_zero(); }   // rhs: 2^261 form, bound 2
        else if (is_zero_mod<FrP>(d)) perm = fe_zero();
        else perm = mul<FrP>(inv<FrP>(rhs), d);                        // (1/rhs)*2^261 times d*2^256 -> x*2^256
    }
    fe v = add(gate, perm);                                            // bound 8
    norm(v);
    const fe out = reduce<FrP>(mul<FrP>(v, fe_const(FrP::ONE)));
    uint32_t wv[8];
    pack(out, wv);
    store8(vals + 8 * i, wv);
}
// Public-input term of src/circuit.rs:410-439 added to the numerator values: vals[row_k] += alpha * (a[row_k] - pi_k).
// The reference assigns constraint[row] per public input, so a later input replaces an earlier one on the same row: the host passes
// one entry per distinct row (the last one) and the additions do not collide.
__global__ void __launch_bounds__(256) k_public_input_term(const uint32_t* __restrict__ a, const uint64_t* __restrict__ rows, const uint32_t* __restrict__ pi,
                                                           uint64_t count, W8 alpha, uint32_t* __restrict__ vals) {
    const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    const uint64_t row = rows[k];
    fe d = sub2<FrP>(ld_fe(a + 8 * row), ld_fe(pi + 8 * k)); norm(d);             // bound 3
    fe v = add(ld_fe(vals + 8 * row), mul<FrP>(to_rp(fe_of(alpha)), d));          // bound 3
    norm(v);
    st_fe<FrP>(vals + 8 * row, mul<FrP>(v, fe_const(FrP::ONE)));
}
__global__ void __launch_bounds__(256) k_any_nonzero(const uint32_t* __restrict__ data, uint64_t n_words, uint32_t* __restrict__ flag) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t v = 0;
    for (uint64_t k = i; k < n_words; k += (uint64_t)gridDim.x * 256) v |= data[k];
    if (v) atomicOr(flag, 1u);
}

// ------------------------------------------------------------------------------------------ permutation polynomials
// vals_id[i] = w^i, vals_sigma[i] = w^(sigma_pad(i)) on the domain of size 2^log_n4 (src/poly_utils.rs:18-25)
__global__ void __launch_bounds__(256) k_perm_values(const uint64_t* __restrict__ sigma, uint64_t sigma_len, uint32_t log_n4, const uint32_t* __restrict__ W,
                                                     uint32_t* __restrict__ vals_id, uint32_t* __restrict__ vals_sigma, uint32_t* __restrict__ flag) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t n4 = 1ull << log_n4;
    if (i >= n4) return;
    // domain.element(s) is w^s for any s (src/poly_utils.rs:25): an entry beyond the domain wraps around, nothing panics
    const uint64_t s = (i < sigma_len ? sigma[i] : i) & (n4 - 1);
    const fe to_r = fe_const(FrP::TO_R);
    const uint64_t e[2] = {i, s};
    uint32_t* dst[2] = {vals_id, vals_sigma};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        uint64_t x = e[k];
        const bool neg = x >= (n4 >> 1);
        if (neg) x -= n4 >> 1;
        fe y = mul<FrP>(ld_fe(W + 8 * x), to_r);
        if (neg) { y = sub2<FrP>(fe_zero(), y); norm(y); }
        st_fe<FrP>(dst[k] + 8 * i, y);
    }
}

// ------------------------------------------------------------------------------------------ quotient on the extended coset
// Labels of the copy constraint as field elements (x*2^256): idx[i] = i, sg_k[i] = sigma(3i + k)   (src/prover.rs:205, src/circuit.rs:356-368)
__global__ void __launch_bounds__(256) k_coset_labels(const uint64_t* __restrict__ sigma, uint64_t n, uint32_t* __restrict__ idx, uint32_t* __restrict__ sg0,
                                                      uint32_t* __restrict__ sg1, uint32_t* __restrict__ sg2, uint32_t* __restrict__ flag) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const fe r2 = fe_const(FrP::R2), to_r = fe_const(FrP::TO_R);
    uint32_t* dst[4] = {idx, sg0, sg1, sg2};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint64_t v = k == 0 ? i : sigma[3 * i + (k - 1)];
        if (v >= 3 * n) { atomicOr(flag, 2u); v = 0; }
        fe j = fe_zero();
        j.v[0] = (uint32_t)v;                                          // < 2^29
        st_fe<FrP>(dst[k] + 8 * i, mul<FrP>(mul<FrP>(j, r2), to_r));   // j * 2^261, then j * 2^256
    }
}
struct CosetQuotientArgs {
    W8 beta, gamma, alpha;
    W8 zh_inv[4];   // 1 / (x^n - 1) at coset point i depends on i mod 4 only: x^n = g^n w_4^i
};
// One coset point per thread: all ten polynomials are read once (z twice: at x and at w x, four slots further on), one value is written.
//   out = [ q_add (a + b - c) + q_mul (a b - c) + alpha ( z prod_k (w_k + beta (3 I + k) + gamma) - z' prod_k (w_k + beta SG_k + gamma) ) ] / (x^n - 1)
__global__ void __launch_bounds__(256) k_quotient_coset(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, const uint32_t* __restrict__ c,
                                                        const uint32_t* __restrict__ qa, const uint32_t* __restrict__ qm, const uint32_t* __restrict__ z,
                                                        const uint32_t* __restrict__ idx, const uint32_t* __restrict__ sg0, const uint32_t* __restrict__ sg1,
                                                        const uint32_t* __restrict__ sg2, uint64_t n4, CosetQuotientArgs args, uint32_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const fe beta1 = to_rp(fe_of(args.beta));                          // beta * 2^261
    const fe gamma1 = to_rp(fe_of(args.gamma)), alpha1 = to_rp(fe_of(args.alpha));
    // data are x*2^256 and to_rp(x) = x*2^261; beta in the form beta * 2^266 turns x*2^256 into beta x * 2^261 with one product
    const fe beta_up = to_rp(beta1);
    const fe w[3] = {ld_fe(a + 8 * i), ld_fe(b + 8 * i), ld_fe(c + 8 * i)};
    const fe w1[3] = {to_rp(w[0]), to_rp(w[1]), to_rp(w[2])};
    // gate
    fe s = add(w[0], w[1]);
    s = sub2<FrP>(s, w[2]); norm(s);                                   // a + b - c, bound 4
    fe m = sub2<FrP>(mul<FrP>(w1[0], w[1]), w[2]); norm(m);            // a b - c, bound 4
    const fe gate = add(mul<FrP>(to_rp(ld_fe(qa + 8 * i)), s), mul<FrP>(to_rp(ld_fe(qm + 8 * i)), m));   // bound 4, limbs < 2^30
    // copy constraint: terms in the 2^261 form so that the running products keep z's 2^256 form
    const fe bi = mul<FrP>(beta_up, ld_fe(idx + 8 * i));               // beta I * 2^261, bound 2
    fe bi3 = add(add(bi, bi), bi);                                     // bound 6
    const uint32_t* sg[3] = {sg0, sg1, sg2};
    fe l = ld_fe(z + 8 * i), r = ld_fe(z + 8 * ((i + 4) & (n4 - 1)));
    fe kb = fe_zero();                                                 // k * beta * 2^261, bound k
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        fe tl = add(add(add(w1[k], bi3), kb), gamma1);                 // bound <= 2 + 6 + 2 + 2
        norm(tl);
        fe tr = add(add(w1[k], mul<FrP>(beta_up, ld_fe(sg[k] + 8 * i))), gamma1);   // bound <= 6
        norm(tr);
        l = mul<FrP>(l, tl);
        r = mul<FrP>(r, tr);
        kb = add(kb, beta1);
    }
    fe d = sub2<FrP>(l, r); norm(d);                                   // bound 4
    fe v = add(gate, mul<FrP>(alpha1, d));                             // bound 6
    norm(v);
    st_fe<FrP>(out + 8 * i, mul<FrP>(to_rp(fe_of(args.zh_inv[i & 3])), v));
}

}  // namespace plk

using namespace plk;

// ================================================================================================ entry points
extern "C" int32_t plk_field_op(plk_ctx* ctx, int32_t field, int32_t op, const uint64_t* a_host, const uint64_t* b_host, uint64_t* out_host, uint64_t n) {
    if (!ctx || !a_host || !b_host || !out_host || op < 0 || op > 2) return PLK_ERR_INVALID_ARG;
    if (n == 0) return PLK_OK;
    PLK_HIP(hipSetDevice(ctx->device));
    DevBuf a, b, o;
    PLK_HIP(a.alloc_copy(a_host, n * 32));
    PLK_HIP(b.alloc_copy(b_host, n * 32));
    PLK_HIP(o.alloc(n * 32));
    const dim3 grid((unsigned)((n + 255) / 256));
    if (field == 0) hipLaunchKernelGGL(k_field_op<FrP>, grid, dim3(256), 0, ctx->stream, op, (const uint32_t*)a.p, (const uint32_t*)b.p, (uint32_t*)o.p, n);
    else hipLaunchKernelGGL(k_field_op<FqP>, grid, dim3(256), 0, ctx->stream, op, (const uint32_t*)a.p, (const uint32_t*)b.p, (uint32_t*)o.p, n);
    PLK_HIP(hipGetLastError());
    PLK_HIP(hipStreamSynchronize(ctx->stream));
    PLK_HIP(hipMemcpy(out_host, o.p, n * 32, hipMemcpyDeviceToHost));
    return PLK_OK;
}

extern "C" int32_t plk_poly_eval_dev(plk_ctx* ctx, const void* coeffs_dev, uint64_t len, const uint64_t x[4], uint64_t out[4]) {
    if (!ctx || !x || !out || (len && !coeffs_dev)) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    const uint64_t words = division_work_words(len);
    PLK_TRY(ensure_scratch(ctx, words * 4));
    PLK_TRY(division_enqueue(ctx, ctx->stream, (const uint32_t*)coeffs_dev, len, x, nullptr, nullptr, ctx->scratch, ctx->scratch + 8, 0, 0));
    PLK_HIP(hipMemcpyAsync(out, ctx->scratch, 32, hipMemcpyDeviceToHost, ctx->stream));
    PLK_HIP(hipStreamSynchronize(ctx->stream));
    prof_collect(ctx, false);
    return PLK_OK;
}
extern "C" int32_t plk_poly_eval(plk_ctx* ctx, const uint64_t* coeffs_host, uint64_t len, const uint64_t x[4], uint64_t out[4]) {
    if (!ctx || (len && !coeffs_host)) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    DevBuf d;
    if (len) PLK_HIP(d.alloc_copy(coeffs_host, len * 32));
    return plk_poly_eval_dev(ctx, d.p, len, x, out);
}

// ---- evaluation for callers that cannot allocate ---------------------------------------------------
// plk_poly_eval_dev needs per-block sums and carries between its kernels (division_work_words(len) words); it takes them from the
// context's grow-on-demand buffer.  The variant below takes them from a fixed array in device memory instead, so that it never
// calls hipMalloc / hipFree.  Every GPU the library runs on holds its own copy of the array at its own address until the process
// ends; `scratch` itself is only a host-side handle (not a device address), so the address is asked for once per device, with
// that device current, and kept.  The kernels overwrite every word they later read (k_div_sums writes all nb block sums before
// k_div_carries reads them), so nothing has to be cleared between calls; two calls that overlap on one GPU would share the
// array and race.
constexpr uint64_t EVAL_STATIC_MAX_LEN = PLK_POLY_EVAL_STATIC_MAX_LEN;
constexpr uint64_t EVAL_STATIC_WORDS = 2 * (EVAL_STATIC_MAX_LEN / ((uint64_t)plk::DIV_THREADS << plk::DIV_LOG_L_LONG) + 1) * 8 + 64;   // = division_work_words(EVAL_STATIC_MAX_LEN): 524,608 bytes
static_assert(2 * (plk::DIV_LONG_FROM / ((uint64_t)plk::DIV_THREADS << plk::DIV_LOG_L_SHORT) + 1) * 8 + 64 <= EVAL_STATIC_WORDS, "the short form's block sums fit as well");
static __device__ __align__(256) uint32_t scratch[EVAL_STATIC_WORDS];
namespace {
constexpr int STATIC_MAX_DEVICES = 64;
std::mutex g_static_mutex;
uint32_t* g_static_ptr[STATIC_MAX_DEVICES] = {};
}  // namespace
namespace plk {
int32_t static_scratch_init(plk_ctx* ctx) {
    if (ctx->device < 0 || ctx->device >= STATIC_MAX_DEVICES) { ctx->err = "device index beyond the static scratch table"; return PLK_ERR_INVALID_ARG; }
    std::lock_guard<std::mutex> lock(g_static_mutex);
    if (g_static_ptr[ctx->device]) return PLK_OK;
    PLK_HIP(hipSetDevice(ctx->device));
    void* p = nullptr;
    PLK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) { ctx->err = "hipGetSymbolAddress returned no address for the static scratch"; return PLK_ERR_HIP; }
    g_static_ptr[ctx->device] = (uint32_t*)p;
    return PLK_OK;
}
}  // namespace plk
extern "C" int32_t plk_poly_eval_dev_static(plk_ctx* ctx, const void* coeffs_dev, uint64_t len, const uint64_t x[4], void* out_dev) {
    if (!ctx || !x || !out_dev || (len && !coeffs_dev)) return PLK_ERR_INVALID_ARG;
    if (len > EVAL_STATIC_MAX_LEN) { ctx->err = "plk_poly_eval_dev_static: more than PLK_POLY_EVAL_STATIC_MAX_LEN coefficients"; return PLK_ERR_INVALID_ARG; }
    uint32_t* work = ctx->device >= 0 && ctx->device < STATIC_MAX_DEVICES ? g_static_ptr[ctx->device] : nullptr;
    if (!work) { ctx->err = "plk_poly_eval_dev_static: static scratch of this device was not set up"; return PLK_ERR_HIP; }
    if (division_work_words(len) > EVAL_STATIC_WORDS) { ctx->err = "plk_poly_eval_dev_static: work area exceeds the static scratch"; return PLK_ERR_INVALID_ARG; }
    PLK_HIP(hipSetDevice(ctx->device));
    PLK_TRY(compute_guard(ctx));                 // a stream wait at most: nothing is allocated
    const int profiling = ctx->profiling;
    ctx->profiling = 0;                       // profiling brackets create events: not on this path
    const int32_t rc = division_enqueue(ctx, ctx->stream, (const uint32_t*)coeffs_dev, len, x, nullptr, nullptr, (uint32_t*)out_dev, work, 0, 0);
    ctx->profiling = profiling;
    return rc;
}

static int32_t read_flag(plk_ctx* ctx, uint32_t* out) {
    PLK_HIP(hipMemcpyAsync(ctx->flag_host, ctx->flag_dev, 4, hipMemcpyDeviceToHost, ctx->stream));
    PLK_HIP(hipStreamSynchronize(ctx->stream));
    *out = *ctx->flag_host;
    return PLK_OK;
}

extern "C" int32_t plk_grand_product_values_dev(plk_ctx* ctx, const void* witness_flat_dev, const void* sigma_dev, uint32_t log_n, const uint64_t beta[4],
                                                const uint64_t gamma[4], void* z_vals_out_dev, void* z_coeffs_out_dev);
extern "C" int32_t plk_grand_product_dev(plk_ctx* ctx, const void* witness_flat_dev, const void* sigma_dev, uint32_t log_n, const uint64_t beta[4],
                                         const uint64_t gamma[4], void* z_coeffs_out_dev) {
    return plk_grand_product_values_dev(ctx, witness_flat_dev, sigma_dev, log_n, beta, gamma, nullptr, z_coeffs_out_dev);
}
// wait == false: returns with the last kernels (values -> coefficients) still running on the compute stream
static int32_t grand_product(plk_ctx* ctx, const void* witness_flat_dev, const void* sigma_dev, uint32_t log_n, const uint64_t beta[4], const uint64_t gamma[4],
                             void* z_vals_out_dev, void* z_coeffs_out_dev, bool wait);
extern "C" int32_t plk_grand_product_values_dev(plk_ctx* ctx, const void* witness_flat_dev, const void* sigma_dev, uint32_t log_n, const uint64_t beta[4],
                                                const uint64_t gamma[4], void* z_vals_out_dev, void* z_coeffs_out_dev) {
    return grand_product(ctx, witness_flat_dev, sigma_dev, log_n, beta, gamma, z_vals_out_dev, z_coeffs_out_dev, true);
}
extern "C" int32_t plk_grand_product_dev_enqueue(plk_ctx* ctx, const void* witness_flat_dev, const void* sigma_dev, uint32_t log_n, const uint64_t beta[4],
                                                 const uint64_t gamma[4], void* z_vals_out_dev, void* z_coeffs_out_dev) {
    return grand_product(ctx, witness_flat_dev, sigma_dev, log_n, beta, gamma, z_vals_out_dev, z_coeffs_out_dev, false);
}
static int32_t grand_product(plk_ctx* ctx, const void* witness_flat_dev, const void* sigma_dev, uint32_t log_n, const uint64_t beta[4], const uint64_t gamma[4],
                             void* z_vals_out_dev, void* z_coeffs_out_dev, bool wait) {
    // the variant that does not wait may leave out the coefficients (values only, no interpolation)
    if (!ctx || !witness_flat_dev || !sigma_dev || !beta || !gamma || log_n > 26) return PLK_ERR_INVALID_ARG;
    if (!z_coeffs_out_dev && (wait || !z_vals_out_dev)) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    const uint64_t n = 1ull << log_n;
    PLK_TRY(compute_guard(ctx));
    hipStream_t st = ctx->stream;
    // scratch: num[n], den[n], scan work (2 * n/1024 + slack), all 8-word elements
    const uint64_t work_elems = 2 * (n / 1024 + 64);
    PLK_TRY(ensure_scratch(ctx, (2 * n + work_elems) * 32));
    uint32_t *num = ctx->scratch, *den = num + 8 * n, *work = den + 8 * n;
    PLK_HIP(hipMemsetAsync(ctx->flag_dev, 0, 4, st));
    prof_begin(ctx, st, "pointwise");
    const uint64_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK, tot_stride = 8 * nb;    // work: block products of the numerators, of the denominators, deeper levels
    hipLaunchKernelGGL(k_gp_local, dim3((unsigned)nb, 2), dim3(256), 0, st, (const uint32_t*)witness_flat_dev, (const uint64_t*)sigma_dev, n, w8_of(beta), w8_of(gamma), num, den,
                       work, tot_stride, ctx->flag_dev);
    if (nb > 1) PLK_TRY(scan_mul(ctx, st, work, tot_stride, 2, nb, false, work + 2 * tot_stride));
    prof_end(ctx, st);
    // one inversion of the full denominator product (the last scanned block product) on the host
    uint64_t dtot[4];
    PLK_HIP(hipMemcpyAsync(dtot, work + tot_stride + 8 * (nb - 1), 32, hipMemcpyDeviceToHost, st));
    uint32_t flag;
    PLK_TRY(read_flag(ctx, &flag));
    if (flag & 2u) { ctx->err = "sigma entry out of range"; return PLK_ERR_INDEX_OUT_OF_RANGE; }
    // dtot = D * 2^261 read as the 2^256 form of 32 D; its inverse times 32 is D^-1 * 2^256
    host::el d = host::el_load(dtot);
    while (host::geq(d.v, host::FR.p)) host::sub_inplace(d.v, host::FR.p);
    if (host::is_zero(d)) { ctx->err = "grand product: denominator is zero"; return PLK_ERR_INVERSE_OF_ZERO; }
    const host::el dinv = host::mul(host::FR, host::inv(host::FR, d), host::from_u64(host::FR, 32));
    prof_begin(ctx, st, "pointwise");
    hipLaunchKernelGGL(k_gp_finish, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, num, den, work, tot_stride, n, w8_of(dinv),
                       (uint32_t*)(z_coeffs_out_dev ? z_coeffs_out_dev : z_vals_out_dev));
    prof_end(ctx, st);
    PLK_HIP(hipGetLastError());
    if (!z_coeffs_out_dev) return PLK_OK;
    if (z_vals_out_dev) PLK_HIP(hipMemcpyAsync(z_vals_out_dev, z_coeffs_out_dev, n * 32, hipMemcpyDeviceToDevice, st));   // z on H, before the interpolation
    PLK_TRY(ntt_run(ctx, st, (uint32_t*)z_coeffs_out_dev, log_n, true, nullptr));
    if (!wait) return PLK_OK;
    PLK_HIP(hipStreamSynchronize(st));
    prof_collect(ctx, false);
    return PLK_OK;
}
extern "C" int32_t plk_grand_product(plk_ctx* ctx, const uint64_t* witness_flat_host, const uint64_t* sigma_host, uint32_t log_n, const uint64_t beta[4],
                                     const uint64_t gamma[4], uint64_t* z_coeffs_out_host) {
    if (!ctx || !witness_flat_host || !sigma_host || !z_coeffs_out_host || log_n > 26) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    const uint64_t n = 1ull << log_n;
    DevBuf w, s, z;
    PLK_HIP(w.alloc_copy(witness_flat_host, 3 * n * 32));
    PLK_HIP(s.alloc_copy(sigma_host, 3 * n * 8));
    PLK_HIP(z.alloc(n * 32));
    PLK_TRY(plk_grand_product_dev(ctx, w.p, s.p, log_n, beta, gamma, z.p));
    PLK_HIP(hipMemcpy(z_coeffs_out_host, z.p, n * 32, hipMemcpyDeviceToHost));
    return PLK_OK;
}

// t_len == nullptr: enqueue only (the verdict is collected by plk_quotient_dev_finish)
static int32_t quotient_enqueue(plk_ctx* ctx, const void* a, const void* b, const void* c, const void* q_add_vals, const void* q_mul_vals,
                                const void* z_vals, const void* sigma, uint32_t log_n, const uint64_t beta[4], const uint64_t gamma[4],
                                const uint64_t alpha[4], const uint64_t* pi_rows_host, const uint64_t* pi_values_host, uint64_t pi_count,
                                void* t_out_dev, uint64_t* t_len) {
    if (!ctx || !a || !b || !c || !q_add_vals || !q_mul_vals || !z_vals || !sigma || !beta || !gamma || !t_out_dev || log_n > 26)
        return PLK_ERR_INVALID_ARG;
    if (pi_count && (!alpha || !pi_rows_host || !pi_values_host)) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    const uint64_t n = 1ull << log_n;
    PLK_TRY(compute_guard(ctx));
    hipStream_t st = ctx->stream;
    uint32_t* t = (uint32_t*)t_out_dev;
    // public inputs: one entry per distinct row, the last assignment wins (src/circuit.rs:432 assigns, it does not add)
    std::vector<uint64_t> rows, vals;
    if (pi_count) {
        std::map<uint64_t, uint64_t> last;
        for (uint64_t k = 0; k < pi_count; ++k) {
            if (pi_rows_host[k] >= n) { ctx->err = "public input row outside the domain"; return PLK_ERR_INDEX_OUT_OF_RANGE; }
            last[pi_rows_host[k]] = k;
        }
        for (const auto& e : last) {
            rows.push_back(e.first);
            vals.insert(vals.end(), pi_values_host + 4 * e.second, pi_values_host + 4 * e.second + 4);
        }
    }
    DevBuf rows_dev, vals_dev;
    if (!rows.empty()) {
        PLK_HIP(rows_dev.alloc_copy(rows.data(), rows.size() * 8));
        PLK_HIP(vals_dev.alloc_copy(vals.data(), vals.size() * 8));
    }
    PLK_HIP(hipMemsetAsync(ctx->flag_dev, 0, 4, st));
    prof_begin(ctx, st, "pointwise");
    hipLaunchKernelGGL(k_quotient_vals, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const uint32_t*)a, (const uint32_t*)b, (const uint32_t*)c,
                       (const uint32_t*)q_add_vals, (const uint32_t*)q_mul_vals, (const uint32_t*)z_vals, (const uint64_t*)sigma, n, w8_of(beta), w8_of(gamma), t,
                       ctx->flag_dev);
    if (!rows.empty())
        hipLaunchKernelGGL(k_public_input_term, dim3((unsigned)((rows.size() + 255) / 256)), dim3(256), 0, st, (const uint32_t*)a, (const uint64_t*)rows_dev.p,
                           (const uint32_t*)vals_dev.p, (uint64_t)rows.size(), w8_of(alpha), t);
    prof_end(ctx, st);
    PLK_HIP(hipGetLastError());
    // numerator in coefficient form: ifft(gate) + ifft(perm) + ifft(pub) = ifft(gate + perm + pub)
    PLK_TRY(ntt_run(ctx, st, t, log_n, true, nullptr));
    // divide_by_vanishing_poly on a numerator of degree < n: quotient 0, remainder = the numerator, which must vanish
    hipLaunchKernelGGL(k_any_nonzero, dim3(256), dim3(256), 0, st, t, n * 8, ctx->flag_dev);
    PLK_HIP(hipGetLastError());
    PLK_HIP(hipMemcpyAsync(ctx->flag_host, ctx->flag_dev, 4, hipMemcpyDeviceToHost, st));
    ctx->quotient_stream = st;
    if (!t_len) return PLK_OK;     // split form: plk_quotient_dev_finish collects the verdict (no public inputs there, so nothing to keep alive)
    return plk_quotient_dev_finish(ctx, t_len);
}
extern "C" int32_t plk_quotient_dev_enqueue(plk_ctx* ctx, const void* a, const void* b, const void* c, const void* q_add_vals, const void* q_mul_vals,
                                            const void* z_vals, const void* sigma, uint32_t log_n, const uint64_t beta[4], const uint64_t gamma[4],
                                            void* t_out_dev) {
    if (ctx && ctx->quotient_stream) { ctx->err = "plk_quotient_dev_enqueue: the previous quotient was not finished"; return PLK_ERR_INVALID_ARG; }
    return quotient_enqueue(ctx, a, b, c, q_add_vals, q_mul_vals, z_vals, sigma, log_n, beta, gamma, nullptr, nullptr, nullptr, 0, t_out_dev, nullptr);
}
extern "C" int32_t plk_quotient_dev_finish(plk_ctx* ctx, uint64_t* t_len) {
    if (!ctx || !t_len) return PLK_ERR_INVALID_ARG;
    if (!ctx->quotient_stream) { ctx->err = "plk_quotient_dev_finish: no quotient enqueued"; return PLK_ERR_INVALID_ARG; }
    PLK_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->quotient_stream;
    ctx->quotient_stream = nullptr;
    PLK_HIP(hipStreamSynchronize(st));
    const uint32_t flag = *ctx->flag_host;
    prof_collect(ctx, false);
    if (flag & 2u) { ctx->err = "sigma entry out of range"; return PLK_ERR_INDEX_OUT_OF_RANGE; }
    if (flag & 4u) { ctx->err = "permutation constraint: inverse of zero"; return PLK_ERR_INVERSE_OF_ZERO; }
    if (flag & 1u) { ctx->err = "t(X) not divisible by Z_H(X)"; return PLK_ERR_NOT_DIVISIBLE; }
    *t_len = 0;
    return PLK_OK;
}
extern "C" int32_t plk_quotient_pi_dev(plk_ctx* ctx, const void* a, const void* b, const void* c, const void* q_add_vals, const void* q_mul_vals,
                                       const void* z_vals, const void* sigma, uint32_t log_n, const uint64_t beta[4], const uint64_t gamma[4],
                                       const uint64_t alpha[4], const uint64_t* pi_rows_host, const uint64_t* pi_values_host, uint64_t pi_count,
                                       void* t_out_dev, uint64_t* t_len) {
    if (!t_len) return PLK_ERR_INVALID_ARG;
    if (ctx && ctx->quotient_stream) { ctx->err = "a quotient enqueued with plk_quotient_dev_enqueue is still open"; return PLK_ERR_INVALID_ARG; }
    return quotient_enqueue(ctx, a, b, c, q_add_vals, q_mul_vals, z_vals, sigma, log_n, beta, gamma, alpha, pi_rows_host, pi_values_host, pi_count, t_out_dev, t_len);
}
extern "C" int32_t plk_quotient_dev(plk_ctx* ctx, const void* a, const void* b, const void* c, const void* q_add_vals, const void* q_mul_vals,
                                    const void* z_vals, const void* sigma, uint32_t log_n, const uint64_t beta[4], const uint64_t gamma[4], void* t_out_dev,
                                    uint64_t* t_len) {
    return plk_quotient_pi_dev(ctx, a, b, c, q_add_vals, q_mul_vals, z_vals, sigma, log_n, beta, gamma, nullptr, nullptr, nullptr, 0, t_out_dev, t_len);
}
extern "C" int32_t plk_quotient(plk_ctx* ctx, const uint64_t* a, const uint64_t* b, const uint64_t* c, const uint64_t* q_add_vals, const uint64_t* q_mul_vals,
                                const uint64_t* z_vals, const uint64_t* sigma, uint32_t log_n, const uint64_t beta[4], const uint64_t gamma[4],
                                uint64_t* t_out_host, uint64_t* t_len) {
    return plk_quotient_pi(ctx, a, b, c, q_add_vals, q_mul_vals, z_vals, sigma, log_n, beta, gamma, nullptr, nullptr, nullptr, 0, t_out_host, t_len);
}
extern "C" int32_t plk_quotient_pi(plk_ctx* ctx, const uint64_t* a, const uint64_t* b, const uint64_t* c, const uint64_t* q_add_vals, const uint64_t* q_mul_vals,
                                   const uint64_t* z_vals, const uint64_t* sigma, uint32_t log_n, const uint64_t beta[4], const uint64_t gamma[4],
                                   const uint64_t alpha[4], const uint64_t* pi_rows, const uint64_t* pi_values, uint64_t pi_count, uint64_t* t_out_host,
                                   uint64_t* t_len) {
    if (!ctx || !a || !b || !c || !q_add_vals || !q_mul_vals || !z_vals || !sigma || !t_out_host || log_n > 26) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    const uint64_t n = 1ull << log_n;
    const uint64_t* src[6] = {a, b, c, q_add_vals, q_mul_vals, z_vals};
    DevBuf d[6], s, t;
    for (int i = 0; i < 6; ++i) PLK_HIP(d[i].alloc_copy(src[i], n * 32));
    PLK_HIP(s.alloc_copy(sigma, 3 * n * 8));
    PLK_HIP(t.alloc(n * 32));
    PLK_TRY(plk_quotient_pi_dev(ctx, d[0].p, d[1].p, d[2].p, d[3].p, d[4].p, d[5].p, s.p, log_n, beta, gamma, alpha, pi_rows, pi_values, pi_count, t.p, t_len));
    PLK_HIP(hipMemcpy(t_out_host, t.p, n * 32, hipMemcpyDeviceToHost));
    return PLK_OK;
}

extern "C" int32_t plk_interpolate_permutation_dev(plk_ctx* ctx, const void* sigma_dev, uint64_t sigma_len, uint32_t log_n4, void* s_id_out_dev,
                                                   void* s_sigma_out_dev) {
    if (!ctx || (sigma_len && !sigma_dev) || !s_id_out_dev || !s_sigma_out_dev || log_n4 < 1 || log_n4 > 27 || sigma_len > (1ull << log_n4))
        return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    const uint64_t n4 = 1ull << log_n4;
    PLK_TRY(compute_guard(ctx));
    hipStream_t st = ctx->stream;
    const uint32_t* W;
    PLK_TRY(ntt_twiddles(ctx, log_n4, &W));
    PLK_HIP(hipMemsetAsync(ctx->flag_dev, 0, 4, st));
    hipLaunchKernelGGL(k_perm_values, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (const uint64_t*)sigma_dev, sigma_len, log_n4, W,
                       (uint32_t*)s_id_out_dev, (uint32_t*)s_sigma_out_dev, ctx->flag_dev);
    PLK_HIP(hipGetLastError());
    PLK_TRY(ntt_run(ctx, st, (uint32_t*)s_id_out_dev, log_n4, true, nullptr));
    PLK_TRY(ntt_run(ctx, st, (uint32_t*)s_sigma_out_dev, log_n4, true, nullptr));
    uint32_t flag;
    PLK_TRY(read_flag(ctx, &flag));
    prof_collect(ctx, false);
    (void)flag;
    return PLK_OK;
}
extern "C" int32_t plk_interpolate_permutation(plk_ctx* ctx, const uint64_t* sigma_host, uint64_t sigma_len, uint32_t log_n4, uint64_t* s_id_out_host,
                                               uint64_t* s_sigma_out_host) {
    if (!ctx || !s_id_out_host || !s_sigma_out_host || log_n4 < 1 || log_n4 > 27) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    const uint64_t n4 = 1ull << log_n4;
    DevBuf s, o0, o1;
    if (sigma_len) PLK_HIP(s.alloc_copy(sigma_host, sigma_len * 8));
    PLK_HIP(o0.alloc(n4 * 32));
    PLK_HIP(o1.alloc(n4 * 32));
    PLK_TRY(plk_interpolate_permutation_dev(ctx, s.p, sigma_len, log_n4, o0.p, o1.p));
    PLK_HIP(hipMemcpy(s_id_out_host, o0.p, n4 * 32, hipMemcpyDeviceToHost));
    PLK_HIP(hipMemcpy(s_sigma_out_host, o1.p, n4 * 32, hipMemcpyDeviceToHost));
    return PLK_OK;
}
extern "C" int32_t plk_interpolate_selector(plk_ctx* ctx, const uint64_t* values_host, uint64_t len, uint64_t* coeffs_out_host, uint32_t* log_n_out) {
    if (!ctx || !values_host || !coeffs_out_host || !log_n_out || len == 0) return PLK_ERR_INVALID_ARG;
    uint32_t log_n = 0;
    while ((1ull << log_n) < len) ++log_n;
    const uint64_t n = 1ull << log_n;
    memcpy(coeffs_out_host, values_host, len * 32);
    memset(coeffs_out_host + 4 * len, 0, (n - len) * 32);
    *log_n_out = log_n;
    return plk_ntt(ctx, coeffs_out_host, log_n, 1, nullptr);
}

// Extended-coset quotient: see include/plonk_hip.h.  Ten polynomials go from values on H to values on the coset g H_4n
// (inverse NTT of n points, zero padding, coset NTT of 4n points), one pointwise kernel, one inverse coset NTT.
extern "C" int32_t plk_quotient_coset_dev(plk_ctx* ctx, const void* a, const void* b, const void* c, const void* q_add_vals, const void* q_mul_vals,
                                          const void* z_vals, const void* sigma, uint32_t log_n, const uint64_t beta[4], const uint64_t gamma[4],
                                          const uint64_t alpha[4], const uint64_t* coset_gen, void* t_out_dev) {
    if (!ctx || !a || !b || !c || !q_add_vals || !q_mul_vals || !z_vals || !sigma || !beta || !gamma || !alpha || !t_out_dev || log_n > 25)
        return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    const uint64_t n = 1ull << log_n, n4 = 4 * n;
    PLK_TRY(compute_guard(ctx));
    hipStream_t st = ctx->stream;
    host::el g = coset_gen ? host::el_load(coset_gen) : host::from_u64(host::FR, 5);
    // 1 / (g^n w_4^i - 1), i < 4
    host::el gn = g;
    for (uint32_t k = 0; k < log_n; ++k) gn = host::mul(host::FR, gn, gn);
    static const uint64_t ROOT28[4] = {0x636e735580d13d9cULL, 0xa22bf3742445ffd6ULL, 0x56452ac01eb203d8ULL, 0x1860ef942963f9e7ULL};
    host::el w4 = host::el_load(ROOT28);
    for (int k = 2; k < 28; ++k) w4 = host::mul(host::FR, w4, w4);
    CosetQuotientArgs args;
    args.beta = w8_of(beta); args.gamma = w8_of(gamma); args.alpha = w8_of(alpha);
    host::el x = gn;
    for (int i = 0; i < 4; ++i) {
        const host::el d = host::sub(host::FR, x, host::one(host::FR));
        if (host::is_zero(d)) { ctx->err = "the coset meets the domain: Z_H vanishes on it"; return PLK_ERR_INVERSE_OF_ZERO; }
        args.zh_inv[i] = w8_of(host::inv(host::FR, d));
        x = host::mul(host::FR, x, w4);
    }
    DevBuf work_buf;
    if (work_buf.alloc(10 * n4 * 32) != hipSuccess) { (void)hipGetLastError(); ctx->err = "out of device memory for the coset evaluations"; return PLK_ERR_ALLOC; }
    uint32_t* work = (uint32_t*)work_buf.p;
    uint32_t* p[10];
    for (int k = 0; k < 10; ++k) p[k] = work + (size_t)k * n4 * 8;
    const void* cols[6] = {a, b, c, q_add_vals, q_mul_vals, z_vals};
    int32_t rc = PLK_OK;
    auto fail = [&](int32_t code) { (void)hipStreamSynchronize(st); return code; };   // kernels may still be using the work buffer
    if (hipMemsetAsync(ctx->flag_dev, 0, 4, st) != hipSuccess) return fail(PLK_ERR_HIP);
    for (int k = 0; k < 6; ++k)
        if (hipMemcpyAsync(p[k], cols[k], n * 32, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(PLK_ERR_HIP);
    prof_begin(ctx, st, "pointwise");
    hipLaunchKernelGGL(k_coset_labels, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const uint64_t*)sigma, n, p[6], p[7], p[8], p[9], ctx->flag_dev);
    prof_end(ctx, st);
    const uint64_t gen_limbs[4] = {g.v[0], g.v[1], g.v[2], g.v[3]};
    for (int k = 0; k < 10 && rc == PLK_OK; ++k) {
        rc = ntt_run(ctx, st, p[k], log_n, true, nullptr);
        if (rc == PLK_OK && hipMemsetAsync(p[k] + n * 8, 0, 3 * n * 32, st) != hipSuccess) rc = PLK_ERR_HIP;
        if (rc == PLK_OK) rc = ntt_run(ctx, st, p[k], log_n + 2, false, gen_limbs);
    }
    if (rc != PLK_OK) return fail(rc);
    uint32_t* t = (uint32_t*)t_out_dev;
    prof_begin(ctx, st, "pointwise");
    hipLaunchKernelGGL(k_quotient_coset, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], n4, args, t);
    prof_end(ctx, st);
    if (hipGetLastError() != hipSuccess) return fail(PLK_ERR_HIP);
    rc = ntt_run(ctx, st, t, log_n + 2, true, gen_limbs);
    if (rc != PLK_OK) return fail(rc);
    // the numerator has degree <= 4n - 4, so it is a multiple of X^n - 1 exactly when coefficients 3n - 3 .. 4n - 1 of t vanish
    const uint64_t first = 3 * n >= 3 ? 3 * n - 3 : 0;
    hipLaunchKernelGGL(k_any_nonzero, dim3(64), dim3(256), 0, st, t + first * 8, (n4 - first) * 8, ctx->flag_dev);
    uint32_t flag = 0;
    rc = read_flag(ctx, &flag);   // waits for the stream: the work buffer is idle from here on
    if (rc != PLK_OK) return fail(rc);
    prof_collect(ctx, false);
    PLK_TRY(rc);
    if (flag & 2u) { ctx->err = "sigma entry out of range"; return PLK_ERR_INDEX_OUT_OF_RANGE; }
    if (flag & 1u) { ctx->err = "t(X) not divisible by Z_H(X)"; return PLK_ERR_NOT_DIVISIBLE; }
    return PLK_OK;
}
extern "C" int32_t plk_quotient_coset(plk_ctx* ctx, const uint64_t* a, const uint64_t* b, const uint64_t* c, const uint64_t* q_add_vals,
                                      const uint64_t* q_mul_vals, const uint64_t* z_vals, const uint64_t* sigma, uint32_t log_n, const uint64_t beta[4],
                                      const uint64_t gamma[4], const uint64_t alpha[4], const uint64_t* coset_gen, uint64_t* t_out_host) {
    if (!ctx || !a || !b || !c || !q_add_vals || !q_mul_vals || !z_vals || !sigma || !t_out_host || log_n > 25) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    const uint64_t n = 1ull << log_n;
    const uint64_t* src[6] = {a, b, c, q_add_vals, q_mul_vals, z_vals};
    DevBuf d[6], s, t;
    for (int i = 0; i < 6; ++i) PLK_HIP(d[i].alloc_copy(src[i], n * 32));
    PLK_HIP(s.alloc_copy(sigma, 3 * n * 8));
    PLK_HIP(t.alloc(4 * n * 32));
    const int32_t rc = plk_quotient_coset_dev(ctx, d[0].p, d[1].p, d[2].p, d[3].p, d[4].p, d[5].p, s.p, log_n, beta, gamma, alpha, coset_gen, t.p);
    // the coefficients are returned in both cases: a caller may want to look at a quotient that does not divide
    if (rc == PLK_OK || rc == PLK_ERR_NOT_DIVISIBLE) PLK_HIP(hipMemcpy(t_out_host, t.p, 4 * n * 32, hipMemcpyDeviceToHost));
    return rc;
}
