// Internal entry points shared between the translation units of libplonk_hip.
#pragma once
#include "common.cuh"
#include "host_math.hpp"

namespace plk {

// msm.hip
int32_t msm_acquire_slot(plk_ctx* ctx, int* out);
// lane: the stream that carries this MSM's sort (front_lane(); nullptr = the context's front stream)
// idle_hint: -1 = the library looks whether the device is idle (the first sort of a round gets more workgroups), 0 / 1 = the caller knows.
// blind_dev: where the n_blind scalars over the gamma powers sit (null: the slot's own buffer).
int32_t msm_enqueue(plk_ctx* ctx, MsmSlot& s, const uint32_t* scalars_dev, uint64_t k, uint64_t base_offset, uint32_t n_blind, hipStream_t lane = nullptr,
                    int idle_hint = -1, const uint32_t* blind_dev = nullptr);
// Stream for the front work (division, sort) of the next commitment or opening.  The front work of consecutive jobs alternates between
// the front stream and the reduction stream, so that the sort of one job runs beside the sort of the next: on one stream the sorts
// (latency-bound, a fraction of the chip each) follow each other and feed the accumulation kernels too slowly.  One stream while
// compute work is detached to the reduction stream or uploads are queued on it.
hipStream_t front_lane(plk_ctx* ctx);
// Several MSMs that share one window plan and their launches (one sort, one accumulation): the short MSMs of a round.
struct MsmJob {
    int slot;                    // index into ctx->slot
    const uint32_t* scalars;     // device
    uint64_t k, base_offset;
    uint32_t n_blind;
    const uint32_t* blind_dev;   // null: the slot's own buffer
};
int msm_plan_for(const plk_ctx* ctx, uint64_t k, uint64_t base_offset);   // MSM_C_LONG, MSM_C or MSM_C_SHORT
int msm_plan_windows(int plan_c);
uint64_t msm_share(const plk_ctx* ctx, uint64_t k, uint64_t base_offset);   // scalars of that range on this rank
int32_t msm_enqueue_group(plk_ctx* ctx, const MsmJob* jobs, uint32_t n, hipStream_t lane, int idle_hint, int plan_c);
int32_t msm_wait(plk_ctx* ctx, MsmSlot& s, host::pt* out);
// the host tail of an MSM in independent pieces (for several host threads) and their combination
constexpr int MSM_TAIL_PIECES = 5;
void msm_tail_piece(const uint32_t* marg, int plan_c, int piece, host::pt* out);
void msm_tail_combine(const host::pt* pieces, int plan_c, host::pt* out);   // enqueues the pending reductions first
int32_t msm_flush_reductions(plk_ctx* ctx);
void slot_destroy(MsmSlot& s);

// ntt.hip
int32_t ntt_run(plk_ctx* ctx, hipStream_t st, uint32_t* data_dev, uint32_t log_n, bool inverse, const uint64_t* coset_gen);
int32_t ntt_twiddles(plk_ctx* ctx, uint32_t log_n, const uint32_t** out);   // w^i, i < n/2 (x*2^261 words); n >= 2
void ntt_release(plk_ctx* ctx);
// half 1: this rank's work before the exchange of the four-step transform, half 2: after it (in place on this rank's N / world elements)
int32_t ntt_sharded_half(plk_ctx* ctx, hipStream_t st, uint32_t* buf, uint32_t log_n, bool inverse, const uint64_t* coset_gen, uint32_t rank, uint32_t world, int half);

// poly.hip
// Suffix Horner scan on stream st:  h_i = sum_{k>=i} c_k zeta^(k-i).  Writes scale*h_i to q_out[i-1] for the quotient indices
// i-1 in [q_first, q_first + q_count) (nothing when q_out is null; a sharded context only needs its own range) and h_0 (unscaled)
// to h0_out (8 words, canonical).  work: >= division_work_words(len) words.
int32_t division_enqueue(plk_ctx* ctx, hipStream_t st, const uint32_t* coeffs_dev, uint64_t len, const uint64_t zeta[4], const uint64_t scale[4],
                         uint32_t* q_out, uint32_t* h0_out, uint32_t* work, uint64_t q_first, uint64_t q_count);
uint64_t division_work_words(uint64_t len);
// The same scan for several polynomials at one point and one scale, as one batch of launches.
constexpr int DIV_BATCH_MAX = MSM_SLOTS;
struct DivJob {
    const uint32_t* coeffs;
    uint64_t len;
    uint32_t* q_out;      // may be null
    uint32_t* h0_out;     // may be null
    uint32_t* work;       // >= division_work_words(len) words
    uint64_t q_first, q_count;
};
int32_t division_enqueue_batch(plk_ctx* ctx, hipStream_t st, uint32_t count, const DivJob* jobs, const uint64_t zeta[4], const uint64_t scale[4]);
// asks the device for the address of the fixed scratch array behind plk_poly_eval_dev_static (once per device; called by plk_create)
int32_t static_scratch_init(plk_ctx* ctx);

}  // namespace plk
