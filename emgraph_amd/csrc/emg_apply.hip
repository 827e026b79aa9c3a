// emg_apply.hip — K8: deterministic row-sparse optimizer apply.
//
// The reference hands TF IndexedSlices (row ids + gradient rows) to Keras optimizers
// (training/sgd.py:97, momentum.py:63, adagrad.py:42, adam.py:45).  Here the backward pass writes one
// gradient row per (positive group, role) without atomics; emg_group.hip groups them by destination (stable:
// equal destinations keep slot order, so the float sum order is fixed => bit-reproducible training, the reference's
// refit-determinism test tests/emgraph/models/test_models.py:338-367) and this file sums each destination's rows in
// that order and performs the optimizer update of the table row exactly once:
//   apply_segments_kernel   the training path (counting grouping): a persistent grid works from the SEGMENT
//                           DESCRIPTORS the grouping's scan emitted — destinations with 2..32 contributions,
//                           singletons (unless the backward kernel already updated them in place), 64-row block
//                           tasks of longer segments — both tables of a step in ONE launch
//   apply_rows_*_kernel     window kernels that find their segments in the sorted keys themselves: the sort
//                           backend (wide keys), rows of <= 16 chunks (four segments per wave) and scalar rows
#include <stdlib.h>
#include <string.h>
#include <cstring>

#include <atomic>

#include "emg_group_kernels.hpp"

namespace emg {

struct ApplyParams {
    float* table; int64_t n_rows; int64_t ld; int32_t k_int;
    float* state0; float* state1; int32_t* tag; int32_t step;
    int32_t half_rows;   // 1: rows of at most 32 chunks two at a time, one per half-wave (segment_update_half; 0: EMG_APPLY_HALF=0)
    int32_t state_lag;   // 1 (Adam, deferred dense pass): m, v of a multi / single destination are as of tag[row], w is current (below)
    const float* contrib; int64_t ldc;
    const uint32_t* keys; const uint32_t* vals; int64_t n;
    int32_t skip_single;   // 1: singletons were updated in place by the scoring kernel
    int32_t win;  // sorted positions per wave
    OptParams opt;
    // segments longer than `defer` rows leave the window kernel as BLOCK TASKS (apply_long_kernel): 64-row blocks of
    // the segment, summed by independent waves anywhere on the chip; arrive[head / 64] counts a segment's finished blocks
    LongTask* long_list; uint32_t* long_count; uint32_t long_cap; int32_t* arrive;
    double* lp_accum;  // += sum |w_pre|^p over the rows this launch updates (the caller scales by lambda); may be null
    int32_t defer;     // segments longer than this go to apply_long_kernel
    // Where the gradient row of the contribution at SORTED position t lives and what it is multiplied by.
    //   full rows (emg_apply_grouped):  srcrow = vals (the contribution's own row), coef = nullptr (1)
    //   FACTORED (bilinear models; emg_prepare_args.factored + emg_backward_args.fac_ws_ent): the gradient row of a
    //   negative's replacement entity is  coef * q  with q one of the two query rows of its triple group, so the
    //   backward kernel stores q once per group and one float per negative.  contrib rows: [0,B) subject rows, [B,2B)
    //   object rows, [2B,3B) q of the object side, [3B,4B) q of the subject side.  srcrow[t] is resolved by the
    //   grouping (mark_single_kernel: the codes are known there), coef[t] is written by the backward kernel straight
    //   into sorted order (pos_of_slot) — so a window's sources are ONE coalesced load each, like its keys.
    const uint32_t* srcrow; const float* coef;
    // segment descriptors of the counting grouping (apply_segments_kernel): see emg_group.hpp
    const Seg* multi; const uint32_t* single; const LongTask* tasks; const uint32_t* counters; uint32_t task_cap;
    int32_t which;             // 0: entity table, 1: relation table (which hyper-parameters of a StepCtl apply)
    const StepCtl* ctl;        // graph node: step number and learning rates from the device record
    // dense_here = 1: the rows no contribution touches (Keras Adam's dense-equivalent update) are visited by apply_segments_kernel's
    // own waves after their items — a row is untouched when the grouping counted nothing for it (off[r + 1] == off[r]) — instead of
    // by a launch of their own afterwards (untouched_rows_kernel, which asks tag[r] != step)
    const uint32_t* off; int32_t dense_here;
};

struct Src { uint32_t row; float coef; };
__device__ __forceinline__ Src contrib_src(const ApplyParams& P, int64_t t) {
    return Src{P.srcrow[t], P.coef ? P.coef[t] : 1.f};
}
// acc += coef * v, the product rounded on its own (never contracted into an fma): the same bits as adding a row the
// backward kernel stored as coef * q
__device__ __forceinline__ void add_scaled(float4& acc, const float4& v, float coef) {
#pragma clang fp contract(off)
    acc.x += coef * v.x; acc.y += coef * v.y; acc.z += coef * v.z; acc.w += coef * v.w;
}
__device__ __forceinline__ void add_scaled(float& acc, float v, float coef) {
#pragma clang fp contract(off)
    acc += coef * v;
}

// One wave per WIN (a power of two <= 64) consecutive SORTED positions: the wave finds the segment heads inside its window with
// one coalesced key load + a ballot (no head list, no atomics — a shared append counter saturates at
// ~90 atomics/us, slower than the whole sort) and processes them one after the other.  A segment may run
// past the window's end; its head's wave handles all of it.  Singleton segments are skipped when the
// backward kernel already applied them in place.
//   DEPTH     : contribution rows in flight per trip beyond the 2-row tail loop (2 = none: 70 VGPRs, 7 waves/SIMD;
//               the 8- and 16-deep forms of earlier versions ran at 4 and 2 waves/SIMD and were slower on both tables).
//   LONG > 0  : a segment of more than LONG rows (a hub entity of a Zipf-distributed graph collects thousands) is
//               NOT summed here by one wave (2 300 rows take 0.5 ms that way) but appended to a list for
//               apply_long_kernel, which spreads its 64-row blocks over the waves of a workgroup.

template <int W, int DEPTH>
__device__ __forceinline__ void sum_and_update(const ApplyParams& P, uint32_t key, int64_t t, int64_t end, int64_t w0,
                                               int64_t wend, Src mysrc, int lane, int nchunks, float& lp_acc) {
    auto source = [&](int64_t u) -> Src {  // wave-uniform u
        if (u < wend) return Src{(uint32_t)__shfl(mysrc.row, (int)(u - w0), 64), __shfl(mysrc.coef, (int)(u - w0), 64)};
        return contrib_src(P, u);
    };
    float* wrow = P.table + (int64_t)key * P.ld;
    float* s0row = P.state0 ? P.state0 + (int64_t)key * P.ld : nullptr;
    float* s1row = P.state1 ? P.state1 + (int64_t)key * P.ld : nullptr;
    if constexpr (W == 4) {
        // two row chunks per lane (columns 4*lane.. and 4*(lane+64)..) x DEPTH contributions per trip: independent
        // 16-byte loads in flight, added in contribution order (bit-reproducible sums)
        for (int c0 = 0; c0 < nchunks; c0 += 128) {
            const int ca = c0 + lane, cb = c0 + 64 + lane;
            const bool oa = ca < nchunks, ob = cb < nchunks;
            float4 accA = make_float4(0.f, 0.f, 0.f, 0.f), accB = accA, wA = accA, wB = accA;
            const float4 zero = accA;
            if (oa) wA = *reinterpret_cast<const float4*>(wrow + 4 * ca);
            if (ob) wB = *reinterpret_cast<const float4*>(wrow + 4 * cb);
            int64_t u = t;
            if constexpr (DEPTH > 2) {
                for (; u + DEPTH <= end; u += DEPTH) {
                    float4 va[DEPTH], vb[DEPTH];
                    float cf[DEPTH];
#pragma unroll
                    for (int j = 0; j < DEPTH; ++j) {
                        const Src sj = source(u + j);
                        const float* rj = P.contrib + (int64_t)sj.row * P.ldc;
                        cf[j] = sj.coef;
                        va[j] = oa ? *reinterpret_cast<const float4*>(rj + 4 * ca) : zero;
                        vb[j] = ob ? *reinterpret_cast<const float4*>(rj + 4 * cb) : zero;
                    }
#pragma unroll
                    for (int j = 0; j < DEPTH; ++j) {  // added in contribution order
                        if (oa) add_scaled(accA, va[j], cf[j]);
                        if (ob) add_scaled(accB, vb[j], cf[j]);
                    }
                }
            }
            for (; u < end; u += 2) {
                const bool two = u + 1 < end;
                const Src s0 = source(u), s1 = two ? source(u + 1) : s0;
                const float* r0 = P.contrib + (int64_t)s0.row * P.ldc;
                const float* r1 = P.contrib + (int64_t)s1.row * P.ldc;
                float4 v0a = zero, v0b = zero, v1a = zero, v1b = zero;
                if (oa) v0a = *reinterpret_cast<const float4*>(r0 + 4 * ca);
                if (ob) v0b = *reinterpret_cast<const float4*>(r0 + 4 * cb);
                if (oa && two) v1a = *reinterpret_cast<const float4*>(r1 + 4 * ca);
                if (ob && two) v1b = *reinterpret_cast<const float4*>(r1 + 4 * cb);
                if (oa) add_scaled(accA, v0a, s0.coef);
                if (ob) add_scaled(accB, v0b, s0.coef);
                if (oa && two) add_scaled(accA, v1a, s1.coef);
                if (ob && two) add_scaled(accB, v1b, s1.coef);
            }
            auto finish = [&](int c, float4 wv, const float4& g) {
                const int64_t off = 4 * (int64_t)c;
                float w[4] = {wv.x, wv.y, wv.z, wv.w};
                float gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    lp_fold(P.opt, w[j], gg[j], lp_acc);
                    opt_update_elem(P.opt, w[j], gg[j], s0row ? s0row + off + j : nullptr, s1row ? s1row + off + j : nullptr);
                }
                *reinterpret_cast<float4*>(wrow + off) = make_float4(w[0], w[1], w[2], w[3]);
            };
            if (oa) finish(ca, wA, accA);
            if (ob) finish(cb, wB, accB);
        }
    } else {
        for (int c = lane; c < nchunks; c += 64) {
            float acc = 0.f;
            for (int64_t u = t; u < end; ++u) {
                const Src su = source(u);
                add_scaled(acc, P.contrib[(int64_t)su.row * P.ldc + c], su.coef);
            }
            float wv = wrow[c];
            lp_fold(P.opt, wv, acc, lp_acc);
            opt_update_elem(P.opt, wv, acc, s0row ? s0row + c : nullptr, s1row ? s1row + c : nullptr);
            wrow[c] = wv;
        }
    }
    if (P.tag && lane == 0) P.tag[key] = P.step;
}

// first position after `lo` whose key differs, keys[lo] == key given (sorted keys): 64 probes per trip
__device__ __forceinline__ int64_t segment_end_search(const ApplyParams& P, int64_t lo, uint32_t key, int lane) {
    int64_t hi = P.n;   // keys[hi] != key or hi == n
    for (;;) {
        const int64_t span = hi - lo;          // > 0
        if (span == 1) return hi;
        const int64_t step = (span + 63) / 64;  // probes lo + step, lo + 2 step, ...: the last one may reach past hi
        const int64_t q = lo + (int64_t)(lane + 1) * step;
        const unsigned long long same = __ballot(q < hi && P.keys[q] == key);   // a prefix of ones
        const int cnt = __popcll(same);
        const int64_t lo2 = lo + (int64_t)cnt * step;
        if (cnt < 64 && lo2 + step < hi) hi = lo2 + step;
        lo = lo2;
    }
}

template <int W, int DEPTH>
__device__ __forceinline__ void apply_rows_body(const ApplyParams& P, int64_t block) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (block * blockDim.x + threadIdx.x) >> 6;
    const int64_t w0 = wave * P.win;
    const int64_t t0 = w0 + lane;
    const bool in = lane < P.win && t0 < P.n;
    const uint32_t mykey = in ? P.keys[t0] : 0u;
    const Src mysrc = in ? contrib_src(P, t0) : Src{0u, 0.f};   // the window's contribution rows: one coalesced load
    const bool head = in && (t0 == 0 || P.keys[t0 - 1] != mykey);
    const bool last = in && (t0 + 1 == P.n || P.keys[t0 + 1] != mykey);
    const unsigned long long heads = __ballot(head);
    unsigned long long todo = __ballot(head && !(P.skip_single && last));
    const int64_t wend = min(w0 + (int64_t)P.win, P.n);  // end of this window
    const int nchunks = P.k_int / W;
    float lp_acc = 0.f;
    while (todo) {
        const int b = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int64_t t = w0 + b;
        const uint32_t key = __shfl(mykey, b, 64);
        if ((int64_t)key >= P.n_rows) continue;  // defensive: never write outside the table
        // segment end: the next head inside the window (ballot, no memory traffic), else scan on past the window
        const unsigned long long after = heads & ~((2ull << b) - 1ull);
        int64_t end = after ? w0 + (__ffsll((long long)after) - 1) : wend;
        if (!after) {  // 64 keys per trip: a serial scan costs one dependent load per row of a long segment
            for (;;) {
                const int64_t q = end + lane;
                const unsigned long long same = __ballot(q < P.n && P.keys[q] == key);
                if (same == ~0ull) {
                    end += 64;
                    if (P.long_list && end - t > kLongSegment) {   // long (a hub row may have thousands): 64-ary search
                        end = segment_end_search(P, end - 1, key, lane);
                        break;
                    }
                    continue;
                }
                end += __ffsll((long long)~same) - 1;
                break;
            }
        }
        if (P.long_list && end - t > P.defer) {  // hand the segment over as block tasks (their order is irrelevant)
            const uint32_t len = (uint32_t)(end - t), nblk = (len + kLongSegment - 1) / kLongSegment;
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(P.long_count, nblk);
            base = __builtin_amdgcn_readfirstlane(base);
            const bool room = base + nblk <= P.long_cap;   // (always: sum of ceil(len / 64) over segments > defer <= n / defer)
            for (uint32_t b = lane; b < nblk && base + b < P.long_cap; b += 64)
                P.long_list[base + b] = LongTask{(uint32_t)t, b, room ? len : 0u};
            if (room) continue;
        }
        sum_and_update<W, DEPTH>(P, key, t, end, w0, wend, mysrc, lane, nchunks, lp_acc);
    }
    if (P.opt.lp_lambda != 0.f) wave_add_double(P.lp_accum, lp_acc);
}

template <int W, int DEPTH>
__global__ __launch_bounds__(256) void apply_rows_kernel(const ApplyParams P) {
    apply_rows_body<W, DEPTH>(P, (int64_t)blockIdx.x);
}

// the entity and the relation table of one training step in ONE launch: blocks [0, blocks0) work on P0, the rest on
// P1.  (On separate streams the two launches only slowed each other down; one after the other they pay two launches.)
template <int W, int DEPTH>
__global__ __launch_bounds__(256) void apply_rows_pair_kernel(const ApplyParams P0, const ApplyParams P1, unsigned blocks0) {
    if (blockIdx.x < blocks0) apply_rows_body<W, DEPTH>(P0, (int64_t)blockIdx.x);
    else apply_rows_body<W, DEPTH>(P1, (int64_t)(blockIdx.x - blocks0));
}

// Deferred segments arrive here as BLOCK TASKS (P.long_list): block b = rows [64 b, 64 b + 64) of its segment.
//   one block (33..64 rows): the task's wave sums the rows left to right, 16 in flight per trip, and updates the table row;
//   more blocks            : each task's wave — any wave of the launch, so a hub row's thousands of contributions spread
//       over the whole chip instead of one workgroup — sums its block left to right into the block's partial row; the
//       wave that finishes a segment's LAST block (arrival counter) adds the block sums left to right and applies the
//       optimizer.  The reduction tree is defined by the segment alone, so the bits do not depend on which wave, window
//       or GPU did what (a one-block segment is the same tree: plain left to right, like the window kernel's).
template <int W, int RIF = 16>   // RIF: contribution rows in flight per trip
__device__ __forceinline__ void sum_block(const ApplyParams& P, int64_t u0, int64_t u1, int c, uint32_t my_row, float my_coef,
                                          float (&out)[W], bool carry = false) {
    // (my_row, my_coef): source row and factor of position u0 + lane, loaded by the WHOLE wave in one instruction before
    // the column loop (block_sources) and handed out with v_readlane — one load round trip per block instead of one per
    // 16 rows in front of the row loads that depend on it
    if constexpr (W == 4) {
        float4 acc = carry ? make_float4(out[0], out[1], out[2], out[3]) : make_float4(0.f, 0.f, 0.f, 0.f);   // carry: continue a running sum
        const int n = (int)(u1 - u0);
        int j0 = 0;
        for (; j0 + RIF <= n; j0 += RIF) {
            float4 v[RIF];
            float cf[RIF];
#pragma unroll
            for (int j = 0; j < RIF; ++j) {
                const uint32_t row = (uint32_t)__builtin_amdgcn_readlane((int)my_row, j0 + j);
                cf[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_coef), j0 + j));
                v[j] = *reinterpret_cast<const float4*>(P.contrib + (int64_t)row * P.ldc + 4 * c);
            }
#pragma unroll
            for (int j = 0; j < RIF; ++j) add_scaled(acc, v[j], cf[j]);
        }
        for (; j0 < n; ++j0) {
            const uint32_t row = (uint32_t)__builtin_amdgcn_readlane((int)my_row, j0);
            const float cf = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_coef), j0));
            add_scaled(acc, *reinterpret_cast<const float4*>(P.contrib + (int64_t)row * P.ldc + 4 * c), cf);
        }
        out[0] = acc.x; out[1] = acc.y; out[2] = acc.z; out[3] = acc.w;
    } else {
        float acc = carry ? out[0] : 0.f;
        for (int64_t u = u0; u < u1; ++u) {
            const Src su = contrib_src(P, u);
            add_scaled(acc, P.contrib[(int64_t)su.row * P.ldc + c], su.coef);
        }
        out[0] = acc;
    }
}

// sources of the (at most 64) positions [u0, u1): lane j holds position u0 + j — call with all 64 lanes active
__device__ __forceinline__ Src block_sources(const ApplyParams& P, int64_t u0, int64_t u1, int lane) {
    return u0 + lane < u1 ? contrib_src(P, u0 + lane) : Src{0u, 0.f};
}

__device__ __forceinline__ int64_t segment_end(const ApplyParams& P, int64_t t, uint32_t key, int lane) {
    int64_t end = t;  // first position whose key differs (keys are sorted); wave-uniform result
    for (;;) {
        const int64_t q = end + lane;
        const unsigned long long same = __ballot(q < P.n && P.keys[q] == key);
        if (same == ~0ull) { end += 64; continue; }
        return end + (__ffsll((long long)~same) - 1);
    }
}

// one (block task, column half) — the work of ONE wave.  per = 1: the wave walks every column chunk; per = 2: every other
// group of 64 chunks (half = which), arrivals counted in the two 16-bit fields of the segment's counter.
template <int W, int RIF = 16, int CMB = 8, bool PLAIN = false>   // rows in flight per trip of a block sum / block sums in flight of the combine
__device__ __forceinline__ void long_task_wave(const ApplyParams& P, const OptParams& opt, int32_t step, float* __restrict__ partial,
                                               int64_t ldp, const LongTask tk, unsigned half, unsigned per, int lane,
                                               float& lp_acc) {
    const int nchunks = P.k_int / W;
    const int col0 = 64 * (int)half, cstep = 64 * (int)per;
    if (tk.len == 0) return;
    const int64_t t = tk.head, end = t + tk.len;
    const uint32_t key = P.keys[t];
    if ((int64_t)key >= P.n_rows) return;  // defensive: never write outside the table
    float* wrow = P.table + (int64_t)key * P.ld;
    float* s0row = (!PLAIN && P.state0) ? P.state0 + (int64_t)key * P.ld : nullptr;
    float* s1row = (!PLAIN && P.state1) ? P.state1 + (int64_t)key * P.ld : nullptr;
    auto update = [&](int c, float (&g)[W]) {   // optimizer step of columns [W c, W c + W) with summed gradient g
        float w[W], a0[W], a1[W];
        if constexpr (W == 4) {   // table row and state as 16-byte accesses
            const float4 wv = *reinterpret_cast<const float4*>(wrow + 4 * c);
            w[0] = wv.x; w[1] = wv.y; w[2] = wv.z; w[3] = wv.w;
            if (s0row) { const float4 t4 = *reinterpret_cast<const float4*>(s0row + 4 * c); a0[0] = t4.x; a0[1] = t4.y; a0[2] = t4.z; a0[3] = t4.w; }
            if (s1row) { const float4 t4 = *reinterpret_cast<const float4*>(s1row + 4 * c); a1[0] = t4.x; a1[1] = t4.y; a1[2] = t4.z; a1[3] = t4.w; }
        } else {
            w[0] = wrow[c];
            if (s0row) a0[0] = s0row[c];
            if (s1row) a1[0] = s1row[c];
        }
#pragma unroll
        for (int j = 0; j < W; ++j) {
            lp_fold(opt, w[j], g[j], lp_acc);
            opt_update_elem(opt, w[j], g[j], &a0[j], &a1[j]);
        }
        if constexpr (W == 4) {
            *reinterpret_cast<float4*>(wrow + 4 * c) = make_float4(w[0], w[1], w[2], w[3]);
            if (s0row) *reinterpret_cast<float4*>(s0row + 4 * c) = make_float4(a0[0], a0[1], a0[2], a0[3]);
            if (s1row) *reinterpret_cast<float4*>(s1row + 4 * c) = make_float4(a1[0], a1[1], a1[2], a1[3]);
        } else {
            wrow[c] = w[0];
            if (s0row) s0row[c] = a0[0];
            if (s1row) s1row[c] = a1[0];
        }
    };
    const int64_t nblk = (tk.len + kLongSegment - 1) / kLongSegment;
    if (nblk == 1) {
        const Src mine = block_sources(P, t, end, lane);
        for (int c = lane + col0; c < nchunks; c += cstep) {
            float g[W];
            sum_block<W, RIF>(P, t, end, c, mine.row, mine.coef, g);
            update(c, g);
        }
        if (P.tag && lane == 0 && half == 0) P.tag[key] = step;
        return;
    }
    // partial row of the block that starts at sorted position u0: 2 * (u0 / 64) + (first block of its segment).
    // Collision-free: a 64-aligned bucket of positions holds at most one non-first block start and one
    // first-block start (a long segment spans more than 64 positions) — and for the same reason t / 64 names the
    // segment's arrival counter.
    const int64_t u0 = t + (int64_t)tk.block * kLongSegment, u1 = min(u0 + kLongSegment, end);
    float* prow = partial + (2 * (u0 / kLongSegment) + (tk.block == 0 ? 1 : 0)) * ldp;
    const Src mine = block_sources(P, u0, u1, lane);
    for (int c = lane + col0; c < nchunks; c += cstep) {
        float g[W];
        sum_block<W, RIF>(P, u0, u1, c, mine.row, mine.coef, g);
        if constexpr (W == 4) store4_through(prow + 4 * c, g[0], g[1], g[2], g[3]);
        else __hip_atomic_store(prow + c, g[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    wait_memory();   // this wave's block sum has left for memory
    // arrivals of the two column halves are counted in the two 16-bit fields of the segment's counter
    // (split only below 2^22 contributions: a segment then has fewer than 65536 blocks)
    const int shift = 16 * (int)half;
    int before = 0;
    if (lane == 0) before = atomicAdd(P.arrive + t / kLongSegment, 1 << shift);
    before = (__builtin_amdgcn_readfirstlane(before) >> shift) & 0xffff;
    if (before != (int)nblk - 1) return;
    // the segment's last block (of this column half): every block sum is in memory
    if (lane == 0) atomicSub(P.arrive + t / kLongSegment, (int)nblk << shift);   // ready for the next apply on this workspace
    const int64_t m0 = t / kLongSegment;             // (t + 64 b) / 64 = t / 64 + b
    if constexpr (W == 4) {
        // two chunks per lane x CMB block sums per trip in flight, added left to right: 0 + first block sum + ...
        for (int c0 = 0; c0 < nchunks; c0 += 128) {
            const int ca = c0 + lane, cb = c0 + 64 + lane;
            const bool oa = ca < nchunks && (per == 1u || half == 0u), ob = cb < nchunks && (per == 1u || half == 1u);
            vfloat4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
            const float* first = partial + (2 * m0 + 1) * ldp;
            if (oa) a = load4_through(first + 4 * ca);
            if (ob) b = load4_through(first + 4 * cb);
            wait_memory();
            landed(a); landed(b);
            for (int64_t bq = 1; bq < nblk; bq += CMB) {
                const int cnt = (int)min((int64_t)CMB, nblk - bq);
                vfloat4 va[CMB], vb[CMB];
#pragma unroll
                for (int j = 0; j < CMB; ++j) {
                    va[j] = vfloat4{0.f, 0.f, 0.f, 0.f}; vb[j] = va[j];
                    if (j < cnt) {
                        const float* r = partial + 2 * (m0 + bq + j) * ldp;
                        if (oa) va[j] = load4_through(r + 4 * ca);
                        if (ob) vb[j] = load4_through(r + 4 * cb);
                    }
                }
                wait_memory();
#pragma unroll
                for (int j = 0; j < CMB; ++j) {
                    landed(va[j]); landed(vb[j]);
                    if (j < cnt) { a += va[j]; b += vb[j]; }
                }
            }
            float ga[4] = {a.x, a.y, a.z, a.w}, gb[4] = {b.x, b.y, b.z, b.w};
            if (oa) update(ca, ga);
            if (ob) update(cb, gb);
        }
    } else {
        for (int c = lane + col0; c < nchunks; c += cstep) {
            float acc[W];
            float a = __hip_atomic_load(partial + (2 * m0 + 1) * ldp + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int64_t bq = 1; bq < nblk; ++bq)
                a += __hip_atomic_load(partial + 2 * (m0 + bq) * ldp + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc[0] = a;
            update(c, acc);
        }
    }
    if (P.tag && lane == 0 && half == 0) P.tag[key] = step;
}

// Rows wider than 64 column chunks: TWO waves per task, each with every other group of 64 chunks — a task is a chain
// of dependent loads (source row index, then 16 rows at a time), so the second wave halves its length instead of
// walking the rows a second time (measured on the Zipf batch: see DESIGN.md 4.1).  Column sums are unchanged.
__device__ __forceinline__ unsigned waves_per_task(const ApplyParams& P, int nchunks) {
    return (nchunks > 64 && P.n < ((int64_t)1 << 22)) ? 2u : 1u;   // (block counts stay below 2^16: see the arrival fields)
}

template <int W>
__device__ __forceinline__ void apply_long_body(const ApplyParams& P, float* __restrict__ partial, int64_t ldp, unsigned block,
                                                unsigned n_blocks) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    const unsigned n_tasks = min(*P.long_count, P.long_cap);
    float lp_acc = 0.f;
    const unsigned per = waves_per_task(P, P.k_int / W);
    for (unsigned i2 = block * nwv + wv; i2 < n_tasks * per; i2 += n_blocks * nwv)   // one (task, column half) per wave at a time
        long_task_wave<W>(P, P.opt, P.step, partial, ldp, P.long_list[i2 / per], i2 % per, per, lane, lp_acc);
    if (P.opt.lp_lambda != 0.f) wave_add_double(P.lp_accum, lp_acc);
    // the last workgroup to finish empties the list, so that a second emg_apply_grouped on the same grouping (or the
    // next batch that reuses the workspace) starts from zero; every workgroup has read the count by then
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(P.long_count + 1, 1u) == n_blocks - 1) { P.long_count[0] = 0u; P.long_count[1] = 0u; }
    }
}

template <int W>
__global__ __launch_bounds__(1024) void apply_long_kernel(const ApplyParams P, float* __restrict__ partial, int64_t ldp) {
    apply_long_body<W>(P, partial, ldp, blockIdx.x, gridDim.x);
}

template <int W>
__global__ __launch_bounds__(1024) void apply_long_pair_kernel(const ApplyParams P0, float* __restrict__ partial0, int64_t ldp0,
                                                               unsigned blocks0, const ApplyParams P1,
                                                               float* __restrict__ partial1, int64_t ldp1) {
    if (blockIdx.x < blocks0) apply_long_body<W>(P0, partial0, ldp0, blockIdx.x, blocks0);
    else apply_long_body<W>(P1, partial1, ldp1, blockIdx.x - blocks0, gridDim.x - blocks0);
}

// Variant of apply_rows_kernel for SKINNY rows (<= 16 sixteen-byte chunks: a column slab of an 8-GPU job, TransE
// k <= 64): LPS lanes work on one segment (16-byte chunk c of the row on lane c % LPS), so a wave processes 64/LPS segments
// at once — with all 64 lanes on one segment a 200-byte row keeps 13 lanes busy (0.81 -> 0.55 ms on the 8-rank
// share).  Wider rows stay on apply_rows_kernel: its contribution indices come from v_readlane (wave-uniform),
// here they need a ds_bpermute per row, which costs more than the idle lanes once a row fills >= 25 lanes.
template <int W, int LPS>
__global__ __launch_bounds__(256) void apply_rows_sub_kernel(const ApplyParams P) {
    constexpr int NSUB = 64 / LPS;
    constexpr unsigned long long SUBMASK = LPS == 64 ? ~0ull : ((1ull << (LPS & 63)) - 1ull);
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPS, sl = lane % LPS;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t w0 = wave * P.win;
    const int64_t t0 = w0 + lane;
    const bool in = lane < P.win && t0 < P.n;
    const uint32_t mykey = in ? P.keys[t0] : 0u;
    const Src mysrc = in ? contrib_src(P, t0) : Src{0u, 0.f};   // the window's contribution rows: one coalesced load
    const bool head = in && (t0 == 0 || P.keys[t0 - 1] != mykey);
    const bool last = in && (t0 + 1 == P.n || P.keys[t0 + 1] != mykey);
    const unsigned long long heads = __ballot(head);
    const unsigned long long todo = __ballot(head && !(P.skip_single && last));
    const int64_t wend = min(w0 + (int64_t)P.win, P.n);  // end of this window
    const int nchunks = P.k_int / W;
    const int ntodo = __popcll(todo);
    float lp_acc = 0.f;
    for (int it = 0; it * NSUB < ntodo; ++it) {
        // this round's heads: ranks it*NSUB .. it*NSUB+NSUB-1 among the todo bits; subgroup `sub` takes the sub-th
        unsigned long long m = todo;
        for (int j = 0; j < it * NSUB + sub && m; ++j) m &= m - 1;
        const bool has0 = m != 0ull;
        const int b = has0 ? __ffsll((long long)m) - 1 : 0;
        const int64_t t = w0 + b;
        const uint32_t key = __shfl(mykey, b, 64);
        const bool has = has0 && (int64_t)key < P.n_rows;  // defensive: never write outside the table
        // segment end: the next head inside the window (ballot, no memory traffic), else scan on past the window
        const unsigned long long after = heads & ~((2ull << b) - 1ull);
        int64_t end = after ? w0 + (__ffsll((long long)after) - 1) : wend;
        bool open = has && !after;
        while (__ballot(open)) {  // LPS keys per trip (a serial scan costs one dependent load per row of a long segment)
            const int64_t q = end + sl;
            const unsigned long long same = __ballot(open && q < P.n && P.keys[q] == key);
            const unsigned long long mine = (same >> (sub * LPS)) & SUBMASK;
            if (open) {
                if (mine == SUBMASK) {
                    end += LPS;
                } else {
                    end += __ffsll((long long)~mine) - 1;
                    open = false;
                }
            }
        }
        if (!has) end = t;  // nothing to do for this subgroup in this round
        auto source = [&](int64_t u) -> Src {  // u is uniform inside a subgroup
            const int from = (int)(u < wend ? u - w0 : 0);
            const Src inwin{(uint32_t)__shfl(mysrc.row, from, 64), __shfl(mysrc.coef, from, 64)};
            return u < wend ? inwin : contrib_src(P, u);
        };
        float* wrow = P.table + (int64_t)key * P.ld;
        float* s0row = P.state0 ? P.state0 + (int64_t)key * P.ld : nullptr;
        float* s1row = P.state1 ? P.state1 + (int64_t)key * P.ld : nullptr;
        if constexpr (W == 4) {
            // two row chunks per lane x two contributions per trip (eight for long segments): independent 16-byte
            // loads in flight, added in contribution order (bit-reproducible sums).  The trip counts depend on the
            // subgroup's segment: the shuffles inside source() need every lane, so all subgroups run the
            // longest trip count and idle ones repeat their last row into a discarded sum.
            for (int c0 = 0; c0 < nchunks; c0 += 2 * LPS) {
                const int ca = c0 + sl, cb = c0 + LPS + sl;
                const bool oa = has && ca < nchunks, ob = has && cb < nchunks;
                float4 accA = make_float4(0.f, 0.f, 0.f, 0.f), accB = accA, wA = accA, wB = accA;
                if (oa) wA = *reinterpret_cast<const float4*>(wrow + 4 * ca);
                if (ob) wB = *reinterpret_cast<const float4*>(wrow + 4 * cb);
                int64_t u = t;
                while (__ballot(u + 8 <= end)) {  // (wave-uniform loop: see above)
                    const bool act = u + 8 <= end;
                    float4 va[8], vb[8];
                    float cf[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const Src sj = source(act ? u + j : t);
                        const float* rj = P.contrib + (int64_t)sj.row * P.ldc;
                        cf[j] = sj.coef;
                        va[j] = (act && oa) ? *reinterpret_cast<const float4*>(rj + 4 * ca) : accA;
                        vb[j] = (act && ob) ? *reinterpret_cast<const float4*>(rj + 4 * cb) : accA;
                    }
                    if (act) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {  // added in contribution order
                            if (oa) add_scaled(accA, va[j], cf[j]);
                            if (ob) add_scaled(accB, vb[j], cf[j]);
                        }
                        u += 8;
                    }
                }
                while (__ballot(u < end)) {
                    const bool act = u < end, two = u + 1 < end;
                    const Src s0 = source(act ? u : t), s1 = source(two ? u + 1 : t);
                    const float* r0 = P.contrib + (int64_t)s0.row * P.ldc;
                    const float* r1 = P.contrib + (int64_t)s1.row * P.ldc;
                    float4 v0a = accA, v0b = accA, v1a = accA, v1b = accA;
                    if (act && oa) v0a = *reinterpret_cast<const float4*>(r0 + 4 * ca);
                    if (act && ob) v0b = *reinterpret_cast<const float4*>(r0 + 4 * cb);
                    if (two && oa) v1a = *reinterpret_cast<const float4*>(r1 + 4 * ca);
                    if (two && ob) v1b = *reinterpret_cast<const float4*>(r1 + 4 * cb);
                    if (act && oa) add_scaled(accA, v0a, s0.coef);
                    if (act && ob) add_scaled(accB, v0b, s0.coef);
                    if (two && oa) add_scaled(accA, v1a, s1.coef);
                    if (two && ob) add_scaled(accB, v1b, s1.coef);
                    if (act) u += 2;
                }
                auto finish = [&](int c, float4 wv, const float4& g) {
                    const int64_t off = 4 * (int64_t)c;
                    float w[4] = {wv.x, wv.y, wv.z, wv.w};
                    float gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        lp_fold(P.opt, w[j], gg[j], lp_acc);
                        opt_update_elem(P.opt, w[j], gg[j], s0row ? s0row + off + j : nullptr, s1row ? s1row + off + j : nullptr);
                    }
                    *reinterpret_cast<float4*>(wrow + off) = make_float4(w[0], w[1], w[2], w[3]);
                };
                if (oa) finish(ca, wA, accA);
                if (ob) finish(cb, wB, accB);
            }
        } else {
            for (int c0 = 0; c0 < nchunks; c0 += LPS) {
                const int c = c0 + sl;
                const bool oc = has && c < nchunks;
                float acc = 0.f;
                int64_t u = t;
                while (__ballot(u < end)) {
                    const bool act = u < end;
                    const Src su = source(act ? u : t);
                    if (act && oc) add_scaled(acc, P.contrib[(int64_t)su.row * P.ldc + c], su.coef);
                    if (act) ++u;
                }
                if (oc) {
                    float wv = wrow[c];
                    lp_fold(P.opt, wv, acc, lp_acc);
                    opt_update_elem(P.opt, wv, acc, s0row ? s0row + c : nullptr, s1row ? s1row + c : nullptr);
                    wrow[c] = wv;
                }
            }
        }
        if (P.tag && sl == 0 && has) P.tag[key] = P.step;
    }
    if (P.opt.lp_lambda != 0.f) wave_add_double(P.lp_accum, lp_acc);
}

// Rows no contribution of this step touched (tag[r] != step).  Two reasons to visit them:
//   * Keras Adam's sparse apply is dense-equivalent: every row's m, v decay and every row moves (g = 0 there);
//   * an LP regulariser makes the gradient dense (regularizers/lp.py:107-113: the penalty covers the FULL tables,
//     EmbeddingModel.py:818-820): an untouched row still has g = lambda * p * |w|^(p-1) * sign(w), and its |w|^p
//     belongs to the loss.  Touched rows got the same term folded into their update (lp_fold), so the regulariser
//     costs ONE pass over the rows nothing else visited instead of n_rows extra contribution rows.
// one untouched row: the optimizer's update with g = 0 (+ the regulariser's gradient), a wave per row
__device__ __forceinline__ void untouched_row_update(const ApplyParams& P, int64_t r, int lane, bool vec, float& lp_acc) {
    float* w = P.table + r * P.ld;
    float* s0 = P.state0 ? P.state0 + r * P.ld : nullptr;
    float* s1 = P.state1 ? P.state1 + r * P.ld : nullptr;
    if (vec) {   // 16-byte chunks: table row and state rows as float4 (the pass is pure bandwidth)
        for (int c = lane; 4 * c < P.k_int; c += 64) {
            float4 wv = *reinterpret_cast<const float4*>(w + 4 * c);
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
            if (s0) a = *reinterpret_cast<const float4*>(s0 + 4 * c);
            if (s1) b = *reinterpret_cast<const float4*>(s1 + 4 * c);
            float ww[4] = {wv.x, wv.y, wv.z, wv.w}, aa[4] = {a.x, a.y, a.z, a.w}, bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float g = 0.f;
                lp_fold(P.opt, ww[j], g, lp_acc);
                opt_update_elem(P.opt, ww[j], g, &aa[j], &bb[j]);
            }
            *reinterpret_cast<float4*>(w + 4 * c) = make_float4(ww[0], ww[1], ww[2], ww[3]);
            if (s0) *reinterpret_cast<float4*>(s0 + 4 * c) = make_float4(aa[0], aa[1], aa[2], aa[3]);
            if (s1) *reinterpret_cast<float4*>(s1 + 4 * c) = make_float4(bb[0], bb[1], bb[2], bb[3]);
        }
        return;
    }
    for (int c = lane; c < P.k_int; c += 64) {
        float wv = w[c], g = 0.f;
        lp_fold(P.opt, wv, g, lp_acc);
        opt_update_elem(P.opt, wv, g, s0 ? s0 + c : nullptr, s1 ? s1 + c : nullptr);
        w[c] = wv;
    }
}
__device__ __forceinline__ bool untouched_vec(const ApplyParams& P) {
    return (P.k_int % 4 == 0) && (P.ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(P.table) & 15u) == 0) &&
           (!P.state0 || (reinterpret_cast<uintptr_t>(P.state0) & 15u) == 0) &&
           (!P.state1 || (reinterpret_cast<uintptr_t>(P.state1) & 15u) == 0);
}

__device__ __forceinline__ void untouched_rows_body(const ApplyParams& P0, int64_t block, int64_t n_blocks) {
    const int lane = threadIdx.x & 63;
    float lp_acc = 0.f;
    ApplyParams P = P0;
    if (P0.ctl) {   // graph node: step number and learning rates from the device record
        const float* h = P0.which ? P0.ctl->hyper_rel : P0.ctl->hyper_ent;
        P.opt.lr = h[0]; P.opt.lr_t = h[5]; P.step = P0.ctl->step;
    }
    // a wave per row, grid-stride: the launch is capped (a wave per row of a 1M-row table made 1M atomics on the ONE
    // double that accumulates sum |w|^p — 9 ms per step of the LP-regularised C3 before; now one atomic per wave of a
    // few thousand)
    const int64_t nw = (n_blocks * blockDim.x) >> 6;
    const bool vec = untouched_vec(P);
    for (int64_t r = (block * blockDim.x + threadIdx.x) >> 6; r < P.n_rows; r += nw) {
        if (P.tag[r] == P.step) continue;
        untouched_row_update(P, r, lane, vec, lp_acc);
    }
    if (P.opt.lp_lambda != 0.f) block_add_double(P.lp_accum, lp_acc);   // (every thread of the workgroup arrives here)
}

// ---------------------------------------------------------------------------------------------------------------
// Deferred dense pass (include/emgraph_hip.h: emg_deferred_catchup / emg_deferred_materialize): replay the dense pass's
// update (Keras Adam's decay, the LP regulariser's gradient) of the steps a row has missed, with each step's own learning rate
// ---------------------------------------------------------------------------------------------------------------
struct ReplayParams {
    float* table; float* s0; float* s1; int32_t* tag; int64_t n_rows, ld; int32_t k_int; OptParams opt;
    const float* lr_hist; int32_t upto; double* lp_accum;
    int32_t lag;   // 1 (Adam, no regulariser): multi / single destinations get w only; the apply redoes the decay of m, v (ApplyParams.state_lag)
    const Seg* multi; const uint32_t* single; const LongTask* tasks; const uint32_t* keys; const uint32_t* counters; uint32_t task_cap;
    const uint32_t* vals; int64_t single_from;   // single_from >= 0: singletons whose contribution slot is >= it are not this pass's (the scoring kernel replays them)
};

// one row, steps tag[r]+1 .. upto: what untouched_rows_body does to it in each of them (g = the regulariser's gradient alone).
// OPT / LPK (0: no regulariser, 1: p in {1, 2, 3}, 2: any p, 3: p == 2 — two multiplications per step, same bits) are compile-time: the replay is ALU work — as many row-steps as
// the dense pass visits, only without their memory traffic — and the generic optimizer switch / powf cost it twice the time
template <int OPT, int LPK>
__device__ __forceinline__ void replay_row(const ReplayParams& P, int64_t r, int lane, float& lp_acc) {
    const int32_t from = P.tag[r];
    if (from >= P.upto) return;
    if (from == 0 && LPK == 0) {   // never written, no regulariser: zero gradient on the initial state moves nothing
        if (lane == 0) P.tag[r] = P.upto;
        return;
    }
    float* w = P.table + r * P.ld;
    float* s0 = P.s0 ? P.s0 + r * P.ld : nullptr;
    float* s1 = P.s1 ? P.s1 + r * P.ld : nullptr;
    OptParams opt = P.opt;
    opt.opt = OPT;
    if (LPK == 0) opt.lp_lambda = 0.f;
    const bool vec = (P.k_int % 4 == 0) && (P.ld % 4 == 0);
    for (int c = lane; (vec ? 4 * c : c) < P.k_int; c += 64) {
        float ww[4] = {0.f, 0.f, 0.f, 0.f}, aa[4] = {0.f, 0.f, 0.f, 0.f}, bb[4] = {0.f, 0.f, 0.f, 0.f};
        const int n = vec ? 4 : 1;
        if (vec) {
            const float4 wv = *reinterpret_cast<const float4*>(w + 4 * c);
            ww[0] = wv.x; ww[1] = wv.y; ww[2] = wv.z; ww[3] = wv.w;
            if (s0) { const float4 a = *reinterpret_cast<const float4*>(s0 + 4 * c); aa[0] = a.x; aa[1] = a.y; aa[2] = a.z; aa[3] = a.w; }
            if (s1) { const float4 b = *reinterpret_cast<const float4*>(s1 + 4 * c); bb[0] = b.x; bb[1] = b.y; bb[2] = b.z; bb[3] = b.w; }
        } else {
            ww[0] = w[c];
            if (s0) aa[0] = s0[c];
            if (s1) bb[0] = s1[c];
        }
        for (int32_t st = from + 1; st <= P.upto; ++st) {
            opt.lr = opt.lr_t = P.lr_hist[st];   // (Adam reads lr_t, the others lr: the table holds what that step's update used)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < n) {
                    float g = 0.f;
                    if constexpr (LPK == 1) lp_fold_p123(opt, ww[j], g, lp_acc);
                    else if constexpr (LPK == 3) lp_fold_p2(opt, ww[j], g, lp_acc);
                    else if constexpr (LPK == 2) lp_fold(opt, ww[j], g, lp_acc);
                    opt_update_elem(opt, ww[j], g, &aa[j], &bb[j]);
                }
            }
        }
        if (vec) {
            *reinterpret_cast<float4*>(w + 4 * c) = make_float4(ww[0], ww[1], ww[2], ww[3]);
            if (s0) *reinterpret_cast<float4*>(s0 + 4 * c) = make_float4(aa[0], aa[1], aa[2], aa[3]);
            if (s1) *reinterpret_cast<float4*>(s1 + 4 * c) = make_float4(bb[0], bb[1], bb[2], bb[3]);
        } else {
            w[c] = ww[0];
            if (s0) s0[c] = aa[0];
            if (s1) s1[c] = bb[0];
        }
    }
    if (lane == 0) P.tag[r] = P.upto;
}

// the rows a prepared batch will read and update = the destinations of its grouping's lists (each exactly once).  A wave takes
// a contiguous stretch of items, 64 at a time: lane k fetches item k's destination and its tag in one go (two dependent
// loads for 64 rows instead of two per row), then the rows that missed a step are replayed one after the other
template <int OPT, int LPK>
__global__ __launch_bounds__(256) void deferred_catchup_kernel(const ReplayParams P) {
    float lp_acc = 0.f;
    const int lane = threadIdx.x & 63;
    const int64_t gw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    // (single_from == 0: every singleton is the scoring kernel's — none in this pass's item list, so that the waves' shares hold real items)
    const int64_t n_multi = P.counters[GC_MULTI], n_single = P.single_from == 0 ? 0 : P.counters[GC_SINGLE];
    const int64_t n_tasks = min(P.counters[GC_TASKS], P.task_cap);
    const int64_t total = n_multi + n_single + n_tasks;
    const int64_t share = (total + nw - 1) / nw;
    const int64_t i0 = gw * share, i1 = min(total, i0 + share);
    for (int64_t base = i0; base < i1; base += 64) {
        const int64_t i = base + lane;
        int64_t dest = -1;
        if (i < i1) {
            if (i < n_multi) dest = P.multi[i].dest;
            else if (i < n_multi + n_single) {
                const uint32_t at = P.single[i - n_multi];
                if (!(P.single_from >= 0 && (int64_t)P.vals[at] >= P.single_from)) dest = P.keys[at];
            } else {
                const LongTask tk = P.tasks[i - n_multi - n_single];
                if (tk.block == 0u) dest = P.keys[tk.head];   // one entry per long segment: its first block's
            }
        }
        const bool due = dest >= 0 && dest < P.n_rows && P.tag[dest] < P.upto;
        unsigned long long todo = __ballot(due);
        while (todo) {
            const int k = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            replay_row<OPT, LPK>(P, __shfl(dest, k, 64), lane, lp_acc);
        }
    }
    if (LPK != 0) block_add_double(P.lp_accum, lp_acc);
}

// ---- the same, rows pipelined (16-byte rows of at most 64 T chunks).  The one-row-at-a-time form above is a chain of dependent
// latencies per row (tag -> row loads -> per-step learning-rate loads -> stores, then the next chunk trip): 317 us for C3's
// entity table with SGD + LP where the bytes take 140.  Here a row's loads (all trips, all state rows, its 64 next learning
// rates: lane l holds step from+1+l) are issued while the row before it is replayed; nothing in the loop is conditional —
// lanes past the row's end clamp to its last chunk and redo that chunk's work, storing the same bytes — so the compiler's
// s_waitcnt leaves the next row's loads in flight.  Rows that missed more than 64 steps take the generic path afterwards.
template <int OPT, int T>
struct ReplayRegs {
    float4 w[T], a[T], b[T];
    float lr;
};
template <int OPT, int T>
__device__ __forceinline__ void replay_issue(const ReplayParams& P, ReplayRegs<OPT, T>& R, int32_t row, int32_t from, int lane, int nchunks) {
    const int64_t base = (int64_t)row * P.ld;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int c = min(lane + 64 * t, nchunks - 1);
        R.w[t] = *reinterpret_cast<const float4*>(P.table + base + 4 * c);
        if constexpr (OPT != EMG_OPT_SGD) R.a[t] = *reinterpret_cast<const float4*>(P.s0 + base + 4 * c);
        if constexpr (OPT == EMG_OPT_ADAM) R.b[t] = *reinterpret_cast<const float4*>(P.s1 + base + 4 * c);
    }
    R.lr = P.lr_hist[min(from + 1 + lane, P.upto)];
}
template <int OPT, int LPK, int T, bool LAG>
__device__ __forceinline__ void replay_finish(const ReplayParams& P, ReplayRegs<OPT, T>& R, int32_t row, int32_t from, int lane, int nchunks,
                                              float& lp_acc) {
    OptParams opt = P.opt;
    opt.opt = OPT;
    if (LPK == 0) opt.lp_lambda = 0.f;
    const int n = P.upto - from;   // 1..64, the same for the whole wave
    float ww[T][4], aa[T][4], bb[T][4];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        ww[t][0] = R.w[t].x; ww[t][1] = R.w[t].y; ww[t][2] = R.w[t].z; ww[t][3] = R.w[t].w;
        if constexpr (OPT != EMG_OPT_SGD) { aa[t][0] = R.a[t].x; aa[t][1] = R.a[t].y; aa[t][2] = R.a[t].z; aa[t][3] = R.a[t].w; }
        else { aa[t][0] = aa[t][1] = aa[t][2] = aa[t][3] = 0.f; }
        if constexpr (OPT == EMG_OPT_ADAM) { bb[t][0] = R.b[t].x; bb[t][1] = R.b[t].y; bb[t][2] = R.b[t].z; bb[t][3] = R.b[t].w; }
        else { bb[t][0] = bb[t][1] = bb[t][2] = bb[t][3] = 0.f; }
    }
    float acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = 0.f;
    for (int i = 0; i < n; ++i) {
        opt.lr = opt.lr_t = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(R.lr), i));
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (OPT == EMG_OPT_ADAM && LPK == 0) {
                    adam_zero_grad_elem(opt, ww[t][j], aa[t][j], bb[t][j]);
                } else {
                    float g = 0.f;
                    if constexpr (LPK == 1) lp_fold_p123(opt, ww[t][j], g, acc[t]);
                    else if constexpr (LPK == 3) lp_fold_p2(opt, ww[t][j], g, acc[t]);
                    opt_update_elem(opt, ww[t][j], g, &aa[t][j], &bb[t][j]);
                }
            }
        }
    }
    const int64_t base = (int64_t)row * P.ld;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const bool on = lane + 64 * t < nchunks;
        const int c = min(lane + 64 * t, nchunks - 1);
        *reinterpret_cast<float4*>(P.table + base + 4 * c) = make_float4(ww[t][0], ww[t][1], ww[t][2], ww[t][3]);
        if constexpr (OPT != EMG_OPT_SGD && !LAG) *reinterpret_cast<float4*>(P.s0 + base + 4 * c) = make_float4(aa[t][0], aa[t][1], aa[t][2], aa[t][3]);
        if constexpr (OPT == EMG_OPT_ADAM && !LAG) *reinterpret_cast<float4*>(P.s1 + base + 4 * c) = make_float4(bb[t][0], bb[t][1], bb[t][2], bb[t][3]);
        if (LPK != 0) lp_acc += on ? acc[t] : 0.f;
    }
}

template <int OPT, int LPK, int T, bool LAG = false>
__global__ __launch_bounds__(256) void deferred_catchup_rows_kernel(const ReplayParams P) {
    float lp_acc = 0.f;
    const int lane = threadIdx.x & 63;
    const int nchunks = P.k_int / 4;
    const int64_t gw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    // (single_from == 0: every singleton is the scoring kernel's — none in this pass's item list, so that the waves' shares hold real items)
    const int64_t n_multi = P.counters[GC_MULTI], n_single = P.single_from == 0 ? 0 : P.counters[GC_SINGLE];
    const int64_t n_tasks = min(P.counters[GC_TASKS], P.task_cap);
    const int64_t total = n_multi + n_single + n_tasks;
    const int64_t share = (total + nw - 1) / nw;
    const int64_t i0 = gw * share, i1 = min(total, i0 + share);
    for (int64_t base = i0; base < i1; base += 64) {
        const int64_t i = base + lane;
        int32_t dest = -1;
        if (i < i1) {
            if (i < n_multi) dest = (int32_t)P.multi[i].dest;
            else if (i < n_multi + n_single) {
                const uint32_t at = P.single[i - n_multi];
                if (!(P.single_from >= 0 && (int64_t)P.vals[at] >= P.single_from)) dest = (int32_t)P.keys[at];
            } else {
                const LongTask tk = P.tasks[i - n_multi - n_single];
                if (tk.block == 0u) dest = (int32_t)P.keys[tk.head];
            }
        }
        const bool valid = dest >= 0 && dest < P.n_rows;
        const int32_t from = valid ? P.tag[dest] : P.upto;
        const bool due = from < P.upto;
        const bool moves = due && !(from == 0 && LPK == 0);   // (never written, no regulariser: zero gradient on the initial state moves nothing)
        // LAG: only w of a multi / single destination is written — m, v and the tag stay as of `from`, segment_update redoes
        // their decay; the heads of block tasks (several waves finish those rows) take the generic replay, complete
        const bool fast = moves && P.upto - from <= 64 && (!LAG || i < n_multi + n_single);
        if (due && ((fast && !LAG) || !moves)) P.tag[dest] = P.upto;    // (each destination is in the lists once: nobody else looks at this tag here)
        unsigned long long todo = __ballot(fast);
        if (todo) {
            ReplayRegs<OPT, T> ra, rb;   // two rows' registers, taken in turns (no copies: a copy would wait for the loads it hides)
            int k = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            int32_t row = __builtin_amdgcn_readlane(dest, k), fr = __builtin_amdgcn_readlane(from, k);
            replay_issue<OPT, T>(P, ra, row, fr, lane, nchunks);
            for (;;) {
                bool more = todo != 0ull;
                k = more ? __ffsll((long long)todo) - 1 : k;   // (no row left: the same row once more, unused — the loop stays branch-free)
                todo &= todo - 1ull;
                const int32_t rown = __builtin_amdgcn_readlane(dest, k), frn = __builtin_amdgcn_readlane(from, k);
                replay_issue<OPT, T>(P, rb, rown, frn, lane, nchunks);
                replay_finish<OPT, LPK, T, LAG>(P, ra, row, fr, lane, nchunks, lp_acc);
                if (!more) break;
                more = todo != 0ull;
                k = more ? __ffsll((long long)todo) - 1 : k;
                todo &= todo - 1ull;
                row = __builtin_amdgcn_readlane(dest, k); fr = __builtin_amdgcn_readlane(from, k);
                replay_issue<OPT, T>(P, ra, row, fr, lane, nchunks);
                replay_finish<OPT, LPK, T, LAG>(P, rb, rown, frn, lane, nchunks, lp_acc);
                if (!more) break;
            }
        }
        unsigned long long slow = __ballot(moves && !fast);
        while (slow) {
            const int k = __ffsll((long long)slow) - 1;
            slow &= slow - 1ull;
            replay_row<OPT, LPK>(P, (int64_t)__builtin_amdgcn_readlane(dest, k), lane, lp_acc);
        }
    }
    if (LPK != 0) block_add_double(P.lp_accum, lp_acc);
}

template <int OPT, int LPK>
__global__ __launch_bounds__(256) void deferred_materialize_kernel(const ReplayParams P) {
    const int lane = threadIdx.x & 63;
    const int64_t gw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    float lp_acc = 0.f;
    for (int64_t r = gw; r < P.n_rows; r += nw) replay_row<OPT, LPK>(P, r, lane, lp_acc);
    if (LPK != 0) block_add_double(P.lp_accum, lp_acc);
}

static inline unsigned untouched_blocks(int64_t n_rows) {
    const int64_t b = cdiv(n_rows * 64, 256);
    return (unsigned)(b < 4096 ? b : 4096);
}

__global__ __launch_bounds__(256) void untouched_rows_kernel(const ApplyParams P) { untouched_rows_body(P, (int64_t)blockIdx.x, (int64_t)gridDim.x); }

// both tables' dense passes in one launch (a launch is ~7 us of a 0.1 ms small-batch step)
__global__ __launch_bounds__(256) void untouched_rows_pair_kernel(const ApplyParams P0, const ApplyParams P1, unsigned blocks0) {
    if (blockIdx.x < blocks0) untouched_rows_body(P0, (int64_t)blockIdx.x, (int64_t)blocks0);
    else untouched_rows_body(P1, (int64_t)(blockIdx.x - blocks0), (int64_t)(gridDim.x - blocks0));
}

// ---------------------------------------------------------------------------------------------------------------
// The training path: descriptor-driven apply (counting grouping).  A persistent grid; every wave
//   1. takes its share of the BLOCK TASKS (round robin: they are the heavy items), as apply_long_kernel does;
//   2. takes a CONTIGUOUS stretch of the segment list (destinations with 2..32 contributions, then — unless the backward
//      kernel updated them in place — the singletons): its descriptors arrive in ONE coalesced load (lane k = item k),
//      the source rows / factors of item k + 1 are fetched while item k's rows are in flight, and a segment is
//      table row + up to 4 contribution rows in flight per trip, summed in contribution order (same bits as ever).
// Nothing is searched: no key loads, no ballots, no window preamble (DESIGN.md 4.1 has the before / after).
// ---------------------------------------------------------------------------------------------------------------
#ifndef EMG_SEG_DEPTH
#define EMG_SEG_DEPTH 4
#endif
__device__ __forceinline__ Src segment_sources(const ApplyParams& P, uint32_t start, uint32_t len, int lane) {
    return (uint32_t)lane < len ? contrib_src(P, (int64_t)start + lane) : Src{0u, 0.f};
}

template <int DEPTH, bool PLAIN>   // contribution rows in flight per trip; PLAIN: no optimizer state rows (SGD; the caller decides the regulariser through opt)
__device__ __forceinline__ void segment_update(const ApplyParams& P, const OptParams& opt, int32_t step, uint32_t dest, int len,
                                               const Src mine, int lane, int nchunks, float& lp_acc) {
    float* wrow = P.table + (int64_t)dest * P.ld;
    float* s0row = (!PLAIN && P.state0) ? P.state0 + (int64_t)dest * P.ld : nullptr;
    float* s1row = (!PLAIN && P.state1) ? P.state1 + (int64_t)dest * P.ld : nullptr;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    // state_lag: the catch-up brought w of this row to step - 1 and left m, v (and the tag) where the row was last written —
    // their decay over the missed steps is two multiplications per element and step, redone here instead of a write + read
    // of both state rows (emg_deferred_catchup)
    int lagn = 0;
    if constexpr (!PLAIN) {
        if (P.state_lag) {
            const int32_t from = P.tag[dest];
            lagn = from > 0 && from < step - 1 ? step - 1 - from : 0;
        }
    }
    for (int c0 = 0; c0 < nchunks; c0 += 128) {
        const int ca = c0 + lane, cb = c0 + 64 + lane;
        const bool oa = ca < nchunks, ob = cb < nchunks;
        float4 accA = zero, accB = zero, wA = zero, wB = zero;
        // the table row AND its optimizer state (momentum / Adagrad accumulator / Adam m, v) are fetched with the first
        // contribution rows, as 16-byte loads: the update after the sum then waits for nothing (the state used to be read
        // element by element after the sum — a dependent round trip per segment and 4x the load instructions; C1 / C2 / C5
        // with Adam: DESIGN.md 4.1)
        float4 m0A = zero, m0B = zero, m1A = zero, m1B = zero;
        if (oa) wA = *reinterpret_cast<const float4*>(wrow + 4 * ca);
        if (ob) wB = *reinterpret_cast<const float4*>(wrow + 4 * cb);
        if constexpr (!PLAIN) {
            if (s0row && oa) m0A = *reinterpret_cast<const float4*>(s0row + 4 * ca);
            if (s0row && ob) m0B = *reinterpret_cast<const float4*>(s0row + 4 * cb);
            if (s1row && oa) m1A = *reinterpret_cast<const float4*>(s1row + 4 * ca);
            if (s1row && ob) m1B = *reinterpret_cast<const float4*>(s1row + 4 * cb);
        }
        for (int u = 0; u < len; u += DEPTH) {
            float4 va[DEPTH], vb[DEPTH];
            float cf[DEPTH];
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) {
                va[j] = zero; vb[j] = zero; cf[j] = 0.f;
                if (u + j < len) {
                    const uint32_t row = (uint32_t)__builtin_amdgcn_readlane((int)mine.row, u + j);
                    cf[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.coef), u + j));
                    const float* rj = P.contrib + (int64_t)row * P.ldc;
                    if (oa) va[j] = *reinterpret_cast<const float4*>(rj + 4 * ca);
                    if (ob) vb[j] = *reinterpret_cast<const float4*>(rj + 4 * cb);
                }
            }
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) {   // added in contribution order
                if (u + j < len) {
                    if (oa) add_scaled(accA, va[j], cf[j]);
                    if (ob) add_scaled(accB, vb[j], cf[j]);
                }
            }
        }
        auto finish = [&](int c, float4 wv, const float4& g, float4 m0, float4 m1) {
            const int64_t off = 4 * (int64_t)c;
            float w[4] = {wv.x, wv.y, wv.z, wv.w};
            float gg[4] = {g.x, g.y, g.z, g.w};
            float a0[4] = {m0.x, m0.y, m0.z, m0.w}, a1[4] = {m1.x, m1.y, m1.z, m1.w};
            if constexpr (!PLAIN) {
                for (int i = 0; i < lagn; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) adam_decay_elem(opt, a0[j], a1[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lp_fold(opt, w[j], gg[j], lp_acc);
                opt_update_elem(opt, w[j], gg[j], &a0[j], &a1[j]);
            }
            *reinterpret_cast<float4*>(wrow + off) = make_float4(w[0], w[1], w[2], w[3]);
            if constexpr (!PLAIN) {
                if (s0row) *reinterpret_cast<float4*>(s0row + off) = make_float4(a0[0], a0[1], a0[2], a0[3]);
                if (s1row) *reinterpret_cast<float4*>(s1row + off) = make_float4(a1[0], a1[1], a1[2], a1[3]);
            }
        };
        if (oa) finish(ca, wA, accA, m0A, m1A);
        if (ob) finish(cb, wB, accB, m0B, m1B);
    }
    if (P.tag && lane == 0) P.tag[dest] = step;
}

// Rows of 17..32 sixteen-byte chunks — k = 100 of the real-valued models, the reference's default width — would run
// segment_update with 25 of the wave's 64 lanes and one segment's chain of dependent trips per wave.  Here each HALF of the
// wave takes its own item (a segment has at most kDeferSegment = 32 contributions: its sources fit the half's lanes, and
// travel inside the half by ds_bpermute instead of v_readlane): two chains per wave, 50 of 64 lanes at k = 100.  The same
// additions in the same order — the same bits.  dest / len / on are per lane (the same within a half).
template <int DEPTH, bool PLAIN>
__device__ __forceinline__ void segment_update_half(const ApplyParams& P, const OptParams& opt, int32_t step, uint32_t dest, int len, bool on,
                                                    const Src mine, int lane, int nchunks, float& lp_acc) {
    const int l = lane & 31, hb = lane & 32;
    const bool oc = on && l < nchunks;
    const int64_t base = on ? (int64_t)dest * P.ld : 0;
    float* wrow = P.table + base;
    float* s0row = (!PLAIN && P.state0) ? P.state0 + base : nullptr;
    float* s1row = (!PLAIN && P.state1) ? P.state1 + base : nullptr;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    int lagn = 0;   // (state_lag: see segment_update)
    if constexpr (!PLAIN) {
        if (P.state_lag && on) {
            const int32_t from = P.tag[dest];
            lagn = from > 0 && from < step - 1 ? step - 1 - from : 0;
        }
    }
    float4 acc = zero, wv = zero, m0 = zero, m1 = zero;
    if (oc) wv = *reinterpret_cast<const float4*>(wrow + 4 * l);
    if constexpr (!PLAIN) {
        if (s0row && oc) m0 = *reinterpret_cast<const float4*>(s0row + 4 * l);
        if (s1row && oc) m1 = *reinterpret_cast<const float4*>(s1row + 4 * l);
    }
    const int lenm = on ? len : 0;
    const int maxlen = max(__builtin_amdgcn_readlane(lenm, 0), __builtin_amdgcn_readlane(lenm, 32));
    for (int u = 0; u < maxlen; u += DEPTH) {
        float4 v[DEPTH];
        float cf[DEPTH];
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            v[j] = zero;
            const int from_lane = hb + min(u + j, 31);
            const uint32_t row = (uint32_t)__shfl((int)mine.row, from_lane, 64);
            cf[j] = __shfl(mine.coef, from_lane, 64);
            if (oc && u + j < len) v[j] = *reinterpret_cast<const float4*>(P.contrib + (int64_t)row * P.ldc + 4 * l);
        }
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {   // added in contribution order
            if (oc && u + j < len) add_scaled(acc, v[j], cf[j]);
        }
    }
    if (oc) {
        float w[4] = {wv.x, wv.y, wv.z, wv.w};
        float gg[4] = {acc.x, acc.y, acc.z, acc.w};
        float a0[4] = {m0.x, m0.y, m0.z, m0.w}, a1[4] = {m1.x, m1.y, m1.z, m1.w};
        if constexpr (!PLAIN) {
            for (int i = 0; i < lagn; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) adam_decay_elem(opt, a0[j], a1[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lp_fold(opt, w[j], gg[j], lp_acc);
            opt_update_elem(opt, w[j], gg[j], &a0[j], &a1[j]);
        }
        *reinterpret_cast<float4*>(wrow + 4 * l) = make_float4(w[0], w[1], w[2], w[3]);
        if constexpr (!PLAIN) {
            if (s0row) *reinterpret_cast<float4*>(s0row + 4 * l) = make_float4(a0[0], a0[1], a0[2], a0[3]);
            if (s1row) *reinterpret_cast<float4*>(s1row + 4 * l) = make_float4(a1[0], a1[1], a1[2], a1[3]);
        }
    }
    if (P.tag && on && l == 0) P.tag[dest] = step;
}

// a segment of more than kDeferSegment rows in a workspace WITHOUT partial rows (sized by emg_apply_workspace_bytes):
// one wave sums all of it left to right (slow for hub rows; the documented small-workspace behaviour)
__device__ __forceinline__ void long_segment_serial(const ApplyParams& P, const OptParams& opt, int32_t step, const LongTask tk,
                                                    int lane, int nchunks, float& lp_acc) {
    const int64_t t = tk.head, end = t + tk.len;
    const uint32_t key = P.keys[t];
    if ((int64_t)key >= P.n_rows) return;
    float* wrow = P.table + (int64_t)key * P.ld;
    float* s0row = P.state0 ? P.state0 + (int64_t)key * P.ld : nullptr;
    float* s1row = P.state1 ? P.state1 + (int64_t)key * P.ld : nullptr;
    for (int c0 = 0; c0 < nchunks; c0 += 64) {   // (uniform loop: block_sources needs every lane)
        const int c = c0 + lane;
        const bool ok = c < nchunks;
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        for (int64_t u0 = t; u0 < end; u0 += kLongSegment) {
            const int64_t u1 = min(u0 + kLongSegment, end);
            const Src mine = block_sources(P, u0, u1, lane);
            sum_block<4, 8>(P, u0, u1, ok ? c : 0, mine.row, mine.coef, g, true);
        }
        if (ok) {
            float w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                w[j] = wrow[4 * c + j];
                lp_fold(opt, w[j], g[j], lp_acc);
                opt_update_elem(opt, w[j], g[j], s0row ? s0row + 4 * c + j : nullptr, s1row ? s1row + 4 * c + j : nullptr);
                wrow[4 * c + j] = w[j];
            }
        }
    }
    if (P.tag && lane == 0) P.tag[key] = step;
}

#ifdef EMG_TRACE   // timing aid (tools/trace_waves.py): wall-clock stamps (10 ns) of every wave of the last launch
__device__ unsigned long long emg_trace_buf[4 * 65536];
#define EMG_STAMP(slot) do { if (lane == 0 && gw < 65536) emg_trace_buf[4 * gw + (slot)] = wall_clock64(); } while (0)
#else
#define EMG_STAMP(slot) do { } while (0)
#endif

// heavy / relief: the last `heavy` waves of the grid take `relief` items fewer each (they carry the other table's long
// segments, see apply_segments_kernel); the others share what that leaves
// FIX: the optimizer known at compile time for the stateful instantiations (0: run-time switch; EMG_OPT_ADAM / EMG_OPT_ADAGRAD: that
// rule, no regulariser).  The run-time form walks opt_update_elem's four branches and lp_fold's test for EVERY element — scalar
// compares and branches that a wave issues in line with its vector work: per-wave stamps put an item of C1 / C2 at 4 - 5 us where
// its round trips are 0.3 us each (TCP->TCC latency counters), i.e. at the SIMD's instruction issue.
constexpr int kFixSgdLp2 = 100;   // FIX: plain SGD with the LP regulariser at p = 2 folded in (C3 + LP: the reference's default regulariser)
template <bool PLAIN, bool HALF, int FIX = 0>
__device__ __forceinline__ void apply_segments_table(const ApplyParams& P, float* __restrict__ partial, int64_t ldp, int64_t gw,
                                                     int64_t nw, int lane, int64_t heavy = 0, int64_t relief = 0) {
    OptParams opt = P.opt;
    if constexpr (PLAIN) { opt.opt = EMG_OPT_SGD; opt.lp_lambda = 0.f; }   // (known at compile time: the update folds to w - lr g)
    if constexpr (FIX == kFixSgdLp2) { opt.opt = EMG_OPT_SGD; opt.lp_p = 2; }   // (lambda stays a run-time value: the tables' may differ)
    else if constexpr (FIX != 0) { opt.opt = FIX; opt.lp_lambda = 0.f; }
    // plain SGD has no optimizer state, with or without the regulariser: the segment forms then carry no state rows, no Adam lag and
    // no dense pass (the registers of eight float4 state values per lane are what kept this form at 102 VGPRs beside PLAIN's 78)
    constexpr bool NS = PLAIN || FIX == kFixSgdLp2;
    int32_t step = P.step;
    if (P.ctl) {   // the step's number and learning rates from the device record (a captured graph cannot bake them)
        const float* h = P.which ? P.ctl->hyper_rel : P.ctl->hyper_ent;
        opt.lr = h[0]; opt.lr_t = h[5];
        step = P.ctl->step;
    }
    const int nchunks = P.k_int / 4;
    const uint32_t n_multi = P.counters[GC_MULTI];
    const uint32_t n_single = P.skip_single ? 0u : P.counters[GC_SINGLE];
    const uint32_t n_tasks = min(P.counters[GC_TASKS], P.task_cap);
    float lp_acc = 0.f;
    if (n_tasks && partial) {
        const unsigned per = waves_per_task(P, nchunks);
        for (int64_t i2 = gw; i2 < (int64_t)n_tasks * per; i2 += nw)
            long_task_wave<4, 8, 4, NS>(P, opt, step, partial, ldp, P.tasks[i2 / per], (unsigned)(i2 % per), per, lane, lp_acc);
    } else if (n_tasks) {
        for (int64_t i = gw; i < (int64_t)n_tasks; i += nw) {
            const LongTask tk = P.tasks[i];
            if (tk.block == 0u && tk.len != 0u) long_segment_serial(P, opt, step, tk, lane, nchunks, lp_acc);
        }
    }
    const int64_t total = (int64_t)n_multi + n_single;
    int64_t share = (total + nw - 1) / nw, i0, i1;
    if (heavy > 0 && relief > 0) {
        relief = min(relief, share);
        share = (total + heavy * relief + nw - 1) / nw;
        const int64_t light = nw - heavy;
        if (gw < light) { i0 = gw * share; i1 = i0 + share; }
        else { i0 = light * share + (gw - light) * (share - relief); i1 = i0 + (share - relief); }
        i0 = min(i0, total); i1 = min(i1, total);
    } else {
        i0 = gw * share; i1 = min(total, i0 + share);
    }
    static_assert(kDeferSegment <= 32, "segment_update_half keeps a segment's sources in 32 lanes");
    constexpr bool halves = HALF;   // (its own instantiation: both forms in one kernel cost the wide rows a wave per SIMD)
    for (int64_t base = i0; base < i1; base += 64) {
        const int64_t it = base + lane;
        Seg sg{0u, 0u, 0u};
        if (it < i1) {
            if (it < (int64_t)n_multi) sg = P.multi[it];
            else { const uint32_t at = P.single[it - n_multi]; sg = Seg{at, 1u, P.keys[at]}; }
        }
        const int cnt = (int)min((int64_t)64, i1 - base);
        if constexpr (halves) {   // two items at a time, one per half of the wave (segment_update_half)
            const int l = lane & 31, hi = lane >> 5;
            auto item = [&](int k, uint32_t& len, uint32_t& dest, bool& on) -> Src {
                const int kk = k + hi;
                const int from_lane = min(kk, 63);
                len = (uint32_t)__shfl((int)sg.len, from_lane, 64);
                dest = (uint32_t)__shfl((int)sg.dest, from_lane, 64);
                const uint32_t start = (uint32_t)__shfl((int)sg.start, from_lane, 64);
                on = kk < cnt && (int64_t)dest < P.n_rows;   // (defensive: never write outside the table)
                return (on && (uint32_t)l < len) ? contrib_src(P, (int64_t)start + l) : Src{0u, 0.f};
            };
            uint32_t len_n, dest_n;
            bool on_n;
            Src nxt = item(0, len_n, dest_n, on_n);
            for (int k = 0; k < cnt; k += 2) {
                const uint32_t len = len_n, dest = dest_n;
                const bool on = on_n;
                const Src mine = nxt;
                if (k + 2 < cnt) nxt = item(k + 2, len_n, dest_n, on_n);
                segment_update_half<EMG_SEG_DEPTH, NS>(P, opt, step, dest, (int)len, on, mine, lane, nchunks, lp_acc);
            }
        } else {
        Src nxt = segment_sources(P, (uint32_t)__builtin_amdgcn_readlane((int)sg.start, 0), (uint32_t)__builtin_amdgcn_readlane((int)sg.len, 0), lane);
        for (int k = 0; k < cnt; ++k) {
            const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)sg.len, k);
            const uint32_t dest = (uint32_t)__builtin_amdgcn_readlane((int)sg.dest, k);
            const Src mine = nxt;
            if (k + 1 < cnt)
                nxt = segment_sources(P, (uint32_t)__builtin_amdgcn_readlane((int)sg.start, k + 1),
                                      (uint32_t)__builtin_amdgcn_readlane((int)sg.len, k + 1), lane);
            if ((int64_t)dest >= P.n_rows) continue;   // defensive: never write outside the table
            segment_update<EMG_SEG_DEPTH, NS>(P, opt, step, dest, (int)len, mine, lane, nchunks, lp_acc);
        }
        }
    }
    if (P.which == 0) EMG_STAMP(3);   // (trace builds: the entity table's items done, its untouched rows next)
    if constexpr (!NS) if (P.dense_here) {   // the rows nothing touched: 64 rows' counts in one load, then a wave per untouched row
        ApplyParams Q = P;
        Q.opt = opt;
        const bool vec = untouched_vec(P);
        const int64_t per = (P.n_rows + nw - 1) / nw;
        const int64_t r0 = gw * per, r1 = min(P.n_rows, r0 + per);
        for (int64_t base = r0; base < r1; base += 64) {
            const int64_t r = base + lane;
            const bool idle = r < r1 && P.off[r + 1] == P.off[r];
            unsigned long long todo = __ballot(idle);
            if (vec && nchunks <= 64 && P.state0 && P.state1) {
                // FOUR rows in flight (one row at a time is a chain of load -> update -> store round trips: 5 untouched rows per wave
                // of C1's launch took as long as its 8 destinations).  Past the list's end: the last row again, its stores skipped.
                while (todo) {
                    int64_t row[4];
                    bool live[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        live[q] = todo != 0ull;
                        const int k = live[q] ? __ffsll((long long)todo) - 1 : 0;
                        row[q] = live[q] ? base + k : (q ? row[q - 1] : base);
                        todo &= todo - 1ull;   // (0 stays 0)
                    }
                    const int c = min(lane, nchunks - 1);
                    float4 wv[4], av[4], bv[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        wv[q] = *reinterpret_cast<const float4*>(P.table + row[q] * P.ld + 4 * c);
                        av[q] = *reinterpret_cast<const float4*>(P.state0 + row[q] * P.ld + 4 * c);
                        bv[q] = *reinterpret_cast<const float4*>(P.state1 + row[q] * P.ld + 4 * c);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float ww[4] = {wv[q].x, wv[q].y, wv[q].z, wv[q].w}, aa[4] = {av[q].x, av[q].y, av[q].z, av[q].w},
                              bb[4] = {bv[q].x, bv[q].y, bv[q].z, bv[q].w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float g = 0.f;
                            opt_update_elem(opt, ww[j], g, &aa[j], &bb[j]);
                        }
                        if (live[q] && lane < nchunks) {
                            *reinterpret_cast<float4*>(P.table + row[q] * P.ld + 4 * c) = make_float4(ww[0], ww[1], ww[2], ww[3]);
                            *reinterpret_cast<float4*>(P.state0 + row[q] * P.ld + 4 * c) = make_float4(aa[0], aa[1], aa[2], aa[3]);
                            *reinterpret_cast<float4*>(P.state1 + row[q] * P.ld + 4 * c) = make_float4(bb[0], bb[1], bb[2], bb[3]);
                        }
                    }
                }
                continue;
            }
            while (todo) {
                const int k = __ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                untouched_row_update(Q, base + k, lane, vec, lp_acc);
            }
        }
    }
    // (every wave of the persistent grid ends here at the same moment: one double atomic per WAVE to one address was 54 us of
    //  C3 + LP's 124 us apply — same-address device-scope atomics retire one per ~10 ns; block_add_double: one per workgroup)
    if (opt.lp_lambda != 0.f) block_add_double(P.lp_accum, lp_acc);
}

struct SegmentsLaunch { ApplyParams P[2]; float* partial[2]; int64_t ldp[2]; int32_t n_tables; int32_t relief; };


#ifndef EMG_SEG_MINWAVES
#define EMG_SEG_MINWAVES 1   // A/B aid: waves per SIMD the stateful instantiations are compiled for (a register cap)
#endif
template <bool PLAIN, bool RIDE, bool HALF = false, int FIX = 0>   // HALF: rows of 17..32 chunks, two items per wave (segment_update_half)
__global__ __launch_bounds__(256, ((PLAIN || FIX == kFixSgdLp2) ? 1 : EMG_SEG_MINWAVES)) void apply_segments_kernel(const SegmentsLaunch K, const Riders riders) {
    // RIDE: the first workgroups of the launch do preparation stages of the next batches (emg_group_kernels.hpp)
    unsigned bx = blockIdx.x, nbx = gridDim.x;
    if constexpr (RIDE) {
        if (run_riders(riders, &bx)) return;
        nbx -= riders.total;
    }
    const int lane = threadIdx.x & 63;
    const int64_t gw = ((int64_t)bx * blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)nbx * blockDim.x) >> 6;
    EMG_STAMP(0);
    // (a loop with a run-time index, not two inlined copies: one set of live registers.  The second table's few items go
    // to the other end of the grid, where waves have less of the first table's work)
    // Two tables: the second one's items (relations: few rows, many contributions each) go one per wave to the far end of
    // the grid.  A segment costs its dependent trips of EMG_SEG_DEPTH rows, so those waves take that many entity items fewer
    // (per-wave stamps, C3: 4600 waves ended at 52-57 us, the 540 that also carried a 16-row relation segment at 71).
    int64_t heavy = 0, relief = 0;
    if (K.n_tables == 2 && K.relief) {
        const ApplyParams& R = K.P[1];
        const int64_t n_multi = R.counters[GC_MULTI], n_single = R.skip_single ? 0 : R.counters[GC_SINGLE];
        const int64_t rows = (int64_t)R.counters[GC_VALID] - (int64_t)R.counters[GC_SINGLE];
        const int64_t items = n_multi + n_single;
        if (items > 0 && items <= nw) {
            const int64_t trips = n_multi ? (rows / n_multi + EMG_SEG_DEPTH - 1) / EMG_SEG_DEPTH : 1;   // of an average multi-row segment
            heavy = items; relief = trips;
        }
    }
    apply_segments_table<PLAIN, HALF, FIX>(K.P[0], K.partial[0], K.ldp[0], gw, nw, lane, heavy, relief);
    EMG_STAMP(1);
    if (K.n_tables == 2) {
        apply_segments_table<PLAIN, HALF, FIX>(K.P[1], K.partial[1], K.ldp[1], nw - 1 - gw, nw, lane);
        EMG_STAMP(2);
    }
}

#ifdef EMG_TRACE
extern "C" int emg_trace_read(unsigned long long* host, int64_t n_words) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(emg_trace_buf), (size_t)n_words * 8) == hipSuccess ? 0 : -1;
}
extern "C" int emg_trace_clear(void) {
    static unsigned long long zeros[4 * 65536];
    return hipMemcpyToSymbol(HIP_SYMBOL(emg_trace_buf), zeros, sizeof(zeros)) == hipSuccess ? 0 : -1;
}
#endif

}  // namespace emg

using namespace emg;

// launch geometry of one table's apply, decided once so that two tables can share their launches
struct ApplyLaunch {
    bool any = false, vec = false, skinny = false, dense = false;
    bool segs = false;           // descriptor-driven kernel (counting grouping, 16-byte rows of more than 16 chunks)
    unsigned grid = 0, nb = 0;   // window-kernel workgroups (segs: persistent workgroups); task-kernel workgroups (0: no task list)
    float* partial = nullptr; int64_t ldp = 0;
};

static bool plain_sgd(const ApplyParams& P) { return P.opt.opt == EMG_OPT_SGD && P.opt.lp_lambda == 0.f; }

static bool segments_path_enabled() {
    static const bool on = [] { const char* e = getenv("EMG_APPLY"); return !(e && strcmp(e, "window") == 0); }();   // A/B aid
    return on;
}

static int apply_setup(const emg_apply_args* a, ApplyParams& P, ApplyLaunch& A) {
    const int opt = a->opt;
    const int32_t k_int = a->k_int;
    const int64_t n_rows = a->n_rows, ld = a->ld, ldc = a->ldc, n_contrib = a->n_contrib;
    const float* hyper = a->hyper;
    EMG_REQUIRE(opt >= EMG_OPT_SGD && opt <= EMG_OPT_ADAM_LAZY, "emg_apply_grouped: unknown optimizer %d", opt);
    EMG_REQUIRE(a->table && n_rows > 0 && ld >= k_int && k_int > 0, "emg_apply_grouped: bad table arguments");
    EMG_REQUIRE(n_rows < ((int64_t)1 << 31), "emg_apply_grouped: too many rows");
    EMG_REQUIRE(n_contrib == 0 || (a->contrib && a->workspace && ldc >= k_int), "emg_apply_grouped: bad contribution arguments");
    EMG_REQUIRE(!(opt == EMG_OPT_MOMENTUM || opt == EMG_OPT_ADAGRAD) || a->state0, "emg_apply_grouped: optimizer needs state0");
    EMG_REQUIRE(!(opt == EMG_OPT_ADAM || opt == EMG_OPT_ADAM_LAZY) || (a->state0 && a->state1),
                "emg_apply_grouped: adam needs state0 and state1");
    EMG_REQUIRE(opt != EMG_OPT_ADAM || a->tag, "emg_apply_grouped: dense-equivalent adam needs the tag array");
    EMG_REQUIRE(hyper[6] == 0.f || (a->tag && hyper[7] >= 1.f), "emg_apply_grouped: a folded LP regulariser needs the tag array and p >= 1");
    EMG_REQUIRE(a->layout_n == 0 || a->layout_n >= n_contrib, "emg_apply_grouped: layout_n < n_contrib");
    P = ApplyParams{};
    A = ApplyLaunch{};
    P.table = a->table; P.n_rows = n_rows; P.ld = ld; P.k_int = k_int;
    P.state0 = a->state0; P.state1 = a->state1; P.tag = a->tag; P.step = a->step;
    P.contrib = a->contrib; P.ldc = ldc; P.n = n_contrib; P.skip_single = a->skip_single ? 1 : 0;
    P.opt = make_opt_params(opt, hyper);
    P.lp_accum = a->lp_accum;
    P.ctl = (const StepCtl*)a->ctl; P.which = a->table_index;
    A.dense = (opt == EMG_OPT_ADAM || P.opt.lp_lambda != 0.f) && !a->deferred_dense;
    // (skip_single = 1 with deferred_dense = 2: the scoring kernel replayed and updated every singleton itself — emg_backward_args.lr_hist)
    EMG_REQUIRE(a->deferred_dense != 2 || (opt == EMG_OPT_ADAM && P.opt.lp_lambda == 0.f && a->tag),
                "emg_apply_grouped: deferred_dense = 2 (m, v lag behind w) is Adam's, without a regulariser");
    EMG_REQUIRE(a->deferred_dense != 2 || segments_path_enabled(), "emg_apply_grouped: deferred_dense = 2 needs the descriptor-driven apply");
    P.state_lag = a->deferred_dense == 2 ? 1 : 0;
    { const char* e = getenv("EMG_APPLY_HALF"); P.half_rows = (e && e[0] == '0') ? 0 : 1; }   // A/B aid (read per call: tests flip it)
    if (n_contrib <= 0) return EMG_OK;
    A.any = true;
    A.vec = (k_int % 4 == 0) && (ld % 4 == 0) && (ldc % 4 == 0) && aligned16(a->table) && aligned16(a->contrib) &&
            (!a->state0 || aligned16(a->state0)) && (!a->state1 || aligned16(a->state1));
    A.ldp = (k_int + 3) / 4 * 4;
    const int64_t N = a->layout_n > 0 ? a->layout_n : n_contrib;   // what the workspace was laid out for
    GroupWs w;
    int rc = group_ws_layout(a->workspace, a->workspace_bytes, N, n_rows, A.ldp, &w);
    if (rc != EMG_OK) return rc;
    P.keys = w.keys;
    P.vals = w.vals;
    P.srcrow = a->factored ? w.srcrow : w.vals;
    P.coef = a->factored ? w.coef : nullptr;
    P.arrive = w.arrive;
    const int nch = A.vec ? k_int / 4 : k_int;
    A.skinny = nch <= 16;
    A.segs = w.counting && A.vec && !A.skinny && segments_path_enabled();
    EMG_REQUIRE(!P.state_lag || A.segs, "emg_apply_grouped: deferred_dense = 2 needs the descriptor-driven apply (counting grouping, "
                                        "16-byte aligned rows of more than 16 chunks)");
    EMG_REQUIRE(!P.ctl || A.segs, "emg_apply_grouped: a device-side step record needs the descriptor-driven apply (counting "
                                  "grouping, 16-byte aligned rows of more than 16 chunks)");
    if (A.segs) {
        P.multi = w.multi; P.single = w.single; P.tasks = w.tasks; P.counters = w.counters; P.task_cap = w.task_cap;
        P.off = w.off;
        A.partial = w.partial;
        // persistent grid: enough waves to fill the chip at 8 per SIMD, fewer for small batches (every wave of the launch
        // reads the list counters and its descriptors before it has anything to do)
        static const int env_blocks = getenv("EMG_SEG_BLOCKS") ? atoi(getenv("EMG_SEG_BLOCKS")) : 0;   // A/B aid
        int64_t waves = N / 2;
        waves = waves < 256 ? 256 : (waves > 8192 ? 8192 : waves);
        A.grid = env_blocks > 0 ? (unsigned)env_blocks : (unsigned)cdiv(waves, 4);
        return EMG_OK;
    }
    // window per wave: large enough to amortise wave launches, small enough for >= ~16k waves in flight
    int win = 64;
    while (win > 1 && n_contrib / win < 16384) win >>= 1;
    if (const char* e = getenv("EMG_APPLY_WIN")) { const int v = atoi(e); if (v >= 1 && v <= 64 && (v & (v - 1)) == 0) win = v; }  // A/B aid
    P.win = win;
    A.grid = (unsigned)cdiv(cdiv(n_contrib, win) * 64, 256);
    static const bool no_long = getenv("EMG_NO_LONG") != nullptr;     // A/B aids
    static const int defer_env = getenv("EMG_DEFER") ? atoi(getenv("EMG_DEFER")) : 0;
    P.defer = defer_env >= 8 ? defer_env : kDeferSegment;   // (>= 8: the task list has room for n / 8 tasks)
    if (w.partial && !A.skinny && !no_long) {  // long segments go to apply_long_kernel (count zeroed by the grouping)
        P.long_list = w.tasks; P.long_count = w.counters + GC_LONG_COUNT;
        P.long_cap = w.task_cap;
        A.partial = w.partial;
        // one wave per block task, 16 waves per workgroup.  Tables with few rows (relations) defer most of their
        // segments: a workgroup per CU; tables with many rows (entities) defer hub rows only: a smaller grid,
        // whose cost when the list is empty is a few microseconds
        const int64_t possible = n_contrib / (kDeferSegment + 1) < n_rows ? n_contrib / (kDeferSegment + 1) : n_rows;
        const int64_t most = n_rows <= 4096 ? 256 : 64;
        A.nb = (unsigned)(possible < 1 ? 1 : (possible < most ? possible : most));
    }
    return EMG_OK;
}

// Workgroups of apply_segments_kernel<PLAIN, RIDE> the device holds at once.  The grid is persistent with STATIC shares: a
// launch of more waves than are resident runs as a full round followed by a partly empty one (per-wave stamps,
// tools/trace_waves.py, C3: 5120 of 8192 waves start at 0 and live 38 us, the other 3072 start at 34-41 us: 78 us for
// 1.6 rounds of work) — so never launch more than fit.
typedef void (*SegmentsKernel)(const SegmentsLaunch, const Riders);
// fix: 0 run-time optimizer switch, 1 Adam without regulariser, 2 Adagrad without regulariser, 3 plain SGD + LP at p = 2 (compile-time forms)
static SegmentsKernel segments_kernel(bool plain, bool ride, bool half, int fix = 0) {
    static const SegmentsKernel fns[8] = {
        apply_segments_kernel<false, false, false>, apply_segments_kernel<false, false, true>,
        apply_segments_kernel<false, true, false>,  apply_segments_kernel<false, true, true>,
        apply_segments_kernel<true, false, false>,  apply_segments_kernel<true, false, true>,
        apply_segments_kernel<true, true, false>,   apply_segments_kernel<true, true, true>};
    static const SegmentsKernel adam[4] = {
        apply_segments_kernel<false, false, false, EMG_OPT_ADAM>, apply_segments_kernel<false, false, true, EMG_OPT_ADAM>,
        apply_segments_kernel<false, true, false, EMG_OPT_ADAM>,  apply_segments_kernel<false, true, true, EMG_OPT_ADAM>};
    static const SegmentsKernel adagrad[4] = {
        apply_segments_kernel<false, false, false, EMG_OPT_ADAGRAD>, apply_segments_kernel<false, false, true, EMG_OPT_ADAGRAD>,
        apply_segments_kernel<false, true, false, EMG_OPT_ADAGRAD>,  apply_segments_kernel<false, true, true, EMG_OPT_ADAGRAD>};
    static const SegmentsKernel sgdlp2[4] = {
        apply_segments_kernel<false, false, false, kFixSgdLp2>, apply_segments_kernel<false, false, true, kFixSgdLp2>,
        apply_segments_kernel<false, true, false, kFixSgdLp2>,  apply_segments_kernel<false, true, true, kFixSgdLp2>};
    if (!plain && fix == 3) return sgdlp2[(ride ? 2 : 0) + (half ? 1 : 0)];
    if (!plain && fix == 1) return adam[(ride ? 2 : 0) + (half ? 1 : 0)];
    if (!plain && fix == 2) return adagrad[(ride ? 2 : 0) + (half ? 1 : 0)];
    return fns[(plain ? 4 : 0) + (ride ? 2 : 0) + (half ? 1 : 0)];
}
static bool segments_half(const ApplyParams& P) { return P.half_rows && P.k_int / 4 <= 32; }
// which compile-time optimizer form serves this table (EMG_APPLY_FIX = 0: the run-time switch everywhere — A/B aid)
static int segments_fix(const ApplyParams& P) {
    static const bool off = getenv("EMG_APPLY_FIX") && atoi(getenv("EMG_APPLY_FIX")) == 0;
    if (off) return 0;
    if (P.opt.lp_lambda != 0.f) return (P.opt.opt == EMG_OPT_SGD && P.opt.lp_p == 2) ? 3 : 0;
    return P.opt.opt == EMG_OPT_ADAM ? 1 : (P.opt.opt == EMG_OPT_ADAGRAD ? 2 : 0);
}
static unsigned segments_capacity(bool plain, bool ride, bool half, int fix) {
    static std::atomic<unsigned> cached[4][8][64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1024u;
    std::atomic<unsigned>& c = cached[plain ? 0 : fix][(plain ? 4 : 0) + (ride ? 2 : 0) + (half ? 1 : 0)][dev & 63];
    unsigned v = c.load(std::memory_order_relaxed);
    if (v) return v;
    const void* fn = (const void*)segments_kernel(plain, ride, half, fix);
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) != hipSuccess || per_cu <= 0) per_cu = 4;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    v = (unsigned)per_cu * (unsigned)cus;
    c.store(v, std::memory_order_relaxed);
    return v;
}
static unsigned segments_grid(unsigned wanted, bool plain, bool ride, bool half, int fix) {
    static const bool fixed = getenv("EMG_SEG_BLOCKS") != nullptr;   // A/B aid: the grid as given
    const unsigned cap = segments_capacity(plain, ride, half, fix);
    return fixed || wanted <= cap ? wanted : cap;
}

// the dense pass inside the descriptor-driven launch (ApplyParams.dense_here): small tables, where a launch of its own costs more
// than its rows (the reference's own configurations: 12 of a 73 us step); EMG_DENSE_FUSED = 0 / 1 forces it off / on (A/B aid)
static bool dense_in_segments(const ApplyParams& P, const ApplyLaunch& A) {
    static const int env = getenv("EMG_DENSE_FUSED") ? atoi(getenv("EMG_DENSE_FUSED")) : -1;
    if (!(A.any && A.segs && A.dense) || P.opt.lp_lambda != 0.f) return false;
    return (env >= 0 ? env != 0 : true) && P.n_rows <= kDenseHereMaxRows;   // (never above: the bucket grouping writes the offset array only for tables of up to this size — emg_group_bucket.hip: BucketTable::off)
}

static int apply_launch(const ApplyParams& P0, const ApplyLaunch& A0, hipStream_t st) {
    ApplyParams P = P0;
    ApplyLaunch A = A0;
    if (dense_in_segments(P, A)) { P.dense_here = 1; A.dense = false; }
    if (A.any && A.segs) {
        SegmentsLaunch K{};
        K.P[0] = P; K.partial[0] = A.partial; K.ldp[0] = A.ldp; K.n_tables = 1;
        static const Riders none{};
        const bool half = segments_half(P);
        const int fix = segments_fix(P);
        const unsigned g1 = segments_grid(A.grid, plain_sgd(P), false, half, fix);
        hipLaunchKernelGGL(segments_kernel(plain_sgd(P), false, half, fix), dim3(g1), dim3(256), 0, st, K, none);
        EMG_LAUNCH_CHECK();
    } else if (A.any) {
        const dim3 grid(A.grid), block(256);
        // DEPTH 2 everywhere (measured, C3: relation table 0.121 ms vs 0.148 ms with 16 rows in flight at 2 waves/SIMD,
        // entity table 0.112 vs 0.22; alone on the chip the relation apply takes 0.047 / 0.054 / 0.064 ms at 2 / 8 / 16):
        // segments of up to 64 rows gain more from 7 waves/SIMD than from deeper trips
        if (A.skinny) {  // skinny rows: four segments per wave
            if (A.vec) hipLaunchKernelGGL((apply_rows_sub_kernel<4, 16>), grid, block, 0, st, P);
            else hipLaunchKernelGGL((apply_rows_sub_kernel<1, 16>), grid, block, 0, st, P);
        } else if (A.vec) hipLaunchKernelGGL((apply_rows_kernel<4, 2>), grid, block, 0, st, P);
        else hipLaunchKernelGGL((apply_rows_kernel<1, 2>), grid, block, 0, st, P);
        EMG_LAUNCH_CHECK();
        if (A.nb) {
            if (A.vec) hipLaunchKernelGGL((apply_long_kernel<4>), dim3(A.nb), dim3(1024), 0, st, P, A.partial, A.ldp);
            else hipLaunchKernelGGL((apply_long_kernel<1>), dim3(A.nb), dim3(1024), 0, st, P, A.partial, A.ldp);
            EMG_LAUNCH_CHECK();
        }
    }
    if (A.dense) {
        hipLaunchKernelGGL(untouched_rows_kernel, dim3(untouched_blocks(P.n_rows)), dim3(256), 0, st, P);
        EMG_LAUNCH_CHECK();
    }
    return EMG_OK;
}

extern "C" int emg_apply_grouped_ex(const emg_apply_args* a, void* stream) {
    EMG_REQUIRE(a, "emg_apply_grouped_ex: null args");
    ApplyParams P;
    ApplyLaunch A;
    int rc = apply_setup(a, P, A);
    if (rc != EMG_OK) return rc;
    return apply_launch(P, A, (hipStream_t)stream);
}

// Two tables (the entity and the relation table of a training step) through SHARED launches: one apply kernel (or one
// window kernel + one task kernel), one dense pass.  Same results as two emg_apply_grouped_ex calls; falls back to exactly
// those where the shapes differ.
namespace emg {
int apply_pair_impl(const emg_apply_args* a, const emg_apply_args* b, const Riders* riders, void* stream);
}
extern "C" int emg_apply_grouped_pair(const emg_apply_args* a, const emg_apply_args* b, void* stream) {
    return emg::apply_pair_impl(a, b, nullptr, stream);
}
// riders (optional): preparation stages of later batches; carried by the shared descriptor-driven launch, launched on
// their own first where the two tables do not share one
int emg::apply_pair_impl(const emg_apply_args* a, const emg_apply_args* b, const Riders* riders, void* stream) {
    EMG_REQUIRE(a && b, "emg_apply_grouped_pair: null args");
    ApplyParams P0, P1;
    ApplyLaunch A0, A1;
    int rc = apply_setup(a, P0, A0);
    if (rc == EMG_OK) rc = apply_setup(b, P1, A1);
    if (rc != EMG_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const bool share_segs = A0.any && A1.any && A0.segs && A1.segs;
    const bool share = A0.any && A1.any && !A0.segs && !A1.segs && A0.vec && A1.vec && !A0.skinny && !A1.skinny && (A0.nb != 0) == (A1.nb != 0);
    if (riders && riders->total && !share_segs) {
        rc = launch_riders_alone(*riders, st);
        if (rc != EMG_OK) return rc;
        riders = nullptr;
    }
    if (!share && !share_segs) {
        rc = apply_launch(P0, A0, st);
        return rc != EMG_OK ? rc : apply_launch(P1, A1, st);
    }
    if (share_segs) {
        if (dense_in_segments(P0, A0) && dense_in_segments(P1, A1)) { P0.dense_here = P1.dense_here = 1; A0.dense = A1.dense = false; }
        SegmentsLaunch K{};
        K.P[0] = P0; K.partial[0] = A0.partial; K.ldp[0] = A0.ldp;
        K.P[1] = P1; K.partial[1] = A1.partial; K.ldp[1] = A1.ldp; K.n_tables = 2;
        static const bool no_relief = getenv("EMG_APPLY_RELIEF") && atoi(getenv("EMG_APPLY_RELIEF")) == 0;   // A/B aid
        K.relief = no_relief ? 0 : 1;
        const bool plain = plain_sgd(P0) && plain_sgd(P1);
        const bool half = segments_half(P0) && segments_half(P1);   // (one width for both tables)
        const int fix = segments_fix(P0) == segments_fix(P1) ? segments_fix(P0) : 0;   // (one optimizer form for both tables)
        if (riders && riders->total) {
            const dim3 grid(segments_grid(A0.grid > A1.grid ? A0.grid : A1.grid, plain, true, half, fix) + riders->total);
            hipLaunchKernelGGL(segments_kernel(plain, true, half, fix), grid, dim3(256), 0, st, K, *riders);
        } else {
            static const Riders none{};
            const dim3 grid(segments_grid(A0.grid > A1.grid ? A0.grid : A1.grid, plain, false, half, fix));
            hipLaunchKernelGGL(segments_kernel(plain, false, half, fix), grid, dim3(256), 0, st, K, none);
        }
        EMG_LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL((apply_rows_pair_kernel<4, 2>), dim3(A0.grid + A1.grid), dim3(256), 0, st, P0, P1, A0.grid);
        EMG_LAUNCH_CHECK();
        if (A0.nb) {
            hipLaunchKernelGGL((apply_long_pair_kernel<4>), dim3(A0.nb + A1.nb), dim3(1024), 0, st, P0, A0.partial, A0.ldp, A0.nb, P1,
                               A1.partial, A1.ldp);
            EMG_LAUNCH_CHECK();
        }
    }
    if (A0.dense && A1.dense) {   // the dense passes (Keras Adam, folded LP) of both tables: one launch too
        const unsigned b0 = untouched_blocks(P0.n_rows), b1 = untouched_blocks(P1.n_rows);
        hipLaunchKernelGGL(untouched_rows_pair_kernel, dim3(b0 + b1), dim3(256), 0, st, P0, P1, b0);
        EMG_LAUNCH_CHECK();
        return EMG_OK;
    }
    ApplyLaunch D0 = A0, D1 = A1;
    D0.any = D1.any = false;
    rc = apply_launch(P0, D0, st);
    return rc != EMG_OK ? rc : apply_launch(P1, D1, st);
}

static int apply_grouped_impl(int opt, float* table, int64_t n_rows, int64_t ld, int32_t k_int, float* state0,
                              float* state1, int32_t* tag, int32_t step, const float* contrib, int64_t ldc,
                              int64_t n_contrib, int32_t skip_single, const float* hyper, double* lp_accum,
                              void* workspace, int64_t workspace_bytes, bool factored, void* stream) {
    EMG_REQUIRE(hyper, "emg_apply_grouped: null hyper");
    emg_apply_args a{};
    a.opt = opt; a.k_int = k_int; a.table = table; a.n_rows = n_rows; a.ld = ld; a.state0 = state0; a.state1 = state1;
    a.tag = tag; a.step = step; a.skip_single = skip_single; a.contrib = contrib; a.ldc = ldc; a.n_contrib = n_contrib;
    for (int i = 0; i < 8; ++i) a.hyper[i] = hyper[i];
    a.lp_accum = lp_accum; a.workspace = workspace; a.workspace_bytes = workspace_bytes; a.factored = factored ? 1 : 0;
    return emg_apply_grouped_ex(&a, stream);
}

extern "C" int emg_apply_grouped(int opt, float* table, int64_t n_rows, int64_t ld, int32_t k_int, float* state0,
                                 float* state1, int32_t* tag, int32_t step, const float* contrib, int64_t ldc,
                                 int64_t n_contrib, int32_t skip_single, const float* hyper, double* lp_accum,
                                 void* workspace, int64_t workspace_bytes, void* stream) {
    return apply_grouped_impl(opt, table, n_rows, ld, k_int, state0, state1, tag, step, contrib, ldc, n_contrib, skip_single,
                              hyper, lp_accum, workspace, workspace_bytes, false, stream);
}

extern "C" int emg_apply_grouped_factored(int opt, float* table, int64_t n_rows, int64_t ld, int32_t k_int, float* state0,
                                          float* state1, int32_t* tag, int32_t step, const float* contrib, int64_t ldc,
                                          int64_t n_contrib, int32_t skip_single, const float* hyper, double* lp_accum,
                                          void* workspace, int64_t workspace_bytes, void* stream) {
    return apply_grouped_impl(opt, table, n_rows, ld, k_int, state0, state1, tag, step, contrib, ldc, n_contrib, skip_single,
                              hyper, lp_accum, workspace, workspace_bytes, true, stream);
}

extern "C" int emg_apply_rows(int opt, float* table, int64_t n_rows, int64_t ld, int32_t k_int, float* state0,
                              float* state1, int32_t* tag, int32_t step, const float* contrib, int64_t ldc,
                              const int32_t* dest, int64_t n_contrib, const float* hyper, void* workspace,
                              int64_t workspace_bytes, void* stream) {
    EMG_REQUIRE(n_contrib == 0 || dest, "emg_apply_rows: null dest");
    EMG_REQUIRE(n_rows > 0, "emg_apply_rows: bad table arguments");
    int rc = emg_group_dest(dest, n_contrib, n_rows, workspace, workspace_bytes, nullptr, stream);
    if (rc != EMG_OK) return rc;
    return emg_apply_grouped(opt, table, n_rows, ld, k_int, state0, state1, tag, step, contrib, ldc, n_contrib, 0,
                             hyper, nullptr, workspace, workspace_bytes, stream);
}

template <int OPT>
static void launch_replay_opt(bool catchup, int lpk, const ReplayParams& P, dim3 grid, hipStream_t st) {
#define EMG_RP(K_) do { if (catchup) hipLaunchKernelGGL((deferred_catchup_kernel<OPT, K_>), grid, dim3(256), 0, st, P); \
                        else hipLaunchKernelGGL((deferred_materialize_kernel<OPT, K_>), grid, dim3(256), 0, st, P); } while (0)
    if (lpk == 0) EMG_RP(0); else if (lpk == 1) EMG_RP(1); else EMG_RP(2);
#undef EMG_RP
}
static void launch_catchup_rows_lag(int trips, const ReplayParams& P, dim3 grid, hipStream_t st) {
    if (trips == 1) hipLaunchKernelGGL((deferred_catchup_rows_kernel<EMG_OPT_ADAM, 0, 1, true>), grid, dim3(256), 0, st, P);
    else if (trips == 2) hipLaunchKernelGGL((deferred_catchup_rows_kernel<EMG_OPT_ADAM, 0, 2, true>), grid, dim3(256), 0, st, P);
    else hipLaunchKernelGGL((deferred_catchup_rows_kernel<EMG_OPT_ADAM, 0, 4, true>), grid, dim3(256), 0, st, P);
}
template <int OPT, int LPK>
static void launch_catchup_rows(int trips, const ReplayParams& P, dim3 grid, hipStream_t st) {
    if (trips == 1) hipLaunchKernelGGL((deferred_catchup_rows_kernel<OPT, LPK, 1>), grid, dim3(256), 0, st, P);
    else if (trips == 2) hipLaunchKernelGGL((deferred_catchup_rows_kernel<OPT, LPK, 2>), grid, dim3(256), 0, st, P);
    else hipLaunchKernelGGL((deferred_catchup_rows_kernel<OPT, LPK, 4>), grid, dim3(256), 0, st, P);
}
static bool env_replay_rows() {
    static const bool on = [] { const char* e = getenv("EMG_REPLAY_ROWS"); return !(e && e[0] == '0'); }();
    return on;
}
static void launch_replay(bool catchup, const ReplayParams& P, dim3 grid, hipStream_t st) {
    const int lpk = P.opt.lp_lambda == 0.f ? 0 : (P.opt.lp_p <= 3 ? 1 : 2);
    const bool aligned = aligned16(P.table) && (!P.s0 || aligned16(P.s0)) && (!P.s1 || aligned16(P.s1));
    if (catchup && lpk != 2 && P.k_int % 4 == 0 && P.ld % 4 == 0 && P.k_int <= 1024 && aligned && env_replay_rows()) {
        const int trips = (int)cdiv((int64_t)P.k_int / 4, 64);
        if (P.lag) { launch_catchup_rows_lag(trips, P, grid, st); return; }
#define EMG_RR(O_) do { if (lpk == 0) launch_catchup_rows<O_, 0>(trips, P, grid, st); else if (P.opt.lp_p == 2) launch_catchup_rows<O_, 3>(trips, P, grid, st); \
                        else launch_catchup_rows<O_, 1>(trips, P, grid, st); } while (0)
        switch (P.opt.opt) {
            case EMG_OPT_SGD: EMG_RR(EMG_OPT_SGD); break;
            case EMG_OPT_MOMENTUM: EMG_RR(EMG_OPT_MOMENTUM); break;
            case EMG_OPT_ADAGRAD: EMG_RR(EMG_OPT_ADAGRAD); break;
            default: EMG_RR(EMG_OPT_ADAM); break;
        }
#undef EMG_RR
        return;
    }
    switch (P.opt.opt) {
        case EMG_OPT_SGD: launch_replay_opt<EMG_OPT_SGD>(catchup, lpk, P, grid, st); break;
        case EMG_OPT_MOMENTUM: launch_replay_opt<EMG_OPT_MOMENTUM>(catchup, lpk, P, grid, st); break;
        case EMG_OPT_ADAGRAD: launch_replay_opt<EMG_OPT_ADAGRAD>(catchup, lpk, P, grid, st); break;
        default: launch_replay_opt<EMG_OPT_ADAM>(catchup, lpk, P, grid, st); break;
    }
}

static int replay_params(ReplayParams& P, int opt, float* table, int64_t n_rows, int64_t ld, int32_t k_int, float* state0, float* state1,
                         int32_t* tag, const float* hyper, const float* lr_hist, int32_t upto_step, double* lp_accum) {
    EMG_REQUIRE(opt >= EMG_OPT_SGD && opt <= EMG_OPT_ADAM, "emg_deferred_catchup / materialize: unknown optimizer %d", opt);
    EMG_REQUIRE(table && tag && hyper && lr_hist, "emg_deferred_catchup / materialize: null pointer");
    EMG_REQUIRE(!(opt == EMG_OPT_MOMENTUM || opt == EMG_OPT_ADAGRAD) || state0, "emg_deferred_catchup / materialize: optimizer needs state0");
    EMG_REQUIRE(opt != EMG_OPT_ADAM || (state0 && state1), "emg_deferred_catchup / materialize: adam needs both state tables");
    EMG_REQUIRE(n_rows > 0 && ld >= k_int && k_int > 0 && upto_step >= 0, "emg_deferred_catchup / materialize: bad arguments");
    EMG_REQUIRE(hyper[6] == 0.f || hyper[7] >= 1.f, "emg_deferred_catchup / materialize: LP needs p >= 1");
    P = ReplayParams{};
    P.table = table; P.s0 = state0; P.s1 = state1; P.tag = tag; P.n_rows = n_rows; P.ld = ld; P.k_int = k_int;
    P.opt = make_opt_params(opt, hyper);
    P.lr_hist = lr_hist; P.upto = upto_step; P.lp_accum = lp_accum;
    return EMG_OK;
}

extern "C" int emg_deferred_catchup(int opt, float* table, int64_t n_rows, int64_t ld, int32_t k_int, float* state0, float* state1,
                                    int32_t* tag, const float* hyper, const float* lr_hist, int32_t upto_step, double* lp_accum,
                                    const void* workspace, int64_t workspace_bytes, int64_t layout_n, int32_t w_only, int64_t skip_single_from,
                                    void* stream) {
    ReplayParams P;
    int rc = replay_params(P, opt, table, n_rows, ld, k_int, state0, state1, tag, hyper, lr_hist, upto_step, lp_accum);
    if (rc != EMG_OK) return rc;
    EMG_REQUIRE(!w_only || (opt == EMG_OPT_ADAM && hyper[6] == 0.f), "emg_deferred_catchup: w_only is Adam's, without a regulariser");
    P.lag = w_only ? 1 : 0;
    EMG_REQUIRE(workspace && layout_n > 0, "emg_deferred_catchup: needs the grouping workspace of the batch (emg_prepare_batch)");
    GroupWs w;
    rc = group_ws_layout(const_cast<void*>(workspace), workspace_bytes, layout_n, n_rows, 0, &w);
    if (rc != EMG_OK) return rc;
    EMG_REQUIRE(w.counting, "emg_deferred_catchup: needs the counting grouping (segment descriptors)");
    P.multi = w.multi; P.single = w.single; P.tasks = w.tasks; P.keys = w.keys; P.counters = w.counters; P.task_cap = w.task_cap;
    P.vals = w.vals; P.single_from = skip_single_from >= 0 ? skip_single_from : -1;
    if (upto_step == 0) return EMG_OK;
    static const int64_t cap_env = getenv("EMG_CATCHUP_WAVES") ? atoll(getenv("EMG_CATCHUP_WAVES")) : 0;   // A/B aid
    // (4096 waves = 1024 workgroups: with a regulariser every workgroup ends on one double atomic to ONE address — 4096 of them
    //  were 16 of C3 + LP's 59 us catch-up; Adam's, without atomics, is 4 us shorter too: 0.090 -> 0.086)
    const int64_t cap = cap_env >= 256 ? cap_env : 4096;
    int64_t waves = layout_n / 2;
    waves = waves < 256 ? 256 : (waves > cap ? cap : waves);
    launch_replay(true, P, dim3((unsigned)cdiv(waves, 4)), (hipStream_t)stream);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

extern "C" int emg_deferred_materialize(int opt, float* table, int64_t n_rows, int64_t ld, int32_t k_int, float* state0, float* state1,
                                        int32_t* tag, const float* hyper, const float* lr_hist, int32_t upto_step, double* lp_accum,
                                        void* stream) {
    ReplayParams P;
    int rc = replay_params(P, opt, table, n_rows, ld, k_int, state0, state1, tag, hyper, lr_hist, upto_step, lp_accum);
    if (rc != EMG_OK) return rc;
    if (upto_step == 0) return EMG_OK;
    launch_replay(false, P, dim3(untouched_blocks(n_rows)), (hipStream_t)stream);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}
