// emg_score_kernels.hpp — device code of the gather + score kernels (see emg_score.hip for the design): shared by
// emg_score.hip (forward / backward forms, dispatch, C-ABI) and emg_fused_m*.hip (one translation unit per model for the
// fused in-place forms: they are the largest kernels of the library and compile in parallel that way).
#pragma once
#include "emg_group_kernels.hpp"

// Every multiply-add of the score and gradient formulas is written out (fmaf where one rounding is meant): the compiler
// contracts a*b + c*d either way round, and differently in different instantiations of the same template — the in-place
// and the contribution-row forms of one step must produce the same bits.
#pragma clang fp contract(off)

namespace emg {

template <int MODEL>
struct is_complex {
    static constexpr bool value = (MODEL == EMG_COMPLEX || MODEL == EMG_HOLE);
};

// A model row in registers: real models E floats; complex models [re(E) | im(E)].
template <int MODEL, int W, int NV>
struct Row {
    static constexpr int E = W * NV;
    static constexpr int N = is_complex<MODEL>::value ? 2 * E : E;
    float x[N];
};

template <int MODEL, int W, int NV, int LPG, bool ZERO = true>
__device__ __forceinline__ void load_row(Row<MODEL, W, NV>& r, const float* __restrict__ base, int lg, int nchunks,
                                         int khalf) {
    constexpr int E = W * NV;
    RowTile<W, NV> t;
    load_tile<W, NV, LPG, ZERO>(t, base, lg, nchunks);
#pragma unroll
    for (int e = 0; e < E; ++e) r.x[e] = t.x[e];
    if constexpr (is_complex<MODEL>::value) {
        load_tile<W, NV, LPG, ZERO>(t, base + khalf, lg, nchunks);
#pragma unroll
        for (int e = 0; e < E; ++e) r.x[E + e] = t.x[e];
    }
}

#ifndef EMG_INPLACE_NT
#define EMG_INPLACE_NT 0
#endif
#ifndef EMG_STREAM_STORES
#define EMG_STREAM_STORES 1
#endif
// (rows of the contribution buffers: written here, read once by the apply — streamed past the caches)
template <int MODEL, int W, int NV, int LPG>
__device__ __forceinline__ void store_row(const Row<MODEL, W, NV>& r, float* __restrict__ base, int lg, int nchunks,
                                          int khalf) {
    constexpr int E = W * NV;
    constexpr bool STREAM = EMG_STREAM_STORES != 0;
    RowTile<W, NV> t;
#pragma unroll
    for (int e = 0; e < E; ++e) t.x[e] = r.x[e];
    store_tile<W, NV, LPG, STREAM>(t, base, lg, nchunks);
    if constexpr (is_complex<MODEL>::value) {
#pragma unroll
        for (int e = 0; e < E; ++e) t.x[e] = r.x[E + e];
        store_tile<W, NV, LPG, STREAM>(t, base + khalf, lg, nchunks);
    }
}

// per-lane partial of the k-reduction for roles (a = subject row, p = relation row, b = object row)
template <int MODEL, int W, int NV>
__device__ __forceinline__ float partial_score(const Row<MODEL, W, NV>& a, const Row<MODEL, W, NV>& p,
                                               const Row<MODEL, W, NV>& b) {
    constexpr int E = W * NV;
    float acc = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if constexpr (MODEL == EMG_TRANSE_L1) {
            acc += fabsf((a.x[e] + p.x[e]) - b.x[e]);  // TransE.py:210: e_s + e_p - e_o, ord=1
        } else if constexpr (MODEL == EMG_TRANSE_L2) {
            const float d = (a.x[e] + p.x[e]) - b.x[e];
            acc = fmaf(d, d, acc);
        } else if constexpr (MODEL == EMG_DISTMULT) {
            acc = fmaf(a.x[e] * p.x[e], b.x[e], acc);  // DistMult.py:201
        } else {
            const float sr = a.x[e], si = a.x[E + e], pr = p.x[e], pi = p.x[E + e], orr = b.x[e], oi = b.x[E + e];
            // ComplEx.py:293-297
            acc = fmaf(pr * sr, orr, acc);
            acc = fmaf(pr * si, oi, acc);
            acc = fmaf(pi * sr, oi, acc);
            acc = fmaf(-(pi * si), orr, acc);
        }
    }
    return acc;
}

template <int MODEL>
__device__ __forceinline__ float finalize_score(float sum, float scale, int flags) {
    if constexpr (MODEL == EMG_TRANSE_L1) return -sum;
    if constexpr (MODEL == EMG_TRANSE_L2) return (flags & EMG_SCORE_PARTIAL) ? sum : -sqrtf(sum);
    if constexpr (MODEL == EMG_HOLE) return (flags & EMG_SCORE_PARTIAL) ? sum : scale * sum;
    return sum;
}

// inner coefficient from g = dL/dscore and the k-reduced sum (TransE-L2: nrm = sqrt(sum of squares))
template <int MODEL>
__device__ __forceinline__ float inner_coef(float g, float nrm, float scale) {
    if constexpr (MODEL == EMG_TRANSE_L2) return nrm > 0.f ? g / nrm : 0.f;
    if constexpr (MODEL == EMG_HOLE) return g * scale;
    return g;
}

// ga += dscore/da * gi etc. for roles (a, p, b)
template <int MODEL, int W, int NV>
__device__ __forceinline__ void accum_grads(const Row<MODEL, W, NV>& a, const Row<MODEL, W, NV>& p,
                                            const Row<MODEL, W, NV>& b, float gi, Row<MODEL, W, NV>& ga,
                                            Row<MODEL, W, NV>& gp, Row<MODEL, W, NV>& gb) {
    constexpr int E = W * NV;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if constexpr (MODEL == EMG_TRANSE_L1) {
            const float t = gi * sgnf((a.x[e] + p.x[e]) - b.x[e]);
            ga.x[e] -= t; gp.x[e] -= t; gb.x[e] += t;
        } else if constexpr (MODEL == EMG_TRANSE_L2) {
            const float t = gi * ((a.x[e] + p.x[e]) - b.x[e]);
            ga.x[e] -= t; gp.x[e] -= t; gb.x[e] += t;
        } else if constexpr (MODEL == EMG_DISTMULT) {
            ga.x[e] = fmaf(gi, p.x[e] * b.x[e], ga.x[e]);
            gp.x[e] = fmaf(gi, a.x[e] * b.x[e], gp.x[e]);
            gb.x[e] = fmaf(gi, a.x[e] * p.x[e], gb.x[e]);
        } else {
            const float sr = a.x[e], si = a.x[E + e], pr = p.x[e], pi = p.x[E + e], orr = b.x[e], oi = b.x[E + e];
            ga.x[e] = fmaf(gi, fmaf(pr, orr, pi * oi), ga.x[e]);
            ga.x[E + e] = fmaf(gi, fmaf(pr, oi, -(pi * orr)), ga.x[E + e]);
            gp.x[e] = fmaf(gi, fmaf(sr, orr, si * oi), gp.x[e]);
            gp.x[E + e] = fmaf(gi, fmaf(sr, oi, -(si * orr)), gp.x[E + e]);
            gb.x[e] = fmaf(gi, fmaf(pr, sr, -(pi * si)), gb.x[e]);
            gb.x[E + e] = fmaf(gi, fmaf(pr, si, pi * sr), gb.x[E + e]);
        }
    }
}

// A value that is identical in all lanes of a group.  When the group IS the wave (LPG == 64) tell the
// compiler so (readfirstlane -> SGPR): branches on it become scalar branches instead of being if-converted
// into both-sides-plus-select, which doubles the live registers of the role-dependent code.
template <int LPG>
__device__ __forceinline__ int uniform_if_wave(int v) {
    if constexpr (LPG == 64) return __builtin_amdgcn_readfirstlane(v);
    return v;
}

// value held by lane `first + j` of the wave (first = the group's lane 0), j uniform within the group
template <int LPG>
__device__ __forceinline__ int group_lane_value(int v, int first, int j) {
    if constexpr (LPG == 64) return __builtin_amdgcn_readlane(v, j);
    return __shfl(v, first + j, 64);
}

struct GroupParams {
    const float* ent; int64_t n_ent; int64_t ld_ent;
    const float* rel; int64_t n_rel; int64_t ld_rel;
    int32_t k_int; int32_t khalf; int32_t nchunks; float scale;
    int32_t width;                                       // columns (per half for complex models) this launch covers; 0 = all
    const int32_t* pos; int64_t B; int32_t eta; const int32_t* codes; int32_t flags;
    float* scores_pos; float* scores_neg;
    // backward / fused
    const float* g_pos; const float* g_neg;              // external dL/dscore (fused_loss < 0)
    const float* bw_scores_pos; const float* bw_scores_neg;  // global final scores (TransE-L2 on a k-slice)
    int32_t fused_loss; float margin; double* loss_accum; uint32_t loss_mask;   // loss_mask: workgroup b adds to loss_accum[b & loss_mask]
    float* contrib_ent; float* contrib_rel; int64_t ldc;
    const uint8_t* single_ent;                           // per entity-contribution slot: 1 = update in place
    float* ent_rw; float* ent_state0; float* ent_state1; int32_t* tag_ent; int32_t step;
    OptParams opt;
    double* lp_accum;                                    // IP 3: += sum |w_pre|^p over the rows updated in place
    FactorView fac;                                      // fac.coef != nullptr: FACTORED contributions (bilinear models), see emg_backward_args
    const StepCtl* ctl;                                  // graph node: batch rows, step number and learning rates from the device record
    int32_t window;                                      // stateful in-place updates through the window forms (IP 4 / 5 / 6)
    const float* lr_hist; int32_t upto;                  // IP 6 (Adam, deferred dense pass): learning rate of every step; rows are replayed to step `upto`
};

// In-place forms (template parameter IP of the backward / fused kernels):
//   0  none: every gradient row goes to the contribution buffer
//   1  plain SGD                       3  plain SGD folding an LP regulariser (p <= 3)
//   2  stateful optimizer, state rows read chunk by chunk at the update (a dependent round trip per row: the fallback shape)
//   4  one state row (momentum / Adagrad)   }  round 4: the state rows of a singleton's replacement entity TRAVEL WITH ITS TABLE ROW
//   5  two state rows (Adam, dense pass)    }  in the rolling window (a slot that is no singleton re-reads the table row instead:
//   6  two state rows, LAGGING (Adam with   }  cache hits, and no load sits under a condition) — the update waits for nothing.
//      the deferred dense pass): (w, m, v) of a singleton are as of tag[row]; the missed steps tag+1 .. upto are replayed in
//      registers BEFORE the row is scored (the dense pass's own update with g = 0 and each step's lr_t) — what
//      emg_deferred_catchup does for the other rows with a pass of its own.
//   The subject / object slots (2 of a group's 22): forms 4 / 5 update their singletons in place at the end of the group, state read
//   chunk by chunk (form 2's way).  Form 6: a singleton subject / object row lags like a singleton negative — its (w, m, v) are
//   fetched together at the group's START, replayed, the queries are built from the replayed row, and the three replayed rows
//   wait in LDS (6 rows per wave) for the update at the group's end: no register lives across the loop over the negatives.
//   (Measured, C3 + Adagrad: s / o through the apply in form 4 — three waves per SIMD instead of two — 0.65 against 0.55-0.60 ms
//   per step: the scoring kernel no faster, the apply 0.08 ms longer.)
template <int IP>
struct ip_traits {
    static constexpr int n_state = IP == 4 ? 1 : ((IP == 5 || IP == 6) ? 2 : 0);
    static constexpr bool window_state = n_state != 0;
    static constexpr bool replay = IP == 6;
    static constexpr bool so_inplace = IP != 0;              // the subject / object slots' singletons are updated in place too
    // 7 (round 5): form 3 — plain SGD with the LP regulariser folded in — under the DEFERRED dense pass: a singleton negative's row is as
    // of tag[row]; the regulariser's steps it missed (w -= lr_i * lambda p |w|^(p-1) sgn w, step by step: the dense pass's bits) are
    // replayed on the row in registers before it is scored, then form 3's update writes it: the row is read once and written once
    // where emg_deferred_catchup read and wrote it first (two row moves per singleton, 0.56 GB of C3 + LP's step)
    static constexpr bool lp_replay = IP == 7;
    static constexpr int chunkwise = (IP == 4 || IP == 5) ? 2 : (IP == 7 ? 3 : IP);   // the form inplace_update runs (negatives of 1 / 2 / 3 / 7, s / o slots)
};

#ifndef EMG_BW_THREADS
#define EMG_BW_THREADS 256
#endif
constexpr int kThreads = EMG_BW_THREADS;
constexpr int kUnroll = 4;

// ---------------------------------------------------------------------------------------------
// forward: scores_pos[g], scores_neg[j*B+g]
// ---------------------------------------------------------------------------------------------
template <int MODEL, int W, int NV, int LPG>
__global__ __launch_bounds__(kThreads) void train_forward_kernel(const GroupParams P) {
    using R = Row<MODEL, W, NV>;
    const int lg = threadIdx.x % LPG;
    int64_t g = ((int64_t)blockIdx.x * kThreads + threadIdx.x) / LPG;
    const bool active = g < P.B;
    if (!active) g = P.B - 1;  // keep the whole wave convergent for the shuffles

    const int32_t s = P.pos[3 * g + 0], p = P.pos[3 * g + 1], o = P.pos[3 * g + 2];
    R rs, rp, ro;
    load_row<MODEL, W, NV, LPG>(rs, P.ent + (int64_t)s * P.ld_ent, lg, P.nchunks, P.khalf);
    load_row<MODEL, W, NV, LPG>(rp, P.rel + (int64_t)p * P.ld_rel, lg, P.nchunks, P.khalf);
    load_row<MODEL, W, NV, LPG>(ro, P.ent + (int64_t)o * P.ld_ent, lg, P.nchunks, P.khalf);

    {
        const float sum = group_sum<LPG>(partial_score<MODEL, W, NV>(rs, rp, ro));
        if (active && lg == 0) P.scores_pos[g] = finalize_score<MODEL>(sum, P.scale, P.flags);
    }

    for (int j0 = 0; j0 < P.eta; j0 += kUnroll) {
        int32_t code[kUnroll];
        R re[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int j = min(j0 + u, P.eta - 1);
            code[u] = uniform_if_wave<LPG>(P.codes[(int64_t)j * P.B + g]);
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int32_t repl = code[u] & 0x7fffffff;
            load_row<MODEL, W, NV, LPG>(re[u], P.ent + (int64_t)repl * P.ld_ent, lg, P.nchunks, P.khalf);
        }
        float part[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const bool keep_s = code[u] < 0;  // bit 31
            if (keep_s) part[u] = partial_score<MODEL, W, NV>(rs, rp, re[u]);
            else part[u] = partial_score<MODEL, W, NV>(re[u], rp, ro);
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const float sum = group_sum<LPG>(part[u]);
            const int j = j0 + u;
            if (active && lg == 0 && j < P.eta)
                P.scores_neg[(int64_t)j * P.B + g] = finalize_score<MODEL>(sum, P.scale, P.flags);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// in-place optimizer update of one table row from registers (singleton destinations)
// ---------------------------------------------------------------------------------------------
//   IP == 1: plain SGD (no state; stays lean)   IP == 2: any optimizer (state RMW; elements are fenced with
//   sched_barrier so the compiler does not interleave eight sqrt/div expansions and blow up the VGPR budget)
// IP 7: the LP regulariser's missed steps from + 1 .. P.upto of a lagging row, in registers — emg_apply.hip::replay_finish's arithmetic
// for plain SGD (g = the regulariser's gradient alone, w -= lr_i g; sum |w|^p of every replayed step into lp_acc), so the dense pass's
// bits.  `from` is wave-uniform: the learning rates come by scalar loads (no vector-memory operation joins the row window's queue).
template <int MODEL, int W, int NV, int LPG>
__device__ __forceinline__ void lp_replay_row(const GroupParams& P, int from, float lr_lane, Row<MODEL, W, NV>& r, int lg, float& lp_acc) {
    constexpr int E = W * NV;
    constexpr int HALVES = is_complex<MODEL>::value ? 2 : 1;
    OptParams opt = P.opt;
    float accs[HALVES * NV];   // (per 16-byte chunk of the lane, over all steps: replay_finish's association)
#pragma unroll
    for (int q = 0; q < HALVES * NV; ++q) accs[q] = 0.f;
    for (int i = from + 1; i <= P.upto; ++i) {
        // lane l of the group holds the learning rate of step upto - l (one vector load per GROUP, in its prologue): a step's rate is a
        // v_readlane; only a row that lags by more than 64 steps reads the table (a scalar load per step, its latency exposed — the
        // first version did that for every step: the kernel 0.268 -> 0.322 ms at C3 + LP)
        const int back = P.upto - i;
        opt.lr = back < 64 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lr_lane), back)) : P.lr_hist[i];
#pragma unroll
        for (int h = 0; h < HALVES; ++h)
#pragma unroll
            for (int it = 0; it < NV; ++it)
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    float& x = r.x[h * E + it * W + w];
                    float g = 0.f;
                    if (opt.lp_p == 2) lp_fold_p2(opt, x, g, accs[h * NV + it]); else lp_fold_p123(opt, x, g, accs[h * NV + it]);
                    x = opt_sgd_elem(opt, x, g);
                }
    }
#pragma unroll
    for (int h = 0; h < HALVES; ++h)
#pragma unroll
        for (int it = 0; it < NV; ++it)
            lp_acc += (lg + it * LPG < P.nchunks) ? accs[h * NV + it] : 0.f;   // (lanes past the row's end hold copies: replayed harmlessly, uncounted)
}

template <int MODEL, int W, int NV, int LPG, int IP>
__device__ __forceinline__ void inplace_update(const GroupParams& P, int64_t row, const Row<MODEL, W, NV>& cur,
                                               const Row<MODEL, W, NV>& grad, int lg, float& lp_acc) {
    // chunk-wise (one 16-byte chunk live at a time) so the singleton path costs almost no extra VGPRs:
    // occupancy is what keeps enough row loads in flight for this HBM-bound kernel
    constexpr int E = W * NV;
    constexpr int HALVES = is_complex<MODEL>::value ? 2 : 1;
    float* wrow = P.ent_rw + row * P.ld_ent;
    float* s0row = (IP == 2 && P.ent_state0) ? P.ent_state0 + row * P.ld_ent : nullptr;   // (IP 1 / 3: plain SGD, no state)
    float* s1row = (IP == 2 && P.ent_state1) ? P.ent_state1 + row * P.ld_ent : nullptr;
#pragma unroll
    for (int h = 0; h < HALVES; ++h) {
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const int c = lg + it * LPG;
            if (c < P.nchunks) {
                const int off = h * P.khalf + c * W;
                float wv[W], s0v[W], s1v[W];
                if constexpr (W == 4) {
                    if (s0row) { const float4 t = *reinterpret_cast<const float4*>(s0row + off); s0v[0] = t.x; s0v[1] = t.y; s0v[2] = t.z; s0v[3] = t.w; }
                    if (s1row) { const float4 t = *reinterpret_cast<const float4*>(s1row + off); s1v[0] = t.x; s1v[1] = t.y; s1v[2] = t.z; s1v[3] = t.w; }
                } else {
                    if (s0row) s0v[0] = s0row[off];
                    if (s1row) s1v[0] = s1row[off];
                }
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    wv[w] = cur.x[h * E + it * W + w];
                    const float g = grad.x[h * E + it * W + w];
                    if constexpr (IP == 1) {
                        wv[w] = opt_sgd_elem(P.opt, wv[w], g);
                    } else if constexpr (IP == 3) {   // SGD with the LP regulariser folded in: the rule of emg_apply_grouped's finish
                        float gl = g;
                        lp_fold_p123(P.opt, wv[w], gl, lp_acc);
                        wv[w] = opt_sgd_elem(P.opt, wv[w], gl);
                    } else {
                        opt_update_elem(P.opt, wv[w], g, &s0v[w], &s1v[w]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if constexpr (W == 4) {
#if EMG_INPLACE_NT
                    typedef float nt_float4 __attribute__((ext_vector_type(4)));
                    const nt_float4 nv = {wv[0], wv[1], wv[2], wv[3]};
                    __builtin_nontemporal_store(nv, reinterpret_cast<nt_float4*>(wrow + off));
#else
                    *reinterpret_cast<float4*>(wrow + off) = make_float4(wv[0], wv[1], wv[2], wv[3]);
#endif
                    if (s0row) *reinterpret_cast<float4*>(s0row + off) = make_float4(s0v[0], s0v[1], s0v[2], s0v[3]);
                    if (s1row) *reinterpret_cast<float4*>(s1row + off) = make_float4(s1v[0], s1v[1], s1v[2], s1v[3]);
                } else {
                    wrow[off] = wv[0];
                    if (s0row) s0row[off] = s0v[0];
                    if (s1row) s1row[off] = s1v[0];
                }
            }
        }
    }
    if (P.tag_ent && lg == 0) P.tag_ent[row] = P.step;
}

// EMG_OPT_GROUP: elements whose update chains (mul -> v_sqrt -> add -> v_rcp -> mul -> sub) the scheduler may interleave in the window
// forms — one at a time leaves every transcendental's wait states as s_nop (56 of the ~600 instructions per negative of form 6)
#ifndef EMG_OPT_GROUP
#define EMG_OPT_GROUP 1
#endif
// The same update with the state rows ALREADY IN REGISTERS (IP 4 / 5 / 6: they arrived with the table row): nothing is waited for.
template <int MODEL, int W, int NV, int LPG, int NS>
__device__ __forceinline__ void inplace_update_regs(const GroupParams& P, const OptParams& opt, int64_t row, const Row<MODEL, W, NV>& cur,
                                                    const Row<MODEL, W, NV>& grad, Row<MODEL, W, NV>& s0, Row<MODEL, W, NV>& s1, int lg) {
    static_assert(W == 4, "16-byte rows");
    constexpr int E = W * NV;
    constexpr int HALVES = is_complex<MODEL>::value ? 2 : 1;
    float* wrow = P.ent_rw + row * P.ld_ent;
    float* s0row = P.ent_state0 + row * P.ld_ent;
    float* s1row = NS == 2 ? P.ent_state1 + row * P.ld_ent : nullptr;
#pragma unroll
    for (int h = 0; h < HALVES; ++h) {
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const int c = lg + it * LPG;
            float wv[W];
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const int e = h * E + it * W + w;
                wv[w] = cur.x[e];
                opt_update_elem(opt, wv[w], grad.x[e], &s0.x[e], NS == 2 ? &s1.x[e] : nullptr);
                if (w % EMG_OPT_GROUP == EMG_OPT_GROUP - 1) __builtin_amdgcn_sched_barrier(0);   // (EMG_OPT_GROUP sqrt / reciprocal chains at a time: see inplace_update)
            }
            if (c < P.nchunks) {
                const int off = h * P.khalf + c * W;
                *reinterpret_cast<float4*>(wrow + off) = make_float4(wv[0], wv[1], wv[2], wv[3]);
                *reinterpret_cast<float4*>(s0row + off) = make_float4(s0.x[h * E + it * W], s0.x[h * E + it * W + 1], s0.x[h * E + it * W + 2], s0.x[h * E + it * W + 3]);
                if constexpr (NS == 2)
                    *reinterpret_cast<float4*>(s1row + off) = make_float4(s1.x[h * E + it * W], s1.x[h * E + it * W + 1], s1.x[h * E + it * W + 2], s1.x[h * E + it * W + 3]);
            }
        }
    }
    if (P.tag_ent && lg == 0) P.tag_ent[row] = P.step;
}

// IP 6: (w, m, v) of a row last written at step `from`, brought to step P.upto in registers: the dense pass's update of each
// missed step (g = 0, that step's lr_t) — emg_apply.hip::replay_finish's arithmetic, so the same bits.  lrv: lane l holds the
// learning rate of step from + 1 + l (fetched with the row); steps beyond 64 read the table.  Lanes past the row's end keep w.
template <int MODEL, int W, int NV, int LPG>
__device__ __forceinline__ void replay_in_window(const GroupParams& P, const OptParams& opt0, int32_t from, float lrv, Row<MODEL, W, NV>& w,
                                                 Row<MODEL, W, NV>& m, Row<MODEL, W, NV>& v, int lg) {
    static_assert(LPG == 64, "the replay's step count is per wave");
    constexpr int E = W * NV;
    OptParams opt = opt0;
    const int n = P.upto - from;
    auto one_step = [&](float lr) {
        opt.lr = opt.lr_t = lr;
#pragma unroll
        for (int e = 0; e < Row<MODEL, W, NV>::N; ++e) {
            const bool on = lg + ((e % E) / W) * LPG < P.nchunks;
            float wv = w.x[e];
            adam_zero_grad_elem(opt, wv, m.x[e], v.x[e]);   // (the dense pass's update of a row with no gradient: the same values)
            w.x[e] = on ? wv : w.x[e];
            if (e % EMG_OPT_GROUP == EMG_OPT_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
        }
    };
    // the first 64 steps: learning rates from the lane register — NO memory operation in this loop (a load here, even one never
    // executed, would make every wait of the rolling window a vmcnt(0))
    const int n1 = n < 64 ? n : 64;
    for (int i = 0; i < n1; ++i) one_step(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(lrv), i)));
    for (int i = 64; i < n; ++i) one_step(P.lr_hist[from + 1 + i]);   // (a row untouched for more than 64 steps)
}

// ---------------------------------------------------------------------------------------------
// backward / fused kernel, register-lean form.
//
// Every model here is (bi)linear enough that the loop over a positive's negatives needs only
//   q_o, q_s : the two HOISTED query rows (object side from (s,p), subject side from (p,o)); a negative's
//              score is <q, e> (TransE: -||q - e||), its gradient row is gi*q (TransE: +-gi*sgn/diff);
//   A_o, A_s : two ACCUMULATORS  sum_j gi_j * e_j  (TransE: sum_j t_j) over the object- / subject-corrupted
//              negatives; the gradients of the kept rows s, p, o are linear in them and are formed once,
//              after the loop, from s, p, o RE-LOADED at that point (L2-hot) instead of kept live.
// => 4 persistent rows instead of 6 + no per-negative role copies: ~100 VGPRs with FOUR replacement rows in
// flight per wave (occupancy x loads in flight is what an HBM-bound gather kernel lives on).
//   FUSED = true : scores, pair-local loss and dL/dscore are computed here (P.fused_loss)
//   FUSED = false: dL/dscore comes from P.g_pos / P.g_neg
//   IP    = 0: all rows to the contribution buffer; 1 (SGD) / 2 (stateful) / 3 (SGD + folded LP regulariser): singleton
//           destinations updated in place
// ---------------------------------------------------------------------------------------------
template <int MODEL, int W, int NV>
__device__ __forceinline__ void make_queries(const Row<MODEL, W, NV>& s, const Row<MODEL, W, NV>& p,
                                             const Row<MODEL, W, NV>& o, Row<MODEL, W, NV>& qo,
                                             Row<MODEL, W, NV>& qs) {
    constexpr int E = W * NV;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if constexpr (MODEL == EMG_TRANSE_L1 || MODEL == EMG_TRANSE_L2) {
            qo.x[e] = s.x[e] + p.x[e];  // (s+p) - e
            qs.x[e] = o.x[e] - p.x[e];  // (e+p) - o = e - (o-p)
        } else if constexpr (MODEL == EMG_DISTMULT) {
            qo.x[e] = p.x[e] * s.x[e];
            qs.x[e] = p.x[e] * o.x[e];
        } else {
            const float sr = s.x[e], si = s.x[E + e], pr = p.x[e], pi = p.x[E + e], orr = o.x[e], oi = o.x[E + e];
            qo.x[e] = fmaf(pr, sr, -(pi * si));   qo.x[E + e] = fmaf(pr, si, pi * sr);      // SURVEY B-2, object side
            qs.x[e] = fmaf(pr, orr, pi * oi);     qs.x[E + e] = fmaf(pr, oi, -(pi * orr));  // subject side
        }
    }
}

template <int MODEL, int W, int NV>
__device__ __forceinline__ float neg_partial(const Row<MODEL, W, NV>& q, const Row<MODEL, W, NV>& e) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < Row<MODEL, W, NV>::N; ++i) {
        if constexpr (MODEL == EMG_TRANSE_L1) acc += fabsf(q.x[i] - e.x[i]);
        else if constexpr (MODEL == EMG_TRANSE_L2) { const float d = q.x[i] - e.x[i]; acc = fmaf(d, d, acc); }
        else acc = fmaf(q.x[i], e.x[i], acc);
    }
    return acc;
}

// gradient row of the replacement entity + accumulator update.  OBJ: the object was replaced (d = q - e)
template <int MODEL, int W, int NV, bool OBJ>
__device__ __forceinline__ void neg_grads(const Row<MODEL, W, NV>& q, const Row<MODEL, W, NV>& e, float gi,
                                          Row<MODEL, W, NV>& row, Row<MODEL, W, NV>& acc) {
#pragma unroll
    for (int i = 0; i < Row<MODEL, W, NV>::N; ++i) {
        if constexpr (MODEL == EMG_TRANSE_L1 || MODEL == EMG_TRANSE_L2) {
            const float d = OBJ ? q.x[i] - e.x[i] : e.x[i] - q.x[i];
            const float t = (MODEL == EMG_TRANSE_L1) ? gi * sgnf(d) : gi * d;
            row.x[i] = OBJ ? t : -t;  // dscore/de = +sgn(d) when e is the object, -sgn(d) when it is the subject
            acc.x[i] += t;
        } else {
            row.x[i] = gi * q.x[i];
            acc.x[i] = fmaf(gi, e.x[i], acc.x[i]);
        }
    }
}

// gradients of the kept rows from the accumulators (+ the positive's own term, inner coefficient gp_i)
template <int MODEL, int W, int NV>
__device__ __forceinline__ void finish_grads(const Row<MODEL, W, NV>& s, const Row<MODEL, W, NV>& p,
                                             const Row<MODEL, W, NV>& o, const Row<MODEL, W, NV>& Ao,
                                             const Row<MODEL, W, NV>& As, float gp_i, Row<MODEL, W, NV>& gs,
                                             Row<MODEL, W, NV>& gp, Row<MODEL, W, NV>& go) {
    constexpr int E = W * NV;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if constexpr (MODEL == EMG_TRANSE_L1 || MODEL == EMG_TRANSE_L2) {
            const float d = (s.x[e] + p.x[e]) - o.x[e];
            const float tp = (MODEL == EMG_TRANSE_L1) ? gp_i * sgnf(d) : gp_i * d;
            gs.x[e] = -Ao.x[e] - tp;
            gp.x[e] = -Ao.x[e] - As.x[e] - tp;
            go.x[e] = As.x[e] + tp;
        } else if constexpr (MODEL == EMG_DISTMULT) {
            const float bo = fmaf(gp_i, o.x[e], Ao.x[e]);  // effective object row seen by (s,p)
            const float bs = fmaf(gp_i, s.x[e], As.x[e]);  // effective subject row seen by (p,o)
            gs.x[e] = p.x[e] * bo;
            go.x[e] = p.x[e] * bs;
            gp.x[e] = fmaf(s.x[e], bo, As.x[e] * o.x[e]);
        } else {
            const float sr = s.x[e], si = s.x[E + e], pr = p.x[e], pi = p.x[E + e], orr = o.x[e], oi = o.x[E + e];
            const float br = fmaf(gp_i, orr, Ao.x[e]), bi = fmaf(gp_i, oi, Ao.x[E + e]);  // b = Ao + gp_i*o
            const float ar = fmaf(gp_i, sr, As.x[e]), ai = fmaf(gp_i, si, As.x[E + e]);   // a = As + gp_i*s
            gs.x[e] = fmaf(pr, br, pi * bi);          gs.x[E + e] = fmaf(pr, bi, -(pi * br));   // GA(p, b)
            go.x[e] = fmaf(pr, ar, -(pi * ai));       go.x[E + e] = fmaf(pr, ai, pi * ar);      // GB(p, a)
            // GP(s, b) + GP(As, o)
            gp.x[e] = fmaf(sr, br, si * bi) + fmaf(As.x[e], orr, As.x[E + e] * oi);
            gp.x[E + e] = fmaf(sr, bi, -(si * br)) + fmaf(As.x[e], oi, -(As.x[E + e] * orr));
        }
    }
}

#ifndef EMG_BW_MINWAVES
#define EMG_BW_MINWAVES 1
#endif
// U = replacement rows in flight per wave and trip.  Measured on C3 (MI355X, fused kernel alone): U = 2: 0.302 ms,
// 4: 0.263, 5: 0.302, 6: 0.314, 8: 0.311, 10: 0.305; a software pipeline (next U rows in flight during the arithmetic
// of this trip) 0.285 (U = 2) / 0.300 (U = 4: 194 VGPRs, 2 waves/SIMD); workgroups of 64 / 128 / 512 threads 0.288 /
// 0.239 / 0.274 against 0.223 with 256 on the same box.  Four waves per SIMD (round 3, one box, step ms): U = 3 without KEEP
// forced to 128 VGPRs (44 B scratch) 0.392-0.394, U = 2 without KEEP (123 VGPRs) 0.373, U = 4 without KEEP (150 VGPRs, 3 waves)
// 0.374 against 0.369-0.370 as built: occupancy beyond 3 waves buys nothing, rows in flight per wave do.
#ifndef EMG_BW_U
#define EMG_BW_U 4
#endif
#ifndef EMG_BW_ROLL
#define EMG_BW_ROLL 1   // 0: A/B aid — U rows per trip, none in flight across trips
#endif

// ASYNC replacement rows (EMG_BW_ASYNC, bilinear fused forms with 16-byte rows): the row loads of the rolling window are
// inline assembly, so hipcc neither counts nor waits for them; the wait before a row's first use is written by hand:
// s_waitcnt vmcnt((U - 1) * pieces) — at most the loads of the U - 1 younger rows may still be in flight (stores issued
// since then only make the wait a little earlier than necessary; memory operations of a wave complete in order).  With
// compiler-visible loads every wait in this loop came out as vmcnt(0): the loop has branches whose sides issue different
// numbers of stores, and the compiler's count of "operations younger than this load" is its minimum over all paths.
#ifndef EMG_BW_ASYNC
#define EMG_BW_ASYNC 0
#endif
typedef float emg_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ emg_f4 vm_load16_async(const float* p) {
    emg_f4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ float vm_load4_async(const float* p) {
    float v;
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// registers written by an asynchronous load issued before a hand-written wait: no use of them may move above it
__device__ __forceinline__ void vm_landed(emg_f4& a) { asm volatile("" : "+v"(a) : : "memory"); }
__device__ __forceinline__ void vm_landed(float& a) { asm volatile("" : "+v"(a) : : "memory"); }
template <int N> __device__ __forceinline__ void vm_wait_only() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void vm_wait(emg_f4& a) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void vm_wait(emg_f4& a, emg_f4& b) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void vm_wait(emg_f4& a, emg_f4& b, emg_f4& c, emg_f4& d) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}

// KEEP: the s, p, o rows stay in registers across the loop over the negatives instead of being re-read at the end
// (the re-read was 1.10x the algorithmic traffic by PMC: after 20 replacement rows per wave on every CU of the XCD
// they are no longer in L2).  Only where it is free: instantiations whose occupancy does not drop (checked with
// -Rpass-analysis=kernel-resource-usage: 16-byte single-chunk rows; complex rows only in the fused in-place form,
// 140 -> 158 VGPRs at the same 3 waves/SIMD).
template <int MODEL, int W, int NV, bool FUSED, int IP>
struct keep_rows {
#ifndef EMG_BW_KEEP
#define EMG_BW_KEEP 1   // 0: A/B aid — never keep s, p, o across the loop
#endif
    // (not with two state rows per window slot — forms 5 / 6: the three kept rows are the registers DistMult's form 5 spilled)
    static constexpr bool value = EMG_BW_KEEP != 0 && W == 4 && NV == 1 && ip_traits<IP>::n_state != 2 &&
                                  (!is_complex<MODEL>::value || (FUSED && (IP == 1 || IP == 2)));
};
template <int MODEL, int W, int NV, int LPG, bool FUSED, int IP, int UW = EMG_BW_U>
__device__ __forceinline__ void train_backward_body(const GroupParams& P0, unsigned bx) {
    using R = Row<MODEL, W, NV>;
    GroupParams P = P0;
    if (P0.ctl) {   // a node of a captured step graph: which rows, which step, which learning rates come from the device record
        P.B = P0.ctl->B; P.pos = P0.pos + 3 * P0.ctl->start; P.step = P0.ctl->step;
        P.opt.lr = P0.ctl->hyper_ent[0]; P.opt.lr_t = P0.ctl->hyper_ent[5];
    }
    const int lg = threadIdx.x % LPG;
    int64_t g = ((int64_t)bx * kThreads + threadIdx.x) / LPG;
    if ((int64_t)bx * (kThreads / LPG) >= P.B) return;   // (a launch sized for the plan's capacity: workgroups past the batch)
    const bool active = g < P.B;
    if (!active) g = P.B - 1;
    const int64_t B = P.B;
    constexpr bool kBilinear = !(MODEL == EMG_TRANSE_L1 || MODEL == EMG_TRANSE_L2);   // replacement row's gradient = gi * q

    const int32_t s = P.pos[3 * g + 0], p = P.pos[3 * g + 1], o = P.pos[3 * g + 2];
    const float* srow = P.ent + (int64_t)s * P.ld_ent;
    const float* prow = P.rel + (int64_t)p * P.ld_rel;
    const float* orow = P.ent + (int64_t)o * P.ld_ent;
    // The corruption codes, in-place flags and factor positions of LPG negatives at a time in ONE gather each (lane j holds
    // negative c0 + j's), the first chunk issued together with the s, p, o rows and before anything is stored; lanes 0 / 1 of
    // my_flag_so hold the flags of the subject / object slot.  NOTHING is loaded inside the loop over the negatives except
    // their rows: a load in a branch there — round 2's fallback for eta + 2 > LPG was one, never taken — makes hipcc close
    // every join with s_waitcnt vmcnt(0), which also waits for the replacement rows just requested: the rolling window
    // below then holds ONE row in flight, not U (PMC, C3: 1572 cycles mean read latency, 1.7 MB in flight on the chip).
    const int first = (threadIdx.x & 63) / LPG * LPG;
    using IT = ip_traits<IP>;
    constexpr int NS = IT::n_state;
    static_assert(!IT::window_state || (W == 4 && LPG == 64 && FUSED), "IP 4 / 5 / 6: fused kernels of 16-byte rows, a wave per group");
    static_assert(!IT::lp_replay || (W == 4 && LPG == 64 && FUSED), "IP 7: fused kernels of 16-byte rows, a wave per group");
    int my_code = 0, my_flag = 0, my_pos = 0, my_flag_so = 0;   // my_pos: where the negative's factor goes (its slot's sorted position)
    int my_tag = 0;                                              // IP 6: the step the singleton's row was last written at
    auto gather = [&](int c0) {
        const int j = c0 + lg;
        my_code = j < P.eta ? P.codes[(int64_t)j * B + g] : 0;
        if (kBilinear && P.fac.coef) my_pos = j < P.eta ? (int)P.fac.pos_of_slot[(int64_t)j * B + g] : 0;
        if constexpr (IP != 0) my_flag = j < P.eta ? (int)P.single_ent[2 * B + (int64_t)j * B + g] : 0;
        if constexpr (IT::replay || IT::lp_replay) my_tag = (j < P.eta && my_flag) ? P.tag_ent[my_code & 0x7fffffff] : P.upto;
    };
    gather(0);
    float lr_lane = 0.f;   // IP 7: the learning rates of the last 64 steps (lane l: step upto - l), fetched with the group's first gather
    if constexpr (IT::lp_replay) lr_lane = P.lr_hist[max(P.upto - lg, 0)];
    if constexpr (IT::so_inplace) { if (lg < 2) my_flag_so = P.single_ent[(int64_t)lg * B + g]; }
    OptParams wopt = P.opt;   // (the window forms know their optimizer family: the update's switch folds away)
    if constexpr (NS == 2) wopt.opt = EMG_OPT_ADAM;
    R qo, qs, Ao, As;
    float pos_nrm = 0.f, pos_score = 0.f, gpos = 0.f, loss_acc = 0.f, lp_acc = 0.f;
    constexpr bool KEEP = keep_rows<MODEL, W, NV, FUSED, IP>::value;
    R ks, kp, ko;  // live across the loop only when KEEP (dead otherwise: no registers)
    // window forms: the state rows of a singleton subject / object row are fetched with the row at the group's START (form 6: and
    // the row replayed to the current step before the queries are built from it) and wait, with the row, in LDS for the update at
    // the group's end — nothing of it lives in registers across the loop over the negatives
    constexpr bool SOP = IT::window_state;
    float* stash = nullptr;
    int fso0 = 0, fso1 = 0;
    int tso0 = 0, tso1 = 0;   // IP 7: the steps a singleton subject / object row was last written at (P.upto: nothing to replay)
    if constexpr (SOP) {
        __shared__ float so_stash_mem[kThreads / 64][2 * (1 + NS)][R::N][64];
        stash = &so_stash_mem[threadIdx.x >> 6][0][0][0];
    }
    auto park = [&](int slot, const R& r) {
#pragma unroll
        for (int e = 0; e < R::N; ++e) stash[(slot * R::N + e) * 64 + lg] = r.x[e];
    };
    auto unpark = [&](int slot, R& r) {
#pragma unroll
        for (int e = 0; e < R::N; ++e) r.x[e] = stash[(slot * R::N + e) * 64 + lg];
    };
    {
        R rs, rp, ro;
        load_row<MODEL, W, NV, LPG>(rs, srow, lg, P.nchunks, P.khalf);
        load_row<MODEL, W, NV, LPG>(rp, prow, lg, P.nchunks, P.khalf);
        load_row<MODEL, W, NV, LPG>(ro, orow, lg, P.nchunks, P.khalf);
        if constexpr (SOP) {
            int my_tag_so = 0;
            if constexpr (IT::replay) { my_tag_so = P.upto; if (lg < 2) my_tag_so = P.tag_ent[lg == 0 ? s : o]; }
            fso0 = group_lane_value<LPG>(my_flag_so, first, 0); fso1 = group_lane_value<LPG>(my_flag_so, first, 1);
            R ms, vs, mo, vo;   // (the state rows of a singleton; any other row: its table row again — address selected, load unconditional)
            load_row<MODEL, W, NV, LPG, false>(ms, fso0 ? P.ent_state0 + (int64_t)s * P.ld_ent : srow, lg, P.nchunks, P.khalf);
            load_row<MODEL, W, NV, LPG, false>(mo, fso1 ? P.ent_state0 + (int64_t)o * P.ld_ent : orow, lg, P.nchunks, P.khalf);
            if constexpr (NS == 2) {
                load_row<MODEL, W, NV, LPG, false>(vs, fso0 ? P.ent_state1 + (int64_t)s * P.ld_ent : srow, lg, P.nchunks, P.khalf);
                load_row<MODEL, W, NV, LPG, false>(vo, fso1 ? P.ent_state1 + (int64_t)o * P.ld_ent : orow, lg, P.nchunks, P.khalf);
            }
            if constexpr (IT::replay) {
                const int ts = group_lane_value<LPG>(my_tag_so, first, 0), to = group_lane_value<LPG>(my_tag_so, first, 1);
                const float lrs = P.lr_hist[min(ts + 1 + lg, P.upto)], lro = P.lr_hist[min(to + 1 + lg, P.upto)];
                if (fso0 && ts > 0 && ts < P.upto) replay_in_window<MODEL, W, NV, LPG>(P, wopt, ts, lrs, rs, ms, vs, lg);
                if (fso1 && to > 0 && to < P.upto) replay_in_window<MODEL, W, NV, LPG>(P, wopt, to, lro, ro, mo, vo, lg);
            }
            park(0, rs); park(1, ms); park(1 + NS, ro); park(2 + NS, mo);
            if constexpr (NS == 2) { park(2, vs); park(5, vo); }
        }
        if constexpr (IT::lp_replay) {   // a lagging singleton subject / object row: the regulariser's missed steps BEFORE the queries are built from it
            int my_tag_so = P.upto;
            if (lg < 2) my_tag_so = P.tag_ent[lg == 0 ? s : o];
            fso0 = group_lane_value<LPG>(my_flag_so, first, 0); fso1 = group_lane_value<LPG>(my_flag_so, first, 1);
            tso0 = fso0 ? __builtin_amdgcn_readfirstlane(group_lane_value<LPG>(my_tag_so, first, 0)) : P.upto;
            tso1 = fso1 ? __builtin_amdgcn_readfirstlane(group_lane_value<LPG>(my_tag_so, first, 1)) : P.upto;
            float counted = 0.f;
            if (tso0 < P.upto) lp_replay_row<MODEL, W, NV, LPG>(P, tso0, lr_lane, rs, lg, counted);
            if (tso1 < P.upto) lp_replay_row<MODEL, W, NV, LPG>(P, tso1, lr_lane, ro, lg, counted);
            if (active) lp_acc += counted;   // (a group past the batch's end is a copy of the last one: its replayed steps are not counted twice)
        }
        make_queries<MODEL, W, NV>(rs, rp, ro, qo, qs);
        if constexpr (KEEP) { ks = rs; kp = rp; ko = ro; }
        if constexpr (FUSED || MODEL == EMG_TRANSE_L2) {
            if (MODEL == EMG_TRANSE_L2 && P.bw_scores_pos) {
                pos_nrm = -P.bw_scores_pos[g];
            } else {
                const float sum = group_sum<LPG>(partial_score<MODEL, W, NV>(rs, rp, ro));
                pos_score = finalize_score<MODEL>(sum, P.scale, 0);
                if constexpr (MODEL == EMG_TRANSE_L2) pos_nrm = sqrtf(sum);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < R::N; ++e) Ao.x[e] = As.x[e] = 0.f;
    PosTerms pos_terms{0.f, 0.f};
    if constexpr (FUSED) {
        if (P.scores_pos && active && lg == 0) P.scores_pos[g] = pos_score;
        pos_terms = local_loss_pos(P.fused_loss, pos_score);
    } else {
        gpos = P.g_pos[g];
    }

    // rows in flight per wave: U table rows — with their state rows (IP 4 / 5 / 6) U * (1 + NS) rows, held to ~48 registers
#ifndef EMG_WIN_BUDGET
#define EMG_WIN_BUDGET 48
#endif
    constexpr int U = !IT::window_state ? UW : (R::N * (1 + NS) * 4 <= EMG_WIN_BUDGET ? 4 : (R::N * (1 + NS) * 3 <= EMG_WIN_BUDGET ? 3 : 2));
    int chunk0 = 0, chunk1 = min(P.eta, LPG);   // the negatives [chunk0, chunk1) are the ones my_code / my_flag / my_pos describe
    auto code_of = [&](int j) -> int32_t { return group_lane_value<LPG>(my_code, first, j - chunk0); };
    auto flag_of = [&](int j) -> int {   // negative j of the current chunk
        if constexpr (IP == 0) return 0;
        return group_lane_value<LPG>(my_flag, first, j - chunk0);
    };
    auto flag_so = [&](int which) -> int {   // 0: the subject slot, 1: the object slot
        if constexpr (IP == 0) return 0;
        return group_lane_value<LPG>(my_flag_so, first, which);
    };
    // the replacement rows of negatives j0 .. j0+U-1.  Bilinear models take them RAW (lanes past the row's end hold a copy of
    // its last chunk): the query rows are zero there, so products, factored rows and in-place updates (guarded by the chunk
    // index) never see them — and no select sits between the load and its first use
    // IP 4 / 5 / 6: a slot of the window = the replacement row AND the optimizer state rows of a singleton (any other slot re-reads
    // its table row: the same lines, just requested — the loads are unconditional, only their address is selected) AND, for the
    // lagging form, the learning rates of the steps the row missed.  All of them asynchronous loads (inline assembly) with a
    // hand-written wait: between a slot's loads and its use lie the other slots' consumption — the replay LOOP, the in-place /
    // contribution branch — and hipcc's wait for compiler-visible loads across that control flow is vmcnt(0), which also waits
    // for the refill just issued (ISA of the first version: no load in flight while a slot is worked on).
    constexpr int PIECES_S = NV * (is_complex<MODEL>::value ? 2 : 1);              // 16-byte loads per row and lane
    constexpr int LPS = PIECES_S * (1 + NS) + (IT::replay ? 1 : 0);                 // vector-memory loads per slot
    static_assert(!IT::window_state || (U - 1) * LPS <= 63, "vmcnt is a 6-bit counter");
    R st0[IT::window_state ? U : 1], st1[NS == 2 ? U : 1];
    int32_t tg[IT::replay ? U : 1];
    float lrv[IT::replay ? U : 1];
    emg_f4 aw[IT::window_state ? U : 1][IT::window_state ? PIECES_S : 1], a0[IT::window_state ? U : 1][IT::window_state ? PIECES_S : 1],
           a1[NS == 2 ? U : 1][NS == 2 ? PIECES_S : 1];
    // TransE sums |q - e| over every lane, so lanes past the row's end must hold zeros.  The select sits at the row's USE, not at its
    // load: a select behind the load makes hipcc wait for the refill it has just issued (ISA of the first form: four
    // s_waitcnt vmcnt(0) at the end of every trip — the window went U -> 0 -> U, one exposed round trip per U negatives:
    // tools/sweep_small.py, C1: 0.4 - 0.55 us per negative whatever U)
    constexpr bool kZeroAtLoad = !kBilinear && EMG_BW_ROLL == 0;
    constexpr bool kZeroAtUse = !kBilinear && !kZeroAtLoad;
    auto zero_tail = [&](R& r) {
#pragma unroll
        for (int e = 0; e < R::N; ++e) r.x[e] = (lg + ((e % (W * NV)) / W) * LPG < P.nchunks) ? r.x[e] : 0.f;
    };
    auto issue_slot = [&](int u, int jn, int32_t repl) {
        if constexpr (IT::window_state) {
            const bool f = flag_of(jn) != 0;
            const float* wb = P.ent + (int64_t)repl * P.ld_ent;
            const float* b0 = f ? P.ent_state0 + (int64_t)repl * P.ld_ent : wb;
            const float* b1 = (NS == 2 && f) ? P.ent_state1 + (int64_t)repl * P.ld_ent : wb;
#pragma unroll
            for (int h = 0; h < (is_complex<MODEL>::value ? 2 : 1); ++h)
#pragma unroll
                for (int it = 0; it < NV; ++it) {
                    const int off = h * P.khalf + 4 * min(lg + it * LPG, P.nchunks - 1);   // (past the row's end: its last chunk again)
                    aw[u][h * NV + it] = vm_load16_async(wb + off);
                    a0[u][h * NV + it] = vm_load16_async(b0 + off);
                    if constexpr (NS == 2) a1[u][h * NV + it] = vm_load16_async(b1 + off);
                }
            if constexpr (IT::replay) {
                tg[u] = group_lane_value<LPG>(my_tag, first, jn - chunk0);
                lrv[u] = vm_load4_async(P.lr_hist + min(tg[u] + 1 + lg, P.upto));
            }
        }
    };
    auto unpack = [&](const emg_f4 (&src)[IT::window_state ? PIECES_S : 1], R& r, bool zero_tail) {
#pragma unroll
        for (int q = 0; q < PIECES_S; ++q) {
            const bool on = !zero_tail || lg + (q % NV) * LPG < P.nchunks;
            r.x[4 * q + 0] = on ? src[q].x : 0.f; r.x[4 * q + 1] = on ? src[q].y : 0.f;
            r.x[4 * q + 2] = on ? src[q].z : 0.f; r.x[4 * q + 3] = on ? src[q].w : 0.f;
        }
    };
    auto take_slot = [&](int u, R& w_row) {   // wait for THIS slot's loads: at most the (U - 1) younger slots' may still be in flight
        if constexpr (IT::window_state) {
            vm_wait_only<(U - 1) * LPS>();
#pragma unroll
            for (int q = 0; q < PIECES_S; ++q) { vm_landed(aw[u][q]); vm_landed(a0[u][q]); if constexpr (NS == 2) vm_landed(a1[u][q]); }
            if constexpr (IT::replay) vm_landed(lrv[u]);
            unpack(aw[u], w_row, !kBilinear);   // (TransE: lanes past the row's end must hold zeros; bilinear models meet zero query rows there)
            unpack(a0[u], st0[u], false);
            if constexpr (NS == 2) unpack(a1[u], st1[u], false);
        }
    };
    auto fetch = [&](int j0, int32_t (&code)[U], R (&re)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) code[u] = code_of(min(j0 + u, chunk1 - 1));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int32_t repl = code[u] & 0x7fffffff;
            if constexpr (IT::window_state) issue_slot(u, min(j0 + u, chunk1 - 1), repl);
            else load_row<MODEL, W, NV, LPG, kZeroAtLoad>(re[u], P.ent + (int64_t)repl * P.ld_ent, lg, P.nchunks, P.khalf);
        }
    };
    // ROLLING window of U replacement rows: as soon as a negative's row has been consumed (score, gradient, in-place
    // update or contribution), ITS registers take the load of the negative U places later — the wave keeps U rows in flight
    // through the whole loop instead of U -> 0 -> U per trip, with no register beyond the U rows (at 3 waves per SIMD it is
    // the bytes in flight per wave that bound this kernel: tools/hbm_ceiling's bare mix runs 8 waves deep)
    int32_t code[U];
    float gj[U];
    R re[U];
    constexpr int PIECES = NV * (is_complex<MODEL>::value ? 2 : 1);   // 16-byte loads per row and lane
    constexpr bool kAsync = EMG_BW_ASYNC != 0 && EMG_BW_ROLL != 0 && W == 4 && FUSED && (PIECES == 1 || PIECES == 2 || PIECES == 4) &&
                            (U - 1) * PIECES <= 15 && !IT::window_state;
    static_assert(!IT::window_state || EMG_BW_ROLL != 0, "IP 4 / 5 / 6 use the rolling window");
    emg_f4 pa[kAsync ? U : 1][kAsync ? PIECES : 1];
    auto issue_row = [&](emg_f4 (&dst)[kAsync ? PIECES : 1], int32_t repl) {   // all lanes load: past the row's end, its last chunk again
        const float* base = P.ent + (int64_t)repl * P.ld_ent;
#pragma unroll
        for (int h = 0; h < (is_complex<MODEL>::value ? 2 : 1); ++h)
#pragma unroll
            for (int it = 0; it < NV; ++it) {
                const int c = min(lg + it * LPG, P.nchunks - 1);
                dst[h * NV + it] = vm_load16_async(base + h * P.khalf + 4 * c);
            }
    };
    // (the body indexes src[1 .. 3] under `if constexpr` on PIECES; where kAsync is off the array has one entry and the lambda is
    // never called, but it is still instantiated: the diagnostic is silenced HERE, not for the translation unit)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Warray-bounds"
    auto take_row = [&](emg_f4 (&src)[kAsync ? PIECES : 1], R& r) {   // wait for THIS row (the U - 1 younger ones may fly on)
        if constexpr (PIECES == 1) vm_wait<(U - 1) * PIECES>(src[0]);
        else if constexpr (PIECES == 2) vm_wait<(U - 1) * PIECES>(src[0], src[1]);
        else vm_wait<(U - 1) * PIECES>(src[0], src[1], src[2], src[3]);
#pragma unroll
        for (int q = 0; q < PIECES; ++q) { r.x[4 * q + 0] = src[q].x; r.x[4 * q + 1] = src[q].y; r.x[4 * q + 2] = src[q].z; r.x[4 * q + 3] = src[q].w; }
    };
#pragma clang diagnostic pop
    for (; chunk0 < P.eta; chunk0 += LPG) {   // (one trip unless eta > LPG)
    chunk1 = min(P.eta, chunk0 + LPG);
    if (chunk0 > 0) gather(chunk0);
    if constexpr (kAsync) {
#pragma unroll
        for (int u = 0; u < U; ++u) { code[u] = code_of(min(chunk0 + u, chunk1 - 1)); issue_row(pa[u], code[u] & 0x7fffffff); }
    } else if constexpr (EMG_BW_ROLL != 0) {
        fetch(chunk0, code, re);
        if constexpr (!FUSED) {
#pragma unroll
            for (int u = 0; u < U; ++u) gj[u] = P.g_neg[(int64_t)min(chunk0 + u, chunk1 - 1) * B + g];
        }
    }
    for (int j0 = chunk0; j0 < chunk1; j0 += U) {
        if constexpr (EMG_BW_ROLL == 0) {
            fetch(j0, code, re);
            if constexpr (!FUSED) {
#pragma unroll
                for (int u = 0; u < U; ++u) gj[u] = P.g_neg[(int64_t)min(j0 + u, chunk1 - 1) * B + g];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u;
            if (j >= chunk1) break;
            const bool keep_s = code[u] < 0;  // subject kept => the OBJECT was replaced
            const int32_t repl = code[u] & 0x7fffffff;
            if constexpr (kAsync) take_row(pa[u], re[u]);
            if constexpr (IT::window_state) take_slot(u, re[u]);
            if constexpr (kZeroAtUse && !IT::window_state) zero_tail(re[u]);
            if constexpr (IT::replay) {   // a singleton behind the table's step: replay the steps it missed, THEN score it
                if (flag_of(j) && tg[u] > 0 && tg[u] < P.upto) replay_in_window<MODEL, W, NV, LPG>(P, wopt, tg[u], lrv[u], re[u], st0[u], st1[u], lg);
            }
            if constexpr (IT::lp_replay) {   // (the tag is the group's — a wave per group — so the step loop and its learning rates are scalar)
                const int from = __builtin_amdgcn_readfirstlane(group_lane_value<LPG>(my_tag, first, j - chunk0));
                float counted = 0.f;
                if (from < P.upto) lp_replay_row<MODEL, W, NV, LPG>(P, from, lr_lane, re[u], lg, counted);
                if (active) lp_acc += counted;
            }
            float nrm = 0.f;
            if constexpr (FUSED || MODEL == EMG_TRANSE_L2) {
                if (MODEL == EMG_TRANSE_L2 && P.bw_scores_neg) {
                    nrm = -P.bw_scores_neg[(int64_t)j * B + g];
                } else {
                    float part;
                    if (keep_s) part = neg_partial<MODEL, W, NV>(qo, re[u]);
                    else part = neg_partial<MODEL, W, NV>(qs, re[u]);
                    const float sum = group_sum<LPG>(part);
                    if constexpr (MODEL == EMG_TRANSE_L2) nrm = sqrtf(sum);
                    if constexpr (FUSED) {
                        const float neg = finalize_score<MODEL>(sum, P.scale, 0);
                        gj[u] = local_loss_neg(P.fused_loss, pos_score, pos_terms, neg, P.margin, loss_acc, gpos);
                        if (P.scores_neg && active && lg == 0) P.scores_neg[(int64_t)j * B + g] = neg;
                    }
                }
            }
            const float gi = inner_coef<MODEL>(gj[u], nrm, P.scale);
            R row;
            if (keep_s) neg_grads<MODEL, W, NV, true>(qo, re[u], gi, row, Ao);
            else neg_grads<MODEL, W, NV, false>(qs, re[u], gi, row, As);
            if (active) {
                const int64_t slot = 2 * B + (int64_t)j * B + g;
                if (IP != 0 && flag_of(j)) {
                    if constexpr (IT::window_state) inplace_update_regs<MODEL, W, NV, LPG, NS>(P, wopt, repl, re[u], row, st0[u], st1[NS == 2 ? u : 0], lg);
                    else inplace_update<MODEL, W, NV, LPG, IT::chunkwise>(P, repl, re[u], row, lg, lp_acc);
                }
                else if (kBilinear && P.fac.coef) {   // row = gi * q: q is stored once, below; gi goes where the apply reads it
                    const int at = group_lane_value<LPG>(my_pos, first, j - chunk0);
                    if (lg == 0) P.fac.coef[at] = gi;
                }
                else store_row<MODEL, W, NV, LPG>(row, P.contrib_ent + slot * P.ldc, lg, P.nchunks, P.khalf);
            }
            if constexpr (EMG_BW_ROLL != 0) {   // refill this row's registers with the negative U places later
                // UNCONDITIONALLY (past the end: the chunk's last row again, a cache hit): a load under a condition leaves a
                // path without it, and the wait for an older row then has to be vmcnt(0) — with every refill on every path
                // hipcc counts them (vmcnt(2 (U - 1))) and U rows really are in flight
                const int jn = min(j + U, chunk1 - 1);
                code[u] = code_of(jn);
                if constexpr (kAsync) issue_row(pa[u], code[u] & 0x7fffffff);
                else if constexpr (IT::window_state) issue_slot(u, jn, code[u] & 0x7fffffff);
                else load_row<MODEL, W, NV, LPG, kZeroAtLoad>(re[u], P.ent + (int64_t)(code[u] & 0x7fffffff) * P.ld_ent, lg, P.nchunks, P.khalf);
                if constexpr (!FUSED) gj[u] = P.g_neg[(int64_t)jn * B + g];
            }
        }
    }
    if constexpr (kAsync) {
        // the refills issued past the chunk's end are never taken: their registers must stay theirs until the loads have landed
        // (hipcc does not know they are being written) — one wait for everything, with every row as its operand.  At the end of
        // EVERY chunk trip (eta > LPG): on the back edge the next fetch() redefines these registers with pure outputs, so the
        // compiler may hand them to gather()'s address arithmetic while the loads are still landing (round-4 advisor finding)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (PIECES == 1) vm_wait<0>(pa[u][0]);
            else if constexpr (PIECES == 2) vm_wait<0>(pa[u][0], pa[u][1]);
            else vm_wait<0>(pa[u][0], pa[u][1], pa[u][2], pa[u][3]);
        }
    }
    if constexpr (IT::window_state) {   // (the refills issued past the end are never taken: their registers stay theirs until the loads have landed)
        vm_wait_only<0>();
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int q = 0; q < PIECES_S; ++q) { vm_landed(aw[u][q]); vm_landed(a0[u][q]); if constexpr (NS == 2) vm_landed(a1[u][q]); }
            if constexpr (IT::replay) vm_landed(lrv[u]);
        }
    }
    }
    if (kBilinear && P.fac.coef && active) {   // the two query rows every factored negative of this group points at
        store_row<MODEL, W, NV, LPG>(qo, P.contrib_ent + (2 * B + g) * P.ldc, lg, P.nchunks, P.khalf);
        store_row<MODEL, W, NV, LPG>(qs, P.contrib_ent + (3 * B + g) * P.ldc, lg, P.nchunks, P.khalf);
    }
    // The epilogue's row addresses are formed from the ids HERE.  Window forms: from an OPAQUE copy of the ids — left to hipcc the
    // addresses are computed in the prologue and live across the whole loop over the negatives as 64-bit pairs, registers the cap
    // of three waves per SIMD does not have (round 4: 12 .. 84 bytes of scratch per lane in every window-form kernel)
    int32_t es = s, ep = p, eo = o;
    int64_t eg = g;
    if constexpr (IT::window_state) asm volatile("" : "+v"(es), "+v"(ep), "+v"(eo), "+v"(eg));
    const float* srow_e = P.ent + (int64_t)es * P.ld_ent;
    const float* prow_e = P.rel + (int64_t)ep * P.ld_rel;
    const float* orow_e = P.ent + (int64_t)eo * P.ld_ent;
    if (active) {
        // kept rows: form their gradients from the accumulators
        R rs, rp, ro, gs, gp, go;
        if constexpr (KEEP) {
            rs = ks; rp = kp; ro = ko;
        } else {  // re-load s, p, o (read a moment ago) instead of holding three more rows per group
            load_row<MODEL, W, NV, LPG>(rs, srow_e, lg, P.nchunks, P.khalf);
            load_row<MODEL, W, NV, LPG>(rp, prow_e, lg, P.nchunks, P.khalf);
            load_row<MODEL, W, NV, LPG>(ro, orow_e, lg, P.nchunks, P.khalf);
        }
        if constexpr (IT::replay) {   // (a singleton's table row still lags: its replayed row is the parked one)
            if (fso0) unpark(0, rs);
            if (fso1) unpark(1 + NS, ro);
        }
        if constexpr (IT::lp_replay && !KEEP) {   // (re-read rows still lag: the same replay again — ~130 instructions a row —, its loss terms counted once, above)
            float uncounted = 0.f;
            if (tso0 < P.upto) lp_replay_row<MODEL, W, NV, LPG>(P, tso0, lr_lane, rs, lg, uncounted);
            if (tso1 < P.upto) lp_replay_row<MODEL, W, NV, LPG>(P, tso1, lr_lane, ro, lg, uncounted);
        }
        finish_grads<MODEL, W, NV>(rs, rp, ro, Ao, As, inner_coef<MODEL>(gpos, pos_nrm, P.scale), gs, gp, go);
        store_row<MODEL, W, NV, LPG>(gp, P.contrib_rel + eg * P.ldc, lg, P.nchunks, P.khalf);
        if constexpr (SOP) {
            if (fso0) { R ms, vs; unpark(1, ms); if constexpr (NS == 2) unpark(2, vs); inplace_update_regs<MODEL, W, NV, LPG, NS>(P, wopt, es, rs, gs, ms, vs, lg); }
            else store_row<MODEL, W, NV, LPG>(gs, P.contrib_ent + eg * P.ldc, lg, P.nchunks, P.khalf);
            if (fso1) { R mo, vo; unpark(2 + NS, mo); if constexpr (NS == 2) unpark(5, vo); inplace_update_regs<MODEL, W, NV, LPG, NS>(P, wopt, eo, ro, go, mo, vo, lg); }
            else store_row<MODEL, W, NV, LPG>(go, P.contrib_ent + (B + eg) * P.ldc, lg, P.nchunks, P.khalf);
        } else {
            if (IT::so_inplace && flag_so(0)) inplace_update<MODEL, W, NV, LPG, IT::chunkwise>(P, es, rs, gs, lg, lp_acc);
            else store_row<MODEL, W, NV, LPG>(gs, P.contrib_ent + eg * P.ldc, lg, P.nchunks, P.khalf);
            if (IT::so_inplace && flag_so(1)) inplace_update<MODEL, W, NV, LPG, IT::chunkwise>(P, eo, ro, go, lg, lp_acc);
            else store_row<MODEL, W, NV, LPG>(go, P.contrib_ent + (B + eg) * P.ldc, lg, P.nchunks, P.khalf);
        }
    }
    // the regulariser's value over the rows updated (and replayed) in place: one double atomic per WORKGROUP (per wave — 16 k
    // adds to one address per launch, one retiring per ~10 ns — they were 58 us of C3 + LP's 311 us scoring kernel)
    if constexpr (IP == 3 || IP == 7) block_add_double(P.lp_accum, lp_acc);
    if constexpr (FUSED) {
        // loss: one value per group (lane 0), block-reduced in double, one atomic per block
        double v = (active && lg == 0) ? (double)loss_acc : 0.0;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        __shared__ double part[kThreads / 64];
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < kThreads / 64; ++w) t += part[w];
            if (t != 0.0) atomicAdd(P.loss_accum + (bx & P.loss_mask), t);   // (slots: emg_backward_args.loss_slots)
        }
    }
}

template <int MODEL, int W, int NV, int LPG, bool FUSED, int IP>
__global__ __launch_bounds__(kThreads, EMG_BW_MINWAVES) void train_backward_kernel(const GroupParams P) {
    train_backward_body<MODEL, W, NV, LPG, FUSED, IP>(P, blockIdx.x);
}
// the same with RIDERS: the first workgroups of the launch do the table-independent preparation of the next batches
// (emg_plan.hip); instantiated for the fused 16-byte-row forms only (compile time)
#ifdef EMG_TRACE   // timing aid (tools/trace_waves.py fused): wall-clock stamps (10 ns) of every wave of the last fused launch
static __device__ unsigned long long emg_trace_fused_buf[4 * 65536];
#endif
#ifndef EMG_IP6_MINWAVES
#define EMG_IP6_MINWAVES 3   // window forms: three waves per SIMD (form 6, ComplEx k = 200: 168 VGPRs + 64 bytes of scratch; left alone 186 VGPRs, two waves:
#endif                       // C3 + Adam 0.86 against 0.93 ms per step)
// UW: replacement rows in flight per wave (EMG_BW_U).  A/B aid (EMG_DEEP_B): a SMALL batch (fewer waves than the SIMDs can hold at
// once: the reference's own configurations, 1.7 - 4.7 waves per SIMD) has registers to spare, and a deeper window (EMG_BW_U_DEEP)
// was expected to shorten a wave's chain of round trips.  Measured: C1 / C2 +-0, C5 slower — those kernels are one wave's
// instruction stream, not its row loads.  Same bits: the negatives are consumed in the same order.
#ifndef EMG_BW_U_DEEP
#define EMG_BW_U_DEEP 10
#endif
template <int MODEL, int W, int NV, int LPG, int IP, int UW = EMG_BW_U>
__global__ __launch_bounds__(kThreads, (ip_traits<IP>::window_state ? EMG_IP6_MINWAVES : EMG_BW_MINWAVES)) void train_fused_riders_kernel(const GroupParams P, const Riders riders) {
    unsigned bx;
    if (run_riders(riders, &bx)) return;
#ifdef EMG_TRACE
    const unsigned tw = (bx * kThreads + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0 && tw < 65536) emg_trace_fused_buf[4 * tw] = wall_clock64();
#endif
    train_backward_body<MODEL, W, NV, LPG, true, IP, UW>(P, bx);
#ifdef EMG_TRACE
    if ((threadIdx.x & 63) == 0 && tw < 65536) emg_trace_fused_buf[4 * tw + 1] = wall_clock64();
#endif
}

// destination ids of the contribution rows of a batch (depends only on the batch ids and codes)
static __global__ void build_dest_kernel(const int32_t* __restrict__ pos, int64_t B, int eta, const int32_t* __restrict__ codes,
                                  int32_t* __restrict__ dest_ent, int32_t* __restrict__ dest_rel) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < B) {
        dest_ent[t] = pos[3 * t + 0];
        dest_ent[B + t] = pos[3 * t + 2];
        dest_rel[t] = pos[3 * t + 1];
    }
    if (t < (int64_t)eta * B) dest_ent[2 * B + t] = codes[t] & 0x7fffffff;
}

// ---------------------------------------------------------------------------------------------
// generic forward fallback (any k, any alignment): one wave per group, strided scalar loads
// ---------------------------------------------------------------------------------------------
template <int MODEL>
__device__ __forceinline__ float strided_partial(const float* __restrict__ a, const float* __restrict__ p,
                                                 const float* __restrict__ b, int khalf, int n, int lane) {
    float acc = 0.f;
    for (int c = lane; c < n; c += 64) {
        if constexpr (MODEL == EMG_TRANSE_L1) {
            acc += fabsf((a[c] + p[c]) - b[c]);
        } else if constexpr (MODEL == EMG_TRANSE_L2) {
            const float d = (a[c] + p[c]) - b[c];
            acc = fmaf(d, d, acc);
        } else if constexpr (MODEL == EMG_DISTMULT) {
            acc = fmaf(a[c] * p[c], b[c], acc);
        } else {
            const float sr = a[c], si = a[khalf + c], pr = p[c], pi = p[khalf + c], orr = b[c], oi = b[khalf + c];
            acc = fmaf(pr * sr, orr, acc);
            acc = fmaf(pr * si, oi, acc);
            acc = fmaf(pi * sr, oi, acc);
            acc = fmaf(-(pi * si), orr, acc);
        }
    }
    return acc;
}

template <int MODEL>
__global__ __launch_bounds__(kThreads) void train_forward_generic_kernel(const GroupParams P) {
    const int lane = threadIdx.x & 63;
    int64_t g = ((int64_t)blockIdx.x * kThreads + threadIdx.x) / 64;
    const bool active = g < P.B;
    if (!active) g = P.B - 1;
    const int n = is_complex<MODEL>::value ? P.khalf : P.k_int;
    const float* rs = P.ent + (int64_t)P.pos[3 * g + 0] * P.ld_ent;
    const float* rp = P.rel + (int64_t)P.pos[3 * g + 1] * P.ld_rel;
    const float* ro = P.ent + (int64_t)P.pos[3 * g + 2] * P.ld_ent;
    float sum = group_sum<64>(strided_partial<MODEL>(rs, rp, ro, P.khalf, n, lane));
    if (active && lane == 0) P.scores_pos[g] = finalize_score<MODEL>(sum, P.scale, P.flags);
    for (int j = 0; j < P.eta; ++j) {
        const int32_t code = P.codes[(int64_t)j * P.B + g];
        const float* re = P.ent + (int64_t)(code & 0x7fffffff) * P.ld_ent;
        const bool keep_s = code < 0;
        sum = group_sum<64>(strided_partial<MODEL>(keep_s ? rs : re, rp, keep_s ? re : ro, P.khalf, n, lane));
        if (active && lane == 0) P.scores_neg[(int64_t)j * P.B + g] = finalize_score<MODEL>(sum, P.scale, P.flags);
    }
}


// the fused forms of one model (emg_fused_m<model>.hip): shape 0..3 = 16 / 32 / 64 lanes per group with one
// 16-byte chunk per lane, 64 lanes with two; ip 0 = no in-place updates, 1 = SGD in place, 2 = any optimizer in place
typedef void (*fused_launch_fn)(int shape, int ip, unsigned grid, hipStream_t st, const GroupParams& P, const Riders& riders);
void launch_fused_m0(int, int, unsigned, hipStream_t, const GroupParams&, const Riders&);
void launch_fused_m1(int, int, unsigned, hipStream_t, const GroupParams&, const Riders&);
void launch_fused_m2(int, int, unsigned, hipStream_t, const GroupParams&, const Riders&);
void launch_fused_m3(int, int, unsigned, hipStream_t, const GroupParams&, const Riders&);
void launch_fused_m4(int, int, unsigned, hipStream_t, const GroupParams&, const Riders&);

}  // namespace emg
