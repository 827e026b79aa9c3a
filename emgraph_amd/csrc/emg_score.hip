// emg_score.hip — fused embedding gather + score (K1+K2+K4), its adjoint (K7) and the fused train kernel.
//
// Replaces, per training batch, EmbeddingModel._lookup_embeddings (EmbeddingModel.py:490-533, three
// materialised tf.nn.embedding_lookup gathers), Model._fn (TransE.py:208-216, DistMult.py:201,
// ComplEx.py:288-298, HolE.py:189), the positive tiling (EmbeddingModel.py:724-729), Loss.apply for the
// pair-local losses, and the TF-autodiff backward of all of them.
//
// Mapping to CDNA4: one LANE GROUP (16/32/64 lanes of a wave64) owns one positive triple.  The
// group keeps the s, p, o rows in VGPRs (16-byte global_load_dwordx4 per lane, rows are 16-byte
// aligned), scores the positive, then streams the eta replacement rows of that positive — the
// relation row and the kept side are read ONCE per group instead of once per negative:
// (3+eta) row reads per group instead of 3(1+eta).  k-reductions are __shfl_xor butterflies.
//
// Backward/fused kernel: for the pair-local losses (pairwise, nll, absolute_margin) dL/dneg_j depends only
// on (pos_i, neg_j), so score -> loss -> gradient happens in ONE pass over the rows.  A gradient row whose
// destination is hit exactly once in the batch (the common case for uniform negatives) is applied to the
// table IN PLACE from the registers that already hold the row: 1 read + 1 write instead of
// read + contribution write + contribution read + RMW.  Race-free: a singleton destination is, by
// definition, read by no other group of this batch.  All other rows go to the contribution buffer and
// are summed in a fixed order by emg_apply_grouped (no float atomics anywhere).
// HBM-bound by design: algorithmic bytes per group in DESIGN.md §4.
#include "emg_score_kernels.hpp"

namespace emg {

// ---------------------------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------------------------
enum class Pass { Forward, Backward, Fused };

template <int MODEL, int W, int NV, int LPG>
static void launch_group(Pass pass, const GroupParams& P, hipStream_t st, const Riders& riders) {
    const int groups_per_block = kThreads / LPG;
    const unsigned grid = (unsigned)cdiv(P.B, groups_per_block) + riders.total;   // (riders.total != 0 only where can_ride())
    if (pass == Pass::Forward)
        hipLaunchKernelGGL((train_forward_kernel<MODEL, W, NV, LPG>), dim3(grid), dim3(kThreads), 0, st, P);
    else {
        int ip = !P.single_ent ? 0 : (P.opt.opt == EMG_OPT_SGD ? (P.opt.lp_lambda != 0.f ? 3 : 1) : 2);
        const bool fused = pass == Pass::Fused;
        // window (emg_backward_args.inplace_window): the window forms — the state rows travel with the table rows (ip 4 / 5; 6: Adam
        // whose dense pass is deferred, emg_backward_args.lr_hist); run_group_pass has checked the shape
        if (ip == 2 && P.window) ip = P.lr_hist ? 6 : ((P.opt.opt == EMG_OPT_ADAM || P.opt.opt == EMG_OPT_ADAM_LAZY) ? 5 : 4);
        if (ip == 3 && P.lr_hist) ip = 7;   // SGD + LP under the deferred dense pass: lagging singleton negatives replayed in registers
#define EMG_BW(F, I) hipLaunchKernelGGL((train_backward_kernel<MODEL, W, NV, LPG, F, I>), dim3(grid), dim3(kThreads), 0, st, P)
        if constexpr (W == 4) {
            if (fused) {   // the fused forms, with or without riders: one translation unit per model
                static const fused_launch_fn by_model[5] = {launch_fused_m0, launch_fused_m1, launch_fused_m2, launch_fused_m3, launch_fused_m4};
                const int shape = NV == 2 ? 3 : (LPG == 16 ? 0 : (LPG == 32 ? 1 : 2));
                by_model[MODEL](shape, ip, grid, st, P, riders);
                return;
            }
            if (ip == 0) EMG_BW(false, 0); else if (ip == 1) EMG_BW(false, 1); else if (ip == 2) EMG_BW(false, 2); else EMG_BW(false, 3);
        } else {   // scalar rows (k not a multiple of 4)
            // (ip == 3 — in-place SGD folding the LP regulariser — exists for 16-byte rows only: run_group_pass refuses it here)
            if (fused) { if (ip == 0) EMG_BW(true, 0); else if (ip == 1) EMG_BW(true, 1); else EMG_BW(true, 2); }
            else { if (ip == 0) EMG_BW(false, 0); else if (ip == 1) EMG_BW(false, 1); else EMG_BW(false, 2); }
        }
#undef EMG_BW
    }
}

// returns false when no register-tiled variant fits
template <int MODEL>
static bool dispatch_model(Pass pass, GroupParams& P, bool vec, hipStream_t st, const Riders& riders) {
    const int n = P.width > 0 ? P.width : (is_complex<MODEL>::value ? P.khalf : P.k_int);
    if (vec) {
        P.nchunks = n / 4;
        const int c = P.nchunks;
        // Narrow rows share a wave (4 / 2 groups of 16 / 32 lanes) — unless the batch is so small that the launch would not
        // even put two waves on every SIMD: then a wave per group (idle lanes cost nothing when every wave is waiting for
        // memory; two groups in lock-step execute BOTH sides of every in-place / contribution branch).  Same bits: the
        // lane reduction's extra levels add zeros.
        static const int wide_env = getenv("EMG_WIDE_GROUPS") ? atoi(getenv("EMG_WIDE_GROUPS")) : -1;   // A/B aid
        // (in-place updates of a stateful optimizer: always a wave per group — the form whose state rows travel with the table rows)
        const bool stateful_ip = pass == Pass::Fused && P.single_ent && (P.window || P.lr_hist);   // (ip 7 too: a wave per group)
        // (stateful_ip wins over the A/B switch: forms 4 / 5 / 6 exist for LPG = 64 only — with EMG_WIDE_GROUPS=0 a narrow row would
        // otherwise reach a shape that launches nothing)
        const bool wide = pass != Pass::Forward && (stateful_ip || (wide_env >= 0 ? wide_env != 0 : P.B <= 2048));
        if (c <= 16 && !wide) launch_group<MODEL, 4, 1, 16>(pass, P, st, riders);
        else if (c <= 32 && !wide) launch_group<MODEL, 4, 1, 32>(pass, P, st, riders);
        else if (c <= 64) launch_group<MODEL, 4, 1, 64>(pass, P, st, riders);
        else if (c <= 128) launch_group<MODEL, 4, 2, 64>(pass, P, st, riders);
        else return false;
    } else {
        P.nchunks = n;
        const int c = P.nchunks;
        if (c <= 64) launch_group<MODEL, 1, 1, 64>(pass, P, st, riders);
        else if (c <= 128) launch_group<MODEL, 1, 2, 64>(pass, P, st, riders);
        else if (c <= 256) launch_group<MODEL, 1, 4, 64>(pass, P, st, riders);
        else if (c <= 512) launch_group<MODEL, 1, 8, 64>(pass, P, st, riders);
        else return false;
    }
    return true;
}

static int run_group_pass(Pass pass, int model, GroupParams& P, hipStream_t st, const Riders* riders_in = nullptr) {
    static const Riders no_riders{};
    const Riders* riders_p = riders_in && riders_in->total ? riders_in : nullptr;
    const bool cplx = (model == EMG_COMPLEX || model == EMG_HOLE);
    EMG_REQUIRE(model >= 0 && model <= EMG_HOLE, "unknown model id %d", model);
    EMG_REQUIRE(P.k_int > 0 && (!cplx || P.k_int % 2 == 0), "bad k_int %d for model %d", P.k_int, model);
    EMG_REQUIRE(P.ld_ent >= P.k_int && P.ld_rel >= P.k_int, "row stride smaller than k_int");
    EMG_REQUIRE(P.B >= 0 && P.eta >= 0, "negative sizes");
    if (P.B == 0) return EMG_OK;
    EMG_REQUIRE(P.B * (int64_t)kThreads < ((int64_t)1 << 37), "batch too large");
    P.khalf = cplx ? P.k_int / 2 : 0;
    const int n = cplx ? P.khalf : P.k_int;
    bool vec = (n % 4 == 0) && (P.ld_ent % 4 == 0) && (P.ld_rel % 4 == 0) && aligned16(P.ent) && aligned16(P.rel);
    if (pass != Pass::Forward) {
        vec = vec && (P.ldc % 4 == 0) && aligned16(P.contrib_ent) && aligned16(P.contrib_rel);
        if (P.single_ent)
            vec = vec && (!P.ent_state0 || aligned16(P.ent_state0)) && (!P.ent_state1 || aligned16(P.ent_state1));
    }
    if (pass != Pass::Forward && P.single_ent && P.opt.opt == EMG_OPT_SGD && P.opt.lp_lambda != 0.f && !vec)
        return fail(EMG_ENOSUP, "train backward: in-place updates fold an LP regulariser for 16-byte aligned rows only (k, or k per "
                                "half for complex models, a multiple of 4); pass single_ent = NULL");
    if (P.window && !(vec && pass == Pass::Fused && n / 4 <= 64 && P.single_ent && P.opt.opt != EMG_OPT_SGD))
        return fail(EMG_ENOSUP, "train backward: inplace_window (a stateful optimizer's state rows travelling with the table rows) needs the "
                                "fused kernel on 16-byte aligned rows of at most 64 chunks (per half for complex models)");
    if (riders_p) {   // only the fused 16-byte-row kernels carry riders; everything else: the stages alone, first
        const bool can_ride = pass == Pass::Fused && vec && n <= 512;   // (= the train_fused_riders_kernel forms)
        if (!can_ride) {
            int rc = launch_riders_alone(*riders_p, st);
            if (rc != EMG_OK) return rc;
            riders_p = nullptr;
        }
    }
    const Riders& riders = riders_p ? *riders_p : no_riders;
    bool ok = false;
    switch (model) {
        case EMG_TRANSE_L1: ok = dispatch_model<EMG_TRANSE_L1>(pass, P, vec, st, riders); break;
        case EMG_TRANSE_L2: ok = dispatch_model<EMG_TRANSE_L2>(pass, P, vec, st, riders); break;
        case EMG_DISTMULT: ok = dispatch_model<EMG_DISTMULT>(pass, P, vec, st, riders); break;
        case EMG_COMPLEX: ok = dispatch_model<EMG_COMPLEX>(pass, P, vec, st, riders); break;
        case EMG_HOLE: ok = dispatch_model<EMG_HOLE>(pass, P, vec, st, riders); break;
    }
    if (!ok && pass == Pass::Backward) {
        // WIDE rows (more than 512 columns per half): given dL/dscore every gradient is separable by column, so the
        // register-tiled kernel runs once per block of 512 columns on offset pointers (the k-sharded multi-GPU step
        // does the same across ranks).  TransE-L2's gradient needs the FULL norm: the caller passes the final scores.
        EMG_REQUIRE(model != EMG_TRANSE_L2 || (P.bw_scores_pos && (P.eta == 0 || P.bw_scores_neg)),
                    "train backward: TransE-L2 rows wider than 512 columns need bw_scores_pos / bw_scores_neg (the full norms)");
        constexpr int kBlock = 512;
        for (int c0 = 0; c0 < n; c0 += kBlock) {
            GroupParams Q = P;
            Q.width = n - c0 < kBlock ? n - c0 : kBlock;
            Q.ent += c0; Q.rel += c0; Q.contrib_ent += c0; Q.contrib_rel += c0;
            if (Q.ent_rw) Q.ent_rw += c0;
            if (Q.ent_state0) Q.ent_state0 += c0;
            if (Q.ent_state1) Q.ent_state1 += c0;
            bool ok2 = false;
            switch (model) {
                case EMG_TRANSE_L1: ok2 = dispatch_model<EMG_TRANSE_L1>(pass, Q, vec, st, no_riders); break;
                case EMG_TRANSE_L2: ok2 = dispatch_model<EMG_TRANSE_L2>(pass, Q, vec, st, no_riders); break;
                case EMG_DISTMULT: ok2 = dispatch_model<EMG_DISTMULT>(pass, Q, vec, st, no_riders); break;
                case EMG_COMPLEX: ok2 = dispatch_model<EMG_COMPLEX>(pass, Q, vec, st, no_riders); break;
                case EMG_HOLE: ok2 = dispatch_model<EMG_HOLE>(pass, Q, vec, st, no_riders); break;
            }
            if (!ok2) return fail(EMG_ENOSUP, "train backward: column block of %d does not fit", Q.width);
            EMG_LAUNCH_CHECK();
        }
        return EMG_OK;
    }
    if (!ok) {
        if (pass != Pass::Forward)
            return fail(EMG_ENOSUP, "fused score+loss+gradient: rows of k_int=%d are wider than the register-tiled kernel holds "
                                    "(512 columns per half) — use emg_train_forward + emg_loss + emg_train_backward_ex(fused_loss = -1), "
                                    "which splits wide rows into column blocks", P.k_int);
        const unsigned grid = (unsigned)cdiv(P.B, kThreads / 64);
        switch (model) {
            case EMG_TRANSE_L1: hipLaunchKernelGGL(train_forward_generic_kernel<EMG_TRANSE_L1>, dim3(grid), dim3(kThreads), 0, st, P); break;
            case EMG_TRANSE_L2: hipLaunchKernelGGL(train_forward_generic_kernel<EMG_TRANSE_L2>, dim3(grid), dim3(kThreads), 0, st, P); break;
            case EMG_DISTMULT: hipLaunchKernelGGL(train_forward_generic_kernel<EMG_DISTMULT>, dim3(grid), dim3(kThreads), 0, st, P); break;
            case EMG_COMPLEX: hipLaunchKernelGGL(train_forward_generic_kernel<EMG_COMPLEX>, dim3(grid), dim3(kThreads), 0, st, P); break;
            case EMG_HOLE: hipLaunchKernelGGL(train_forward_generic_kernel<EMG_HOLE>, dim3(grid), dim3(kThreads), 0, st, P); break;
        }
    }
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

__global__ void finalize_scores_kernel(int model, float scale, float* s, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (model == EMG_TRANSE_L2) s[i] = -sqrtf(s[i]);
    else if (model == EMG_HOLE) s[i] = scale * s[i];
}

}  // namespace emg

using namespace emg;

// TransE with any positive order of the norm (EMG_TRANSE_P, `ord` = INFINITY: the largest |component|; TransE.py:208-216
// hands `norm` to tf.norm as ord).  Generic kernels, a wave per triple (inference) or per positive group (training): any
// width, rows read through the caches — orders 1 and 2 are the tuned models, this is the reference's remaining freedom.
//   score      f = -(sum_c |d_c|^ord)^(1/ord),  d = (e_s + e_p) - e_o;  ord = inf: f = -max_c |d_c|
//   gradient   df/dd_c = -sgn(d_c) |d_c|^(ord-1) / ||d||^(ord-1);  ord = inf: -sgn(d_c) [|d_c| = max] / #maxima (tf.reduce_max's
//              gradient is shared by tied maxima); a zero vector has gradient zero
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
// ||(a + p) - b||_ord of one triple (all 64 lanes; the same value in every lane); *ties: number of components equal to the maximum
__device__ __forceinline__ float transe_p_norm(const float* a, const float* p, const float* b, int k_int, float ord, int lane, float* ties) {
    const bool mx = isinf(ord);
    float acc = 0.f;
    for (int c = lane; c < k_int; c += 64) {
        const float d = fabsf((a[c] + p[c]) - b[c]);
        acc = mx ? fmaxf(acc, d) : acc + powf(d, ord);
    }
    if (!mx) return powf(wave_sum_f(acc), 1.0f / ord);
    acc = wave_max_f(acc);
    if (ties) {
        float cnt = 0.f;
        for (int c = lane; c < k_int; c += 64) cnt += fabsf((a[c] + p[c]) - b[c]) == acc ? 1.f : 0.f;
        *ties = wave_sum_f(cnt);
    }
    return acc;
}

__global__ __launch_bounds__(256) void score_transe_p_kernel(const float* __restrict__ ent, int64_t ld_ent, const float* __restrict__ rel,
                                                             int64_t ld_rel, int k_int, float ord, const int32_t* __restrict__ spo, int64_t n,
                                                             float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (t >= n) return;
    const float nrm = transe_p_norm(ent + (int64_t)spo[3 * t] * ld_ent, rel + (int64_t)spo[3 * t + 1] * ld_rel,
                                    ent + (int64_t)spo[3 * t + 2] * ld_ent, k_int, ord, lane, nullptr);
    if (lane == 0) out[t] = -nrm;
}

struct TransePTrain {
    const float* ent; int64_t ld_ent; const float* rel; int64_t ld_rel; int32_t k_int; float ord;
    const int32_t* pos; int64_t B; int32_t eta; const int32_t* codes;
    float* scores_pos; float* scores_neg;               // forward
    const float* g_pos; const float* g_neg;             // backward: dL/dscore
    float* contrib_ent; float* contrib_rel; int64_t ldc;
};

// scores of a positive and its eta negatives (eta-major codes: replacement | keep_subject << 31)
__global__ __launch_bounds__(256) void transe_p_forward_kernel(const TransePTrain P) {
    const int lane = threadIdx.x & 63;
    const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (g >= P.B) return;
    const float* es = P.ent + (int64_t)P.pos[3 * g] * P.ld_ent;
    const float* ep = P.rel + (int64_t)P.pos[3 * g + 1] * P.ld_rel;
    const float* eo = P.ent + (int64_t)P.pos[3 * g + 2] * P.ld_ent;
    const float f = -transe_p_norm(es, ep, eo, P.k_int, P.ord, lane, nullptr);
    if (lane == 0) P.scores_pos[g] = f;
    for (int j = 0; j < P.eta; ++j) {
        const int32_t code = P.codes[(int64_t)j * P.B + g];
        const float* er = P.ent + (int64_t)(code & 0x7fffffff) * P.ld_ent;
        const float fn = code < 0 ? -transe_p_norm(es, ep, er, P.k_int, P.ord, lane, nullptr) : -transe_p_norm(er, ep, eo, P.k_int, P.ord, lane, nullptr);
        if (lane == 0) P.scores_neg[(int64_t)j * P.B + g] = fn;
    }
}

// gradient rows of a positive group, in the contribution layout of every model: slot g = dE[s], B + g = dE[o], 2B + jB + g = the
// replacement of negative j, relation slot g = dR[p].  The shared rows accumulate in their slots (the lane that owns a column
// adds to it, positive first, negatives in ascending j: a fixed order).
__global__ __launch_bounds__(256) void transe_p_backward_kernel(const TransePTrain P) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (g >= P.B) return;
    const float* es = P.ent + (int64_t)P.pos[3 * g] * P.ld_ent;
    const float* ep = P.rel + (int64_t)P.pos[3 * g + 1] * P.ld_rel;
    const float* eo = P.ent + (int64_t)P.pos[3 * g + 2] * P.ld_ent;
    float* cs = P.contrib_ent + g * P.ldc;
    float* co = P.contrib_ent + (P.B + g) * P.ldc;
    float* cp = P.contrib_rel + g * P.ldc;
    const bool mx = isinf(P.ord);
    // u_c = -df/dd_c * (-1) ... the triple (a, p, b) with upstream gradient gs gets  da = dp = -gs u,  db = +gs u
    auto unit = [&](float d, float nrm, float ties) -> float {
        const float ad = fabsf(d);
        if (nrm == 0.f || ad == 0.f) return 0.f;
        const float sg = d > 0.f ? 1.f : -1.f;
        if (mx) return ad == nrm ? sg / ties : 0.f;
        return sg * powf(ad, P.ord - 1.f) / powf(nrm, P.ord - 1.f);
    };
    {
        float ties = 1.f;
        const float nrm = transe_p_norm(es, ep, eo, P.k_int, P.ord, lane, &ties);
        const float gs = P.g_pos[g];
        for (int c = lane; c < P.k_int; c += 64) {
            const float v = gs * unit((es[c] + ep[c]) - eo[c], nrm, ties);
            cs[c] = -v; cp[c] = -v; co[c] = v;
        }
    }
    for (int j = 0; j < P.eta; ++j) {
        const int32_t code = P.codes[(int64_t)j * P.B + g];
        const bool keep_s = code < 0;   // subject kept: the OBJECT was replaced
        const float* er = P.ent + (int64_t)(code & 0x7fffffff) * P.ld_ent;
        const float* a = keep_s ? es : er;
        const float* b = keep_s ? er : eo;
        float* cr = P.contrib_ent + (2 * P.B + (int64_t)j * P.B + g) * P.ldc;
        float ties = 1.f;
        const float nrm = transe_p_norm(a, ep, b, P.k_int, P.ord, lane, &ties);
        const float gs = P.g_neg[(int64_t)j * P.B + g];
        for (int c = lane; c < P.k_int; c += 64) {
            const float v = gs * unit((a[c] + ep[c]) - b[c], nrm, ties);
            cp[c] = cp[c] - v;
            if (keep_s) { cs[c] = cs[c] - v; cr[c] = v; }
            else { co[c] = co[c] + v; cr[c] = -v; }
        }
    }
}

static int transe_p_check(int32_t k_int, float ord, int64_t ld_ent, int64_t ld_rel) {
    EMG_REQUIRE(ord > 0.f && k_int > 0 && ld_ent >= k_int && ld_rel >= k_int, "EMG_TRANSE_P: the order of the norm (passed as `scale`) must be positive");
    return EMG_OK;
}

extern "C" int emg_train_forward(int model, const float* ent, int64_t n_ent, int64_t ld_ent, const float* rel,
                                 int64_t n_rel, int64_t ld_rel, int32_t k_int, float scale, const int32_t* pos,
                                 int64_t B, int32_t eta, const int32_t* codes, int32_t flags, float* scores_pos,
                                 float* scores_neg, void* stream) {
    if (B == 0) return EMG_OK;
    EMG_REQUIRE(ent && rel && pos && scores_pos, "emg_train_forward: null pointer");
    if (model == EMG_TRANSE_P) {
        int rc = transe_p_check(k_int, scale, ld_ent, ld_rel);
        if (rc != EMG_OK) return rc;
        EMG_REQUIRE(flags == EMG_SCORE_FINAL, "EMG_TRANSE_P: no partial (column-slab) scores");
        if (eta == 0) {
            hipLaunchKernelGGL(score_transe_p_kernel, dim3((unsigned)cdiv(B * 64, 256)), dim3(256), 0, (hipStream_t)stream, ent, ld_ent, rel, ld_rel,
                               (int)k_int, scale, pos, B, scores_pos);
        } else {
            EMG_REQUIRE(codes && scores_neg, "emg_train_forward: eta>0 needs codes and scores_neg");
            TransePTrain T{};
            T.ent = ent; T.ld_ent = ld_ent; T.rel = rel; T.ld_rel = ld_rel; T.k_int = k_int; T.ord = scale; T.pos = pos; T.B = B; T.eta = eta;
            T.codes = codes; T.scores_pos = scores_pos; T.scores_neg = scores_neg;
            hipLaunchKernelGGL(transe_p_forward_kernel, dim3((unsigned)cdiv(B * 64, 256)), dim3(256), 0, (hipStream_t)stream, T);
        }
        EMG_LAUNCH_CHECK();
        return EMG_OK;
    }
    EMG_REQUIRE(eta == 0 || (codes && scores_neg), "emg_train_forward: eta>0 needs codes and scores_neg");
    GroupParams P{};
    P.ent = ent; P.n_ent = n_ent; P.ld_ent = ld_ent; P.rel = rel; P.n_rel = n_rel; P.ld_rel = ld_rel;
    P.k_int = k_int; P.scale = scale; P.pos = pos; P.B = B; P.eta = eta; P.codes = codes; P.flags = flags;
    P.scores_pos = scores_pos; P.scores_neg = scores_neg;
    return run_group_pass(Pass::Forward, model, P, (hipStream_t)stream);
}

extern "C" int emg_score_triples(int model, const float* ent, int64_t n_ent, int64_t ld_ent, const float* rel,
                                 int64_t n_rel, int64_t ld_rel, int32_t k_int, float scale, const int32_t* spo,
                                 int64_t n, int32_t flags, float* out, void* stream) {
    return emg_train_forward(model, ent, n_ent, ld_ent, rel, n_rel, ld_rel, k_int, scale, spo, n, 0, nullptr, flags,
                             out, nullptr, stream);
}

extern "C" int emg_build_dest(const int32_t* pos, int64_t B, int32_t eta, const int32_t* codes, int32_t* dest_ent,
                              int32_t* dest_rel, void* stream) {
    if (B <= 0) return EMG_OK;
    EMG_REQUIRE(pos && dest_ent && dest_rel && (eta == 0 || codes), "emg_build_dest: null pointer");
    const int64_t n = B * (int64_t)(eta > 1 ? eta : 1);
    hipLaunchKernelGGL(build_dest_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, pos, B,
                       (int)eta, codes, dest_ent, dest_rel);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

namespace emg {
int train_backward_impl(const emg_backward_args* a, const Riders* riders, void* stream);
}
extern "C" int emg_train_backward_ex(const emg_backward_args* a, void* stream) { return emg::train_backward_impl(a, nullptr, stream); }

// riders (optional): preparation stages of later batches carried by this launch; a pass that cannot carry them (column
// blocks of wide rows, B == 0) launches them on their own first
int emg::train_backward_impl(const emg_backward_args* a, const Riders* riders, void* stream) {
    EMG_REQUIRE(a, "emg_train_backward_ex: null args");
    if (a->B == 0) return riders ? launch_riders_alone(*riders, (hipStream_t)stream) : EMG_OK;
    EMG_REQUIRE(a->ent && a->rel && a->pos && a->contrib_ent && a->contrib_rel, "emg_train_backward_ex: null pointer");
    EMG_REQUIRE(a->eta == 0 || a->codes, "emg_train_backward_ex: eta>0 needs codes");
    EMG_REQUIRE(a->ldc >= a->k_int, "emg_train_backward_ex: ldc < k_int");
    const bool fused = a->fused_loss >= 0;
    if (fused) {
        EMG_REQUIRE(a->fused_loss == EMG_LOSS_PAIRWISE || a->fused_loss == EMG_LOSS_NLL ||
                        a->fused_loss == EMG_LOSS_ABSOLUTE_MARGIN,
                    "emg_train_backward_ex: loss %d is not pair-local, use emg_train_forward + emg_loss", a->fused_loss);
        EMG_REQUIRE(a->loss_accum, "emg_train_backward_ex: fused loss needs loss_accum");
        EMG_REQUIRE(!a->bw_scores_pos && !a->bw_scores_neg, "emg_train_backward_ex: fused loss cannot take bw_scores");
    } else {
        EMG_REQUIRE(a->g_pos && (a->eta == 0 || a->g_neg), "emg_train_backward_ex: external dL/dscore missing");
    }
    if (a->model == EMG_TRANSE_P) {   // any order of the norm: generic kernels, external dL/dscore, every row through the apply
        EMG_REQUIRE(!fused && !a->single_ent && !a->fac_ws_ent && !a->ctl,
                    "emg_train_backward_ex: EMG_TRANSE_P trains through emg_train_forward + emg_loss + this call with fused_loss = -1, "
                    "without in-place updates, factored contributions or device-side step records");
        int rc = transe_p_check(a->k_int, a->scale, a->ld_ent, a->ld_rel);
        if (rc == EMG_OK && riders && riders->total) rc = launch_riders_alone(*riders, (hipStream_t)stream);
        if (rc != EMG_OK) return rc;
        TransePTrain T{};
        T.ent = a->ent; T.ld_ent = a->ld_ent; T.rel = a->rel; T.ld_rel = a->ld_rel; T.k_int = a->k_int; T.ord = a->scale; T.pos = a->pos;
        T.B = a->B; T.eta = a->eta; T.codes = a->codes; T.g_pos = a->g_pos; T.g_neg = a->g_neg;
        T.contrib_ent = a->contrib_ent; T.contrib_rel = a->contrib_rel; T.ldc = a->ldc;
        hipLaunchKernelGGL(transe_p_backward_kernel, dim3((unsigned)cdiv(a->B * 64, 256)), dim3(256), 0, (hipStream_t)stream, T);
        EMG_LAUNCH_CHECK();
        return EMG_OK;
    }
    GroupParams P{};
    P.ent = a->ent; P.n_ent = a->n_ent; P.ld_ent = a->ld_ent; P.rel = a->rel; P.n_rel = a->n_rel; P.ld_rel = a->ld_rel;
    P.k_int = a->k_int; P.scale = a->scale; P.pos = a->pos; P.B = a->B; P.eta = a->eta; P.codes = a->codes;
    P.g_pos = a->g_pos; P.g_neg = a->g_neg; P.bw_scores_pos = a->bw_scores_pos; P.bw_scores_neg = a->bw_scores_neg;
    P.fused_loss = a->fused_loss; P.margin = a->margin; P.loss_accum = a->loss_accum;
    EMG_REQUIRE(a->loss_slots >= 0 && a->loss_slots <= 4096 && (a->loss_slots & (a->loss_slots - 1)) == 0,
                "emg_train_backward_ex: loss_slots must be 0 or a power of two <= 4096");
    P.loss_mask = a->loss_slots > 1 ? (uint32_t)a->loss_slots - 1u : 0u;
    P.scores_pos = a->scores_pos_out; P.scores_neg = a->scores_neg_out;
    P.contrib_ent = a->contrib_ent; P.contrib_rel = a->contrib_rel; P.ldc = a->ldc;
    if (a->fac_ws_ent) {
        EMG_REQUIRE(!(a->model == EMG_TRANSE_L1 || a->model == EMG_TRANSE_L2),
                    "emg_train_backward_ex: factored contributions need a bilinear model — a TransE gradient row depends "
                    "on the replacement entity");
        const int64_t Bl = a->layout_B > 0 ? a->layout_B : a->B;
        int rc = factor_view(a->fac_ws_ent, a->fac_ws_ent_bytes, (2 + (int64_t)a->eta) * Bl, a->n_ent, &P.fac);
        if (rc != EMG_OK) return rc;
    }
    P.single_ent = a->single_ent;
    P.window = a->single_ent && a->inplace_window ? 1 : 0;
    if (a->single_ent) {
        EMG_REQUIRE(a->opt >= EMG_OPT_SGD && a->opt <= EMG_OPT_ADAM_LAZY, "emg_train_backward_ex: unknown optimizer");
        EMG_REQUIRE(!(a->opt == EMG_OPT_MOMENTUM || a->opt == EMG_OPT_ADAGRAD) || a->ent_state0,
                    "emg_train_backward_ex: optimizer needs ent_state0");
        EMG_REQUIRE(!(a->opt == EMG_OPT_ADAM || a->opt == EMG_OPT_ADAM_LAZY) || (a->ent_state0 && a->ent_state1),
                    "emg_train_backward_ex: adam needs both state tables");
        P.ent_rw = const_cast<float*>(a->ent);
        P.ent_state0 = a->ent_state0; P.ent_state1 = a->ent_state1; P.tag_ent = a->tag_ent; P.step = a->step;
        P.opt = make_opt_params(a->opt, a->hyper);
        // A folded LP regulariser (hyper[6] = lambda, hyper[7] = p): plain SGD has its own in-place form (IP 3: the pow / sign
        // code stays out of the other forms, where it costs the fused kernel a wave per SIMD); with a stateful optimizer every
        // gradient row goes through emg_apply_grouped, which folds it in
        if (a->hyper[6] != 0.f) {
            EMG_REQUIRE(a->opt == EMG_OPT_SGD, "emg_train_backward_ex: in-place singleton updates fold an LP regulariser for plain SGD "
                                               "only (pass single_ent = NULL and let emg_apply_grouped apply every row)");
            EMG_REQUIRE(a->lp_accum && a->hyper[7] >= 1.f && a->hyper[7] <= 3.f && a->tag_ent,
                        "emg_train_backward_ex: a folded LP regulariser with in-place updates needs lp_accum, p in {1, 2, 3} and the tag array");
            P.opt.lp_lambda = a->hyper[6]; P.opt.lp_p = (int)a->hyper[7];
            P.lp_accum = a->lp_accum;
        }
    }
    if (a->lr_hist && a->opt == EMG_OPT_SGD) {   // SGD + LP under the deferred dense pass: lagging singleton negatives replayed in the kernel (ip 7)
        EMG_REQUIRE(a->single_ent && !a->inplace_window && a->hyper[6] != 0.f && a->tag_ent && fused && a->step >= 1 && !a->ctl,
                    "emg_train_backward_ex: lr_hist with EMG_OPT_SGD is for the fused kernel with in-place updates and a folded LP regulariser");
        const bool cplx = a->model == EMG_COMPLEX || a->model == EMG_HOLE;
        const int n = cplx ? a->k_int / 2 : a->k_int;
        EMG_REQUIRE(n % 4 == 0 && n / 4 <= 128 && a->ld_ent % 4 == 0 && aligned16(a->ent),
                    "emg_train_backward_ex: lr_hist needs 16-byte rows of at most 128 chunks (per half for complex models)");
        P.lr_hist = a->lr_hist; P.upto = a->step - 1;
    } else if (a->lr_hist) {   // Adam's dense pass is deferred: singletons among the negatives lag and are replayed in the kernel (ip 6)
        EMG_REQUIRE(a->single_ent && a->inplace_window && a->opt == EMG_OPT_ADAM && a->hyper[6] == 0.f && a->tag_ent && fused && a->step >= 1,
                    "emg_train_backward_ex: lr_hist (lagging singletons) is for the fused kernel with in-place EMG_OPT_ADAM updates, no regulariser");
        const bool cplx = a->model == EMG_COMPLEX || a->model == EMG_HOLE;
        const int n = cplx ? a->k_int / 2 : a->k_int;
        EMG_REQUIRE(n % 4 == 0 && n / 4 <= 64 && a->ld_ent % 4 == 0 && aligned16(a->ent) && aligned16(a->ent_state0) && aligned16(a->ent_state1) && !a->ctl,
                    "emg_train_backward_ex: lr_hist needs 16-byte rows of at most 64 chunks (per half for complex models) and no device-side step record");
        P.lr_hist = a->lr_hist; P.upto = a->step - 1;
    }
    EMG_REQUIRE(a->layout_B == 0 || a->layout_B >= a->B, "emg_train_backward_ex: layout_B < B");
    P.ctl = (const StepCtl*)a->ctl;
    if (P.ctl) {   // the launch covers the capacity; the kernel reads the batch's rows and size from the record
        EMG_REQUIRE(a->layout_B > 0, "emg_train_backward_ex: a device-side step record needs layout_B (the launch size)");
        P.B = a->layout_B;
    }
    return run_group_pass(fused ? Pass::Fused : Pass::Backward, a->model, P, (hipStream_t)stream, riders);
}

extern "C" int emg_train_backward(int model, const float* ent, int64_t n_ent, int64_t ld_ent, const float* rel,
                                  int64_t n_rel, int64_t ld_rel, int32_t k_int, float scale, const int32_t* pos,
                                  int64_t B, int32_t eta, const int32_t* codes, const float* g_pos,
                                  const float* g_neg, float* contrib_ent, float* contrib_rel, int64_t ldc,
                                  int32_t* dest_ent, int32_t* dest_rel, void* stream) {
    if (B == 0) return EMG_OK;
    EMG_REQUIRE(dest_ent && dest_rel, "emg_train_backward: null pointer");
    emg_backward_args a{};
    a.model = model; a.ent = ent; a.n_ent = n_ent; a.ld_ent = ld_ent; a.rel = rel; a.n_rel = n_rel; a.ld_rel = ld_rel;
    a.k_int = k_int; a.scale = scale; a.pos = pos; a.B = B; a.eta = eta; a.codes = codes; a.fused_loss = -1;
    a.g_pos = g_pos; a.g_neg = g_neg; a.contrib_ent = contrib_ent; a.contrib_rel = contrib_rel; a.ldc = ldc;
    int rc = emg_train_backward_ex(&a, stream);
    if (rc != EMG_OK) return rc;
    return emg_build_dest(pos, B, eta, codes, dest_ent, dest_rel, stream);
}

extern "C" int emg_finalize_scores(int model, float scale, float* scores, int64_t n, void* stream) {
    EMG_REQUIRE(scores || n == 0, "emg_finalize_scores: null pointer");
    if (n == 0 || !(model == EMG_TRANSE_L2 || model == EMG_HOLE)) return EMG_OK;
    hipLaunchKernelGGL(finalize_scores_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, model,
                       scale, scores, n);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}
