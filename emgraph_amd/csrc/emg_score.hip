// emg_score.hip — fused embedding gather + score (K1+K2+K4) and its adjoint (K7).
//
// Replaces, per training batch, EmbeddingModel._lookup_embeddings (EmbeddingModel.py:490-533, three
// materialised tf.nn.embedding_lookup gathers), Model._fn (TransE.py:208-216, DistMult.py:201,
// ComplEx.py:288-298, HolE.py:189), the positive tiling (EmbeddingModel.py:724-729) and the
// TF-autodiff backward of all of them.
//
// Mapping to CDNA4: one LANE GROUP (16/32/64 lanes of a wave64) owns one positive triple.  The
// group keeps the s, p, o rows in VGPRs (16-byte global_load_dwordx4 per lane, rows are 16-byte
// aligned), scores the positive, then streams the eta replacement rows of that positive — the
// relation row and the kept side are read ONCE per group instead of once per negative:
// (3+eta) row reads per group instead of 3(1+eta).  k-reductions are __shfl_xor butterflies.
// HBM-bound by design: algorithmic bytes per group = 12 + (3+eta)*4*k_int + 4*(1+eta)  (DESIGN.md).
#include "emg_common.hpp"

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

template <int MODEL, int W, int NV, int LPG>
__device__ __forceinline__ void load_row(Row<MODEL, W, NV>& r, const float* __restrict__ base, int lg, int nchunks,
                                         int khalf) {
    constexpr int E = W * NV;
    RowTile<W, NV> t;
    load_tile<W, NV, LPG>(t, base, lg, nchunks);
#pragma unroll
    for (int e = 0; e < E; ++e) r.x[e] = t.x[e];
    if constexpr (is_complex<MODEL>::value) {
        load_tile<W, NV, LPG>(t, base + khalf, lg, nchunks);
#pragma unroll
        for (int e = 0; e < E; ++e) r.x[E + e] = t.x[e];
    }
}

template <int MODEL, int W, int NV, int LPG>
__device__ __forceinline__ void store_row(const Row<MODEL, W, NV>& r, float* __restrict__ base, int lg, int nchunks,
                                          int khalf) {
    constexpr int E = W * NV;
    RowTile<W, NV> t;
#pragma unroll
    for (int e = 0; e < E; ++e) t.x[e] = r.x[e];
    store_tile<W, NV, LPG>(t, base, lg, nchunks);
    if constexpr (is_complex<MODEL>::value) {
#pragma unroll
        for (int e = 0; e < E; ++e) t.x[e] = r.x[E + e];
        store_tile<W, NV, LPG>(t, base + khalf, lg, nchunks);
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
            acc += (pr * sr) * orr + (pr * si) * oi + (pi * sr) * oi - (pi * si) * orr;
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

// inner coefficient from g = dL/dscore and the k-reduced sum
template <int MODEL>
__device__ __forceinline__ float inner_coef(float g, float sum, float scale) {
    if constexpr (MODEL == EMG_TRANSE_L2) {
        const float nrm = sqrtf(sum);
        return nrm > 0.f ? g / nrm : 0.f;
    }
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
            ga.x[e] = fmaf(gi, pr * orr + pi * oi, ga.x[e]);
            ga.x[E + e] = fmaf(gi, pr * oi - pi * orr, ga.x[E + e]);
            gp.x[e] = fmaf(gi, sr * orr + si * oi, gp.x[e]);
            gp.x[E + e] = fmaf(gi, sr * oi - si * orr, gp.x[E + e]);
            gb.x[e] = fmaf(gi, pr * sr - pi * si, gb.x[e]);
            gb.x[E + e] = fmaf(gi, pr * si + pi * sr, gb.x[E + e]);
        }
    }
}

struct GroupParams {
    const float* ent; int64_t n_ent; int64_t ld_ent;
    const float* rel; int64_t n_rel; int64_t ld_rel;
    int32_t k_int; int32_t khalf; int32_t nchunks; float scale;
    const int32_t* pos; int64_t B; int32_t eta; const int32_t* codes; int32_t flags;
    float* scores_pos; float* scores_neg;
    // backward only
    const float* g_pos; const float* g_neg;
    float* contrib_ent; float* contrib_rel; int64_t ldc;
    int32_t* dest_ent; int32_t* dest_rel;
};

constexpr int kThreads = 256;
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
            code[u] = P.codes[(int64_t)j * P.B + g];
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
            R a, b;
#pragma unroll
            for (int e = 0; e < R::N; ++e) {
                a.x[e] = keep_s ? rs.x[e] : re[u].x[e];
                b.x[e] = keep_s ? re[u].x[e] : ro.x[e];
            }
            part[u] = partial_score<MODEL, W, NV>(a, rp, b);
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
// backward: gradient rows per group (no atomics; summed per destination by emg_apply_rows)
// ---------------------------------------------------------------------------------------------
template <int MODEL, int W, int NV, int LPG>
__global__ __launch_bounds__(kThreads) void train_backward_kernel(const GroupParams P) {
    using R = Row<MODEL, W, NV>;
    const int lg = threadIdx.x % LPG;
    int64_t g = ((int64_t)blockIdx.x * kThreads + threadIdx.x) / LPG;
    const bool active = g < P.B;
    if (!active) g = P.B - 1;

    const int32_t s = P.pos[3 * g + 0], p = P.pos[3 * g + 1], o = P.pos[3 * g + 2];
    R rs, rp, ro, gs, gp, go;
    load_row<MODEL, W, NV, LPG>(rs, P.ent + (int64_t)s * P.ld_ent, lg, P.nchunks, P.khalf);
    load_row<MODEL, W, NV, LPG>(rp, P.rel + (int64_t)p * P.ld_rel, lg, P.nchunks, P.khalf);
    load_row<MODEL, W, NV, LPG>(ro, P.ent + (int64_t)o * P.ld_ent, lg, P.nchunks, P.khalf);
#pragma unroll
    for (int e = 0; e < R::N; ++e) gs.x[e] = gp.x[e] = go.x[e] = 0.f;

    {
        float sum = 0.f;
        if constexpr (MODEL == EMG_TRANSE_L2) sum = group_sum<LPG>(partial_score<MODEL, W, NV>(rs, rp, ro));
        const float gi = inner_coef<MODEL>(P.g_pos[g], sum, P.scale);
        accum_grads<MODEL, W, NV>(rs, rp, ro, gi, gs, gp, go);
    }

    const int64_t B = P.B;
    constexpr int U = 2;
    for (int j0 = 0; j0 < P.eta; j0 += U) {
        int32_t code[U];
        float gj[U];
        R re[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = min(j0 + u, P.eta - 1);
            code[u] = P.codes[(int64_t)j * B + g];
            gj[u] = P.g_neg[(int64_t)j * B + g];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int32_t repl = code[u] & 0x7fffffff;
            load_row<MODEL, W, NV, LPG>(re[u], P.ent + (int64_t)repl * P.ld_ent, lg, P.nchunks, P.khalf);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u;
            if (j >= P.eta) break;
            const bool keep_s = code[u] < 0;
            R a, b, ta, tb;
#pragma unroll
            for (int e = 0; e < R::N; ++e) {
                a.x[e] = keep_s ? rs.x[e] : re[u].x[e];
                b.x[e] = keep_s ? re[u].x[e] : ro.x[e];
                ta.x[e] = 0.f;
                tb.x[e] = 0.f;
            }
            float sum = 0.f;
            if constexpr (MODEL == EMG_TRANSE_L2) sum = group_sum<LPG>(partial_score<MODEL, W, NV>(a, rp, b));
            const float gi = inner_coef<MODEL>(gj[u], sum, P.scale);
            accum_grads<MODEL, W, NV>(a, rp, b, gi, ta, gp, tb);
            R row;
#pragma unroll
            for (int e = 0; e < R::N; ++e) {
                gs.x[e] += keep_s ? ta.x[e] : 0.f;
                go.x[e] += keep_s ? 0.f : tb.x[e];
                row.x[e] = keep_s ? tb.x[e] : ta.x[e];
            }
            if (active) {
                const int64_t slot = 2 * B + (int64_t)j * B + g;
                store_row<MODEL, W, NV, LPG>(row, P.contrib_ent + slot * P.ldc, lg, P.nchunks, P.khalf);
                if (lg == 0) P.dest_ent[slot] = code[u] & 0x7fffffff;
            }
        }
    }
    if (active) {
        store_row<MODEL, W, NV, LPG>(gs, P.contrib_ent + g * P.ldc, lg, P.nchunks, P.khalf);
        store_row<MODEL, W, NV, LPG>(go, P.contrib_ent + (B + g) * P.ldc, lg, P.nchunks, P.khalf);
        store_row<MODEL, W, NV, LPG>(gp, P.contrib_rel + g * P.ldc, lg, P.nchunks, P.khalf);
        if (lg == 0) {
            P.dest_ent[g] = s;
            P.dest_ent[B + g] = o;
            P.dest_rel[g] = p;
        }
    }
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
            acc += (pr * sr) * orr + (pr * si) * oi + (pi * sr) * oi - (pi * si) * orr;
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

// ---------------------------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------------------------
enum class Pass { Forward, Backward };

template <int MODEL, int W, int NV, int LPG>
static void launch_group(Pass pass, const GroupParams& P, hipStream_t st) {
    const int groups_per_block = kThreads / LPG;
    const unsigned grid = (unsigned)cdiv(P.B, groups_per_block);
    if (pass == Pass::Forward)
        hipLaunchKernelGGL((train_forward_kernel<MODEL, W, NV, LPG>), dim3(grid), dim3(kThreads), 0, st, P);
    else
        hipLaunchKernelGGL((train_backward_kernel<MODEL, W, NV, LPG>), dim3(grid), dim3(kThreads), 0, st, P);
}

// returns false when no register-tiled variant fits
template <int MODEL>
static bool dispatch_model(Pass pass, GroupParams& P, bool vec, hipStream_t st) {
    const int n = is_complex<MODEL>::value ? P.khalf : P.k_int;
    if (vec) {
        P.nchunks = n / 4;
        const int c = P.nchunks;
        if (c <= 16) launch_group<MODEL, 4, 1, 16>(pass, P, st);
        else if (c <= 32) launch_group<MODEL, 4, 1, 32>(pass, P, st);
        else if (c <= 64) launch_group<MODEL, 4, 1, 64>(pass, P, st);
        else if (c <= 128) launch_group<MODEL, 4, 2, 64>(pass, P, st);
        else return false;
    } else {
        P.nchunks = n;
        const int c = P.nchunks;
        if (c <= 64) launch_group<MODEL, 1, 1, 64>(pass, P, st);
        else if (c <= 128) launch_group<MODEL, 1, 2, 64>(pass, P, st);
        else if (c <= 256) launch_group<MODEL, 1, 4, 64>(pass, P, st);
        else if (c <= 512) launch_group<MODEL, 1, 8, 64>(pass, P, st);
        else return false;
    }
    return true;
}

static int run_group_pass(Pass pass, int model, GroupParams& P, hipStream_t st) {
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
    if (pass == Pass::Backward) vec = vec && (P.ldc % 4 == 0) && aligned16(P.contrib_ent) && aligned16(P.contrib_rel);
    bool ok = false;
    switch (model) {
        case EMG_TRANSE_L1: ok = dispatch_model<EMG_TRANSE_L1>(pass, P, vec, st); break;
        case EMG_TRANSE_L2: ok = dispatch_model<EMG_TRANSE_L2>(pass, P, vec, st); break;
        case EMG_DISTMULT: ok = dispatch_model<EMG_DISTMULT>(pass, P, vec, st); break;
        case EMG_COMPLEX: ok = dispatch_model<EMG_COMPLEX>(pass, P, vec, st); break;
        case EMG_HOLE: ok = dispatch_model<EMG_HOLE>(pass, P, vec, st); break;
    }
    if (!ok) {
        if (pass == Pass::Backward)
            return fail(EMG_ENOSUP, "train_backward: k_int=%d exceeds the register-tiled limit", P.k_int);
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

extern "C" int emg_train_forward(int model, const float* ent, int64_t n_ent, int64_t ld_ent, const float* rel,
                                 int64_t n_rel, int64_t ld_rel, int32_t k_int, float scale, const int32_t* pos,
                                 int64_t B, int32_t eta, const int32_t* codes, int32_t flags, float* scores_pos,
                                 float* scores_neg, void* stream) {
    if (B == 0) return EMG_OK;
    EMG_REQUIRE(ent && rel && pos && scores_pos, "emg_train_forward: null pointer");
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

extern "C" int emg_train_backward(int model, const float* ent, int64_t n_ent, int64_t ld_ent, const float* rel,
                                  int64_t n_rel, int64_t ld_rel, int32_t k_int, float scale, const int32_t* pos,
                                  int64_t B, int32_t eta, const int32_t* codes, const float* g_pos,
                                  const float* g_neg, float* contrib_ent, float* contrib_rel, int64_t ldc,
                                  int32_t* dest_ent, int32_t* dest_rel, void* stream) {
    if (B == 0) return EMG_OK;
    EMG_REQUIRE(ent && rel && pos && g_pos && contrib_ent && contrib_rel && dest_ent && dest_rel,
                "emg_train_backward: null pointer");
    EMG_REQUIRE(eta == 0 || (codes && g_neg), "emg_train_backward: eta>0 needs codes and g_neg");
    EMG_REQUIRE(ldc >= k_int, "emg_train_backward: ldc < k_int");
    GroupParams P{};
    P.ent = ent; P.n_ent = n_ent; P.ld_ent = ld_ent; P.rel = rel; P.n_rel = n_rel; P.ld_rel = ld_rel;
    P.k_int = k_int; P.scale = scale; P.pos = pos; P.B = B; P.eta = eta; P.codes = codes;
    P.g_pos = g_pos; P.g_neg = g_neg; P.contrib_ent = contrib_ent; P.contrib_rel = contrib_rel; P.ldc = ldc;
    P.dest_ent = dest_ent; P.dest_rel = dest_rel;
    return run_group_pass(Pass::Backward, model, P, (hipStream_t)stream);
}

extern "C" int emg_finalize_scores(int model, float scale, float* scores, int64_t n, void* stream) {
    EMG_REQUIRE(scores || n == 0, "emg_finalize_scores: null pointer");
    if (n == 0 || !(model == EMG_TRANSE_L2 || model == EMG_HOLE)) return EMG_OK;
    hipLaunchKernelGGL(finalize_scores_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, model,
                       scale, scores, n);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}
