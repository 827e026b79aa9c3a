// emg_common.hpp — shared host/device helpers for libemgraph_hip.so (gfx950 / wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "emgraph_hip.h"

namespace emg {

int fail(int code, const char* fmt, ...);

#define EMG_HIP(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return ::emg::fail(EMG_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),   \
                               __FILE__, __LINE__);                                                \
    } while (0)

#define EMG_REQUIRE(cond, ...)                                   \
    do {                                                         \
        if (!(cond)) return ::emg::fail(EMG_EINVAL, __VA_ARGS__); \
    } while (0)

#define EMG_LAUNCH_CHECK() EMG_HIP(hipGetLastError())

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11).  Bit-for-bit the generator restated in
// oracle/emgraph_oracle.py::philox4x32_10 and oracle/emg_oracle.c.
// ---------------------------------------------------------------------------------------------
struct Philox4 {
    uint32_t v[4];
};

__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                          uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0;
        const uint64_t p1 = (uint64_t)M1 * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    Philox4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

// draws for corruption row j: keep_subject bit and an index in [0, n_choices)
__host__ __device__ __forceinline__ void corruption_draw(uint64_t seed, uint64_t counter, uint64_t j,
                                                         uint64_t n_choices, uint32_t* keep_subject,
                                                         uint32_t* idx) {
    const Philox4 o = philox4x32_10((uint32_t)j, (uint32_t)(j >> 32), (uint32_t)counter, (uint32_t)(counter >> 32),
                                    (uint32_t)seed, (uint32_t)(seed >> 32));
    *keep_subject = o.v[0] & 1u;
    const uint64_t r64 = ((uint64_t)o.v[2] << 32) | (uint64_t)o.v[1];
#if defined(__HIP_DEVICE_COMPILE__)
    *idx = (uint32_t)__umul64hi(r64, n_choices);
#else
    *idx = (uint32_t)(((unsigned __int128)r64 * n_choices) >> 64);
#endif
}

// ---------------------------------------------------------------------------------------------
// Register tile of one embedding row slice held by a lane group.
//   W  = floats per chunk (4: 16-byte global_load_dwordx4; 1: scalar fallback)
//   NV = chunks per lane;  a group of LPG lanes covers LPG*NV chunks.
// chunk c of a lane: c = lg + it*LPG  (lg = lane in group)  -> floats [c*W, c*W+W)
// ---------------------------------------------------------------------------------------------
template <int W, int NV>
struct RowTile {
    float x[W * NV];
};

// ZERO = false: lanes past the row's end keep what they re-read (finite values of the row itself) — for operands whose
// partner is zero there anyway; the select that zeroes a value needs the value, i.e. it ends the load's flight
template <int W, int NV, int LPG, bool ZERO = true>
__device__ __forceinline__ void load_tile(RowTile<W, NV>& t, const float* __restrict__ row, int lg, int nchunks) {
#pragma unroll
    for (int it = 0; it < NV; ++it) {
        // NO branch around the load: a lane past the row's end re-reads the last chunk (the same cache line as its
        // neighbour's) and zeroes the value.  A load under a lane predicate compiles to `s_cbranch_execz` around it, i.e.
        // a path without the load, and hipcc then waits for every OLDER load with s_waitcnt vmcnt(0) — which also waits for
        // the loads issued after it: the fused kernel's rolling window of replacement rows degenerated to one row in flight.
        const int c = lg + it * LPG;
        const bool on = c < nchunks;
        const int cc = on ? c : nchunks - 1;
        if constexpr (W == 4) {
            const float4 v = *reinterpret_cast<const float4*>(row + 4 * cc);
            t.x[4 * it + 0] = (on || !ZERO) ? v.x : 0.f; t.x[4 * it + 1] = (on || !ZERO) ? v.y : 0.f;
            t.x[4 * it + 2] = (on || !ZERO) ? v.z : 0.f; t.x[4 * it + 3] = (on || !ZERO) ? v.w : 0.f;
        } else {
            const float v = row[cc];
            t.x[it] = (on || !ZERO) ? v : 0.f;
        }
    }
}

// STREAM: the row is written once and read once by a later kernel (a contribution row): non-temporal store, so that it
// does not displace table rows from the caches on its way to memory (tools/hbm_ceiling: the fused kernel's access mix
// moves 4.7 TB/s with plain stores, 5.1 TB/s with its streamed rows stored non-temporally)
template <int W, int NV, int LPG, bool STREAM = false>
__device__ __forceinline__ void store_tile(const RowTile<W, NV>& t, float* __restrict__ row, int lg, int nchunks) {
    typedef float nt_float4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int it = 0; it < NV; ++it) {
        const int c = lg + it * LPG;
        if (c < nchunks) {
            if constexpr (W == 4) {
                if constexpr (STREAM) {
                    const nt_float4 v = {t.x[4 * it + 0], t.x[4 * it + 1], t.x[4 * it + 2], t.x[4 * it + 3]};
                    __builtin_nontemporal_store(v, reinterpret_cast<nt_float4*>(row + 4 * c));
                } else {
                    *reinterpret_cast<float4*>(row + 4 * c) =
                        make_float4(t.x[4 * it + 0], t.x[4 * it + 1], t.x[4 * it + 2], t.x[4 * it + 3]);
                }
            } else {
                row[c] = t.x[it];
            }
        }
    }
}

// Sum over the LPG lanes of a group, the same bits in every lane.  A butterfly, lowest lane bit first: lanes (i, i ^ 1), (i, i ^ 2),
// (i, i ^ 4), (i, i ^ 8) as DPP operands of the adds themselves (quad_perm; row_half_mirror / row_mirror reach the other quad /
// the other half of the row, whose lanes all hold the same partial sum by then), the four rows of 16 through v_readlane:
// (r0 + r1) + (r2 + r3).  __shfl_xor is ds_bpermute_b32 — six dependent trips through the LDS crossbar, ~0.5 us per negative of a
// wave's chain: nothing where 12 waves per SIMD wait in line (C3), the whole kernel time where there are 1.7 (the reference's own
// batch sizes; tools/sweep_small.py: fused time = 16 us + 0.55 us per negative whatever the depth of the row window).
// A group narrower than the wave: the missing levels would add zeros, so every LPG gives the same bits for the same row.
#ifndef EMG_DPP_SUM
#define EMG_DPP_SUM 1
#endif
template <int CTRL>
__device__ __forceinline__ float dpp_lane_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int LPG>
__device__ __forceinline__ float group_sum(float v) {
#if EMG_DPP_SUM
    static_assert(LPG == 16 || LPG == 32 || LPG == 64, "a group is 16, 32 or 64 lanes");
    v += dpp_lane_f<0xB1>(v);    // quad_perm [1, 0, 3, 2]
    v += dpp_lane_f<0x4E>(v);    // quad_perm [2, 3, 0, 1]
    v += dpp_lane_f<0x141>(v);   // row_half_mirror
    v += dpp_lane_f<0x140>(v);   // row_mirror
    if constexpr (LPG == 16) return v;
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16)),
                r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    const float a = r0 + r1, b = r2 + r3;
    if constexpr (LPG == 32) return (__lane_id() & 32) ? b : a;
    return a + b;
#else
#pragma unroll
    for (int off = 1; off < LPG; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
#endif
}

// sign(d) in {-1, 0, +1} (-0 for d = -0).  Two compares, two selects and a subtraction — with their wait states eight issues per
// element, a third of TransE-L1's instructions per negative — or: d 2^100 2^100 saturates every nonzero d (denormals included)
// far beyond +-1, and the median of (that, -1, 1) clamps it: three instructions, the multiplications packed two elements each.
// (A NaN gives -1 / +1 / NaN instead of 0: a diverged model either way.)
#ifndef EMG_SGN_MED3
#define EMG_SGN_MED3 1
#endif
__device__ __forceinline__ float sgnf(float d) {
#if EMG_SGN_MED3
#pragma clang fp contract(off)
    const float h = 0x1p+100f;
    return __builtin_amdgcn_fmed3f((d * h) * h, -1.f, 1.f);
#else
    return (float)(d > 0.f) - (float)(d < 0.f);
#endif
}

// ---------------------------------------------------------------------------------------------
// Optimizer element update (Keras / TF-2.2 rules; training/{sgd,momentum,adagrad,adam}.py).
// Shared by the segmented apply kernel and by the in-place singleton path of the backward kernel
// so that both produce bit-identical results.
// ---------------------------------------------------------------------------------------------
struct OptParams {
    int opt;
    float lr, mu, beta1, beta2, eps, lr_t;
    float lp_lambda;  // LP regulariser folded into the update (0 = none): g += lambda * p * |w|^(p-1) * sign(w)
    int lp_p;
};

// LP regulariser (regularizers/lp.py:107-113), per element: value |w|^p and gradient of lambda * |w|^p
__device__ __forceinline__ float lp_pow(float a, int p) {
    if (p == 1) return a;
    if (p == 2) return a * a;
    if (p == 3) return a * a * a;
    return powf(a, (float)p);
}
__device__ __forceinline__ float lp_grad(const OptParams& P, float w) {
#pragma clang fp contract(off)
    const float a = fabsf(w);
    return P.lp_lambda * (float)P.lp_p * (P.lp_p == 1 ? 1.f : lp_pow(a, P.lp_p - 1)) * sgnf(w);
}
// p == 2 (the reference's default, regularizers/_regularizer_constants.py), taken by lp_fold / lp_fold_p123 themselves: lambda * 2 * |w| * sgn(w) is fl(2 lambda * w)
// — the product with the sign is exact and rounding is sign-symmetric, signed zeros included — and |w|^2 is fl(w * w): two
// multiplications where the generic form spends an abs, a square, two selects, three multiplications and the sign's med3.  The same
// VALUE as the generic expressions at p = 2 and the same bits up to the sign of a zero: for w = -0 the generic product is +0 (the
// sign function returns 0), this one -0, which shows only when the data gradient is itself -0.  Every device path takes THIS form at
// p = 2 (lp_fold and lp_fold_p123 branch here), so the paths agree with each other bit for bit; against the oracle's generic numpy
// expression the comparison is by value (tests/test_hip_kernels.py::test_lp_fold_forms_agree...).  (The deferred replay is bound by
// exactly this arithmetic: DESIGN 7)
__device__ __forceinline__ void lp_fold_p2(const OptParams& P, float w, float& g, float& lp_acc) {
#pragma clang fp contract(off)
    g += (P.lp_lambda * 2.f) * w;
    lp_acc += w * w;
}
// the same for p in {1, 2, 3} only (the fused kernel's in-place form: no powf, whose inlined code costs it a wave per SIMD);
// same expressions, same bits as lp_fold for these p
__device__ __forceinline__ void lp_fold_p123(const OptParams& P, float w, float& g, float& lp_acc) {
#pragma clang fp contract(off)
    if (P.lp_p == 2) { lp_fold_p2(P, w, g, lp_acc); return; }   // (same bits; a uniform branch, folded where p is a compile-time constant)
    const float a = fabsf(w);
    const float a2 = a * a;
    const float pm1 = P.lp_p == 1 ? 1.f : (P.lp_p == 2 ? a : a2);          // |w|^(p-1)
    g += P.lp_lambda * (float)P.lp_p * pm1 * sgnf(w);
    lp_acc += P.lp_p == 1 ? a : (P.lp_p == 2 ? a2 : a2 * a);
}
// every row's update sees the gradient of the WHOLE loss: data term (summed contributions, 0 for a row no triple of
// the batch touches) + the regulariser's, both evaluated at the pre-update value (EmbeddingModel.py:786-820)
__device__ __forceinline__ void lp_fold(const OptParams& P, float w, float& g, float& lp_acc) {
    if (P.lp_lambda != 0.f) {
        if (P.lp_p == 2) { lp_fold_p2(P, w, g, lp_acc); return; }   // (same bits)
        g += lp_grad(P, w);
        lp_acc += lp_pow(fabsf(w), P.lp_p);
    }
}

__device__ __forceinline__ float opt_sgd_elem(const OptParams& P, float w, float g) {
#pragma clang fp contract(off)
    return w - P.lr * g;
}

// num / (sqrt(root) + eps) of Adagrad's and Adam's update.  EMG_OPT_FAST_RECIP (default, round 4): the hardware's v_sqrt_f32 and
// v_rcp_f32 (1 ulp each) and one multiplication — the step is within 3 ulp of the correctly rounded form (sqrtf and an IEEE
// division: ~25 instructions of range fix-ups each, the bulk of the arithmetic of a replayed Adam step, which is what bounds the
// deferred pass's catch-up and the scoring kernel's in-register replay).  eps >= 1e-7 keeps the reciprocal's argument in the normal
// range.  ONE function for every path (dense pass, catch-up, apply, in-place forms): their results stay bit-identical to each
// other; against Keras' CPU arithmetic (parity unpinned) the difference is the one TF's own GPU kernels have.
#ifndef EMG_OPT_FAST_RECIP
#define EMG_OPT_FAST_RECIP 1
#endif
__device__ __forceinline__ float opt_ratio(float num, float root, float eps) {
#pragma clang fp contract(off)
#if EMG_OPT_FAST_RECIP
    // (eps >= FLT_MIN by construction — make_opt_params raises a denormal or zero eps of a C-ABI caller to it, so that root == 0, an
    // untouched row of the dense pass, never meets v_rcp_f32's flushed argument: inf, and 0 * inf would poison the table — and there
    // is NO max on the device side: a NaN root or state stays a NaN here, as in Keras, instead of becoming a huge finite step)
    return num * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(root) + eps);
#else
    return num / (sqrtf(root) + eps);
#endif
}

__device__ __forceinline__ void opt_update_elem(const OptParams& P, float& w, float g, float* s0, float* s1) {
#pragma clang fp contract(off)  // every op rounded separately: identical bits wherever this is inlined
    if (P.opt == EMG_OPT_SGD) {
        w = w - P.lr * g;
    } else if (P.opt == EMG_OPT_MOMENTUM) {  // Keras SGD(momentum): v = mu*v - lr*g ; w += v
        const float v = P.mu * (*s0) - P.lr * g;
        *s0 = v;
        w = w + v;
    } else if (P.opt == EMG_OPT_ADAGRAD) {  // acc += g^2 ; w -= lr*g/(sqrt(acc)+eps)
        const float a = *s0 + g * g;
        *s0 = a;
        w = w - opt_ratio(P.lr * g, a, P.eps);
    } else {  // Adam: m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; w -= lr_t m/(sqrt(v)+eps)
        const float m = P.beta1 * (*s0) + (1.f - P.beta1) * g;
        const float v = P.beta2 * (*s1) + (1.f - P.beta2) * g * g;
        *s0 = m;
        *s1 = v;
        w = w - opt_ratio(P.lr_t * m, v, P.eps);
    }
}

// Adam's step with a ZERO gradient (the dense pass on a row no triple touched, replayed by the deferred pass): opt_update_elem's
// values with g = +0, in 9 instructions per element instead of 13 — the replay is arithmetic, and the kernels that carry it
// (catch-up, the scoring kernel's form 6) are bound by instruction issue (PMC, C3 + Adam: 12.3 k VALU instructions per wave, the
// SIMDs issuing 79 % of the time).
//   m: b1 m + (1 - b1) 0 = b1 m + (+0): the ADDITION stays — it turns a product that is -0 into +0, as the dense pass's own does;
//   v: b2 v + (1 - b2) 0 0 = b2 v + (+0) = b2 v exactly (v is a sum of squares: never -0, never negative).
#ifndef EMG_ZERO_GRAD_TRIM
#define EMG_ZERO_GRAD_TRIM 1   // 0: A/B aid — opt_update_elem's expressions with a zero the compiler cannot see through
#endif
__device__ __forceinline__ void adam_decay_elem(const OptParams& P, float& m, float& v) {
#pragma clang fp contract(off)
#if EMG_ZERO_GRAD_TRIM
    m = P.beta1 * m + 0.0f;
    v = P.beta2 * v;
#else
    float g = 0.f;
    asm volatile("" : "+v"(g));
    m = P.beta1 * m + (1.f - P.beta1) * g;
    v = P.beta2 * v + (1.f - P.beta2) * g * g;
#endif
}
__device__ __forceinline__ void adam_zero_grad_elem(const OptParams& P, float& w, float& m, float& v) {
#pragma clang fp contract(off)
    adam_decay_elem(P, m, v);
    w = w - opt_ratio(P.lr_t * m, v, P.eps);
}

// ---------------------------------------------------------------------------------------------
// Loss pieces shared by loss_kernel (emg_train.hip) and the fused train kernel (emg_score.hip).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float clip75(float v) { return fminf(fmaxf(v, -75.0f), 75.0f); }  // losses/utils.py:44-53
__device__ __forceinline__ float in75(float v) { return (v >= -75.0f && v <= 75.0f) ? 1.f : 0.f; }
__device__ __forceinline__ float log1pexp_naive(float x) { return logf(1.0f + expf(x)); }  // nll.py:59 literal form
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }
// tf.maximum(v, 0) propagates NaN (fmaxf would swallow it and hide a diverged model from the NaN check)
__device__ __forceinline__ float relu_nan(float v) { return (v >= 0.f || v != v) ? v : 0.f; }

// The three losses whose dL/dneg_j depends only on (pos_i, neg_j): pairwise.py:69, nll.py:55-59,
// absolute_margin.py:69.  Returns dL/dneg; adds this pair's loss and dL/dpos share.
// The positive's share is the same for each of its negatives (the reference tiles the positive eta times,
// EmbeddingModel.py:724-729), so it is evaluated once per positive (PosTerms) and added once per negative.
struct PosTerms {
    float loss, grad;
};

// The per-negative NLL terms with the hardware's 1-ulp transcendental instructions (v_exp_f32, v_log_f32, v_rcp_f32) instead of
// libm's expf / logf and an IEEE division: 14 instructions instead of 45, in a loop whose cost at the reference's batch sizes is
// the instruction stream of ONE wave per SIMD (tools/sweep_small.py: ~0.4 us per negative).  exp(x), |x| <= 75: 2^(x log2 e) with the
// product's rounding error carried into a first-order correction (<= 2 ulp); sigmoid = e rcp(1 + e) (<= 2 ulp); the loss VALUE's
// log(1 + e) = ln 2 log2(1 + e) (absolute error ~1e-7 per term, summed in double).  EMG_LOSS_FAST = 0 (default): libm / IEEE forms —
// measured on one box, ms per step, fast against libm: C2 0.0519 / 0.0530, C5 0.1191 / 0.1202, C3 0.3478 / 0.3437 (the large batch is
// memory-bound): a microsecond where it helps, so the accurate forms stay.
#ifndef EMG_LOSS_FAST
#define EMG_LOSS_FAST 0
#endif
__device__ __forceinline__ float exp75_fast(float x) {
    const float t = x * 1.44269502f;                         // log2(e) = 1.44269502 + 1.92596299e-8
    float r = fmaf(x, 1.44269502f, -t);                      // the product's rounding error, exact
    r = fmaf(x, 1.92596299e-8f, r);
    const float e0 = __builtin_amdgcn_exp2f(t);
    return fmaf(e0 * r, 0.693147181f, e0);                   // 2^(t + r) = 2^t (1 + r ln 2 + ...)
}

__device__ __forceinline__ PosTerms local_loss_pos(int loss, float pos) {
    PosTerms t;
    t.loss = 0.f;
    t.grad = 0.f;
    if (loss == EMG_LOSS_NLL) {
        const float pc = clip75(pos);
        t.loss = log1pexp_naive(-pc);
        t.grad = -in75(pos) * sigmoidf(-pc);
    }
    return t;
}

__device__ __forceinline__ float local_loss_neg(int loss, float pos, const PosTerms& t, float neg, float margin,
                                                float& loss_acc, float& gpos_acc) {
    if (loss == EMG_LOSS_PAIRWISE) {
        const float v = margin - pos + neg;
        const float act = v >= 0.f ? 1.f : 0.f;  // TF MaximumGrad: x >= y takes the gradient
        loss_acc += relu_nan(v);
        gpos_acc -= act;
        return act;
    }
    if (loss == EMG_LOSS_NLL) {
#if EMG_LOSS_FAST
        const float e = exp75_fast(clip75(neg));
        const float one_e = 1.0f + e;
        loss_acc += t.loss + 0.693147181f * __builtin_amdgcn_logf(one_e);   // nll.py:59 literal log(1+exp(x))
        gpos_acc += t.grad;
        return in75(neg) * (e * __builtin_amdgcn_rcpf(one_e));              // sigmoid(clip(neg))
#else
        const float e = expf(clip75(neg));    // <= e^75 = 3.7e32, finite in f32
        loss_acc += t.loss + logf(1.0f + e);  // nll.py:59 literal log(1+exp(x))
        gpos_acc += t.grad;
        return in75(neg) * (e / (1.0f + e));  // sigmoid(clip(neg))
#endif
    }
    const float v = margin + neg;  // absolute_margin
    loss_acc += relu_nan(v) - pos;
    gpos_acc -= 1.f;
    return v >= 0.f ? 1.f : 0.f;
}

// add the lanes' partial sums of a wave into *dst (double): one atomic per wave that has something to add
__device__ __forceinline__ void wave_add_double(double* dst, float partial) {
    double v = (double)partial;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if (dst && (threadIdx.x & 63) == 0 && v != 0.0) atomicAdd(dst, v);
}

// the regulariser's partial sums of a workgroup's waves -> ONE double atomic (a launch of 16 k waves adding to one address one
// by one took longer than the replay itself)
__device__ __forceinline__ void block_add_double(double* dst, float partial) {
    __shared__ double s_part[16];   // (workgroups of up to 1024 threads)
    double v = (double)partial;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (unsigned w = 0; w < (blockDim.x + 63) / 64; ++w) t += s_part[w];
        if (dst && t != 0.0) atomicAdd(dst, t);
    }
    __syncthreads();   // (s_part may be written again by the next call of the same launch)
}

// ---------------------------------------------------------------------------------------------
// Hand-off of rows between waves of ONE launch that may sit on different CUs / XCDs (whose L2s are not coherent):
// the producer writes THROUGH to memory (sc1), waits until its stores have left (s_waitcnt vmcnt(0)) and bumps a
// device-scope atomic counter; the consumer that sees the final count reads past its caches (sc1 loads).
// ---------------------------------------------------------------------------------------------
typedef float vfloat4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store4_through(float* p, float a, float b, float c, float d) {
    const vfloat4 v = {a, b, c, d};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ vfloat4 load4_through(const float* p) {   // issue only: wait_memory() + landed() before use
    vfloat4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void wait_memory() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void landed(vfloat4& v) { asm volatile("" : "+v"(v)); }   // orders uses of v behind wait_memory()

static inline OptParams make_opt_params(int opt, const float* hyper) {
    OptParams o;
    o.opt = opt == EMG_OPT_ADAM_LAZY ? EMG_OPT_ADAM : opt;
    o.lr = hyper[0]; o.mu = hyper[1]; o.beta1 = hyper[2]; o.beta2 = hyper[3]; o.eps = hyper[4]; o.lr_t = hyper[5];
    if (o.eps >= 0.f && o.eps < 1.17549435e-38f) o.eps = 1.17549435e-38f;   // opt_ratio's reciprocal needs a normal denominator (a NaN eps stays a NaN)
    o.lp_lambda = hyper[6]; o.lp_p = (int)hyper[7];
    return o;
}

}  // namespace emg
