// emg_apply.hip — K8: deterministic row-sparse optimizer apply.
//
// The reference hands TF IndexedSlices (row ids + gradient rows) to Keras optimizers
// (training/sgd.py:97, momentum.py:63, adagrad.py:42, adam.py:45).  Here the backward pass has
// written one gradient row per (positive group, role) without atomics; this file
//   1. radix-sorts (destination row, contribution index) — stable, so equal destinations keep
//      index order and the float sum order is fixed => bit-reproducible training (the reference's
//      refit-determinism test, tests/emgraph/models/test_models.py:338-367);
//   2. one wave per segment head sums the segment's rows (16-byte coalesced loads) and performs
//      the optimizer update of that table row exactly once.
// HBM-bound streaming: reads every contribution row once, read-modify-writes each touched row once.
#include <string.h>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "emg_common.hpp"

namespace emg {

struct ApplyParams {
    float* table; int64_t n_rows; int64_t ld; int32_t k_int;
    float* state0; float* state1; int32_t* tag; int32_t step;
    const float* contrib; int64_t ldc;
    const uint32_t* keys; const uint32_t* vals; int64_t n;
    float lr, mu, beta1, beta2, eps, lr_t;
};

__global__ void iota_kernel(uint32_t* v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (uint32_t)i;
}

template <int OPT>
__device__ __forceinline__ void update_elem(float& w, float g, float* s0, float* s1, const ApplyParams& P) {
    if constexpr (OPT == EMG_OPT_SGD) {
        w = w - P.lr * g;
    } else if constexpr (OPT == EMG_OPT_MOMENTUM) {  // Keras SGD(momentum): v = mu*v - lr*g ; w += v
        const float v = P.mu * (*s0) - P.lr * g;
        *s0 = v;
        w = w + v;
    } else if constexpr (OPT == EMG_OPT_ADAGRAD) {  // acc += g^2 ; w -= lr*g/(sqrt(acc)+eps)
        const float a = *s0 + g * g;
        *s0 = a;
        w = w - P.lr * g / (sqrtf(a) + P.eps);
    } else {  // Adam: m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; w -= lr_t m/(sqrt(v)+eps)
        const float m = P.beta1 * (*s0) + (1.f - P.beta1) * g;
        const float v = P.beta2 * (*s1) + (1.f - P.beta2) * g * g;
        *s0 = m;
        *s1 = v;
        w = w - P.lr_t * m / (sqrtf(v) + P.eps);
    }
}

// one wave per sorted position; only segment heads work
template <int OPT, int W>
__global__ __launch_bounds__(256) void apply_rows_kernel(const ApplyParams P) {
    const int lane = threadIdx.x & 63;
    const int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (t >= P.n) return;
    const uint32_t key = P.keys[t];
    if (t > 0 && P.keys[t - 1] == key) return;
    if ((int64_t)key >= P.n_rows) return;  // defensive: never write outside the table
    int64_t end = t + 1;
    while (end < P.n && P.keys[end] == key) ++end;

    float* wrow = P.table + (int64_t)key * P.ld;
    float* s0row = P.state0 ? P.state0 + (int64_t)key * P.ld : nullptr;
    float* s1row = P.state1 ? P.state1 + (int64_t)key * P.ld : nullptr;
    const int nchunks = P.k_int / W;
    for (int c = lane; c < nchunks; c += 64) {
        float acc[W];
#pragma unroll
        for (int w = 0; w < W; ++w) acc[w] = 0.f;
        for (int64_t u = t; u < end; ++u) {
            const float* src = P.contrib + (int64_t)P.vals[u] * P.ldc + (int64_t)c * W;
            if constexpr (W == 4) {
                const float4 v = *reinterpret_cast<const float4*>(src);
                acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
            } else {
                acc[0] += src[0];
            }
        }
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const int64_t off = (int64_t)c * W + w;
            float wv = wrow[off];
            update_elem<OPT>(wv, acc[w], s0row ? s0row + off : nullptr, s1row ? s1row + off : nullptr, P);
            wrow[off] = wv;
        }
    }
    if (P.tag && lane == 0) P.tag[key] = P.step;
}

// Keras Adam's sparse apply is dense-equivalent (every row: m*=b1, v*=b2, w -= lr_t m/(sqrt v + eps));
// rows touched this step were fully handled by apply_rows_kernel and are skipped via tag.
__global__ __launch_bounds__(256) void adam_untouched_kernel(const ApplyParams P) {
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= P.n_rows) return;
    if (P.tag[r] == P.step) return;
    float* w = P.table + r * P.ld;
    float* m = P.state0 + r * P.ld;
    float* v = P.state1 + r * P.ld;
    for (int c = lane; c < P.k_int; c += 64) {
        const float mm = P.beta1 * m[c];
        const float vv = P.beta2 * v[c];
        m[c] = mm;
        v[c] = vv;
        w[c] = w[c] - P.lr_t * mm / (sqrtf(vv) + P.eps);
    }
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static int sort_temp_bytes(int64_t n, size_t* bytes) {
    *bytes = 0;
    if (n <= 0) return EMG_OK;
    EMG_HIP(rocprim::radix_sort_pairs(nullptr, *bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                      (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n, 0, 32, (hipStream_t)0,
                                      false));
    return EMG_OK;
}

}  // namespace emg

using namespace emg;

extern "C" int64_t emg_apply_workspace_bytes(int64_t n_contrib, int64_t n_rows) {
    (void)n_rows;
    if (n_contrib <= 0) return 256;
    size_t tmp = 0;
    if (sort_temp_bytes(n_contrib, &tmp) != EMG_OK) return -1;
    return (int64_t)(3 * align256((size_t)n_contrib * 4) + align256(tmp) + 256);
}

extern "C" int emg_apply_rows(int opt, float* table, int64_t n_rows, int64_t ld, int32_t k_int, float* state0,
                              float* state1, int32_t* tag, int32_t step, const float* contrib, int64_t ldc,
                              const int32_t* dest, int64_t n_contrib, const float* hyper, void* workspace,
                              int64_t workspace_bytes, void* stream) {
    EMG_REQUIRE(opt >= EMG_OPT_SGD && opt <= EMG_OPT_ADAM_LAZY, "emg_apply_rows: unknown optimizer %d", opt);
    EMG_REQUIRE(table && hyper && n_rows > 0 && ld >= k_int && k_int > 0, "emg_apply_rows: bad table arguments");
    EMG_REQUIRE(n_rows < ((int64_t)1 << 31), "emg_apply_rows: too many rows");
    EMG_REQUIRE(n_contrib == 0 || (contrib && dest && workspace && ldc >= k_int), "emg_apply_rows: bad contribution arguments");
    EMG_REQUIRE(!(opt == EMG_OPT_MOMENTUM || opt == EMG_OPT_ADAGRAD) || state0, "emg_apply_rows: optimizer needs state0");
    EMG_REQUIRE(!(opt == EMG_OPT_ADAM || opt == EMG_OPT_ADAM_LAZY) || (state0 && state1),
                "emg_apply_rows: adam needs state0 and state1");
    EMG_REQUIRE(opt != EMG_OPT_ADAM || tag, "emg_apply_rows: dense-equivalent adam needs the tag array");
    hipStream_t st = (hipStream_t)stream;

    ApplyParams P{};
    P.table = table; P.n_rows = n_rows; P.ld = ld; P.k_int = k_int;
    P.state0 = state0; P.state1 = state1; P.tag = tag; P.step = step;
    P.contrib = contrib; P.ldc = ldc; P.n = n_contrib;
    P.lr = hyper[0]; P.mu = hyper[1]; P.beta1 = hyper[2]; P.beta2 = hyper[3]; P.eps = hyper[4]; P.lr_t = hyper[5];

    if (n_contrib > 0) {
        size_t tmp = 0;
        int rc = sort_temp_bytes(n_contrib, &tmp);
        if (rc != EMG_OK) return rc;
        const size_t kb = align256((size_t)n_contrib * 4);
        EMG_REQUIRE((int64_t)(3 * kb + align256(tmp)) <= workspace_bytes, "emg_apply_rows: workspace too small (%lld < %lld)",
                    (long long)workspace_bytes, (long long)(3 * kb + align256(tmp)));
        char* ws = (char*)workspace;
        uint32_t* keys_out = (uint32_t*)ws;
        uint32_t* vals_in = (uint32_t*)(ws + kb);
        uint32_t* vals_out = (uint32_t*)(ws + 2 * kb);
        void* temp = ws + 3 * kb;
        hipLaunchKernelGGL(iota_kernel, dim3((unsigned)cdiv(n_contrib, 256)), dim3(256), 0, st, vals_in, n_contrib);
        EMG_LAUNCH_CHECK();
        int end_bit = 1;
        while (end_bit < 32 && ((int64_t)1 << end_bit) < n_rows) ++end_bit;
        EMG_HIP(rocprim::radix_sort_pairs(temp, tmp, (const uint32_t*)dest, keys_out, (const uint32_t*)vals_in, vals_out,
                                          (size_t)n_contrib, 0, end_bit, st, false));
        P.keys = keys_out;
        P.vals = vals_out;
        const bool vec = (k_int % 4 == 0) && (ld % 4 == 0) && (ldc % 4 == 0) && aligned16(table) && aligned16(contrib) &&
                         (!state0 || aligned16(state0)) && (!state1 || aligned16(state1));
        const dim3 grid((unsigned)cdiv(n_contrib * 64, 256)), block(256);
#define EMG_LAUNCH_APPLY(O)                                                                   \
    do {                                                                                      \
        if (vec) hipLaunchKernelGGL((apply_rows_kernel<O, 4>), grid, block, 0, st, P);        \
        else hipLaunchKernelGGL((apply_rows_kernel<O, 1>), grid, block, 0, st, P);            \
    } while (0)
        switch (opt) {
            case EMG_OPT_SGD: EMG_LAUNCH_APPLY(EMG_OPT_SGD); break;
            case EMG_OPT_MOMENTUM: EMG_LAUNCH_APPLY(EMG_OPT_MOMENTUM); break;
            case EMG_OPT_ADAGRAD: EMG_LAUNCH_APPLY(EMG_OPT_ADAGRAD); break;
            default: EMG_LAUNCH_APPLY(EMG_OPT_ADAM); break;
        }
#undef EMG_LAUNCH_APPLY
        EMG_LAUNCH_CHECK();
    }
    if (opt == EMG_OPT_ADAM) {
        hipLaunchKernelGGL(adam_untouched_kernel, dim3((unsigned)cdiv(n_rows * 64, 256)), dim3(256), 0, st, P);
        EMG_LAUNCH_CHECK();
    }
    return EMG_OK;
}
