// emg_train.hip — corruption draws (K3/K14), loss + dL/dscore (K5), LP regulariser (K6), row clip (K9).
#include "emg_common.hpp"

namespace emg {

// ---------------------------------------------------------------------------------------------
// K3/K14: codes[j] = replacement | keep_subject<<31   (protocol.py:598-641)
// ---------------------------------------------------------------------------------------------
__global__ void corrupt_codes_kernel(int64_t n, int side, uint64_t n_choices, const int32_t* __restrict__ entities_list,
                                     uint64_t seed, uint64_t counter, const int32_t* __restrict__ inj_mask,
                                     const int32_t* __restrict__ inj_repl, int32_t* __restrict__ codes) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint32_t keep, idx;
    if (inj_repl) {
        idx = (uint32_t)inj_repl[j];
        keep = inj_mask ? (uint32_t)(inj_mask[j] != 0) : 0u;
    } else {
        corruption_draw(seed, counter, (uint64_t)j, n_choices, &keep, &idx);
    }
    if (side == EMG_SIDE_O) keep = 1u;       // protocol.py:606 keep subject, corrupt object
    else if (side == EMG_SIDE_S) keep = 0u;  // :607-608
    const uint32_t repl = entities_list ? (uint32_t)entities_list[idx] : idx;  // :616-619 / :635-641
    codes[j] = (int32_t)((repl & 0x7fffffffu) | (keep << 31));
}

// protocol.py:643-656: subjects = keep_s*s + keep_o*repl ; objects = keep_o*o + keep_s*repl
__global__ void corrupt_expand_kernel(const int32_t* __restrict__ pos, int64_t B, int64_t n,
                                      const int32_t* __restrict__ codes, int32_t* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int64_t i = j % B;
    const int32_t code = codes[j];
    const int32_t repl = code & 0x7fffffff;
    const bool keep_s = code < 0;
    out[3 * j + 0] = keep_s ? pos[3 * i + 0] : repl;
    out[3 * j + 1] = pos[3 * i + 1];
    out[3 * j + 2] = keep_s ? repl : pos[3 * i + 2];
}

// ---------------------------------------------------------------------------------------------
// K5 losses.  One thread per positive i; negatives of i are rows (sd*eta + j)*B + i.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float log_sigmoid(float x) {  // tf.math.log_sigmoid = -softplus(-x)
    return -(fmaxf(-x, 0.f) + log1pf(expf(-fabsf(x))));
}

template <int LOSS>
__global__ __launch_bounds__(256) void loss_kernel(const float* __restrict__ sp, const float* __restrict__ sn,
                                                   int64_t B, int eta, int n_sides, float margin, float alpha,
                                                   double* __restrict__ loss_accum, float* __restrict__ g_pos,
                                                   float* __restrict__ g_neg) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float loss = 0.f;
    if (i < B) {
        const float pos = sp[i];
        float gp = 0.f;
        for (int sd = 0; sd < n_sides; ++sd) {
            const float* snp = sn + (int64_t)sd * eta * B + i;
            float* gnp = g_neg + (int64_t)sd * eta * B + i;
            if constexpr (LOSS == EMG_LOSS_PAIRWISE || LOSS == EMG_LOSS_NLL || LOSS == EMG_LOSS_ABSOLUTE_MARGIN) {
                const PosTerms pt = local_loss_pos(LOSS, pos);
                for (int j = 0; j < eta; ++j)
                    gnp[(int64_t)j * B] = local_loss_neg(LOSS, pos, pt, snp[(int64_t)j * B], margin, loss, gp);
            } else if constexpr (LOSS == EMG_LOSS_SELF_ADVERSARIAL) {  // self_adversarial.py:98-110
                loss += -log_sigmoid(margin + pos);
                gp += -sigmoidf(-(margin + pos));
                float mx = -INFINITY;
                for (int j = 0; j < eta; ++j) mx = fmaxf(mx, alpha * snp[(int64_t)j * B]);
                float den = 0.f, wsum = 0.f;
                for (int j = 0; j < eta; ++j) {
                    const float nv = snp[(int64_t)j * B];
                    const float w = expf(alpha * nv - mx);
                    den += w;
                    wsum += w * log_sigmoid(-nv - margin);
                }
                const float sbar = wsum / den;  // sum_j p_j * ell_j
                loss -= sbar;
                for (int j = 0; j < eta; ++j) {
                    const float nv = snp[(int64_t)j * B];
                    const float w = expf(alpha * nv - mx) / den;
                    const float ell = log_sigmoid(-nv - margin);
                    const float dell = -sigmoidf(nv + margin);
                    gnp[(int64_t)j * B] = -(w * dell + alpha * w * (ell - sbar));  // softmax not stop-gradiented
                }
            } else {  // multiclass_nll: nll_multiclass.py:70-81
                const float pc = clip75(pos);
                const float pe = expf(pc);
                float den = pe;
                for (int j = 0; j < eta; ++j) den += expf(clip75(snp[(int64_t)j * B]));
                loss += -logf(pe / den);
                gp += -(1.f - pe / den) * in75(pos);
                for (int j = 0; j < eta; ++j) {
                    const float nv = snp[(int64_t)j * B];
                    gnp[(int64_t)j * B] = in75(nv) * expf(clip75(nv)) / den;
                }
            }
        }
        g_pos[i] = gp;
    }
    // block reduction of the loss (double), one atomic per block
    double v = (double)loss;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double t = part[0] + part[1] + part[2] + part[3];
        if (t != 0.0) atomicAdd(loss_accum, t);
    }
}

// ---------------------------------------------------------------------------------------------
// K6: LP regulariser over a full table (regularizers/lp.py:107-113)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float powi_abs(float a, int p) {
    if (p == 1) return a;
    if (p == 2) return a * a;
    if (p == 3) return a * a * a;
    return powf(a, (float)p);
}

__global__ __launch_bounds__(256) void lp_kernel(float* __restrict__ w, int64_t n_rows, int64_t ld, int k_int,
                                                 float lambda, int p, float step, double* __restrict__ loss_accum,
                                                 float* __restrict__ contrib, int64_t ldc, int32_t* __restrict__ dest) {
    double acc = 0.0;
    const int64_t total = n_rows * (int64_t)k_int;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / k_int;
        const int c = (int)(t - r * k_int);
        float* ptr = w + r * ld + c;
        const float x = *ptr;
        const float a = fabsf(x);
        acc += (double)powi_abs(a, p);
        if (step != 0.f || contrib) {
            // d/dx lambda*|x|^p = lambda*p*|x|^(p-1)*sign(x)
            const float g = lambda * (float)p * (p == 1 ? 1.f : powi_abs(a, p - 1)) * sgnf(x);
            if (contrib) {
                contrib[r * ldc + c] = g;
                if (c == 0) dest[r] = (int32_t)r;
            } else {
                *ptr = x - step * g;
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && loss_accum) atomicAdd(loss_accum, (double)lambda * (part[0] + part[1] + part[2] + part[3]));
}

// K9: tf.clip_by_norm(W, clip_norm, axes=1): rows with ||row|| > c are scaled by c/||row||
__global__ __launch_bounds__(256) void clip_rows_kernel(float* __restrict__ w, int64_t n_rows, int64_t ld, int k_int,
                                                        float max_norm) {
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= n_rows) return;
    float* row = w + r * ld;
    float acc = 0.f;
    for (int c = lane; c < k_int; c += 64) acc = fmaf(row[c], row[c], acc);
    acc = group_sum<64>(acc);
    const float nrm = sqrtf(acc);
    if (nrm > max_norm) {
        const float f = max_norm / nrm;
        for (int c = lane; c < k_int; c += 64) row[c] *= f;
    }
}

}  // namespace emg

using namespace emg;

extern "C" int emg_corrupt_codes(int64_t B, int32_t eta, int side, int64_t n_choices, const int32_t* entities_list,
                                 uint64_t seed, uint64_t draw_counter, const int32_t* inj_mask,
                                 const int32_t* inj_repl, int32_t* codes, void* stream) {
    EMG_REQUIRE(side >= EMG_SIDE_S && side <= EMG_SIDE_SO, "emg_corrupt_codes: bad side %d", side);
    EMG_REQUIRE(B >= 0 && eta >= 0, "emg_corrupt_codes: negative sizes");
    const int64_t n = B * eta;
    if (n == 0) return EMG_OK;
    EMG_REQUIRE(codes, "emg_corrupt_codes: null codes");
    EMG_REQUIRE(n_choices > 0 && n_choices < ((int64_t)1 << 31), "emg_corrupt_codes: n_choices=%lld out of range",
                (long long)n_choices);
    EMG_REQUIRE(!(side == EMG_SIDE_SO && inj_repl && !inj_mask), "emg_corrupt_codes: injected 's+o' needs inj_mask");
    hipLaunchKernelGGL(corrupt_codes_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, n, side,
                       (uint64_t)n_choices, entities_list, seed, draw_counter, inj_mask, inj_repl, codes);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

extern "C" int emg_corrupt_expand(const int32_t* pos, int64_t B, int32_t eta, const int32_t* codes, int32_t* out_spo,
                                  void* stream) {
    const int64_t n = B * eta;
    if (n <= 0) return EMG_OK;
    EMG_REQUIRE(pos && codes && out_spo, "emg_corrupt_expand: null pointer");
    hipLaunchKernelGGL(corrupt_expand_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, pos, B, n,
                       codes, out_spo);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

extern "C" int emg_loss(int loss, const float* scores_pos, const float* scores_neg, int64_t B, int32_t eta,
                        int32_t n_sides, float margin, float alpha, double* loss_accum, float* g_pos, float* g_neg,
                        void* stream) {
    EMG_REQUIRE(loss >= EMG_LOSS_PAIRWISE && loss <= EMG_LOSS_MULTICLASS_NLL, "emg_loss: unknown loss %d", loss);
    EMG_REQUIRE(B >= 0 && eta >= 1 && n_sides >= 1, "emg_loss: bad sizes");
    if (B == 0) return EMG_OK;
    EMG_REQUIRE(scores_pos && scores_neg && loss_accum && g_pos && g_neg, "emg_loss: null pointer");
    const dim3 grid((unsigned)cdiv(B, 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define EMG_LAUNCH_LOSS(L) \
    hipLaunchKernelGGL(loss_kernel<L>, grid, block, 0, st, scores_pos, scores_neg, B, eta, n_sides, margin, alpha, loss_accum, g_pos, g_neg)
    switch (loss) {
        case EMG_LOSS_PAIRWISE: EMG_LAUNCH_LOSS(EMG_LOSS_PAIRWISE); break;
        case EMG_LOSS_NLL: EMG_LAUNCH_LOSS(EMG_LOSS_NLL); break;
        case EMG_LOSS_ABSOLUTE_MARGIN: EMG_LAUNCH_LOSS(EMG_LOSS_ABSOLUTE_MARGIN); break;
        case EMG_LOSS_SELF_ADVERSARIAL: EMG_LAUNCH_LOSS(EMG_LOSS_SELF_ADVERSARIAL); break;
        default: EMG_LAUNCH_LOSS(EMG_LOSS_MULTICLASS_NLL); break;
    }
#undef EMG_LAUNCH_LOSS
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

extern "C" int emg_lp_regularizer(float* table, int64_t n_rows, int64_t ld, int32_t k_int, float lambda, int32_t p,
                                  float grad_scale_lr, double* loss_accum, void* stream) {
    EMG_REQUIRE(table && n_rows >= 0 && ld >= k_int && k_int > 0 && p >= 1, "emg_lp_regularizer: bad arguments");
    if (n_rows == 0) return EMG_OK;
    const int64_t total = n_rows * (int64_t)k_int;
    const unsigned grid = (unsigned)(cdiv(total, 256) < 4096 ? cdiv(total, 256) : 4096);
    hipLaunchKernelGGL(lp_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, table, n_rows, ld, (int)k_int, lambda,
                       (int)p, grad_scale_lr, loss_accum, (float*)nullptr, (int64_t)0, (int32_t*)nullptr);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

extern "C" int emg_lp_grad_rows(const float* table, int64_t n_rows, int64_t ld, int32_t k_int, float lambda, int32_t p,
                                float* contrib, int64_t ldc, int32_t* dest, double* loss_accum, void* stream) {
    EMG_REQUIRE(table && contrib && dest && n_rows >= 0 && ld >= k_int && ldc >= k_int && k_int > 0 && p >= 1,
                "emg_lp_grad_rows: bad arguments");
    if (n_rows == 0) return EMG_OK;
    const int64_t total = n_rows * (int64_t)k_int;
    const unsigned grid = (unsigned)(cdiv(total, 256) < 4096 ? cdiv(total, 256) : 4096);
    hipLaunchKernelGGL(lp_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, const_cast<float*>(table), n_rows, ld,
                       (int)k_int, lambda, (int)p, 0.f, loss_accum, contrib, ldc, dest);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

// ---------------------------------------------------------------------------------------------
// Table initialisers on the device (initializers/{glorot_uniform,uniform,normal}.py: the reference's draws come from
// TensorFlow's generators and cannot be reproduced — parity-unpinned — so the contract is the distribution):
// element (r, c) of the table takes word (c & 3) of Philox4x32-10(counter = (r * k_int + c) / 4, stream, seed).
// ---------------------------------------------------------------------------------------------
__global__ void init_table_kernel(int kind, float* __restrict__ table, int64_t n_rows, int64_t ld, int k_int, float a, float b,
                                  uint64_t seed, uint64_t stream_id) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // group of 4 consecutive elements of the flat [n_rows, k_int] table
    const int64_t total = n_rows * (int64_t)k_int;
    if (4 * q >= total) return;
    const Philox4 o = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32),
                                    (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint32_t w[4] = {o.v[0], o.v[1], o.v[2], o.v[3]};
    float v[4];
    if (kind == 0) {   // U[a, b)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = a + (b - a) * ((float)(w[i] >> 8) * 5.9604644775390625e-08f);
    } else {           // N(a, b^2): Box-Muller on the two pairs of words
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
            const float u1 = ((float)(w[i] >> 8) + 0.5f) * 5.9604644775390625e-08f;   // (0, 1)
            const float u2 = (float)(w[i + 1] >> 8) * 5.9604644775390625e-08f;
            const float rad = sqrtf(-2.0f * logf(u1));
            float sn, cs;
            sincosf(6.283185307179586f * u2, &sn, &cs);
            v[i] = a + b * rad * cs;
            v[i + 1] = a + b * rad * sn;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t e = 4 * q + i;
        if (e < total) table[(e / k_int) * ld + (e % k_int)] = v[i];
    }
}

extern "C" int emg_init_table(int kind, float* table, int64_t n_rows, int64_t ld, int32_t k_int, float a, float b,
                              uint64_t seed, uint64_t stream_id, void* stream) {
    EMG_REQUIRE(kind == 0 || kind == 1, "emg_init_table: kind must be 0 (uniform [a, b)) or 1 (normal mean a, std b)");
    EMG_REQUIRE(table && n_rows >= 0 && ld >= k_int && k_int > 0, "emg_init_table: bad arguments");
    if (n_rows == 0) return EMG_OK;
    const int64_t groups = (n_rows * (int64_t)k_int + 3) / 4;
    hipLaunchKernelGGL(init_table_kernel, dim3((unsigned)cdiv(groups, 256)), dim3(256), 0, (hipStream_t)stream, kind, table, n_rows,
                       ld, (int)k_int, a, b, seed, stream_id);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

// rows[j] -> table[ids[j]] for the ids inside the table (a wave per row; the batch-sharded step with the optimizer state sharded by
// owner: every replica takes the owners' UPDATED rows; ids are distinct, the padding of the fixed-capacity all-gather is >= n_rows)
__global__ __launch_bounds__(256) void scatter_rows_kernel(float* __restrict__ table, int64_t n_rows, int64_t ld, int k_int,
                                                           const float* __restrict__ rows, int64_t ldr, const int32_t* __restrict__ ids, int64_t n) {
    const int lane = threadIdx.x & 63;
    const int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (j >= n) return;
    const int64_t id = ids[j];
    if (id < 0 || id >= n_rows) return;
    for (int c = lane; c < k_int; c += 64) table[id * ld + c] = rows[j * ldr + c];
}

extern "C" int emg_scatter_rows(float* table, int64_t n_rows, int64_t ld, int32_t k_int, const float* rows, int64_t ldr,
                                const int32_t* ids, int64_t n, void* stream) {
    EMG_REQUIRE(table && rows && ids && n_rows >= 0 && ld >= k_int && ldr >= k_int && k_int > 0 && n >= 0, "emg_scatter_rows: bad arguments");
    if (n == 0) return EMG_OK;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)cdiv(n * 64, 256)), dim3(256), 0, (hipStream_t)stream, table, n_rows, ld,
                       (int)k_int, rows, ldr, ids, n);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

extern "C" int emg_clip_rows(float* table, int64_t n_rows, int64_t ld, int32_t k_int, float max_norm, void* stream) {
    EMG_REQUIRE(table && n_rows >= 0 && ld >= k_int && k_int > 0, "emg_clip_rows: bad arguments");
    if (n_rows == 0) return EMG_OK;
    hipLaunchKernelGGL(clip_rows_kernel, dim3((unsigned)cdiv(n_rows * 64, 256)), dim3(256), 0, (hipStream_t)stream, table,
                       n_rows, ld, (int)k_int, max_norm);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}
