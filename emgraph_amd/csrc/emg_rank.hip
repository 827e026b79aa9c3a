// emg_rank.hip — filtered 1-vs-all ranking (K10-K13).
//
// Replaces, per test triple, generate_corruptions_for_eval (protocol.py:448-528: a [2|E|,3] int
// array), three gathers of [2|E|,k] rows + _fn over them (EmbeddingModel.py:1856-1866),
// perform_comparision (:1989-2033) and the filter gathers (:1942-1963).  Nothing of that is
// materialised: DistMult/ComplEx/HolE become a [rows x k_int]·[k_int x |E|] contraction on the
// f32-input MFMA (v_mfma_f32_32x32x2_f32: exact f32, k-ordered fmaf chain) with the
// int32(score*1e5) comparison against the positive fused into the epilogue, so only two int32
// counters per query row ever reach HBM.  TransE (not a contraction) is an LDS-tiled VALU kernel.
//
// CANONICAL ORDER (what makes ranks bit-exact against oracle/emg_oracle.c): every score is the
// chain acc_{k+1} = fmaf(q_k, e_k, acc_k), k ascending from acc_0 = +0 (TransE-L1:
// acc + |q_k - e_k|; L2: fmaf(d,d,acc)).  The MFMA kernel, the positive scorer and the filter
// scorer all produce exactly that chain, so the test entity and every filter entity compare
// identically wherever they are scored.
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "emg_common.hpp"

// The canonical order below is only canonical if the compiler never fuses a*b+c on its own:
// every fused multiply-add in this file is an explicit __fmaf_rn / MFMA.
#pragma clang fp contract(off)

namespace emg {

typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef float float16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int cmp_int(float score) { return (int)__fmul_rn(score, 100000.0f); }  // EmbeddingModel.py:2010-2014

__device__ __forceinline__ bool is_dot_model(int model) { return model >= EMG_DISTMULT; }

// ---------------------------------------------------------------------------------------------
// query construction
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void row_to_query(int64_t r, int64_t n_q, int side_mode, int64_t* qi, bool* obj_side) {
    if (side_mode == EMG_EVAL_S) { *qi = r; *obj_side = false; }
    else if (side_mode == EMG_EVAL_O) { *qi = r; *obj_side = true; }
    else { *obj_side = r < n_q; *qi = r < n_q ? r : r - n_q; }
}

__global__ void build_queries_kernel(int model, const float* __restrict__ ent, int64_t ld_ent,
                                     const float* __restrict__ rel, int64_t ld_rel, int k_int,
                                     const int32_t* __restrict__ test, int64_t n_q, int64_t n_rows, int side_mode,
                                     float* __restrict__ Q, int64_t ldq) {
    const bool cplx = (model == EMG_COMPLEX || model == EMG_HOLE);
    const int n = cplx ? k_int / 2 : k_int;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rows * n) return;
    const int64_t r = t / n;
    const int c = (int)(t - r * n);
    int64_t qi; bool obj;
    row_to_query(r, n_q, side_mode, &qi, &obj);
    const int32_t s = test[3 * qi + 0], p = test[3 * qi + 1], o = test[3 * qi + 2];
    const float* x = ent + (int64_t)(obj ? s : o) * ld_ent;  // the KEPT entity
    const float* pr = rel + (int64_t)p * ld_rel;
    float* q = Q + r * ldq;
    if (model == EMG_TRANSE_L1 || model == EMG_TRANSE_L2 || model == EMG_TRANSE_P) {
        // object side: |(s+p) - e| ; subject side: |(e+p) - o| = |e - (o-p)|
        q[c] = obj ? __fadd_rn(x[c], pr[c]) : __fsub_rn(x[c], pr[c]);
    } else if (model == EMG_DISTMULT) {
        q[c] = __fmul_rn(pr[c], x[c]);
    } else {
        const float p_r = pr[c], p_i = pr[n + c], x_r = x[c], x_i = x[n + c];
        if (obj) {  // <[p_r s_r - p_i s_i | p_r s_i + p_i s_r], [e_r | e_i]>
            q[c] = __fmaf_rn(p_r, x_r, -__fmul_rn(p_i, x_i));
            q[n + c] = __fmaf_rn(p_r, x_i, __fmul_rn(p_i, x_r));
        } else {    // <[p_r o_r + p_i o_i | p_r o_i - p_i o_r], [e_r | e_i]>
            q[c] = __fmaf_rn(p_r, x_r, __fmul_rn(p_i, x_i));
            q[n + c] = __fmaf_rn(p_r, x_i, -__fmul_rn(p_i, x_r));
        }
    }
}

// canonical chain of one (query row, entity row) pair
__device__ __forceinline__ float chain_score(int model, const float* __restrict__ q, const float* __restrict__ e, int k_int,
                                             float scale) {
    float acc = 0.f;
    if (model == EMG_TRANSE_L1) {
        for (int k = 0; k < k_int; ++k) acc = __fadd_rn(acc, fabsf(__fsub_rn(q[k], e[k])));
        return -acc;
    }
    if (model == EMG_TRANSE_L2) {
        for (int k = 0; k < k_int; ++k) {
            const float d = __fsub_rn(q[k], e[k]);
            acc = __fmaf_rn(d, d, acc);
        }
        return -sqrtf(acc);
    }
    if (model == EMG_TRANSE_P) {   // any positive order (scale = ord): -(sum |d|^ord)^(1/ord); ord = inf: -max |d|
        if (isinf(scale)) {
            for (int k = 0; k < k_int; ++k) acc = fmaxf(acc, fabsf(__fsub_rn(q[k], e[k])));
            return -acc;
        }
        for (int k = 0; k < k_int; ++k) acc = __fadd_rn(acc, powf(fabsf(__fsub_rn(q[k], e[k])), scale));
        return -powf(acc, 1.0f / scale);
    }
    for (int k = 0; k < k_int; ++k) acc = __fmaf_rn(q[k], e[k], acc);
    return model == EMG_HOLE ? __fmul_rn(acc, scale) : acc;
}

__global__ void pos_int_kernel(int model, const float* __restrict__ ent, int64_t ld_ent, int k_int, float scale,
                               const int32_t* __restrict__ test, int64_t n_q, int64_t n_rows, int side_mode,
                               const float* __restrict__ Q, int64_t ldq, int32_t* __restrict__ pos_int) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    int64_t qi; bool obj;
    row_to_query(r, n_q, side_mode, &qi, &obj);
    const int32_t tgt = obj ? test[3 * qi + 2] : test[3 * qi + 0];  // the TRUE entity on the corrupted side
    pos_int[r] = cmp_int(chain_score(model, Q + r * ldq, ent + (int64_t)tgt * ld_ent, k_int, scale));
}

// ---------------------------------------------------------------------------------------------
// f32 MFMA count kernel (DistMult / ComplEx / HolE)
// ---------------------------------------------------------------------------------------------
struct CountParams {
    const float* Q; int64_t ldq; const int32_t* pos_int; int64_t n_rows;
    const float* ent; int64_t n_cand; int64_t ld_ent; const int32_t* cand;
    int32_t k_int; float scale; int32_t model;
    int32_t* cnt_gt; int32_t* cnt_eq;
    float* S; int64_t lds;
    int64_t n_qb; int64_t n_cb; int64_t n_tiles; int32_t tiles_per_chunk;
};

constexpr int BM = 128, BN = 128, BK = 16, LDT = BK + 1;

template <bool VEC>
__device__ __forceinline__ void load_frag4(float (&v)[4], const float* __restrict__ row, bool row_ok, int kbase, int k_int) {
    if constexpr (VEC) {
        if (row_ok && kbase < k_int) {
            const float4 t = *reinterpret_cast<const float4*>(row + kbase);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
            v[0] = v[1] = v[2] = v[3] = 0.f;
        }
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = (row_ok && kbase + c < k_int) ? row[kbase + c] : 0.f;
    }
}

template <bool VEC, bool DENSE>
__global__ __launch_bounds__(256) void count_mfma_kernel(const CountParams P) {
    __shared__ float As[BM * LDT];
    __shared__ float Bs[BN * LDT];
    __shared__ int pos_s[BM];

    // XCD-aware decode: blocks on one XCD (id % 8) walk the query tiles of the SAME entity chunk,
    // so an entity tile is fetched from HBM once per XCD and re-read from that XCD's L2.
    const int64_t id = blockIdx.x;
    const int64_t xcd = id & 7, slot = id >> 3;
    const int64_t qb = slot % P.n_qb;
    const int64_t cb = xcd + 8 * (slot / P.n_qb);
    if (cb >= P.n_cb) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int lrow = tid >> 2, kq = tid & 3;  // loader: rows lrow, lrow+64 ; floats [4kq,4kq+4)
    const int l31 = lane & 31, lhi = lane >> 5;

    if (tid < BM) {
        const int64_t qr = qb * BM + tid;
        pos_s[tid] = (!DENSE && qr < P.n_rows) ? P.pos_int[qr] : 0x7fffffff;
    }

    const float* arow[2];
    bool aok[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int64_t qr = qb * BM + lrow + 64 * r;
        aok[r] = qr < P.n_rows;
        arow[r] = P.Q + (aok[r] ? qr : 0) * P.ldq;
    }

    unsigned cnt[2][16];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) cnt[a][r] = 0u;

    const int64_t tile0 = cb * P.tiles_per_chunk;
    const int64_t tile1 = min(tile0 + (int64_t)P.tiles_per_chunk, P.n_tiles);
    for (int64_t tile = tile0; tile < tile1; ++tile) {
        const float* brow[2];
        bool bok[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int64_t el = tile * BN + lrow + 64 * r;
            bok[r] = el < P.n_cand;
            const int64_t erow = bok[r] ? (P.cand ? (int64_t)P.cand[el] : el) : 0;
            brow[r] = P.ent + erow * P.ld_ent;
        }
        float16v acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

        for (int k0 = 0; k0 < P.k_int; k0 += BK) {
            float av[2][4], bv[2][4];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                load_frag4<VEC>(av[r], arow[r], aok[r], k0 + 4 * kq, P.k_int);
                load_frag4<VEC>(bv[r], brow[r], bok[r], k0 + 4 * kq, P.k_int);
            }
            __syncthreads();  // previous k-step's LDS reads done
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    As[(lrow + 64 * r) * LDT + 4 * kq + c] = av[r][c];
                    Bs[(lrow + 64 * r) * LDT + 4 * kq + c] = bv[r][c];
                }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                const int k = 2 * kk + lhi;  // A[i][k=lane>>5], B[k=lane>>5][j]
                float a[2], b[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    a[t] = As[(wr * 64 + t * 32 + l31) * LDT + k];
                    b[t] = Bs[(wc * 64 + t * 32 + l31) * LDT + k];
                }
#pragma unroll
                for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                    for (int tb = 0; tb < 2; ++tb)
                        acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
            }
        }
        // epilogue: D[row][col]: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                const int64_t ecol = tile * BN + wc * 64 + tb * 32 + l31;
                const bool cok = ecol < P.n_cand;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = wr * 64 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    float v = acc[ta][tb][r];
                    if (P.model == EMG_HOLE) v = __fmul_rn(v, P.scale);
                    if constexpr (DENSE) {
                        const int64_t qr = qb * BM + rl;
                        if (cok && qr < P.n_rows) P.S[qr * P.lds + ecol] = v;
                    } else {
                        const int ci = cmp_int(v);
                        const int p = pos_s[rl];
                        cnt[ta][r] += (unsigned)(cok && ci > p) + ((unsigned)(cok && ci == p) << 16);
                    }
                }
            }
    }
    if constexpr (!DENSE) {
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                unsigned c = cnt[ta][r];
#pragma unroll
                for (int off = 16; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
                if (l31 == 0) {
                    const int rl = wr * 64 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    const int64_t qr = qb * BM + rl;
                    if (qr < P.n_rows) {
                        if (c & 0xffffu) atomicAdd(&P.cnt_gt[qr], (int)(c & 0xffffu));
                        if (c >> 16) atomicAdd(&P.cnt_eq[qr], (int)(c >> 16));
                    }
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// TransE count kernel (f32 VALU, LDS-tiled): 64 queries x 64 entities per block, 4x4 per thread
// ---------------------------------------------------------------------------------------------
constexpr int TQ = 64, TE = 64, TK = 32;

template <bool L2, bool DENSE>
__global__ __launch_bounds__(256) void count_transe_kernel(const CountParams P) {
    __shared__ __attribute__((aligned(16))) float Qs[TK * TQ];
    __shared__ __attribute__((aligned(16))) float Es[TK * TE];
    __shared__ int pos_s[TQ];
    __shared__ unsigned cnt_s[TQ];

    const int64_t id = blockIdx.x;
    const int64_t xcd = id & 7, slot = id >> 3;
    const int64_t qb = slot % P.n_qb;
    const int64_t cb = xcd + 8 * (slot / P.n_qb);
    if (cb >= P.n_cb) return;

    const int tid = threadIdx.x;
    const int tq = tid & 15, te = tid >> 4;
    const int lrow = tid & 63, lkq = tid >> 6;  // loader: row lrow, float4 slots lkq and lkq+4 of the k-tile

    if (tid < TQ) {
        const int64_t qr = qb * TQ + tid;
        pos_s[tid] = (!DENSE && qr < P.n_rows) ? P.pos_int[qr] : 0x7fffffff;
        cnt_s[tid] = 0u;
    }
    const int64_t qrow_g = qb * TQ + lrow;
    const bool qok = qrow_g < P.n_rows;
    const float* qptr = P.Q + (qok ? qrow_g : 0) * P.ldq;

    unsigned cnt[4] = {0u, 0u, 0u, 0u};
    const int64_t tile0 = cb * P.tiles_per_chunk;
    const int64_t tile1 = min(tile0 + (int64_t)P.tiles_per_chunk, P.n_tiles);
    for (int64_t tile = tile0; tile < tile1; ++tile) {
        const int64_t el = tile * TE + lrow;
        const bool eok = el < P.n_cand;
        const float* eptr = P.ent + (eok ? (P.cand ? (int64_t)P.cand[el] : el) : 0) * P.ld_ent;
        float acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
        for (int k0 = 0; k0 < P.k_int; k0 += TK) {
            float qv[2][4], ev[2][4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int kb = k0 + 4 * (lkq + 4 * h);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    qv[h][c] = (qok && kb + c < P.k_int) ? qptr[kb + c] : 0.f;
                    ev[h][c] = (eok && kb + c < P.k_int) ? eptr[kb + c] : 0.f;
                }
            }
            __syncthreads();
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int kl = 4 * (lkq + 4 * h) + c;
                    Qs[kl * TQ + lrow] = qv[h][c];
                    Es[kl * TE + lrow] = ev[h][c];
                }
            __syncthreads();
#pragma unroll 8
            for (int k = 0; k < TK; ++k) {
                const float4 q4 = *reinterpret_cast<const float4*>(&Qs[k * TQ + 4 * tq]);
                const float4 e4 = *reinterpret_cast<const float4*>(&Es[k * TE + 4 * te]);
                const float q[4] = {q4.x, q4.y, q4.z, q4.w};
                const float e[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const float d = __fsub_rn(q[a], e[b]);
                        if constexpr (L2) acc[a][b] = __fmaf_rn(d, d, acc[a][b]);
                        else acc[a][b] = __fadd_rn(acc[a][b], fabsf(d));
                    }
            }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int ql = 4 * tq + a;
            const int p = pos_s[ql];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int64_t ecol = tile * TE + 4 * te + b;
                const bool cok = ecol < P.n_cand;
                const float v = L2 ? -sqrtf(acc[a][b]) : -acc[a][b];
                if constexpr (DENSE) {
                    const int64_t qr = qb * TQ + ql;
                    if (cok && qr < P.n_rows) P.S[qr * P.lds + ecol] = v;
                } else {
                    const int ci = cmp_int(v);
                    cnt[a] += (unsigned)(cok && ci > p) + ((unsigned)(cok && ci == p) << 16);
                }
            }
        }
    }
    if constexpr (!DENSE) {
        // per thread <= 4*tiles_per_chunk per field; 16 threads share a query row
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if (cnt[a]) atomicAdd(&cnt_s[4 * tq + a], cnt[a]);
        __syncthreads();
        if (tid < TQ) {
            const int64_t qr = qb * TQ + tid;
            const unsigned c = cnt_s[tid];
            if (qr < P.n_rows) {
                if (c & 0xffffu) atomicAdd(&P.cnt_gt[qr], (int)(c & 0xffffu));
                if (c >> 16) atomicAdd(&P.cnt_eq[qr], (int)(c >> 16));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// TransE count kernel, large form: 128 queries x 128 entities per block, 8 x 8 per thread, 16-wide k tiles double-
// buffered in LDS with the next tile's 16-byte global loads in flight under the arithmetic.  Per k-step a thread
// reads 4 x 16 bytes from LDS for 64 (query, entity) pairs x 2 VALU instructions (the 4 x 4 form above: 2 reads per
// 16 pairs, single-buffered, scalar global loads: 33 % of the VALU peak).  Every pair still accumulates |q_k - e_k|
// (or (q_k - e_k)^2) in ascending k in one register: same bits.  Needs 16-byte aligned rows.
// ---------------------------------------------------------------------------------------------
constexpr int UQ = 128, UE = 128, UK = 16;

template <bool L2>
__global__ __launch_bounds__(256) void count_transe_big_kernel(const CountParams P) {
    __shared__ __attribute__((aligned(16))) float Qs[2][UK * UQ];
    __shared__ __attribute__((aligned(16))) float Es[2][UK * UE];
    __shared__ int pos_s[UQ];
    __shared__ unsigned cnt_s[UQ];

    const int64_t id = blockIdx.x;
    const int64_t xcd = id & 7, slot = id >> 3;
    const int64_t qb = slot % P.n_qb;
    const int64_t cb = xcd + 8 * (slot / P.n_qb);
    if (cb >= P.n_cb) return;

    const int tid = threadIdx.x;
    const int tq = tid & 15, te = tid >> 4;
    const int lrow = tid & 127, lk = tid >> 7;   // loader: row lrow, 16-byte slots lk and lk + 2 of the 16-wide k tile

    if (tid < UQ) {
        const int64_t qr = qb * UQ + tid;
        pos_s[tid] = qr < P.n_rows ? P.pos_int[qr] : 0x7fffffff;
        cnt_s[tid] = 0u;
    }
    const int64_t qrow_g = min(qb * UQ + lrow, P.n_rows - 1);   // clamped: rows past the end count nothing (pos = INT_MAX)
    const float* qptr = P.Q + qrow_g * P.ldq;
    const int nkt = (P.k_int + UK - 1) / UK;

    unsigned cnt[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) cnt[a] = 0u;
    const int64_t tile0 = cb * P.tiles_per_chunk;
    const int64_t tile1 = min(tile0 + (int64_t)P.tiles_per_chunk, P.n_tiles);
    for (int64_t tile = tile0; tile < tile1; ++tile) {
        const int64_t el = min(tile * UE + lrow, P.n_cand - 1);
        const float* eptr = P.ent + (P.cand ? (int64_t)P.cand[el] : el) * P.ld_ent;
        float4 gq[2], ge[2];
        auto fetch = [&](int kt) {   // k-tile kt of this thread's query row and entity row -> registers (zeros past k_int)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int kb = kt * UK + 4 * (lk + 2 * h);
                gq[h] = ge[h] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kb + 4 <= P.k_int) {
                    gq[h] = *reinterpret_cast<const float4*>(qptr + kb);
                    ge[h] = *reinterpret_cast<const float4*>(eptr + kb);
                } else if (kb < P.k_int) {
                    float tq4[4] = {0.f, 0.f, 0.f, 0.f}, te4[4] = {0.f, 0.f, 0.f, 0.f};
                    for (int c = 0; c < 4; ++c)
                        if (kb + c < P.k_int) { tq4[c] = qptr[kb + c]; te4[c] = eptr[kb + c]; }
                    gq[h] = make_float4(tq4[0], tq4[1], tq4[2], tq4[3]);
                    ge[h] = make_float4(te4[0], te4[1], te4[2], te4[3]);
                }
            }
        };
        auto stage = [&](int buf) {   // registers -> LDS, k-major
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int kl = 4 * (lk + 2 * h);
                Qs[buf][(kl + 0) * UQ + lrow] = gq[h].x; Qs[buf][(kl + 1) * UQ + lrow] = gq[h].y;
                Qs[buf][(kl + 2) * UQ + lrow] = gq[h].z; Qs[buf][(kl + 3) * UQ + lrow] = gq[h].w;
                Es[buf][(kl + 0) * UE + lrow] = ge[h].x; Es[buf][(kl + 1) * UE + lrow] = ge[h].y;
                Es[buf][(kl + 2) * UE + lrow] = ge[h].z; Es[buf][(kl + 3) * UE + lrow] = ge[h].w;
            }
        };
        float acc[8][8];
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[a][b] = 0.f;
        fetch(0);
        __syncthreads();       // (the previous tile's readers are done with buffer 0)
        stage(0);
        for (int kt = 0; kt < nkt; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nkt) fetch(kt + 1);   // in flight under the arithmetic below
            __syncthreads();                   // buffer `buf` is staged; buffer 1 - buf is free again
#pragma unroll 4
            for (int k = 0; k < UK; ++k) {
                const float4 q0 = *reinterpret_cast<const float4*>(&Qs[buf][k * UQ + 4 * tq]);
                const float4 q1 = *reinterpret_cast<const float4*>(&Qs[buf][k * UQ + 64 + 4 * tq]);
                const float4 e0 = *reinterpret_cast<const float4*>(&Es[buf][k * UE + 4 * te]);
                const float4 e1 = *reinterpret_cast<const float4*>(&Es[buf][k * UE + 64 + 4 * te]);
                const float q[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                const float e[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
#pragma unroll
                for (int a = 0; a < 8; ++a)
#pragma unroll
                    for (int b = 0; b < 8; ++b) {
                        const float d = q[a] - e[b];   // (nothing here can contract: one rounding per operation, as __f*_rn)
                        if constexpr (L2) acc[a][b] = __builtin_fmaf(d, d, acc[a][b]);
                        else asm("v_add_f32 %0, %0, |%1|" : "+v"(acc[a][b]) : "v"(d));   // |d| as a source modifier of the add: hipcc emits
                                                                                             // v_and + v_pk_add otherwise (2.5 VALU per pair)
                    }
            }
            if (kt + 1 < nkt) stage(1 - buf);
        }
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const int ql = (a < 4 ? 0 : 64) + 4 * tq + (a & 3);
            const int p = pos_s[ql];
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int64_t ecol = tile * UE + (b < 4 ? 0 : 64) + 4 * te + (b & 3);
                const bool cok = ecol < P.n_cand;
                const float v = L2 ? -sqrtf(acc[a][b]) : -acc[a][b];
                const int ci = cmp_int(v);
                cnt[a] += (unsigned)(cok && ci > p) + ((unsigned)(cok && ci == p) << 16);
            }
        }
    }
    // per thread <= 8 * tiles_per_chunk per field; 16 threads share a query row
#pragma unroll
    for (int a = 0; a < 8; ++a)
        if (cnt[a]) atomicAdd(&cnt_s[(a < 4 ? 0 : 64) + 4 * tq + (a & 3)], cnt[a]);
    __syncthreads();
    if (tid < UQ) {
        const int64_t qr = qb * UQ + tid;
        const unsigned c = cnt_s[tid];
        if (qr < P.n_rows) {
            if (c & 0xffffu) atomicAdd(&P.cnt_gt[qr], (int)(c & 0xffffu));
            if (c >> 16) atomicAdd(&P.cnt_eq[qr], (int)(c >> 16));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// filter counts: one wave per query row, one lane per filter entry, canonical chain
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void filter_count_kernel(int model, const float* __restrict__ Q, int64_t ldq,
                                                           const int32_t* __restrict__ pos_int, int64_t n_rows,
                                                           const float* __restrict__ ent, int64_t n_local,
                                                           int64_t ld_ent, int64_t ent_offset, int k_int, float scale,
                                                           const int64_t* __restrict__ fptr,
                                                           const int32_t* __restrict__ fidx,
                                                           int32_t* __restrict__ fgt, int32_t* __restrict__ feq) {
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= n_rows) return;
    const int p = pos_int[r];
    const float* q = Q + r * ldq;
    int gt = 0, eq = 0;
    for (int64_t u = fptr[r] + lane; u < fptr[r + 1]; u += 64) {
        const int64_t e = (int64_t)fidx[u] - ent_offset;
        if (e < 0 || e >= n_local) continue;
        const int ci = cmp_int(chain_score(model, q, ent + e * ld_ent, k_int, scale));
        gt += ci > p;
        eq += ci == p;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        gt += __shfl_xor(gt, off, 64);
        eq += __shfl_xor(eq, off, 64);
    }
    if (lane == 0) {
        if (gt) atomicAdd(&fgt[r], gt);
        if (eq) atomicAdd(&feq[r], eq);
    }
}

// ---------------------------------------------------------------------------------------------
// exact re-scoring of the (row, entity) pairs the bf16 prefilter could not decide (emg_rank_bf16.hip, MODE 2):
// one thread per pair, the canonical chain and comparison of the parity path, counters updated atomically
// ---------------------------------------------------------------------------------------------
struct RescoreParams {
    int model; const float* Q; int64_t ldq; const int32_t* pos_int; const float* ent; int64_t ld_ent; int64_t ent_offset;
    int k_int; float scale; const uint64_t* pairs; uint32_t cap; const uint32_t* seg_count; uint32_t n_seg;  // segment s: pairs[s*cap ..+min(count, cap))
    uint32_t groups_per_block;   // 4-segment groups per workgroup of the prefilter that wrote the pairs (see the kernel)
    uint32_t min_pairs, max_pairs;   // this launch takes the segments with min_pairs <= pairs < max_pairs
    int32_t* cnt_gt; int32_t* cnt_eq;
};

__device__ __forceinline__ void rescore_finish(const RescoreParams& P, int64_t row, float acc) {
    const float score = P.model == EMG_HOLE ? __fmul_rn(acc, P.scale)
                      : (P.model == EMG_TRANSE_L1 ? -acc : (P.model == EMG_TRANSE_L2 ? -sqrtf(acc) : acc));
    const int ci = cmp_int(score), p = P.pos_int[row];
    if (ci > p) atomicAdd(&P.cnt_gt[row], 1);
    else if (ci == p) atomicAdd(&P.cnt_eq[row], 1);
}

// One WAVE per segment of the pair buffer (= the pairs one wave of the prefilter emitted: a handful of query rows
// against the entities of a few tiles), 64 pairs at a time.  16-byte-aligned rows (VEC): 16-float slices of the 64
// query and entity rows are staged through the wave's own LDS region with COALESCED loads (4 lanes x 16 B per row
// slice: 16 pairs per load instruction, the next slice in flight under the current slice's chain), then every lane
// runs the canonical chain over ITS pair's slice.  One thread per pair reading global memory directly issues 64
// scattered 16-byte requests per load instruction (measured 0.4-0.8 G pairs/s against 3 G pairs/s staged).
constexpr int RS_KC = 16, RS_LD = RS_KC + 4;

__device__ __forceinline__ void wave_lds_sync() {   // this wave's LDS writes are visible to its own later reads
    __builtin_amdgcn_s_waitcnt(0xc07f);             // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// KIND: the chain step of chain_score — 0: fmaf(q, e, acc); 1: acc + |q - e|; 2: fmaf(d, d, acc), d = q - e
template <int KIND>
__device__ __forceinline__ float chain_step(float q, float e, float acc) {
    if constexpr (KIND == 0) return __fmaf_rn(q, e, acc);
    else if constexpr (KIND == 1) return __fadd_rn(acc, fabsf(__fsub_rn(q, e)));
    else { const float d = __fsub_rn(q, e); return __fmaf_rn(d, d, acc); }
}

template <bool VEC, int KIND>
__global__ __launch_bounds__(256) void rescore_pairs_kernel(const RescoreParams P) {
    __shared__ __attribute__((aligned(16))) float qs[4][64 * RS_LD];
    __shared__ __attribute__((aligned(16))) float es[4][64 * RS_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* myq = qs[wave];
    float* mye = es[wave];
    const int sub = lane >> 2, part = lane & 3;   // loader role: pair (16 it + sub) of the wave's 64, 16-byte piece `part`
    // Workgroup b of the prefilter ran on XCD b & 7 and walked the query blocks of one entity chunk after another; its
    // segments are re-scored on the SAME XCD in the same order, so the chunk's entity rows (3-6 MB) are still in that XCD's
    // L2 when the next query block's pairs ask for them (a row is wanted by ~10 query rows of a 2048-row tile).
    const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3, r = P.groups_per_block;
    const uint32_t group = r * (8u * (j / r) + xcd) + (j % r);
    const uint32_t seg = group * 4u + wave;   // one wave per segment (the grid is padded: groups past the end do nothing)
    if (seg < P.n_seg) {
        uint32_t n = min(P.seg_count[seg], P.cap);
        if (n < P.min_pairs || n >= P.max_pairs) n = 0u;   // the other launch's segment
        const uint64_t* sp = P.pairs + (uint64_t)seg * P.cap;
        for (uint32_t c0 = 0; c0 < n; c0 += 64u) {
            const bool live = c0 + lane < n;
            int64_t row = 0, e = 0;
            if (live) {
                const uint64_t pr = sp[c0 + lane];
                row = (int64_t)(pr >> 32);
                e = (int64_t)(uint32_t)pr - P.ent_offset;
            }
            float acc = 0.f;
            if constexpr (!VEC) {
                const float* q = P.Q + row * P.ldq;
                const float* er = P.ent + e * P.ld_ent;
                for (int k = 0; k < P.k_int; ++k) acc = chain_step<KIND>(q[k], er[k], acc);
            } else {
                const float* qp[4];
                const float* ep[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    qp[it] = P.Q + __shfl(row, 16 * it + sub, 64) * P.ldq + 4 * part;
                    ep[it] = P.ent + __shfl(e, 16 * it + sub, 64) * P.ld_ent + 4 * part;
                }
                float4 qv[4], ev[4];
                auto fetch = [&](int k0) {   // k_int % 4 == 0: a 4-float piece is whole or absent
                    const bool pin = k0 + 4 * part < P.k_int;
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        qv[it] = pin ? *reinterpret_cast<const float4*>(qp[it] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
                        ev[it] = pin ? *reinterpret_cast<const float4*>(ep[it] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                };
                fetch(0);
                for (int k0 = 0; k0 < P.k_int; k0 += RS_KC) {
                    wave_lds_sync();   // the previous slice has been consumed by every lane
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        *reinterpret_cast<float4*>(myq + (16 * it + sub) * RS_LD + 4 * part) = qv[it];
                        *reinterpret_cast<float4*>(mye + (16 * it + sub) * RS_LD + 4 * part) = ev[it];
                    }
                    wave_lds_sync();
                    if (k0 + RS_KC < P.k_int) fetch(k0 + RS_KC);   // the next slice flies while this one is multiplied
                    const int kn = min(RS_KC, P.k_int - k0);
#pragma unroll
                    for (int c = 0; c < RS_KC / 4; ++c) {
                        if (4 * c < kn) {
                            const float4 a = *reinterpret_cast<const float4*>(myq + lane * RS_LD + 4 * c);
                            const float4 b2 = *reinterpret_cast<const float4*>(mye + lane * RS_LD + 4 * c);
                            acc = chain_step<KIND>(a.x, b2.x, acc); acc = chain_step<KIND>(a.y, b2.y, acc);
                            acc = chain_step<KIND>(a.z, b2.z, acc); acc = chain_step<KIND>(a.w, b2.w, acc);
                        }
                    }
                }
            }
            if (live) rescore_finish(P, row, acc);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Re-scoring of the f16 prefilter's segments, QUERY ROWS IN LDS.  A segment holds the undecided pairs of ONE wave of the
// prefilter: 32 query rows against the entities of one chunk, ~1000 pairs when the positives rank in the middle of the
// table.  Re-read per pair, the two rows are 2 x 4 k_int bytes of L2 traffic each time (PMC at C4 size: 2.0 G L2
// requests of which 44 % missed — the query rows of the ~500 segments in flight on an XCD alone are 26 MB, six times its
// L2).  Here one WORKGROUP takes one segment: its query rows are read once into LDS (32 x (k_int + 4) floats), the four
// waves share the segment's pairs 64 at a time, and only entity rows still travel — 16-float slices, coalesced, staged
// through the wave's own LDS region as in rescore_pairs_kernel, two slices ahead of the chain.  The prefilter walks the
// tiles of its chunk in a per-block ROTATED order; every workgroup starts at the wrap point of its segment instead, so
// the workgroups an XCD runs side by side (same chunk, neighbouring query blocks) sweep the chunk's entity rows in the
// same direction at the same time and meet them in that XCD's L2.  Counters: per-row LDS atomics, one global atomic per
// row and workgroup.  Segments with few pairs, or whose rows do not fit the LDS image, take the per-pair form.
// ---------------------------------------------------------------------------------------------
#ifndef RQ_ABLATE
#define RQ_ABLATE 0   // timing experiments only (wrong results): 1 no entity loads, 2 no LDS staging / chain
#endif
#ifndef RQ_NPF
#define RQ_NPF 2
#endif
constexpr int RQ_ROWS = 32;        // query rows of a segment (one wave of count_mfma_bf16_v3_kernel)
constexpr int RQ_MIN_PAIRS = 512;  // shorter segments go to rescore_pairs_kernel: a workgroup's eight waves need a batch of 64 each
constexpr int RQ_KC = 32, RQ_LD = RQ_KC + 4;   // 32-float slices: every request a whole 128-byte line (16-float slices: PMC 77 B / request)

// RQ_WAVES = 8 up to ~570 columns; 4 where the image of wider rows leaves room for four staging regions only (<= 830)
static inline size_t rescore_segment_lds(int k_int, int waves) { return ((size_t)RQ_ROWS * (k_int + 4) + (size_t)waves * 64 * RQ_LD) * sizeof(float); }

// Asynchronous 16-byte row loads with a hand-counted wait (as emg_score_kernels.hpp's rolling window): compiler-visible loads under
// the slice loop's conditions made every wait s_waitcnt vmcnt(0) — the ISA of round 4's kernels: ONE slice in flight per wave, each
// slice a full L2 round trip (1.4 us at C4's size: 13 slices x 221 batches per wave = the kernel's 4 ms).  Here the loads are
// inline assembly, unconditional (past the row's end: its last piece again), every consumed slice issues exactly one refill, and
// the wait before slice j is vmcnt(8 (NPF - 1)): only the NPF - 1 younger slices may still be in flight.
typedef float rq_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rq_f4 rq_load16(const float* p) {
    rq_f4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int N> __device__ __forceinline__ void rq_wait(rq_f4 (&a)[8]) {
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "n"(N) : "memory");
}

template <int KIND, int RQ_WAVES>
__global__ __launch_bounds__(64 * RQ_WAVES) void rescore_segment_kernel(const RescoreParams P) {
    constexpr int RQ_THREADS = 64 * RQ_WAVES;
    extern __shared__ __attribute__((aligned(16))) float rq_lds[];
    __shared__ int s_rmin, s_rmax, s_wrap, s_gt[RQ_ROWS], s_eq[RQ_ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ldq_s = P.k_int + 4;
    float* const qrows = rq_lds;
    float* const mye = rq_lds + RQ_ROWS * ldq_s + wave * (64 * RQ_LD);
    // XCD b & 7 re-scores the segments of the prefilter workgroups that ran there, in their order (see rescore_pairs_kernel)
    const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3, spb = P.groups_per_block * 4u;
    const uint32_t seg = (xcd + 8u * (j / spb)) * spb + (j % spb);
    if (seg >= P.n_seg) return;
    const uint32_t n = min(P.seg_count[seg], P.cap);
    if (n < P.min_pairs || n >= P.max_pairs) return;   // (short segments: rescore_pairs_kernel, a wave each)
    const uint64_t* sp = P.pairs + (uint64_t)seg * P.cap;

    // ---- one pass over the pairs: the span of query rows, and where the tile order wraps ----------------------
    if (tid == 0) { s_rmin = INT_MAX; s_rmax = -1; s_wrap = INT_MAX; }
    if (tid < RQ_ROWS) { s_gt[tid] = 0; s_eq[tid] = 0; }
    __syncthreads();
    {
        int rmin = INT_MAX, rmax = -1, wrap = INT_MAX;
        for (uint32_t i = tid; i < n; i += RQ_THREADS) {
            const uint64_t pr = sp[i];
            const int row = (int)(pr >> 32);
            rmin = min(rmin, row); rmax = max(rmax, row);
            if (i > 0 && (((uint32_t)pr - (uint32_t)P.ent_offset) >> 7) < (((uint32_t)sp[i - 1] - (uint32_t)P.ent_offset) >> 7))
                wrap = min(wrap, (int)i);   // the entity tile went down
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            rmin = min(rmin, __shfl_xor(rmin, off, 64)); rmax = max(rmax, __shfl_xor(rmax, off, 64));
            wrap = min(wrap, __shfl_xor(wrap, off, 64));
        }
        if (lane == 0) { atomicMin(&s_rmin, rmin); atomicMax(&s_rmax, rmax); atomicMin(&s_wrap, wrap); }
    }
    __syncthreads();
    const int rmin = s_rmin, nrows = s_rmax - rmin + 1;
    const uint32_t start = s_wrap == INT_MAX ? 0u : (uint32_t)s_wrap;
    const bool staged = nrows <= RQ_ROWS;   // workgroup-uniform
    if (staged) {
        const int pieces = P.k_int >> 2;   // 16-byte pieces per row
        for (int t = tid; t < nrows * pieces; t += RQ_THREADS) {
            const int r = t / pieces, c = t - r * pieces;
            *reinterpret_cast<float4*>(qrows + r * ldq_s + 4 * c) = *reinterpret_cast<const float4*>(P.Q + (int64_t)(rmin + r) * P.ldq + 4 * c);
        }
    }
    __syncthreads();

    const int sub = lane >> 3, part = lane & 7;   // loader role: pair (8 it + sub) of the wave's 64, 16-byte piece `part` of a slice
    constexpr int NPF = RQ_NPF;                   // entity slices in flight per wave beside the one being multiplied
    for (uint32_t c0 = (uint32_t)wave * 64u; c0 < n; c0 += RQ_THREADS) {
        const bool live = c0 + lane < n;
        int64_t row = rmin, e = 0;
        if (live) {
            uint32_t at = start + c0 + lane;
            if (at >= n) at -= n;
            const uint64_t pr = sp[at];
            row = (int64_t)(pr >> 32);
            e = (int64_t)(uint32_t)pr - P.ent_offset;
        }
        const float* ep[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) ep[it] = P.ent + __shfl(e, 8 * it + sub, 64) * P.ld_ent + 4 * part;
        const float* const ql = qrows + (int)(row - rmin) * ldq_s;   // staged: this pair's query row in the LDS image
        const float* const qg = P.Q + row * P.ldq;                   // otherwise (rows outside one 32-row span): read in place
        static_assert(8 * (NPF - 1) <= 63, "vmcnt is a 6-bit counter");
        rq_f4 ev[NPF][8];
        const int k_last = ((P.k_int - 1) / RQ_KC) * RQ_KC;   // start of the row's last slice
        auto fetch = [&](int k0, rq_f4 (&evs)[8]) {   // k_int % 4 == 0; past the row's end: its last piece again (never multiplied)
            const int kk = min(min(k0, k_last) + 4 * part, P.k_int - 4) - 4 * part;   // (the row pointers carry the lane's piece already)
#pragma unroll
            for (int it = 0; it < 8; ++it) evs[it] = rq_load16(ep[it] + kk);
        };
        float acc = 0.f;
        auto slice = [&](int k0, rq_f4 (&evs)[8]) {
            rq_wait<8 * (NPF - 1)>(evs);   // this slice has landed; the NPF - 1 younger ones fly on
            wave_lds_sync();   // the previous slice has been consumed by every lane
#pragma unroll
            for (int it = 0; it < 8; ++it) *reinterpret_cast<rq_f4*>(mye + (8 * it + sub) * RQ_LD + 4 * part) = evs[it];
            wave_lds_sync();
            fetch(k0 + NPF * RQ_KC, evs);   // the refill, unconditional: NPF slices ahead of the chain
            const int kn = min(RQ_KC, P.k_int - k0);
            // A whole slice (all but a row's last): straight-line code — its sixteen LDS reads are issued together and waited for
            // ONCE.  With the per-chunk guard below hipcc put a branch and an s_waitcnt lgkmcnt(0) in front of every four chain
            // steps: eight exposed LDS round trips per slice, which was the kernel's time (round 5: neither deeper prefetch nor
            // conflict-free reads nor L2-resident rows changed it)
            if (kn == RQ_KC && staged) {
                float4 a[RQ_KC / 4], b2[RQ_KC / 4];
#pragma unroll
                for (int c = 0; c < RQ_KC / 4; ++c) {
                    a[c] = *reinterpret_cast<const float4*>(ql + k0 + 4 * c);
                    b2[c] = *reinterpret_cast<const float4*>(mye + lane * RQ_LD + 4 * c);
                }
#pragma unroll
                for (int c = 0; c < RQ_KC / 4; ++c) {
                    acc = chain_step<KIND>(a[c].x, b2[c].x, acc); acc = chain_step<KIND>(a[c].y, b2[c].y, acc);
                    acc = chain_step<KIND>(a[c].z, b2[c].z, acc); acc = chain_step<KIND>(a[c].w, b2[c].w, acc);
                }
                return;
            }
#pragma unroll
            for (int c = 0; c < RQ_KC / 4; ++c) {
                if (4 * c < kn) {
                    const float4 a = staged ? *reinterpret_cast<const float4*>(ql + k0 + 4 * c) : *reinterpret_cast<const float4*>(qg + k0 + 4 * c);
                    const float4 b2 = *reinterpret_cast<const float4*>(mye + lane * RQ_LD + 4 * c);
                    acc = chain_step<KIND>(a.x, b2.x, acc); acc = chain_step<KIND>(a.y, b2.y, acc);
                    acc = chain_step<KIND>(a.z, b2.z, acc); acc = chain_step<KIND>(a.w, b2.w, acc);
                }
            }
        };
#pragma unroll
        for (int u = 0; u < NPF; ++u) fetch(u * RQ_KC, ev[u]);
        for (int k0 = 0; k0 < P.k_int; k0 += NPF * RQ_KC) {
#pragma unroll
            for (int u = 0; u < NPF; ++u)
                if (k0 + u * RQ_KC < P.k_int) slice(k0 + u * RQ_KC, ev[u]);
        }
#pragma unroll
        for (int u = 0; u < NPF; ++u) rq_wait<0>(ev[u]);   // refills never taken: their registers stay theirs until the loads have landed
        if (live) {
            if (!staged) rescore_finish(P, row, acc);
            else {
                const float score = P.model == EMG_HOLE ? __fmul_rn(acc, P.scale)
                                  : (P.model == EMG_TRANSE_L1 ? -acc : (P.model == EMG_TRANSE_L2 ? -sqrtf(acc) : acc));
                const int ci = cmp_int(score), p = P.pos_int[row];
                if (ci > p) atomicAdd(&s_gt[row - rmin], 1);
                else if (ci == p) atomicAdd(&s_eq[row - rmin], 1);
            }
        }
    }
    if (staged) {
        __syncthreads();
        if (tid < nrows) {
            if (s_gt[tid]) atomicAdd(&P.cnt_gt[rmin + tid], s_gt[tid]);
            if (s_eq[tid]) atomicAdd(&P.cnt_eq[rmin + tid], s_eq[tid]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// ENTITY-TILE-MAJOR re-scoring (round 5).  The segment form above keeps 32 query rows in LDS and streams an entity row per pair:
// 59 M pairs x 1600 B at C4's size, of which 22 % miss L2 (the table is 1.6 GB) — 20 GB from HBM per 8192 query rows.  Here the
// pairs are first bucketed by TILE of 32 entity rows (the prefilter's segments list them tile by tile, so runs of a tile take
// ONE atomic: histogram | scan | scatter, 0.5 GB read and written), then one workgroup per tile keeps the tile's 32 entity rows
// in LDS and streams the QUERY rows of its pairs — the query matrix of a call is 13 MB: every streamed byte comes from L2 / MALL.
// Same chain, same comparison; counters are integers, so the order of the additions is free.
// ---------------------------------------------------------------------------------------------
constexpr int RT_TILE_LOG = 5;   // 32 entity rows per tile = RQ_ROWS (the LDS image)
struct TileParams {
    RescoreParams R; int64_t n_local; uint32_t n_tiles;
    uint32_t *tcnt, *toff, *cursor; uint64_t* sorted; uint64_t sorted_cap;
};

// runs of equal tile among the 64 pairs a wave holds (a segment lists its pairs tile by tile): the run's first lane acts for it
__device__ __forceinline__ void tile_runs(bool live, uint32_t tile, int lane, bool* head, int* head_lane, uint32_t* run) {
    const uint32_t prev = __shfl_up(tile, 1, 64);
    const bool prev_live = __shfl_up((int)live, 1, 64) != 0;
    *head = live && (lane == 0 || !prev_live || tile != prev);
    const unsigned long long hm = __ballot(*head), lm = __ballot(live);
    const unsigned long long below = hm & ((2ull << lane) - 1ull);            // heads at or below this lane
    *head_lane = below ? 63 - __clzll((long long)below) : 0;
    const unsigned long long above = lane == 63 ? 0ull : (hm >> (lane + 1));
    // the run ends at the next head, or at the first dead lane above (live lanes need not be a prefix: a short last batch is)
    const unsigned long long dead_above = lane == 63 ? 0ull : ((~lm) >> (lane + 1));
    const unsigned long long stop = above | dead_above;
    *run = (uint32_t)(stop ? __ffsll((long long)stop) : 64 - lane);
}

template <bool SCATTER>
__global__ __launch_bounds__(256) void tile_sort_kernel(const TileParams T) {
    const RescoreParams& P = T.R;
    const int lane = threadIdx.x & 63;
    const uint32_t seg = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (seg >= P.n_seg) return;
    const uint32_t n = min(P.seg_count[seg], P.cap);
    const uint64_t* sp = P.pairs + (uint64_t)seg * P.cap;
    for (uint32_t c0 = 0; c0 < n; c0 += 64u) {
        const bool live = c0 + lane < n;
        const uint64_t pr = live ? sp[c0 + lane] : 0ull;
        const uint32_t el = (uint32_t)pr - (uint32_t)P.ent_offset;
        const uint32_t tile = live ? el >> RT_TILE_LOG : 0xffffffffu;
        bool head; int head_lane; uint32_t run;
        tile_runs(live, tile, lane, &head, &head_lane, &run);
        if constexpr (!SCATTER) {
            if (head) atomicAdd(T.tcnt + tile, run);
        } else {
            uint32_t base = 0u;
            if (head) base = atomicAdd(T.cursor + tile, run);
            base = __shfl(base, head_lane, 64);
            const uint64_t at = (uint64_t)base + (uint32_t)(lane - head_lane);
            if (live && at < T.sorted_cap) T.sorted[at] = ((uint64_t)el << 32) | (pr >> 32);   // (local entity row, query row)
        }
    }
}

// exclusive offsets of the tiles' stretches (one workgroup: the tile counts of 1M entities are 31 k words)
__global__ __launch_bounds__(1024) void tile_scan_kernel(const TileParams T) {
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0u;
    __syncthreads();
    for (uint32_t base = 0; base < T.n_tiles; base += 1024u * 8u) {
        uint32_t c[8], sum = 0u;
        const uint32_t i0 = base + threadIdx.x * 8u;
#pragma unroll
        for (int j = 0; j < 8; ++j) { c[j] = i0 + j < T.n_tiles ? T.tcnt[i0 + j] : 0u; sum += c[j]; }
        uint32_t inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        uint32_t pre = s_carry, tot = 0u;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const uint32_t v = s_w[w]; if (w < wv) pre += v; tot += v; }
        uint32_t run = pre + inc - sum;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (i0 + j < T.n_tiles) { T.toff[i0 + j] = run; T.cursor[i0 + j] = run; T.tcnt[i0 + j] = 0u; }   // (the counts return to zero)
            run += c[j];
        }
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) T.toff[T.n_tiles] = s_carry;
}

template <int KIND, int RQ_WAVES>
__global__ __launch_bounds__(64 * RQ_WAVES) void rescore_tile_kernel(const TileParams T) {
    constexpr int RQ_THREADS = 64 * RQ_WAVES;
    const RescoreParams& P = T.R;
    extern __shared__ __attribute__((aligned(16))) float rq_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lde_s = P.k_int + 4;
    float* const erows = rq_lds;
    float* const myq = rq_lds + RQ_ROWS * lde_s + wave * (64 * RQ_LD);
    const uint32_t tile = blockIdx.x;
    const uint64_t p0 = T.toff[tile], p1 = min((uint64_t)T.toff[tile + 1], T.sorted_cap);
    if (p1 <= p0) return;
    const uint32_t n = (uint32_t)(p1 - p0);
    const uint64_t* sp = T.sorted + p0;
    const int64_t e_first = (int64_t)tile << RT_TILE_LOG;
    const int nrows = (int)min((int64_t)RQ_ROWS, T.n_local - e_first);
    {   // the tile's entity rows: read once
        const int pieces = P.k_int >> 2;
        for (int t = tid; t < nrows * pieces; t += RQ_THREADS) {
            const int r = t / pieces, c = t - r * pieces;
            *reinterpret_cast<float4*>(erows + r * lde_s + 4 * c) = *reinterpret_cast<const float4*>(P.ent + (e_first + r) * P.ld_ent + 4 * c);
        }
    }
    __syncthreads();
    const int sub = lane >> 3, part = lane & 7;   // loader role: pair (8 it + sub) of the wave's 64, 16-byte piece `part` of a slice
    constexpr int NPF = RQ_NPF;                   // query slices in flight per wave beside the one being multiplied
    for (uint32_t c0 = (uint32_t)wave * 64u; c0 < n; c0 += RQ_THREADS) {
        const bool live = c0 + lane < n;
        int64_t row = 0;
        int er = 0;
        if (live) {
            const uint64_t pr = sp[c0 + lane];
            row = (int64_t)(uint32_t)pr;
            er = (int)((pr >> 32) - (uint64_t)e_first);
        }
        const float* qp[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) qp[it] = P.Q + __shfl(row, 8 * it + sub, 64) * P.ldq + 4 * part;
        const float* const el = erows + er * lde_s;   // this pair's entity row in the LDS image
        static_assert(8 * (NPF - 1) <= 63, "vmcnt is a 6-bit counter");
        rq_f4 qv[NPF][8];
        const int k_last = ((P.k_int - 1) / RQ_KC) * RQ_KC;   // start of the row's last slice
        auto fetch = [&](int k0, rq_f4 (&qs)[8]) {   // k_int % 4 == 0; past the row's end: its last piece again (never multiplied)
            const int kk = min(min(k0, k_last) + 4 * part, P.k_int - 4) - 4 * part;   // (the row pointers carry the lane's piece already)
#pragma unroll
            for (int it = 0; it < 8; ++it) qs[it] = rq_load16(qp[it] + kk);
        };
        float acc = 0.f;
        auto slice = [&](int k0, rq_f4 (&qs)[8]) {
            rq_wait<8 * (NPF - 1)>(qs);   // this slice has landed; the NPF - 1 younger ones fly on
            wave_lds_sync();   // the previous slice has been consumed by every lane
#pragma unroll
            for (int it = 0; it < 8; ++it) *reinterpret_cast<rq_f4*>(myq + (8 * it + sub) * RQ_LD + 4 * part) = qs[it];
            wave_lds_sync();
            fetch(k0 + NPF * RQ_KC, qs);   // the refill, unconditional: NPF slices ahead of the chain
            const int kn = min(RQ_KC, P.k_int - k0);
            if (kn == RQ_KC) {   // a whole slice: sixteen LDS reads issued together, one wait (see rescore_segment_kernel)
                float4 a[RQ_KC / 4], b2[RQ_KC / 4];
#pragma unroll
                for (int c = 0; c < RQ_KC / 4; ++c) {
                    a[c] = *reinterpret_cast<const float4*>(myq + lane * RQ_LD + 4 * c);
                    b2[c] = *reinterpret_cast<const float4*>(el + k0 + 4 * c);
                }
#pragma unroll
                for (int c = 0; c < RQ_KC / 4; ++c) {
                    acc = chain_step<KIND>(a[c].x, b2[c].x, acc); acc = chain_step<KIND>(a[c].y, b2[c].y, acc);
                    acc = chain_step<KIND>(a[c].z, b2[c].z, acc); acc = chain_step<KIND>(a[c].w, b2[c].w, acc);
                }
                return;
            }
#pragma unroll
            for (int c = 0; c < RQ_KC / 4; ++c) {
                if (4 * c < kn) {
                    const float4 a = *reinterpret_cast<const float4*>(myq + lane * RQ_LD + 4 * c);
                    const float4 b2 = *reinterpret_cast<const float4*>(el + k0 + 4 * c);
                    acc = chain_step<KIND>(a.x, b2.x, acc); acc = chain_step<KIND>(a.y, b2.y, acc);
                    acc = chain_step<KIND>(a.z, b2.z, acc); acc = chain_step<KIND>(a.w, b2.w, acc);
                }
            }
        };
#pragma unroll
        for (int u = 0; u < NPF; ++u) fetch(u * RQ_KC, qv[u]);
        for (int k0 = 0; k0 < P.k_int; k0 += NPF * RQ_KC) {
#pragma unroll
            for (int u = 0; u < NPF; ++u)
                if (k0 + u * RQ_KC < P.k_int) slice(k0 + u * RQ_KC, qv[u]);
        }
#pragma unroll
        for (int u = 0; u < NPF; ++u) rq_wait<0>(qv[u]);   // refills never taken: their registers stay theirs until the loads have landed
        if (live) rescore_finish(P, row, acc);
    }
}

__global__ void to_f16_kernel(const float* __restrict__ src, int64_t n_rows, int64_t ld_src, int k_int,
                              _Float16* __restrict__ dst, int64_t ld_dst) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rows * ld_dst) return;
    const int64_t r = t / ld_dst;
    const int c = (int)(t - r * ld_dst);
    dst[t] = c < k_int ? (_Float16)src[r * ld_src + c] : (_Float16)0.f;   // v_cvt_f16_f32: round to nearest even
}

// ---------------------------------------------------------------------------------------------
// precision mode 2: the error band of the half-precision prefilter (derivation in the header / DESIGN.md 4.2).
//   prefilter_bounds_kernel: (max ||e||, max ||e~||, max ||e~ - e||) over the rows of the f32 table and its half copy,
//   prefilter_band_kernel  : per query row  ||dq|| E~max + ||q|| dEmax + g (||q~|| E~max + ||q|| Emax), rounded UP to float.
// Norms in float64 (their own roundoff, ~k 2^-53 relative, is far inside the 1e-6 the band is inflated by); a wave per row.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum_double(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ void atomic_max_nonneg(double* dst, double v) {   // v >= 0: the bit patterns order like the values
    atomicMax(reinterpret_cast<unsigned long long*>(dst), (unsigned long long)__double_as_longlong(v));
}

__global__ __launch_bounds__(256) void prefilter_bounds_kernel(const float* __restrict__ ent, int64_t n_rows, int64_t ld,
                                                               const _Float16* __restrict__ enth, int64_t ldh, int k_int,
                                                               double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    double m_e = 0.0, m_h = 0.0, m_d = 0.0;   // this wave's maxima of the SQUARED norms
    for (int64_t r = wave; r < n_rows; r += n_waves) {
        double se = 0.0, sh = 0.0, sd = 0.0;
        for (int c = lane; c < k_int; c += 64) {
            const double e = (double)ent[r * ld + c], h = (double)(float)enth[r * ldh + c], d = h - e;
            se = fma(e, e, se); sh = fma(h, h, sh); sd = fma(d, d, sd);
        }
        m_e = fmax(m_e, wave_sum_double(se)); m_h = fmax(m_h, wave_sum_double(sh)); m_d = fmax(m_d, wave_sum_double(sd));
    }
    if (lane == 0) {   // (sqrt is monotonic: the maximum of the norms is the root of the maximum of the squares)
        atomic_max_nonneg(out + 0, sqrt(m_e)); atomic_max_nonneg(out + 1, sqrt(m_h)); atomic_max_nonneg(out + 2, sqrt(m_d));
    }
}

__global__ __launch_bounds__(256) void prefilter_band_kernel(const float* __restrict__ Q, int64_t n_rows, int64_t ldq,
                                                             const _Float16* __restrict__ Qh, int64_t ldqh, int k_int,
                                                             const double* __restrict__ bounds, float* __restrict__ band) {
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= n_rows) return;
    double sq = 0.0, sh = 0.0, sd = 0.0;
    for (int c = lane; c < k_int; c += 64) {
        const double q = (double)Q[r * ldq + c], h = (double)(float)Qh[r * ldqh + c], d = h - q;
        sq = fma(q, q, sq); sh = fma(h, h, sh); sd = fma(d, d, sd);
    }
    const double nq = sqrt(wave_sum_double(sq)), nh = sqrt(wave_sum_double(sh)), nd = sqrt(wave_sum_double(sd));
    if (lane != 0) return;
    const double e_max = bounds[0], h_max = bounds[1], d_max = bounds[2];
    const double g = 2.0 * (double)(k_int + 32) * 5.9604644775390625e-08;   // 2 (k + 32) 2^-24
    const double b = (nd * h_max + nq * d_max + g * (nh * h_max + nq * e_max)) * (1.0 + 1e-6);
    band[r] = __double2float_ru(b) + 1e-37f;
}

// ---------------------------------------------------------------------------------------------
// TransE-L2 through the half-precision MFMA prefilter: ||q - e||^2 = |q|^2 - (2 q.e - |e|^2) is a contraction over
// k + 2 coordinates, q' = [2q | -1 | -1], e' = [e | n_hi | n_lo] with n_hi + n_lo ~ |e|^2 split over two halves.
// to_f16_l2_kernel writes these rows (one wave per row); for entity rows the largest |(n_hi + n_lo) - |e|^2| goes to
// *n_res (its actual value enters the error band, so a clamped or badly split norm only widens the band), for query
// rows the doubled f32 row goes to `doubled` (what emg_eval_prefilter_band measures the rounding residual against).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void to_f16_l2_kernel(const float* __restrict__ src, int64_t n_rows, int64_t ld_src, int k_int,
                                                        int is_query, _Float16* __restrict__ dst, int64_t ld_dst,
                                                        float* __restrict__ doubled, double* __restrict__ n_res) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    double worst = 0.0;
    for (int64_t r = wave; r < n_rows; r += n_waves) {
        const float* x = src + r * ld_src;
        _Float16* out = dst + r * ld_dst;
        double n = 0.0;
        for (int c = lane; c < (int)ld_dst; c += 64) {
            float v = 0.f;
            if (c < k_int) {
                v = x[c];
                n = fma((double)v, (double)v, n);
                if (is_query) {
                    v = __fmul_rn(v, 2.0f);
                    if (doubled) doubled[r * ld_src + c] = v;
                }
                v = fminf(fmaxf(v, -65000.f), 65000.f);   // (no infinities in the MFMA operands; the residual norms see the clamp)
            } else if (is_query && c < k_int + 2) {
                v = -1.f;
            }
            if (is_query || c < k_int || c >= k_int + 2) out[c] = (_Float16)v;
        }
        if (!is_query) {
            n = wave_sum_double(n);
            const _Float16 hi = (_Float16)fminf((float)n, 60000.f);
            const double rest = n - (double)(float)hi;
            const _Float16 lo = (_Float16)fminf(fmaxf((float)rest, -60000.f), 60000.f);
            if (lane == 0) { out[k_int] = hi; out[k_int + 1] = lo; }
            worst = fmax(worst, fabs(rest - (double)(float)lo));
        }
    }
    if (!is_query && n_res && lane == 0 && worst > 0.0) atomic_max_nonneg(n_res, worst);
}

// accumulator thresholds of the TransE-L2 prefilter (derivation: DESIGN.md 4.2).  thr[r]: counted when acc >= it,
// thr[n_rows + r]: emitted for exact re-scoring when acc >= it (and not counted).
__global__ void l2_thresholds_kernel(const float* __restrict__ Q, int64_t n_rows, int64_t ldq, const int32_t* __restrict__ pos_int,
                                     const float* __restrict__ band, const double* __restrict__ bounds, int k_int,
                                     float* __restrict__ thr) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    double nq2 = 0.0;
    for (int c = 0; c < k_int; ++c) { const double q = (double)Q[r * ldq + c]; nq2 = fma(q, q, nq2); }
    const double u = 5.9604644775390625e-08;                               // 2^-24
    const double n_max = bounds[0] * bounds[0] * (1.0 + 1e-6);             // largest |e|^2
    const double g_mfma = 2.0 * (double)(k_int + 34) * u;                  // the two norm coordinates join the k products
    const double E = ((double)band[r] + g_mfma * n_max + bounds[3]) * (1.0 + 1e-6) + 1e-30;
    const double gamma = (double)(k_int + 2) * u * 1.01;                   // the exact kernel's chain of k fused squares
    const double m = -(double)pos_int[r];
    const double tg = m > 0.0 ? (m * 1e-5) * (m * 1e-5) * (1.0 - 8.0 * u) : 0.0;
    const double tl = (m + 1.0) > 0.0 ? ((m + 1.0) * 1e-5) * ((m + 1.0) * 1e-5) * (1.0 + 8.0 * u) : 0.0;
    float g = INFINITY, e = -INFINITY;
    if (E == E && E < 1e30) {
        if (m > 0.0) g = nextafterf(__double2float_ru(nq2 * (1.0 + 1e-12) + E - tg / (1.0 + gamma)), INFINITY);
        e = nextafterf(__double2float_rd(nq2 * (1.0 - 1e-12) - E - tl / (1.0 - gamma)), -INFINITY);
    }
    thr[r] = g;
    thr[n_rows + r] = e;
}

__global__ void to_bf16_kernel(const float* __restrict__ src, int64_t n_rows, int64_t ld_src, int k_int,
                               uint16_t* __restrict__ dst, int64_t ld_dst) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rows * ld_dst) return;
    const int64_t r = t / ld_dst;
    const int c = (int)(t - r * ld_dst);
    uint16_t out = 0;
    if (c < k_int) {
        const uint32_t u = __float_as_uint(src[r * ld_src + c]);
        if ((u & 0x7f800000u) == 0x7f800000u && (u & 0x7fffffu)) out = (uint16_t)((u >> 16) | 0x40u);  // NaN
        else out = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);                                  // RNE
    }
    dst[t] = out;
}

// ---------------------------------------------------------------------------------------------
// Pipelined form of count_mfma_kernel for 16-byte-aligned rows (k_int % 4 == 0): same MFMA chain (so the same
// bits), but (1) the next k-slice's global loads are issued before the MFMAs of the current one and stay in
// registers across the tile boundary, (2) rows past the end are CLAMPED instead of predicated (their results
// are masked in the epilogue), (3) LDS is k-major [16][130]: a lane group reads 32 consecutive floats, and
// the stride 130 (= 2 mod 8) makes the transposing dwordx4 -> 4 x ds_write_b32 stores conflict-free too
// (the row-major stride-17 layout of the kernel above has 2-way write conflicts: 25 % of its LDS cycles).
// ---------------------------------------------------------------------------------------------
constexpr int LDK = 130;
#ifndef PIPE_NTB
#define PIPE_NTB 4
#endif

template <bool DENSE, int NTB>  // NTB: 32-column MFMA tiles per wave along the entities (wave tile 64 x 32*NTB)
__global__ __launch_bounds__(256, 2) void count_mfma_pipe_kernel(const CountParams P) {
    constexpr int BNW = 2 * 32 * NTB;       // entities per workgroup tile (two waves across)
    constexpr int LDB = BNW + 2;            // = 2 mod 8: conflict-free transposing stores (see header)
    constexpr int NBR = BNW / 64;           // entity rows each loader thread stages per slice
    __shared__ float As[BK * LDK];
    __shared__ float Bs[BK * LDB];
    __shared__ int pos_s[BM];

    const int64_t id = blockIdx.x;
    const int64_t xcd = id & 7, slot = id >> 3;
    const int64_t qb = slot % P.n_qb;
    const int64_t cb = xcd + 8 * (slot / P.n_qb);
    if (cb >= P.n_cb) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int lrow = tid >> 2, kq = tid & 3;  // loader: rows lrow, lrow+64 ; floats [4kq,4kq+4) of the 16-wide slice
    const int l31 = lane & 31, lhi = lane >> 5;

    if (tid < BM) {
        const int64_t qr = qb * BM + tid;
        pos_s[tid] = (!DENSE && qr < P.n_rows) ? P.pos_int[qr] : 0x7fffffff;
    }
    const float* arow[2];
    const float* brow[NBR];
#pragma unroll
    for (int r = 0; r < 2; ++r) arow[r] = P.Q + min(qb * BM + lrow + 64 * r, P.n_rows - 1) * P.ldq + 4 * kq;
    auto point_b = [&](int64_t tile) {
#pragma unroll
        for (int r = 0; r < NBR; ++r) {
            const int64_t el = min(tile * BNW + lrow + 64 * r, P.n_cand - 1);
            brow[r] = P.ent + (P.cand ? (int64_t)P.cand[el] : el) * P.ld_ent + 4 * kq;
        }
    };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 av[2], bv[NBR];
    auto fetch = [&](int k0) {  // k_int % 4 == 0: a 4-float piece is either whole or past the end
        const bool in = k0 + 4 * kq < P.k_int;
#pragma unroll
        for (int r = 0; r < 2; ++r) av[r] = in ? *reinterpret_cast<const f32x4*>(arow[r] + k0) : zero4;
#pragma unroll
        for (int r = 0; r < NBR; ++r) bv[r] = in ? *reinterpret_cast<const f32x4*>(brow[r] + k0) : zero4;
    };

    unsigned cnt[2][16];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) cnt[a][r] = 0u;

    const int64_t tile0 = cb * P.tiles_per_chunk;
    const int64_t tile1 = min(tile0 + (int64_t)P.tiles_per_chunk, P.n_tiles);
    point_b(tile0);
    fetch(0);
    for (int64_t tile = tile0; tile < tile1; ++tile) {
        float16v acc[2][NTB];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < NTB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

        for (int k0 = 0; k0 < P.k_int; k0 += BK) {
            __syncthreads();  // previous slice's LDS reads done
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int r = 0; r < 2; ++r) As[(4 * kq + c) * LDK + lrow + 64 * r] = av[r][c];
#pragma unroll
                for (int r = 0; r < NBR; ++r) Bs[(4 * kq + c) * LDB + lrow + 64 * r] = bv[r][c];
            }
            __syncthreads();
            // next slice (or the next tile's first one) flies while this one is multiplied
            if (k0 + BK < P.k_int) fetch(k0 + BK);
            else if (tile + 1 < tile1) { point_b(tile + 1); fetch(0); }
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                const int k = 2 * kk + lhi;  // A[i][k=lane>>5], B[k=lane>>5][j]
                float a[2], b[NTB];
#pragma unroll
                for (int t = 0; t < 2; ++t) a[t] = As[k * LDK + wr * 64 + t * 32 + l31];
#pragma unroll
                for (int t = 0; t < NTB; ++t) b[t] = Bs[k * LDB + wc * (32 * NTB) + t * 32 + l31];
#pragma unroll
                for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                    for (int tb = 0; tb < NTB; ++tb)
                        acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
            }
        }
        // epilogue: D[row][col]: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int tb = 0; tb < NTB; ++tb) {
                const int64_t ecol = tile * BNW + wc * (32 * NTB) + tb * 32 + l31;
                const bool cok = ecol < P.n_cand;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = wr * 64 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    float v = acc[ta][tb][r];
                    if (P.model == EMG_HOLE) v = __fmul_rn(v, P.scale);
                    if constexpr (DENSE) {
                        const int64_t qr = qb * BM + rl;
                        if (cok && qr < P.n_rows) P.S[qr * P.lds + ecol] = v;
                    } else {
                        const int ci = cmp_int(v);
                        const int p = pos_s[rl];
                        cnt[ta][r] += (unsigned)(cok && ci > p) + ((unsigned)(cok && ci == p) << 16);
                    }
                }
            }
    }
    if constexpr (!DENSE) {
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                unsigned c = cnt[ta][r];
#pragma unroll
                for (int off = 16; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
                if (l31 == 0) {
                    const int rl = wr * 64 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    const int64_t qr = qb * BM + rl;
                    if (qr < P.n_rows) {
                        if (c & 0xffffu) atomicAdd(&P.cnt_gt[qr], (int)(c & 0xffffu));
                        if (c >> 16) atomicAdd(&P.cnt_eq[qr], (int)(c >> 16));
                    }
                }
            }
    }
}

// Any model through the canonical chain itself: a thread per candidate, query rows one after the other (a wave's
// comparisons of a row are counted with two ballots).  The path of EMG_TRANSE_P (any order of the norm: powf per
// coordinate, no tiling to exploit) — correctness first; orders 1 and 2 have the tiled / v_sad / MFMA kernels above.
template <bool DENSE>
__global__ __launch_bounds__(256) void count_chain_kernel(const CountParams P) {
    const int lane = threadIdx.x & 63;
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = c < P.n_cand;
    const int64_t e = ok ? (P.cand ? (int64_t)P.cand[c] : c) : 0;
    const float* er = P.ent + e * P.ld_ent;
    for (int64_t r = 0; r < P.n_rows; ++r) {
        const float sc = chain_score(P.model, P.Q + r * P.ldq, er, P.k_int, P.scale);
        if constexpr (DENSE) {
            if (ok) P.S[r * P.lds + c] = sc;
        } else {
            const int ci = cmp_int(sc), p = P.pos_int[r];
            const unsigned long long gt = __ballot(ok && ci > p), eq = __ballot(ok && ci == p);
            if (lane == 0) {
                if (gt) atomicAdd(&P.cnt_gt[r], __popcll(gt));
                if (eq) atomicAdd(&P.cnt_eq[r], __popcll(eq));
            }
        }
    }
}

static int launch_count(bool dense, int model, CountParams& P, int precision, hipStream_t st) {
    EMG_REQUIRE(model >= 0 && model <= EMG_TRANSE_P, "unknown model id %d", model);
    if (model == EMG_TRANSE_P) {
        EMG_REQUIRE(P.scale > 0.f, "EMG_TRANSE_P: the order of the norm (passed as `scale`) must be positive");
        if (precision != 0) return fail(EMG_ENOSUP, "EMG_TRANSE_P is evaluated by the exact kernel only");
        if (P.n_rows == 0 || P.n_cand == 0) return EMG_OK;
        const dim3 grid((unsigned)cdiv(P.n_cand, 256)), block(256);
        if (dense) hipLaunchKernelGGL(count_chain_kernel<true>, grid, block, 0, st, P);
        else hipLaunchKernelGGL(count_chain_kernel<false>, grid, block, 0, st, P);
        EMG_LAUNCH_CHECK();
        return EMG_OK;
    }
    if (precision != 0) return fail(EMG_ENOSUP, "eval precision mode %d is not built in this version", precision);
    if (P.n_rows == 0 || P.n_cand == 0) return EMG_OK;
    const bool transe = model <= EMG_TRANSE_L2;
    const int bm = transe ? TQ : BM, bn = transe ? TE : BN;
    P.n_qb = cdiv(P.n_rows, bm);
    P.n_tiles = cdiv(P.n_cand, bn);
    // packed 16-bit per-lane counters stay < 65536; an f32 chunk of 16 tiles (2048 rows, 3.3 MB at k_int=400)
    // stays in the XCD's 4 MB L2 next to the query tiles that stream past it (32 tiles = 6.5 MB thrashed: 27 % misses)
    P.tiles_per_chunk = dense ? 4 : (transe ? 64 : 16);
    P.n_cb = cdiv(P.n_tiles, P.tiles_per_chunk);
    const int64_t blocks = 8 * P.n_qb * cdiv(P.n_cb, 8);
    EMG_REQUIRE(blocks < ((int64_t)1 << 31), "emg_eval_count: grid too large");
    const dim3 grid((unsigned)blocks), block(256);
    static const bool transe_big = getenv("EMG_TRANSE_BIG") == nullptr || atoi(getenv("EMG_TRANSE_BIG")) != 0;   // A/B aid
    if (transe && !dense && transe_big && (P.ldq % 4 == 0) && (P.ld_ent % 4 == 0) && aligned16(P.Q) && aligned16(P.ent)) {
        P.n_qb = cdiv(P.n_rows, UQ);
        P.n_tiles = cdiv(P.n_cand, UE);
        P.tiles_per_chunk = 32;   // 4096 entities per chunk; 8 x 32 counts per thread and field stay far below 16 bits
        P.n_cb = cdiv(P.n_tiles, P.tiles_per_chunk);
        const int64_t blocksb = 8 * P.n_qb * cdiv(P.n_cb, 8);
        EMG_REQUIRE(blocksb < ((int64_t)1 << 31), "emg_eval_count: grid too large");
        if (model == EMG_TRANSE_L1) hipLaunchKernelGGL((count_transe_big_kernel<false>), dim3((unsigned)blocksb), block, 0, st, P);
        else hipLaunchKernelGGL((count_transe_big_kernel<true>), dim3((unsigned)blocksb), block, 0, st, P);
    } else if (transe) {
        if (model == EMG_TRANSE_L1) {
            if (dense) hipLaunchKernelGGL((count_transe_kernel<false, true>), grid, block, 0, st, P);
            else hipLaunchKernelGGL((count_transe_kernel<false, false>), grid, block, 0, st, P);
        } else {
            if (dense) hipLaunchKernelGGL((count_transe_kernel<true, true>), grid, block, 0, st, P);
            else hipLaunchKernelGGL((count_transe_kernel<true, false>), grid, block, 0, st, P);
        }
    } else {
        const bool vec = (P.k_int % 4 == 0) && (P.ldq % 4 == 0) && (P.ld_ent % 4 == 0) && aligned16(P.Q) && aligned16(P.ent);
        if (vec && !dense && PIPE_NTB == 4) {  // 128 x 256 workgroup tiles
            P.n_tiles = cdiv(P.n_cand, 256);
            P.tiles_per_chunk = 8;
            P.n_cb = cdiv(P.n_tiles, P.tiles_per_chunk);
            const int64_t blocks4 = 8 * P.n_qb * cdiv(P.n_cb, 8);
            hipLaunchKernelGGL((count_mfma_pipe_kernel<false, 4>), dim3((unsigned)blocks4), block, 0, st, P);
        } else if (vec) {
            if (dense) hipLaunchKernelGGL((count_mfma_pipe_kernel<true, 2>), grid, block, 0, st, P);
            else hipLaunchKernelGGL((count_mfma_pipe_kernel<false, 2>), grid, block, 0, st, P);
        } else {
            if (dense) hipLaunchKernelGGL((count_mfma_kernel<false, true>), grid, block, 0, st, P);
            else hipLaunchKernelGGL((count_mfma_kernel<false, false>), grid, block, 0, st, P);
        }
    }
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

}  // namespace emg

using namespace emg;

extern "C" int emg_eval_build_queries(int model, const float* ent, int64_t n_ent, int64_t ld_ent, const float* rel,
                                      int64_t n_rel, int64_t ld_rel, int32_t k_int, float scale,
                                      const int32_t* test_spo, int64_t n_q, int side_mode, float* Q, int64_t ldq,
                                      int32_t* pos_int, void* stream) {
    (void)n_ent; (void)n_rel;
    EMG_REQUIRE(model >= 0 && model <= EMG_TRANSE_P, "emg_eval_build_queries: unknown model %d", model);
    EMG_REQUIRE(side_mode >= EMG_EVAL_S && side_mode <= EMG_EVAL_S_O, "emg_eval_build_queries: bad side_mode %d", side_mode);
    EMG_REQUIRE(k_int > 0 && ld_ent >= k_int && ld_rel >= k_int && ldq >= k_int, "emg_eval_build_queries: bad strides");
    const bool cplx = (model == EMG_COMPLEX || model == EMG_HOLE);
    EMG_REQUIRE(!cplx || k_int % 2 == 0, "emg_eval_build_queries: odd k_int for a complex model");
    if (n_q == 0) return EMG_OK;
    EMG_REQUIRE(ent && rel && test_spo && Q && pos_int, "emg_eval_build_queries: null pointer");
    const int64_t n_rows = (side_mode >= EMG_EVAL_SPO) ? 2 * n_q : n_q;
    const int n = cplx ? k_int / 2 : k_int;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(build_queries_kernel, dim3((unsigned)cdiv(n_rows * n, 256)), dim3(256), 0, st, model, ent, ld_ent,
                       rel, ld_rel, (int)k_int, test_spo, n_q, n_rows, side_mode, Q, ldq);
    EMG_LAUNCH_CHECK();
    hipLaunchKernelGGL(pos_int_kernel, dim3((unsigned)cdiv(n_rows, 64)), dim3(64), 0, st, model, ent, ld_ent, (int)k_int,
                       scale, test_spo, n_q, n_rows, side_mode, Q, ldq, pos_int);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

extern "C" int emg_eval_count(int model, const float* Q, int64_t ldq, const int32_t* pos_int, int64_t n_rows,
                              const float* ent, int64_t n_cand, int64_t ld_ent, const int32_t* cand, int32_t k_int,
                              float scale, int precision, const void* ent_bf16, int64_t ld_bf16, int32_t* cnt_gt,
                              int32_t* cnt_eq, void* stream) {
    (void)ent_bf16; (void)ld_bf16;
    EMG_REQUIRE(n_rows >= 0 && n_cand >= 0 && k_int > 0 && ldq >= k_int && ld_ent >= k_int, "emg_eval_count: bad sizes");
    EMG_REQUIRE((n_rows == 0 || n_cand == 0) || (Q && pos_int && ent && cnt_gt && cnt_eq), "emg_eval_count: null pointer");
    CountParams P{};
    P.Q = Q; P.ldq = ldq; P.pos_int = pos_int; P.n_rows = n_rows; P.ent = ent; P.n_cand = n_cand; P.ld_ent = ld_ent;
    P.cand = cand; P.k_int = k_int; P.scale = scale; P.model = model; P.cnt_gt = cnt_gt; P.cnt_eq = cnt_eq;
    return launch_count(false, model, P, precision, (hipStream_t)stream);
}

extern "C" int emg_eval_scores_dense(int model, const float* Q, int64_t ldq, int64_t n_rows, const float* ent,
                                     int64_t n_cand, int64_t ld_ent, const int32_t* cand, int32_t k_int, float scale,
                                     int precision, const void* ent_bf16, int64_t ld_bf16, float* S, int64_t lds,
                                     void* stream) {
    (void)ent_bf16; (void)ld_bf16;
    EMG_REQUIRE(n_rows >= 0 && n_cand >= 0 && k_int > 0 && ldq >= k_int && ld_ent >= k_int && lds >= n_cand,
                "emg_eval_scores_dense: bad sizes");
    EMG_REQUIRE((n_rows == 0 || n_cand == 0) || (Q && ent && S), "emg_eval_scores_dense: null pointer");
    CountParams P{};
    P.Q = Q; P.ldq = ldq; P.n_rows = n_rows; P.ent = ent; P.n_cand = n_cand; P.ld_ent = ld_ent; P.cand = cand;
    P.k_int = k_int; P.scale = scale; P.model = model; P.S = S; P.lds = lds;
    return launch_count(true, model, P, precision, (hipStream_t)stream);
}

extern "C" int emg_eval_filter_count(int model, const float* Q, int64_t ldq, const int32_t* pos_int, int64_t n_rows,
                                     const float* ent, int64_t n_local, int64_t ld_ent, int64_t ent_offset,
                                     int32_t k_int, float scale, int precision, const int64_t* filt_ptr,
                                     const int32_t* filt_idx, int32_t* fcnt_gt, int32_t* fcnt_eq, void* stream) {
    EMG_REQUIRE(model >= 0 && model <= EMG_TRANSE_P, "emg_eval_filter_count: unknown model %d", model);
    if (precision != 0) return fail(EMG_ENOSUP, "eval precision mode %d is not built in this version", precision);
    EMG_REQUIRE(n_rows >= 0 && k_int > 0 && ldq >= k_int && ld_ent >= k_int, "emg_eval_filter_count: bad sizes");
    if (n_rows == 0) return EMG_OK;
    EMG_REQUIRE(Q && pos_int && ent && filt_ptr && fcnt_gt && fcnt_eq, "emg_eval_filter_count: null pointer");
    hipLaunchKernelGGL(filter_count_kernel, dim3((unsigned)cdiv(n_rows * 64, 256)), dim3(256), 0, (hipStream_t)stream,
                       model, Q, ldq, pos_int, n_rows, ent, n_local, ld_ent, ent_offset, (int)k_int, scale, filt_ptr,
                       filt_idx, fcnt_gt, fcnt_eq);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

extern "C" int emg_eval_rescore_pairs(int model, const float* Q, int64_t ldq, const int32_t* pos_int, const float* ent,
                                      int64_t ld_ent, int64_t ent_offset, int32_t k_int, float scale,
                                      const uint64_t* pairs, int64_t pairs_capacity, const uint32_t* pair_count,
                                      int64_t n_segments, int32_t* cnt_gt, int32_t* cnt_eq, void* stream) {
    return emg_eval_rescore_pairs_ex(model, Q, ldq, pos_int, ent, ld_ent, ent_offset, k_int, scale, pairs, pairs_capacity, pair_count,
                                     n_segments, 4, cnt_gt, cnt_eq, stream);
}

extern "C" int emg_eval_rescore_pairs_ex(int model, const float* Q, int64_t ldq, const int32_t* pos_int, const float* ent,
                                         int64_t ld_ent, int64_t ent_offset, int32_t k_int, float scale,
                                         const uint64_t* pairs, int64_t pairs_capacity, const uint32_t* pair_count,
                                         int64_t n_segments, int32_t segments_per_block, int32_t* cnt_gt, int32_t* cnt_eq,
                                         void* stream) {
    return emg_eval_rescore_pairs_rows(model, Q, ldq, pos_int, ent, ld_ent, ent_offset, k_int, scale, pairs, pairs_capacity, pair_count,
                                       n_segments, segments_per_block, segments_per_block == 8 ? 32 : 0, cnt_gt, cnt_eq, stream);
}

extern "C" int emg_eval_rescore_pairs_rows(int model, const float* Q, int64_t ldq, const int32_t* pos_int, const float* ent,
                                           int64_t ld_ent, int64_t ent_offset, int32_t k_int, float scale,
                                           const uint64_t* pairs, int64_t pairs_capacity, const uint32_t* pair_count,
                                           int64_t n_segments, int32_t segments_per_block, int32_t rows_per_segment,
                                           int32_t* cnt_gt, int32_t* cnt_eq, void* stream) {
    EMG_REQUIRE(segments_per_block >= 4 && segments_per_block % 4 == 0 && segments_per_block <= 64,
                "emg_eval_rescore_pairs: segments_per_block must be a multiple of 4 (emg_eval_prefilter_waves; 4 = emg_eval_prefilter_sad)");
    EMG_REQUIRE(model >= EMG_TRANSE_L1 && model <= EMG_HOLE, "emg_eval_rescore_pairs: unknown model id %d", model);
    EMG_REQUIRE(Q && pos_int && ent && pairs && pair_count && cnt_gt && cnt_eq, "emg_eval_rescore_pairs: null pointer");
    if (n_segments <= 0) return EMG_OK;
    EMG_REQUIRE(pairs_capacity >= n_segments, "emg_eval_rescore_pairs: pair buffer smaller than one entry per segment");
    RescoreParams P{};
    P.model = model; P.Q = Q; P.ldq = ldq; P.pos_int = pos_int; P.ent = ent; P.ld_ent = ld_ent; P.ent_offset = ent_offset;
    P.k_int = k_int; P.scale = scale; P.pairs = pairs; P.cap = (uint32_t)(pairs_capacity / n_segments);
    P.seg_count = pair_count; P.n_seg = (uint32_t)n_segments;
    P.cnt_gt = cnt_gt; P.cnt_eq = cnt_eq;
    P.min_pairs = 0u; P.max_pairs = 0xffffffffu;
    const bool vec = (k_int % 4 == 0) && (ldq % 4 == 0) && (ld_ent % 4 == 0) && aligned16(Q) && aligned16(ent);
    P.groups_per_block = (uint32_t)(segments_per_block / 4);
    const int64_t per = 8 * (int64_t)P.groups_per_block;                    // one workgroup per 4-segment group, XCD-aligned
    const int64_t blocks = cdiv(cdiv(n_segments, 4), per) * per;
    EMG_REQUIRE(blocks < ((int64_t)1 << 31), "emg_eval_rescore_pairs: too many segments");
    const dim3 grid((unsigned)blocks), block(256);
    hipStream_t st = (hipStream_t)stream;
    // the f16 prefilter's segments (8 per workgroup, 32 query rows each): query rows in LDS, one workgroup per segment
    static const bool seg_off = [] { const char* e = getenv("EMG_RESCORE"); return e && !strcmp(e, "pairs"); }();
    const int seg_waves = rescore_segment_lds(k_int, 8) <= 144 * 1024 ? 8 : (rescore_segment_lds(k_int, 4) <= 144 * 1024 ? 4 : 0);
    if (vec && rows_per_segment > 0 && rows_per_segment <= RQ_ROWS && !seg_off && seg_waves) {
        static const uint32_t min_pairs = [] { const char* e = getenv("EMG_RESCORE_MIN"); return e ? (uint32_t)atoi(e) : (uint32_t)RQ_MIN_PAIRS; }();
        P.min_pairs = seg_waves == 8 ? min_pairs : min_pairs / 2;
        const int64_t nb = cdiv(n_segments, segments_per_block);
        const int64_t sblocks = cdiv(nb, 8) * 8 * segments_per_block;
        EMG_REQUIRE(sblocks < ((int64_t)1 << 31), "emg_eval_rescore_pairs: too many segments");
        const size_t lds = rescore_segment_lds(k_int, seg_waves);
        const int kind = model == EMG_TRANSE_L1 ? 1 : (model == EMG_TRANSE_L2 ? 2 : 0);
        static std::atomic<uint64_t> done[6];
        const void* fn8[3] = {(const void*)rescore_segment_kernel<0, 8>, (const void*)rescore_segment_kernel<1, 8>, (const void*)rescore_segment_kernel<2, 8>};
        const void* fn4[3] = {(const void*)rescore_segment_kernel<0, 4>, (const void*)rescore_segment_kernel<1, 4>, (const void*)rescore_segment_kernel<2, 4>};
        const void* fn = seg_waves == 8 ? fn8[kind] : fn4[kind];
        if (lds > 48 * 1024) {   // opt in to > 64 KB of dynamic LDS once per device and kernel
            int dev = 0;
            EMG_HIP(hipGetDevice(&dev));
            const uint64_t bit = 1ull << (dev & 63);
            std::atomic<uint64_t>& flag = done[kind + (seg_waves == 8 ? 0 : 3)];
            if (!(flag.load(std::memory_order_acquire) & bit)) {
                EMG_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
                flag.fetch_or(bit, std::memory_order_release);
            }
        }
        const dim3 sgrid((unsigned)sblocks), sblock(64 * seg_waves);
        if (seg_waves == 8) {
            if (kind == 1) hipLaunchKernelGGL((rescore_segment_kernel<1, 8>), sgrid, sblock, lds, st, P);
            else if (kind == 2) hipLaunchKernelGGL((rescore_segment_kernel<2, 8>), sgrid, sblock, lds, st, P);
            else hipLaunchKernelGGL((rescore_segment_kernel<0, 8>), sgrid, sblock, lds, st, P);
        } else {
            if (kind == 1) hipLaunchKernelGGL((rescore_segment_kernel<1, 4>), sgrid, sblock, lds, st, P);
            else if (kind == 2) hipLaunchKernelGGL((rescore_segment_kernel<2, 4>), sgrid, sblock, lds, st, P);
            else hipLaunchKernelGGL((rescore_segment_kernel<0, 4>), sgrid, sblock, lds, st, P);
        }
        EMG_LAUNCH_CHECK();
        P.max_pairs = P.min_pairs; P.min_pairs = 0u;   // the rest, below: a wave per segment
        if (P.max_pairs == 0u) return EMG_OK;
    }
    if (model == EMG_TRANSE_L1) {
        if (vec) hipLaunchKernelGGL((rescore_pairs_kernel<true, 1>), grid, block, 0, st, P);
        else hipLaunchKernelGGL((rescore_pairs_kernel<false, 1>), grid, block, 0, st, P);
    } else if (model == EMG_TRANSE_L2) {
        if (vec) hipLaunchKernelGGL((rescore_pairs_kernel<true, 2>), grid, block, 0, st, P);
        else hipLaunchKernelGGL((rescore_pairs_kernel<false, 2>), grid, block, 0, st, P);
    } else {
        if (vec) hipLaunchKernelGGL((rescore_pairs_kernel<true, 0>), grid, block, 0, st, P);
        else hipLaunchKernelGGL((rescore_pairs_kernel<false, 0>), grid, block, 0, st, P);
    }
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

extern "C" int64_t emg_eval_rescore_tiles_ws_bytes(int64_t n_local) {
    if (n_local <= 0) return 256;
    const int64_t n_tiles = cdiv(n_local, (int64_t)1 << RT_TILE_LOG);
    return 3 * 4 * (n_tiles + 64) + 256;
}

extern "C" int emg_eval_rescore_pairs_tiles(int model, const float* Q, int64_t ldq, const int32_t* pos_int, const float* ent,
                                            int64_t ld_ent, int64_t ent_offset, int64_t n_local, int32_t k_int, float scale,
                                            const uint64_t* pairs, int64_t pairs_capacity, const uint32_t* pair_count,
                                            int64_t n_segments, uint64_t* sorted, int64_t sorted_capacity, void* tile_ws,
                                            int64_t tile_ws_bytes, int32_t* cnt_gt, int32_t* cnt_eq, void* stream) {
    EMG_REQUIRE(model >= EMG_TRANSE_L1 && model <= EMG_HOLE, "emg_eval_rescore_pairs_tiles: unknown model id %d", model);
    EMG_REQUIRE(Q && pos_int && ent && pairs && pair_count && cnt_gt && cnt_eq && sorted && tile_ws, "emg_eval_rescore_pairs_tiles: null pointer");
    if (n_segments <= 0 || n_local <= 0) return EMG_OK;
    EMG_REQUIRE(pairs_capacity >= n_segments && sorted_capacity >= pairs_capacity, "emg_eval_rescore_pairs_tiles: pair buffers too small");
    EMG_REQUIRE(tile_ws_bytes >= emg_eval_rescore_tiles_ws_bytes(n_local), "emg_eval_rescore_pairs_tiles: tile workspace too small (must be zero on first use)");
    EMG_REQUIRE(n_local < ((int64_t)1 << 32), "emg_eval_rescore_pairs_tiles: too many entities");
    const bool vec = (k_int % 4 == 0) && (ldq % 4 == 0) && (ld_ent % 4 == 0) && aligned16(Q) && aligned16(ent);
    const int waves = rescore_segment_lds(k_int, 8) <= 144 * 1024 ? 8 : (rescore_segment_lds(k_int, 4) <= 144 * 1024 ? 4 : 0);
    if (!vec || !waves) return fail(EMG_ENOSUP, "emg_eval_rescore_pairs_tiles: needs 16-byte aligned rows whose 32-row image fits LDS");
    TileParams T{};
    RescoreParams& P = T.R;
    P.model = model; P.Q = Q; P.ldq = ldq; P.pos_int = pos_int; P.ent = ent; P.ld_ent = ld_ent; P.ent_offset = ent_offset;
    P.k_int = k_int; P.scale = scale; P.pairs = pairs; P.cap = (uint32_t)(pairs_capacity / n_segments);
    P.seg_count = pair_count; P.n_seg = (uint32_t)n_segments; P.cnt_gt = cnt_gt; P.cnt_eq = cnt_eq;
    T.n_local = n_local; T.n_tiles = (uint32_t)cdiv(n_local, (int64_t)1 << RT_TILE_LOG);
    T.tcnt = (uint32_t*)tile_ws; T.toff = T.tcnt + T.n_tiles + 64; T.cursor = T.toff + T.n_tiles + 64;
    T.sorted = sorted; T.sorted_cap = (uint64_t)sorted_capacity;
    hipStream_t st = (hipStream_t)stream;
    const dim3 sgrid((unsigned)cdiv(n_segments, 4)), sblock(256);
    hipLaunchKernelGGL(tile_sort_kernel<false>, sgrid, sblock, 0, st, T);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, st, T);
    hipLaunchKernelGGL(tile_sort_kernel<true>, sgrid, sblock, 0, st, T);
    EMG_LAUNCH_CHECK();
    const size_t lds = rescore_segment_lds(k_int, waves);
    const int kind = model == EMG_TRANSE_L1 ? 1 : (model == EMG_TRANSE_L2 ? 2 : 0);
    static std::atomic<uint64_t> done[6];
    const void* fn8[3] = {(const void*)rescore_tile_kernel<0, 8>, (const void*)rescore_tile_kernel<1, 8>, (const void*)rescore_tile_kernel<2, 8>};
    const void* fn4[3] = {(const void*)rescore_tile_kernel<0, 4>, (const void*)rescore_tile_kernel<1, 4>, (const void*)rescore_tile_kernel<2, 4>};
    const void* fn = waves == 8 ? fn8[kind] : fn4[kind];
    if (lds > 48 * 1024) {   // opt in to > 64 KB of dynamic LDS once per device and kernel
        int dev = 0;
        EMG_HIP(hipGetDevice(&dev));
        const uint64_t bit = 1ull << (dev & 63);
        std::atomic<uint64_t>& flag = done[kind + (waves == 8 ? 0 : 3)];
        if (!(flag.load(std::memory_order_acquire) & bit)) {
            EMG_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
            flag.fetch_or(bit, std::memory_order_release);
        }
    }
    const dim3 tgrid(T.n_tiles), tblock(64 * waves);
    if (waves == 8) {
        if (kind == 1) hipLaunchKernelGGL((rescore_tile_kernel<1, 8>), tgrid, tblock, lds, st, T);
        else if (kind == 2) hipLaunchKernelGGL((rescore_tile_kernel<2, 8>), tgrid, tblock, lds, st, T);
        else hipLaunchKernelGGL((rescore_tile_kernel<0, 8>), tgrid, tblock, lds, st, T);
    } else {
        if (kind == 1) hipLaunchKernelGGL((rescore_tile_kernel<1, 4>), tgrid, tblock, lds, st, T);
        else if (kind == 2) hipLaunchKernelGGL((rescore_tile_kernel<2, 4>), tgrid, tblock, lds, st, T);
        else hipLaunchKernelGGL((rescore_tile_kernel<0, 4>), tgrid, tblock, lds, st, T);
    }
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

extern "C" int emg_to_f16(const float* src, int64_t n_rows, int64_t ld_src, int32_t k_int, void* dst_f16, int64_t ld_dst,
                          void* stream) {
    EMG_REQUIRE(src && dst_f16 && n_rows >= 0 && ld_src >= k_int && ld_dst >= k_int, "emg_to_f16: bad arguments");
    if (n_rows == 0) return EMG_OK;
    hipLaunchKernelGGL(to_f16_kernel, dim3((unsigned)cdiv(n_rows * ld_dst, 256)), dim3(256), 0, (hipStream_t)stream, src,
                       n_rows, ld_src, (int)k_int, (_Float16*)dst_f16, ld_dst);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

extern "C" int emg_eval_prefilter_bounds(const float* ent, int64_t n_rows, int64_t ld_ent, const void* ent_f16,
                                         int64_t ld_f16, int32_t k_int, double* bounds3, void* stream) {
    EMG_REQUIRE(ent && ent_f16 && bounds3 && n_rows >= 0 && ld_ent >= k_int && ld_f16 >= k_int && k_int > 0,
                "emg_eval_prefilter_bounds: bad arguments");
    EMG_HIP(hipMemsetAsync(bounds3, 0, 3 * sizeof(double), (hipStream_t)stream));
    if (n_rows == 0) return EMG_OK;
    const int64_t waves = n_rows < 16384 ? n_rows : 16384;   // a few rows per wave, one atomic triple per wave
    hipLaunchKernelGGL(prefilter_bounds_kernel, dim3((unsigned)cdiv(waves * 64, 256)), dim3(256), 0, (hipStream_t)stream, ent,
                       n_rows, ld_ent, (const _Float16*)ent_f16, ld_f16, (int)k_int, bounds3);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

extern "C" int emg_eval_prefilter_band(const float* q, int64_t n_rows, int64_t ldq, const void* q_f16, int64_t ldq_f16,
                                       int32_t k_int, const double* bounds3, float* band, void* stream) {
    EMG_REQUIRE(n_rows >= 0 && k_int > 0 && ldq >= k_int && ldq_f16 >= k_int, "emg_eval_prefilter_band: bad sizes");
    if (n_rows == 0) return EMG_OK;
    EMG_REQUIRE(q && q_f16 && bounds3 && band, "emg_eval_prefilter_band: null pointer");
    hipLaunchKernelGGL(prefilter_band_kernel, dim3((unsigned)cdiv(n_rows * 64, 256)), dim3(256), 0, (hipStream_t)stream, q,
                       n_rows, ldq, (const _Float16*)q_f16, ldq_f16, (int)k_int, bounds3, band);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

extern "C" int emg_to_f16_l2(const float* src, int64_t n_rows, int64_t ld_src, int32_t k_int, int is_query, void* dst_f16,
                             int64_t ld_dst, float* doubled, double* n_residual_max, void* stream) {
    EMG_REQUIRE(n_rows >= 0 && k_int > 0 && ld_src >= k_int && ld_dst >= k_int + 2, "emg_to_f16_l2: bad sizes");
    if (n_rows == 0) return EMG_OK;
    EMG_REQUIRE(src && dst_f16, "emg_to_f16_l2: null pointer");
    if (!is_query && n_residual_max) EMG_HIP(hipMemsetAsync(n_residual_max, 0, sizeof(double), (hipStream_t)stream));
    const int64_t b = cdiv(n_rows, 4 * 4);
    hipLaunchKernelGGL(to_f16_l2_kernel, dim3((unsigned)(b > 65536 ? 65536 : b)), dim3(256), 0, (hipStream_t)stream, src, n_rows,
                       ld_src, k_int, is_query, (_Float16*)dst_f16, ld_dst, doubled, n_residual_max);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

extern "C" int emg_eval_l2_thresholds(const float* Q, int64_t n_rows, int64_t ldq, const int32_t* pos_int, const float* band,
                                      const double* bounds4, int32_t k_int, float* thr, void* stream) {
    EMG_REQUIRE(n_rows >= 0 && k_int > 0 && ldq >= k_int, "emg_eval_l2_thresholds: bad sizes");
    if (n_rows == 0) return EMG_OK;
    EMG_REQUIRE(Q && pos_int && band && bounds4 && thr, "emg_eval_l2_thresholds: null pointer");
    hipLaunchKernelGGL(l2_thresholds_kernel, dim3((unsigned)cdiv(n_rows, 128)), dim3(128), 0, (hipStream_t)stream, Q, n_rows, ldq,
                       pos_int, band, bounds4, k_int, thr);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

extern "C" int emg_to_bf16(const float* src, int64_t n_rows, int64_t ld_src, int32_t k_int, void* dst_bf16,
                           int64_t ld_dst, void* stream) {
    EMG_REQUIRE(src && dst_bf16 && n_rows >= 0 && ld_src >= k_int && ld_dst >= k_int, "emg_to_bf16: bad arguments");
    if (n_rows == 0) return EMG_OK;
    hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)cdiv(n_rows * ld_dst, 256)), dim3(256), 0, (hipStream_t)stream, src,
                       n_rows, ld_src, (int)k_int, (uint16_t*)dst_bf16, ld_dst);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}
