// emg_rank_bf16.hip — bf16 MFMA variant of the 1-vs-all count kernel (DistMult / ComplEx / HolE).
//
// Same contraction and the same fused compare-and-count epilogue as emg_rank.hip::count_mfma_kernel, but the
// query rows and the entity table are bf16 (round-to-nearest-even copies made by emg_to_bf16) and the inner
// product runs on v_mfma_f32_32x32x16_bf16 (16x the f32-MFMA rate, half the bytes).  NOT a parity mode:
// bf16 inputs carry ~3 significant digits while ranks compare int32(score*1e5), so ranks agree with the exact
// f32 path only statistically (tests/test_hip_kernels.py::test_bf16_rank_agreement reports the rate).  To keep
// the one comparison that matters structurally exact, the TRUE entity of each query row is excluded from the
// count by INDEX (self_ent) instead of by score equality; the caller adds it back as one tie.
//
// Tile: 128 query rows x 128 entities x BK=32 per step; 4 waves (2x2), each 64x64 = 2x2 MFMA 32x32x16.
// LDS rows are 64 B (32 bf16); a lane reads its 8-element fragment with ds_read_b128; the 16-byte slot index
// is XOR-swizzled with (row>>2)&3 so the 16 lanes of a b128 lane group hit 16 distinct slots of the 256-byte
// bank row (conflict-free) — the writer applies the same XOR.
#include "emg_common.hpp"

namespace emg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct CountBf16Params {
    const uint16_t* Q; int64_t ldq; const int32_t* pos_int; const int32_t* self_ent; int64_t n_rows;
    const uint16_t* ent; int64_t n_cand; int64_t ld_ent; const int32_t* cand; int64_t ent_offset;
    int32_t k_pad; float scale; int32_t model;
    int32_t* cnt_gt; int32_t* cnt_eq;
    float* S; int64_t lds;
    int64_t n_qb; int64_t n_cb; int64_t n_tiles; int32_t tiles_per_chunk;
};

constexpr int HBM_ = 128, HBN_ = 128, HBK_ = 32;

__device__ __forceinline__ int lds_slot_off(int row, int slot) {  // byte offset of a 16-byte slot
    return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4);
}

template <bool DENSE>
__global__ __launch_bounds__(256) void count_mfma_bf16_kernel(const CountBf16Params P) {
    __shared__ __attribute__((aligned(16))) unsigned char As[HBM_ * 64];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[HBN_ * 64];
    __shared__ int pos_s[HBM_];
    __shared__ int self_s[HBM_];

    const int64_t id = blockIdx.x;
    const int64_t xcd = id & 7, slot_id = id >> 3;
    const int64_t qb = slot_id % P.n_qb;
    const int64_t cb = xcd + 8 * (slot_id / P.n_qb);
    if (cb >= P.n_cb) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int lrow = tid >> 2, lslot = tid & 3;  // loader: rows lrow, lrow+64; 16-byte slot lslot of the 64-byte k-slice
    const int l31 = lane & 31, lhi = lane >> 5;

    if (tid < HBM_) {
        const int64_t qr = qb * HBM_ + tid;
        pos_s[tid] = (!DENSE && qr < P.n_rows) ? P.pos_int[qr] : 0x7fffffff;
        self_s[tid] = (!DENSE && P.self_ent && qr < P.n_rows) ? P.self_ent[qr] : -1;
    }

    const uint16_t* arow[2];
    bool aok[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int64_t qr = qb * HBM_ + lrow + 64 * r;
        aok[r] = qr < P.n_rows;
        arow[r] = P.Q + (aok[r] ? qr : 0) * P.ldq + 8 * lslot;
    }

    unsigned cnt[2][16];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) cnt[a][r] = 0u;

    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    const int64_t tile0 = cb * P.tiles_per_chunk;
    const int64_t tile1 = min(tile0 + (int64_t)P.tiles_per_chunk, P.n_tiles);
    for (int64_t tile = tile0; tile < tile1; ++tile) {
        const uint16_t* brow[2];
        bool bok[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int64_t el = tile * HBN_ + lrow + 64 * r;
            bok[r] = el < P.n_cand;
            const int64_t erow = bok[r] ? (P.cand ? (int64_t)P.cand[el] : el) : 0;
            brow[r] = P.ent + erow * P.ld_ent + 8 * lslot;
        }
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

        // register-staged prefetch: tile k0+32 is loaded while tile k0 is multiplied
        uint4 av[2], bv[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            av[r] = aok[r] ? *reinterpret_cast<const uint4*>(arow[r]) : zero4;
            bv[r] = bok[r] ? *reinterpret_cast<const uint4*>(brow[r]) : zero4;
        }
        for (int k0 = 0; k0 < P.k_pad; k0 += HBK_) {
            __syncthreads();  // previous step's LDS reads are done
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                *reinterpret_cast<uint4*>(As + lds_slot_off(lrow + 64 * r, lslot)) = av[r];
                *reinterpret_cast<uint4*>(Bs + lds_slot_off(lrow + 64 * r, lslot)) = bv[r];
            }
            __syncthreads();
            if (k0 + HBK_ < P.k_pad) {
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    av[r] = aok[r] ? *reinterpret_cast<const uint4*>(arow[r] + k0 + HBK_) : zero4;
                    bv[r] = bok[r] ? *reinterpret_cast<const uint4*>(brow[r] + k0 + HBK_) : zero4;
                }
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {   // two K=16 MFMA steps per 32-wide k-slice
                const int slot = 2 * ks + lhi;  // lane holds k = 16*ks + 8*(lane>>5) .. +8
                bf16x8 a[2], b[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    a[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(As + lds_slot_off(wr * 64 + t * 32 + l31, slot)));
                    b[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Bs + lds_slot_off(wc * 64 + t * 32 + l31, slot)));
                }
#pragma unroll
                for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                    for (int tb = 0; tb < 2; ++tb)
                        acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
            }
        }
        // epilogue: D[row][col]: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                const int64_t ecol = tile * HBN_ + wc * 64 + tb * 32 + l31;
                const bool cok = ecol < P.n_cand;
                const int gid = cok ? (int)((P.cand ? (int64_t)P.cand[ecol] : ecol) + P.ent_offset) : -2;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = wr * 64 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    float v = acc[ta][tb][r];
                    if (P.model == EMG_HOLE) v = v * P.scale;
                    if constexpr (DENSE) {
                        const int64_t qr = qb * HBM_ + rl;
                        if (cok && qr < P.n_rows) P.S[qr * P.lds + ecol] = v;
                    } else {
                        const int ci = (int)(v * 100000.0f);
                        const int p = pos_s[rl];
                        const bool use = cok && gid != self_s[rl];
                        cnt[ta][r] += (unsigned)(use && ci > p) + ((unsigned)(use && ci == p) << 16);
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
                    const int64_t qr = qb * HBM_ + rl;
                    if (qr < P.n_rows) {
                        if (c & 0xffffu) atomicAdd(&P.cnt_gt[qr], (int)(c & 0xffffu));
                        if (c >> 16) atomicAdd(&P.cnt_eq[qr], (int)(c >> 16));
                    }
                }
            }
    }
}

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// positive's comparison integer from the bf16-rounded operands (fp32 accumulate), + the true entity id
__global__ void pos_int_bf16_kernel(int model, const uint16_t* __restrict__ ent, int64_t ld_ent, int k_int, float scale,
                                    const int32_t* __restrict__ test, int64_t n_q, int64_t n_rows, int side_mode,
                                    const uint16_t* __restrict__ Q, int64_t ldq, int32_t* __restrict__ pos_int,
                                    int32_t* __restrict__ self_ent) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    int64_t qi; bool obj;
    if (side_mode == EMG_EVAL_S) { qi = r; obj = false; }
    else if (side_mode == EMG_EVAL_O) { qi = r; obj = true; }
    else { obj = r < n_q; qi = r < n_q ? r : r - n_q; }
    const int32_t tgt = obj ? test[3 * qi + 2] : test[3 * qi + 0];
    const uint16_t* q = Q + r * ldq;
    const uint16_t* e = ent + (int64_t)tgt * ld_ent;
    float acc = 0.f;
    for (int k = 0; k < k_int; ++k) acc = fmaf(bf16_to_f32(q[k]), bf16_to_f32(e[k]), acc);
    if (model == EMG_HOLE) acc *= scale;
    pos_int[r] = (int)(acc * 100000.0f);
    self_ent[r] = tgt;
}

// filter counts on the bf16-rounded operands; the row's own entity is skipped (it was excluded from the count)
__global__ __launch_bounds__(256) void filter_count_bf16_kernel(int model, const uint16_t* __restrict__ Q, int64_t ldq,
                                                                const int32_t* __restrict__ pos_int,
                                                                const int32_t* __restrict__ self_ent, int64_t n_rows,
                                                                const uint16_t* __restrict__ ent, int64_t n_local,
                                                                int64_t ld_ent, int64_t ent_offset, int k_int,
                                                                float scale, const int64_t* __restrict__ fptr,
                                                                const int32_t* __restrict__ fidx,
                                                                int32_t* __restrict__ fgt, int32_t* __restrict__ feq) {
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= n_rows) return;
    const int p = pos_int[r];
    const int self = self_ent[r];
    const uint16_t* q = Q + r * ldq;
    int gt = 0, eq = 0;
    for (int64_t u = fptr[r] + lane; u < fptr[r + 1]; u += 64) {
        const int gidx = fidx[u];
        const int64_t e = (int64_t)gidx - ent_offset;
        if (e < 0 || e >= n_local || gidx == self) continue;
        const uint16_t* er = ent + e * ld_ent;
        float acc = 0.f;
        for (int k = 0; k < k_int; ++k) acc = fmaf(bf16_to_f32(q[k]), bf16_to_f32(er[k]), acc);
        if (model == EMG_HOLE) acc *= scale;
        const int ci = (int)(acc * 100000.0f);
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

static int launch_bf16(bool dense, CountBf16Params& P, hipStream_t st) {
    EMG_REQUIRE(P.model >= EMG_DISTMULT && P.model <= EMG_HOLE, "bf16 eval: model %d is not a contraction (TransE stays f32 VALU)", P.model);
    EMG_REQUIRE(P.k_pad > 0 && P.k_pad % HBK_ == 0 && P.ldq >= P.k_pad && P.ld_ent >= P.k_pad,
                "bf16 eval: rows must be zero-padded to a multiple of %d elements (k_pad=%d ldq=%lld ld=%lld)", HBK_,
                P.k_pad, (long long)P.ldq, (long long)P.ld_ent);
    EMG_REQUIRE(P.ldq % 8 == 0 && P.ld_ent % 8 == 0 && aligned16(P.Q) && aligned16(P.ent), "bf16 eval: rows must be 16-byte aligned");
    if (P.n_rows == 0 || P.n_cand == 0) return EMG_OK;
    P.n_qb = cdiv(P.n_rows, HBM_);
    P.n_tiles = cdiv(P.n_cand, HBN_);
    P.tiles_per_chunk = dense ? 4 : 32;
    P.n_cb = cdiv(P.n_tiles, P.tiles_per_chunk);
    const int64_t blocks = 8 * P.n_qb * cdiv(P.n_cb, 8);
    EMG_REQUIRE(blocks < ((int64_t)1 << 31), "bf16 eval: grid too large");
    if (dense) hipLaunchKernelGGL(count_mfma_bf16_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, P);
    else hipLaunchKernelGGL(count_mfma_bf16_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, P);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

}  // namespace emg

using namespace emg;

extern "C" int emg_eval_pos_int_bf16(int model, const void* ent_bf16, int64_t ld_ent, int32_t k_int, float scale,
                                     const int32_t* test_spo, int64_t n_q, int side_mode, const void* q_bf16,
                                     int64_t ldq, int32_t* pos_int, int32_t* self_ent, void* stream) {
    EMG_REQUIRE(model >= EMG_DISTMULT && model <= EMG_HOLE, "emg_eval_pos_int_bf16: model %d unsupported", model);
    EMG_REQUIRE(side_mode >= EMG_EVAL_S && side_mode <= EMG_EVAL_S_O, "emg_eval_pos_int_bf16: bad side_mode");
    if (n_q == 0) return EMG_OK;
    EMG_REQUIRE(ent_bf16 && test_spo && q_bf16 && pos_int && self_ent, "emg_eval_pos_int_bf16: null pointer");
    const int64_t n_rows = side_mode >= EMG_EVAL_SPO ? 2 * n_q : n_q;
    hipLaunchKernelGGL(pos_int_bf16_kernel, dim3((unsigned)cdiv(n_rows, 64)), dim3(64), 0, (hipStream_t)stream, model,
                       (const uint16_t*)ent_bf16, ld_ent, (int)k_int, scale, test_spo, n_q, n_rows, side_mode,
                       (const uint16_t*)q_bf16, ldq, pos_int, self_ent);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

extern "C" int emg_eval_count_bf16(int model, const void* q_bf16, int64_t ldq, const int32_t* pos_int,
                                   const int32_t* self_ent, int64_t n_rows, const void* ent_bf16, int64_t n_cand,
                                   int64_t ld_ent, const int32_t* cand, int64_t ent_offset, int32_t k_pad, float scale,
                                   int32_t* cnt_gt, int32_t* cnt_eq, void* stream) {
    EMG_REQUIRE((n_rows == 0 || n_cand == 0) || (q_bf16 && pos_int && ent_bf16 && cnt_gt && cnt_eq),
                "emg_eval_count_bf16: null pointer");
    CountBf16Params P{};
    P.Q = (const uint16_t*)q_bf16; P.ldq = ldq; P.pos_int = pos_int; P.self_ent = self_ent; P.n_rows = n_rows;
    P.ent = (const uint16_t*)ent_bf16; P.n_cand = n_cand; P.ld_ent = ld_ent; P.cand = cand; P.ent_offset = ent_offset;
    P.k_pad = k_pad; P.scale = scale; P.model = model; P.cnt_gt = cnt_gt; P.cnt_eq = cnt_eq;
    return launch_bf16(false, P, (hipStream_t)stream);
}

extern "C" int emg_eval_scores_dense_bf16(int model, const void* q_bf16, int64_t ldq, int64_t n_rows,
                                          const void* ent_bf16, int64_t n_cand, int64_t ld_ent, const int32_t* cand,
                                          int32_t k_pad, float scale, float* S, int64_t lds, void* stream) {
    EMG_REQUIRE((n_rows == 0 || n_cand == 0) || (q_bf16 && ent_bf16 && S && lds >= n_cand),
                "emg_eval_scores_dense_bf16: bad arguments");
    CountBf16Params P{};
    P.Q = (const uint16_t*)q_bf16; P.ldq = ldq; P.n_rows = n_rows; P.ent = (const uint16_t*)ent_bf16; P.n_cand = n_cand;
    P.ld_ent = ld_ent; P.cand = cand; P.k_pad = k_pad; P.scale = scale; P.model = model; P.S = S; P.lds = lds;
    return launch_bf16(true, P, (hipStream_t)stream);
}

extern "C" int emg_eval_filter_count_bf16(int model, const void* q_bf16, int64_t ldq, const int32_t* pos_int,
                                          const int32_t* self_ent, int64_t n_rows, const void* ent_bf16,
                                          int64_t n_local, int64_t ld_ent, int64_t ent_offset, int32_t k_int,
                                          float scale, const int64_t* filt_ptr, const int32_t* filt_idx,
                                          int32_t* fcnt_gt, int32_t* fcnt_eq, void* stream) {
    EMG_REQUIRE(model >= EMG_DISTMULT && model <= EMG_HOLE, "emg_eval_filter_count_bf16: model %d unsupported", model);
    if (n_rows == 0) return EMG_OK;
    EMG_REQUIRE(q_bf16 && pos_int && self_ent && ent_bf16 && filt_ptr && fcnt_gt && fcnt_eq,
                "emg_eval_filter_count_bf16: null pointer");
    hipLaunchKernelGGL(filter_count_bf16_kernel, dim3((unsigned)cdiv(n_rows * 64, 256)), dim3(256), 0,
                       (hipStream_t)stream, model, (const uint16_t*)q_bf16, ldq, pos_int, self_ent, n_rows,
                       (const uint16_t*)ent_bf16, n_local, ld_ent, ent_offset, (int)k_int, scale, filt_ptr, filt_idx,
                       fcnt_gt, fcnt_eq);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}
