// emg_rank_bf16.hip — bf16 MFMA variant of the 1-vs-all count kernel (DistMult / ComplEx / HolE).
//
// Same contraction and the same fused compare-and-count epilogue as emg_rank.hip::count_mfma_kernel, but the
// query rows and the entity table are bf16 (round-to-nearest-even copies made by emg_to_bf16) and the inner
// product runs on v_mfma_f32_32x32x16_bf16 (16x the f32-MFMA rate, half the bytes).  NOT a parity mode:
// bf16 inputs carry ~3 significant digits while ranks compare int32(score*1e5), so ranks agree with the exact
// f32 path only statistically (tests/test_hip_kernels.py::test_bf16_rank_agreement reports the rate).
//
// The one comparison that matters structurally — the positive against ITSELF — stays exact: the positive's
// comparison integer is produced by this same kernel (MODE 2, "diag": B rows = the true entities of the tile's
// query rows), i.e. by the identical MFMA k-order, so in the count pass the true entity always lands on
// `ci == pos` (one tie), exactly like the f32 path where it ties by construction.
//
// Tile: 128 query rows x 128 entities x BK=64 per step; 4 waves (2x2), each 64x64 = 2x2 MFMA 32x32x16.
// LDS rows are 128 B (64 bf16) = eight 16-byte slots; a lane reads its 8-element fragment with ds_read_b128;
// the slot index is XOR-swizzled with (row>>1)&7 so the 16 lanes of a b128 lane group hit 16 distinct slots of
// the two 256-byte bank rows they span (conflict-free) — the writer applies the same XOR.  Global loads are
// register-staged one k-slice ahead (also across the tile boundary, so the epilogue hides the first slice of
// the next tile); out-of-range rows are CLAMPED (never predicated) and masked in the epilogue.
// Row storage contract: ld >= round_up(k_pad, 64) elements (k_pad itself is a multiple of 32; a trailing
// half slice is loaded but not multiplied).
#include "emg_common.hpp"

namespace emg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct CountBf16Params {
    const uint16_t* Q; int64_t ldq; const int32_t* pos_int; const int32_t* self_ent; int64_t n_rows;
    const uint16_t* ent; int64_t n_cand; int64_t ld_ent; const int32_t* cand; int64_t ent_offset;
    int32_t k_pad; float scale; int32_t model;
    int32_t* cnt_gt; int32_t* cnt_eq;
    float* S; int64_t lds;
    int32_t* pos_out;
    int64_t n_qb; int64_t n_cb; int64_t n_tiles; int32_t tiles_per_chunk;
};

constexpr int HBM_ = 128, HBN_ = 128, HBK_ = 64;
enum { BF_COUNT = 0, BF_DENSE = 1, BF_DIAG = 2 };

__device__ __forceinline__ int lds_slot_off(int row, int slot) {  // byte offset of a 16-byte slot
    return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4);
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void count_mfma_bf16_kernel(const CountBf16Params P) {
    __shared__ __attribute__((aligned(16))) unsigned char As[HBM_ * 128];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[HBN_ * 128];
    __shared__ int pos_s[HBM_];

    int64_t qb, cb;
    if constexpr (MODE == BF_DIAG) {
        qb = blockIdx.x; cb = 0;
    } else {
        const int64_t id = blockIdx.x;
        const int64_t xcd = id & 7, slot_id = id >> 3;
        qb = slot_id % P.n_qb;
        cb = xcd + 8 * (slot_id / P.n_qb);
        if (cb >= P.n_cb) return;
    }

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int lrow = tid >> 3, lslot = tid & 7;  // loader: rows lrow + 32 r; 16-byte slot lslot of the 128-byte k-slice
    const int l31 = lane & 31, lhi = lane >> 5;

    if constexpr (MODE == BF_COUNT) {
        if (tid < HBM_) {
            const int64_t qr = qb * HBM_ + tid;
            pos_s[tid] = qr < P.n_rows ? P.pos_int[qr] : 0x7fffffff;
        }
    }

    const uint16_t* arow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t qr = min(qb * HBM_ + lrow + 32 * r, P.n_rows - 1);
        arow[r] = P.Q + qr * P.ldq + 8 * lslot;
    }
    const uint16_t* brow[4];
    auto point_b = [&](int64_t tile) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int64_t erow;
            if constexpr (MODE == BF_DIAG) {
                erow = P.self_ent[min(qb * HBM_ + lrow + 32 * r, P.n_rows - 1)];
            } else {
                const int64_t el = min(tile * HBN_ + lrow + 32 * r, P.n_cand - 1);
                erow = P.cand ? (int64_t)P.cand[el] : el;
            }
            brow[r] = P.ent + erow * P.ld_ent + 8 * lslot;
        }
    };

    unsigned cnt[2][16];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) cnt[a][r] = 0u;

    const bool hole = P.model == EMG_HOLE;
    const int64_t tile0 = MODE == BF_DIAG ? 0 : cb * P.tiles_per_chunk;
    const int64_t tile1 = MODE == BF_DIAG ? 1 : min(tile0 + (int64_t)P.tiles_per_chunk, P.n_tiles);

    u32x4 av[4], bv[4];
    point_b(tile0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        av[r] = *reinterpret_cast<const u32x4*>(arow[r]);
        bv[r] = *reinterpret_cast<const u32x4*>(brow[r]);
    }
    for (int64_t tile = tile0; tile < tile1; ++tile) {
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

        for (int k0 = 0; k0 < P.k_pad; k0 += HBK_) {
            __syncthreads();  // previous step's LDS reads are done
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                *reinterpret_cast<u32x4*>(As + lds_slot_off(lrow + 32 * r, lslot)) = av[r];
                *reinterpret_cast<u32x4*>(Bs + lds_slot_off(lrow + 32 * r, lslot)) = bv[r];
            }
            __syncthreads();
            // register-staged prefetch of the NEXT slice (next tile's first slice at the end of this one)
            if (k0 + HBK_ < P.k_pad) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    av[r] = *reinterpret_cast<const u32x4*>(arow[r] + k0 + HBK_);
                    bv[r] = *reinterpret_cast<const u32x4*>(brow[r] + k0 + HBK_);
                }
            } else if (tile + 1 < tile1) {
                point_b(tile + 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    av[r] = *reinterpret_cast<const u32x4*>(arow[r]);
                    bv[r] = *reinterpret_cast<const u32x4*>(brow[r]);
                }
            }
            const int nks = min(4, (P.k_pad - k0) >> 4);  // K=16 MFMA steps in this slice (2 on a trailing half slice)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks >= nks) break;
                const int slot = 2 * ks + lhi;  // lane holds k = 16*ks + 8*(lane>>5) .. +8
                bf16x8 a[2], b[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    a[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(As + lds_slot_off(wr * 64 + t * 32 + l31, slot)));
                    b[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Bs + lds_slot_off(wc * 64 + t * 32 + l31, slot)));
                }
#pragma unroll
                for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                    for (int tb = 0; tb < 2; ++tb)
                        acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
            }
        }
        // epilogue: D[row][col]: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
        if constexpr (MODE == BF_COUNT) {
            const bool full = (tile + 1) * HBN_ <= P.n_cand;  // block-uniform
#pragma unroll
            for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                for (int tb = 0; tb < 2; ++tb) {
                    const bool cok = full || tile * HBN_ + wc * 64 + tb * 32 + l31 < P.n_cand;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[ta][tb][r];
                        if (hole) v = v * P.scale;
                        const int ci = (int)(v * 100000.0f);
                        const int p = pos_s[wr * 64 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi];
                        const unsigned inc = (ci > p ? 1u : 0u) + (ci == p ? 0x10000u : 0u);
                        cnt[ta][r] += cok ? inc : 0u;
                    }
                }
        } else {
#pragma unroll
            for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                for (int tb = 0; tb < 2; ++tb) {
                    const int cl = wc * 64 + tb * 32 + l31;
                    const int64_t ecol = tile * HBN_ + cl;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = wr * 64 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                        const int64_t qr = qb * HBM_ + rl;
                        float v = acc[ta][tb][r];
                        if (hole) v = v * P.scale;
                        if constexpr (MODE == BF_DENSE) {
                            if (ecol < P.n_cand && qr < P.n_rows) P.S[qr * P.lds + ecol] = v;
                        } else {
                            if (cl == rl && qr < P.n_rows) P.pos_out[qr] = (int)(v * 100000.0f);
                        }
                    }
                }
        }
    }
    if constexpr (MODE == BF_COUNT) {
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

// the true entity of every query row (the row's own candidate)
__global__ void self_ent_kernel(const int32_t* __restrict__ test, int64_t n_q, int64_t n_rows, int side_mode,
                                int32_t* __restrict__ self_ent) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    int64_t qi; bool obj;
    if (side_mode == EMG_EVAL_S) { qi = r; obj = false; }
    else if (side_mode == EMG_EVAL_O) { qi = r; obj = true; }
    else { obj = r < n_q; qi = r < n_q ? r : r - n_q; }
    self_ent[r] = obj ? test[3 * qi + 2] : test[3 * qi + 0];
}

// filter counts on the bf16-rounded operands; the row's own entity is one tie by construction (see header)
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
        if (e < 0 || e >= n_local) continue;
        if (gidx == self) { eq += 1; continue; }
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

static int launch_bf16(int mode, CountBf16Params& P, hipStream_t st) {
    EMG_REQUIRE(P.model >= EMG_DISTMULT && P.model <= EMG_HOLE, "bf16 eval: model %d is not a contraction (TransE stays f32 VALU)", P.model);
    const int64_t ld_min = (P.k_pad + HBK_ - 1) / HBK_ * HBK_;
    EMG_REQUIRE(P.k_pad > 0 && P.k_pad % 32 == 0 && P.ldq >= ld_min && P.ld_ent >= ld_min,
                "bf16 eval: k_pad must be a multiple of 32 and rows stored with ld >= round_up(k_pad, %d) (k_pad=%d ldq=%lld ld=%lld)",
                HBK_, P.k_pad, (long long)P.ldq, (long long)P.ld_ent);
    EMG_REQUIRE(P.ldq % 8 == 0 && P.ld_ent % 8 == 0 && aligned16(P.Q) && aligned16(P.ent), "bf16 eval: rows must be 16-byte aligned");
    if (P.n_rows == 0 || P.n_cand == 0) return EMG_OK;
    P.n_qb = cdiv(P.n_rows, HBM_);
    P.n_tiles = cdiv(P.n_cand, HBN_);
    P.tiles_per_chunk = mode == BF_DENSE ? 4 : 32;
    P.n_cb = cdiv(P.n_tiles, P.tiles_per_chunk);
    const int64_t blocks = mode == BF_DIAG ? P.n_qb : 8 * P.n_qb * cdiv(P.n_cb, 8);
    EMG_REQUIRE(blocks < ((int64_t)1 << 31), "bf16 eval: grid too large");
    if (mode == BF_DENSE) hipLaunchKernelGGL(count_mfma_bf16_kernel<BF_DENSE>, dim3((unsigned)blocks), dim3(256), 0, st, P);
    else if (mode == BF_DIAG) hipLaunchKernelGGL(count_mfma_bf16_kernel<BF_DIAG>, dim3((unsigned)blocks), dim3(256), 0, st, P);
    else hipLaunchKernelGGL(count_mfma_bf16_kernel<BF_COUNT>, dim3((unsigned)blocks), dim3(256), 0, st, P);
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
    hipLaunchKernelGGL(self_ent_kernel, dim3((unsigned)cdiv(n_rows, 256)), dim3(256), 0, (hipStream_t)stream, test_spo,
                       n_q, n_rows, side_mode, self_ent);
    EMG_LAUNCH_CHECK();
    // the positive's integer comes from the SAME MFMA arithmetic the count pass uses (diag mode, see header)
    CountBf16Params P{};
    P.Q = (const uint16_t*)q_bf16; P.ldq = ldq; P.self_ent = self_ent; P.n_rows = n_rows;
    P.ent = (const uint16_t*)ent_bf16; P.n_cand = HBN_; P.ld_ent = ld_ent; P.k_pad = (k_int + 31) / 32 * 32;
    P.scale = scale; P.model = model; P.pos_out = pos_int;
    return launch_bf16(BF_DIAG, P, (hipStream_t)stream);
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
    return launch_bf16(BF_COUNT, P, (hipStream_t)stream);
}

extern "C" int emg_eval_scores_dense_bf16(int model, const void* q_bf16, int64_t ldq, int64_t n_rows,
                                          const void* ent_bf16, int64_t n_cand, int64_t ld_ent, const int32_t* cand,
                                          int32_t k_pad, float scale, float* S, int64_t lds, void* stream) {
    EMG_REQUIRE((n_rows == 0 || n_cand == 0) || (q_bf16 && ent_bf16 && S && lds >= n_cand),
                "emg_eval_scores_dense_bf16: bad arguments");
    CountBf16Params P{};
    P.Q = (const uint16_t*)q_bf16; P.ldq = ldq; P.n_rows = n_rows; P.ent = (const uint16_t*)ent_bf16; P.n_cand = n_cand;
    P.ld_ent = ld_ent; P.cand = cand; P.k_pad = k_pad; P.scale = scale; P.model = model; P.S = S; P.lds = lds;
    return launch_bf16(BF_DENSE, P, (hipStream_t)stream);
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
