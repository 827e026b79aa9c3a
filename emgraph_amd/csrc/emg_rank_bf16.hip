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
#include <atomic>
#include <type_traits>
#include <utility>

namespace emg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct CountBf16Params {
    const uint16_t* Q; int64_t ldq; const int32_t* pos_int; const int32_t* self_ent; int64_t n_rows;
    const uint16_t* ent; int64_t n_cand; int64_t ld_ent; const int32_t* cand; int64_t ent_offset;
    int32_t k_pad; float scale; int32_t model;
    int32_t need;  // 0: both counters; 1: cnt_gt receives #(>=) only; 2: cnt_gt receives #(>) only (cnt_eq untouched)
    int32_t k16;  // ceil(k_int / 16): MFMA k-steps that hold real data (v3)
    int32_t qs;  // v2: LDS query-row stride in bytes (odd multiple of 64)
    float cmul;  // score -> comparison integer: int(acc * cmul), cmul = 1e5 (* 2/k for HolE), one rounding
    int32_t* cnt_gt; int32_t* cnt_eq;
    float* S; int64_t lds;
    int32_t* pos_out;
    int64_t n_qb; int64_t n_cb; int64_t n_tiles; int32_t tiles_per_chunk;
    // prefilter mode (exact ranks at MFMA speed, see count_mfma_bf16_v3_kernel MODE 2)
    const float* band;         // per query row: rigorous bound on |bf16 accumulator - exact f32 chain| over all candidates
    const float* thr_direct;   // or (TransE-L2 through the augmented contraction): the two accumulator thresholds themselves,
                               //   [0, n_rows): counted when acc >= it; [n_rows, 2 n_rows): emitted when acc >= it (and not counted)
    uint64_t* pairs;           // (row << 32 | global entity id) of the candidates the bound cannot decide:
    uint32_t* pair_count;      //   wave w of the grid owns pairs[w * pair_cap ...), pair_count[w] = how many it wrote
    uint32_t pair_cap;         //   (no atomics: a returning atomic per emission cost more than the MFMAs of the tile);
    uint32_t n_segments;       //   pair_count[n_segments] != 0: some wave ran out of room
    int32_t ties;              // prefilter, MODE 4: candidates PROVEN to tie with the positive are counted into cnt_eq (see the kernel)
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
                        const int ci = (int)(acc[ta][tb][r] * P.cmul);
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
                        const float v = acc[ta][tb][r];
                        if constexpr (MODE == BF_DENSE) {
                            if (ecol < P.n_cand && qr < P.n_rows) P.S[qr * P.lds + ecol] = hole ? v * P.scale : v;
                        } else {
                            if (cl == rl && qr < P.n_rows) P.pos_out[qr] = (int)(v * P.cmul);
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

// ---------------------------------------------------------------------------------------------------------
// v2 count kernel: query-stationary, LDS-DMA streamed entities.
//
// One workgroup = 8 waves (2 x 4, each 64 query rows x 64 entities) owns 128 query rows for its whole life:
// their bf16 rows (<= 128 x 832 B) are copied into LDS ONCE, so the k-loop streams ONLY the entity table.
// Entity slices (256 rows x 32 k = 16 KB) arrive by global_load_lds_dwordx4 into a 3-slot LDS ring, issued two
// k-steps ahead; the loop has ONE raw s_barrier per k-step and a counted `s_waitcnt vmcnt(2)` (never 0 in
// steady state), and the stream runs straight across tile boundaries.  LDS-DMA writes lane-linearly, so the
// bank swizzle is applied on the SOURCE side (which 16-byte k-chunk a lane fetches) and again on the read.
// All LDS lives in one dynamic array (a second __shared__ object makes hipcc drain vmcnt before every ds_read).
//
// Epilogue: compare against per-row float thresholds instead of converting every score: with c = cmul > 0,
// int(fl(v*c)) > p  <=>  fl(v*c) >= G(p)  <=>  v >= g*, where g* is the smallest float whose rounded product
// reaches G (fl(v*c) is monotone in v) — found once per row per block.  Same for >=.  Results are IDENTICAL to
// the v1 kernel's integer compare for every non-NaN score (tests/test_hip_kernels.py compares the two).
constexpr int V2_BM = 128, V2_BN = 256, V2_NS = 3, V2_STAGE = V2_BN * 64;  // bytes per ring slot (32 bf16 per row)
constexpr int V2_KPAD_MAX = 416;

__device__ __forceinline__ void glds16(const uint16_t* gsrc, unsigned lds_base) {
    // wave-uniform LDS base (M0) + lane*16 <- 16 bytes from each lane's own global address
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)(uintptr_t)lds_base, 16, 0, 0);
}

// thresholds in the ACCUMULATOR domain (see header): smallest v with fl(v*c) >= T
__device__ __forceinline__ float acc_threshold(float T, float c) {
#pragma clang fp contract(off)
    float a = T / c;
#pragma unroll 1
    for (int i = 0; i < 4; ++i) { const float d = nextafterf(a, -INFINITY); if (d * c >= T) a = d; }
#pragma unroll 1
    for (int i = 0; i < 4; ++i) { if (!(a * c >= T)) a = nextafterf(a, INFINITY); }
    return a;
}
__device__ __forceinline__ float gt_threshold(int p) {   // int(x) > p  <=>  x >= G
    const float pf = (float)p;
    return (p >= 0 && p < (1 << 24)) ? (float)(p + 1) : nextafterf(pf, INFINITY);
}
__device__ __forceinline__ float ge_threshold(int p) {   // int(x) >= p  <=>  x >= E
    const float pf = (float)p;
    return (p > 0 || p < -(1 << 24)) ? pf : nextafterf((float)(p - 1), INFINITY);
}

__global__ __launch_bounds__(512, 1) void count_mfma_bf16_v2_kernel(const CountBf16Params P) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];  // ring | Q rows | thresholds | counts
    const int qs = P.qs;                                                     // LDS query row stride, bytes
    unsigned char* Qs = smem + V2_NS * V2_STAGE;
    float* thr_s = reinterpret_cast<float*>(Qs + V2_BM * qs);  // [0,128): gt thresholds, [128,256): ge thresholds
    unsigned* cnt_s = reinterpret_cast<unsigned*>(thr_s + 2 * V2_BM);

    const int64_t id = blockIdx.x;
    const int64_t xcd = id & 7, slot_id = id >> 3;
    const int64_t qb = slot_id % P.n_qb;
    const int64_t cb = xcd + 8 * (slot_id / P.n_qb);
    if (cb >= P.n_cb) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int xq = (l31 >> 2) & 3;

    // ---- one-time: per-row thresholds -> LDS, query rows -> LDS (swizzled) ------------------------------
    if (tid < 2 * V2_BM) {
        const int row = tid & (V2_BM - 1);
        const int64_t qr = qb * V2_BM + row;
        float t = INFINITY;  // rows past the end count nothing
        if (qr < P.n_rows) {
            const int p = P.pos_int[qr];
            t = acc_threshold(tid < V2_BM ? gt_threshold(p) : ge_threshold(p), P.cmul);
        }
        thr_s[tid] = t;
    } else if (tid < 3 * V2_BM) {
        cnt_s[tid - 2 * V2_BM] = 0u;
    }
    {
        const int nslots = P.k_pad >> 3;
        for (int i = tid; i < V2_BM * nslots; i += 512) {
            const int row = i / nslots, slot = i - row * nslots;
            const int64_t qr = min(qb * V2_BM + row, P.n_rows - 1);
            const u32x4 v = *reinterpret_cast<const u32x4*>(P.Q + qr * P.ldq + 8 * slot);
            *reinterpret_cast<u32x4*>(Qs + row * qs + ((slot ^ ((row >> 2) & 3)) << 4)) = v;
        }
    }
    __syncthreads();

    unsigned cnt[16];  // per accumulator register: four packed 8-bit counters (<= 2 x tiles_per_chunk = 32 each)
#pragma unroll
    for (int r = 0; r < 16; ++r) cnt[r] = 0u;

    // ---- LDS-DMA producer state: this lane's two 16-byte pieces of every ring slot -----------------------
    const int KS = P.k_pad >> 5;  // k-steps per tile
    const int64_t tile0 = cb * P.tiles_per_chunk;
    const int ntile = (int)(min(tile0 + (int64_t)P.tiles_per_chunk, P.n_tiles) - tile0);
    const int S = ntile * KS;
    int prow[2], pslot[2];
    const uint16_t* gp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        prow[j] = (2 * wave + j) * 16 + (lane >> 2);
        pslot[j] = (lane & 3) ^ ((prow[j] >> 2) & 3);
    }
    // Tiles are visited in a per-block ROTATED order: the workgroups that share this entity chunk (same XCD,
    // consecutive qb) would otherwise stream identical addresses in lockstep and all wait on the same HBM lines.
    const int rot = (int)(qb % ntile);
    int itile = rot;  // producer's tile (chunk-local)
    int ik = 0, islot = 0;
    auto point = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t el = min((tile0 + itile) * V2_BN + prow[j], P.n_cand - 1);
            gp[j] = P.ent + el * P.ld_ent + 8 * pslot[j];
        }
    };
    const unsigned ring0 = (unsigned)(uintptr_t)smem + (unsigned)(2 * wave) * 1024u;
    auto issue = [&]() {
        const unsigned base = __builtin_amdgcn_readfirstlane(ring0 + (unsigned)islot * V2_STAGE);
        glds16(gp[0] + ik * 32, base);
        glds16(gp[1] + ik * 32, base + 1024u);
        islot = islot == V2_NS - 1 ? 0 : islot + 1;
        if (++ik == KS) { ik = 0; itile = itile + 1 == ntile ? 0 : itile + 1; point(); }
    };
    point();
    issue();
    if (S > 1) issue();
    if (S > 2) issue();

    // ---- consumer addressing ----------------------------------------------------------------------------
    int aoff[2], boff[2], sl[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        aoff[t] = (wr * 64 + t * 32 + l31) * qs;
        boff[t] = (wc * 64 + t * 32 + l31) * 64;
        sl[t] = ((2 * t + lhi) ^ xq) << 4;  // t = MFMA k-half of the 32-wide step
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    int kk = 0, cslot = 0;
    int ctile = rot;  // consumer's tile (chunk-local)

    // Fragments for step s+1 are read while the MFMAs of step s execute: two register sets swap roles every
    // step, which is why the loop body is written out twice.  Schedule of one step (s):
    //   vmcnt: my pieces of stage s+1 landed -> s_barrier: everyone's landed AND every wave has finished reading
    //   stage s (its fragments sit in registers) -> refill that slot with stage s+3  |  ds_read the fragments of
    //   stage s+1 and the next query k-slice (they land under the MFMAs)  |  8 MFMAs(s)  |  (epilogue of a tile)
    bf16x8 A0[2][2], A1[2][2], B0[2][2], B1[2][2];
    auto load_frags = [&](bf16x8 (&af)[2][2], bf16x8 (&bf)[2][2], int k_idx, int slot) {
        const unsigned char* Bst = smem + slot * V2_STAGE;
        const unsigned char* Ak = Qs + k_idx * 64;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bf[ks][t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Bst + boff[t] + sl[ks]));
                af[ks][t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Ak + aoff[t] + sl[ks]));
            }
    };
    // stage 0
    if (S > 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (S > 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    load_frags(A0, B0, 0, 0);

    auto step = [&](int s, bf16x8 (&acur)[2][2], bf16x8 (&bcur)[2][2], bf16x8 (&anxt)[2][2], bf16x8 (&bnxt)[2][2]) {
        const int knext = kk + 1 == KS ? 0 : kk + 1;
        cslot = cslot == V2_NS - 1 ? 0 : cslot + 1;
        // (the last step runs the same sequence on a stale slot: no branch around the ds_reads, so hipcc can
        // count them — a join would force lgkmcnt(0) in front of the MFMAs)
        // compiler-visible lgkmcnt(0): the CURRENT fragments (read a whole MFMA group ago) are retired here, so
        // hipcc does not put an lgkmcnt(0) between the reads below and the MFMAs that do not depend on them
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (s + 2 < S) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s + 3 < S) issue();
        load_frags(anxt, bnxt, knext, cslot);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                for (int tb = 0; tb < 2; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[ks][ta], bcur[ks][tb], acc[ta][tb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        kk = knext;
        if (knext == 0) {  // the tile's last k-step: compare-and-count, then clear the accumulators
            auto epilogue = [&](auto FULL) {
#pragma unroll
                for (int ta = 0; ta < 2; ++ta) {
                    float thrG[16], thrE[16];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r0 = wr * 64 + ta * 32 + 8 * j + 4 * lhi;  // rows of accumulator regs 4j..4j+3
                        const f32x4 g4 = *reinterpret_cast<const f32x4*>(thr_s + r0);
                        const f32x4 e4 = *reinterpret_cast<const f32x4*>(thr_s + V2_BM + r0);
#pragma unroll
                        for (int i = 0; i < 4; ++i) { thrG[4 * j + i] = g4[i]; thrE[4 * j + i] = e4[i]; }
                    }
                    const unsigned one = ta ? 0x10000u : 1u;  // packed 8-bit fields: gt0 | eq0<<8 | gt1<<16 | eq1<<24
#pragma unroll
                    for (int tb = 0; tb < 2; ++tb) {
                        if constexpr (FULL.value) {
                            unsigned long long tie = 0ull;  // lanes holding a score equal to the positive's (rare)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const float v = acc[ta][tb][r];
                                const unsigned long long mg = __builtin_amdgcn_fcmpf(v, thrG[r], 3);  // v >= thr
                                const unsigned long long me = __builtin_amdgcn_fcmpf(v, thrE[r], 3);
                                cnt[r] += (v >= thrG[r]) ? one : 0u;
                                tie |= mg ^ me;
                            }
                            if (tie) {
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const float v = acc[ta][tb][r];
                                    cnt[r] += (v >= thrE[r] && !(v >= thrG[r])) ? (one << 8) : 0u;
                                }
                            }
                        } else {
                            const bool cok = (tile0 + ctile) * V2_BN + wc * 64 + tb * 32 + l31 < P.n_cand;
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const float v = acc[ta][tb][r];
                                const bool gt = v >= thrG[r], ge = v >= thrE[r];
                                cnt[r] += (cok && gt) ? one : 0u;
                                cnt[r] += (cok && ge && !gt) ? (one << 8) : 0u;
                            }
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[ta][tb][r] = 0.f;
                    }
                }
            };
            if ((tile0 + ctile + 1) * V2_BN <= P.n_cand) epilogue(std::true_type{});  // block-uniform
            else epilogue(std::false_type{});
            ctile = ctile + 1 == ntile ? 0 : ctile + 1;
        }
    };
    for (int s = 0; s < S; s += 2) {
        step(s, A0, B0, A1, B1);
        if (s + 1 >= S) break;
        step(s + 1, A1, B1, A0, B0);
    }

    // ---- block reduction: lanes -> rows (shuffles), 4 column-waves -> LDS, one global atomic per row ------
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned pk = cnt[r] >> (16 * ta);
            unsigned c = (pk & 0xffu) | ((pk & 0xff00u) << 8);  // gt | eq<<16, wide enough for the lane sum
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
            if (l31 == 0 && c) atomicAdd(&cnt_s[wr * 64 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi], c);
        }
    __syncthreads();
    if (tid < V2_BM) {
        const int64_t qr = qb * V2_BM + tid;
        const unsigned c = cnt_s[tid];
        if (qr < P.n_rows) {
            if (c & 0xffffu) atomicAdd(&P.cnt_gt[qr], (int)(c & 0xffffu));
            if (c >> 16) atomicAdd(&P.cnt_eq[qr], (int)(c >> 16));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// v3 count kernel: query fragments in REGISTERS, LDS holds nothing but a deep LDS-DMA ring of entity slices.
//
// Measured on v2: one 16 KB LDS-DMA fill takes ~1.1-1.3 us issue->landed under load, so with the two fills
// v2's ring can keep in flight a CU streams ~25 GB/s and the MFMAs wait (33 % MFMA busy).  v3 removes the
// query tile from LDS: wave w keeps the A fragments of ITS 32 query rows for the whole k range in VGPRs
// (NQ x 4 registers, NQ = ceil(k_int/16) MFMA k-steps; the k-loop is fully unrolled so the indices are
// static, and a trailing all-padding k-step is never multiplied).  A workgroup is 8 waves = 256 query rows
// (twice v2's reuse of every streamed entity byte) against 128-entity tiles, and all of LDS is a ring of
// 128 KB of entity slices (128 entities x 16*SQ k): all but two of them in flight.
// Per slice and wave: 4*SQ ds_read_b128 (every wave reads the whole slice), 4*SQ MFMA 32x32x16 (32 x 128
// outputs), ONE raw s_barrier, SQ/2 global_load_lds_dwordx4, a constant counted `s_waitcnt vmcnt`: the loop
// body is branch-free (past the end the producer keeps issuing wrapped, never-read fills so the count stays
// constant), which lets hipcc count the ds_reads instead of draining them.  B fragments are double-buffered
// by k-step: each is read under the previous k-step's MFMAs.  Epilogue and thresholds as in v2; each wave
// owns its rows, so the block reduction is a lane shuffle and one global atomic per row.
#ifndef V3_ABLATE
#define V3_ABLATE 0  // timing experiments only (wrong results): 1 no DMA wait, 2 no barrier, 4 no DMA issue
#endif
constexpr int V3_BM = 256, V3_BN = 128, V3_RING = 128 * 1024;

// ONE: only one comparison per score (P.need = 1: count `>=`, the 'worst' strategy's only input; 2: count `>`,
// 'best'): half the epilogue's VALU work.  Both counters are produced only when the caller needs ties ('middle').
// MODE 2 (PREFILTER): exact ranks at MFMA speed.  The caller passes, per query row, a rigorous bound `band` on the
// distance between this kernel's accumulator (bf16-rounded operands) and the exact f32 chain of the parity path.
// A candidate whose accumulator clears the `>` threshold by more than the band is counted here (cnt_gt); one that
// misses the `>=` threshold by more than the band is dropped; the few in between are EMITTED as (row, entity) pairs
// and re-scored exactly by emg_eval_rescore_pairs — so the counters, and the ranks, equal the exact path's bit for bit.
// WAVES = 8: 256 query rows per workgroup, two waves per SIMD (<= 256 registers: NQ <= 25, k_int <= 400).  WAVES = 4 (the
// prefilter above k_int = 400): 128 query rows, ONE wave per SIMD with the whole 512-register file — NQ up to 50 query
// fragments (200 registers) beside the accumulators; every streamed entity byte is used by half as many rows.
//
// MODE 3 (round 6): the prefilter WITHOUT emission.  Measured on MODE 2 (profiles/r6_b_eval_kernel_stats.md): the same MFMAs take
// 5.7 ms with one counter, 7.2 ms as the prefilter of a trained-like table (0.08 % undecided) and 9.0 ms on random positives
// (0.72 %): the per-group scalar branches and the emission — 37 % of the accumulator registers hold an undecided candidate on
// random positives — stall the wave, and its pair stores sit in the same counted queue as the LDS-DMA fills.  Here a tile's
// epilogue is branch-free: per value two compares, each shifted into a 32-bit word of the lane by its own carry
// (v_cmp + v_addc w, w, w: w = 2 w + bit), then count += popcount(greater word), undecided = greater-or-equal word XOR greater
// word, and ONE 8-byte store per lane and tile of the undecided BITMAP — into the wave's own segment of the pair buffer
// (64 lanes x 8 B x <= 32 tiles = the segment's 16 KB at 2048 entries).  prefilter_compact_kernel then turns each segment's
// bitmap into the (row, entity) pairs the re-scoring kernels read, in place (a wave per segment reads the 16 KB, then writes).
//
// MODE 4 (round 6): MODE 3 that also PROVES TIES.  The reference compares int32(score * 1e5) (EmbeddingModel.py:2010-2014): a table
// whose scores are small against 1e-5 — a freshly initialised model, the first epochs of a fit, what early stopping evaluates —
// has EVERY candidate tie with the positive, and modes 2 / 3 call every tie undecided (its accumulator lies between the `>` and the
// `>=` threshold): the pair buffer overflows and the exact kernel does the tile (47 ms per 8192 x 1M pass).  But a tie can be
// decided like the other two outcomes: the integer cell of the positive is [E, G) in the accumulator domain, and an accumulator in
// [E + band, G - band) belongs to a score inside the cell whatever the rounding.  Four thresholds per row — E - b, E + b, G - b, G + b
// —, four compare-and-shift pairs per value; greater = above G + b, EQUAL = in [E + b, G - b) (counted into cnt_eq), undecided = the
// two bands around E and G; where the band is wider than the cell (E + b > G - b: any table with scores of order one) the equal zone is
// empty and the two bands merge into mode 3's.  Eight VALU instructions per value instead of four: the host runs this form only
// where mode 3's probe found too many undecided candidates (ranking.py).
template <int NQ, int SQ, int MODE, int WAVES = 8>  // NQ: 16-wide k-steps per row; SQ: k-steps per slice (2 or 4); MODE 0 both | 1 one | 2 prefilter | 3 prefilter, bitmap | 4 ... proving ties
__global__ __launch_bounds__(64 * WAVES, 1) void count_mfma_bf16_v3_kernel(const CountBf16Params P) {
    constexpr int V3_BM = 32 * WAVES;           // query rows per workgroup
    constexpr bool ONE = MODE == 1;
    constexpr bool PRE = MODE >= 2;
    constexpr bool BMP = MODE >= 3;
    constexpr bool TIES = MODE == 4;
    constexpr int RB = SQ * 32;                 // slice row bytes (64 / 128)
    constexpr int SPR = SQ * 2;                 // 16-byte slots per slice row
    constexpr int RPB = 256 / RB;               // rows per 256-byte LDS bank row
    constexpr int STAGE = V3_BN * RB;           // 8 KB / 16 KB
    constexpr int NS = V3_RING / STAGE;         // 16 / 8 slots
    constexpr int G = STAGE / (1024 * WAVES);   // LDS-DMA instructions per wave and slice
    constexpr int D = (NQ + SQ - 1) / SQ;       // slices per tile
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];  // ring | thresholds
    float* thr_s = reinterpret_cast<float*>(smem + V3_RING);                // [0,V3_BM): gt, [V3_BM, 2 V3_BM): ge

    const int64_t id = blockIdx.x;
    const int64_t xcd = id & 7, slot_id = id >> 3;
    const int64_t qb = slot_id % P.n_qb;
    const int64_t cb = xcd + 8 * (slot_id / P.n_qb);
    if (cb >= P.n_cb) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: the wave's LDS ring
    const int l31 = lane & 31, lhi = lane >> 5;                                                      //  base, pair segment and row base too)
    const int64_t n_cand = P.n_cand, ld_ent = P.ld_ent;
    const uint16_t* const ent = P.ent;

    // ---- one-time: thresholds -> LDS, this wave's query fragments -> registers ---------------------------
    {
        const int row = tid & (V3_BM - 1);
        const int64_t qr = qb * V3_BM + row;
        float t = INFINITY;  // rows past the end count nothing
        if (PRE && P.thr_direct != nullptr) {
            if (qr < P.n_rows) t = P.thr_direct[(tid < V3_BM ? 0 : P.n_rows) + qr];
        } else if (qr < P.n_rows) {
            const int p = P.pos_int[qr];
            const bool want_gt = ONE ? P.need == 2 : tid < V3_BM;
            t = acc_threshold(want_gt ? gt_threshold(p) : ge_threshold(p), P.cmul);
            if constexpr (PRE) {
                // widen by the band (+ the few ulps by which the exact path's two roundings, fl(fl(acc*scale)*1e5), can
                // differ from this kernel's single one), rounding outwards
                const float b = P.band[qr] + 1e-6f * fabsf(t) + 1e-30f;
                t = want_gt ? nextafterf(t + b, INFINITY) : nextafterf(t - b, -INFINITY);
            }
        }
        thr_s[tid] = t;
    }
    bf16x8 A[NQ];
    {
        const int64_t qr = min(qb * V3_BM + wave * 32 + l31, P.n_rows - 1);
        const uint16_t* qp = P.Q + qr * P.ldq + 8 * lhi;
#pragma unroll
        for (int q = 0; q < NQ; ++q) A[q] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(qp + q * 16));
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // compiler-visible vmcnt(0): from here on only LDS-DMA is in the VMEM queue
    __syncthreads();

    // TRANSPOSED products (round 6): the entity fragment is the MFMA's A operand, the query fragment its B operand, so an
    // accumulator tile is [entity][query row] and lane l holds query row l & 31 in ALL of its registers (16 entities each:
    // 8 (r >> 2) + (r & 3) + 4 (l >> 5) of the 32-entity block).  One lane, one row: the row's two thresholds are two registers
    // (no LDS reads in the epilogue), a counted value is v_cmp + v_addc into ONE 32-bit counter per lane (was: compare, select,
    // add into packed 8-bit counters per register pair), and the lanes l / l + 32 of a row meet in one shuffle at the end.
    // The products a[m][k] b[k][n] and their k order are those of the untransposed form: the same scores bit for bit.
    const float g_l = thr_s[wave * 32 + l31], e_l = thr_s[V3_BM + wave * 32 + l31];
    unsigned cgt = 0u, cge = 0u;   // scores that reached the first threshold; MODE 0: ... the second (ties = cge - cgt)
    [[maybe_unused]] unsigned ucnt = 0u;   // MODE 3: undecided candidates of this lane
    // MODE 4: the INNER thresholds of this lane's row (rows past the end: +inf, nothing is ever equal) and its proven ties
    [[maybe_unused]] float gi_l = INFINITY, ei_l = INFINITY;
    [[maybe_unused]] unsigned ceq_l = 0u;
    if constexpr (TIES) {
        const int64_t qr = qb * V3_BM + wave * 32 + l31;
        if (qr < P.n_rows) {
            const int p = P.pos_int[qr];
            const float tg = acc_threshold(gt_threshold(p), P.cmul), te = acc_threshold(ge_threshold(p), P.cmul);
            const float bg = P.band[qr] + 1e-6f * fabsf(tg) + 1e-30f, be = P.band[qr] + 1e-6f * fabsf(te) + 1e-30f;
            gi_l = nextafterf(tg - bg, -INFINITY);   // an accumulator below it: the score is below the `>` threshold for certain
            ei_l = nextafterf(te + be, INFINITY);    // an accumulator at or above it: the score reaches the `>=` threshold for certain
        }
    }

    // ---- LDS-DMA producer: this lane's 16-byte pieces of every slice -------------------------------------
    const int64_t tile0 = cb * P.tiles_per_chunk;
    const int ntile = (int)(min(tile0 + (int64_t)P.tiles_per_chunk, P.n_tiles) - tile0);
    int prow[G], pslot[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
        const int pos = (G * wave + j) * 64 + lane;  // 16-byte position inside the slice
        prow[j] = pos / SPR;
        pslot[j] = (pos % SPR) ^ ((prow[j] / RPB) & (SPR - 1));
    }
    // Tiles are visited in a per-block ROTATED order (see v2).
    const int rot = (int)(qb % ntile);
    int itile = rot, islot = 0;
    const uint16_t* gp[G];
    auto point = [&]() {
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int64_t el = min((tile0 + itile) * V3_BN + prow[j], n_cand - 1);
            gp[j] = ent + el * ld_ent + 8 * pslot[j];
        }
    };
    const unsigned ring0 = (unsigned)(uintptr_t)smem + (unsigned)(G * wave) * 1024u;
    auto issue = [&](int d) {  // d: slice of the tile (static at every call site)
        const unsigned base = __builtin_amdgcn_readfirstlane(ring0 + (unsigned)islot * STAGE);
#pragma unroll
        for (int j = 0; j < G; ++j) glds16(gp[j] + d * (16 * SQ), base + 1024u * j);
        islot = (islot + 1) & (NS - 1);
        if (d == D - 1) { itile = itile + 1 == ntile ? 0 : itile + 1; point(); }
    };
    point();
    // slices 0 .. NS-2 (the stream simply wraps around the chunk; fills past the last real slice are never read)
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) issue(t % D);

    // ---- consumer ---------------------------------------------------------------------------------------
    const int xq = (l31 / RPB) & (SPR - 1);
    const unsigned char* const bptr = smem + l31 * RB;
    int sl[SQ];
#pragma unroll
    for (int t = 0; t < SQ; ++t) sl[t] = ((2 * t + lhi) ^ xq) << 4;

    f32x16 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

    bf16x8 X[4], Y[4];
    int cslot = 0;
    // prefilter: this wave's private segment of the pair buffer
    unsigned pair_n = 0u, pair_over = 0u;
    uint64_t* const pair_base = PRE ? P.pairs + ((uint64_t)blockIdx.x * (unsigned)WAVES + (unsigned)wave) * P.pair_cap : nullptr;
    auto load_step = [&](bf16x8 (&dst)[4], int slot, int ks) {
        const unsigned char* st = bptr + slot * STAGE + sl[ks];
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
            dst[tb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(st + tb * (32 * RB)));
    };
    auto mma_step = [&](const bf16x8& a, bf16x8 (&bq)[4]) {
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
            // the prefilter's operands are IEEE half (11 significant bits: an 8x narrower error band than bf16, same
            // MFMA rate); the fragments are bit containers, only the instruction differs
            // (entity fragment first: the transposed tile, see the counters above)
            if constexpr (PRE) acc[tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bq[tb]), __builtin_bit_cast(f16x8, a), acc[tb], 0, 0, 0);
            else acc[tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[tb], a, acc[tb], 0, 0, 0);
        }
    };
    __builtin_amdgcn_s_waitcnt(0x0F70 | (G * (NS - 2)));  // slice 0 landed (mine) ...
    __builtin_amdgcn_s_barrier();                          // ... and everyone's
    load_step(X, 0, 0);

    for (int ti = 0; ti < ntile; ++ti) {
        const int ctile = rot + ti >= ntile ? rot + ti - ntile : rot + ti;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int d = q / SQ, ks = q % SQ;                         // static
            const bool last_of_slice = ks == SQ - 1 || q == NQ - 1;    // static
            bf16x8 (&cur)[4] = (q & 1) ? Y : X;
            bf16x8 (&nxt)[4] = (q & 1) ? X : Y;
            if (!last_of_slice) {
                load_step(nxt, cslot, ks + 1);
            } else {
                // next slice: mine landed when at most the G(NS-3) younger fills are still flying; the barrier
                // makes it everyone's, and says every wave is done with the PREVIOUS slice -> refill that slot
#if !(V3_ABLATE & 1)
                __builtin_amdgcn_s_waitcnt(0x0F70 | (G * (NS - 3)));
#endif
#if !(V3_ABLATE & 2)
                __builtin_amdgcn_s_barrier();
#endif
                cslot = (cslot + 1) & (NS - 1);
                load_step(nxt, cslot, 0);
                __builtin_amdgcn_sched_barrier(0);
#if !(V3_ABLATE & 4)
                issue((d + NS - 1) % D);  // after the reads: the DMA issue is slow and would delay them
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
            mma_step(A[q], cur);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (NQ & 1) {  // an odd k-step count leaves the next tile's first fragments in Y: swap roles back
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) X[tb] = Y[tb];
        }
        // ---- prefilter: the (row, entity) pairs of accumulators inside [lo, hi) of their row --------------------
        // The two compares of a value leave LANE MASKS in scalar registers; `undecided` is one s_andn2 of them and the
        // (wave-uniform) test for "any in these four registers" a scalar branch: a tile with nothing to emit costs one
        // compare per value more than the counting epilogue.  A mask that is not empty is emitted on the spot — slot =
        // pairs so far + v_mbcnt of the mask — so no per-lane bitmap, prefix scan or bit loop is ever built.
        const uint32_t row_s = (uint32_t)(qb * V3_BM + wave * 32), col_s = (uint32_t)(P.ent_offset + (tile0 + ctile) * V3_BN);   // scalars
        const unsigned lhi4 = 4u * (unsigned)lhi;
        [[maybe_unused]] auto emit_mask = [&](uint64_t m, int r, int tb) {   // register r of block tb: entity = col0 + 32 tb + 8 (r >> 2) + (r & 3) + 4 lhi, row = row0 + l31
            const unsigned n = (unsigned)__builtin_popcountll(m);
            if (pair_n + n > P.pair_cap) { pair_over = 1u; return; }   // the caller redoes this query tile with the exact kernel
            const unsigned before = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            // (the scalar halves are made opaque: otherwise the lane-dependent sums are hoisted out of the tile loop and
            // held in — spilled — vector registers for an emission that may never come)
            unsigned hi_s = row_s, lo_s = col_s + (unsigned)(32 * tb + 8 * (r >> 2) + (r & 3));
            asm volatile("" : "+s"(hi_s), "+s"(lo_s));
            if (__builtin_amdgcn_inverse_ballot_w64(m))   // low word: entity, high word: query row
                *reinterpret_cast<uint2*>(pair_base + pair_n + before) = make_uint2(lo_s + lhi4, hi_s + (unsigned)l31);
            pair_n = (unsigned)__builtin_amdgcn_readfirstlane((int)(pair_n + n));
        };
        // ---- tile epilogue: compare-and-count, clear -----------------------------------------------------
        auto epilogue = [&](auto FULL) {
            const int64_t ent0 = (tile0 + ctile) * V3_BN + 4 * lhi;   // this lane's first entity of the tile
            if constexpr (BMP) {
                // value i = 16 (tb & 1) + r of word tb >> 1 ends at bit 31 - i (32 shifts: whatever the word held before is gone)
                unsigned wg[2] = {0u, 0u}, we[2] = {0u, 0u};
                [[maybe_unused]] unsigned wgi[2] = {0u, 0u}, wei[2] = {0u, 0u};
#pragma unroll
                for (int tb = 0; tb < 4; ++tb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        // (the table's last, partial tile: a NaN reaches no threshold — one code path, no mask)
                        const float v = (FULL.value || ent0 + 32 * tb + 8 * (r >> 2) + (r & 3) < n_cand) ? acc[tb][r] : __builtin_nanf("");
                        asm volatile("v_cmp_ge_f32 vcc, %2, %3\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                                     "v_cmp_ge_f32 vcc, %2, %4\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc"
                                     : "+v"(wg[tb >> 1]), "+v"(we[tb >> 1]) : "v"(v), "v"(g_l), "v"(e_l) : "vcc");
                        if constexpr (TIES)
                            asm volatile("v_cmp_ge_f32 vcc, %2, %3\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                                         "v_cmp_ge_f32 vcc, %2, %4\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc"
                                         : "+v"(wgi[tb >> 1]), "+v"(wei[tb >> 1]) : "v"(v), "v"(gi_l), "v"(ei_l) : "vcc");
                    }
                // (the thresholds are ordered E - b <= E + b, G - b <= G + b: each inner word is a subset of its outer one)
                unsigned u0 = we[0] ^ wg[0], u1 = we[1] ^ wg[1];   // modes 3: between the outer thresholds
                if constexpr (TIES) {
                    u0 = (we[0] ^ wei[0]) | (wgi[0] ^ wg[0]); u1 = (we[1] ^ wei[1]) | (wgi[1] ^ wg[1]);   // the two bands around E and G
                    ceq_l += (unsigned)__builtin_popcount(wei[0] & ~wgi[0]) + (unsigned)__builtin_popcount(wei[1] & ~wgi[1]);   // [E + b, G - b)
                }
                cgt += (unsigned)__builtin_popcount(wg[0]) + (unsigned)__builtin_popcount(wg[1]);
                ucnt += (unsigned)__builtin_popcount(u0) + (unsigned)__builtin_popcount(u1);
                reinterpret_cast<uint2*>(pair_base)[ctile * 64 + lane] = make_uint2(u0, u1);
#pragma unroll
                for (int tb = 0; tb < 4; ++tb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tb][r] = 0.f;
                return;
            }
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    [[maybe_unused]] uint64_t mu[4], any = 0ull;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float v = acc[tb][4 * j + i];
                        const bool cok = FULL.value || ent0 + 32 * tb + 8 * j + i < n_cand;
                        if constexpr (PRE) {
                            // the two compares leave LANE MASKS in scalar registers: `undecided` is one s_andn2 of them, the count
                            // the first mask as the carry of a v_addc, and "any in these four registers" a scalar branch
                            uint64_t mg, me;
                            if constexpr (FULL.value) {   // (as v4's: hipcc turns the mask back into a select and an add)
                                asm volatile("v_cmp_ge_f32_e64 %1, %3, %4\n\tv_cmp_ge_f32_e64 %2, %3, %5\n\tv_addc_co_u32_e64 %0, vcc, 0, %0, %1"
                                             : "+v"(cgt), "=&s"(mg), "=&s"(me) : "v"(v), "v"(g_l), "v"(e_l) : "vcc");
                            } else {
                                const uint64_t cokm = __builtin_amdgcn_ballot_w64(cok);
                                mg = __builtin_amdgcn_ballot_w64(v >= g_l) & cokm; me = __builtin_amdgcn_ballot_w64(v >= e_l) & cokm;
                                cgt += __builtin_amdgcn_inverse_ballot_w64(mg) ? 1u : 0u;
                            }
                            mu[i] = me & ~mg;
                            any |= mu[i];
                        } else {
                            cgt += (cok && v >= g_l) ? 1u : 0u;
                            if constexpr (!ONE) cge += (cok && v >= e_l) ? 1u : 0u;
                        }
                    }
                    if constexpr (PRE) {
                        // (count here: left to itself the compiler defers the counter updates of a tile to its end and carries
                        // their lane masks there — scalar registers it does not have)
                        asm volatile("" : "+v"(cgt));
                        if (any) {   // wave-uniform
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (mu[i]) emit_mask(mu[i], 4 * j + i, tb);
                        }
                    }
                }
            }
#pragma unroll
            for (int tb = 0; tb < 4; ++tb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tb][r] = 0.f;
        };
        if ((tile0 + ctile + 1) * V3_BN <= n_cand) epilogue(std::true_type{});  // block-uniform
        else epilogue(std::false_type{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may land after this workgroup has left
    if constexpr (BMP) {   // the segment's undecided candidates: the wave's sum (the bitmap holds them whatever their number)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) ucnt += __shfl_xor(ucnt, off, 64);
        // more than the re-scoring pass's segment holds: the caller redoes this query tile exactly, and the segment counts as EMPTY
        // (its words stay a bitmap: nothing may read them as pairs)
        pair_over = ucnt > P.pair_cap ? 1u : 0u;
        pair_n = pair_over ? 0u : ucnt;
    }
    if constexpr (PRE) {
        if (lane == 0) {
            P.pair_count[blockIdx.x * (unsigned)WAVES + (unsigned)wave] = pair_n;
            if (pair_over) atomicOr(P.pair_count + P.n_segments, 1u);
        }
    }

    // ---- rows are private to the wave, a row to the lanes l and l + 32: one shuffle, one global atomic per row and counter ----
    {
        unsigned ceq = TIES ? ceq_l : ((ONE || PRE) ? 0u : cge - cgt);   // (the second threshold is the lower one: every score counted in cgt is in cge)
        cgt += __shfl_xor(cgt, 32, 64);
        ceq += __shfl_xor(ceq, 32, 64);
        const int64_t qr = qb * V3_BM + wave * 32 + l31;
        if (lhi == 0 && qr < P.n_rows) {
            if (cgt) atomicAdd(&P.cnt_gt[qr], (int)cgt);
            if (ceq) atomicAdd(&P.cnt_eq[qr], (int)ceq);
        }
    }
}

// The opt-in to > 64 KB of dynamic LDS is a per-DEVICE function attribute: remember it per device (bit d of `done`)
// so that a process driving several GPUs sets it on each, and so that two host threads may race here harmlessly
// (hipFuncSetAttribute is idempotent; the flag is only ever set after a successful call).
static int allow_full_lds(const void* kernel, std::atomic<uint64_t>& done) {
    int dev = 0;
    EMG_HIP(hipGetDevice(&dev));
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return EMG_OK;
    EMG_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    done.fetch_or(bit, std::memory_order_release);
    return EMG_OK;
}

template <int NQ, int SQ, int MODE, int WAVES = 8>
static int launch_v3(const CountBf16Params& P, int64_t nblk, hipStream_t st) {
    const size_t lds_bytes = (size_t)V3_RING + 2 * (32 * WAVES) * sizeof(float);
    static std::atomic<uint64_t> devices_done{0};  // one flag per template instance and device
    int rc = allow_full_lds((const void*)count_mfma_bf16_v3_kernel<NQ, SQ, MODE, WAVES>, devices_done);
    if (rc != EMG_OK) return rc;
    hipLaunchKernelGGL((count_mfma_bf16_v3_kernel<NQ, SQ, MODE, WAVES>), dim3((unsigned)nblk), dim3(64 * WAVES), lds_bytes, st, P);
    return EMG_OK;
}

// MODE 3's second half: a segment's undecided bitmap -> its (row << 32 | entity) pairs, IN PLACE.  One wave per segment: all of the
// segment's words are in registers (32 x 8 bytes per lane) before the first pair is written over them; per tile a lane's pairs go
// behind those of the lanes below it (wave scan of the popcounts), tiles ascending — the order the re-scoring kernels sweep in.
struct CompactParams {
    uint64_t* pairs; const uint32_t* pair_count; uint32_t pair_cap, n_segments;
    int64_t n_qb, n_cb, n_tiles, ent_offset; int32_t tiles_per_chunk, waves;
    int64_t last32;   // v4 (its stage of 32 entities never runs past the table: it is shifted back to end there): n_cand - 32; v3: INT64_MAX
};
__global__ __launch_bounds__(256) void prefilter_compact_kernel(const CompactParams C) {
    const int lane = threadIdx.x & 63;
    const uint32_t seg = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (seg >= C.n_segments) return;
    const uint32_t n = C.pair_count[seg];
    if (n == 0u || n > C.pair_cap) return;   // (nothing to re-score; an overflowing segment was recorded as empty)
    const int64_t blk = seg / (uint32_t)C.waves;
    const int wave = (int)(seg % (uint32_t)C.waves);
    const int64_t xcd = blk & 7, slot_id = blk >> 3, qb = slot_id % C.n_qb, cb = xcd + 8 * (slot_id / C.n_qb);   // (the prefilter's block -> tile map)
    const int64_t tile0 = cb * C.tiles_per_chunk;
    const int ntile = (int)(min(tile0 + (int64_t)C.tiles_per_chunk, C.n_tiles) - tile0);
    uint64_t* const base = C.pairs + (uint64_t)seg * C.pair_cap;
    const uint2* const bm = reinterpret_cast<const uint2*>(base);
    uint2 w[32];
#pragma unroll
    for (int t = 0; t < 32; ++t) w[t] = t < ntile ? bm[t * 64 + lane] : make_uint2(0u, 0u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every word has been read before the first pair overwrites one
    const uint64_t row_hi = (uint64_t)(uint32_t)(qb * 32 * C.waves + wave * 32 + (lane & 31)) << 32;
    const uint32_t lhi4 = 4u * (uint32_t)(lane >> 5);
    uint32_t done = 0u;   // pairs written so far (wave-uniform)
#pragma unroll
    for (int t = 0; t < 32; ++t) {
        const uint32_t c = (uint32_t)__builtin_popcount(w[t].x) + (uint32_t)__builtin_popcount(w[t].y);
        if (__builtin_amdgcn_ballot_w64(c != 0u) == 0ull) continue;   // wave-uniform
        uint32_t incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(incl, off, 64); if (lane >= off) incl += v; }
        uint32_t at = done + incl - c;
        const int64_t col = (tile0 + t) * V3_BN;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint32_t m = h ? w[t].y : w[t].x;
            while (m) {
                const int b = 31 - __builtin_clz(m);      // bit 31 - i holds value i = 16 (tb & 1) + r of blocks 2 h, 2 h + 1: ascending entities first
                m &= ~(1u << b);
                const int i = 31 - b, r = i & 15;
                const int64_t blk0 = min(col + 32 * (2 * h + (i >> 4)), C.last32);   // the block's first entity (v4: shifted back at the table's end)
                base[at++] = row_hi | (uint64_t)((uint32_t)(C.ent_offset + blk0) + lhi4 + (uint32_t)(8 * (r >> 2) + (r & 3)));
            }
        }
        done += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
}

// ---------------------------------------------------------------------------------------------------------
// v4 count kernel: 64 query rows per wave, ONE wave per SIMD, the entity stream entity-group-major.
//
// Measured on v3: every one of its eight waves reads the whole entity slice from LDS for its 32 query rows —
// 8 x 16 KB per 16 KB streamed, the LDS pipe as busy as the MFMA pipe — and the two waves of a SIMD reach the tile's compare
// epilogue together, behind the same barriers: MFMA busy 47 %.  Here a wave owns 64 query rows (two 32-row halves: 2 x NQ x 4
// registers of fragments in the AGPRs — a wave addresses 256 VGPRs + 256 AGPRs, an MFMA reads its A operand from either), so a
// B fragment read from LDS feeds TWO MFMAs and a workgroup of four waves covers the same 256 rows with half the LDS reads.
// With no second wave to hide behind, the epilogue is interleaved with the MFMAs by construction: the stream is reordered so
// that a STAGE is 32 entities x the whole contraction (a quarter tile, all k), a stage's 2 NQ MFMAs accumulate 64 x 32 scores
// in 32 VGPRs, and the compare-and-count of the PREVIOUS stage's 32 registers sits in the slots between this stage's MFMAs.
// Accumulators ping-pong between two VGPR sets (the stage loop is unrolled twice); the first MFMA of a stage takes srcC = 0.
// What the first version (compiler-scheduled builtins) measured, and what this one does about it:
//   * hipcc puts the MFMA results in AGPRs and copies all 32 to VGPRs at the stage's end (v_accvgpr_read x 32 behind the last
//     MFMA's latency), answers every use of a ds_read with s_waitcnt lgkmcnt(0) (the fragments read AHEAD drained with it) and
//     gives a counted value v_cmp -> s_and -> v_cndmask -> v_add: the 143 VALU instructions of a stage were NOT hidden (2 NQ MFMAs
//     = 1650 cycles, the stage took 3000).  Here the MFMAs, the LDS reads and the compare-and-count are inline asm: the
//     accumulators are VGPR operands of the MFMA itself, LDS reads are counted by hand (they return in order: "at most N younger
//     reads outstanding" is a fragment's arrival), a counted value is v_cmp + v_addc (the lane's carry-in IS the comparison).
//     The compiler knows none of the MFMA hazards inside asm; the schedule keeps every dependent pair apart by construction
//     (an accumulator is read by the VALU a whole MFMA after its last write at the earliest, B fragments arrive through
//     s_waitcnt, nothing else writes an MFMA operand).
//   * one barrier per stage with all four waves arriving together and the next stage's first fragments read behind it cost
//     ~550 cycles per stage: the barrier now stands PF k-steps before the stage's end and the next stage's first PF fragments
//     are read across it.
//   * the refill was one owner wave per stage behind a branch in every slot: every wave now issues its share of every stage's
//     LDS-DMA instructions (instruction j by wave j mod 4, the odd ones out twice), branch-free, a constant vmcnt.
// LDS: rows of 2 NQ + 1 sixteen-byte slots (an ODD pitch: conflict-free ds_read_b128 without a swizzle: 0 conflicts by PMC), a
// stage = NQ + 1 LDS-DMA instructions of 1 KB (the last one half padding), a ring of NS stages (6 at NQ = 25: 156 KB).
// The MFMA k-order per score is v3's (k-steps ascending into one accumulator): the same bits, the same counts — tests compare.
// The prefilter's pairs go to TWO segments per wave (one per 32-row half): the pair buffer keeps v3's geometry (eight
// segments of 32 query rows per 256-row workgroup), the re-scoring pass sees no difference.
template <class F, int... I> __device__ __forceinline__ void static_for(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int OFF> __device__ __forceinline__ void lds_read16(bf16x8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N> __device__ __forceinline__ void lds_wait(bf16x8& frag) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(frag) : "n"(N));
}
// (entity fragment as the A operand, query fragment — an AGPR — as B: the transposed tile of MODE 3, one query row per lane)
template <bool FIRST> __device__ __forceinline__ void mfma_asm_t(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %1, 0" : "=v"(acc) : "a"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %1, %0" : "+v"(acc) : "a"(a), "v"(b));
}
template <bool F16, bool FIRST> __device__ __forceinline__ void mfma_asm(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    if constexpr (FIRST) {
        if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(acc) : "a"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "a"(a), "v"(b));
    } else {
        if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
    }
}
constexpr int v4_ring(int nq) {   // B-fragment ring: the smallest divisor of NQ that is >= 4 (a stage then starts at ring index 0)
    for (int r = 4; r <= nq; ++r)
        if (nq % r == 0) return r;
    return nq;
}
template <int NQ> struct V4Geo {
    static constexpr int SLOTS = 2 * NQ + 1;            // 16-byte slots per LDS row
    static constexpr int PITCH = 16 * SLOTS;
    static constexpr int INSTR = NQ + 1;                // LDS-DMA instructions (1 KB each) per stage of 32 rows
    static constexpr int FI = (INSTR + 3) / 4;          // ... per wave
    static constexpr int STAGE = 1024 * INSTR;
    static constexpr int NS_FIT = (160 * 1024 - 2048) / STAGE;
    static constexpr int NS = NS_FIT > 8 ? 8 : NS_FIT;  // ring slots
    static constexpr int RB = v4_ring(NQ);              // B fragments held
    static constexpr int PF = RB - 1 > 6 ? 6 : RB - 1;  // ... read ahead (k-steps; <= 15 outstanding LDS reads encode)
    static_assert(NS >= 4 && FI * (NS - 2) <= 63 && PF >= 1 && PF < NQ, "v4: ring too short / vmcnt does not encode");
};
constexpr int vmcnt_imm(int n) { return 0x0F70 | (n & 15) | ((n >> 4) << 14); }   // s_waitcnt vmcnt(n), gfx9 encoding (6 bits, split)

#ifndef V4_ABLATE
#define V4_ABLATE 0  // timing experiments only (wrong results): 1 no refills, 2 no compare epilogue, 4 no MFMAs, 8 no LDS reads, 16 no wait / barrier
#endif
// MODE 3 (round 6): the prefilter as a BITMAP, as v3's MODE 3 — the products transposed (the query fragment is the MFMA's B
// operand: lane = query row, the stage's 32 entities in the accumulator registers), per value two compares shifted into two words
// of the lane by their own carries (4 VALU instructions, no scalar work, no branch: they sit in the slots between the MFMAs like
// the one-counter form's two), and after every second stage of a tile — 32 values per word — count += popcount, undecided =
// XOR, one 4-byte store per lane and half into the half's segment of the pair buffer ([tile][lane][word], v3's layout:
// prefilter_compact_kernel reads both).  The thresholds are two registers per half (64 in the other modes).
template <int NQ, int MODE>   // MODE 0 both counters | 1 one | 2 prefilter | 3 prefilter, bitmap (see v3)
__global__ __launch_bounds__(256, 1) void count_mfma_bf16_v4_kernel(const CountBf16Params P) {
    using G = V4Geo<NQ>;
    constexpr bool ONE = MODE == 1, PRE = MODE >= 2, BMP = MODE == 3;
    constexpr int NS = G::NS, STAGE = G::STAGE, INSTR = G::INSTR, FI = G::FI, SLOTS = G::SLOTS, PF = G::PF, RB = G::RB;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];  // ring | thresholds
    float* thr_s = reinterpret_cast<float*>(smem + NS * STAGE);             // [0,256): gt, [256,512): ge

    const int64_t id = blockIdx.x;
    const int64_t xcd = id & 7, slot_id = id >> 3;
    const int64_t qb = slot_id % P.n_qb;
    const int64_t cb = xcd + 8 * (slot_id / P.n_qb);
    if (cb >= P.n_cb) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int64_t n_cand = P.n_cand, ld_ent = P.ld_ent;

    // ---- one-time: thresholds -> LDS -> registers, this wave's query fragments -> AGPRs --------------------
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const int64_t qr = qb * 256 + tid;
        float t = INFINITY;  // rows past the end count nothing
        if (PRE && P.thr_direct != nullptr) {
            if (qr < P.n_rows) t = P.thr_direct[(half ? P.n_rows : 0) + qr];
        } else if (qr < P.n_rows) {
            const int p = P.pos_int[qr];
            const bool want_gt = ONE ? P.need == 2 : half == 0;
            t = acc_threshold(want_gt ? gt_threshold(p) : ge_threshold(p), P.cmul);
            if constexpr (PRE) {   // widened by the band, outwards (v3)
                const float b = P.band[qr] + 1e-6f * fabsf(t) + 1e-30f;
                t = want_gt ? nextafterf(t + b, INFINITY) : nextafterf(t - b, -INFINITY);
            }
        }
        thr_s[half * 256 + tid] = t;
    }
    bf16x8 A[2][NQ];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int64_t qr = min(qb * 256 + wave * 64 + 32 * h + l31, P.n_rows - 1);
        const uint16_t* qp = P.Q + qr * P.ldq + 8 * lhi;
#pragma unroll
        for (int q = 0; q < NQ; ++q) A[h][q] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(qp + q * 16));
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // compiler-visible vmcnt(0): from here on only LDS-DMA (and pair stores) are in the VMEM queue
    __syncthreads();
    f32x4 gth[2][4], eth[2][4];   // this lane's rows: 64 w + 32 h + 8 j + i + 4 lhi
    float g_t[2] = {0.f, 0.f}, e_t[2] = {0.f, 0.f};   // MODE 3: this lane's ROW 64 w + 32 h + l31
    if constexpr (BMP) {
#pragma unroll
        for (int h = 0; h < 2; ++h) { g_t[h] = thr_s[wave * 64 + 32 * h + l31]; e_t[h] = thr_s[256 + wave * 64 + 32 * h + l31]; }
    } else {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r0 = wave * 64 + 32 * h + 8 * j + 4 * lhi;
            gth[h][j] = *reinterpret_cast<const f32x4*>(thr_s + r0);
            if constexpr (!ONE) eth[h][j] = *reinterpret_cast<const f32x4*>(thr_s + 256 + r0);
        }
    }

    __builtin_amdgcn_s_waitcnt(0xC07F);   // compiler-visible lgkmcnt(0): no LDS read of the compiler's is pending inside the stage loop
                                          // (it would answer each use there with an s_waitcnt lgkmcnt(0) that drains the hand-counted reads)
    unsigned cgt[2][16];   // per accumulator register: how often its score reached the first threshold (<= 4 x tiles_per_chunk)
    unsigned ceq[2][4];    // MODE 0: ties, four packed 8-bit counters per register quadruple (rare)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int r = 0; r < 16; ++r) cgt[h][r] = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) ceq[h][j] = 0u;
    }
    // MODE 3: the words the compares are shifted into (32 values = two stages fill one), the lane's counts
    unsigned bw_g[2] = {0u, 0u}, bw_e[2] = {0u, 0u}, bc_gt[2] = {0u, 0u}, bc_un[2] = {0u, 0u};

    // ---- LDS-DMA producer: every wave issues its share (instruction j by wave j mod 4) of every stage ---------
    const int64_t tile0 = cb * P.tiles_per_chunk;
    const int ntile = (int)(min(tile0 + (int64_t)P.tiles_per_chunk, P.n_tiles) - tile0);
    const int rot = (int)(qb % ntile);   // tiles are visited in a per-block rotated order (v2)
    unsigned off[FI];                    // byte offset of this lane's piece of my fill instruction i from the stage's first row
    unsigned ldo[FI];                    // ... and its 1 KB place in the stage (scalar)
#pragma unroll
    for (int i = 0; i < FI; ++i) {
        int j = wave + 4 * i;
        if (j >= INSTR) j -= 4;          // (the odd ones out: an instruction of mine again — the same bytes to the same place)
        const int p = 64 * j + lane;
        off[i] = (unsigned)min(p / SLOTS, 31) * (unsigned)(ld_ent * 2) + (unsigned)min(p % SLOTS, 2 * NQ - 1) * 16u;
        ldo[i] = (unsigned)__builtin_amdgcn_readfirstlane(1024 * j);
    }
    // (the table's last rows: a stage that would run past them is SHIFTED back to end at the last row — every fetched row exists,
    //  nothing is clamped per lane; the epilogue masks the rows it has seen in the stage before and names entities from the shifted base)
    int ftile = rot, ftb = 0, fslot = 0;   // chunk-local tile / quarter / ring slot of the next stage to fill
    const unsigned char* fsrc = nullptr;
    unsigned fbase = 0u;
    const unsigned ring0 = (unsigned)(uintptr_t)smem;
    auto fill_point = [&]() __attribute__((always_inline)) {
        const int64_t row0 = min((tile0 + ftile) * V3_BN + 32 * ftb, n_cand - 32);
        fsrc = reinterpret_cast<const unsigned char*>(P.ent) + row0 * ld_ent * 2;
        fbase = __builtin_amdgcn_readfirstlane(ring0 + (unsigned)fslot * STAGE);
    };
    auto fill_one = [&](int i) __attribute__((always_inline)) {
        if (!(V4_ABLATE & 1)) glds16(reinterpret_cast<const uint16_t*>(fsrc + off[i]), fbase + ldo[i]);
    };
    auto fill_next = [&]() __attribute__((always_inline)) {
        ftb = (ftb + 1) & 3;
        if (ftb == 0) ftile = ftile + 1 == ntile ? 0 : ftile + 1;
        fslot = fslot + 1 == NS ? 0 : fslot + 1;
    };
#pragma unroll 1
    for (int t = 0; t < NS - 1; ++t) {   // stages 0 .. NS-2 (the stream wraps around the chunk; fills past its end are never read)
        fill_point();
#pragma unroll
        for (int i = 0; i < FI; ++i) fill_one(i);
        fill_next();
    }

    // ---- consumer ---------------------------------------------------------------------------------------
    const unsigned rd_base = ring0 + (unsigned)(l31 * G::PITCH + 16 * lhi);
    f32x16 accA[2], accB[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accA[h][r] = 0.f; accB[h][r] = 0.f; }
    int cslot = 0;
    uint64_t cokP = 0ull;   // lanes whose entity of the PREVIOUS stage is new (none before the first stage)
    bool fullP = false;     // ... all of them (else: patch)
    uint32_t colP = 0u;     // global id of that stage's first entity
    int shiftP = 64;        // MODE 3: entities of the previous stage the stage before has seen (its rows were shifted back; 64: all — no stage yet)
    int slotP = -1;         // MODE 3: 2 x (the previous stage's tile in the chunk) + its word (tb >> 1); -1: none
    unsigned pair_n[2] = {0u, 0u}, pair_over = 0u;
    uint64_t* pair_base[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
        pair_base[h] = PRE ? P.pairs + ((uint64_t)blockIdx.x * 8u + (unsigned)(2 * wave + h)) * P.pair_cap : nullptr;
    const uint32_t row_s = (uint32_t)(qb * 256 + wave * 64);
    const unsigned lhi4 = 4u * (unsigned)lhi;

    auto emit_mask = [&](uint64_t m, int h, int r) __attribute__((always_inline)) {   // (v3's emission; r = 4 j + i: row = row_s + 32 h + 8 j + i + 4 lhi)
        const unsigned n = (unsigned)__builtin_popcountll(m);
        if (pair_n[h] + n > P.pair_cap) { pair_over = 1u; return; }   // the caller redoes this query tile with the exact kernel
        const unsigned before = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        // (the lane terms are made opaque: otherwise the 32 lane-dependent row sums are hoisted out of the stage loop and held in
        // vector registers for an emission that may never come)
        unsigned lv = (unsigned)l31, hv = lhi4;
        asm volatile("" : "+v"(lv), "+v"(hv));
        if (__builtin_amdgcn_inverse_ballot_w64(m))   // low word: entity, high word: query row
            *reinterpret_cast<uint2*>(pair_base[h] + pair_n[h] + before) =
                make_uint2(colP + lv, row_s + (unsigned)(32 * h + 8 * (r >> 2) + (r & 3)) + hv);
        pair_n[h] = (unsigned)__builtin_amdgcn_readfirstlane((int)(pair_n[h] + n));
    };
    // one accumulator register of the previous stage: compare and count; v = 16 h + 4 j + i.  Two counters / the prefilter: what
    // the two lane masks say — a tie, an undecided candidate — is looked at ONE VALUE LATER (resolve): the scalar test of masks
    // the VALU has just produced would stall the wave, and the MFMA behind it, for the compare's latency.
    uint64_t smg = 0ull, sme = 0ull;
    auto resolve = [&](int v) __attribute__((always_inline)) {
        const int h = v >> 4, r = v & 15, j = r >> 2, i = r & 3;
        const uint64_t mu = sme & ~smg;
        if (mu) {   // wave-uniform; ties are rare, the prefilter's undecided are a few per stage
            if constexpr (PRE) emit_mask(mu, h, r);
            else ceq[h][j] += __builtin_amdgcn_inverse_ballot_w64(mu) ? (1u << (8 * i)) : 0u;
        }
    };
    auto judge = [&](f32x16 (&aP)[2], int v) __attribute__((always_inline)) {
        const int h = v >> 4, r = v & 15, j = r >> 2, i = r & 3;
        const float x = aP[h][r];
        if constexpr (BMP) {
            // (both compares first, into scalar pairs: a carry is not read by the instruction right behind the compare that made it)
            uint64_t c0_, c1_;
            asm volatile("v_cmp_ge_f32_e64 %2, %4, %5\n\tv_cmp_ge_f32_e64 %3, %4, %6\n\t"
                         "v_addc_co_u32_e64 %0, vcc, %0, %0, %2\n\tv_addc_co_u32_e64 %1, vcc, %1, %1, %3"
                         : "+v"(bw_g[h]), "+v"(bw_e[h]), "=&s"(c0_), "=&s"(c1_) : "v"(x), "v"(g_t[h]), "v"(e_t[h]) : "vcc");
        } else if constexpr (ONE) {
            asm volatile("v_cmp_ge_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(cgt[h][r]) : "v"(x), "v"(gth[h][j][i]) : "vcc");
        } else {
            if (v > 0) resolve(v - 1);
            asm volatile("v_cmp_ge_f32_e64 %1, %3, %4\n\tv_cmp_ge_f32_e64 %2, %3, %5\n\tv_addc_co_u32_e64 %0, vcc, 0, %0, %1"
                         : "+v"(cgt[h][r]), "=&s"(smg), "=&s"(sme) : "v"(x), "v"(gth[h][j][i]), "v"(eth[h][j][i]) : "vcc");
        }
    };
    uint32_t* const bm32[2] = {reinterpret_cast<uint32_t*>(pair_base[0]), reinterpret_cast<uint32_t*>(pair_base[1])};
    auto judge_flush = [&]() __attribute__((always_inline)) {   // the stage's last value (before colP moves on)
        if constexpr (BMP) {
            // a word is full after the tile's second and fourth stage: count, undecided, store ([tile][lane][word] of the half's segment)
            if (slotP >= 0 && (slotP & 0x10000)) {   // (wave-uniform; bit 16: the previous stage was an odd one)
                const int sl_ = slotP & 0xffff;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const unsigned u = bw_e[h] ^ bw_g[h];   // (the first threshold is the higher one)
                    bc_gt[h] += (unsigned)__builtin_popcount(bw_g[h]);
                    bc_un[h] += (unsigned)__builtin_popcount(u);
                    bm32[h][((sl_ >> 1) * 64 + lane) * 2 + (sl_ & 1)] = u;
                }
            }
        } else if constexpr (!ONE) resolve(31);
    };

    // The table's last stage was shifted back over rows the stage before has counted: their lanes get a NaN score — it reaches no
    // threshold — and the compare-and-count needs no mask.  (An accumulator the VALU writes behind an MFMA's back: the nops let
    // the stage's last MFMA retire first; once per table.)
    auto patch = [&](f32x16 (&aP)[2]) __attribute__((always_inline)) {
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(aP[0]), "+v"(aP[1]));
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if constexpr (BMP) aP[h][r] = 8 * (r >> 2) + (r & 3) + 4 * lhi >= shiftP ? aP[h][r] : __builtin_nanf("");   // (transposed: the entities are registers)
                else aP[h][r] = __builtin_amdgcn_inverse_ballot_w64(cokP) ? aP[h][r] : __builtin_nanf("");
            }
        asm volatile("" : "+v"(aP[0]), "+v"(aP[1]));
    };
    // the compare-and-count of the previous stage goes into the slots 1 .. JS - 1, the refill's instructions behind the barrier
    constexpr int JS = 2 * (NQ - PF) > 2 ? 2 * (NQ - PF) : 2;
    auto run_stage = [&](f32x16 (&aC)[2], f32x16 (&aP)[2], bf16x8 (&B)[RB], int s) __attribute__((always_inline)) {
        if (!fullP) patch(aP);   // (block-uniform, the table's last stage only)
        const unsigned rd = rd_base + (unsigned)cslot * STAGE;
        cslot = cslot + 1 == NS ? 0 : cslot + 1;
        const unsigned rdn = rd_base + (unsigned)cslot * STAGE;
        static_for([&](auto Qc) __attribute__((always_inline)) {
            constexpr int q = decltype(Qc)::value;
            if constexpr (q == NQ - PF) {
                // the next stage has landed (mine: at most the NS - 3 younger batches still fly; the barrier: everyone's), and
                // every wave is done with the stage BEFORE this one: its slot is refilled behind the barrier
                if constexpr (!(V4_ABLATE & 16)) {
                    __builtin_amdgcn_s_waitcnt(vmcnt_imm(FI * (NS - 3)));
                    __builtin_amdgcn_s_barrier();
                }
                fill_point();
            }
            if constexpr (!(V4_ABLATE & 8)) {
                if constexpr (q + PF < NQ) lds_read16<32 * (q + PF)>(B[(q + PF) % RB], rd);
                else lds_read16<32 * (q + PF - NQ)>(B[(q + PF) % RB], rdn);   // the next stage's first fragments, across the barrier
            }
            lds_wait<PF>(B[q % RB]);   // this k-step's fragment has arrived; the PF younger reads fly on
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                if constexpr (!(V4_ABLATE & 4)) {
                    if constexpr (BMP) mfma_asm_t<q == 0>(aC[hh], A[hh][q], B[q % RB]);
                    else mfma_asm<PRE, q == 0>(aC[hh], A[hh][q], B[q % RB]);
                }
                // the slot behind this MFMA
                const int sl = 2 * q + hh;
                if (sl >= 1 && sl < JS && !(V4_ABLATE & 2)) {
#pragma unroll
                    for (int v = (sl - 1) * 32 / (JS - 1); v < sl * 32 / (JS - 1); ++v) judge(aP, v);
                }
                if (sl == JS && !(V4_ABLATE & 2)) judge_flush();
                if (sl >= JS || JS == 2 * NQ) {
                    const int u = JS == 2 * NQ ? sl : sl - JS, nu = 2 * NQ - (JS == 2 * NQ ? 0 : JS);
#pragma unroll
                    for (int i = u * FI / nu; i < (u + 1) * FI / nu; ++i) fill_one(i);
                }
            }
        }, std::make_integer_sequence<int, NQ>{});
        fill_next();
        const int tb = s & 3, ti = s >> 2;
        const int ctile = rot + ti >= ntile ? rot + ti - ntile : rot + ti;
        const int64_t col0 = (tile0 + ctile) * V3_BN + 32 * tb, col0s = min(col0, n_cand - 32);   // (shifted like the fill)
        fullP = col0s == col0;
        cokP = __builtin_amdgcn_ballot_w64(col0s + l31 >= col0);
        colP = (uint32_t)(P.ent_offset + col0s);
        shiftP = (int)(col0 - col0s);
        slotP = 2 * ctile + (tb >> 1) + ((tb & 1) << 16);
    };
    // stage 0 has landed; its first PF fragments
    bf16x8 B[RB];
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(FI * (NS - 2)));
    __builtin_amdgcn_s_barrier();
    static_for([&](auto Qc) __attribute__((always_inline)) { constexpr int q = decltype(Qc)::value; lds_read16<32 * q>(B[q], rd_base); }, std::make_integer_sequence<int, PF>{});
#pragma unroll 1
    for (int s = 0; s < 4 * ntile; s += 2) {
        run_stage(accA, accB, B, s);
        run_stage(accB, accA, B, s + 1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the fragments read past the last stage)
    if (!fullP) patch(accB);
#pragma unroll
    for (int v = 0; v < 32; ++v) judge(accB, v);   // the last stage's
    judge_flush();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may land after this workgroup has left
    if constexpr (BMP) {   // a half's undecided candidates: the wave's sum; more than the segment's entries: the segment counts as EMPTY (v3)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            unsigned u = bc_un[h];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) u += __shfl_xor(u, o, 64);
            if (u > P.pair_cap) { pair_over = 1u; u = 0u; }
            pair_n[h] = u;
        }
    }
    if constexpr (PRE) {
        if (lane == 0) {
#pragma unroll
            for (int h = 0; h < 2; ++h) P.pair_count[blockIdx.x * 8u + (unsigned)(2 * wave + h)] = pair_n[h];
            if (pair_over) atomicOr(P.pair_count + P.n_segments, 1u);
        }
    }
    if constexpr (BMP) {   // a row is the lanes l and l + 32 of its half
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            unsigned c = bc_gt[h];
            c += __shfl_xor(c, 32, 64);
            const int64_t qr = qb * 256 + wave * 64 + 32 * h + l31;
            if (lhi == 0 && qr < P.n_rows && c) atomicAdd(&P.cnt_gt[qr], (int)c);
        }
        return;
    }
    // ---- rows are private to the wave: lane shuffle, one global atomic per row and counter ----------------
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            unsigned c = cgt[h][r] | (((ceq[h][r >> 2] >> (8 * (r & 3))) & 0xffu) << 16);  // gt | eq << 16: room for the 32-lane sum
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) c += __shfl_xor(c, o, 64);
            if (l31 == 0) {
                const int64_t qr = qb * 256 + wave * 64 + 32 * h + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (qr < P.n_rows) {
                    if (c & 0xffffu) atomicAdd(&P.cnt_gt[qr], (int)(c & 0xffffu));
                    if (c >> 16) atomicAdd(&P.cnt_eq[qr], (int)(c >> 16));
                }
            }
        }
}

template <int NQ, int MODE>
static int launch_v4(const CountBf16Params& P, int64_t nblk, hipStream_t st) {
    const size_t lds_bytes = (size_t)V4Geo<NQ>::NS * V4Geo<NQ>::STAGE + 2 * 256 * sizeof(float);
    static std::atomic<uint64_t> devices_done{0};  // one flag per template instance and device
    int rc = allow_full_lds((const void*)count_mfma_bf16_v4_kernel<NQ, MODE>, devices_done);
    if (rc != EMG_OK) return rc;
    hipLaunchKernelGGL((count_mfma_bf16_v4_kernel<NQ, MODE>), dim3((unsigned)nblk), dim3(256), lds_bytes, st, P);
    return EMG_OK;
}
static int v4_mode() {   // EMG_BF16_V4: 0 the v3 kernel everywhere, 1 (default) v4 where it wins (one counter), 2 v4 in every mode at 400 columns (A/B, tests)
    const char* e = getenv("EMG_BF16_V4");
    return e ? atoi(e) : 1;
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

// Filter counts of the bf16 mode, scored through the SAME MFMA arithmetic as the count kernels: one wave takes 32
// (query row, filter entity) pairs of the CSR, puts pair i's query row in row i of the A fragments and its entity row in
// column i of the B fragments and multiplies k-step by k-step in the count kernels' order — element (i, i) of the
// 32 x 32 product is pair i's score with exactly the bits the count kernel produced for that (row, entity), so what the
// filter subtracts is what was counted (a sequential fmaf chain differs from the MFMA's internal order in the last
// bit, which flips (int)(score * 1e5) for a few candidates per million).  31/32 of the product is discarded; filter
// sets are a few entities per row.  The row's own entity is one tie by construction (see header).
__global__ __launch_bounds__(256) void filter_count_bf16_kernel(int model, const uint16_t* __restrict__ Q, int64_t ldq,
                                                                const int32_t* __restrict__ pos_int,
                                                                const int32_t* __restrict__ self_ent, int64_t n_rows,
                                                                const uint16_t* __restrict__ ent, int64_t n_local,
                                                                int64_t ld_ent, int64_t ent_offset, int k16,
                                                                float scale, const int64_t* __restrict__ fptr,
                                                                const int32_t* __restrict__ fidx,
                                                                int32_t* __restrict__ fgt, int32_t* __restrict__ feq) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, lhi = lane >> 5;
    const int64_t total = fptr[n_rows];  // (the host does not know it: the grid is fixed and strides over the pairs)
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; wave * 32 < total; wave += n_waves) {
    const int64_t u = wave * 32 + l31;   // this lane's pair (both half-waves hold the same 32 pairs)
    // the pair's row: the last r with fptr[r] <= u
    int64_t lo = 0, hi = n_rows;         // fptr[lo] <= u < fptr[hi]
    const int64_t uu = min(u, total - 1);
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (fptr[mid] <= uu) lo = mid; else hi = mid;
    }
    const int64_t r = lo;
    const int gidx = fidx[uu];
    const int64_t e = (int64_t)gidx - ent_offset;
    const bool in_range = u < total && e >= 0 && e < n_local;
    const bool is_self = in_range && gidx == self_ent[r];
    const uint16_t* qp = Q + r * ldq + 8 * lhi;
    const uint16_t* ep = ent + (in_range ? e : 0) * ld_ent + 8 * lhi;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int q = 0; q < k16; ++q) {
        const bf16x8 a = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(qp + q * 16));
        const bf16x8 b = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(ep + q * 16));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    // element (row i, column i): column = l31 = i; row i sits in register (i & 3) + 4 (i >> 3) of the half-wave (i >> 2) & 1
    const int want = (l31 & 3) + 4 * (l31 >> 3);
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) v = want == i ? acc[i] : v;
    if (lhi != ((l31 >> 2) & 1) || !in_range) continue;
    if (is_self) { atomicAdd(&feq[r], 1); continue; }
    const float cmul = (model == EMG_HOLE ? scale : 1.0f) * 100000.0f;
    const int ci = (int)(v * cmul), p = pos_int[r];
    if (ci > p) atomicAdd(&fgt[r], 1);
    else if (ci == p) atomicAdd(&feq[r], 1);
    }
}

// k-steps of 16 the prefilter kernel is instantiated for, and what they ask of the operand rows
// Entity tiles per chunk of the PREFILTER: the chunk is what the re-scoring pass sweeps with the query rows of one segment
// in LDS, and its f32 rows (tiles x 128 x 4 k_int bytes) should sit in one XCD's 4 MB L2 next to the other segments' sweeps.
static int v3_prefilter_tiles() {
    static const int t = [] { const char* e = getenv("EMG_PRE_TILES"); const int v = e ? atoi(e) : 32; return v >= 1 && v <= 32 ? v : 32; }();
    return t;
}

constexpr int V3_WIDE_FROM = 26;   // k-steps from which the prefilter runs as 4 waves x 128 query rows (query fragments > 100 registers)
static int v3_prefilter_steps(int k16) {
    static const int have[] = {4, 7, 8, 10, 13, 16, 19, 22, 25, 32, 38, 44, 50};
    for (int nq : have)
        if (k16 <= nq) return nq;
    return 0;
}
static int64_t v3_prefilter_ld(int k16) {   // the entity slices are fetched 64 columns at a time
    const int nq = v3_prefilter_steps(k16);
    return nq ? 64 * (int64_t)((nq + 3) / 4) : 0;
}

static int launch_bf16(int mode, CountBf16Params& P, hipStream_t st) {
    EMG_REQUIRE(P.model >= EMG_DISTMULT && P.model <= EMG_HOLE, "bf16 eval: model %d is not a contraction (TransE stays f32 VALU)", P.model);
    const int64_t ld_min = (P.k_pad + HBK_ - 1) / HBK_ * HBK_;
    EMG_REQUIRE(P.k_pad > 0 && P.k_pad % 16 == 0 && P.ldq >= ld_min && P.ld_ent >= ld_min,
                "bf16 eval: k_pad must be a multiple of 16 and rows stored zero-padded with ld >= round_up(k_pad, %d) (k_pad=%d ldq=%lld ld=%lld)",
                HBK_, P.k_pad, (long long)P.ldq, (long long)P.ld_ent);
    P.k16 = P.k_pad / 16;               // MFMA k-steps holding real data
    P.k_pad = (P.k_pad + 31) / 32 * 32;  // the v1/v2 kernels multiply whole 32-wide steps (the tail is zero padding)
    EMG_REQUIRE(P.ldq % 8 == 0 && P.ld_ent % 8 == 0 && aligned16(P.Q) && aligned16(P.ent), "bf16 eval: rows must be 16-byte aligned");
    if (P.n_rows == 0 || P.n_cand == 0) return EMG_OK;
    P.cmul = (P.model == EMG_HOLE ? P.scale : 1.0f) * 100000.0f;
    P.n_qb = cdiv(P.n_rows, HBM_);
    P.n_tiles = cdiv(P.n_cand, HBN_);
    P.tiles_per_chunk = mode == BF_DENSE ? 4 : 32;
    P.n_cb = cdiv(P.n_tiles, P.tiles_per_chunk);
    const int64_t blocks = mode == BF_DIAG ? P.n_qb : 8 * P.n_qb * cdiv(P.n_cb, 8);
    EMG_REQUIRE(blocks < ((int64_t)1 << 31), "bf16 eval: grid too large");
    if (mode == BF_DENSE) hipLaunchKernelGGL(count_mfma_bf16_kernel<BF_DENSE>, dim3((unsigned)blocks), dim3(256), 0, st, P);
    else if (mode == BF_DIAG) hipLaunchKernelGGL(count_mfma_bf16_kernel<BF_DIAG>, dim3((unsigned)blocks), dim3(256), 0, st, P);
    else if (P.cand == nullptr && P.cmul > 0.f && P.cmul < INFINITY && P.n_rows > V2_BM &&
             (P.k16 == 25 || P.k16 == 13 || P.k16 == 8 ||
              (P.pairs != nullptr && v3_prefilter_steps(P.k16) != 0 && P.ldq >= v3_prefilter_ld(P.k16) && P.ld_ent >= v3_prefilter_ld(P.k16)))) {
        // register-stationary query fragments + deep LDS-DMA ring (see its header); common k only
        P.n_qb = cdiv(P.n_rows, V3_BM);
        P.n_tiles = cdiv(P.n_cand, V3_BN);
        P.tiles_per_chunk = P.pairs ? v3_prefilter_tiles() : 32;  // <= 32: the epilogue's packed counters are 8 bits wide
        P.n_cb = cdiv(P.n_tiles, P.tiles_per_chunk);
        const int64_t nblk = 8 * P.n_qb * cdiv(P.n_cb, 8);
        EMG_REQUIRE(nblk < ((int64_t)1 << 31), "bf16 eval: grid too large");
        int rc;  // 64-wide slices (SQ = 4): one barrier per 16 MFMAs measured 1.5-3.5 % faster than 32-wide
        // the prefilter as a bitmap (MODE 3) wherever a wave's segment of the pair buffer holds its tiles' words (64 entries per tile:
        // 2048 entries at 32 tiles — what ranking.py allocates up to 65536 segments); EMG_PRE_BITMAP=0: the emitting form (A/B)
        const char* bm_env = getenv("EMG_PRE_BITMAP");   // (read per call: tests compare the two forms inside one process)
        const bool bitmap_ok = !(bm_env && bm_env[0] && atoi(bm_env) == 0);
        const bool bmp = P.pairs && bitmap_ok && (int64_t)P.pair_cap >= 64 * (int64_t)P.tiles_per_chunk &&
                         !(v4_mode() == 2 && v3_prefilter_steps(P.k16) == 25);   // (EMG_BF16_V4=2: the v4 kernel's emitting prefilter, an A/B form)
        EMG_REQUIRE(!P.ties || bmp, "bf16 prefilter (ties form): a wave's segment must hold its bitmap (64 entries per entity tile)");
        const int md = P.pairs ? (P.ties ? 4 : (bmp ? 3 : 2)) : (P.need != 0 ? 1 : 0);
        const char* pv4 = getenv("EMG_PRE_V4");   // 0: the bitmap prefilter through v3 at every width (A/B; read per call)
        const bool pre_v4 = !(pv4 && pv4[0] && atoi(pv4) == 0);
        bool bmp_v4 = false;
#define EMG_V3P(NQ_) (md == 4 ? launch_v3<NQ_, 4, 4>(P, nblk, st) : md == 3 ? launch_v3<NQ_, 4, 3>(P, nblk, st) : launch_v3<NQ_, 4, 2>(P, nblk, st))
#define EMG_V3(NQ_) (md >= 2 ? EMG_V3P(NQ_) : md == 1 ? launch_v3<NQ_, 4, 1>(P, nblk, st) : launch_v3<NQ_, 4, 0>(P, nblk, st))
        // the prefilter (exact-fast mode, what evaluate_performance uses by default) at EVERY width up to 400: the next
        // instantiated step count, the extra k-steps multiply the rows' zero padding (exact zeros: nothing changes)
        const int nq = P.pairs != nullptr ? v3_prefilter_steps(P.k16) : P.k16;
        if (nq >= V3_WIDE_FROM) {   // 400 < k_int <= 800: one wave per SIMD, 128 query rows per workgroup
            P.n_qb = cdiv(P.n_rows, 128);
            const int64_t wblk = 8 * P.n_qb * cdiv(P.n_cb, 8);
            EMG_REQUIRE(wblk < ((int64_t)1 << 31), "bf16 eval: grid too large");
#define EMG_V3W(NQ_) (md == 4 ? launch_v3<NQ_, 4, 4, 4>(P, wblk, st) : md == 3 ? launch_v3<NQ_, 4, 3, 4>(P, wblk, st) : launch_v3<NQ_, 4, 2, 4>(P, wblk, st))
            if (nq == 32) rc = EMG_V3W(32);
            else if (nq == 38) rc = EMG_V3W(38);
            else if (nq == 44) rc = EMG_V3W(44);
            else rc = EMG_V3W(50);
#undef EMG_V3W
        }
        else if (md == 1 && v4_mode() >= 1 && P.n_cand >= V3_BN) {   // one counter: 64 query rows per wave (v4: -9 % at 400 columns)
            if (nq == 25) rc = launch_v4<25, 1>(P, nblk, st);
            else if (nq == 13) rc = launch_v4<13, 1>(P, nblk, st);
            else rc = launch_v4<8, 1>(P, nblk, st);
        }
        else if (md == 3 && v4_mode() >= 1 && pre_v4 && P.n_cand >= V3_BN && (nq == 25 || nq == 13)) {   // the bitmap prefilter, 64 query rows per wave
            rc = nq == 25 ? launch_v4<25, 3>(P, nblk, st) : launch_v4<13, 3>(P, nblk, st);
            bmp_v4 = true;
        }
        else if (nq == 25 && v4_mode() == 2 && P.n_cand >= V3_BN) {   // A/B only: v4 loses to v3 with two counters / as the prefilter
            rc = md == 2 ? launch_v4<25, 2>(P, nblk, st) : launch_v4<25, 0>(P, nblk, st);
        }
        else if (nq == 25) rc = EMG_V3(25);
        else if (nq == 13) rc = EMG_V3(13);
        else if (nq == 8) rc = EMG_V3(8);
        else if (nq == 4) rc = EMG_V3P(4);
        else if (nq == 7) rc = EMG_V3P(7);
        else if (nq == 10) rc = EMG_V3P(10);
        else if (nq == 16) rc = EMG_V3P(16);
        else if (nq == 19) rc = EMG_V3P(19);
        else rc = EMG_V3P(22);
#undef EMG_V3
#undef EMG_V3P
        if (rc != EMG_OK) return rc;
        if (md >= 3) {   // the segments' bitmaps -> pairs, in place
            EMG_LAUNCH_CHECK();
            CompactParams C{};
            C.pairs = P.pairs; C.pair_count = P.pair_count; C.pair_cap = P.pair_cap; C.n_segments = P.n_segments;
            C.n_qb = P.n_qb; C.n_cb = P.n_cb; C.n_tiles = P.n_tiles; C.ent_offset = P.ent_offset; C.tiles_per_chunk = P.tiles_per_chunk;
            C.waves = nq >= V3_WIDE_FROM ? 4 : 8;
            C.last32 = bmp_v4 ? P.n_cand - 32 : INT64_MAX;
            hipLaunchKernelGGL(prefilter_compact_kernel, dim3((unsigned)cdiv((int64_t)P.n_segments, 4)), dim3(256), 0, st, C);
        }
    } else if (P.pairs) {
        return fail(EMG_ENOSUP, "bf16 prefilter: contraction widths up to 400 on rows of at least emg_eval_prefilter_ld columns, more than 128 query rows, no candidate list");
    } else if (P.cand == nullptr && P.k_pad <= V2_KPAD_MAX && P.cmul > 0.f && P.cmul < INFINITY) {
        // query-stationary LDS-DMA kernel (see its header); anything else takes the v1 tile kernel above
        const int m = P.k_pad / 32;
        P.qs = (m & 1) ? 64 * m : 64 * (m + 1);
        P.n_tiles = cdiv(P.n_cand, V2_BN);
        P.tiles_per_chunk = 16;
        P.n_cb = cdiv(P.n_tiles, P.tiles_per_chunk);
        const int64_t nblk = 8 * P.n_qb * cdiv(P.n_cb, 8);
        EMG_REQUIRE(nblk < ((int64_t)1 << 31), "bf16 eval: grid too large");
        const size_t lds_bytes = (size_t)V2_NS * V2_STAGE + (size_t)V2_BM * P.qs + 3 * V2_BM * sizeof(float);
        static std::atomic<uint64_t> devices_done{0};
        int rc2 = allow_full_lds((const void*)count_mfma_bf16_v2_kernel, devices_done);
        if (rc2 != EMG_OK) return rc2;
        hipLaunchKernelGGL(count_mfma_bf16_v2_kernel, dim3((unsigned)nblk), dim3(512), lds_bytes, st, P);
    } else hipLaunchKernelGGL(count_mfma_bf16_kernel<BF_COUNT>, dim3((unsigned)blocks), dim3(256), 0, st, P);
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
                                   int32_t* cnt_gt, int32_t* cnt_eq, int32_t need, void* stream) {
    EMG_REQUIRE((n_rows == 0 || n_cand == 0) || (q_bf16 && pos_int && ent_bf16 && cnt_gt && cnt_eq),
                "emg_eval_count_bf16: null pointer");
    EMG_REQUIRE(need >= 0 && need <= 2, "emg_eval_count_bf16: need must be 0 (both), 1 (>=) or 2 (>)");
    CountBf16Params P{};
    P.Q = (const uint16_t*)q_bf16; P.ldq = ldq; P.pos_int = pos_int; P.self_ent = self_ent; P.n_rows = n_rows;
    P.ent = (const uint16_t*)ent_bf16; P.n_cand = n_cand; P.ld_ent = ld_ent; P.cand = cand; P.ent_offset = ent_offset;
    P.k_pad = k_pad; P.scale = scale; P.model = model; P.cnt_gt = cnt_gt; P.cnt_eq = cnt_eq; P.need = need;
    // the tile kernels always produce both counters: for need = 1 their ties are added into cnt_gt as well
    if (need == 1) P.cnt_eq = cnt_gt;
    return launch_bf16(BF_COUNT, P, (hipStream_t)stream);
}

// grid of the register-stationary kernel for (n_rows, n_cand) at `k_cols` contraction columns: the prefilter's pair buffer
// has one segment per wave (8 waves x 256 query rows per workgroup up to 400 columns, 4 waves x 128 rows above)
static bool v3_wide(int32_t k_cols) { return v3_prefilter_steps((k_cols + 15) / 16) >= V3_WIDE_FROM; }
static int64_t v3_blocks(int64_t n_rows, int64_t n_cand, int32_t k_cols) {
    const int64_t n_qb = cdiv(n_rows, v3_wide(k_cols) ? 128 : V3_BM), n_cb = cdiv(cdiv(n_cand, V3_BN), v3_prefilter_tiles());
    return 8 * n_qb * cdiv(n_cb, 8);
}

extern "C" int64_t emg_eval_prefilter_ld(int32_t k_cols) {
    const int64_t plain = ((int64_t)k_cols + 63) / 64 * 64;
    if (k_cols <= 0) return 0;
    const int64_t need = v3_prefilter_ld((k_cols + 15) / 16);
    return need > plain ? need : plain;
}

extern "C" int32_t emg_eval_prefilter_max_cols(void) { return 800; }

extern "C" int32_t emg_eval_prefilter_waves(int32_t k_cols) { return v3_wide(k_cols) ? 4 : 8; }

extern "C" int64_t emg_eval_prefilter_segments_k(int64_t n_rows, int64_t n_cand, int32_t k_cols) {
    return (n_rows <= 0 || n_cand <= 0) ? 0 : emg_eval_prefilter_waves(k_cols) * v3_blocks(n_rows, n_cand, k_cols);
}

extern "C" int64_t emg_eval_prefilter_segments(int64_t n_rows, int64_t n_cand) {
    return emg_eval_prefilter_segments_k(n_rows, n_cand, 400);
}

extern "C" int emg_eval_prefilter_f16(int model, const void* q_f16, int64_t ldq, const int32_t* pos_int, const float* band,
                                      int64_t n_rows, const void* ent_f16, int64_t n_cand, int64_t ld_ent,
                                      int64_t ent_offset, int32_t k_pad, float scale, int32_t* cnt_gt, uint64_t* pairs,
                                      uint32_t* pair_count, int64_t pairs_capacity, void* stream) {
    EMG_REQUIRE(q_f16 && pos_int && band && ent_f16 && cnt_gt && pairs && pair_count, "emg_eval_prefilter_f16: null pointer");
    EMG_REQUIRE(n_rows < ((int64_t)1 << 31) && ent_offset + n_cand < ((int64_t)1 << 31), "emg_eval_prefilter_f16: ids must fit 31 bits");
    if (n_rows == 0 || n_cand == 0) return EMG_OK;
    const int64_t n_seg = emg_eval_prefilter_segments_k(n_rows, n_cand, k_pad);
    EMG_REQUIRE(pairs_capacity >= n_seg && pairs_capacity / n_seg < ((int64_t)1 << 31), "emg_eval_prefilter_f16: pair buffer smaller than one entry per wave (%lld)", (long long)n_seg);
    CountBf16Params P{};
    P.Q = (const uint16_t*)q_f16; P.ldq = ldq; P.pos_int = pos_int; P.n_rows = n_rows;
    P.ent = (const uint16_t*)ent_f16; P.n_cand = n_cand; P.ld_ent = ld_ent; P.ent_offset = ent_offset;
    P.k_pad = k_pad; P.scale = scale; P.model = model; P.cnt_gt = cnt_gt; P.cnt_eq = cnt_gt; P.need = 0;
    P.band = band; P.pairs = pairs; P.pair_count = pair_count; P.pair_cap = (uint32_t)(pairs_capacity / n_seg);
    P.n_segments = (uint32_t)n_seg;
    EMG_HIP(hipMemsetAsync(pair_count, 0, (size_t)(n_seg + 1) * sizeof(uint32_t), (hipStream_t)stream));
    return launch_bf16(BF_COUNT, P, (hipStream_t)stream);
}

extern "C" int emg_eval_prefilter_f16_ties(int model, const void* q_f16, int64_t ldq, const int32_t* pos_int, const float* band,
                                           int64_t n_rows, const void* ent_f16, int64_t n_cand, int64_t ld_ent,
                                           int64_t ent_offset, int32_t k_pad, float scale, int32_t* cnt_gt, int32_t* cnt_eq,
                                           uint64_t* pairs, uint32_t* pair_count, int64_t pairs_capacity, void* stream) {
    EMG_REQUIRE(q_f16 && pos_int && band && ent_f16 && cnt_gt && cnt_eq && pairs && pair_count, "emg_eval_prefilter_f16_ties: null pointer");
    EMG_REQUIRE(n_rows < ((int64_t)1 << 31) && ent_offset + n_cand < ((int64_t)1 << 31), "emg_eval_prefilter_f16_ties: ids must fit 31 bits");
    if (n_rows == 0 || n_cand == 0) return EMG_OK;
    const int64_t n_seg = emg_eval_prefilter_segments_k(n_rows, n_cand, k_pad);
    EMG_REQUIRE(pairs_capacity >= n_seg && pairs_capacity / n_seg < ((int64_t)1 << 31), "emg_eval_prefilter_f16_ties: pair buffer smaller than one entry per wave (%lld)", (long long)n_seg);
    CountBf16Params P{};
    P.Q = (const uint16_t*)q_f16; P.ldq = ldq; P.pos_int = pos_int; P.n_rows = n_rows;
    P.ent = (const uint16_t*)ent_f16; P.n_cand = n_cand; P.ld_ent = ld_ent; P.ent_offset = ent_offset;
    P.k_pad = k_pad; P.scale = scale; P.model = model; P.cnt_gt = cnt_gt; P.cnt_eq = cnt_eq; P.need = 0;
    P.band = band; P.pairs = pairs; P.pair_count = pair_count; P.pair_cap = (uint32_t)(pairs_capacity / n_seg);
    P.n_segments = (uint32_t)n_seg; P.ties = 1;
    EMG_HIP(hipMemsetAsync(pair_count, 0, (size_t)(n_seg + 1) * sizeof(uint32_t), (hipStream_t)stream));
    return launch_bf16(BF_COUNT, P, (hipStream_t)stream);
}

extern "C" int emg_eval_prefilter_f16_thr(const void* q_f16, int64_t ldq, const float* thr, int64_t n_rows, const void* ent_f16,
                                          int64_t n_cand, int64_t ld_ent, int64_t ent_offset, int32_t k_pad, int32_t* cnt_gt,
                                          uint64_t* pairs, uint32_t* pair_count, int64_t pairs_capacity, void* stream) {
    EMG_REQUIRE(q_f16 && thr && ent_f16 && cnt_gt && pairs && pair_count, "emg_eval_prefilter_f16_thr: null pointer");
    EMG_REQUIRE(n_rows < ((int64_t)1 << 31) && ent_offset + n_cand < ((int64_t)1 << 31), "emg_eval_prefilter_f16_thr: ids must fit 31 bits");
    if (n_rows == 0 || n_cand == 0) return EMG_OK;
    const int64_t n_seg = emg_eval_prefilter_segments_k(n_rows, n_cand, k_pad);
    EMG_REQUIRE(pairs_capacity >= n_seg && pairs_capacity / n_seg < ((int64_t)1 << 31), "emg_eval_prefilter_f16_thr: pair buffer smaller than one entry per wave (%lld)", (long long)n_seg);
    CountBf16Params P{};
    P.Q = (const uint16_t*)q_f16; P.ldq = ldq; P.pos_int = nullptr; P.n_rows = n_rows;
    P.ent = (const uint16_t*)ent_f16; P.n_cand = n_cand; P.ld_ent = ld_ent; P.ent_offset = ent_offset;
    P.k_pad = k_pad; P.scale = 1.0f; P.model = EMG_DISTMULT; P.cnt_gt = cnt_gt; P.cnt_eq = cnt_gt; P.need = 0;
    P.band = nullptr; P.thr_direct = thr; P.pairs = pairs; P.pair_count = pair_count; P.pair_cap = (uint32_t)(pairs_capacity / n_seg);
    P.n_segments = (uint32_t)n_seg;
    EMG_HIP(hipMemsetAsync(pair_count, 0, (size_t)(n_seg + 1) * sizeof(uint32_t), (hipStream_t)stream));
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
    EMG_REQUIRE(filt_idx, "emg_eval_filter_count_bf16: null filter index");
    const int k16 = (k_int + 15) / 16;   // rows are stored zero-padded to a multiple of 64 (emg_eval_count_bf16's contract)
    EMG_REQUIRE(ldq >= 16 * k16 && ld_ent >= 16 * k16 && ldq % 8 == 0 && ld_ent % 8 == 0 && aligned16(q_bf16) && aligned16(ent_bf16),
                "emg_eval_filter_count_bf16: rows must be 16-byte aligned and zero-padded to a multiple of 16");
    const int64_t waves = n_rows < 4096 ? (n_rows < 64 ? 64 : n_rows) : 4096;   // 32 pairs per wave and trip
    hipLaunchKernelGGL(filter_count_bf16_kernel, dim3((unsigned)cdiv(waves * 64, 256)), dim3(256), 0,
                       (hipStream_t)stream, model, (const uint16_t*)q_bf16, ldq, pos_int, self_ent, n_rows,
                       (const uint16_t*)ent_bf16, n_local, ld_ent, ent_offset, k16, scale, filt_ptr, filt_idx,
                       fcnt_gt, fcnt_eq);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}
