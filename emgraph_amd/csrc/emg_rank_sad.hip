// emg_rank_sad.hip — TransE-L1 1-vs-all ranking at integer speed: EXACT ranks through a 16-bit fixed-point prefilter.
//
// The exact TransE-L1 kernel (emg_rank.hip::count_transe_big_kernel) spends 1.5 VALU instructions per (query,
// entity, k) element and sits at ~90 % of the VALU issue rate: the f32 form has nothing left.  CDNA has a
// sum-of-absolute-differences instruction, v_sad_u16: D = |a.lo - b.lo| + |a.hi - b.hi| + c on packed 16-bit
// unsigned operands — ONE instruction for TWO elements of an L1 distance.  So:
//   1. every coordinate x of the query rows and of the entity table is mapped to u = rint((x + R) / delta),
//      delta = 2R / 65535, R >= every |x| involved (R = (max|ent| + max|rel|)(1 + 1e-6); a query row is s+p or o-p,
//      EmbeddingModel.py:1856-1866 through emg_rank.hip::build_queries_kernel);
//   2. S = sum_k |u_q - u_e| (exact integer arithmetic) satisfies |L1(q, e) - delta S| <= k delta (1 + 1e-10);
//   3. the reference compares int(score * 1e5) with the positive's (EmbeddingModel.py:2010-2033), score = -L1 in
//      the canonical f32 chain A (emg_rank.hip::chain_score), |A - L1| <= gamma L1, gamma = (k+2) 2^-24: per query
//      row two integer thresholds LO, HI follow such that S < LO proves int(-A 1e5) > pos_int ("greater": counted
//      here) and S > HI proves int(-A 1e5) < pos_int (dropped); the candidates in between — a few per thousand —
//      are EMITTED as (row, entity) pairs in the layout of the half-precision prefilter (emg_rank_bf16.hip) and
//      re-scored by emg_eval_rescore_pairs with the canonical f32 chain.
// The counters, and therefore the ranks, equal the exact kernel's bit for bit; a wrong bound can only cost speed
// if it is too wide, never correctness if every inequality above holds — tests/test_hip_kernels.py checks the ranks
// against precision 0 on tables with planted ties, duplicates of the positive and near-threshold candidates.
#include "emg_common.hpp"

namespace emg {

// ---------------------------------------------------------------------------------------------
// range of the fixed-point map: out[0] = max|ent|, out[1] = max|rel| (doubles; caller zeroes them)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sad_absmax_kernel(const float* __restrict__ src, int64_t n_rows, int64_t ld,
                                                         int k_int, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    float m = 0.f;
    for (int64_t r = wave; r < n_rows; r += n_waves)
        for (int c = lane; c < k_int; c += 64) m = fmaxf(m, fabsf(src[r * ld + c]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0 && m > 0.f)   // non-negative doubles order like their bit patterns
        atomicMax(reinterpret_cast<unsigned long long*>(out), (unsigned long long)__double_as_longlong((double)m));
}

__device__ __forceinline__ double sad_half_range(const double* __restrict__ range) {
    const double r = (range[0] + range[1]) * (1.0 + 1e-6);
    return r > 1e-30 ? r : 1e-30;
}

// u16 image of the rows: column c < k_int -> clamp(rint((x + R) / delta)), padding columns -> 0 on both sides
__global__ __launch_bounds__(256) void sad_quantize_kernel(const float* __restrict__ src, int64_t n_rows, int64_t ld_src,
                                                           int k_int, const double* __restrict__ range,
                                                           uint16_t* __restrict__ dst, int64_t ld_dst) {
    const double R = sad_half_range(range), inv = 65535.0 / (2.0 * R);
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int n_dw = (int)(ld_dst / 2);
    for (int64_t r = wave; r < n_rows; r += n_waves) {
        const float* x = src + r * ld_src;
        unsigned* out = reinterpret_cast<unsigned*>(dst + r * ld_dst);
        for (int d = lane; d < n_dw; d += 64) {
            unsigned w = 0u;
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (2 * d + h < k_int) {
                    double u = rint(((double)x[2 * d + h] + R) * inv);
                    u = u < 0.0 ? 0.0 : (u > 65535.0 ? 65535.0 : u);
                    w |= (unsigned)u << (16 * h);
                }
            out[d] = w;
        }
    }
}

// per query row: S < lo  =>  counted as "greater";  S > hi  =>  dropped;  lo <= S <= hi  =>  re-scored exactly
__global__ void sad_thresholds_kernel(const int32_t* __restrict__ pos_int, int64_t n_rows, int k_int,
                                      const double* __restrict__ range, uint32_t* __restrict__ lo,
                                      uint32_t* __restrict__ hi) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const double R = sad_half_range(range), delta = 2.0 * R / 65535.0;
    const double u = 5.9604644775390625e-08;                  // 2^-24
    const double gamma = (double)(k_int + 2) * u * 1.01;      // the f32 chain: k subtractions and k additions of non-negative terms
    const double kd = (double)k_int * delta * (1.0 + 1e-6);   // sum over k of the two half-step roundings
    const double m = -(double)pos_int[r];                     // int(-A 1e5) > pos  <=>  floor(fl(A 1e5)) < m  <=>  fl(A 1e5) < m
    // greater is certain when (delta S + kd)(1 + gamma)(1 + u) 1e5 < m
    const double lo_v = (m * 1e-5 / ((1.0 + gamma) * (1.0 + u) * (1.0 + 1e-9)) - kd) / delta;
    // less is certain when (delta S - kd)(1 - gamma)(1 - u) 1e5 >= m + 1
    const double hi_v = ((m + 1.0) * 1e-5 * (1.0 + 1e-9) / ((1.0 - gamma) * (1.0 - u)) + kd) / delta;
    const double lo_f = floor(lo_v), hi_f = ceil(hi_v);
    lo[r] = lo_f <= 0.0 ? 0u : (lo_f >= 4294967295.0 ? 0xffffffffu : (uint32_t)lo_f);
    hi[r] = hi_f <= 0.0 ? 0u : (hi_f >= 4294967295.0 ? 0xffffffffu : (uint32_t)hi_f);
}

// ---------------------------------------------------------------------------------------------
// the count kernel: 128 query rows x 128 entities per workgroup, 8 x 8 per thread, k in tiles of 8 dwords (16
// coordinates) double-buffered in LDS (the next tile's 16-byte global loads fly under the arithmetic).  Per dword
// step a thread reads 4 x 16 bytes from LDS and issues 64 v_sad_u16 for 128 coordinate pairs.
// ---------------------------------------------------------------------------------------------
struct SadParams {
    const uint32_t* Q; int64_t ldq;         // u16 pairs, row strides in dwords
    const uint32_t* lo; const uint32_t* hi; int64_t n_rows;
    const uint32_t* ent; int64_t ld_ent; int64_t n_cand; int64_t ent_offset;
    int32_t kw;                             // dwords per row actually summed (multiple of SW)
    int32_t* cnt_gt;
    uint64_t* pairs; uint32_t* pair_count; uint32_t pair_cap; uint32_t n_segments;
    int64_t n_qb; int64_t n_cb; int64_t n_tiles; int32_t tiles_per_chunk;
};

constexpr int SQ = 128, SE = 128, SW = 8, S_TILES = 32;

__global__ __launch_bounds__(256, 2) void count_sad_kernel(const SadParams P) {
    __shared__ __attribute__((aligned(16))) uint32_t Qs[2][SW * SQ];
    __shared__ __attribute__((aligned(16))) uint32_t Es[2][SW * SE];
    __shared__ __attribute__((aligned(16))) uint32_t lo_s[SQ];
    __shared__ __attribute__((aligned(16))) uint32_t hi_s[SQ];
    __shared__ unsigned cnt_s[SQ];

    const int64_t id = blockIdx.x;
    const int64_t xcd = id & 7, slot = id >> 3;
    const int64_t qb = slot % P.n_qb;
    const int64_t cb = xcd + 8 * (slot / P.n_qb);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t seg = (uint32_t)blockIdx.x * 4u + (uint32_t)wave;
    if (cb >= P.n_cb) return;   // (its segments stay at count 0: the caller cleared pair_count)

    const int tq = tid & 15, te = tid >> 4;
    const int lrow = tid & 127, lh = tid >> 7;   // loader: row lrow, dwords 4 lh .. 4 lh + 3 of the k tile

    if (tid < SQ) {
        const int64_t qr = qb * SQ + tid;
        const bool ok = qr < P.n_rows;           // rows past the end: never greater (lo 0), never emitted (hi < lo is impossible,
        lo_s[tid] = ok ? P.lo[qr] : 0u;          //   so they are masked by row_ok below)
        hi_s[tid] = ok ? P.hi[qr] : 0u;
        cnt_s[tid] = 0u;
    }
    const int64_t qrow_g = min(qb * SQ + lrow, P.n_rows - 1);
    const uint32_t* qptr = P.Q + qrow_g * P.ldq + 4 * lh;
    const int nkt = P.kw / SW;

    unsigned row_ok = 0u;   // bit a: this thread's a-th query row exists
#pragma unroll
    for (int a = 0; a < 8; ++a) row_ok |= (qb * SQ + (a < 4 ? 0 : 64) + 4 * tq + (a & 3) < P.n_rows) ? (1u << a) : 0u;

    unsigned cnt[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) cnt[a] = 0u;
    uint64_t* const pair_base = P.pairs + (uint64_t)seg * P.pair_cap;
    unsigned pair_n = 0u, pair_over = 0u;

    const int64_t tile0 = cb * P.tiles_per_chunk;
    const int64_t tile1 = min(tile0 + (int64_t)P.tiles_per_chunk, P.n_tiles);
    for (int64_t tile = tile0; tile < tile1; ++tile) {
        const int64_t el = min(tile * SE + lrow, P.n_cand - 1);
        const uint32_t* eptr = P.ent + el * P.ld_ent + 4 * lh;
        uint4 gq, ge;
        auto fetch = [&](int kt) {
            gq = *reinterpret_cast<const uint4*>(qptr + kt * SW);
            ge = *reinterpret_cast<const uint4*>(eptr + kt * SW);
        };
        auto stage = [&](int buf) {   // registers -> LDS, k-major
            const int kl = 4 * lh;
            Qs[buf][(kl + 0) * SQ + lrow] = gq.x; Qs[buf][(kl + 1) * SQ + lrow] = gq.y;
            Qs[buf][(kl + 2) * SQ + lrow] = gq.z; Qs[buf][(kl + 3) * SQ + lrow] = gq.w;
            Es[buf][(kl + 0) * SE + lrow] = ge.x; Es[buf][(kl + 1) * SE + lrow] = ge.y;
            Es[buf][(kl + 2) * SE + lrow] = ge.z; Es[buf][(kl + 3) * SE + lrow] = ge.w;
        };
        uint32_t acc[8][8];
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[a][b] = 0u;
        fetch(0);
        __syncthreads();       // (the previous tile's readers are done with buffer 0, and lo_s / hi_s are staged)
        stage(0);
        for (int kt = 0; kt < nkt; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nkt) fetch(kt + 1);
            __syncthreads();
#pragma unroll 4
            for (int k = 0; k < SW; ++k) {
                const uint4 q0 = *reinterpret_cast<const uint4*>(&Qs[buf][k * SQ + 4 * tq]);
                const uint4 q1 = *reinterpret_cast<const uint4*>(&Qs[buf][k * SQ + 64 + 4 * tq]);
                const uint4 e0 = *reinterpret_cast<const uint4*>(&Es[buf][k * SE + 4 * te]);
                const uint4 e1 = *reinterpret_cast<const uint4*>(&Es[buf][k * SE + 64 + 4 * te]);
                const uint32_t q[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                const uint32_t e[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
#pragma unroll
                for (int a = 0; a < 8; ++a)
#pragma unroll
                    for (int b = 0; b < 8; ++b) acc[a][b] = __builtin_amdgcn_sad_u16(q[a], e[b], acc[a][b]);
            }
            if (kt + 1 < nkt) stage(1 - buf);
        }
        // ---- epilogue: count the certain "greater", collect the undecided ------------------------------------
        const bool full = (tile + 1) * SE <= P.n_cand;   // block-uniform
        unsigned long long und = 0ull;                   // bit 8 a + b
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const int ql = (a < 4 ? 0 : 64) + 4 * tq + (a & 3);
            const uint32_t lo = lo_s[ql], hi = hi_s[ql];
            const bool rok = (row_ok >> a) & 1u;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool cok = full || tile * SE + (b < 4 ? 0 : 64) + 4 * te + (b & 3) < P.n_cand;
                const uint32_t s = acc[a][b];
                cnt[a] += (unsigned)(cok && s < lo);
                und |= (cok && rok && s >= lo && s <= hi) ? (1ull << (8 * a + b)) : 0ull;
            }
        }
        if (__any(und != 0ull)) {   // wave-uniform
            const int n_l = __popcll(und);
            int incl = n_l;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int up = __shfl_up(incl, off, 64);
                if (lane >= off) incl += up;
            }
            const int total = __shfl(incl, 63, 64);
            if (pair_n + (unsigned)total > P.pair_cap) {
                pair_over = 1u;   // the caller redoes this query tile with the exact kernel
            } else {
                uint64_t* dst = pair_base + pair_n + (unsigned)(incl - n_l);
                pair_n += (unsigned)total;
                const uint64_t row0 = (uint64_t)(qb * SQ + 4 * tq);
                const uint64_t col0 = (uint64_t)(P.ent_offset + tile * SE + 4 * te);
                while (und) {
                    const int bit = __ffsll((long long)und) - 1;
                    und &= und - 1ull;
                    const int a = bit >> 3, b = bit & 7;
                    *dst++ = ((row0 + (uint64_t)((a < 4 ? 0 : 64) + (a & 3))) << 32) |
                             (uint32_t)(col0 + (uint64_t)((b < 4 ? 0 : 64) + (b & 3)));
                }
            }
        }
    }
    if (lane == 0) {
        P.pair_count[seg] = pair_n;
        if (pair_over) atomicOr(P.pair_count + P.n_segments, 1u);
    }
    // 16 threads share a query row
#pragma unroll
    for (int a = 0; a < 8; ++a)
        if (cnt[a]) atomicAdd(&cnt_s[(a < 4 ? 0 : 64) + 4 * tq + (a & 3)], cnt[a]);
    __syncthreads();
    if (tid < SQ) {
        const int64_t qr = qb * SQ + tid;
        const unsigned c = cnt_s[tid];
        if (qr < P.n_rows && c) atomicAdd(&P.cnt_gt[qr], (int)c);
    }
}

static int64_t sad_blocks(int64_t n_rows, int64_t n_cand) {
    const int64_t n_qb = cdiv(n_rows, SQ), n_cb = cdiv(cdiv(n_cand, SE), S_TILES);
    return 8 * n_qb * cdiv(n_cb, 8);
}

}  // namespace emg

using namespace emg;

extern "C" int emg_eval_sad_range(const float* ent, int64_t n_ent, int64_t ld_ent, const float* rel, int64_t n_rel,
                                  int64_t ld_rel, int32_t k_int, double* range, void* stream) {
    EMG_REQUIRE(ent && rel && range && k_int > 0 && n_ent >= 0 && n_rel >= 0 && ld_ent >= k_int && ld_rel >= k_int,
                "emg_eval_sad_range: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    EMG_HIP(hipMemsetAsync(range, 0, 2 * sizeof(double), st));
    auto blocks = [&](int64_t n) { const int64_t b = cdiv(n, 4 * 8); return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); };
    if (n_ent > 0) hipLaunchKernelGGL(sad_absmax_kernel, dim3(blocks(n_ent)), dim3(256), 0, st, ent, n_ent, ld_ent, k_int, range);
    if (n_rel > 0) hipLaunchKernelGGL(sad_absmax_kernel, dim3(blocks(n_rel)), dim3(256), 0, st, rel, n_rel, ld_rel, k_int, range + 1);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

extern "C" int64_t emg_eval_sad_ld(int32_t k_int) { return k_int <= 0 ? 0 : (int64_t)(k_int + 2 * SW - 1) / (2 * SW) * (2 * SW); }

extern "C" int emg_eval_sad_quantize(const float* src, int64_t n_rows, int64_t ld_src, int32_t k_int, const double* range,
                                     void* dst_u16, int64_t ld_dst, void* stream) {
    EMG_REQUIRE(n_rows >= 0 && k_int > 0 && ld_src >= k_int, "emg_eval_sad_quantize: bad sizes");
    EMG_REQUIRE(ld_dst >= emg_eval_sad_ld(k_int) && ld_dst % 8 == 0,
                "emg_eval_sad_quantize: ld_dst must be a multiple of 8 and at least emg_eval_sad_ld(k_int) = %lld", (long long)emg_eval_sad_ld(k_int));
    if (n_rows == 0) return EMG_OK;
    EMG_REQUIRE(src && range && dst_u16 && aligned16(dst_u16), "emg_eval_sad_quantize: null or misaligned pointer");
    const int64_t b = cdiv(n_rows, 4 * 4);   // 4 rows per wave and trip
    hipLaunchKernelGGL(sad_quantize_kernel, dim3((unsigned)(b > 65536 ? 65536 : b)), dim3(256), 0, (hipStream_t)stream, src, n_rows,
                       ld_src, k_int, range, (uint16_t*)dst_u16, ld_dst);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

extern "C" int emg_eval_sad_thresholds(const int32_t* pos_int, int64_t n_rows, int32_t k_int, const double* range,
                                       uint32_t* lo, uint32_t* hi, void* stream) {
    EMG_REQUIRE(n_rows >= 0 && k_int > 0, "emg_eval_sad_thresholds: bad sizes");
    if (n_rows == 0) return EMG_OK;
    EMG_REQUIRE(pos_int && range && lo && hi, "emg_eval_sad_thresholds: null pointer");
    hipLaunchKernelGGL(sad_thresholds_kernel, dim3((unsigned)cdiv(n_rows, 256)), dim3(256), 0, (hipStream_t)stream, pos_int,
                       n_rows, k_int, range, lo, hi);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

extern "C" int64_t emg_eval_sad_segments(int64_t n_rows, int64_t n_cand) {
    return (n_rows <= 0 || n_cand <= 0) ? 0 : 4 * sad_blocks(n_rows, n_cand);
}

extern "C" int emg_eval_prefilter_sad(const void* q_u16, int64_t ldq, const uint32_t* lo, const uint32_t* hi,
                                      int64_t n_rows, const void* ent_u16, int64_t n_cand, int64_t ld_ent,
                                      int64_t ent_offset, int32_t k_int, int32_t* cnt_gt, uint64_t* pairs,
                                      uint32_t* pair_count, int64_t pairs_capacity, void* stream) {
    EMG_REQUIRE(n_rows >= 0 && n_cand >= 0 && k_int > 0, "emg_eval_prefilter_sad: bad sizes");
    if (n_rows == 0 || n_cand == 0) return EMG_OK;
    EMG_REQUIRE(q_u16 && lo && hi && ent_u16 && cnt_gt && pairs && pair_count, "emg_eval_prefilter_sad: null pointer");
    const int64_t kp = emg_eval_sad_ld(k_int);
    EMG_REQUIRE(ldq >= kp && ld_ent >= kp && ldq % 8 == 0 && ld_ent % 8 == 0 && aligned16(q_u16) && aligned16(ent_u16),
                "emg_eval_prefilter_sad: rows must be 16-byte aligned u16 images of at least emg_eval_sad_ld(k_int) = %lld columns",
                (long long)kp);
    EMG_REQUIRE(n_rows < ((int64_t)1 << 31) && ent_offset + n_cand < ((int64_t)1 << 31), "emg_eval_prefilter_sad: ids must fit 31 bits");
    // S <= 65535 k must fit the 32-bit accumulator
    EMG_REQUIRE(k_int <= 65536, "emg_eval_prefilter_sad: k_int %d too wide for the 32-bit sums", k_int);
    const int64_t n_seg = emg_eval_sad_segments(n_rows, n_cand);
    EMG_REQUIRE(pairs_capacity >= n_seg && pairs_capacity / n_seg < ((int64_t)1 << 31),
                "emg_eval_prefilter_sad: pair buffer smaller than one entry per wave (%lld)", (long long)n_seg);
    SadParams P{};
    P.Q = (const uint32_t*)q_u16; P.ldq = ldq / 2; P.lo = lo; P.hi = hi; P.n_rows = n_rows;
    P.ent = (const uint32_t*)ent_u16; P.ld_ent = ld_ent / 2; P.n_cand = n_cand; P.ent_offset = ent_offset;
    P.kw = (int32_t)(kp / 2); P.cnt_gt = cnt_gt;
    P.pairs = pairs; P.pair_count = pair_count; P.pair_cap = (uint32_t)(pairs_capacity / n_seg); P.n_segments = (uint32_t)n_seg;
    P.n_qb = cdiv(n_rows, SQ); P.n_tiles = cdiv(n_cand, SE); P.tiles_per_chunk = S_TILES; P.n_cb = cdiv(P.n_tiles, S_TILES);
    const int64_t blocks = sad_blocks(n_rows, n_cand);
    EMG_REQUIRE(blocks < ((int64_t)1 << 29), "emg_eval_prefilter_sad: grid too large");
    hipStream_t st = (hipStream_t)stream;
    EMG_HIP(hipMemsetAsync(pair_count, 0, (size_t)(n_seg + 1) * sizeof(uint32_t), st));
    hipLaunchKernelGGL(count_sad_kernel, dim3((unsigned)blocks), dim3(256), 0, st, P);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}
