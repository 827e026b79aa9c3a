// emg_group_bucket.hip — the BUCKET form of a training batch's preparation for tables far larger than L2 (round 5).
//
// What it replaces: the counting grouping of emg_group.hip (id kernel + histogram | scan over the table rows | scatter |
// in-segment order) touches two table-sized int32 arrays at random once or twice per contribution — at C3 (|E| = 1M, 360 k
// contributions per batch) 117 MB of partial-line traffic and 360 k device-scope atomics per batch, streamed BESIDE the
// scoring kernel (profiles/r4_z_c3_kernel_stats.md: 174 us of kernel time in four launches, the scoring kernel 0.221 ms alone
// and 0.255-0.269 ms with them).  Same contract (emg_group.hpp: keys ascending, a destination's contributions in ascending
// slot order, singleton flags, segment descriptors, factored source rows) — the reference's counterpart is what
// tf.IndexedSlices + the Keras sparse apply do with the gradient rows of a batch (EmbeddingModel.py:1388-1440, training/sgd.py:97).
//
//   bucket_ids_kernel   one workgroup per chunk of 4096 contribution SLOTS (slot i < B: subject of row i, < 2B: object, else
//                       the Philox draw of negative i - 2B — the same draws, codes and destination arrays as prepare_ids_kernel).
//                       bucket = destination >> sh.  LDS histogram over the <= 4096 buckets of each table (the LDS atomic's
//                       return value is the contribution's rank inside its bucket and chunk), LDS scan, and the (destination,
//                       slot) pairs leave bucket-ordered inside the chunk's own 4096-pair stretch, with the chunk's row of
//                       exclusive bucket offsets — coalesced, no global atomic, nothing to zero.
//   bucket_sort_kernel  one workgroup (1024 threads) per bucket of <= 2048 table rows: gathers its pairs from every chunk
//                       (column b of the offset matrix says where), row histogram + scan + scatter in LDS, orders every
//                       row's slots (insertion sort per row; a row of more than 32 is ranked by the whole workgroup), emits the
//                       singleton / segment / block-task lists the apply kernel works from, and writes keys / vals / flags /
//                       factored source rows in one coalesced pass.  A bucket of more than 8192 contributions (a hub row, a
//                       restricted corruption pool) takes the same phases through global memory — slower, same result.
#include <stdlib.h>
#include <string.h>

#include "emg_group_kernels.hpp"

namespace emg {

struct BucketTable {
    int64_t R; int32_t sh, nb; int32_t nchunks_cap, pad0;   // row stride of bmat = nb + 1
    uint32_t *pd, *ps;           // chunk-ordered pairs: destination, slot
    uint32_t* bmat;              // [chunk][nb + 1] exclusive offsets of the chunk's buckets (last: its valid contributions)
    uint32_t *keys, *vals, *srcrow, *pos_of_slot; float* coef;
    Seg* multi; uint32_t* single; LongTask* tasks; uint32_t task_cap;
    int32_t* arrive; uint32_t* counters;
    uint8_t* flags; const int32_t* fac_codes;   // factored contributions: the batch's codes (sign bit: which query row a negative points at)
};
struct BucketLaunch { BucketTable t[2]; int64_t B, n_ce; uint32_t cap_lds; uint32_t pad0; };

// exclusive scan of a[0 .. L) in place by a workgroup of NT threads (contiguous stretches per thread); a[L] = the total
template <int NT>
__device__ __forceinline__ void block_scan_inplace(uint32_t* a, int L, uint32_t* s_part) {
    const int per = (L + NT - 1) / NT;
    const int i0 = threadIdx.x * per, i1 = min(L, i0 + per);
    uint32_t sum = 0u;
    for (int i = i0; i < i1; ++i) sum += a[i];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) s_part[wv] = inc;
    __syncthreads();
    uint32_t pre = 0u, tot = 0u;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) { const uint32_t v = s_part[w]; if (w < wv) pre += v; tot += v; }
    uint32_t run = pre + inc - sum;
    for (int i = i0; i < i1; ++i) { const uint32_t c = a[i]; a[i] = run; run += c; }
    if (threadIdx.x == 0) a[L] = tot;
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
// 1. ids + chunk-local bucketing
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bucket_ids_kernel(const PrepParams P, const BucketLaunch L) {
    __shared__ uint32_t s_he[kBucketMaxNB + 1], s_hr[kBucketMaxNB + 1];
    __shared__ uint32_t s_part[4];
    const BucketTable &TE = L.t[0], &TR = L.t[1];
    const int64_t B = P.B, n_ce = L.n_ce;
    const int64_t per_side = (int64_t)P.eta * B;
    const unsigned c = blockIdx.x;
    for (int b = threadIdx.x; b <= TE.nb; b += 256) s_he[b] = 0u;
    for (int b = threadIdx.x; b <= TR.nb; b += 256) s_hr[b] = 0u;
    if (c == 0 && threadIdx.x < 8) { TE.counters[threadIdx.x] = 0u; TR.counters[threadIdx.x] = 0u; }
    __syncthreads();
    constexpr int Q = kBucketChunk / 256;
    uint32_t de[Q], re[Q], dr[Q], rr[Q];   // destination and rank inside (chunk, bucket); 0xffffffff = none
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int64_t i = (int64_t)c * kBucketChunk + q * 256 + threadIdx.x;
        de[q] = dr[q] = 0xffffffffu; re[q] = rr[q] = 0u;
        if (i >= n_ce) continue;
        int32_t d;
        uint32_t keep = 0u;
        if (i < 2 * B) {
            const int64_t row = i < B ? i : i - B;
            d = P.pos[3 * row + (i < B ? 0 : 2)];
            if (i < B) {
                const int32_t p = P.pos[3 * row + 1];
                P.dest_rel[row] = p;
                if (p >= 0 && (int64_t)p < TR.R) { dr[q] = (uint32_t)p; rr[q] = atomicAdd(&s_hr[p >> TR.sh], 1u); }
            }
        } else {
            const int64_t j = i - 2 * B;
            const int sd = (int)(j / per_side);
            int64_t jj = j - sd * per_side;   // the draw index restarts per side (one emg_corrupt_codes call each)
            if (P.B_global != B) {             // rows [row_offset, row_offset + B) of a larger batch: draw what IT would
                const int64_t je = jj / B;
                jj = je * P.B_global + P.row_offset + (jj - je * B);
            }
            const int side = P.sides[sd];
            uint32_t idx;
            if (P.inj_repl) {
                idx = (uint32_t)P.inj_repl[j];
                keep = P.inj_mask ? (uint32_t)(P.inj_mask[j] != 0) : 0u;
            } else {
                corruption_draw(P.seed, P.counter0 + (uint64_t)sd, (uint64_t)jj, P.n_choices, &keep, &idx);
            }
            if (side == EMG_SIDE_O) keep = 1u;
            else if (side == EMG_SIDE_S) keep = 0u;
            const uint32_t repl = (P.entities_list ? (uint32_t)P.entities_list[idx] : idx) & 0x7fffffffu;
            P.codes[j] = (int32_t)(repl | (keep << 31));
            d = (int32_t)repl;
        }
        P.dest_ent[i] = d;
        if (d >= 0 && (int64_t)d < TE.R) { de[q] = (uint32_t)d; re[q] = atomicAdd(&s_he[d >> TE.sh], 1u); }
        else if (TE.flags) TE.flags[i] = 0;   // (an id outside the table is dropped: it has no row to update)
    }
    __syncthreads();
    block_scan_inplace<256>(s_he, TE.nb, s_part);
    const bool has_rel = (int64_t)c * kBucketChunk < B;
    if (has_rel) block_scan_inplace<256>(s_hr, TR.nb, s_part);
    uint32_t* me = TE.bmat + (size_t)c * (TE.nb + 1);
    for (int b = threadIdx.x; b <= TE.nb; b += 256) me[b] = s_he[b];
    if (has_rel) {
        uint32_t* mr = TR.bmat + (size_t)c * (TR.nb + 1);
        for (int b = threadIdx.x; b <= TR.nb; b += 256) mr[b] = s_hr[b];
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const uint32_t i = (uint32_t)((int64_t)c * kBucketChunk + q * 256 + threadIdx.x);
        if (de[q] != 0xffffffffu) {
            const size_t at = (size_t)c * kBucketChunk + s_he[de[q] >> TE.sh] + re[q];
            TE.pd[at] = de[q]; TE.ps[at] = i;
        }
        if (dr[q] != 0xffffffffu) {
            const size_t at = (size_t)c * kBucketChunk + s_hr[dr[q] >> TR.sh] + rr[q];
            TR.pd[at] = dr[q]; TR.ps[at] = i;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 2. one workgroup per bucket
// ---------------------------------------------------------------------------------------------------------------
// every pair of bucket b, once: 64 chunks per wave trip (lane = chunk: where its stretch of the bucket starts and ends), the
// trip's pairs then taken 64 at a time (the lane that holds a pair's chunk is found by bisection over the inclusive lengths)
template <typename F>
__device__ __forceinline__ void for_each_pair(const BucketTable& T, int b, int nchunks, F&& f) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int stride = T.nb + 1;
    for (int cg = wv; cg * 64 < nchunks; cg += 16) {
        const int c = cg * 64 + lane;
        uint32_t s = 0u, e = 0u;
        if (c < nchunks) { s = T.bmat[(size_t)c * stride + b]; e = T.bmat[(size_t)c * stride + b + 1]; }
        const uint32_t len = e - s;
        uint32_t inc = len;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        const uint32_t total = __shfl(inc, 63, 64);
        for (uint32_t k = 0; k < total; k += 64u) {
            const uint32_t x = k + lane;
            int lo = 0;
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) {
                const uint32_t v = __shfl(inc, lo + step - 1, 64);
                if (v <= x) lo += step;
            }
            lo = min(lo, 63);
            const uint32_t ss = __shfl(s, lo, 64), excl = __shfl(inc - len, lo, 64);
            if (x < total) {
                const size_t at = (size_t)(cg * 64 + lo) * kBucketChunk + ss + (x - excl);
                f(T.pd[at], T.ps[at]);
            }
        }
    }
}

template <bool LDS>
__device__ __forceinline__ void bucket_sort_body(const BucketTable& T, int b, int nchunks, uint32_t g0, uint32_t count, int64_t B,
                                                 uint32_t* s_ends, uint32_t* s_slot, uint16_t* s_dl, uint16_t* s_long, uint32_t* s_misc) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int RB = 1 << T.sh;
    const uint32_t r0 = (uint32_t)b << T.sh;
    // phase 1: row histogram
    for_each_pair(T, b, nchunks, [&](uint32_t d, uint32_t) { atomicAdd(&s_ends[d - r0], 1u); });
    __syncthreads();
    // phase 2: scan over the bucket's rows + the segment descriptors (a bucket takes its stretch of each list with one atomic)
    const int rpt = RB > 1024 ? RB / 1024 : 1;
    const int row0 = threadIdx.x * rpt;
    uint32_t cj[2] = {0u, 0u};
    uint32_t loc[4] = {0u, 0u, 0u, 0u};   // contributions | segments of 2..kDefer rows | singletons | block tasks
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (j < rpt && row0 + j < RB) cj[j] = s_ends[row0 + j];
        loc[0] += cj[j];
        loc[1] += (cj[j] >= 2u && cj[j] <= (uint32_t)kDeferSegment) ? 1u : 0u;
        loc[2] += cj[j] == 1u ? 1u : 0u;
        loc[3] += cj[j] > (uint32_t)kDeferSegment ? (cj[j] + kLongSegment - 1) / kLongSegment : 0u;
    }
    uint32_t inc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t v = loc[q];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
        inc[q] = v;
        if (lane == 63) s_misc[8 + wv * 4 + q] = v;
    }
    __syncthreads();
    uint32_t wpre[4], tot[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        wpre[q] = 0u; tot[q] = 0u;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const uint32_t v = s_misc[8 + w * 4 + q]; if (w < wv) wpre[q] += v; tot[q] += v; }
    }
    if (threadIdx.x == 0) {
        s_misc[1] = tot[1] ? atomicAdd(T.counters + GC_MULTI, tot[1]) : 0u;
        s_misc[2] = tot[2] ? atomicAdd(T.counters + GC_SINGLE, tot[2]) : 0u;
        s_misc[3] = tot[3] ? atomicAdd(T.counters + GC_TASKS, tot[3]) : 0u;
        s_misc[4] = 0u;   // rows longer than kDeferSegment (ordered by the whole workgroup below)
    }
    __syncthreads();
    uint32_t run[4];
    run[0] = wpre[0] + inc[0] - loc[0];
#pragma unroll
    for (int q = 1; q < 4; ++q) run[q] = s_misc[q] + wpre[q] + inc[q] - loc[q];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (!(j < rpt && row0 + j < RB)) continue;
        const uint32_t c = cj[j], start = run[0];
        s_ends[row0 + j] = start;   // the scatter cursor
        if (c == 0u) continue;
        const uint32_t r = r0 + row0 + j;
        if (c == 1u) T.single[run[2]++] = g0 + start;
        else if (c <= (uint32_t)kDeferSegment) T.multi[run[1]++] = Seg{g0 + start, c, r};
        else {
            const uint32_t nbk = (c + kLongSegment - 1) / kLongSegment;
            const bool room = run[3] + nbk <= T.task_cap;
            for (uint32_t k = 0; k < nbk && run[3] + k < T.task_cap; ++k) T.tasks[run[3] + k] = LongTask{g0 + start, k, room ? c : 0u};
            run[3] += nbk;
            s_long[atomicAdd(&s_misc[4], 1u)] = (uint16_t)(row0 + j);
        }
        run[0] += c;
    }
    __syncthreads();
    // phase 3: scatter — a pair takes the next free position of its row (order inside a row arbitrary until phase 4)
    for_each_pair(T, b, nchunks, [&](uint32_t d, uint32_t slot) {
        const uint32_t dl = d - r0;
        const uint32_t p = atomicAdd(&s_ends[dl], 1u);
        if constexpr (LDS) { s_slot[p] = slot; s_dl[p] = (uint16_t)dl; }
        else { T.vals[g0 + p] = slot; T.keys[g0 + p] = d; }
    });
    __syncthreads();   // (s_ends[row] = END of the row now; global path: the stores above are visible to the workgroup)
    uint32_t* gv = T.vals + g0;
    // phase 4: ascending slot order inside every row = the stable order.  Rows of 2..32: insertion sort by the row's thread
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (!(j < rpt && row0 + j < RB) || cj[j] < 2u || cj[j] > (uint32_t)kDeferSegment) continue;
        const uint32_t start = row0 + j ? s_ends[row0 + j - 1] : 0u, n = cj[j];
        for (uint32_t i = 1; i < n; ++i) {
            const uint32_t key = LDS ? s_slot[start + i] : gv[start + i];
            uint32_t k = i;
            while (k > 0u) {
                const uint32_t prev = LDS ? s_slot[start + k - 1] : gv[start + k - 1];
                if (prev <= key) break;
                if constexpr (LDS) s_slot[start + k] = prev; else gv[start + k] = prev;
                --k;
            }
            if constexpr (LDS) s_slot[start + k] = key; else gv[start + k] = key;
        }
    }
    // longer rows: every slot counts the slots of its row below it (all threads read the same word: an LDS broadcast)
    const uint32_t n_long = s_misc[4];
    for (uint32_t li = 0; li < n_long; ++li) {
        const uint32_t row = s_long[li];
        const uint32_t start = row ? s_ends[row - 1] : 0u, n = s_ends[row] - start;
        if constexpr (LDS) {
            uint32_t mine[kBucketCap / 1024], rk[kBucketCap / 1024];
#pragma unroll
            for (int q = 0; q < kBucketCap / 1024; ++q) {
                const uint32_t i = threadIdx.x + q * 1024u;
                rk[q] = 0u; mine[q] = i < n ? s_slot[start + i] : 0u;
            }
            for (uint32_t u = 0; u < n; ++u) {
                const uint32_t v = s_slot[start + u];
#pragma unroll
                for (int q = 0; q < kBucketCap / 1024; ++q) rk[q] += v < mine[q] ? 1u : 0u;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < kBucketCap / 1024; ++q)
                if (threadIdx.x + q * 1024u < n) s_slot[start + rk[q]] = mine[q];
            __syncthreads();
        } else {   // through global memory: ranked copies go to the (not yet written) srcrow stretch and come back
            uint32_t* tmp = T.srcrow + g0 + start;
            for (uint32_t i = threadIdx.x; i < n; i += 1024u) {
                const uint32_t m = gv[start + i];
                uint32_t r = 0u;
                for (uint32_t u = 0; u < n; ++u) r += gv[start + u] < m ? 1u : 0u;
                tmp[r] = m;
            }
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < n; i += 1024u) gv[start + i] = tmp[i];
            __syncthreads();
        }
    }
    __syncthreads();
    // phase 5: the grouping's outputs, one coalesced pass over the bucket's sorted positions
    for (uint32_t p = threadIdx.x; p < count; p += 1024u) {
        uint32_t slot, dl;
        if constexpr (LDS) { slot = s_slot[p]; dl = s_dl[p]; }
        else { slot = gv[p]; dl = T.keys[g0 + p] - r0; }
        const uint32_t start = dl ? s_ends[dl - 1] : 0u, len = s_ends[dl] - start;
        const uint32_t at = g0 + p;
        if constexpr (LDS) { T.keys[at] = r0 + dl; T.vals[at] = slot; }
        if (T.flags) T.flags[slot] = len == 1u ? 1 : 0;
        if (T.fac_codes) {
            const uint32_t fB = (uint32_t)B;
            if (slot < 2u * fB) { T.srcrow[at] = slot; T.coef[at] = 1.f; }   // subject / object rows are stored in full
            else {
                const uint32_t i = slot - 2u * fB;
                T.srcrow[at] = (T.fac_codes[i] < 0 ? 2u : 3u) * fB + i % fB;
                T.pos_of_slot[i] = at;
            }
        }
    }
}

__global__ __launch_bounds__(1024) void bucket_sort_kernel(const BucketLaunch L) {
    __shared__ uint32_t s_ends[kBucketRowsMax];
    __shared__ uint32_t s_slot[kBucketCap];
    __shared__ uint16_t s_dl[kBucketCap];
    __shared__ uint16_t s_long[kBucketRowsMax];
    __shared__ uint32_t s_misc[8 + 64];   // 0: g0 | 1..3: list bases | 4: long rows | 5: count | 8..: wave partials
    const int ti = blockIdx.x < (unsigned)L.t[0].nb ? 0 : 1;
    const BucketTable& T = L.t[ti];
    const int b = (int)blockIdx.x - (ti ? L.t[0].nb : 0);
    const int64_t n = ti ? L.B : L.n_ce;
    const int nchunks = (int)((n + kBucketChunk - 1) / kBucketChunk);
    const int RB = 1 << T.sh;
    for (int r = threadIdx.x; r < RB; r += 1024) s_ends[r] = 0u;
    if (threadIdx.x == 0) { s_misc[0] = 0u; s_misc[5] = 0u; }
    // housekeeping of the apply that follows: per-segment block counters, the window path's task list
    const int64_t n_arr = n / kLongSegment + 1, per = (n_arr + T.nb - 1) / T.nb;
    for (int64_t t = (int64_t)b * per + threadIdx.x; t < min(n_arr, ((int64_t)b + 1) * per); t += 1024) T.arrive[t] = 0;
    if (b == 0 && threadIdx.x < 2) T.counters[GC_LONG_COUNT + threadIdx.x] = 0u;
    __syncthreads();
    // where the bucket starts in the sorted order (everything in lower buckets, over all chunks) and how much it holds
    {
        const int stride = T.nb + 1;
        uint32_t g = 0u, cnt = 0u;
        for (int c = threadIdx.x; c < nchunks; c += 1024) {
            const uint32_t s = T.bmat[(size_t)c * stride + b], e = T.bmat[(size_t)c * stride + b + 1];
            g += s; cnt += e - s;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { g += __shfl_xor(g, o, 64); cnt += __shfl_xor(cnt, o, 64); }
        if ((threadIdx.x & 63) == 0 && (g | cnt)) { atomicAdd(&s_misc[0], g); atomicAdd(&s_misc[5], cnt); }
    }
    __syncthreads();
    const uint32_t g0 = s_misc[0], count = s_misc[5];
    if (b == T.nb - 1 && threadIdx.x == 0) T.counters[GC_VALID] = g0 + count;
    __syncthreads();
    if (count == 0u) return;
    if (count <= L.cap_lds) bucket_sort_body<true>(T, b, nchunks, g0, count, L.B, s_ends, s_slot, s_dl, s_long, s_misc);
    else bucket_sort_body<false>(T, b, nchunks, g0, count, L.B, s_ends, s_slot, s_dl, s_long, s_misc);
}

static void fill_bucket_table(BucketTable& T, const GroupWs& w, const BucketGeo& g, int64_t R, uint8_t* flags, const int32_t* fac_codes) {
    T = BucketTable{};
    T.R = R; T.sh = g.sh; T.nb = g.nb; T.nchunks_cap = g.nchunks;
    T.pd = w.tmpv; T.ps = w.tmpv2; T.bmat = w.bmat;
    T.keys = w.keys; T.vals = w.vals; T.srcrow = w.srcrow; T.pos_of_slot = w.pos_of_slot; T.coef = w.coef;
    T.multi = w.multi; T.single = w.single; T.tasks = w.tasks; T.task_cap = w.task_cap;
    T.arrive = w.arrive; T.counters = w.counters; T.flags = flags; T.fac_codes = fac_codes;
}

// the bucket form of emg_prepare_batch (S: its validated stages); false = not eligible, the counting form runs
bool bucket_eligible(const emg_prepare_args* a, const PrepStages& S) {
    if (!S.both || a->ctl || a->n_extra_ent != 0 || a->n_extra_rel != 0 || !group_backend_bucket(a->n_ent)) return false;
    if (!S.we.bmat || !S.wr.bmat || !S.we.tmpv2 || !S.wr.tmpv2) return false;
    return S.cap_ce < ((int64_t)1 << 31);
}

int bucket_prepare(const emg_prepare_args* a, const PrepStages& S, hipStream_t st) {
    // test aid: a smaller LDS capacity sends buckets through the global-memory form
    const char* cap_s = getenv("EMG_BUCKET_CAP");
    const long cap_v = cap_s ? atol(cap_s) : 0;
    const uint32_t cap_env = (uint32_t)(cap_v > 0 && cap_v < kBucketCap ? cap_v : kBucketCap);
    const BucketGeo ge = bucket_geometry(S.cap_ce, a->n_ent), gr = bucket_geometry(S.cap_cr, a->n_rel);
    EMG_REQUIRE(ge.ok && gr.ok, "emg_prepare_batch: bucket grouping without a geometry");
    BucketLaunch L{};
    fill_bucket_table(L.t[0], S.we, ge, a->n_ent, a->single_flags, a->factored ? a->codes : nullptr);
    fill_bucket_table(L.t[1], S.wr, gr, a->n_rel, nullptr, nullptr);
    L.B = a->B; L.n_ce = S.n_ce; L.cap_lds = cap_env;
    PrepParams P = S.prep;
    const unsigned nchunks = (unsigned)cdiv(S.n_ce, kBucketChunk);
    hipLaunchKernelGGL(bucket_ids_kernel, dim3(nchunks), dim3(256), 0, st, P, L);
    EMG_LAUNCH_CHECK();
    hipLaunchKernelGGL(bucket_sort_kernel, dim3((unsigned)(ge.nb + gr.nb)), dim3(1024), 0, st, L);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

}  // namespace emg
