// emg_group_bucket.hip — the BUCKET form of a training batch's preparation for tables far larger than L2 (round 5).
//
// What it replaces: the counting grouping of emg_group.hip (id kernel + histogram | scan over the table rows | scatter |
// in-segment order) touches two table-sized int32 arrays at random once or twice per contribution — at C3 (|E| = 1M, 360 k
// contributions per batch) 117 MB of partial-line traffic and 360 k device-scope atomics per batch in four launches: 69 us
// alone, 174 us of kernel time streamed BESIDE the scoring kernel, which runs 0.200 ms alone and 0.23-0.27 ms with them
// (profiles/r4_z_c3_kernel_stats.md, r5_b_*).  Same contract (emg_group.hpp: keys ascending, a destination's contributions in
// ascending slot order, singleton flags, segment descriptors, factored source rows) — the reference's counterpart is what
// tf.IndexedSlices + the Keras sparse apply do with the gradient rows of a batch (EmbeddingModel.py:1388-1440, training/sgd.py:97).
//
//   bucket_ids_kernel   one workgroup per chunk of >= 1024 contribution SLOTS (slot i < B: subject of row i, < 2B: object, else
//                       the Philox draw of negative i - 2B — the same draws, codes and destination arrays as prepare_ids_kernel).
//                       bucket = destination >> sh.  LDS histogram over the <= 4096 buckets of each table, LDS scan, and the
//                       (destination, slot) pairs leave bucket-ordered inside the chunk's own stretch, with the chunk's column
//                       of exclusive bucket offsets (bmat[bucket][chunk]) — no global atomic, nothing to zero.
//   bucket_sort_kernel  one workgroup (256 threads: a wave per SIMD) per bucket of <= 2048 table rows: two coalesced rows of the
//                       offset matrix say where its pairs lie in every chunk; they are gathered ONCE into LDS, then row
//                       histogram + scan + scatter in LDS, every row's slots ordered (insertion sort per row; a row of more
//                       than 32 is ranked by the whole workgroup), the singleton / segment / block-task lists the apply kernel
//                       works from, and keys / vals / flags / factored source rows in one coalesced pass.  Four global round
//                       trips per workgroup.  A bucket of more than 4096 contributions (a hub row, a restricted corruption
//                       pool) takes the same phases through global memory — slower, same result.
#include <stdlib.h>
#include <string.h>

#include "emg_group_kernels.hpp"

namespace emg {

constexpr int kBT = 256;                    // threads of both kernels
constexpr int kLdm = kBucketChunksMax;      // row stride of the offset matrix
constexpr uint32_t kSlot = 0x7fffffffu;     // a pair's slot; its top bit: the code's sign bit of a negative (see IdOut)

struct BucketTable {
    int64_t R; int32_t sh, nb;
    uint32_t *pd, *ps;           // chunk-ordered pairs: destination, slot
    uint32_t* bmat;              // [nb + 1][kLdm]: bmat[b][c] = pairs of chunk c in buckets below b (row nb: its valid contributions)
    uint32_t *keys, *vals, *srcrow, *pos_of_slot; float* coef;
    Seg* multi; uint32_t* single; LongTask* tasks; uint32_t task_cap;
    int32_t* arrive; uint32_t* counters;
    uint8_t* flags; const int32_t* fac_codes;   // factored contributions: the batch's codes (sign bit: which query row a negative points at)
    uint32_t* off;               // tables of <= kDenseHereMaxRows rows: the exclusive row offsets [R + 1] the apply's in-launch dense pass reads (else nullptr)
};
struct BucketLaunch { BucketTable t[2]; int64_t B, n_ce; uint32_t cap_lds; int32_t chunk_log; };

// exclusive scan of a[0 .. L) in place by the workgroup (contiguous stretches per thread); returns the total to every thread
__device__ __forceinline__ uint32_t block_scan_inplace(uint32_t* a, int L, uint32_t* s_part) {
    const int per = (L + kBT - 1) / kBT;
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
    for (int w = 0; w < kBT / 64; ++w) { const uint32_t v = s_part[w]; if (w < wv) pre += v; tot += v; }
    uint32_t run = pre + inc - sum;
    for (int i = i0; i < i1; ++i) { const uint32_t c = a[i]; a[i] = run; run += c; }
    __syncthreads();
    return tot;
}

// ---------------------------------------------------------------------------------------------------------------
// 1. ids + chunk-local bucketing
// ---------------------------------------------------------------------------------------------------------------
// entity destination of a slot; relation destination (slots below B; else -1); the code's sign bit of a negative's slot (which
// query row a factored negative points at: it travels in the top bit of the pair's slot, so the bucket kernel reads no codes)
struct IdOut { int32_t d; int32_t p; uint32_t kb; };

__device__ __forceinline__ IdOut slot_ids(const PrepParams& P, int64_t i, int64_t B, int64_t per_side) {
    IdOut o{-1, -1, 0u};
    if (i < 2 * B) {
        const int64_t row = i < B ? i : i - B;
        o.d = P.pos[3 * row + (i < B ? 0 : 2)];
        if (i < B) { o.p = P.pos[3 * row + 1]; P.dest_rel[row] = o.p; }
    } else {
        const int64_t j = i - 2 * B;
        const int sd = (int)(j / per_side);
        int64_t jj = j - sd * per_side;   // the draw index restarts per side (one emg_corrupt_codes call each)
        if (P.B_global != B) {             // rows [row_offset, row_offset + B) of a larger batch: draw what IT would
            const int64_t je = jj / B;
            jj = je * P.B_global + P.row_offset + (jj - je * B);
        }
        const int side = P.sides[sd];
        uint32_t idx, keep;
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
        o.d = (int32_t)repl; o.kb = keep << 31;
    }
    P.dest_ent[i] = o.d;
    return o;
}

// SMALL: chunks of kBucketChunkMin slots — a thread's four ids stay in registers between the histogram and the placement
template <bool SMALL>
__global__ __launch_bounds__(kBT) void bucket_ids_kernel(const PrepParams P, const BucketLaunch L) {
    // dynamic LDS, sized for THIS batch's bucket counts (C3: 2 KB, not the 33 KB of two 4097-bin histograms): the workgroups start
    // beside scoring workgroups that hold most of a CU's LDS (the window forms' stash)
    extern __shared__ uint32_t s_hist[];
    __shared__ uint32_t s_part[kBT / 64];
    const BucketTable &TE = L.t[0], &TR = L.t[1];
    uint32_t* const s_he = s_hist;
    uint32_t* const s_hr = s_hist + TE.nb + 1;
    const int64_t B = P.B, n_ce = L.n_ce;
    const int64_t per_side = (int64_t)P.eta * B;
    const unsigned c = blockIdx.x;
    const int64_t i_base = (int64_t)c << L.chunk_log;
    const int trips = SMALL ? kBucketChunkMin / kBT : (1 << L.chunk_log) / kBT;
    const bool has_rel = i_base < B;
    for (int b = threadIdx.x; b <= TE.nb; b += kBT) s_he[b] = 0u;
    if (has_rel) for (int b = threadIdx.x; b <= TR.nb; b += kBT) s_hr[b] = 0u;
    if (c == 0 && threadIdx.x < 8) { TE.counters[threadIdx.x] = 0u; TR.counters[threadIdx.x] = 0u; }
    __syncthreads();
    constexpr int Q = kBucketChunkMin / kBT;
    IdOut keep_ids[SMALL ? Q : 1];
    // pass 1: the ids (Philox draws, codes, destination arrays) and the histogram over the buckets
#pragma unroll
    for (int q = 0; q < (SMALL ? Q : 1); ++q) {
        for (int t = SMALL ? q : 0; t < (SMALL ? q + 1 : trips); ++t) {
            const int64_t i = i_base + (int64_t)t * kBT + threadIdx.x;
            IdOut o{-1, -1, 0u};
            if (i < n_ce) {
                o = slot_ids(P, i, B, per_side);
                if (o.d >= 0 && (int64_t)o.d < TE.R) atomicAdd(&s_he[o.d >> TE.sh], 1u);
                else { o.d = -1; if (TE.flags) TE.flags[i] = 0; }   // (an id outside the table is dropped: it has no row to update)
                if (o.p >= 0 && (int64_t)o.p < TR.R) atomicAdd(&s_hr[o.p >> TR.sh], 1u); else o.p = -1;
            }
            if constexpr (SMALL) keep_ids[q] = o;
        }
    }
    __syncthreads();
    const uint32_t tot_e = block_scan_inplace(s_he, TE.nb, s_part);
    for (int b = threadIdx.x; b <= TE.nb; b += kBT) TE.bmat[(size_t)b * kLdm + c] = b < TE.nb ? s_he[b] : tot_e;
    if (has_rel) {
        const uint32_t tot_r = block_scan_inplace(s_hr, TR.nb, s_part);
        for (int b = threadIdx.x; b <= TR.nb; b += kBT) TR.bmat[(size_t)b * kLdm + c] = b < TR.nb ? s_hr[b] : tot_r;
    }
    __syncthreads();
    // pass 2: every contribution takes the next free place of its bucket's stretch (order inside a (chunk, bucket) stretch
    // arbitrary — the bucket kernel orders by slot).  Larger chunks re-read the ids their own thread wrote
#pragma unroll
    for (int q = 0; q < (SMALL ? Q : 1); ++q) {
        for (int t = SMALL ? q : 0; t < (SMALL ? q + 1 : trips); ++t) {
            const int64_t i = i_base + (int64_t)t * kBT + threadIdx.x;
            int32_t d = -1, p = -1;
            uint32_t kb = 0u;
            if constexpr (SMALL) { d = keep_ids[q].d; p = keep_ids[q].p; kb = keep_ids[q].kb; }
            else if (i < n_ce) {
                d = P.dest_ent[i]; if (!(d >= 0 && (int64_t)d < TE.R)) d = -1;
                if (i >= 2 * B) kb = (uint32_t)P.codes[i - 2 * B] & 0x80000000u;
                if (i < B) { p = P.dest_rel[i]; if (!(p >= 0 && (int64_t)p < TR.R)) p = -1; }
            }
            if (d >= 0) {
                const size_t at = (size_t)i_base + atomicAdd(&s_he[d >> TE.sh], 1u);
                TE.pd[at] = (uint32_t)d; TE.ps[at] = (uint32_t)i | kb;
            }
            if (p >= 0) {
                const size_t at = (size_t)i_base + atomicAdd(&s_hr[p >> TR.sh], 1u);
                TR.pd[at] = (uint32_t)p; TR.ps[at] = (uint32_t)i;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 2. one workgroup per bucket
// ---------------------------------------------------------------------------------------------------------------
struct BucketLds {
    uint32_t ends[kBucketRowsMax];          // per row: count -> exclusive start (scatter cursor) -> end
    uint32_t cs[kBucketChunksMax];          // per chunk: where the bucket's stretch starts inside the chunk
    uint32_t coff[kBucketChunksMax + 1];    // per chunk: pairs of the bucket in earlier chunks
    uint32_t in_slot[kBucketCap];           // the gathered pairs (global form: the list of long rows lives here)
    uint16_t in_dl[kBucketCap];
    uint32_t slot[kBucketCap];              // the bucket's slots in sorted order
    uint16_t dl[kBucketCap];
    uint16_t longs[1024];
    uint32_t misc[8 + 4 * (kBT / 64)];      // 1..3: list bases | 4: long rows | 8..: wave partials
};

// pair x of the bucket (x < count): which chunk holds it (bisection over the chunks' offsets) and where
__device__ __forceinline__ size_t pair_address(const BucketLds& S, int nchunks, int chunk_log, uint32_t x) {
    int lo = 0, hi = nchunks;   // largest c with coff[c] <= x
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (S.coff[mid] <= x) lo = mid; else hi = mid; }
    return ((size_t)lo << chunk_log) + S.cs[lo] + (x - S.coff[lo]);
}

template <bool LDS>
__device__ __forceinline__ void bucket_sort_body(const BucketTable& T, BucketLds& S, int b, int nchunks, int chunk_log, uint32_t g0,
                                                 uint32_t count, int64_t B) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int RB = 1 << T.sh;
    const uint32_t r0 = (uint32_t)b << T.sh;
    uint16_t* longs = LDS ? S.longs : reinterpret_cast<uint16_t*>(S.in_slot);
    // phase 1: gather the bucket's pairs (once: they stay in LDS) + row histogram
    // (four pairs per thread and trip: all eight loads of a trip in flight together)
    for (uint32_t x0 = threadIdx.x; x0 < count; x0 += 4u * kBT) {
        uint32_t d[4], sl[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t x = x0 + q * kBT;
            const size_t at = pair_address(S, nchunks, chunk_log, min(x, count - 1u));
            d[q] = T.pd[at];
            if constexpr (LDS) sl[q] = T.ps[at];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t x = x0 + q * kBT;
            if (x >= count) continue;
            const uint32_t dl = d[q] - r0;
            if constexpr (LDS) { S.in_dl[x] = (uint16_t)dl; S.in_slot[x] = sl[q]; }
            atomicAdd(&S.ends[dl], 1u);
        }
    }
    __syncthreads();
    // phase 2: scan over the bucket's rows; the bucket takes its stretch of each descriptor list with one atomic (issued
    // here, needed in phase 6)
    const int rpt = RB > kBT ? RB / kBT : 1;
    const int row0 = threadIdx.x * rpt;
    constexpr int RPT = kBucketRowsMax / kBT;
    uint32_t cj[RPT];
    uint32_t loc[4] = {0u, 0u, 0u, 0u};   // contributions | segments of 2..kDefer rows | singletons | block tasks
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        cj[j] = 0u;
        if (j < rpt && row0 + j < RB) cj[j] = S.ends[row0 + j];
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
        if (lane == 63) S.misc[8 + wv * 4 + q] = v;
    }
    __syncthreads();
    uint32_t wpre[4], tot[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        wpre[q] = 0u; tot[q] = 0u;
#pragma unroll
        for (int w = 0; w < kBT / 64; ++w) { const uint32_t v = S.misc[8 + w * 4 + q]; if (w < wv) wpre[q] += v; tot[q] += v; }
    }
    if (threadIdx.x == 0) {
        // GC_MULTI and GC_SINGLE are neighbouring words: ONE 64-bit atomic takes the bucket's stretch of both lists (every
        // bucket of a batch hits these counters at the same moment: three atomics per bucket on one line were a third of the kernel)
        static_assert(GC_MULTI == 0 && GC_SINGLE == 1, "the segment and singleton list lengths share a 64-bit word");
        const unsigned long long add = (unsigned long long)tot[1] | ((unsigned long long)tot[2] << 32);
        const unsigned long long old = add ? atomicAdd(reinterpret_cast<unsigned long long*>(T.counters + GC_MULTI), add) : 0ull;
        S.misc[1] = (uint32_t)old; S.misc[2] = (uint32_t)(old >> 32);
        S.misc[3] = tot[3] ? atomicAdd(T.counters + GC_TASKS, tot[3]) : 0u;
    }
    uint32_t run0 = wpre[0] + inc[0] - loc[0];
    const uint32_t my_start = run0;
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        if (!(j < rpt && row0 + j < RB)) continue;
        S.ends[row0 + j] = run0;   // the scatter cursor
        if (T.off && (int64_t)r0 + row0 + j < T.R) T.off[r0 + row0 + j] = g0 + run0;
        if (cj[j] > (uint32_t)kDeferSegment) longs[atomicAdd(&S.misc[4], 1u)] = (uint16_t)(row0 + j);
        run0 += cj[j];
    }
    __syncthreads();
    // the staging array of the bucket's slots: LDS, or (a bucket beyond the LDS capacity) the not yet written srcrow stretch
    uint32_t* gstage = T.srcrow + g0;
    auto ld = [&](uint32_t i) -> uint32_t { if constexpr (LDS) return S.slot[i]; else return gstage[i]; };
    auto stg = [&](uint32_t i, uint32_t v) { if constexpr (LDS) S.slot[i] = v; else gstage[i] = v; };
    // phase 3: scatter — a pair takes the next free position of its row (order inside a row arbitrary until phase 4)
    for (uint32_t x = threadIdx.x; x < count; x += kBT) {
        uint32_t dl, slot;
        if constexpr (LDS) { dl = S.in_dl[x]; slot = S.in_slot[x]; }
        else { const size_t at = pair_address(S, nchunks, chunk_log, x); dl = T.pd[at] - r0; slot = T.ps[at]; }
        const uint32_t p = atomicAdd(&S.ends[dl], 1u);
        stg(p, slot);
        if constexpr (LDS) S.dl[p] = (uint16_t)dl; else T.keys[g0 + p] = r0 + dl;
    }
    __syncthreads();   // (ends[row] = END of the row now; global form: the stores above are visible to the workgroup)
    // phase 4: ascending slot order inside every row = the stable order
    if constexpr (LDS) {
        // every slot counts the slots of its row below it and takes that place in a second array (the gathered pairs' — they are
        // consumed): item-parallel, so a row of 30 costs its thread 30 LDS reads, not 200 dependent exchanges (the relation
        // table: 16 contributions per row on average)
        for (uint32_t p = threadIdx.x; p < count; p += kBT) {
            const uint32_t dl = S.dl[p], mine = S.slot[p];
            const uint32_t start = dl ? S.ends[dl - 1] : 0u, n = S.ends[dl] - start;
            uint32_t rk = 0u;
            for (uint32_t u = 0; u < n; ++u) rk += (S.slot[start + u] & kSlot) < (mine & kSlot) ? 1u : 0u;
            S.in_slot[start + rk] = mine;
        }
    } else {
        {
            uint32_t start = my_start;
    #pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const uint32_t n = cj[j];
                if (n >= 2u && n <= (uint32_t)kDeferSegment) {
                    for (uint32_t i = 1; i < n; ++i) {
                        const uint32_t key = ld(start + i);
                        uint32_t k = i;
                        while (k > 0u) {
                            const uint32_t prev = ld(start + k - 1);
                            if ((prev & kSlot) <= (key & kSlot)) break;
                            stg(start + k, prev);
                            --k;
                        }
                        stg(start + k, key);
                    }
                }
                start += n;
            }
        }
        // longer rows: every slot counts the slots of its row below it (all threads read the same word: an LDS broadcast); four
        // slots per thread and pass, and the ranked copies go straight to their final place in `vals` — phase 5 takes the slots of
        // such a row from there — so nothing is held across passes
        const uint32_t n_long = S.misc[4];
        for (uint32_t li = 0; li < n_long; ++li) {
            const uint32_t row = longs[li];
            const uint32_t start = row ? S.ends[row - 1] : 0u, n = S.ends[row] - start;
            for (uint32_t i0 = 0; i0 < n; i0 += 4u * kBT) {
                uint32_t mine[4], rk[4];
    #pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t i = i0 + q * kBT + threadIdx.x;
                    rk[q] = 0u;
                    mine[q] = i < n ? ld(start + i) : 0u;
                }
                for (uint32_t u = 0; u < n; ++u) {
                    const uint32_t v = ld(start + u);
    #pragma unroll
                    for (int q = 0; q < 4; ++q) rk[q] += (v & kSlot) < (mine[q] & kSlot) ? 1u : 0u;
                }
    #pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (i0 + q * kBT + threadIdx.x < n) T.vals[g0 + start + rk[q]] = mine[q];
            }
        }
    }
    __syncthreads();
    // phase 5: the grouping's outputs, one coalesced pass over the bucket's sorted positions
    const uint32_t fB = (uint32_t)B;
    for (uint32_t p0 = threadIdx.x; p0 < count; p0 += 4u * kBT) {
        uint32_t slot[4], len[4], dlq[4], kb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t p = min(p0 + q * kBT, count - 1u);
            if constexpr (LDS) slot[q] = S.in_slot[p]; else slot[q] = ld(p);
            if constexpr (LDS) dlq[q] = S.dl[p]; else dlq[q] = T.keys[g0 + p] - r0;
            len[q] = S.ends[dlq[q]] - (dlq[q] ? S.ends[dlq[q] - 1] : 0u);
            if (!LDS && len[q] > (uint32_t)kDeferSegment) slot[q] = T.vals[g0 + p];
            kb[q] = slot[q] >> 31; slot[q] &= kSlot;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t p = p0 + q * kBT;
            if (p >= count) continue;
            const uint32_t at = g0 + p;
            T.vals[at] = slot[q];
            if constexpr (LDS) T.keys[at] = r0 + dlq[q];
            if (T.flags) T.flags[slot[q]] = len[q] == 1u ? 1 : 0;
            if (T.fac_codes) {
                if (slot[q] < 2u * fB) { T.srcrow[at] = slot[q]; T.coef[at] = 1.f; }   // subject / object rows are stored in full
                else {
                    const uint32_t i = slot[q] - 2u * fB;
                    T.srcrow[at] = (kb[q] ? 2u : 3u) * fB + i % fB;
                    T.pos_of_slot[i] = at;
                }
            }
        }
    }
    // phase 6: the segment descriptors (inside the bucket's stretch of a list the rows stay ascending)
    {
        uint32_t run[4];
        run[0] = my_start;
#pragma unroll
        for (int q = 1; q < 4; ++q) run[q] = S.misc[q] + wpre[q] + inc[q] - loc[q];
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const uint32_t c = cj[j], start = run[0];
            if (c == 0u) continue;
            const uint32_t r = r0 + row0 + j;
            if (c == 1u) T.single[run[2]++] = g0 + start;
            else if (c <= (uint32_t)kDeferSegment) T.multi[run[1]++] = Seg{g0 + start, c, r};
            else {
                const uint32_t nbk = (c + kLongSegment - 1) / kLongSegment;
                const bool room = run[3] + nbk <= T.task_cap;
                for (uint32_t k = 0; k < nbk && run[3] + k < T.task_cap; ++k) T.tasks[run[3] + k] = LongTask{g0 + start, k, room ? c : 0u};
                run[3] += nbk;
            }
            run[0] += c;
        }
    }
}

__global__ __launch_bounds__(kBT) void bucket_sort_kernel(const BucketLaunch L) {
    __shared__ BucketLds S;
    __shared__ uint32_t s_part[kBT / 64];
    const int ti = blockIdx.x < (unsigned)L.t[0].nb ? 0 : 1;
    const BucketTable& T = L.t[ti];
    const int b = (int)blockIdx.x - (ti ? L.t[0].nb : 0);
    const int64_t n = ti ? L.B : L.n_ce;
    const int nchunks = (int)((n + ((int64_t)1 << L.chunk_log) - 1) >> L.chunk_log);
    const int RB = 1 << T.sh;
    for (int r = threadIdx.x; r < RB; r += kBT) S.ends[r] = 0u;
    if (threadIdx.x == 0) S.misc[4] = 0u;
    // where the bucket's pairs lie: two coalesced rows of the offset matrix
    uint32_t g = 0u;
    for (int c = threadIdx.x; c < kBucketChunksMax; c += kBT) {
        uint32_t s = 0u, e = 0u;
        if (c < nchunks) { s = T.bmat[(size_t)b * kLdm + c]; e = T.bmat[(size_t)(b + 1) * kLdm + c]; }
        S.cs[c] = s; S.coff[c] = e - s;
        g += s;
    }
    // housekeeping of the apply that follows: per-segment block counters, the window path's task list
    const int64_t n_arr = n / kLongSegment + 1, per = (n_arr + T.nb - 1) / T.nb;
    for (int64_t t = (int64_t)b * per + threadIdx.x; t < min(n_arr, ((int64_t)b + 1) * per); t += kBT) T.arrive[t] = 0;
    if (b == 0 && threadIdx.x < 2) T.counters[GC_LONG_COUNT + threadIdx.x] = 0u;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) g += __shfl_xor(g, o, 64);
    if ((threadIdx.x & 63) == 0) S.misc[8 + (threadIdx.x >> 6)] = g;
    __syncthreads();
    uint32_t g0 = 0u;
#pragma unroll
    for (int w = 0; w < kBT / 64; ++w) g0 += S.misc[8 + w];
    __syncthreads();
    const uint32_t count = block_scan_inplace(S.coff, kBucketChunksMax, s_part);   // (everything in lower buckets, over all chunks: g0)
    if (b == T.nb - 1 && threadIdx.x == 0) { T.counters[GC_VALID] = g0 + count; if (T.off) T.off[T.R] = g0 + count; }
    if (count == 0u) {   // (an empty bucket still owns its stretch of the offset array)
        if (T.off) for (int64_t r = ((int64_t)b << T.sh) + threadIdx.x; r < min(T.R, ((int64_t)b + 1) << T.sh); r += kBT) T.off[r] = g0;
        return;
    }
    if (count <= L.cap_lds) bucket_sort_body<true>(T, S, b, nchunks, L.chunk_log, g0, count, L.B);
    else bucket_sort_body<false>(T, S, b, nchunks, L.chunk_log, g0, count, L.B);
}

static void fill_bucket_table(BucketTable& T, const GroupWs& w, const BucketGeo& g, int64_t R, uint8_t* flags, const int32_t* fac_codes) {
    T = BucketTable{};
    T.R = R; T.sh = g.sh; T.nb = g.nb;
    T.pd = w.tmpv; T.ps = w.tmpv2; T.bmat = w.bmat;
    T.keys = w.keys; T.vals = w.vals; T.srcrow = w.srcrow; T.pos_of_slot = w.pos_of_slot; T.coef = w.coef;
    T.multi = w.multi; T.single = w.single; T.tasks = w.tasks; T.task_cap = w.task_cap;
    T.arrive = w.arrive; T.counters = w.counters; T.flags = flags; T.fac_codes = fac_codes;
    T.off = R <= kDenseHereMaxRows ? w.off : nullptr;   // (emg_apply.hip::dense_in_segments: the same bound)
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
    L.B = a->B; L.n_ce = S.n_ce; L.cap_lds = cap_env; L.chunk_log = ge.chunk_log;   // (both tables in the entity table's chunks)
    const PrepParams P = S.prep;
    const unsigned nchunks = (unsigned)cdiv(S.n_ce, (int64_t)1 << ge.chunk_log);
    const size_t hist_lds = (size_t)(ge.nb + gr.nb + 2) * sizeof(uint32_t);   // (<= 2 x 4097 words: below the 64 KB that need no opt-in)
    if (ge.chunk_log == 10) hipLaunchKernelGGL(bucket_ids_kernel<true>, dim3(nchunks), dim3(kBT), hist_lds, st, P, L);
    else hipLaunchKernelGGL(bucket_ids_kernel<false>, dim3(nchunks), dim3(kBT), hist_lds, st, P, L);
    EMG_LAUNCH_CHECK();
    hipLaunchKernelGGL(bucket_sort_kernel, dim3((unsigned)(ge.nb + gr.nb)), dim3(kBT), 0, st, L);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

}  // namespace emg
