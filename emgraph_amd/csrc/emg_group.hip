// emg_group.hip — everything about a training batch that does not depend on the tables: Philox corruption codes,
// the destination row of every gradient contribution, and their GROUPING by destination (what the reference leaves
// to tf.IndexedSlices + the Keras optimizers' sparse apply, training/sgd.py:97 ... adam.py:45; loop
// EmbeddingModel.py:1388-1440).
//
// Grouping contract (both backends): keys[t] ascending destination ids, vals[t] the contribution slot at sorted
// position t, equal destinations in ascending slot order — so the float sum order of a destination's rows is fixed
// (bit-reproducible refit, tests/emgraph/models/test_models.py:338-367) — plus a per-slot SINGLETON flag.
//
// COUNTING backend (the training path).  Destinations are table rows, so no comparison sort is needed:
//   1. histogram    cnt[dest]++            int32 atomics, fused into prepare_ids_kernel (the ids are in registers there)
//   2. scan         one kernel over the R table rows (decoupled look-back between 4096-row tiles): exclusive offsets
//                   off[r], the scatter cursor, and the SEGMENT DESCRIPTORS the apply kernel works from — list of
//                   destinations with 2..32 contributions, list of singletons, 64-row block tasks of longer segments
//   3. scatter      position = cursor[dest]++   (order inside a segment arbitrary)
//   4. order        every contribution takes its rank among its segment's slots (segments are short; a hub row's
//                   thousands are ranked by as many threads): THE stable order; factored source rows resolved here
// Four launches for both tables of a step (rocPRIM's device radix sort took five for one table, the largest line of
// profiles/r2_i_kernel_stats.md), no per-step memset, sizes read from a device record when the step is a graph node.
// SORT backend: rocPRIM device radix sort for wide keys (see emg_group.hpp).
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include <rocprim/rocprim.hpp>

#include "emg_group.hpp"

namespace emg {

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

bool group_backend_counting(int64_t N, int64_t R) {
    static const int forced = [] {
        const char* e = getenv("EMG_GROUPING");   // A/B aid: "sort" = the radix-sort backend + window apply everywhere
        if (!e) return 0;
        return strcmp(e, "sort") == 0 ? 1 : (strcmp(e, "count") == 0 ? 2 : 0);
    }();
    if (forced == 1) return false;
    if (forced == 2) return R < ((int64_t)1 << 31);
    return R <= 16 * N + ((int64_t)1 << 20);
}

// force Onesweep (histogram + scan + one pass per 8-bit digit) above 4096 items: the default picks a
// block sort + ~13 merge launches below 1M items, which is launch-bound at our sizes
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                              rocprim::default_config, 4096>;

// the size query walks rocPRIM's host-side config selection (device lookup included): remember the last few answers
static int sort_temp_bytes(int64_t n, size_t* bytes) {
    *bytes = 0;
    if (n <= 0) return EMG_OK;
    static std::mutex mu;
    static int64_t cached_n[8] = {0};
    static size_t cached_b[8] = {0};
    static int next = 0;
    {
        std::lock_guard<std::mutex> g(mu);
        for (int i = 0; i < 8; ++i)
            if (cached_n[i] == n) { *bytes = cached_b[i]; return EMG_OK; }
    }
    EMG_HIP(rocprim::radix_sort_pairs<SortConfig>(nullptr, *bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                                  (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n, 0, 32,
                                                  (hipStream_t)0, false));
    std::lock_guard<std::mutex> g(mu);
    cached_n[next] = n; cached_b[next] = *bytes; next = (next + 1) & 7;
    return EMG_OK;
}

// [keys | vals | tmpv | srcrow | pos_of_slot | coef | multi | single | tasks || arrive | counters | status | cnt || off]
// (sort backend: ... || arrive | counters || rocPRIM temp); then the partial rows of the long-segment reduction.
// The part between the bars is the control region: zero before the first grouping, left zero by every grouping.
static int layout_impl(char* ws, int64_t ws_bytes, int64_t N, int64_t R, int64_t ldp, GroupWs* o, size_t* total, bool check) {
    if (N < 0) N = 0;
    o->counting = group_backend_counting(N, R);
    o->kb = align256((size_t)N * 4);
    size_t at = 0;
    auto take = [&](size_t b) { const size_t r = at; at += align256(b); return r; };
    const size_t keys = take(o->kb), vals = take(o->kb), tmpv = take(o->kb), srcrow = take(o->kb), pos = take(o->kb), coef = take(o->kb);
    const size_t multi = take(sizeof(Seg) * ((size_t)N / 2 + 1));
    const size_t single = take(o->kb);
    o->task_cap = (uint32_t)(N / 8 + 2);
    const size_t tasks = take(sizeof(LongTask) * (size_t)o->task_cap);
    o->clean_offset = at;
    const size_t arrive = take(4 * ((size_t)N / kLongSegment + 2));
    const size_t counters = take(4 * GC_WORDS);
    size_t status = 0, cnt = 0, off = 0, stmp = 0;
    o->scan_blocks = 0; o->sort_tmp_bytes = 0;
    if (o->counting) {
        o->scan_blocks = (int)cdiv(R + 1, kScanTile);
        status = take(8 * (size_t)o->scan_blocks);
        cnt = take(4 * ((size_t)R + 1));
        o->clean_bytes = at - o->clean_offset;
        off = take(4 * ((size_t)R + 1));
    } else {
        o->clean_bytes = at - o->clean_offset;
        int rc = sort_temp_bytes(N, &o->sort_tmp_bytes);
        if (rc != EMG_OK) return rc;
        stmp = take(o->sort_tmp_bytes);
    }
    const size_t base = at;
    const size_t need = (ldp > 0 && N > kLongSegment) ? partial_rows(N) * (size_t)ldp * sizeof(float) : 0;
    if (total) *total = base + need + 256;
    if (!check) return EMG_OK;
    EMG_REQUIRE((int64_t)base <= ws_bytes, "grouping workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)base);
    o->keys = (uint32_t*)(ws + keys); o->vals = (uint32_t*)(ws + vals); o->tmpv = (uint32_t*)(ws + tmpv);
    o->srcrow = (uint32_t*)(ws + srcrow); o->pos_of_slot = (uint32_t*)(ws + pos); o->coef = (float*)(ws + coef);
    o->multi = (Seg*)(ws + multi); o->single = (uint32_t*)(ws + single); o->tasks = (LongTask*)(ws + tasks);
    o->arrive = (int32_t*)(ws + arrive); o->counters = (uint32_t*)(ws + counters);
    o->status = o->counting ? (unsigned long long*)(ws + status) : nullptr;
    o->cnt = o->counting ? (int32_t*)(ws + cnt) : nullptr;
    o->off = o->counting ? (uint32_t*)(ws + off) : nullptr;
    o->sort_tmp = o->counting ? nullptr : (void*)(ws + stmp);
    o->partial = (need > 0 && (int64_t)(base + need) <= ws_bytes) ? (float*)(ws + base) : nullptr;
    return EMG_OK;
}

int group_ws_layout(void* ws, int64_t ws_bytes, int64_t N, int64_t R, int64_t ldp, GroupWs* out) {
    return layout_impl((char*)ws, ws_bytes, N, R, ldp, out, nullptr, true);
}

int64_t group_ws_bytes(int64_t N, int64_t R, int64_t ldp) {
    GroupWs w;
    size_t total = 0;
    if (layout_impl(nullptr, 0, N, R, ldp, &w, &total, false) != EMG_OK) return -1;
    return (int64_t)total;
}

int factor_view(void* workspace, int64_t workspace_bytes, int64_t N, int64_t R, FactorView* out) {
    GroupWs w;
    int rc = group_ws_layout(workspace, workspace_bytes, N, R, 0, &w);
    if (rc != EMG_OK) return rc;
    out->pos_of_slot = w.pos_of_slot; out->coef = w.coef;
    return EMG_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// one table's grouping problem, as the kernels see it
// ---------------------------------------------------------------------------------------------------------------
struct TableGroup {
    const int32_t* dest; int64_t n_extra; int32_t per_B; int32_t pad0;   // n = n_extra + per_B * B contributions
    int64_t R;
    int32_t* cnt; uint32_t* off;
    uint32_t *keys, *vals, *tmpv, *srcrow, *pos_of_slot; float* coef;
    Seg* multi; uint32_t* single; LongTask* tasks; uint32_t task_cap; int32_t scan_blocks;
    int32_t* arrive; uint32_t* counters; unsigned long long* status;
    uint8_t* flags; const int32_t* fac_codes;   // optional: per-slot singleton flags; factored contributions (codes of the batch)
};
struct GroupLaunch {
    TableGroup t[2]; int32_t n_tables; int32_t pad0;
    int64_t B; const StepCtl* ctl;
    unsigned split_n, split_scan;   // workgroups of table 0 in the per-contribution / the scan launches
};

__device__ __forceinline__ int64_t table_n(const GroupLaunch& G, int ti) {
    const int64_t B = G.ctl ? G.ctl->B : G.B;
    return G.t[ti].n_extra + (int64_t)G.t[ti].per_B * B;
}

// start of a grouping: list counters, scan ticket and tile status words back to zero (thread i of the launch)
__device__ __forceinline__ void group_reset(const TableGroup& T, int64_t i) {
    if (i < 8) T.counters[i] = 0u;
    if (T.status && i < T.scan_blocks) T.status[i] = 0ull;
}

__device__ __forceinline__ void hist_add(const TableGroup& T, int32_t d) {
    if (d >= 0 && (int64_t)d < T.R) atomicAdd(T.cnt + d, 1);   // (an id outside the table is dropped: it has no row to update)
}

// 1. histogram of an existing id array (emg_group_dest; emg_prepare_batch with caller-filled extra rows)
__global__ __launch_bounds__(256) void group_hist_kernel(const GroupLaunch G) {
    const int ti = blockIdx.x < G.split_n ? 0 : 1;
    const TableGroup& T = G.t[ti];
    const int64_t i = (int64_t)(blockIdx.x - (ti ? G.split_n : 0u)) * 256 + threadIdx.x;
    group_reset(T, i);
    if (i < table_n(G, ti)) hist_add(T, T.dest[i]);
}

// 2. scan over the table rows.  Tile = 4096 rows = 256 threads x 16; tiles are taken in ticket order, so every
// predecessor of a tile has started and publishes its aggregate without waiting for anybody (decoupled look-back,
// Merrill & Garland 2016): status word = value << 2 | (1: tile aggregate, 2: inclusive prefix).
__global__ __launch_bounds__(256) void group_scan_kernel(const GroupLaunch G) {
    const int ti = blockIdx.x < G.split_scan ? 0 : 1;
    const TableGroup& T = G.t[ti];
    __shared__ unsigned s_bid;
    __shared__ uint32_t s_wave[4][4];
    __shared__ uint32_t s_base[4];
    if (threadIdx.x == 0) s_bid = atomicAdd(T.counters + GC_SCAN_TICKET, 1u);
    __syncthreads();
    const unsigned bid = s_bid;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)bid * kScanTile + (int64_t)threadIdx.x * 16;
    const int64_t rows = T.R + 1;   // row R is the sentinel (count 0): off[R] = number of grouped contributions
    int c[16];
    if (r0 + 16 <= rows) {
        const int4* p = reinterpret_cast<const int4*>(T.cnt + r0);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int4 v = p[q]; c[4 * q] = v.x; c[4 * q + 1] = v.y; c[4 * q + 2] = v.z; c[4 * q + 3] = v.w; }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) c[j] = r0 + j < rows ? T.cnt[r0 + j] : 0;
    }
    uint32_t loc[4] = {0u, 0u, 0u, 0u};   // contributions | segments of 2..kDefer rows | singletons | block tasks
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t cj = (uint32_t)c[j];
        loc[0] += cj;
        loc[1] += (cj >= 2u && cj <= (uint32_t)kDeferSegment) ? 1u : 0u;
        loc[2] += cj == 1u ? 1u : 0u;
        loc[3] += cj > (uint32_t)kDeferSegment ? (cj + kLongSegment - 1) / kLongSegment : 0u;
    }
    uint32_t inc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t v = loc[q];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
        inc[q] = v;
        if (lane == 63) s_wave[wv][q] = v;
    }
    __syncthreads();
    uint32_t wpre[4], tot[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        wpre[q] = 0u; tot[q] = 0u;
#pragma unroll
        for (int w = 0; w < 4; ++w) { if (w < wv) wpre[q] += s_wave[w][q]; tot[q] += s_wave[w][q]; }
    }
    if (wv == 0) {
        if (lane == 0) {
            __hip_atomic_store(T.status + bid, ((unsigned long long)tot[0] << 2) | (bid == 0u ? 2ull : 1ull), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            // the lists need no global order (a segment's sum is self-contained): a tile takes its stretch of each with
            // one atomic; inside the stretch rows stay ascending
            s_base[1] = tot[1] ? atomicAdd(T.counters + GC_MULTI, tot[1]) : 0u;
            s_base[2] = tot[2] ? atomicAdd(T.counters + GC_SINGLE, tot[2]) : 0u;
            s_base[3] = tot[3] ? atomicAdd(T.counters + GC_TASKS, tot[3]) : 0u;
        }
        uint32_t excl = 0u;
        if (bid > 0u) {
            int64_t look = (int64_t)bid - 1;
            for (;;) {
                const int64_t j = look - lane;
                const unsigned long long sv = j >= 0 ? __hip_atomic_load(T.status + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 2ull;
                const unsigned flag = (unsigned)(sv & 3ull);
                const unsigned long long ready = __ballot(flag != 0u);
                const unsigned long long pref = __ballot(flag == 2u);
                const int p = pref ? __ffsll((long long)pref) - 1 : 63;
                const unsigned long long need = (2ull << p) - 1ull;   // lanes 0..p (p = 63: all)
                if ((ready & need) != need) { __builtin_amdgcn_s_sleep(1); continue; }
                uint32_t v = lane <= p ? (uint32_t)(sv >> 2) : 0u;
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
                excl += v;
                if (pref) break;
                look -= 64;
            }
            if (lane == 0)
                __hip_atomic_store(T.status + bid, ((unsigned long long)(excl + tot[0]) << 2) | 2ull, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) s_base[0] = excl;
    }
    __syncthreads();
    uint32_t run[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) run[q] = s_base[q] + wpre[q] + inc[q] - loc[q];
    if ((int64_t)bid == (int64_t)T.scan_blocks - 1 && threadIdx.x == 255) T.counters[GC_VALID] = run[0] + loc[0];
    uint32_t offs[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t cj = (uint32_t)c[j], start = run[0];
        const int64_t r = r0 + j;
        offs[j] = start;
        if (cj != 0u) {
            T.cnt[r] = (int32_t)start;   // scatter cursor (rows without contributions keep 0)
            if (cj == 1u) T.single[run[2]++] = start;
            else if (cj <= (uint32_t)kDeferSegment) T.multi[run[1]++] = Seg{start, cj, (uint32_t)r};
            else {
                const uint32_t nb = (cj + kLongSegment - 1) / kLongSegment;
                const bool room = run[3] + nb <= T.task_cap;   // (always: tasks <= n / 33 * ... < n / 8)
                for (uint32_t b = 0; b < nb && run[3] + b < T.task_cap; ++b) T.tasks[run[3] + b] = LongTask{start, b, room ? cj : 0u};
                run[3] += nb;
            }
            run[0] += cj;
        }
    }
    if (r0 + 16 <= rows) {
        uint4* p = reinterpret_cast<uint4*>(T.off + r0);
#pragma unroll
        for (int q = 0; q < 4; ++q) p[q] = make_uint4(offs[4 * q], offs[4 * q + 1], offs[4 * q + 2], offs[4 * q + 3]);
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) if (r0 + j < rows) T.off[r0 + j] = offs[j];
    }
}

// 3. scatter: a contribution takes the next free position of its destination's segment
__global__ __launch_bounds__(256) void group_scatter_kernel(const GroupLaunch G) {
    const int ti = blockIdx.x < G.split_n ? 0 : 1;
    const TableGroup& T = G.t[ti];
    const int64_t i = (int64_t)(blockIdx.x - (ti ? G.split_n : 0u)) * 256 + threadIdx.x;
    if (i >= table_n(G, ti)) return;
    const int32_t d = T.dest[i];
    const bool ok = d >= 0 && (int64_t)d < T.R;
    bool single = false;
    if (ok) {
        const uint32_t start = T.off[d];
        single = T.off[d + 1] - start == 1u;
        // (a destination hit once — most of them, for uniform negatives on a large table — needs no cursor)
        const uint32_t pos = single ? start : (uint32_t)atomicAdd(T.cnt + d, 1);
        T.tmpv[pos] = (uint32_t)i;
        T.keys[pos] = (uint32_t)d;
    }
    if (T.flags) T.flags[i] = single ? 1 : 0;
}

// 4. order: rank of a contribution among the slots of its segment = its place in the stable order.
// Factored contributions (see emg_backward_args.fac_ws_ent): srcrow[q] = the row of the 4B-row contribution buffer the slot
// at sorted position q points at, pos_of_slot[slot - 2B] = q for the negatives' slots (where the backward kernel puts
// their factor), coef[q] = 1 for the subject / object slots.
__global__ __launch_bounds__(256) void group_order_kernel(const GroupLaunch G) {
    const int ti = blockIdx.x < G.split_n ? 0 : 1;
    const TableGroup& T = G.t[ti];
    const int64_t t = (int64_t)(blockIdx.x - (ti ? G.split_n : 0u)) * 256 + threadIdx.x;
    const int64_t n = table_n(G, ti);
    if (t < 2) T.counters[GC_LONG_COUNT + t] = 0u;          // window-path task list (apply_rows_kernel) starts empty
    if (t <= n / kLongSegment) T.arrive[t] = 0;             // per-segment block counters of the long-segment reduction
    if (t >= (int64_t)T.off[T.R]) return;
    const uint32_t d = T.keys[t];
    const uint32_t start = T.off[d], len = T.off[d + 1] - start;
    const uint32_t mine = T.tmpv[t];
    uint32_t rank = 0u;
    for (uint32_t j = 0; j < len; ++j) rank += T.tmpv[start + j] < mine ? 1u : 0u;
    const uint32_t q = start + rank;
    T.vals[q] = mine;
    if ((uint32_t)t == start) T.cnt[d] = 0;                 // the cursor has done its work: the histogram is zero again
    if (T.fac_codes) {
        const uint32_t fac_B = (uint32_t)(G.ctl ? G.ctl->B : G.B);
        if (mine < 2u * fac_B) {
            T.srcrow[q] = mine;
            T.coef[q] = 1.f;   // subject / object rows are stored in full (the negatives' factors come from the backward kernel)
        } else {
            const uint32_t i = mine - 2u * fac_B;
            T.srcrow[q] = (T.fac_codes[i] < 0 ? 2u : 3u) * fac_B + i % fac_B;
            T.pos_of_slot[i] = q;
        }
    }
}

// SORT backend epilogue: flags[original index] = 1 iff its destination occurs exactly once; factored source rows
__global__ void mark_single_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, int64_t n,
                                   uint8_t* __restrict__ flags, uint32_t* __restrict__ counters,
                                   const int32_t* __restrict__ fac_codes, uint32_t fac_B, uint32_t* __restrict__ srcrow,
                                   uint32_t* __restrict__ pos_of_slot, float* __restrict__ coef,
                                   int32_t* __restrict__ arrive) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 2) counters[GC_LONG_COUNT + t] = 0u;   // task list of the apply that follows (saves a memset launch)
    if (t <= n / kLongSegment) arrive[t] = 0;        // and its per-segment block counters
    if (t >= n) return;
    const uint32_t key = keys[t];
    const uint32_t slot = vals[t];
    if (flags) {
        const bool head = (t == 0) || keys[t - 1] != key;
        const bool last = (t + 1 == n) || keys[t + 1] != key;
        flags[slot] = (head && last) ? 1 : 0;
    }
    if (fac_codes) {
        if (slot < 2u * fac_B) {
            srcrow[t] = slot;
            coef[t] = 1.f;
        } else {
            const uint32_t i = slot - 2u * fac_B;
            srcrow[t] = (fac_codes[i] < 0 ? 2u : 3u) * fac_B + i % fac_B;
            pos_of_slot[i] = (uint32_t)t;
        }
    }
}

// corruption codes (Philox / injected) + the destination ids they imply, for every corruption side, ONE launch;
// with the counting backend also the histogram of both tables
struct PrepParams {
    const int32_t* pos; int64_t B; int32_t eta; int32_t n_sides; int32_t sides[4];
    uint64_t n_choices; const int32_t* entities_list; uint64_t seed; uint64_t counter0;
    const int32_t* inj_mask; const int32_t* inj_repl;
    int32_t* codes; int32_t* dest_ent; int32_t* dest_rel;
    int64_t B_global; int64_t row_offset;  // draw index of (negative je, local row i) = je * B_global + row_offset + i
    const StepCtl* ctl;                    // graph node: batch = rows [ctl->start, +ctl->B) of `pos`, draws from ctl->draw_counter0
    int32_t hist;                          // 1: histogram + grouping reset of G's tables
};

__global__ __launch_bounds__(256) void prepare_ids_kernel(const PrepParams P, const GroupLaunch G) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t B = P.B;
    const int32_t* pos = P.pos;
    uint64_t counter0 = P.counter0, n_choices = P.n_choices;
    const int32_t* elist = P.entities_list;
    if (P.ctl) {
        B = P.ctl->B; pos += 3 * P.ctl->start; counter0 = P.ctl->draw_counter0;
        if (P.ctl->n_choices > 0) { n_choices = (uint64_t)P.ctl->n_choices; elist = P.ctl->entities_list; }
    }
    if (P.hist) { group_reset(G.t[0], j); group_reset(G.t[1], j); }
    const int64_t per_side = (int64_t)P.eta * B;
    if (j < B) {
        const int32_t s = pos[3 * j + 0], p = pos[3 * j + 1], o = pos[3 * j + 2];
        P.dest_ent[j] = s;
        P.dest_ent[B + j] = o;
        P.dest_rel[j] = p;
        if (P.hist) { hist_add(G.t[0], s); hist_add(G.t[0], o); hist_add(G.t[1], p); }
    }
    if (j >= per_side * P.n_sides) return;
    const int sd = (int)(j / per_side);
    int64_t jj = j - sd * per_side;  // the draw index restarts per side (one emg_corrupt_codes call each)
    if (!P.ctl && P.B_global != B) {  // this batch is rows [row_offset, row_offset + B) of a larger one: draw what IT would
        const int64_t je = jj / B;
        jj = je * P.B_global + P.row_offset + (jj - je * B);
    }
    const int side = P.sides[sd];
    uint32_t keep, idx;
    if (P.inj_repl) {
        idx = (uint32_t)P.inj_repl[j];
        keep = P.inj_mask ? (uint32_t)(P.inj_mask[j] != 0) : 0u;
    } else {
        corruption_draw(P.seed, counter0 + (uint64_t)sd, (uint64_t)jj, n_choices, &keep, &idx);
    }
    if (side == EMG_SIDE_O) keep = 1u;
    else if (side == EMG_SIDE_S) keep = 0u;
    const uint32_t repl = elist ? (uint32_t)elist[idx] : idx;
    P.codes[j] = (int32_t)((repl & 0x7fffffffu) | (keep << 31));
    P.dest_ent[2 * B + j] = (int32_t)(repl & 0x7fffffffu);
    if (P.hist) hist_add(G.t[0], (int32_t)(repl & 0x7fffffffu));
}

static void fill_table(TableGroup& T, const GroupWs& w, const int32_t* dest, int64_t n_extra, int per_B, int64_t R,
                       uint8_t* flags, const int32_t* fac_codes) {
    T = TableGroup{};
    T.dest = dest; T.n_extra = n_extra; T.per_B = per_B; T.R = R;
    T.cnt = w.cnt; T.off = w.off; T.keys = w.keys; T.vals = w.vals; T.tmpv = w.tmpv; T.srcrow = w.srcrow;
    T.pos_of_slot = w.pos_of_slot; T.coef = w.coef; T.multi = w.multi; T.single = w.single; T.tasks = w.tasks;
    T.task_cap = w.task_cap; T.scan_blocks = w.scan_blocks; T.arrive = w.arrive; T.counters = w.counters; T.status = w.status;
    T.flags = flags; T.fac_codes = fac_codes;
}

static int clean_ws(const GroupWs& w, void* ws, hipStream_t st) {
    EMG_HIP(hipMemsetAsync((char*)ws + w.clean_offset, 0, w.clean_bytes, st));
    return EMG_OK;
}

// scan -> scatter -> order of one or two tables whose histograms are complete (cap_n*: launch sizes; the kernels read
// the actual sizes from G.B / G.ctl)
static int counting_tail(GroupLaunch& G, int64_t cap_n0, int64_t cap_n1, hipStream_t st) {
    G.split_scan = (unsigned)G.t[0].scan_blocks;
    const unsigned scan_blocks = G.split_scan + (G.n_tables > 1 ? (unsigned)G.t[1].scan_blocks : 0u);
    hipLaunchKernelGGL(group_scan_kernel, dim3(scan_blocks), dim3(256), 0, st, G);
    EMG_LAUNCH_CHECK();
    // (+1: the order kernel also resets arrive[0 .. n / 64], and thread 0 / 1 the window path's task counters)
    G.split_n = (unsigned)cdiv(cap_n0 + 1, 256);
    const unsigned nb = G.split_n + (G.n_tables > 1 ? (unsigned)cdiv(cap_n1 + 1, 256) : 0u);
    hipLaunchKernelGGL(group_scatter_kernel, dim3(nb), dim3(256), 0, st, G);
    EMG_LAUNCH_CHECK();
    hipLaunchKernelGGL(group_order_kernel, dim3(nb), dim3(256), 0, st, G);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

// stable grouping of n destination ids into the workspace (SORT backend)
static int sort_group(const int32_t* dest, int64_t n, int64_t n_rows, const GroupWs& w, uint8_t* single_flags, hipStream_t st,
                      const int32_t* fac_codes, int64_t fac_B) {
    int end_bit = 1;
    while (end_bit < 32 && ((int64_t)1 << end_bit) < n_rows) ++end_bit;
    size_t tmp = w.sort_tmp_bytes;
    // values = original positions, generated on the fly (no iota array)
    EMG_HIP(rocprim::radix_sort_pairs<SortConfig>(w.sort_tmp, tmp, (const uint32_t*)dest, w.keys,
                                                  rocprim::counting_iterator<uint32_t>(0u), w.vals, (size_t)n, 0,
                                                  end_bit, st, false));
    hipLaunchKernelGGL(mark_single_kernel, dim3((unsigned)cdiv(n + 1, 256)), dim3(256), 0, st, w.keys, w.vals, n,
                       single_flags, w.counters, fac_codes, (uint32_t)fac_B, w.srcrow, w.pos_of_slot, w.coef, w.arrive);
    EMG_LAUNCH_CHECK();
    return EMG_OK;
}

}  // namespace emg

using namespace emg;

extern "C" int64_t emg_apply_workspace_bytes(int64_t n_contrib, int64_t n_rows) {
    if (n_contrib <= 0) return 256;
    return group_ws_bytes(n_contrib, n_rows > 0 ? n_rows : 1, 0);
}

extern "C" int64_t emg_apply_workspace_bytes_ex(int64_t n_contrib, int64_t n_rows, int32_t k_int) {
    if (n_contrib <= 0) return 256;
    return group_ws_bytes(n_contrib, n_rows > 0 ? n_rows : 1, k_int > 0 ? (k_int + 3) / 4 * 4 : 0);
}

extern "C" int emg_group_dest(const int32_t* dest, int64_t n, int64_t n_rows, void* workspace,
                              int64_t workspace_bytes, uint8_t* single_flags, void* stream) {
    EMG_REQUIRE(n >= 0 && n_rows > 0 && n_rows < ((int64_t)1 << 31), "emg_group_dest: bad sizes");
    if (n == 0) return EMG_OK;
    EMG_REQUIRE(dest && workspace, "emg_group_dest: null pointer");
    EMG_REQUIRE(n < ((int64_t)1 << 31), "emg_group_dest: too many contributions");
    hipStream_t st = (hipStream_t)stream;
    GroupWs w;
    int rc = group_ws_layout(workspace, workspace_bytes, n, n_rows, 0, &w);
    if (rc != EMG_OK) return rc;
    if (!w.counting) return sort_group(dest, n, n_rows, w, single_flags, st, nullptr, 0);
    rc = clean_ws(w, workspace, st);   // a caller-owned workspace of unknown content
    if (rc != EMG_OK) return rc;
    GroupLaunch G{};
    G.n_tables = 1; G.B = 0;
    fill_table(G.t[0], w, dest, n, 0, n_rows, single_flags, nullptr);
    G.t[1] = G.t[0];
    G.split_n = (unsigned)cdiv(n > w.scan_blocks ? n : w.scan_blocks, 256);
    hipLaunchKernelGGL(group_hist_kernel, dim3(G.split_n), dim3(256), 0, st, G);
    EMG_LAUNCH_CHECK();
    return counting_tail(G, n, 0, st);
}

// internal form: layout_B > 0 sizes the workspaces' layout for that many positives (a plan's capacity) and ctl, if given,
// is the device record the kernels read the batch from
extern "C" int emg_prepare_batch(const emg_prepare_args* a, void* stream) {
    EMG_REQUIRE(a, "emg_prepare_batch: null args");
    EMG_REQUIRE(a->B >= 0 && a->eta >= 1 && a->n_sides >= 1 && a->n_sides <= 4, "emg_prepare_batch: bad sizes");
    if (a->B == 0) return EMG_OK;
    EMG_REQUIRE(a->pos && a->codes && a->dest_ent && a->dest_rel && a->ws_ent && a->ws_rel, "emg_prepare_batch: null pointer");
    EMG_REQUIRE(a->inj_repl || a->n_choices > 0, "emg_prepare_batch: n_choices must be positive");
    EMG_REQUIRE(a->n_extra_ent >= 0 && a->n_extra_rel >= 0 && a->n_ent > 0 && a->n_rel > 0, "emg_prepare_batch: bad table sizes");
    EMG_REQUIRE(a->n_ent < ((int64_t)1 << 31) && a->n_rel < ((int64_t)1 << 31), "emg_prepare_batch: too many rows");
    hipStream_t st = (hipStream_t)stream;
    const StepCtl* ctl = (const StepCtl*)a->ctl;
    EMG_REQUIRE(!ctl || (a->layout_B >= a->B && !a->inj_repl && a->B_global == 0),
                "emg_prepare_batch: a device-side batch record needs layout_B >= B (its capacity) and excludes injected / sharded draws");
    PrepParams P{};
    P.pos = a->pos; P.B = a->B; P.eta = a->eta; P.n_sides = a->n_sides;
    for (int i = 0; i < a->n_sides; ++i) {
        EMG_REQUIRE(a->sides[i] >= EMG_SIDE_S && a->sides[i] <= EMG_SIDE_SO, "emg_prepare_batch: bad side %d", a->sides[i]);
        P.sides[i] = a->sides[i];
    }
    P.n_choices = (uint64_t)a->n_choices; P.entities_list = a->entities_list; P.seed = a->seed; P.counter0 = a->draw_counter0;
    P.inj_mask = a->inj_mask; P.inj_repl = a->inj_repl; P.codes = a->codes;
    EMG_REQUIRE(a->B_global == 0 || (a->row_offset >= 0 && a->row_offset + a->B <= a->B_global),
                "emg_prepare_batch: rows [row_offset, row_offset + B) must lie inside the global batch");
    P.B_global = a->B_global > 0 ? a->B_global : a->B;
    P.row_offset = a->B_global > 0 ? a->row_offset : 0;
    P.dest_ent = a->dest_ent + a->n_extra_ent; P.dest_rel = a->dest_rel + a->n_extra_rel;
    P.ctl = ctl;
    const int et = a->eta * a->n_sides;
    const int64_t Bl = a->layout_B > 0 ? a->layout_B : a->B;          // the size the workspaces are laid out (and launches sized) for
    const int64_t n_neg = Bl * (int64_t)et;
    const int64_t n_ce = a->n_extra_ent + (2 + (int64_t)et) * a->B, n_cr = a->n_extra_rel + a->B;
    const int64_t cap_ce = a->n_extra_ent + (2 + (int64_t)et) * Bl, cap_cr = a->n_extra_rel + Bl;
    EMG_REQUIRE(!a->factored || (a->n_extra_ent == 0 && cap_ce < ((int64_t)1 << 31)),
                "emg_prepare_batch: factored contributions exclude caller-filled extra entity rows");
    EMG_REQUIRE(cap_ce < ((int64_t)1 << 31), "emg_prepare_batch: too many contributions");
    GroupWs we, wr;
    int rc = group_ws_layout(a->ws_ent, a->ws_ent_bytes, cap_ce, a->n_ent, 0, &we);
    if (rc == EMG_OK) rc = group_ws_layout(a->ws_rel, a->ws_rel_bytes, cap_cr, a->n_rel, 0, &wr);
    if (rc != EMG_OK) return rc;
    GroupLaunch G{};
    G.n_tables = 2; G.B = a->B; G.ctl = ctl;
    fill_table(G.t[0], we, a->dest_ent, a->n_extra_ent, 2 + et, a->n_ent, a->single_flags, a->factored ? a->codes : nullptr);
    fill_table(G.t[1], wr, a->dest_rel, a->n_extra_rel, 1, a->n_rel, nullptr, nullptr);
    const bool both = we.counting && wr.counting;
    if (both && !a->ws_clean) {
        rc = clean_ws(we, a->ws_ent, st);
        if (rc == EMG_OK) rc = clean_ws(wr, a->ws_rel, st);
        if (rc != EMG_OK) return rc;
    }
    // the histogram rides in the id kernel unless caller-filled extra rows come first (the ids of those are in memory)
    const bool fused_hist = both && a->n_extra_ent == 0 && a->n_extra_rel == 0;
    P.hist = fused_hist ? 1 : 0;
    int64_t threads = n_neg > Bl ? n_neg : Bl;
    if (fused_hist) { const int64_t sb = we.scan_blocks > wr.scan_blocks ? we.scan_blocks : wr.scan_blocks; if (sb > threads) threads = sb; }
    hipLaunchKernelGGL(prepare_ids_kernel, dim3((unsigned)cdiv(threads, 256)), dim3(256), 0, st, P, G);
    EMG_LAUNCH_CHECK();
    if (both) {
        if (!fused_hist) {
            G.split_n = (unsigned)cdiv(cap_ce > we.scan_blocks ? cap_ce : we.scan_blocks, 256);
            const unsigned nb = G.split_n + (unsigned)cdiv(cap_cr > wr.scan_blocks ? cap_cr : wr.scan_blocks, 256);
            hipLaunchKernelGGL(group_hist_kernel, dim3(nb), dim3(256), 0, st, G);
            EMG_LAUNCH_CHECK();
        }
        return counting_tail(G, cap_ce, cap_cr, st);
    }
    EMG_REQUIRE(!ctl, "emg_prepare_batch: a device-side batch record needs the counting backend for both tables");
    // mixed / sort backends: table by table
    for (int ti = 0; ti < 2; ++ti) {
        const GroupWs& w = ti ? wr : we;
        const int32_t* dest = ti ? a->dest_rel : a->dest_ent;
        const int64_t n = ti ? n_cr : n_ce, R = ti ? a->n_rel : a->n_ent;
        uint8_t* flags = ti ? nullptr : a->single_flags;
        const int32_t* fc = (ti == 0 && a->factored) ? a->codes : nullptr;
        if (!w.counting) { rc = sort_group(dest, n, R, w, flags, st, fc, a->B); if (rc != EMG_OK) return rc; continue; }
        if (!a->ws_clean) { rc = clean_ws(w, ti ? a->ws_rel : a->ws_ent, st); if (rc != EMG_OK) return rc; }
        GroupLaunch G1{};
        G1.n_tables = 1; G1.B = a->B;
        G1.t[0] = G.t[ti]; G1.t[1] = G.t[ti];
        G1.split_n = (unsigned)cdiv(n > w.scan_blocks ? n : w.scan_blocks, 256);
        hipLaunchKernelGGL(group_hist_kernel, dim3(G1.split_n), dim3(256), 0, st, G1);
        EMG_LAUNCH_CHECK();
        rc = counting_tail(G1, n, 0, st);
        if (rc != EMG_OK) return rc;
    }
    return EMG_OK;
}
