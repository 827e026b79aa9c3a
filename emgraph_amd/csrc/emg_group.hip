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

#include "emg_group_kernels.hpp"

namespace emg {

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

bool group_backend_counting(int64_t N, int64_t R) {
    // A/B aid, read per call (tests switch it inside one process): "sort" = the radix-sort backend + window apply everywhere,
    // "count" = the counting grouping whatever the table's size; anything else (unset, "bucket"): by size
    const char* e = getenv("EMG_GROUPING");
    if (e && strcmp(e, "sort") == 0) return false;
    if (e && strcmp(e, "count") == 0) return R < ((int64_t)1 << 31);
    return R <= 16 * N + ((int64_t)1 << 20);
}

BucketGeo bucket_geometry(int64_t N, int64_t R) {
    BucketGeo g{0, 0, 0, false};
    if (N <= 0 || R <= 0) return g;
    // ~1024 contributions per bucket on average: rows per bucket = 1024 R / N, a power of two in [1, kBucketRowsMax]
    int64_t want = (1024 * R) / N;
    if (want < 1) want = 1;
    int sh = 0;
    while (sh < 11 && ((int64_t)2 << sh) <= want) ++sh;
    while (sh < 11 && cdiv(R, (int64_t)1 << sh) > kBucketMaxNB) ++sh;
    const int64_t nb = cdiv(R, (int64_t)1 << sh);
    int cl = 10;
    while (cl < 30 && cdiv(N, (int64_t)1 << cl) > kBucketChunksMax) ++cl;
    if (nb > kBucketMaxNB || cl > 16) return g;   // (a workgroup of the id kernel walks at most 65536 slots)
    g.sh = sh; g.nb = (int)nb; g.chunk_log = cl; g.ok = true;
    return g;
}

bool group_backend_bucket(int64_t n_ent) {
    const char* e = getenv("EMG_GROUPING");   // (read per call: tests switch it inside one process)
    if (e && e[0]) return strcmp(e, "bucket") == 0 && n_ent > kDenseHereMaxRows;   // count / sort: never; forced: wherever it is valid
    return n_ent >= kBucketMinRows;
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
    size_t status = 0, cnt = 0, off = 0, stmp = 0, tmpv2 = 0, bmat = 0;
    const BucketGeo geo = bucket_geometry(N, R);
    o->scan_blocks = 0; o->sort_tmp_bytes = 0;
    if (o->counting) {
        o->scan_blocks = (int)cdiv(R + 1, kScanTile);
        status = take(8 * (size_t)o->scan_blocks);
        cnt = take(4 * ((size_t)R + 1));
        o->clean_bytes = at - o->clean_offset;
        off = take(4 * ((size_t)R + 1));
        if (geo.ok) { tmpv2 = take(o->kb); bmat = take(4 * (size_t)kBucketChunksMax * ((size_t)geo.nb + 1)); }
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
    o->tmpv2 = (o->counting && geo.ok) ? (uint32_t*)(ws + tmpv2) : nullptr;
    o->bmat = (o->counting && geo.ok) ? (uint32_t*)(ws + bmat) : nullptr;
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

__global__ __launch_bounds__(256) void group_hist_kernel(const GroupLaunch G) { group_hist_body(G, blockIdx.x); }
__global__ __launch_bounds__(256) void group_scan_kernel(const GroupLaunch G) { group_scan_body(G, blockIdx.x); }
__global__ __launch_bounds__(256) void group_scatter_kernel(const GroupLaunch G) { group_scatter_body(G, blockIdx.x); }
__global__ __launch_bounds__(256) void group_order_kernel(const GroupLaunch G) { group_order_body(G, blockIdx.x); }
__global__ __launch_bounds__(256) void prepare_ids_kernel(const PrepParams P, const GroupLaunch G) { prepare_ids_body(P, G, blockIdx.x); }

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
    G.split_n = (unsigned)cdiv(cap_n0 + 1, kPrepBlock2);
    const unsigned nb = G.split_n + (G.n_tables > 1 ? (unsigned)cdiv(cap_n1 + 1, kPrepBlock2) : 0u);
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
    G.split_n = (unsigned)cdiv(n > w.scan_blocks ? n : w.scan_blocks, kPrepBlock);
    hipLaunchKernelGGL(group_hist_kernel, dim3(G.split_n), dim3(256), 0, st, G);
    EMG_LAUNCH_CHECK();
    return counting_tail(G, n, 0, st);
}

// emg_group_dest with an explicit ORDER KEY: the contributions of a destination are ordered by ascending order_key[i] (distinct
// within a destination) instead of by input index — the owner of a table range in the batch-sharded step receives gradient rows
// from every rank and must add them in the order of their slots in the GLOBAL batch (what one GPU does).  The apply then finds
// row i of the receive buffer through the factored arrays (srcrow = i, coef = 1): emg_apply_grouped_factored.
extern "C" int emg_group_dest_keyed(const int32_t* dest, const uint32_t* order_key, int64_t n, int64_t n_rows, void* workspace,
                                    int64_t workspace_bytes, void* stream) {
    EMG_REQUIRE(n >= 0 && n_rows > 0 && n_rows < ((int64_t)1 << 31), "emg_group_dest_keyed: bad sizes");
    if (n == 0) return EMG_OK;
    EMG_REQUIRE(dest && order_key && workspace, "emg_group_dest_keyed: null pointer");
    EMG_REQUIRE(n < ((int64_t)1 << 31), "emg_group_dest_keyed: too many contributions");
    hipStream_t st = (hipStream_t)stream;
    GroupWs w;
    int rc = group_ws_layout(workspace, workspace_bytes, n, n_rows, 0, &w);
    if (rc != EMG_OK) return rc;
    if (!w.counting) return fail(EMG_ENOSUP, "emg_group_dest_keyed: needs the counting grouping (n_rows <= 16 n + 2^20)");
    rc = clean_ws(w, workspace, st);
    if (rc != EMG_OK) return rc;
    GroupLaunch G{};
    G.n_tables = 1; G.B = 0;
    fill_table(G.t[0], w, dest, n, 0, n_rows, nullptr, nullptr);
    G.t[0].order_key = order_key;
    G.t[1] = G.t[0];
    G.split_n = (unsigned)cdiv(n > w.scan_blocks ? n : w.scan_blocks, kPrepBlock);
    hipLaunchKernelGGL(group_hist_kernel, dim3(G.split_n), dim3(256), 0, st, G);
    EMG_LAUNCH_CHECK();
    return counting_tail(G, n, 0, st);
}

// internal form: layout_B > 0 sizes the workspaces' layout for that many positives (a plan's capacity) and ctl, if given,
// is the device record the kernels read the batch from
int launch_riders_alone(const Riders& R, hipStream_t st) {
    for (int i = 0; i < 2; ++i) {
        const Rider& r = R.r[i];
        if (r.kind == RIDE_NONE || r.blocks == 0) continue;
        const dim3 grid(r.blocks), block(256);
        if (r.kind == RIDE_IDS) hipLaunchKernelGGL(prepare_ids_kernel, grid, block, 0, st, R.prep, r.G);
        else if (r.kind == RIDE_SCAN) hipLaunchKernelGGL(group_scan_kernel, grid, block, 0, st, r.G);
        else if (r.kind == RIDE_SCATTER) hipLaunchKernelGGL(group_scatter_kernel, grid, block, 0, st, r.G);
        else if (r.kind == RIDE_ORDER) hipLaunchKernelGGL(group_order_kernel, grid, block, 0, st, r.G);
        EMG_LAUNCH_CHECK();
    }
    return EMG_OK;
}

// validation + layout of emg_prepare_batch without a launch: the four stages of the counting grouping as launchable
// descriptions (the plan attaches them to the step's big launches as riders)
int prepare_stages(const emg_prepare_args* a, PrepStages* o) {
    EMG_REQUIRE(a && o, "emg_prepare_batch: null args");
    EMG_REQUIRE(a->B > 0 && a->eta >= 1 && a->n_sides >= 1 && a->n_sides <= 4, "emg_prepare_batch: bad sizes");
    EMG_REQUIRE(a->pos && a->codes && a->dest_ent && a->dest_rel && a->ws_ent && a->ws_rel, "emg_prepare_batch: null pointer");
    EMG_REQUIRE(a->inj_repl || a->n_choices > 0, "emg_prepare_batch: n_choices must be positive");
    EMG_REQUIRE(a->n_extra_ent >= 0 && a->n_extra_rel >= 0 && a->n_ent > 0 && a->n_rel > 0, "emg_prepare_batch: bad table sizes");
    EMG_REQUIRE(a->n_ent < ((int64_t)1 << 31) && a->n_rel < ((int64_t)1 << 31), "emg_prepare_batch: too many rows");
    const StepCtl* ctl = (const StepCtl*)a->ctl;
    EMG_REQUIRE(!ctl || (a->layout_B >= a->B && !a->inj_repl && a->B_global == 0),
                "emg_prepare_batch: a device-side batch record needs layout_B >= B (its capacity) and excludes injected / sharded draws");
    *o = PrepStages{};
    PrepParams& P = o->prep;
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
    o->n_ce = a->n_extra_ent + (2 + (int64_t)et) * a->B; o->n_cr = a->n_extra_rel + a->B;
    o->cap_ce = a->n_extra_ent + (2 + (int64_t)et) * Bl; o->cap_cr = a->n_extra_rel + Bl;
    EMG_REQUIRE(!a->factored || (a->n_extra_ent == 0 && o->cap_ce < ((int64_t)1 << 31)),
                "emg_prepare_batch: factored contributions exclude caller-filled extra entity rows");
    EMG_REQUIRE(o->cap_ce < ((int64_t)1 << 31), "emg_prepare_batch: too many contributions");
    int rc = group_ws_layout(a->ws_ent, a->ws_ent_bytes, o->cap_ce, a->n_ent, 0, &o->we);
    if (rc == EMG_OK) rc = group_ws_layout(a->ws_rel, a->ws_rel_bytes, o->cap_cr, a->n_rel, 0, &o->wr);
    if (rc != EMG_OK) return rc;
    GroupLaunch& G = o->G;
    G.n_tables = 2; G.B = a->B; G.ctl = ctl;
    fill_table(G.t[0], o->we, a->dest_ent, a->n_extra_ent, 2 + et, a->n_ent, a->single_flags, a->factored ? a->codes : nullptr);
    fill_table(G.t[1], o->wr, a->dest_rel, a->n_extra_rel, 1, a->n_rel, nullptr, nullptr);
    o->both = o->we.counting && o->wr.counting;
    // the histogram rides in the id kernel unless caller-filled extra rows come first (the ids of those are in memory)
    o->fused_hist = o->both && a->n_extra_ent == 0 && a->n_extra_rel == 0;
    P.hist = o->fused_hist ? 1 : 0;
    int64_t threads = n_neg > Bl ? n_neg : Bl;
    if (o->fused_hist) { const int64_t sb = o->we.scan_blocks > o->wr.scan_blocks ? o->we.scan_blocks : o->wr.scan_blocks; if (sb > threads) threads = sb; }
    o->nb_ids = (unsigned)cdiv(threads, kPrepBlock);
    if (o->both) {   // launch geometry of scan / scatter / order (as counting_tail)
        G.split_scan = (unsigned)G.t[0].scan_blocks;
        o->nb_scan = G.split_scan + (unsigned)G.t[1].scan_blocks;
        G.split_n = (unsigned)cdiv(o->cap_ce + 1, kPrepBlock2);
        o->nb_n = G.split_n + (unsigned)cdiv(o->cap_cr + 1, kPrepBlock2);
    }
    return EMG_OK;
}

}  // namespace emg

using namespace emg;

extern "C" int emg_prepare_batch(const emg_prepare_args* a, void* stream) {
    EMG_REQUIRE(a, "emg_prepare_batch: null args");
    EMG_REQUIRE(a->B >= 0, "emg_prepare_batch: bad sizes");
    if (a->B == 0) return EMG_OK;
    hipStream_t st = (hipStream_t)stream;
    PrepStages S;
    int rc = prepare_stages(a, &S);
    if (rc != EMG_OK) return rc;
    if (S.both && !a->ws_clean) {
        rc = clean_ws(S.we, a->ws_ent, st);
        if (rc == EMG_OK) rc = clean_ws(S.wr, a->ws_rel, st);
        if (rc != EMG_OK) return rc;
    }
    if (bucket_eligible(a, S)) return bucket_prepare(a, S, st);
    hipLaunchKernelGGL(prepare_ids_kernel, dim3(S.nb_ids), dim3(256), 0, st, S.prep, S.G);
    EMG_LAUNCH_CHECK();
    if (S.both) {
        if (!S.fused_hist) {
            GroupLaunch H = S.G;
            H.split_n = (unsigned)cdiv(S.cap_ce > S.we.scan_blocks ? S.cap_ce : S.we.scan_blocks, kPrepBlock);
            const unsigned nb = H.split_n + (unsigned)cdiv(S.cap_cr > S.wr.scan_blocks ? S.cap_cr : S.wr.scan_blocks, kPrepBlock);
            hipLaunchKernelGGL(group_hist_kernel, dim3(nb), dim3(256), 0, st, H);
            EMG_LAUNCH_CHECK();
        }
        return counting_tail(S.G, S.cap_ce, S.cap_cr, st);
    }
    EMG_REQUIRE(!a->ctl, "emg_prepare_batch: a device-side batch record needs the counting backend for both tables");
    // mixed / sort backends: table by table
    for (int ti = 0; ti < 2; ++ti) {
        const GroupWs& w = ti ? S.wr : S.we;
        const int32_t* dest = ti ? a->dest_rel : a->dest_ent;
        const int64_t n = ti ? S.n_cr : S.n_ce, R = ti ? a->n_rel : a->n_ent;
        uint8_t* flags = ti ? nullptr : a->single_flags;
        const int32_t* fc = (ti == 0 && a->factored) ? a->codes : nullptr;
        if (!w.counting) { rc = sort_group(dest, n, R, w, flags, st, fc, a->B); if (rc != EMG_OK) return rc; continue; }
        if (!a->ws_clean) { rc = clean_ws(w, ti ? a->ws_rel : a->ws_ent, st); if (rc != EMG_OK) return rc; }
        GroupLaunch G1{};
        G1.n_tables = 1; G1.B = a->B;
        G1.t[0] = S.G.t[ti]; G1.t[1] = S.G.t[ti];
        G1.split_n = (unsigned)cdiv(n > w.scan_blocks ? n : w.scan_blocks, kPrepBlock);
        hipLaunchKernelGGL(group_hist_kernel, dim3(G1.split_n), dim3(256), 0, st, G1);
        EMG_LAUNCH_CHECK();
        rc = counting_tail(G1, n, 0, st);
        if (rc != EMG_OK) return rc;
    }
    return EMG_OK;
}
