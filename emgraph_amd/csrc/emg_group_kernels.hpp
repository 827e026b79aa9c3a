// emg_group_kernels.hpp — device bodies of the counting grouping (see emg_group.hip for the algorithm), written as
// functions of a workgroup index so that they can run as kernels of their own (emg_group.hip) or as RIDERS: extra
// workgroups at the front of the training step's two big launches (emg_score.hip: the fused kernel, emg_apply.hip: the
// apply kernel).  The preparation of the NEXT batches then costs no launch, no side stream and no event: batch t + 2's id
// kernel and batch t + 1's scatter ride with the fused kernel of batch t, their scan / ordering with its apply.
#pragma once
#include "emg_group.hpp"

namespace emg {

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
    // optional (emg_group_dest_keyed): the order of a destination's contributions is ascending order_key[i] instead of ascending
    // input index i; vals[q] then holds the key and srcrow[q] the input index (coef[q] = 1): the apply reads rows through srcrow
    const uint32_t* order_key;
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
__device__ __forceinline__ void group_hist_body(const GroupLaunch& G, unsigned bx) {
    const int ti = bx < G.split_n ? 0 : 1;
    const TableGroup& T = G.t[ti];
    const int64_t base = (int64_t)(bx - (ti ? G.split_n : 0u)) * kPrepBlock + threadIdx.x;
    const int64_t n = table_n(G, ti);
    int32_t d[kPrepItems];
#pragma unroll
    for (int q = 0; q < kPrepItems; ++q) {
        const int64_t i = base + q * 256;
        group_reset(T, i);
        d[q] = i < n ? T.dest[i] : -1;
    }
#pragma unroll
    for (int q = 0; q < kPrepItems; ++q) hist_add(T, d[q]);
}

// 2. scan over the table rows.  Tile = 4096 rows = 256 threads x 16; tiles are taken in ticket order, so every
// predecessor of a tile has started and publishes its aggregate without waiting for anybody (decoupled look-back,
// Merrill & Garland 2016): status word = value << 2 | (1: tile aggregate, 2: inclusive prefix).
__device__ __forceinline__ void group_scan_body(const GroupLaunch& G, unsigned bx) {
    const int ti = bx < G.split_scan ? 0 : 1;
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
__device__ __forceinline__ void group_scatter_body(const GroupLaunch& G, unsigned bx) {
    const int ti = bx < G.split_n ? 0 : 1;
    const TableGroup& T = G.t[ti];
    const int64_t base = (int64_t)(bx - (ti ? G.split_n : 0u)) * kPrepBlock2 + threadIdx.x;
    const int64_t n = table_n(G, ti);
    if (base >= n) return;
    int32_t d[kPrepItems2];
    bool ok[kPrepItems2], single[kPrepItems2];
    uint32_t start[kPrepItems2], next[kPrepItems2], pos[kPrepItems2], key[kPrepItems2];
#pragma unroll
    for (int q = 0; q < kPrepItems2; ++q) {   // phase 1: the ids (an index past the end: the last one again, nothing stored for it)
        const int64_t i = base + q * 256;
        d[q] = T.dest[i < n ? i : n - 1];
        key[q] = T.order_key ? T.order_key[i < n ? i : n - 1] : (uint32_t)i;
    }
#pragma unroll
    for (int q = 0; q < kPrepItems2; ++q) {   // phase 2: their segments
        ok[q] = base + q * 256 < n && d[q] >= 0 && (int64_t)d[q] < T.R;
        const int32_t dd = ok[q] ? d[q] : 0;
        start[q] = T.off[dd];
        next[q] = T.off[dd + 1];
    }
#pragma unroll
    for (int q = 0; q < kPrepItems2; ++q) {   // phase 3: a contribution takes the next free position of its destination's segment
        single[q] = ok[q] && next[q] - start[q] == 1u;
        // (a destination hit once — most of them, for uniform negatives on a large table — needs no cursor)
        pos[q] = start[q];
        if (ok[q] && !single[q]) pos[q] = (uint32_t)atomicAdd(T.cnt + d[q], 1);
    }
#pragma unroll
    for (int q = 0; q < kPrepItems2; ++q) {
        const int64_t i = base + q * 256;
        if (ok[q]) {
            T.tmpv[pos[q]] = key[q];
            if (T.order_key) T.pos_of_slot[pos[q]] = (uint32_t)i;
            T.keys[pos[q]] = (uint32_t)d[q];
        }
        if (T.flags && i < n) T.flags[i] = single[q] ? 1 : 0;
    }
}

// 4. order: rank of a contribution among the slots of its segment = its place in the stable order.
// Factored contributions (see emg_backward_args.fac_ws_ent): srcrow[q] = the row of the 4B-row contribution buffer the slot
// at sorted position q points at, pos_of_slot[slot - 2B] = q for the negatives' slots (where the backward kernel puts
// their factor), coef[q] = 1 for the subject / object slots.
__device__ __forceinline__ void group_order_body(const GroupLaunch& G, unsigned bx) {
    const int ti = bx < G.split_n ? 0 : 1;
    const TableGroup& T = G.t[ti];
    const int64_t base = (int64_t)(bx - (ti ? G.split_n : 0u)) * kPrepBlock2 + threadIdx.x;
    const int64_t n = table_n(G, ti);
#pragma unroll
    for (int q = 0; q < kPrepItems2; ++q) {
        const int64_t t = base + q * 256;
        if (t < 2) T.counters[GC_LONG_COUNT + t] = 0u;          // window-path task list (apply_rows_kernel) starts empty
        if (t <= n / kLongSegment) T.arrive[t] = 0;             // per-segment block counters of the long-segment reduction
    }
    const int64_t total = (int64_t)T.off[T.R];
    if (base >= total) return;
    uint32_t d[kPrepItems2], mine[kPrepItems2], start[kPrepItems2], len[kPrepItems2], rank[kPrepItems2];
    bool on[kPrepItems2];
#pragma unroll
    for (int q = 0; q < kPrepItems2; ++q) {   // phase 1 (a position past the end: the last one again, nothing stored for it)
        const int64_t t = base + q * 256;
        on[q] = t < total;
        const int64_t tt = on[q] ? t : total - 1;
        d[q] = T.keys[tt];
        mine[q] = T.tmpv[tt];
    }
#pragma unroll
    for (int q = 0; q < kPrepItems2; ++q) {   // phase 2
        start[q] = T.off[d[q]];
        len[q] = T.off[d[q] + 1] - start[q];
    }
#pragma unroll
    for (int q = 0; q < kPrepItems2; ++q) {   // phase 3: the rank among the segment's slots (a singleton: nothing to read)
        // A hub row's segment is thousands of slots and every one of its threads walks all of them: 16-byte loads, sixteen
        // values per trip (the scalar loop took 130 us alone / 510 us beside the scoring kernel on the Zipf batch, most of
        // the preparation; short segments never leave the head / tail loops)
        uint32_t r = 0u;
        if (on[q] && len[q] > 1u) {
            const uint32_t* seg = T.tmpv + start[q];
            const uint32_t m = mine[q], ln = len[q];
            uint32_t j = 0u;
            const uint32_t head = min(ln, (4u - (start[q] & 3u)) & 3u);   // up to the first 16-byte boundary
            for (; j < head; ++j) r += seg[j] < m ? 1u : 0u;
            for (; j + 16u <= ln; j += 16u) {
                const uint4 a = *reinterpret_cast<const uint4*>(seg + j), b = *reinterpret_cast<const uint4*>(seg + j + 4),
                            c = *reinterpret_cast<const uint4*>(seg + j + 8), e = *reinterpret_cast<const uint4*>(seg + j + 12);
                r += (a.x < m) + (a.y < m) + (a.z < m) + (a.w < m) + (b.x < m) + (b.y < m) + (b.z < m) + (b.w < m)
                   + (c.x < m) + (c.y < m) + (c.z < m) + (c.w < m) + (e.x < m) + (e.y < m) + (e.z < m) + (e.w < m);
            }
            for (; j + 4u <= ln; j += 4u) {
                const uint4 a = *reinterpret_cast<const uint4*>(seg + j);
                r += (a.x < m) + (a.y < m) + (a.z < m) + (a.w < m);
            }
            for (; j < ln; ++j) r += seg[j] < m ? 1u : 0u;
        }
        rank[q] = r;
    }
    const uint32_t fac_B = (uint32_t)(G.ctl ? G.ctl->B : G.B);
    int32_t fcode[kPrepItems2];
#pragma unroll
    for (int q = 0; q < kPrepItems2; ++q) {   // phase 4 (factored contributions): the codes of the negatives' slots
        fcode[q] = 0;
        if (T.fac_codes && on[q] && mine[q] >= 2u * fac_B) fcode[q] = T.fac_codes[mine[q] - 2u * fac_B];
    }
#pragma unroll
    for (int q = 0; q < kPrepItems2; ++q) {
        if (!on[q]) continue;
        const int64_t t = base + q * 256;
        const uint32_t at = start[q] + rank[q];
        T.vals[at] = mine[q];
        if (T.order_key) { T.srcrow[at] = T.pos_of_slot[t]; T.coef[at] = 1.f; }   // (pos_of_slot: the scatter's second payload here)
        if ((uint32_t)t == start[q]) T.cnt[d[q]] = 0;          // the cursor has done its work: the histogram is zero again
        if (T.fac_codes) {
            if (mine[q] < 2u * fac_B) {
                T.srcrow[at] = mine[q];
                T.coef[at] = 1.f;   // subject / object rows are stored in full (the negatives' factors come from the backward kernel)
            } else {
                const uint32_t i = mine[q] - 2u * fac_B;
                T.srcrow[at] = (fcode[q] < 0 ? 2u : 3u) * fac_B + i % fac_B;
                T.pos_of_slot[i] = at;
            }
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

__device__ __forceinline__ void prepare_ids_body(const PrepParams& P, const GroupLaunch& G, unsigned bx) {
    int64_t B = P.B;
    const int32_t* pos = P.pos;
    uint64_t counter0 = P.counter0, n_choices = P.n_choices;
    const int32_t* elist = P.entities_list;
    if (P.ctl) {
        B = P.ctl->B; pos += 3 * P.ctl->start; counter0 = P.ctl->draw_counter0;
        if (P.ctl->n_choices > 0) { n_choices = (uint64_t)P.ctl->n_choices; elist = P.ctl->entities_list; }
    }
    const int64_t per_side = (int64_t)P.eta * B;
#pragma unroll
    for (int q = 0; q < kPrepItems; ++q) {
        const int64_t j = ((int64_t)bx * kPrepItems + q) * 256 + threadIdx.x;
        if (P.hist) { group_reset(G.t[0], j); group_reset(G.t[1], j); }
        if (j < B) {
            const int32_t s = pos[3 * j + 0], p = pos[3 * j + 1], o = pos[3 * j + 2];
            P.dest_ent[j] = s;
            P.dest_ent[B + j] = o;
            P.dest_rel[j] = p;
            if (P.hist) { hist_add(G.t[0], s); hist_add(G.t[0], o); hist_add(G.t[1], p); }
        }
        if (j >= per_side * P.n_sides) continue;
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
}


// emg_prepare_batch's validation + layout without a launch (emg_group.hip): the stages as launchable descriptions
struct PrepStages {
    PrepParams prep; GroupLaunch G; GroupWs we, wr;
    unsigned nb_ids, nb_scan, nb_n;    // workgroups of the id kernel / the scan / the scatter and order kernels
    bool both, fused_hist;             // both tables on the counting backend; histogram inside the id kernel
    int64_t n_ce, n_cr, cap_ce, cap_cr;
};
int prepare_stages(const emg_prepare_args* a, PrepStages* out);
// the bucket form of the same preparation (emg_group_bucket.hip): two launches, nothing table-sized
bool bucket_eligible(const emg_prepare_args* a, const PrepStages& S);
int bucket_prepare(const emg_prepare_args* a, const PrepStages& S, hipStream_t st);

// ---------------------------------------------------------------------------------------------------------------
// riders: up to two preparation stages in front of a launch's own workgroups
// ---------------------------------------------------------------------------------------------------------------
enum { RIDE_NONE = 0, RIDE_IDS = 1, RIDE_SCAN = 2, RIDE_SCATTER = 3, RIDE_ORDER = 4 };
struct Rider { int32_t kind; uint32_t blocks; GroupLaunch G; };
struct Riders { Rider r[2]; PrepParams prep; uint32_t total; uint32_t pad0; };   // prep: of the RIDE_IDS rider (at most one)

// true: this workgroup was a rider (the kernel returns); false: *bx = the workgroup's index among the kernel's own
__device__ __forceinline__ bool run_riders(const Riders& R, unsigned* bx) {
    unsigned b = blockIdx.x;
    if (b >= R.total) { *bx = b - R.total; return false; }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const Rider& r = R.r[i];
        if (b < r.blocks) {
            if (r.kind == RIDE_IDS) prepare_ids_body(R.prep, r.G, b);
            else if (r.kind == RIDE_SCAN) group_scan_body(r.G, b);
            else if (r.kind == RIDE_SCATTER) group_scatter_body(r.G, b);
            else if (r.kind == RIDE_ORDER) group_order_body(r.G, b);
            return true;
        }
        b -= r.blocks;
    }
    return true;
}

// the same stages as launches of their own (a launch that cannot carry riders: window apply, unfused step)
int launch_riders_alone(const Riders& R, hipStream_t st);
static inline void add_rider(Riders& R, int kind, unsigned blocks, const GroupLaunch& G, const PrepParams* prep) {
    const int i = R.r[0].kind == RIDE_NONE ? 0 : 1;
    R.r[i].kind = kind; R.r[i].blocks = blocks; R.r[i].G = G;
    if (prep) R.prep = *prep;
    R.total += blocks;
}

}  // namespace emg
