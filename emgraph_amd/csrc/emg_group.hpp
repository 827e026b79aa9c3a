// emg_group.hpp — the grouping workspace: what emg_group.hip produces about a batch's gradient contributions and what
// the backward kernel (emg_score.hip) and the row-sparse apply (emg_apply.hip) consume.
//
// Two backends fill it (group_backend()):
//   COUNTING (the training path: destinations are rows of a table, n_rows <= 16 n + 2^20): histogram over the row ids
//       (int32 atomics, fused into the id kernel) -> one decoupled-look-back scan over the rows -> scatter -> in-segment
//       ordering.  Besides the stable order (keys / vals) the scan emits SEGMENT DESCRIPTORS — the list of destinations
//       with 2..kDeferSegment contributions, the list of singletons, the block tasks of longer segments — so that the
//       apply kernel starts streaming rows at instruction 0 instead of searching segment ends.
//   SORT (wide keys: the filter index's (entity, relation) keys, tables far larger than the batch): rocPRIM's
//       device radix sort; the apply then runs the window kernels, which find their segments themselves.
#pragma once
#include "emg_common.hpp"

namespace emg {

constexpr int kLongSegment = 64;   // rows; the block size of the long-segment reduction tree
constexpr int kDeferSegment = 32;  // segments longer than this leave the per-segment path as block tasks
struct LongTask { uint32_t head, block, len; };   // sorted position of the segment's head, block index, rows of the segment (0 = void)
struct Seg { uint32_t start, len, dest; };        // contributions at sorted positions [start, start + len) belong to table row dest

// words of GroupWs::counters
enum { GC_MULTI = 0, GC_SINGLE = 1, GC_TASKS = 2, GC_VALID = 3,      // written by the scan: list lengths, grouped contributions
       GC_SCAN_TICKET = 4, GC_SCAN_DONE = 5,                          // scan bookkeeping (zero between launches)
       GC_LONG_COUNT = 8, GC_LONG_DONE = 9,                           // window path: task list length / finished workgroups
       GC_WORDS = 64 };

// Per-step values a captured step graph cannot bake into kernel arguments (include/emgraph_hip.h: emg_step_ctl)
typedef emg_step_ctl StepCtl;

struct GroupWs {
    bool counting;
    size_t kb;                                   // bytes of one N-entry uint32 region
    uint32_t *keys, *vals, *tmpv, *srcrow, *pos_of_slot;
    float* coef;
    Seg* multi; uint32_t* single; LongTask* tasks; int32_t* arrive; uint32_t* counters;
    unsigned long long* status; int scan_blocks;
    int32_t* cnt; uint32_t* off;                 // counting backend: per-row count / cursor, exclusive offsets [R + 1]
    void* sort_tmp; size_t sort_tmp_bytes;       // sort backend
    uint32_t *tmpv2, *bmat;                      // bucket grouping: second half of the chunk-ordered pairs; chunk x bucket offset matrix
    float* partial;                              // block sums of the long-segment reduction (nullptr: no room)
    size_t clean_offset, clean_bytes;            // the control region that must be zero before the first grouping
    uint32_t task_cap;
};

constexpr int kScanTile = 4096;                  // rows per scan workgroup (256 threads x 16)
// Contributions per THREAD of the per-contribution stages (ids + histogram, scatter, order), taken 256 apart so that every load
// of a wave stays coalesced, each stage written in phases (all loads of a phase issued before anything waits).  These kernels
// stream beside the scoring kernel; a contribution is a chain of dependent round trips (id -> offsets -> cursor -> store) and a
// wave is alive for as long as its chain.  Four chains per thread: a quarter of the waves, each living little longer — measured,
// C3: step 0.3633 -> 0.3555 ms (scoring kernel 0.280 -> 0.274 with the preparation beside it); eight: 0.3551, but the small
// batches' riders then are their launch's long pole (C1 0.062 -> 0.071).  (Not a matter of registers: the scoring kernel of C3
// holds 2 x 200 of a SIMD's 512, a preparation wave of <= 56 fits beside it in either form.)
#ifndef EMG_PREP_ITEMS
#define EMG_PREP_ITEMS 4
#endif
constexpr int kPrepItems = EMG_PREP_ITEMS;
constexpr int kPrepBlock = 256 * kPrepItems;     // contributions per workgroup of the id / histogram stages
// scatter and order can take a count of their own (A/B aid; default: the same)
#ifndef EMG_PREP_ITEMS2
#define EMG_PREP_ITEMS2 EMG_PREP_ITEMS
#endif
constexpr int kPrepItems2 = EMG_PREP_ITEMS2;
constexpr int kPrepBlock2 = 256 * kPrepItems2;   // contributions per workgroup of the scatter / order stages

// BUCKET grouping (emg_group_bucket.hip; round 5): the counting grouping for tables far larger than L2.  The table-sized
// histogram / offset arrays of the counting backend are random 4-byte accesses into 4 MB arrays per contribution (the 117 MB
// per step of profiles/r4_z beside the scoring kernel); here a contribution is bucketed by its destination's high bits inside
// the id kernel's workgroup (LDS histogram of <= 4096 bins, chunk-ordered pairs + one offset row per chunk), and ONE workgroup
// per bucket of <= 2048 table rows does histogram, scan, scatter, in-segment ordering and the segment descriptors in LDS.
// Two launches for both tables, no global atomics but the three list stretches per bucket, nothing table-sized.
constexpr int kBucketChunkMin = 1024;  // contribution slots per workgroup of the id kernel (doubled until a batch is <= 512 chunks)
constexpr int kBucketChunksMax = 512;  // chunks per batch = entries of a row of the offset matrix (bmat[bucket][chunk])
constexpr int kBucketRowsMax = 2048;   // table rows per bucket (the LDS row table of the bucket kernel)
constexpr int kBucketMaxNB = 4096;     // buckets per table (the LDS histogram of the id kernel)
constexpr int kBucketCap = 4096;       // contributions of a bucket sorted in LDS; a fuller bucket is sorted through global memory
// Tables of up to kDenseHereMaxRows rows may run Keras Adam's dense pass INSIDE the descriptor-driven apply launch, which finds the
// untouched rows in the grouping's offset array (emg_apply.hip: dense_here).  The bucket grouping starts ABOVE that size for the
// ENTITY table, forced or not (smaller tables keep the counting grouping: their row arrays live in L2) — but it groups the
// RELATION table of the same batch too, which usually IS that small: for such a table its sort kernel writes the offset array
// as well (BucketTable::off; round 6, the round-5 advisor's finding: a standalone relation apply read an array nobody wrote).
constexpr int64_t kDenseHereMaxRows = 131072;
constexpr int64_t kBucketMinRows = 2 * kDenseHereMaxRows;
struct BucketGeo { int sh, nb, chunk_log; bool ok; };   // bucket = row >> sh; nb buckets; chunks of 1 << chunk_log slots
BucketGeo bucket_geometry(int64_t N, int64_t R);
// 0: counting grouping as ever; 1: bucket grouping where eligible (default); EMG_GROUPING=count|bucket|sort
bool group_backend_bucket(int64_t n_ent);

bool group_backend_counting(int64_t N, int64_t R);
int64_t group_ws_bytes(int64_t N, int64_t R, int64_t ldp);
// N: contribution slots the layout is sized for (a plan: its capacity), R: table rows, ldp: floats per partial row (0: none)
int group_ws_layout(void* ws, int64_t ws_bytes, int64_t N, int64_t R, int64_t ldp, GroupWs* out);

static inline size_t partial_rows(int64_t n) { return 2 * ((size_t)n / kLongSegment + 2); }

// What the backward kernel needs of an entity grouping workspace when contributions are FACTORED: where each negative's
// slot landed in the sorted order, and the per-position factor array it fills.
struct FactorView { const uint32_t* pos_of_slot; float* coef; };
int factor_view(void* workspace, int64_t workspace_bytes, int64_t N, int64_t R, FactorView* out);

}  // namespace emg
