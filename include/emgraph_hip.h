/* emgraph_hip.h — C-ABI of libemgraph_hip.so: the MI355X (gfx950) replacement for
 * bi-graph/Emgraph's per-batch hot path.
 *
 * The reference (pure Python on TensorFlow) has no FFI of its own; the entry points below are
 * what a ctypes binding inside the reference would call INSTEAD of the TF op groups cited at
 * each declaration (file:line relative to the reference tree).  INTEGRATION.md shows that stub.
 *
 * Conventions
 *   - every function returns 0 on success, a negative EMG_E* code otherwise; the message for the
 *     calling thread's last failure is emg_last_error().  No C++ exception crosses the ABI.
 *   - all pointers are DEVICE pointers on the current HIP device unless marked "host"; the caller
 *     owns every buffer (in our host code they are PyTorch-ROCm tensors); the library allocates
 *     nothing persistent — scratch comes from caller-provided workspaces sized by *_workspace_bytes.
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous and ordered on it.
 *   - embedding tables are float32 row-major with a row stride `ld` (in floats, ld >= k_int).
 *     ComplEx/HolE rows are [re(0:k) | im(k:2k)], k_int = 2k (ComplEx.py:224,288-290).
 *   - triples are int32 [n,3] row-major (s,p,o)  (EmbeddingModel.py:503-505).
 *   - eta-major negatives: negative row j (0 <= j < eta*B) corrupts positive j mod B
 *     (protocol.py:598).
 */
#ifndef EMGRAPH_HIP_H
#define EMGRAPH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMG_ABI_VERSION 8

#define EMG_OK 0
#define EMG_EINVAL (-1)   /* bad argument */
#define EMG_EHIP (-2)     /* HIP runtime error (message has hipGetErrorString) */
#define EMG_ENOSUP (-3)   /* combination not supported */

/* model ids: TransE.py:208-216 (norm 1 / 2), DistMult.py:201, ComplEx.py:288-298, HolE.py:189 */
#define EMG_TRANSE_L1 0
#define EMG_TRANSE_L2 1
#define EMG_DISTMULT 2
#define EMG_COMPLEX 3
#define EMG_HOLE 4
/* TransE with ANY positive order of the norm (TransE.py:208-216 hands `norm` to tf.norm as ord): f = -(sum |e_s + r_p - e_o|^ord)^(1/ord),
 * ord = +inf: the largest |component|; the `scale` argument of every call carries ord.  Inference: emg_score_triples and the
 * emg_eval_* / emg_rank_1vsall family at precision 0.  Training (round 4): the unfused step — emg_train_forward (eta > 0), emg_loss,
 * emg_train_backward_ex(fused_loss = -1: gradient -g sgn(d)|d|^(ord-1) / ||d||^(ord-1); ord = inf: shared by the tied maxima) — with
 * every gradient row through emg_apply_grouped (no in-place updates, no fused loss, no column slabs).  Generic kernels: orders 1
 * and 2 are EMG_TRANSE_L1 / _L2, the tuned (fused, MFMA- and v_sad-accelerated) models. */
#define EMG_TRANSE_P 5

/* corruption side: protocol.py:598-608 ('s,o' is an alias of 's+o' there, :591-593) */
#define EMG_SIDE_S 0
#define EMG_SIDE_O 1
#define EMG_SIDE_SO 2

/* losses: losses/pairwise.py:66-70, nll.py:55-59, absolute_margin.py:66-70,
 * self_adversarial.py:90-112, nll_multiclass.py:70-81 */
#define EMG_LOSS_PAIRWISE 0
#define EMG_LOSS_NLL 1
#define EMG_LOSS_ABSOLUTE_MARGIN 2
#define EMG_LOSS_SELF_ADVERSARIAL 3
#define EMG_LOSS_MULTICLASS_NLL 4

/* optimizers: training/sgd.py:97, momentum.py:63, adagrad.py:42, adam.py:45 (Keras rules) */
#define EMG_OPT_SGD 0
#define EMG_OPT_MOMENTUM 1
#define EMG_OPT_ADAGRAD 2
#define EMG_OPT_ADAM 3        /* Keras sparse apply == dense-equivalent: every row decays/updates */
#define EMG_OPT_ADAM_LAZY 4   /* touched rows only (not reference semantics; opt-in) */

/* score flags */
#define EMG_SCORE_FINAL 0     /* the model's score */
#define EMG_SCORE_PARTIAL 1   /* k-slice partial: TransE-L2 returns sum(d^2) (no -sqrt), HolE unscaled;
                                 finish with emg_finalize_scores after summing slices */

int emg_version(void);
const char* emg_last_error(void);
/* name of the compiled GPU target ("gfx950") */
const char* emg_target(void);
/* hash of the kernel sources the library was built from (the rocprofv3 evidence under profiles/ names the binary it measured) */
const char* emg_source_hash(void);

/* ---- K1+K2: fused embedding gather + score (replaces EmbeddingModel._lookup_embeddings
 * :490-533 followed by Model._fn; the predict() path :2132-2133). out[n] f32. */
int emg_score_triples(int model, const float* ent, int64_t n_ent, int64_t ld_ent,
                      const float* rel, int64_t n_rel, int64_t ld_rel, int32_t k_int, float scale,
                      const int32_t* spo, int64_t n, int32_t flags, float* out, void* stream);

/* finish partial scores in place (TransE-L2: -sqrt(x); HolE: scale*x; others unchanged) */
int emg_finalize_scores(int model, float scale, float* scores, int64_t n, void* stream);

/* ---- K3/K14: corruption draws (replaces the tf.random.uniform + mask logic of
 * generate_corruptions_for_fit, protocol.py:598-641).
 * codes[j] = replacement_entity | (keep_subject << 31), j in [0, B*eta).
 *   side S: keep_subject=0 (subject replaced); O: keep_subject=1; SO: Bernoulli(1/2) per row.
 *   replacement = idx if entities_list==NULL else entities_list[idx], idx ~ U{0..n_choices-1}
 *   drawn from Philox4x32-10(key=seed, counter=(j, draw_counter)); or, when inj_repl != NULL,
 *   idx = inj_repl[j] (and keep_subject = inj_mask[j] for side SO) — the "injected draws" mode
 *   used to reproduce the reference's golden vectors. */
int emg_corrupt_codes(int64_t B, int32_t eta, int side, int64_t n_choices,
                      const int32_t* entities_list, uint64_t seed, uint64_t draw_counter,
                      const int32_t* inj_mask, const int32_t* inj_repl, int32_t* codes, void* stream);

/* materialise the [B*eta,3] corruption array exactly as generate_corruptions_for_fit returns it
 * (protocol.py:643-656) from the positives and the codes */
int emg_corrupt_expand(const int32_t* pos, int64_t B, int32_t eta, const int32_t* codes,
                       int32_t* out_spo, void* stream);

/* ---- K1+K2+K4 for a training batch: scores of the B positives and of their eta*B negatives
 * given as codes; the [B*eta,3] array and the three gathered [n,k] temporaries of the reference
 * (EmbeddingModel.py:675-677,788-799) are never materialised.  Rows shared inside a positive
 * group (p and the kept side) are read once.  scores_neg is eta-major. */
int emg_train_forward(int model, const float* ent, int64_t n_ent, int64_t ld_ent,
                      const float* rel, int64_t n_rel, int64_t ld_rel, int32_t k_int, float scale,
                      const int32_t* pos, int64_t B, int32_t eta, const int32_t* codes, int32_t flags,
                      float* scores_pos, float* scores_neg, void* stream);

/* ---- K5: loss value + dL/dscore (replaces Loss.apply, losses/loss.py:123-138, and TF autodiff
 * through it).  scores_neg is [n_sides*eta*B] (side-major, then eta-major); the positive is tiled
 * per side as EmbeddingModel.py:724-729,786-816.  *loss_accum (device f64) += loss.
 * g_pos[B] = sum over tiles/sides of dL/dpos, g_neg[n_sides*eta*B] = dL/dneg. */
int emg_loss(int loss, const float* scores_pos, const float* scores_neg, int64_t B, int32_t eta,
             int32_t n_sides, float margin, float alpha, double* loss_accum, float* g_pos, float* g_neg,
             void* stream);

/* ---- K7: adjoint of gather+score.  Writes per-group gradient rows (no atomics):
 *   contrib_ent[0*B+i] = dL/dE[s_i] from group i,  contrib_ent[1*B+i] = dL/dE[o_i],
 *   contrib_ent[2*B + j*B + i] = dL/dE[replacement of negative j of positive i]   (j < eta)
 *   contrib_rel[i] = dL/dR[p_i];   row stride ldc floats (>= k_int).
 * dest_ent[(2+eta)*B], dest_rel[B] receive the destination row ids of those rows. */
int emg_train_backward(int model, const float* ent, int64_t n_ent, int64_t ld_ent,
                       const float* rel, int64_t n_rel, int64_t ld_rel, int32_t k_int, float scale,
                       const int32_t* pos, int64_t B, int32_t eta, const int32_t* codes,
                       const float* g_pos, const float* g_neg,
                       float* contrib_ent, float* contrib_rel, int64_t ldc,
                       int32_t* dest_ent, int32_t* dest_rel, void* stream);

/* destination ids of the (2+eta)*B entity / B relation contribution rows of a batch (layout above);
 * depends only on the batch ids and codes, so it can run ahead of the scoring kernels */
int emg_build_dest(const int32_t* pos, int64_t B, int32_t eta, const int32_t* codes,
                   int32_t* dest_ent, int32_t* dest_rel, void* stream);

/* Per-step values of a training step that a captured step graph cannot bake into kernel arguments: a DEVICE record
 * every kernel of the step reads instead (emg_prepare_args.ctl, emg_backward_args.ctl, emg_apply_args.ctl; NULL = the
 * values of the argument struct).  emg_plan_run fills an array of them per replay.  The batch is rows
 * [start, start + B) of the resident training set (the `pos` pointer of the call is then the set's row 0). */
typedef struct emg_step_ctl {
    int64_t start; int64_t B;
    uint64_t draw_counter0;                                  /* as emg_prepare_args.draw_counter0 */
    int64_t n_choices; const int32_t* entities_list;         /* corruption pool (n_choices 0: the call's) */
    int32_t step; int32_t reserved0;                         /* optimizer step number (>= 1) */
    float hyper_ent[8]; float hyper_rel[8];                  /* as emg_apply_rows' hyper; entries 0 (lr) and 5 (lr_t) are read */
} emg_step_ctl;

/* ---- everything about a training batch that does not depend on the tables, in one call (what the step
 * pipeline runs ahead on a side stream): the corruption codes of every corruption side (as emg_corrupt_codes,
 * side sd drawing with counter draw_counter0 + sd, codes side-major then eta-major), the destination ids of
 * all gradient rows (as emg_build_dest) and their stable grouping + singleton flags (as emg_group_dest) for
 * the entity and the relation table.  dest_ent / dest_rel start with n_extra_* caller-filled entries (the LP
 * regulariser's dense rows) followed by the batch's; the grouping covers both.
 * Batch-sharded multi-GPU training: `pos` holds rows [row_offset, row_offset + B) of a GLOBAL batch of B_global
 * positives; the draw of (negative jj, local row i) is then the one the whole batch would use for its row
 * jj * B_global + row_offset + i, so the negatives do not depend on how many GPUs share the batch.
 * B_global == 0 means B_global = B, row_offset = 0 (single GPU). */
typedef struct emg_prepare_args {
    const int32_t* pos; int64_t B; int32_t eta; int32_t n_sides; int32_t sides[4];
    int64_t n_choices; const int32_t* entities_list; uint64_t seed; uint64_t draw_counter0;
    const int32_t* inj_mask; const int32_t* inj_repl;       /* optional injected draws [n_sides*eta*B] */
    int32_t* codes;                                          /* out [n_sides*eta*B] */
    int32_t* dest_ent; int64_t n_extra_ent; int64_t n_ent;   /* out [n_extra_ent + (2+n_sides*eta)*B] */
    int32_t* dest_rel; int64_t n_extra_rel; int64_t n_rel;   /* out [n_extra_rel + B] */
    void* ws_ent; int64_t ws_ent_bytes; void* ws_rel; int64_t ws_rel_bytes;  /* emg_apply_workspace_bytes */
    uint8_t* single_flags;                                   /* optional out, per entity contribution row */
    int64_t B_global; int64_t row_offset;                    /* batch-sharded draws (see above); 0, 0 = whole batch */
    /* factored != 0 (bilinear models, n_extra_ent = 0): the entity workspace also receives, per sorted position, the
     * row of the 4*B-row contribution buffer its slot points at, and per negative slot its sorted position — what
     * emg_train_backward_ex (fac_ws_ent) and emg_apply_grouped_factored work from. */
    int32_t factored;
    /* ws_clean != 0: the caller zeroed the workspaces once (hipMemset of the whole buffer) and has used them only through
     * this library since — the grouping then skips re-zeroing its control region (every grouping leaves it zero) */
    int32_t ws_clean;
    /* layout_B > 0: the workspaces are laid out (and the launches sized) for layout_B >= B positives — a plan's capacity,
     * so that batches of different sizes share one layout; ctl: optional device record (see emg_step_ctl) */
    int64_t layout_B; const void* ctl;
} emg_prepare_args;
int emg_prepare_batch(const emg_prepare_args* args, void* stream);

/* ---- K5+K7 fused / extended backward.  One pass over the (3+eta) rows of each positive group:
 *   fused_loss >= 0 (EMG_LOSS_PAIRWISE | EMG_LOSS_NLL | EMG_LOSS_ABSOLUTE_MARGIN — the losses whose
 *       dL/dneg_j depends only on (pos_i, neg_j)): scores, loss (accumulated into *loss_accum) and
 *       dL/dscore are computed in the kernel; scores_*_out optional.
 *   fused_loss < 0: dL/dscore is read from g_pos / g_neg (any loss; see emg_loss).
 *   single_ent != NULL: uint8 per entity contribution slot (from emg_group_dest); slots flagged 1 have a
 *       destination hit exactly once in this batch and are applied to the table IN PLACE (optimizer `opt`,
 *       hyper = 8 floats as emg_apply_rows) instead of being written to contrib_ent — finish with
 *       emg_apply_grouped(..., skip_single=1).
 *   bw_scores_* : optional global final scores, used only by TransE-L2 when `ent`/`rel` are column
 *       slices (k-sharded multi-GPU) so that the norm is the full one.
 * Relation gradients always go to contrib_rel. */
typedef struct emg_backward_args {
    int32_t model; int32_t k_int; float scale; int32_t eta;
    const float* ent; int64_t n_ent; int64_t ld_ent;
    const float* rel; int64_t n_rel; int64_t ld_rel;
    const int32_t* pos; int64_t B; const int32_t* codes;
    int32_t fused_loss; float margin; double* loss_accum;
    const float* g_pos; const float* g_neg;
    const float* bw_scores_pos; const float* bw_scores_neg;
    float* scores_pos_out; float* scores_neg_out;
    float* contrib_ent; float* contrib_rel; int64_t ldc;
    const uint8_t* single_ent; int32_t opt; int32_t step; float hyper[8];
    float* ent_state0; float* ent_state1; int32_t* tag_ent;
    /* FACTORED entity contributions (NULL = off; DistMult / ComplEx / HolE only).  The gradient row of a negative's
     * replacement entity is  gi * q, q being one of the TWO query rows of its triple group (object side: q(s, p);
     * subject side: q(p, o)) — so instead of eta full rows per group the kernel writes q once per side and ONE float
     * per negative.  fac_ws_ent = the entity workspace of the emg_prepare_batch call (factored = 1) for this batch: the
     * float goes straight to the sorted position of the negative's slot there (slots updated in place write nothing).
     * contrib_ent then has 4*B rows: [0,B) subject rows, [B,2B) object rows, [2B,3B) q object side, [3B,4B) q subject
     * side, and the table apply is emg_apply_grouped_factored (same sums, same order, same bits: it adds the rounded
     * product gi * q exactly where the unfactored path adds the stored row). */
    void* fac_ws_ent; int64_t fac_ws_ent_bytes;
    int64_t layout_B; const void* ctl;   /* as in emg_prepare_args: layout of fac_ws_ent; device record, pos = row 0 of the resident set */
    double* lp_accum;   /* folded LP with in-place updates (below): += sum |w|^p (pre-update) over the rows updated in place */
    /* lr_hist != NULL (EMG_OPT_ADAM in place with inplace_window, fused loss, deferred dense pass — see emg_deferred_catchup): the rows of SINGLETON
     * NEGATIVES are as of tag_ent[row]; the kernel fetches (w, m, v) of such a row together, replays the steps tag + 1 .. step - 1 in
     * registers (the dense pass's update with g = 0 and lr_hist[s]), scores the row, applies this step's update and writes
     * (w, m, v, tag = step): one read and one write of the three rows where catch-up + scoring + apply moved twelve.  The same for
     * singleton subject / object rows (replayed before the group's queries are built).  Finish with emg_apply_grouped_ex(skip_single
     * = 1) after emg_deferred_catchup(..., skip_single_from = 0) brought every other row of the batch up to date. */
    const float* lr_hist;
    /* inplace_window != 0 (stateful optimizers; fused loss; 16-byte rows of at most 64 chunks, per half for complex models): a
     * singleton negative's optimizer state rows are fetched together with its table row in the kernel's rolling window — the update
     * waits for nothing (0: the state is read chunk by chunk at the update).  The subject / object singletons are updated in place too
     * (without lr_hist: at the group's end, state read chunk by chunk; with lr_hist: replayed at the group's start, parked in LDS,
     * updated at its end): finish with skip_single = 1.  Required by lr_hist. */
    int32_t inplace_window;
    /* loss_slots: 0 / 1: loss_accum is ONE double.  A power of two n > 1: loss_accum points to n doubles, workgroup b of the fused
     * kernel adds its partial sum to loss_accum[b & (n - 1)], the batch's loss is their sum.  (One double atomic per workgroup to
     * ONE address retires one per ~10 ns at the memory side of the eight L2s: the 680 workgroups of a 20 us launch — the reference's
     * own configurations — all arrive within its last microseconds; measured 5 of C2's 22 us.) */
    int32_t loss_slots;
} emg_backward_args;
/* hyper[6] = lambda, hyper[7] = p of an LP regulariser folded into the update (see emg_apply_grouped): with single_ent != NULL
 * only for opt = EMG_OPT_SGD — a singleton row is then updated in place with g + lambda p |w|^(p-1) sign(w), the rule the
 * apply uses for every other row, tagged, and its |w|^p added to *lp_accum; with a stateful optimizer and a regulariser
 * pass single_ent = NULL (every row through emg_apply_grouped). */
int emg_train_backward_ex(const emg_backward_args* args, void* stream);

/* ---- K8 in two halves (emg_apply_rows = both):
 * emg_group_dest: stable grouping of (dest, index) into `workspace` (+ optional singleton flags[n]) — a counting sort over
 *   the table rows (histogram, one scan, scatter, in-segment ordering: csrc/emg_group.hip) that also emits the segment
 *   descriptors emg_apply_grouped works from; keys wider than 16 n + 2^20 go through a device radix sort instead;
 * emg_apply_grouped: segmented sum + optimizer update using that workspace; skip_single != 0 skips
 * length-1 segments (already applied in place by emg_train_backward_ex). */
int emg_group_dest(const int32_t* dest, int64_t n, int64_t n_rows, void* workspace, int64_t workspace_bytes,
                   uint8_t* single_flags, void* stream);
/* emg_group_dest with an explicit order: a destination's contributions are summed in ascending order_key (distinct within a
 * destination) instead of ascending index — the batch-sharded multi-GPU step: the owner of a table range receives gradient rows
 * from every rank and adds them in the order of their slots in the GLOBAL batch.  Counting backend only (EMG_ENOSUP otherwise).
 * Apply with emg_apply_grouped_factored(contrib = the received rows): it reads row i through the workspace's source array. */
int emg_group_dest_keyed(const int32_t* dest, const uint32_t* order_key, int64_t n, int64_t n_rows, void* workspace,
                         int64_t workspace_bytes, void* stream);
int emg_apply_grouped(int opt, float* table, int64_t n_rows, int64_t ld, int32_t k_int,
                      float* state0, float* state1, int32_t* tag, int32_t step,
                      const float* contrib, int64_t ldc, int64_t n_contrib, int32_t skip_single,
                      const float* hyper, double* lp_accum, void* workspace, int64_t workspace_bytes, void* stream);
/* emg_apply_grouped for an entity table whose contributions the backward kernel wrote FACTORED (emg_backward_args.
 * fac_ws_ent = this workspace, grouped by emg_prepare_batch with factored = 1): n_contrib = (2 + eta) * B contribution
 * slots, `contrib` the 4*B-row buffer. */
int emg_apply_grouped_factored(int opt, float* table, int64_t n_rows, int64_t ld, int32_t k_int,
                               float* state0, float* state1, int32_t* tag, int32_t step,
                               const float* contrib, int64_t ldc, int64_t n_contrib, int32_t skip_single,
                               const float* hyper, double* lp_accum, void* workspace, int64_t workspace_bytes, void* stream);
/* emg_apply_grouped[_factored] with every argument in a struct, and the PAIR form: two tables (a training step's entity
 * and relation table) through shared launches — one window kernel over both groupings, one task kernel over both task
 * lists.  Same results as two calls; where the two shapes cannot share a launch (scalar or <= 16-chunk rows, only one of
 * the workspaces sized for tasks) it IS two calls. */
typedef struct emg_apply_args {
    int32_t opt; int32_t k_int; float* table; int64_t n_rows; int64_t ld;
    float* state0; float* state1; int32_t* tag; int32_t step; int32_t skip_single;
    const float* contrib; int64_t ldc; int64_t n_contrib;
    float hyper[8]; double* lp_accum; void* workspace; int64_t workspace_bytes;
    int32_t factored;
    int32_t table_index;                 /* 0 entity / 1 relation table: which hyper-parameters of `ctl` apply */
    int64_t layout_n; const void* ctl;   /* layout_n > 0: contribution slots the workspace was laid out for (>= n_contrib); device record */
    int32_t deferred_dense; int32_t reserved1;   /* 1: no dense pass (Keras Adam's decay, the LP regulariser's): the caller runs emg_deferred_catchup, below;
                                                  * 2: the same, and the catch-up ran with w_only (m, v of the destinations lag behind w) */
} emg_apply_args;
int emg_apply_grouped_ex(const emg_apply_args* args, void* stream);
int emg_apply_grouped_pair(const emg_apply_args* a, const emg_apply_args* b, void* stream);
/* DEFERRED dense pass.  Keras Adam (training/adam.py:31-48) decays m, v and moves w of EVERY row every step; a folded LP
 * regulariser (regularizers/lp.py:107-113) gives every row a gradient every step: on a 1M x 400 table that is 3.2 - 9.6 GB
 * read + written per step.  Deferred: a row nothing touches keeps (w, state) as of tag[row], the last step it was written
 * at; before a batch is scored, emg_deferred_catchup replays the missed steps tag[row]+1 .. upto — the dense pass's own
 * update (g = the regulariser's gradient alone) with THAT step's learning rate lr_hist[step] (lr_t for Adam, lr otherwise),
 * the same float operations in the same order, so the same bits — for exactly the rows the batch will read and update: the
 * destinations of the grouping in `workspace` (emg_prepare_batch's, counting backend).  The apply then runs with
 * deferred_dense = 1 (no dense pass).  emg_deferred_materialize does the same for every row (before the tables are read,
 * and — with a regulariser — before a loss is reported: *lp_accum += sum |w|^p of every replayed step).  hyper: the 8 values
 * of emg_apply_grouped (hyper[0] / hyper[5] are replaced per step by lr_hist).
 * w_only = 1 (EMG_OPT_ADAM, no regulariser; the apply must then run with deferred_dense = 2): for the destinations the apply
 * finishes with one wave each (the grouping's multi / single lists, gaps of at most 64 steps) only w is written back — m, v
 * and tag[row] stay as of the row's last write, and the apply redoes their decay over the missed steps in registers
 * (two multiplications per element and step) instead of this call writing and the apply re-reading both state rows.
 * Same bits.
 * skip_single_from >= 0: singleton destinations whose contribution slot is >= it are left alone — the scoring kernel replays them
 * itself as it gathers them (emg_backward_args.lr_hist: 0 = every singleton); < 0: none are. */
int emg_deferred_catchup(int opt, float* table, int64_t n_rows, int64_t ld, int32_t k_int, float* state0, float* state1,
                         int32_t* tag, const float* hyper, const float* lr_hist, int32_t upto_step, double* lp_accum,
                         const void* workspace, int64_t workspace_bytes, int64_t layout_n, int32_t w_only, int64_t skip_single_from,
                         void* stream);
int emg_deferred_materialize(int opt, float* table, int64_t n_rows, int64_t ld, int32_t k_int, float* state0, float* state1,
                             int32_t* tag, const float* hyper, const float* lr_hist, int32_t upto_step, double* lp_accum,
                             void* stream);
/* LP regulariser folded into the optimizer step (hyper[6] = lambda != 0, hyper[7] = p): the penalty covers the FULL
 * table (regularizers/lp.py:107-113, EmbeddingModel.py:818-820), so its gradient lambda*p*|w|^(p-1)*sign(w) reaches
 * every row.  Rows with contributions (and rows the backward kernel updates in place) add it to their summed
 * gradient; rows nothing touched in this step (tag[r] != step) are visited by ONE dense pass that applies the
 * optimizer with that gradient alone.  *lp_accum (device double, may be NULL) += sum of |w|^p over all rows at their
 * pre-update values — the caller multiplies by lambda for the loss.  Needs `tag`. */

/* ---- K8: deterministic row-sparse optimizer apply.  Sorts (dest, index) (stable radix sort),
 * sums each destination's contribution rows in index order and updates that table row once.
 * state0/state1: momentum buffer | adagrad accumulator | adam m, v  (same shape/stride as the table;
 * NULL when unused).  `tag` int32[n_rows] scratch owned by the caller (persistently, zero-initialised
 * once) marks rows touched in step `step` (>=1) — needed by EMG_OPT_ADAM's dense pass.
 * hyper: HOST pointer to 8 floats {lr, momentum, beta1, beta2, eps, lr_t, lp_lambda, lp_p} read at call time
 * (lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t), Adam only; lp_lambda = 0: no regulariser folded in, see emg_apply_grouped). */
int64_t emg_apply_workspace_bytes(int64_t n_contrib, int64_t n_rows);
/* the same plus scratch for the long-segment reduction (destinations hit by more than 64 contributions in a batch —
 * hub entities of a skewed graph, relation rows — are summed as 64-row blocks by a whole workgroup instead of by one
 * wave); with the smaller workspace above such segments are summed by a single wave (slow, same reduction order only
 * for segments of up to 64 rows) */
int64_t emg_apply_workspace_bytes_ex(int64_t n_contrib, int64_t n_rows, int32_t k_int);
int emg_apply_rows(int opt, float* table, int64_t n_rows, int64_t ld, int32_t k_int,
                   float* state0, float* state1, int32_t* tag, int32_t step,
                   const float* contrib, int64_t ldc, const int32_t* dest, int64_t n_contrib,
                   const float* hyper, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- K6: LP regulariser over FULL tables (regularizers/lp.py:107-113): *loss_accum +=
 * lambda * sum |W|^p, and (if grad_scale_lr != 0) the SGD-style in-place update
 * W -= grad_scale_lr * lambda * p * |W|^(p-1) * sign(W). */
int emg_lp_regularizer(float* table, int64_t n_rows, int64_t ld, int32_t k_int, float lambda, int32_t p,
                       float grad_scale_lr, double* loss_accum, void* stream);

/* K6, optimizer-agnostic form: the regulariser's gradient as n_rows extra contribution rows
 * (contrib[r] = lambda*p*|W[r]|^(p-1)*sign(W[r]), dest[r] = r) to be appended to the batch's
 * contributions before emg_apply_rows; also accumulates the loss term. */
int emg_lp_grad_rows(const float* table, int64_t n_rows, int64_t ld, int32_t k_int, float lambda, int32_t p,
                     float* contrib, int64_t ldc, int32_t* dest, double* loss_accum, void* stream);

/* ---- K9: optional row-norm clip after a batch (EmbeddingModel.py:1371-1380, clip_by_norm axes=1) */
int emg_clip_rows(float* table, int64_t n_rows, int64_t ld, int32_t k_int, float max_norm, void* stream);
/* rows[j] -> table[ids[j]] for 0 <= ids[j] < n_rows (distinct ids; others skipped): the multi-GPU batch-sharded step with the
 * optimizer state sharded by OWNER writes the owners' updated rows into every replica with it (no reference counterpart: the
 * reference has no distributed code; the update itself is training/{sgd,momentum,adagrad}.py) */
int emg_scatter_rows(float* table, int64_t n_rows, int64_t ld, int32_t k_int, const float* rows, int64_t ldr,
                     const int32_t* ids, int64_t n, void* stream);
/* initializers/{glorot_uniform,uniform,normal}.py on the device: kind 0 = U[a, b), 1 = N(mean a, std b); element (r, c)
 * takes word (c & 3) of Philox4x32-10(counter (r * k_int + c) / 4, stream_id, seed), so a table is a pure function of
 * (seed, stream_id, shape) — the same on any number of GPUs.  The reference draws from TensorFlow's generators (cannot
 * be reproduced: parity-unpinned); Glorot-uniform is U(+-sqrt(6 / (rows + cols))) (glorot_uniform.py:59-74). */
int emg_init_table(int kind, float* table, int64_t n_rows, int64_t ld, int32_t k_int, float a, float b,
                   uint64_t seed, uint64_t stream_id, void* stream);

/* ====================== filtered 1-vs-all ranking (K10-K13) ======================
 * side_mode: 0 's' | 1 'o' | 2 's+o' | 3 's,o'.  Query rows: for modes 2,3 rows [0,n_q) are the
 * OBJECT-side queries (s,p,?) and rows [n_q,2n_q) the SUBJECT-side queries (?,p,o) — the block
 * order of generate_corruptions_for_eval (protocol.py:511-518); modes 0/1 have n_q rows. */
#define EMG_EVAL_S 0
#define EMG_EVAL_O 1
#define EMG_EVAL_SPO 2
#define EMG_EVAL_S_O 3

/* Build the query matrix Q[n_rows, ldq] and the positive's comparison integer
 * pos_int[n_rows] = int32(score_pos * 1e5) (EmbeddingModel.py:1865-1866,2010-2014), with the
 * positive scored through the SAME arithmetic as its corruptions (see DESIGN.md "canonical order").
 * Query vectors: DistMult q=p*x; ComplEx/HolE the Re/Im-hoisted vectors (SURVEY B-2);
 * TransE object side q=s+p, subject side q=o-p. */
int emg_eval_build_queries(int model, const float* ent, int64_t n_ent, int64_t ld_ent,
                           const float* rel, int64_t n_rel, int64_t ld_rel, int32_t k_int, float scale,
                           const int32_t* test_spo, int64_t n_q, int side_mode,
                           float* Q, int64_t ldq, int32_t* pos_int, void* stream);

/* Count, for every query row, the candidates whose comparison integer is > / == the positive's
 * (perform_comparision, EmbeddingModel.py:2010-2033; worst = gt+eq, best = gt,
 * middle = gt+ceil(eq/2) are formed by the caller).  Candidates are rows [0,n_cand) of `ent`
 * (an entity slab) or, if cand != NULL, rows cand[0..n_cand) of it.  cnt_gt/cnt_eq int32[n_rows]
 * are ACCUMULATED (zero them first; sum over slabs / GPUs).
 * precision: must be 0 = fp32 (exact f32-MFMA for DistMult/ComplEx/HolE; f32 VALU for TransE); any other value
 *            returns EMG_ENOSUP.  ent_bf16 / ld_bf16 are ignored (reserved): the bf16 MFMA throughput mode has its
 *            own entry points below (emg_eval_count_bf16 ...) because its operands are the bf16 copies. */
int emg_eval_count(int model, const float* Q, int64_t ldq, const int32_t* pos_int, int64_t n_rows,
                   const float* ent, int64_t n_cand, int64_t ld_ent, const int32_t* cand,
                   int32_t k_int, float scale, int precision, const void* ent_bf16, int64_t ld_bf16,
                   int32_t* cnt_gt, int32_t* cnt_eq, void* stream);

/* Same counts restricted to each query row's filter list (CSR: filt_ptr int64[n_rows+1],
 * filt_idx int32 GLOBAL entity ids; `ent` points at global row `ent_offset` and holds n_local rows;
 * entries outside [ent_offset, ent_offset+n_local) are skipped so every slab owner can be handed
 * the same global list) — replaces the tf.gather(scores, indices_obj/sub) + perform_comparision of
 * EmbeddingModel.py:1942-1963 and the SQLite lookups feeding it (sqlite_adapter.py:449-508).
 * fcnt_gt/fcnt_eq are ACCUMULATED. */
int emg_eval_filter_count(int model, const float* Q, int64_t ldq, const int32_t* pos_int, int64_t n_rows,
                          const float* ent, int64_t n_local, int64_t ld_ent, int64_t ent_offset,
                          int32_t k_int, float scale, int precision,
                          const int64_t* filt_ptr, const int32_t* filt_idx,
                          int32_t* fcnt_gt, int32_t* fcnt_eq, void* stream);

/* Debug/verification: dense scores S[n_rows, lds>=n_cand] through the SAME kernels as
 * emg_eval_count (small sizes only). */
int emg_eval_scores_dense(int model, const float* Q, int64_t ldq, int64_t n_rows,
                          const float* ent, int64_t n_cand, int64_t ld_ent, const int32_t* cand,
                          int32_t k_int, float scale, int precision, const void* ent_bf16, int64_t ld_bf16,
                          float* S, int64_t lds, void* stream);

/* f32 -> bf16 (round-to-nearest-even) copy of a table for precision mode 1 */
int emg_to_bf16(const float* src, int64_t n_rows, int64_t ld_src, int32_t k_int,
                void* dst_bf16, int64_t ld_dst, void* stream);

/* ---- bf16 MFMA variant of the 1-vs-all path (DistMult / ComplEx / HolE only; v_mfma_f32_32x32x16_bf16).
 * Throughput mode, NOT a parity mode: operands are the bf16 (RNE) copies made by emg_to_bf16; the kernels
 * multiply over k_pad = round_up(k_int, 16) elements (one MFMA k-step) and rows are stored zero-padded
 * with ld >= round_up(k_pad, 64).  emg_eval_pos_int_bf16 writes the true entity of each row (self_ent) and the
 * positive's comparison integer computed by the SAME MFMA arithmetic as the count pass, so the true entity
 * counts as exactly one tie (as in the f32 path); emg_eval_filter_count_bf16 counts a filter list's self entry
 * as that same tie by index, so the filtered rank's self-cancellation stays exact, and scores every other filter
 * entry through the same MFMA k-step sequence as the count pass (32 (row, entity) pairs on the diagonal of one
 * 32x32 product), so a filter entity is subtracted with exactly the comparison result it was counted with.
 * ent_offset: global id of row 0 of `ent_bf16` (slabs); cand: optional row indices. */
int emg_eval_pos_int_bf16(int model, const void* ent_bf16, int64_t ld_ent, int32_t k_int, float scale,
                          const int32_t* test_spo, int64_t n_q, int side_mode, const void* q_bf16, int64_t ldq,
                          int32_t* pos_int, int32_t* self_ent, void* stream);
int emg_eval_count_bf16(int model, const void* q_bf16, int64_t ldq, const int32_t* pos_int, const int32_t* self_ent,
                        int64_t n_rows, const void* ent_bf16, int64_t n_cand, int64_t ld_ent, const int32_t* cand,
                        int64_t ent_offset, int32_t k_pad, float scale, int32_t* cnt_gt, int32_t* cnt_eq,
                        int32_t need, void* stream);
/* need: 0 = both counters (cnt_gt += #(>), cnt_eq += #(==));  1 = cnt_gt += #(>=) — all the 'worst' strategy
 * (the reference's default) reads;  2 = cnt_gt += #(>) — all 'best' reads.  With need != 0 the register-stationary
 * kernel does one comparison per score instead of two and the content of cnt_eq is unspecified. */
int emg_eval_filter_count_bf16(int model, const void* q_bf16, int64_t ldq, const int32_t* pos_int,
                               const int32_t* self_ent, int64_t n_rows, const void* ent_bf16, int64_t n_local,
                               int64_t ld_ent, int64_t ent_offset, int32_t k_int, float scale,
                               const int64_t* filt_ptr, const int32_t* filt_idx, int32_t* fcnt_gt,
                               int32_t* fcnt_eq, void* stream);
/* f32 -> IEEE half (round-to-nearest-even) copy of a table / query matrix for precision mode 2; same layout rules
 * as emg_to_bf16 (rows zero-padded to ld_dst >= round_up(k_int, 64)) */
int emg_to_f16(const float* src, int64_t n_rows, int64_t ld_src, int32_t k_int, void* dst_f16, int64_t ld_dst, void* stream);
/* The prefilter's error band.  With q~ = q + dq, e~ = e + de the half-rounded operands:
 *   |MFMA(q~, e~) - chain(q, e)| <= ||dq|| max||e~|| + ||q|| max||de|| + g (||q~|| max||e~|| + ||q|| max||e||),  g = 2 (k + 32) 2^-24
 * (Cauchy-Schwarz on dq.e~ + q.de, plus the accumulation error of either sum).  emg_eval_prefilter_bounds writes
 * bounds3 = (max||e||, max||e~||, max||e~ - e||) over the rows of a table (slab) and its half copy — float64, DEVICE memory;
 * emg_eval_prefilter_band evaluates the bound per query row from the ACTUAL residual norms of this query tile, inflated
 * by 1e-6 and rounded up to float: the `band` input of emg_eval_prefilter_f16. */
int emg_eval_prefilter_bounds(const float* ent, int64_t n_rows, int64_t ld_ent, const void* ent_f16, int64_t ld_f16,
                              int32_t k_int, double* bounds3, void* stream);
int emg_eval_prefilter_band(const float* q, int64_t n_rows, int64_t ldq, const void* q_f16, int64_t ldq_f16,
                            int32_t k_int, const double* bounds3, float* band, void* stream);

/* ---- exact ranks at MFMA speed (precision mode 2): half-precision MFMA prefilter (v_mfma_f32_32x32x16_f16: 11
 * significant bits, an 8x narrower error band than bf16 at the same rate) + exact re-scoring of the undecided.
 * `band[r]` is a rigorous upper bound, for query row r and EVERY candidate, on |accumulator of the f16 kernel -
 * accumulator of the exact f32 chain| (the host derives it from the rounding residuals of the operands, see
 * emgraph_amd/evaluation/ranking.py::prefilter_band).  emg_eval_prefilter_f16 adds to cnt_gt the candidates that
 * beat the positive by more than the band, drops those that lose by more than it, and writes the rest as
 * (row << 32 | global entity id) into `pairs`: the buffer is cut into emg_eval_prefilter_segments(n_rows, n_cand)
 * segments of pairs_capacity / segments entries, one per wave of the kernel (no atomics); pair_count[s] = entries
 * written to segment s, pair_count[segments] != 0 if some wave ran out of room (then the caller must use
 * emg_eval_count for these rows instead).  pair_count (uint32 [segments + 1]) is zeroed by the call.
 * Round 6 (ABI unchanged): with at least 2048 entries per segment (64 per entity tile of the chunk) the kernel first writes each
 * segment's undecided candidates as a BITMAP into the segment itself (64 lanes x 8 bytes per tile, branch-free) and a second
 * launch inside the same call turns every segment's bitmap into its pairs in place, entity tiles ascending; a segment with more
 * undecided candidates than entries is then recorded with pair_count[s] = 0 (its words stay a bitmap) beside the overflow flag.
 * What the caller sees after the call is as before; smaller segments take the emitting kernel (EMG_PRE_BITMAP=0 forces it).
 * EMG_ENOSUP for
 * shapes the register-stationary kernel does not cover.  emg_eval_rescore_pairs scores the pairs with the parity
 * path's arithmetic (k-ordered fmaf chain, int32(score*1e5)) and adds them to cnt_gt / cnt_eq, reading the counts
 * on the device (no host round trip between the two calls).  The resulting counters equal
 * emg_eval_count(precision 0) bit for bit. */
int64_t emg_eval_prefilter_segments(int64_t n_rows, int64_t n_cand);   /* = emg_eval_prefilter_segments_k(.., 400) */
/* Above 400 contraction columns (ComplEx / HolE k > 200, DistMult k > 400; up to emg_eval_prefilter_max_cols() = 800) the
 * prefilter runs as 4 waves x 128 query rows per workgroup — one wave per SIMD, whose 512 registers hold up to 50 query
 * fragments — instead of 8 x 256: the segment count depends on the width, and emg_eval_prefilter_waves(k_cols) (8 or 4)
 * is the segments_per_block of the re-scoring call. */
int64_t emg_eval_prefilter_segments_k(int64_t n_rows, int64_t n_cand, int32_t k_cols);
int32_t emg_eval_prefilter_waves(int32_t k_cols);
int32_t emg_eval_prefilter_max_cols(void);
/* row stride (elements, zero padded) the half-precision prefilter wants of BOTH operands for a contraction over k_cols
 * columns: the kernel is instantiated for 4, 7, 8, 10, 13, 16, 19, 22, 25 (8 waves) and 32, 38, 44, 50 (4 waves) k-steps
 * of 16 and fetches entity rows 64 columns at a time; a width in between runs the next instantiation over the zero
 * padding (k_cols <= 800) */
int64_t emg_eval_prefilter_ld(int32_t k_cols);
int emg_eval_prefilter_f16(int model, const void* q_f16, int64_t ldq, const int32_t* pos_int, const float* band,
                           int64_t n_rows, const void* ent_f16, int64_t n_cand, int64_t ld_ent, int64_t ent_offset,
                           int32_t k_pad, float scale, int32_t* cnt_gt, uint64_t* pairs, uint32_t* pair_count,
                           int64_t pairs_capacity, void* stream);
/* The same pass PROVING TIES (ABI 8, round 6).  The reference compares int32(score * 1e5) (EmbeddingModel.py:2010-2014): on a table
 * whose scores are small against 1e-5 — a freshly initialised model, the first epochs of a fit — every candidate ties with the
 * positive, and emg_eval_prefilter_f16 calls every tie undecided (its accumulator lies between the `>` and the `>=` threshold).
 * Here a candidate whose accumulator lies inside the positive's integer cell by more than the band is counted into cnt_eq, one
 * above it into cnt_gt, and only the two bands around the cell's ends are written as pairs.  With a band wider than the cell
 * (scores of order one) the result is emg_eval_prefilter_f16's.  Eight instead of four VALU instructions per score: for tables the
 * plain form cannot decide.  Needs segments that hold the bitmap form (pairs_capacity / segments >= 2048 at 32 tiles per chunk:
 * EMG_EINVAL otherwise); cnt_gt / cnt_eq as emg_eval_rescore_pairs continues them. */
int emg_eval_prefilter_f16_ties(int model, const void* q_f16, int64_t ldq, const int32_t* pos_int, const float* band,
                                int64_t n_rows, const void* ent_f16, int64_t n_cand, int64_t ld_ent, int64_t ent_offset,
                                int32_t k_pad, float scale, int32_t* cnt_gt, int32_t* cnt_eq, uint64_t* pairs,
                                uint32_t* pair_count, int64_t pairs_capacity, void* stream);
int emg_eval_rescore_pairs(int model, const float* Q, int64_t ldq, const int32_t* pos_int, const float* ent,
                           int64_t ld_ent, int64_t ent_offset, int32_t k_int, float scale, const uint64_t* pairs,
                           int64_t pairs_capacity, const uint32_t* pair_count, int64_t n_segments, int32_t* cnt_gt,
                           int32_t* cnt_eq, void* stream);
/* the same with the number of segments each workgroup of the prefilter wrote (8: emg_eval_prefilter_f16[_thr], 4:
 * emg_eval_prefilter_sad and the plain call): the re-scoring then runs each workgroup's segments on the XCD that produced
 * them, in the same order, so the entity rows of a chunk are re-read from that XCD's L2 instead of HBM */
int emg_eval_rescore_pairs_ex(int model, const float* Q, int64_t ldq, const int32_t* pos_int, const float* ent,
                              int64_t ld_ent, int64_t ent_offset, int32_t k_int, float scale, const uint64_t* pairs,
                              int64_t pairs_capacity, const uint32_t* pair_count, int64_t n_segments,
                              int32_t segments_per_block, int32_t* cnt_gt, int32_t* cnt_eq, void* stream);
/* the same, told how many consecutive query rows a segment covers (32: the waves of emg_eval_prefilter_f16[_thr]; 0: not
 * known): segments of >= 512 pairs are then re-scored by one workgroup each with the segment's query rows held in LDS */
int emg_eval_rescore_pairs_rows(int model, const float* Q, int64_t ldq, const int32_t* pos_int, const float* ent,
                                int64_t ld_ent, int64_t ent_offset, int32_t k_int, float scale, const uint64_t* pairs,
                                int64_t pairs_capacity, const uint32_t* pair_count, int64_t n_segments,
                                int32_t segments_per_block, int32_t rows_per_segment, int32_t* cnt_gt, int32_t* cnt_eq,
                                void* stream);
/* ENTITY-TILE-MAJOR form of the re-scoring (round 5): the undecided pairs of all segments are bucketed by tile of 32 entity rows
 * (histogram | scan | scatter into `sorted`, capacity >= pairs_capacity), then one workgroup per tile holds the tile's f32 rows in
 * LDS and streams the query rows of its pairs — the query matrix of a call fits L2 / MALL where the entity table does not.  Same
 * chain and comparison as emg_eval_rescore_pairs; n_local = rows of `ent`; tile_ws: emg_eval_rescore_tiles_ws_bytes(n_local)
 * bytes, ZERO before its first use (the call leaves it zero).  EMG_ENOSUP for rows that are not 16-byte aligned or whose
 * 32-row image does not fit LDS: use emg_eval_rescore_pairs_rows.  Replaces the same reference lines as emg_eval_rescore_pairs
 * (EmbeddingModel.py:1856-1866, 2010-2033). */
int64_t emg_eval_rescore_tiles_ws_bytes(int64_t n_local);
int emg_eval_rescore_pairs_tiles(int model, const float* Q, int64_t ldq, const int32_t* pos_int, const float* ent,
                                 int64_t ld_ent, int64_t ent_offset, int64_t n_local, int32_t k_int, float scale,
                                 const uint64_t* pairs, int64_t pairs_capacity, const uint32_t* pair_count, int64_t n_segments,
                                 uint64_t* sorted, int64_t sorted_capacity, void* tile_ws, int64_t tile_ws_bytes,
                                 int32_t* cnt_gt, int32_t* cnt_eq, void* stream);
int emg_eval_scores_dense_bf16(int model, const void* q_bf16, int64_t ldq, int64_t n_rows, const void* ent_bf16,
                               int64_t n_cand, int64_t ld_ent, const int32_t* cand, int32_t k_pad, float scale,
                               float* S, int64_t lds, void* stream);

/* TransE-L2 through the same half-precision MFMA prefilter: ||q - e||^2 = |q|^2 - (2 q.e - |e|^2) is a contraction over
 * k_int + 2 coordinates, q' = [2q | -1 | -1], e' = [e | n_hi | n_lo], n_hi + n_lo ~ |e|^2 (TransE.py:208-216 with norm 2:
 * score = -sqrt(sum_k (q_k - e_k)^2), compared as int(score * 1e5), EmbeddingModel.py:2010-2033).
 *   emg_to_f16_l2            f32 rows -> half rows of ld_dst >= k_int + 2 columns.  is_query = 0: [e | n_hi | n_lo | 0..],
 *                            *n_residual_max (device double, optional, written by the call) = max |n_hi + n_lo - |e|^2|;
 *                            is_query = 1: [2q | -1 | -1 | 0..] and, if `doubled` is given, the f32 rows 2q (ld_src stride)
 *   emg_eval_l2_thresholds   per query row the two accumulator thresholds (thr float [2 * n_rows]) from pos_int, the
 *                            band of emg_eval_prefilter_band(2q, half(2q)) and bounds4 = the three norms of
 *                            emg_eval_prefilter_bounds followed by *n_residual_max
 *   emg_eval_prefilter_f16_thr  emg_eval_prefilter_f16 with the thresholds given instead of derived from pos_int / band:
 *                            cnt_gt[row] += #(acc >= thr[row]); pairs with thr[n_rows + row] <= acc < thr[row] are emitted
 * followed by emg_eval_rescore_pairs(EMG_TRANSE_L2, ...): the counters equal emg_eval_count's bit for bit. */
int emg_to_f16_l2(const float* src, int64_t n_rows, int64_t ld_src, int32_t k_int, int is_query, void* dst_f16,
                  int64_t ld_dst, float* doubled, double* n_residual_max, void* stream);
int emg_eval_l2_thresholds(const float* Q, int64_t n_rows, int64_t ldq, const int32_t* pos_int, const float* band,
                           const double* bounds4, int32_t k_int, float* thr, void* stream);
int emg_eval_prefilter_f16_thr(const void* q_f16, int64_t ldq, const float* thr, int64_t n_rows, const void* ent_f16,
                               int64_t n_cand, int64_t ld_ent, int64_t ent_offset, int32_t k_pad, int32_t* cnt_gt,
                               uint64_t* pairs, uint32_t* pair_count, int64_t pairs_capacity, void* stream);

/* TransE-L1: the ranks of emg_eval_count, bit for bit, at integer speed (csrc/emg_rank_sad.hip).  The score of the
 * reference, -sum_k |q_k - e_k| (TransE.py:208-216 with norm 1, compared as int(score * 1e5), EmbeddingModel.py:
 * 2010-2033), is bounded from both sides by a sum of absolute differences of 16-bit fixed-point images of the rows
 * (v_sad_u16: two coordinates per instruction); candidates the bound decides are counted, the rest re-scored by
 * emg_eval_rescore_pairs (which takes TransE model ids for this purpose).
 *   emg_eval_sad_range      range[0] = max|ent|, range[1] = max|rel| (device doubles, written by the call): the image
 *                           maps [-R, R], R = (range[0] + range[1])(1 + 1e-6), onto 0..65535 — every query row
 *                           emg_eval_build_queries makes from these tables (s+p, o-p) lies inside
 *   emg_eval_sad_ld         columns of an image row for k_int coordinates (k_int rounded up to 16)
 *   emg_eval_sad_quantize   f32 rows -> u16 image rows (ld_dst u16 columns, padding columns 0)
 *   emg_eval_sad_thresholds per query row the integer sums below which a candidate certainly compares greater (lo)
 *                           and above which certainly less (hi) than pos_int
 *   emg_eval_prefilter_sad  cnt_gt[row] += #(candidates with sum < lo); (row << 32 | ent_offset + column) of the
 *                           candidates with lo <= sum <= hi into `pairs`, cut into emg_eval_sad_segments(n_rows,
 *                           n_cand) segments exactly as emg_eval_prefilter_f16 does (pair_count zeroed by the call;
 *                           pair_count[segments] != 0: some wave ran out of room, use emg_eval_count for these rows) */
int emg_eval_sad_range(const float* ent, int64_t n_ent, int64_t ld_ent, const float* rel, int64_t n_rel, int64_t ld_rel,
                       int32_t k_int, double* range, void* stream);
int64_t emg_eval_sad_ld(int32_t k_int);
int emg_eval_sad_quantize(const float* src, int64_t n_rows, int64_t ld_src, int32_t k_int, const double* range,
                          void* dst_u16, int64_t ld_dst, void* stream);
int emg_eval_sad_thresholds(const int32_t* pos_int, int64_t n_rows, int32_t k_int, const double* range, uint32_t* lo,
                            uint32_t* hi, void* stream);
int64_t emg_eval_sad_segments(int64_t n_rows, int64_t n_cand);
int emg_eval_prefilter_sad(const void* q_u16, int64_t ldq, const uint32_t* lo, const uint32_t* hi, int64_t n_rows,
                           const void* ent_u16, int64_t n_cand, int64_t ld_ent, int64_t ent_offset, int32_t k_int,
                           int32_t* cnt_gt, uint64_t* pairs, uint32_t* pair_count, int64_t pairs_capacity, void* stream);

/* ====================== one-call forms (compositions of the entry points above, one stream) ==============
 * The C-ABI sketched in SURVEY.md 8b.  A maintainer binding the library from the reference calls these once per
 * batch / per test set; the finer-grained entry points exist so that a host can overlap stages on several
 * streams (as emgraph_amd/training.py does). */

/* generate_corruptions_for_fit (protocol.py:531-659) in one call: out_spo int32 [B*eta, 3], eta-major.
 * entities_size > 0: replacements are ids drawn from [0, entities_size) (:616-619); otherwise they are drawn
 * from entities_list[0..n_list) (:620-641).  Draws: Philox4x32-10 keyed by (seed; row, counter). */
int emg_corrupt_fit(const int32_t* pos, int64_t B, int32_t eta, int side, int64_t entities_size,
                    const int32_t* entities_list, int64_t n_list, uint64_t seed, uint64_t counter,
                    int32_t* out_spo, void* stream);

/* Filtered ranks of n_q test triples against all entities (cand == NULL) or the rows cand[0..n_cand):
 * emg_eval_build_queries -> emg_eval_count[_bf16] -> emg_eval_filter_count[_bf16] -> rank assembly
 * (EmbeddingModel.py:1845-2033).  filt_ptr/filt_idx: optional CSR over the n_rows query rows in the row order
 * documented above (object-side rows first), GLOBAL entity ids, each list containing the row's true entity.
 * strategy: 0 worst | 1 best | 2 middle.  precision_mode: 0 exact f32 | 1 bf16 MFMA (statistical agreement) |
 * 2 the ranks of mode 0, bit for bit, through the half-precision MFMA prefilter + exact re-scoring (emg_to_f16,
 * emg_eval_prefilter_bounds / _band, emg_eval_prefilter_f16, emg_eval_rescore_pairs; one host synchronisation to read the
 * overflow flag; TransE-L1 goes through the 16-bit fixed-point prefilter emg_eval_prefilter_sad instead, TransE-L2
 * through the MFMA prefilter on the augmented rows of emg_to_f16_l2; the exact kernel takes over for candidate lists,
 * shapes the prefilter kernels do not cover and overflowing pair buffers — the pair buffer of a call is capped at
 * 1 GiB, so hand mode 2 a few thousand test triples per call (emgraph_amd/evaluation/ranking.py uses 4096): a much larger
 * call still returns the exact ranks, but through the exact kernel).
 * rank_out int32: [n_q] for side_mode 0,1,2; [n_q,2] = [subject_rank, object_rank] for side_mode 3. */
int emg_rank_1vsall(int model, const float* ent, int64_t n_ent, int64_t ld_ent, const float* rel, int64_t n_rel,
                    int64_t ld_rel, int32_t k_int, float scale, const int32_t* test_spo, int64_t n_q, int side_mode,
                    const int32_t* cand, int64_t n_cand, const int64_t* filt_ptr, const int32_t* filt_idx,
                    int strategy, int precision_mode, int32_t* rank_out, void* stream);

/* One training batch: corruptions of every side -> scores -> loss (accumulated into *loss_accum) -> gradients ->
 * row-sparse optimizer update of both tables (EmbeddingModel.py:614-822 + training/{sgd,momentum,adagrad,adam}.py), without the LP
 * regulariser.  `workspace` (device, emg_train_step_workspace_bytes) holds all per-batch scratch.
 * inplace != 0: rows whose destination occurs once in the batch are updated by the gradient kernel itself. */
typedef struct emg_step_args {
    int32_t model; int32_t k_int; float scale; int32_t eta; int32_t n_sides; int32_t sides[4];
    float* ent; int64_t n_ent; int64_t ld_ent; float* rel; int64_t n_rel; int64_t ld_rel;
    float* ent_state0; float* ent_state1; float* rel_state0; float* rel_state1;   /* optimizer state, NULL if unused */
    int32_t* tag_ent; int32_t* tag_rel;                                           /* int32[n_rows], zeroed once */
    int32_t opt; int32_t step; float hyper[8];                                    /* as emg_apply_rows (hyper[6] = 0) */
    const int32_t* pos; int64_t B;                                                /* device int32 [B,3] */
    int64_t n_choices; const int32_t* entities_list; uint64_t seed; uint64_t draw_counter0;
    const int32_t* inj_mask; const int32_t* inj_repl;                             /* optional injected draws */
    int32_t loss; float margin; float alpha; double* loss_accum;
    int32_t inplace;
    void* workspace; int64_t workspace_bytes;
} emg_step_args;
int64_t emg_train_step_workspace_bytes(int64_t B, int32_t eta_total, int32_t k_int, int64_t n_ent, int64_t n_rel);
int emg_train_step(const emg_step_args* args, void* stream);

/* ====================== the training step as one call on several streams (emg_plan.hip) ======================
 * emg_train_step above is a single-stream composition.  A plan object additionally owns two high-priority side
 * streams (batch preparation of the NEXT batches runs there while the current one computes), a stream for the
 * relation table's apply, and the events between them, so that the multi-stream step emgraph_amd's fit() runs is
 * ONE library call per batch (EmbeddingModel.py:1388-1440 is the loop it stands for).  All buffers are the caller's;
 * the plan owns streams and events only.  Scratch is sized for cap_B positives per batch:
 *   scores, g : float [(1 + eta_total) * cap_B] each (only the unfused step — losses that couple a positive's
 *               negatives — uses them);  contrib_ent [(2 + eta_total) * cap_B, ldc], contrib_rel [cap_B, ldc];
 *   per slot  : codes int32 [eta_total * cap_B], dest_ent int32 [(2 + eta_total) * cap_B], dest_rel int32 [cap_B],
 *               single uint8 [(2 + eta_total) * cap_B], ws_* from emg_apply_workspace_bytes_ex.
 * n_slots = 1 + number of batches prepared ahead (0..2; 1 slot = everything on the caller's stream). */
typedef struct emg_plan_slot {
    int32_t* codes; int32_t* dest_ent; int32_t* dest_rel; uint8_t* single;
    void* ws_ent; int64_t ws_ent_bytes; void* ws_rel; int64_t ws_rel_bytes;
} emg_plan_slot;
typedef struct emg_plan_config {
    int32_t model; int32_t k_int; float scale; int32_t eta; int32_t n_sides; int32_t sides[4];
    float* ent; int64_t n_ent; int64_t ld_ent; float* rel; int64_t n_rel; int64_t ld_rel;
    float* ent_state0; float* ent_state1; float* rel_state0; float* rel_state1; int32_t* tag_ent; int32_t* tag_rel;
    int32_t opt; int32_t loss; float margin; float alpha;
    uint64_t seed; int64_t batches_count;            /* draw counter of (epoch, batch, side): see emg_prepare_args */
    const int32_t* X; int64_t n_triples;             /* the resident, id-mapped training set [n_triples, 3] */
    int64_t cap_B;
    float* scores; float* g; float* contrib_ent; float* contrib_rel; int64_t ldc;
    double* loss_accum; double* lp_sum;              /* lp_sum[2]: sum |w|^p of the entity / relation table */
    int32_t factored;                                /* 1: factored entity contributions (emg_prepare_args / emg_backward_args); contrib_ent then needs 4 * cap_B rows */
    int32_t loss_slots;                              /* loss_accum points to this many doubles (emg_backward_args.loss_slots; 0 / 1: one) */
    float lp_lambda_ent; float lp_lambda_rel; int32_t lp_p;   /* folded LP regulariser (0 = none; excludes inplace) */
    int32_t fused; int32_t inplace; int32_t normalize;   /* inplace: 0 off, 1 singletons in place, 2 the same through a stateful optimizer's
                                                            window form (emg_backward_args.inplace_window; with lr_t_hist: Adam's singleton
                                                            negatives replayed inside the scoring kernel, s / o slots through the apply) */
    int32_t n_slots; emg_plan_slot slots[4];
    int64_t aux_min_rows;                            /* entity contribution rows above which apply_rel gets its stream */
    void* ctl_buf; int64_t ctl_bytes;                /* optional device scratch (>= 32 * sizeof(emg_step_ctl)) for emg_plan_run */
    const float* lr_t_hist;                          /* != NULL (EMG_OPT_ADAM and / or LP): deferred dense pass (emg_deferred_catchup before every
                                                        scoring kernel, none afterwards); [s] = learning rate of step s (Adam: lr_t), filled
                                                        for every step the plan is given */
} emg_plan_config;
typedef struct emg_plan_batch {
    int64_t start; int64_t B; int32_t epoch; int32_t batch;   /* rows [start, start + B) of X; 1-based epoch / batch */
    int64_t n_choices; const int32_t* entities_list;           /* corruption pool (0 / NULL: all n_ent entities) */
    const int32_t* inj_mask; const int32_t* inj_repl;          /* optional injected draws */
} emg_plan_batch;
int emg_plan_create(const emg_plan_config* cfg, void** plan);
/* train on `cur`; `next[0..n_next)` (nearest first) are prepared ahead on the side streams if a slot is free.
 * step >= 1 is the optimizer step number, hyper6 = {lr, momentum, beta1, beta2, eps, lr_t} for it. */
int emg_plan_step(void* plan, const emg_plan_batch* cur, int32_t step, const float* hyper6,
                  const emg_plan_batch* next, int32_t n_next, void* stream);
/* The same steps as GRAPH replays (small batches: a step is shorter than its launches take to issue).  The step's
 * kernels are captured once per graph length (<= 32 steps) with their per-step values read from device records
 * (emg_step_ctl) that each call writes before the replay: two launches per 32 steps from the host.  batches[i] trains as
 * optimizer step first_step + i with hyper6s[6 i .. 6 i + 5]; results are those of n emg_plan_step calls.
 * emg_plan_graph_ok: 1 if the plan can (fused pair-local loss, 16-byte rows of more than 16 chunks, counting grouping,
 * ctl_buf given); injected draws need emg_plan_step. */
int emg_plan_graph_ok(void* plan);
/* 1 if a plan of this shape can DEFER its dense pass (emg_plan_config.lr_t_hist): emg_deferred_catchup walks the segment
 * descriptors of the counting grouping, which is chosen only while a table is not much longer than its batch has gradient
 * rows (n_rows <= 16 n + 2^20; EMG_GROUPING=sort forces the radix-sort backend).  emg_plan_create refuses lr_t_hist otherwise:
 * the caller keeps the dense pass (emgraph_amd/training.py::Trainer._make_plan). */
int emg_plan_deferred_ok(int64_t cap_B, int32_t eta_total, int64_t n_ent, int64_t n_rel);
int emg_plan_run(void* plan, const emg_plan_batch* batches, int32_t n, int32_t first_step, const float* hyper6s,
                 void* stream);
/* HIP-event timing of the next max_samples launches of every stage (0 = off); avg_ms / counts: 9 entries =
 * prepare, fused, forward, loss, backward, apply_ent, apply_rel, clip, catchup (the deferred pass's emg_deferred_catchup calls) */
int emg_plan_timing(void* plan, int32_t max_samples);
int emg_plan_stage_ms(void* plan, float* avg_ms, int32_t* counts);
int emg_plan_destroy(void* plan);

#ifdef __cplusplus
}
#endif
#endif /* EMGRAPH_HIP_H */
