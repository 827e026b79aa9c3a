// emg_plan.hip — the per-batch training step as ONE library call on several HIP streams.
//
// What EmbeddingModel.fit's inner loop (EmbeddingModel.py:1388-1440: tf.data batch -> _get_model_loss ->
// optimizer.minimize) costs per batch here is a dozen kernel launches on four streams:
//
//   side streams (2, high priority, alternating): everything about batches t+1, t+2 that does not depend on the
//       tables — Philox corruption codes, destination ids, their stable grouping, singleton flags (emg_prepare_batch)
//   main stream:  fused gather + score + loss + gradient kernel (in-place singleton updates)  ->  entity apply
//   aux stream :  relation apply, underneath the entity apply
//
// Round 1 issued this plan from Python (ctypes call + torch stream / event objects per launch): 0.14-0.24 ms of host
// time per step, more than the GPU needs for the small configurations (C1: B = 1725, C2: B = 2722).  Here the whole
// step is enqueued by one call: the slots, events and streams live in the plan object, the host cost is the launches.
//
// The plan owns streams and events only; every buffer is the caller's (emg_plan_config), as everywhere in this ABI.
// Not covered: the k-sharded multi-GPU step (it needs a collective between forward and loss: host-driven).
#include <functional>
#include <vector>

#include <stdlib.h>
#include <string.h>

#include "emg_group_kernels.hpp"

namespace emg {

int train_backward_impl(const emg_backward_args* a, const Riders* riders, void* stream);                                  // emg_score.hip
int apply_pair_impl(const emg_apply_args* a, const emg_apply_args* b, const Riders* riders, void* stream);   // emg_apply.hip

enum Stage { ST_PREPARE = 0, ST_FUSED, ST_FORWARD, ST_LOSS, ST_BACKWARD, ST_APPLY_ENT, ST_APPLY_REL, ST_CLIP, ST_CATCHUP, ST_COUNT };

struct SlotState {
    emg_plan_slot buf;
    hipEvent_t ready = nullptr, done = nullptr;
    bool has_key = false, ready_recorded = false;
    bool done_in_capture = false;   // graph capture: `done` was recorded inside this capture (only then may a captured wait name it)
    int64_t key[4] = {0, 0, 0, 0};
};

struct Plan {
    emg_plan_config cfg;
    hipStream_t side[2] = {nullptr, nullptr};
    hipStream_t aux = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    hipEvent_t scored = nullptr;     // recorded after a step's scoring launches: look-ahead preparation starts behind it
    bool wait_scored = false;        // the next prepare() waits for `scored` first
    SlotState slots[4];
    int n_side = 0, side_rr = 0;
    int timing_max = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[ST_COUNT];
    // captured step graphs (emg_plan_run), by length in steps
    std::vector<std::pair<int, hipGraphExec_t>> graphs;
    const StepCtl* ctl = nullptr;   // capture in progress: the record of the step being captured
    hipStream_t cap = nullptr;              // origin stream of the captures (the caller's may be the null stream, which cannot capture)
    bool side_joined[2] = {false, false};   // capture in progress: the side stream has been forked into the capture
    hipEvent_t side_join[2] = {nullptr, nullptr};
};

constexpr int kGraphSteps = 32;     // steps per graph replay = records per ctl_write_kernel launch (kernel arguments: 4 KB)
struct CtlBlock { StepCtl rec[kGraphSteps]; };
__global__ void ctl_write_kernel(const CtlBlock blk, StepCtl* __restrict__ dst, int n) {
    // (the records travel as kernel arguments: copied at launch, so the host buffer may be reused at once)
    const int words = n * (int)(sizeof(StepCtl) / 4);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&blk);
    for (int i = threadIdx.x; i < words; i += blockDim.x) reinterpret_cast<uint32_t*>(dst)[i] = src[i];
}

static bool same_key(const SlotState& s, const emg_plan_batch& b) {
    return s.has_key && s.key[0] == b.start && s.key[1] == b.B && s.key[2] == b.epoch && s.key[3] == b.batch;
}

struct Timed {  // HIP events around one stage (only the first `timing_max` launches of each stage are sampled)
    Plan* p; int st; hipStream_t s; hipEvent_t e0 = nullptr, e1 = nullptr;
    Timed(Plan* p_, int st_, hipStream_t s_) : p(p_), st(st_), s(s_) {
        if (p->timing_max > 0 && (int)p->ev[st].size() < p->timing_max) {
            if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) (void)hipEventRecord(e0, s);
            else e0 = e1 = nullptr;
        }
    }
    ~Timed() {
        if (e0 && e1) { (void)hipEventRecord(e1, s); p->ev[st].push_back({e0, e1}); }
    }
};

// emg_prepare_batch's arguments for a batch in a slot (ctl: the batch comes from that device record instead of b)
static void fill_prepare_args(const Plan* P, const SlotState& sl, const emg_plan_batch& b, const StepCtl* ctl, emg_prepare_args& a) {
    const emg_plan_config& c = P->cfg;
    a = emg_prepare_args{};
    a.pos = c.X + 3 * b.start; a.B = b.B; a.eta = c.eta; a.n_sides = c.n_sides;
    for (int i = 0; i < c.n_sides; ++i) a.sides[i] = c.sides[i];
    a.n_choices = b.n_choices > 0 ? b.n_choices : c.n_ent; a.entities_list = b.entities_list;
    a.seed = c.seed;
    a.draw_counter0 = (uint64_t)(((int64_t)(b.epoch - 1) * c.batches_count + (b.batch - 1)) * c.n_sides);
    a.inj_mask = b.inj_mask; a.inj_repl = b.inj_repl;
    a.codes = sl.buf.codes; a.dest_ent = sl.buf.dest_ent; a.n_ent = c.n_ent; a.dest_rel = sl.buf.dest_rel; a.n_rel = c.n_rel;
    a.ws_ent = sl.buf.ws_ent; a.ws_ent_bytes = sl.buf.ws_ent_bytes; a.ws_rel = sl.buf.ws_rel; a.ws_rel_bytes = sl.buf.ws_rel_bytes;
    a.single_flags = c.inplace ? sl.buf.single : nullptr;
    a.factored = c.factored;
    a.ws_clean = 1;            // emg_plan_create zeroed the control regions; every grouping leaves them zero
    a.layout_B = c.cap_B;      // one workspace layout for every batch size of the run
    if (ctl) {              // graph capture: rows, size and draw counter come from the device record
        a.pos = c.X; a.B = c.cap_B; a.ctl = ctl;
        a.n_choices = c.n_ent; a.entities_list = nullptr; a.inj_mask = a.inj_repl = nullptr;
    }
}

static int prepare(Plan* P, SlotState& sl, const emg_plan_batch& b, hipStream_t main) {
    hipStream_t st = main;
    if (P->n_side > 0) {
        const int si = P->side_rr;
        st = P->side[si];
        P->side_rr = (P->side_rr + 1) % P->n_side;
        if (P->ctl) {   // graph capture: a side stream joins the capture by waiting for an event recorded inside it
            if (!P->side_joined[si]) { EMG_HIP(hipStreamWaitEvent(st, P->fork, 0)); P->side_joined[si] = true; }
            if (sl.done_in_capture) EMG_HIP(hipStreamWaitEvent(st, sl.done, 0));
        } else {
            if (P->wait_scored) EMG_HIP(hipStreamWaitEvent(st, P->scored, 0));      // beside the applies, not beside the scoring kernel
            EMG_HIP(hipStreamWaitEvent(st, sl.done, 0));                           // the compute that last used this slot
            if (sl.has_key && sl.ready_recorded) EMG_HIP(hipStreamWaitEvent(st, sl.ready, 0));  // evicted, never consumed
        }
    }
    emg_prepare_args a{};
    fill_prepare_args(P, sl, b, P->ctl, a);
    int rc;
    {
        Timed t(P, ST_PREPARE, st);
        rc = emg_prepare_batch(&a, st);
    }
    if (rc != EMG_OK) return rc;
    if (P->n_side > 0) { EMG_HIP(hipEventRecord(sl.ready, st)); sl.ready_recorded = true; }
    sl.has_key = true;
    sl.key[0] = b.start; sl.key[1] = b.B; sl.key[2] = b.epoch; sl.key[3] = b.batch;
    return EMG_OK;
}

// after_scoring: called between the scoring launches and the applies (emg_plan_step enqueues the look-ahead preparation there)
static int compute(Plan* P, SlotState& sl, const emg_plan_batch& b, int32_t step, const float* hyper6, hipStream_t main,
                   const Riders* ride_a = nullptr, const Riders* ride_b = nullptr,
                   const std::function<int()>* after_scoring = nullptr) {
    const emg_plan_config& c = P->cfg;
    const int32_t et = c.eta * c.n_sides;
    const int64_t B = b.B, n_ce = (2 + et) * B;
    const bool lp = c.lp_lambda_ent != 0.f || c.lp_lambda_rel != 0.f;
    float he[8], hr[8];
    for (int i = 0; i < 6; ++i) he[i] = hr[i] = hyper6[i];
    he[6] = c.lp_lambda_ent; hr[6] = c.lp_lambda_rel; he[7] = hr[7] = (float)c.lp_p;
    const int32_t* pos = c.X + 3 * b.start;

    emg_backward_args ba{};
    ba.model = c.model; ba.k_int = c.k_int; ba.scale = c.scale; ba.eta = et;
    ba.ent = c.ent; ba.n_ent = c.n_ent; ba.ld_ent = c.ld_ent; ba.rel = c.rel; ba.n_rel = c.n_rel; ba.ld_rel = c.ld_rel;
    ba.pos = pos; ba.B = B; ba.codes = sl.buf.codes; ba.margin = c.margin; ba.loss_accum = c.loss_accum; ba.loss_slots = c.loss_slots;
    ba.contrib_ent = c.contrib_ent; ba.contrib_rel = c.contrib_rel; ba.ldc = c.ldc;
    ba.single_ent = c.inplace ? sl.buf.single : nullptr; ba.opt = c.opt; ba.step = step;
    ba.inplace_window = c.inplace == 2 ? 1 : 0;   // (a stateful optimizer's window form)
    for (int i = 0; i < 6; ++i) ba.hyper[i] = hyper6[i];
    if (c.inplace && c.lp_lambda_ent != 0.f) { ba.hyper[6] = he[6]; ba.hyper[7] = he[7]; ba.lp_accum = c.lp_sum; }   // (plain SGD: checked at creation)
    ba.ent_state0 = c.ent_state0; ba.ent_state1 = c.ent_state1; ba.tag_ent = c.tag_ent;
    if (c.factored) { ba.fac_ws_ent = sl.buf.ws_ent; ba.fac_ws_ent_bytes = sl.buf.ws_ent_bytes; }
    ba.layout_B = c.cap_B;
    if (P->ctl) { ba.pos = c.X; ba.B = c.cap_B; ba.ctl = P->ctl; }
    int rc;
    // Adam without regulariser / in-place singletons: the catch-up writes w alone, the apply redoes the decay of m, v (emgraph_hip.h)
    static const bool lag_env = [] { const char* e = getenv("EMG_DEFERRED_W_ONLY"); return !(e && e[0] == '0'); }();
    // (only where the descriptor-driven apply runs — 16-byte rows of more than 16 chunks —: it is the one that redoes the decay)
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const bool seg_rows = c.k_int % 4 == 0 && c.k_int > 64 && c.ld_ent % 4 == 0 && c.ld_rel % 4 == 0 && c.ldc % 4 == 0 && al16(c.ent) &&
                          al16(c.rel) && al16(c.contrib_ent) && al16(c.contrib_rel) && al16(c.ent_state0) && al16(c.ent_state1) &&
                          al16(c.rel_state0) && al16(c.rel_state1);
    // Adam in place under the deferred pass (round 4): the singletons among the NEGATIVES lag — the scoring kernel fetches (w, m, v)
    // of such a row together, replays its missed steps in registers and updates it (emg_backward_args.lr_hist); the catch-up and
    // the apply handle every other destination of the batch (s / o slots, rows hit more than once)
    const bool lag_ip = c.lr_t_hist && c.inplace == 2 && c.opt == EMG_OPT_ADAM;
    // SGD + LP in place under the deferred pass (round 5, form 7): every singleton's row is replayed by the scoring kernel as it gathers
    // it; the catch-up walks the rows hit more than once
    const int n_cols = (c.model == EMG_COMPLEX || c.model == EMG_HOLE) ? c.k_int / 2 : c.k_int;
    static const bool lp_ip_env = [] { const char* e = getenv("EMG_LP_REPLAY"); return !(e && e[0] == '0'); }();   // A/B aid
    const bool lp_ip = c.lr_t_hist && c.inplace == 1 && c.opt == EMG_OPT_SGD && c.lp_lambda_ent != 0.f && c.lp_p <= 3 && c.fused && !P->ctl &&
                       n_cols % 4 == 0 && n_cols / 4 <= 128 && c.ld_ent % 4 == 0 && al16(c.ent) && lp_ip_env;
    const int32_t w_only = (c.lr_t_hist && c.opt == EMG_OPT_ADAM && !lp && (!c.inplace || lag_ip) && seg_rows && lag_env) ? 1 : 0;
    if (lag_ip || lp_ip) ba.lr_hist = c.lr_t_hist;
    if (c.lr_t_hist) {   // deferred dense pass (Keras Adam / LP): bring the rows this batch reads and updates up to step - 1
        Timed t(P, ST_CATCHUP, main);
        rc = emg_deferred_catchup(c.opt, c.ent, c.n_ent, c.ld_ent, c.k_int, c.ent_state0, c.ent_state1, c.tag_ent, he, c.lr_t_hist, step - 1,
                                  lp ? c.lp_sum : nullptr, sl.buf.ws_ent, sl.buf.ws_ent_bytes, (2 + (int64_t)et) * c.cap_B, w_only,
                                  (lag_ip || lp_ip) ? 0 : -1, main);   // (the scoring kernel replays every singleton itself)
        if (rc != EMG_OK) return rc;
        rc = emg_deferred_catchup(c.opt, c.rel, c.n_rel, c.ld_rel, c.k_int, c.rel_state0, c.rel_state1, c.tag_rel, hr, c.lr_t_hist, step - 1,
                                  lp ? c.lp_sum + 1 : nullptr, sl.buf.ws_rel, sl.buf.ws_rel_bytes, c.cap_B, w_only, -1, main);
        if (rc != EMG_OK) return rc;
    }
    if (c.fused) {
        ba.fused_loss = c.loss;
        Timed t(P, ST_FUSED, main);
        rc = train_backward_impl(&ba, ride_a, main);
        if (rc != EMG_OK) return rc;
    } else {
        float* sp = c.scores; float* sn = sp + B;
        float* gp = c.g; float* gn = gp + B;
        {
            Timed t(P, ST_FORWARD, main);
            rc = emg_train_forward(c.model, c.ent, c.n_ent, c.ld_ent, c.rel, c.n_rel, c.ld_rel, c.k_int, c.scale, pos, B, et,
                                   sl.buf.codes, EMG_SCORE_FINAL, sp, sn, main);
            if (rc != EMG_OK) return rc;
        }
        {
            Timed t(P, ST_LOSS, main);
            rc = emg_loss(c.loss, sp, sn, B, c.eta, c.n_sides, c.margin, c.alpha, c.loss_accum, gp, gn, main);
            if (rc != EMG_OK) return rc;
        }
        ba.fused_loss = -1; ba.g_pos = gp; ba.g_neg = gn;
        // rows wider than the register-tiled kernel run as column blocks: TransE-L2's gradient then needs the full norms
        if (c.model == EMG_TRANSE_L2 && c.k_int > 512) { ba.bw_scores_pos = sp; ba.bw_scores_neg = sn; }
        Timed t(P, ST_BACKWARD, main);
        rc = train_backward_impl(&ba, ride_a, main);
        if (rc != EMG_OK) return rc;
    }
    if (after_scoring) {
        rc = (*after_scoring)();
        if (rc != EMG_OK) return rc;
    }
    // The two tables' applies are independent: ONE pair of launches over both groupings
    // (emg_apply_grouped_pair).  On separate streams the relation apply ran underneath the entity apply but slowed it
    // down by as much as it saved (C3: entity apply alone 0.10 ms, beside the relation apply 0.12; relation apply alone
    // 0.047 ms, nearly all of it launch + window preamble) — the aux stream remains as the EMG_PAIR_APPLY=0 A/B path.
    auto fill = [&](emg_apply_args& aa, bool ent_table) {
        aa = emg_apply_args{};
        aa.opt = c.opt; aa.k_int = c.k_int; aa.step = step;
        const float* h = ent_table ? he : hr;
        for (int i = 0; i < 8; ++i) aa.hyper[i] = h[i];
        if (ent_table) {
            aa.table = c.ent; aa.n_rows = c.n_ent; aa.ld = c.ld_ent; aa.state0 = c.ent_state0; aa.state1 = c.ent_state1;
            aa.tag = c.tag_ent; aa.skip_single = c.inplace ? 1 : 0;
            aa.contrib = c.contrib_ent; aa.n_contrib = n_ce;
            aa.lp_accum = lp ? c.lp_sum : nullptr; aa.workspace = sl.buf.ws_ent; aa.workspace_bytes = sl.buf.ws_ent_bytes;
            aa.factored = c.factored; aa.layout_n = (2 + (int64_t)et) * c.cap_B; aa.table_index = 0;
        } else {
            aa.table = c.rel; aa.n_rows = c.n_rel; aa.ld = c.ld_rel; aa.state0 = c.rel_state0; aa.state1 = c.rel_state1;
            aa.tag = c.tag_rel; aa.skip_single = 0; aa.contrib = c.contrib_rel; aa.n_contrib = B;
            aa.lp_accum = lp ? c.lp_sum + 1 : nullptr; aa.workspace = sl.buf.ws_rel; aa.workspace_bytes = sl.buf.ws_rel_bytes;
            aa.layout_n = c.cap_B; aa.table_index = 1;
        }
        if (P->ctl) { aa.ctl = P->ctl; aa.n_contrib = aa.layout_n; }
        aa.deferred_dense = c.lr_t_hist ? (w_only ? 2 : 1) : 0;
        aa.ldc = c.ldc;
    };
    emg_apply_args ae, ar;
    fill(ae, true);
    fill(ar, false);
    static const bool pair_env = getenv("EMG_PAIR_APPLY") == nullptr || atoi(getenv("EMG_PAIR_APPLY")) != 0;
    const bool pair = pair_env || P->ctl != nullptr;   // (a captured step never forks for the relation apply)
    const bool big = n_ce >= c.aux_min_rows;
    if (pair) {   // (any batch size: C1 0.134 -> 0.126 ms/step, C2 0.100 -> 0.089 against two launch pairs in sequence)
        Timed t(P, ST_APPLY_ENT, main);
        rc = apply_pair_impl(&ae, &ar, ride_b, main);
        if (rc != EMG_OK) return rc;
    } else {
        const bool use_aux = P->aux != nullptr && big;
        hipStream_t rst = main;
        if (use_aux) {
            EMG_HIP(hipEventRecord(P->fork, main));
            EMG_HIP(hipStreamWaitEvent(P->aux, P->fork, 0));
            rst = P->aux;
        }
        auto apply_rel = [&]() {
            Timed t(P, ST_APPLY_REL, rst);
            return emg_apply_grouped_ex(&ar, rst);
        };
        if (use_aux) {
            rc = apply_rel();
            if (rc != EMG_OK) return rc;
            EMG_HIP(hipEventRecord(P->join, P->aux));
        }
        {
            Timed t(P, ST_APPLY_ENT, main);
            rc = emg_apply_grouped_ex(&ae, main);
            if (rc != EMG_OK) return rc;
        }
        if (use_aux) EMG_HIP(hipStreamWaitEvent(main, P->join, 0));
        else {
            rc = apply_rel();
            if (rc != EMG_OK) return rc;
        }
    }
    if (c.normalize) {  // EmbeddingModel.py:1434-1440: tf.clip_by_norm(ent_emb, clip_norm=1, axes=1) after each batch
        Timed t(P, ST_CLIP, main);
        rc = emg_clip_rows(c.ent, c.n_ent, c.ld_ent, c.k_int, 1.0f, main);
        if (rc != EMG_OK) return rc;
    }
    return EMG_OK;
}

}  // namespace emg

using namespace emg;

extern "C" int emg_plan_create(const emg_plan_config* cfg, void** out) {
    EMG_REQUIRE(cfg && out, "emg_plan_create: null pointer");
    EMG_REQUIRE(cfg->n_slots >= 1 && cfg->n_slots <= 4, "emg_plan_create: 1..4 slots (1 + batches prepared ahead)");
    EMG_REQUIRE(cfg->n_sides >= 1 && cfg->n_sides <= 4 && cfg->eta >= 1, "emg_plan_create: bad eta / sides");
    EMG_REQUIRE(cfg->ent && cfg->rel && cfg->X && cfg->contrib_ent && cfg->contrib_rel && cfg->loss_accum && cfg->tag_ent && cfg->tag_rel,
                "emg_plan_create: null buffer");
    EMG_REQUIRE(cfg->fused || (cfg->scores && cfg->g), "emg_plan_create: the unfused step needs the score / gradient buffers");
    EMG_REQUIRE(!(cfg->inplace && (cfg->lp_lambda_ent != 0.f || cfg->lp_lambda_rel != 0.f)) || (cfg->opt == EMG_OPT_SGD && cfg->lp_p <= 3),
                "emg_plan_create: in-place singleton updates fold an LP regulariser for plain SGD and p <= 3 only");
    EMG_REQUIRE((cfg->lp_lambda_ent == 0.f && cfg->lp_lambda_rel == 0.f) || cfg->lp_sum, "emg_plan_create: LP needs lp_sum");
    EMG_REQUIRE(!cfg->lr_t_hist || ((cfg->opt == EMG_OPT_ADAM || cfg->lp_lambda_ent != 0.f || cfg->lp_lambda_rel != 0.f) && !cfg->normalize),
                "emg_plan_create: a deferred dense pass (lr_t_hist) is for EMG_OPT_ADAM and / or an LP regulariser, without row normalisation");
    EMG_REQUIRE(!cfg->lr_t_hist || emg_plan_deferred_ok(cfg->cap_B, cfg->eta * cfg->n_sides, cfg->n_ent, cfg->n_rel),
                "emg_plan_create: a deferred dense pass needs the counting grouping for both tables (emg_plan_deferred_ok): "
                "n_ent = %lld, n_rel = %lld against %lld gradient rows per batch", (long long)cfg->n_ent, (long long)cfg->n_rel,
                (long long)((2 + (int64_t)cfg->eta * cfg->n_sides) * cfg->cap_B));
    EMG_REQUIRE(cfg->inplace >= 0 && cfg->inplace <= 2, "emg_plan_create: inplace is 0 (off), 1 (singletons in place) or 2 (a stateful optimizer's "
                                                        "window form)");
    EMG_REQUIRE(cfg->inplace != 2 || (cfg->opt != EMG_OPT_SGD && cfg->fused), "emg_plan_create: inplace = 2 is for stateful optimizers in the fused step");
    EMG_REQUIRE(!(cfg->lr_t_hist && cfg->inplace == 1 && cfg->opt != EMG_OPT_SGD),
                "emg_plan_create: a stateful optimizer's in-place updates under the deferred dense pass need inplace = 2 (Adam's in-kernel replay)");
    if (cfg->lr_t_hist && cfg->inplace == 2) {   // (what the in-kernel replay needs; the apply that finishes the other rows is the descriptor-driven one)
        const bool cplx = cfg->model == EMG_COMPLEX || cfg->model == EMG_HOLE;
        const int n = cplx ? cfg->k_int / 2 : cfg->k_int;
        EMG_REQUIRE(cfg->opt == EMG_OPT_ADAM && cfg->fused && cfg->lp_lambda_ent == 0.f && cfg->lp_lambda_rel == 0.f && n % 4 == 0 && n / 4 <= 64 && cfg->k_int > 64,
                    "emg_plan_create: in-place Adam under the deferred dense pass needs the fused step, no regulariser and 16-byte rows of 17 "
                    "... 64 chunks (per half for complex models)");
    }
    Plan* P = new Plan();
    P->cfg = *cfg;
    P->n_side = cfg->n_slots - 1 > 2 ? 2 : cfg->n_slots - 1;
    auto bail = [&](const char* what) { emg_plan_destroy(P); return fail(EMG_EHIP, "emg_plan_create: %s failed", what); };
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // hi = numerically smallest = highest priority
    static const bool side_high = getenv("EMG_SIDE_PRIO") && atoi(getenv("EMG_SIDE_PRIO")) != 0;   // A/B aid
    for (int i = 0; i < P->n_side; ++i)
        // LOW priority: the preparation runs two batches ahead and has a whole step of slack; at high priority (round 2: "the
        // small kernels of a chain must not queue behind the big ones") its waves displace scoring waves — C3 0.369 vs
        // 0.363 ms/step in three alternating pairs, B = 131 072 / Zipf / TransE unchanged.  (Streams confined to every 2nd / 4th /
        // 8th CU with hipExtStreamCreateWithCUMask: 0.43-0.48 ms/step — far worse.)
        if (hipStreamCreateWithPriority(&P->side[i], hipStreamNonBlocking, side_high ? hi : lo) != hipSuccess) return bail("hipStreamCreateWithPriority");
    if (P->n_side > 0) {
        if (hipStreamCreateWithFlags(&P->aux, hipStreamNonBlocking) != hipSuccess) return bail("hipStreamCreate");
        if (hipEventCreateWithFlags(&P->fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&P->join, hipEventDisableTiming) != hipSuccess) return bail("hipEventCreate");
    }
    for (int i = 0; i < P->n_side; ++i)
        if (hipEventCreateWithFlags(&P->side_join[i], hipEventDisableTiming) != hipSuccess) return bail("hipEventCreate");
    if (!P->fork && hipEventCreateWithFlags(&P->fork, hipEventDisableTiming) != hipSuccess) return bail("hipEventCreate");
    if (hipStreamCreateWithFlags(&P->cap, hipStreamNonBlocking) != hipSuccess) return bail("hipStreamCreate");
    if (hipEventCreateWithFlags(&P->scored, hipEventDisableTiming) != hipSuccess) return bail("hipEventCreate");
    for (int i = 0; i < cfg->n_slots; ++i) {
        P->slots[i].buf = cfg->slots[i];
        // the grouping workspaces' control regions start out zero (emg_prepare_args.ws_clean)
        const int64_t et = (int64_t)cfg->eta * cfg->n_sides;
        GroupWs we, wr;
        if (group_ws_layout(cfg->slots[i].ws_ent, cfg->slots[i].ws_ent_bytes, (2 + et) * cfg->cap_B, cfg->n_ent, 0, &we) != EMG_OK ||
            group_ws_layout(cfg->slots[i].ws_rel, cfg->slots[i].ws_rel_bytes, cfg->cap_B, cfg->n_rel, 0, &wr) != EMG_OK) {
            emg_plan_destroy(P);
            return EMG_EINVAL;
        }
        if (hipMemset((char*)cfg->slots[i].ws_ent + we.clean_offset, 0, we.clean_bytes) != hipSuccess ||
            hipMemset((char*)cfg->slots[i].ws_rel + wr.clean_offset, 0, wr.clean_bytes) != hipSuccess) return bail("hipMemset");
        if (hipEventCreateWithFlags(&P->slots[i].ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&P->slots[i].done, hipEventDisableTiming) != hipSuccess) return bail("hipEventCreate");
    }
    *out = P;
    return EMG_OK;
}

extern "C" int emg_plan_destroy(void* plan) {
    if (!plan) return EMG_OK;
    Plan* P = (Plan*)plan;
    for (auto& v : P->ev)
        for (auto& e : v) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (int i = 0; i < 4; ++i) {
        if (P->slots[i].ready) (void)hipEventDestroy(P->slots[i].ready);
        if (P->slots[i].done) (void)hipEventDestroy(P->slots[i].done);
    }
    for (auto& g : P->graphs) (void)hipGraphExecDestroy(g.second);
    for (int i = 0; i < 2; ++i)
        if (P->side_join[i]) (void)hipEventDestroy(P->side_join[i]);
    if (P->scored) (void)hipEventDestroy(P->scored);
    if (P->fork) (void)hipEventDestroy(P->fork);
    if (P->join) (void)hipEventDestroy(P->join);
    for (int i = 0; i < 2; ++i)
        if (P->side[i]) { (void)hipStreamSynchronize(P->side[i]); (void)hipStreamDestroy(P->side[i]); }
    if (P->aux) { (void)hipStreamSynchronize(P->aux); (void)hipStreamDestroy(P->aux); }
    if (P->cap) (void)hipStreamDestroy(P->cap);
    delete P;
    return EMG_OK;
}

extern "C" int emg_plan_step(void* plan, const emg_plan_batch* cur, int32_t step, const float* hyper6,
                             const emg_plan_batch* next, int32_t n_next, void* stream) {
    EMG_REQUIRE(plan && cur && hyper6, "emg_plan_step: null pointer");
    Plan* P = (Plan*)plan;
    const emg_plan_config& c = P->cfg;
    if (cur->B <= 0) return EMG_OK;
    EMG_REQUIRE(cur->B <= c.cap_B && cur->start >= 0 && cur->start + cur->B <= c.n_triples, "emg_plan_step: batch [%lld, +%lld) outside the resident training set / scratch capacity",
                (long long)cur->start, (long long)cur->B);
    hipStream_t main = (hipStream_t)stream;
    SlotState* sl = nullptr;
    for (int i = 0; i < c.n_slots; ++i)
        if (same_key(P->slots[i], *cur)) { sl = &P->slots[i]; break; }
    int rc;
    if (!sl) {  // not prepared ahead: take a free slot (else the first one) and prepare now
        sl = &P->slots[0];
        for (int i = 0; i < c.n_slots; ++i)
            if (!P->slots[i].has_key) { sl = &P->slots[i]; break; }
        rc = prepare(P, *sl, *cur, main);
        if (rc != EMG_OK) return rc;
    }
    // batches ahead (nearest first): each goes to a free slot unless already held.  Their preparation chains are enqueued
    // BETWEEN this step's scoring kernel and its applies and start behind the scoring kernel: at 3 waves per SIMD (168
    // VGPRs) that kernel has no room for a co-resident wave, so a preparation wave beside it displaces a scoring wave one
    // for one, while the apply kernel (5 waves per SIMD of 81 VGPRs) leaves room (EMG_PREP_AT_START=1: the round-2 order)
    auto look_ahead = [&]() -> int {
        for (int j = 0; j < n_next && P->n_side > 0; ++j) {
            const emg_plan_batch& nb = next[j];
            if (nb.B <= 0 || nb.B > c.cap_B || nb.start < 0 || nb.start + nb.B > c.n_triples) continue;
            bool held = false;
            for (int i = 0; i < c.n_slots; ++i) held = held || same_key(P->slots[i], nb);
            if (held) continue;
            SlotState* fr = nullptr;
            for (int i = 0; i < c.n_slots; ++i)
                if (&P->slots[i] != sl && !P->slots[i].has_key) { fr = &P->slots[i]; break; }
            if (!fr) break;
            int r = prepare(P, *fr, nb, main);
            if (r != EMG_OK) return r;
        }
        return EMG_OK;
    };
    // Measured (same box, alternating): C3 (360 k contributions per batch) 0.333-0.367 ms/step from the start vs 0.368-0.374
    // behind the scoring kernel — the apply is as short as the chain (0.08 ms) and slows by what it hides (0.085 -> 0.125) —,
    // B = 131 072 (2.9 M contributions) 2.84-2.89 vs 2.68-2.80: the scoring kernel runs clean (1.56 -> 1.11-1.17 ms) and the
    // 1.3 ms apply absorbs the 0.45 ms chain.  So: behind the scoring kernel from a million contributions per batch up.
    static const int at_env = getenv("EMG_PREP_AT_START") ? atoi(getenv("EMG_PREP_AT_START")) : -1;   // A/B aid
    const bool at_start = at_env >= 0 ? at_env != 0 : (2 + (int64_t)c.eta * c.n_sides) * cur->B < 1000000;
    const std::function<int()> between = [&]() -> int {
        EMG_HIP(hipEventRecord(P->scored, main));
        P->wait_scored = true;
        const int r = look_ahead();
        P->wait_scored = false;
        return r;
    };
    if (at_start || P->n_side == 0) {
        rc = look_ahead();
        if (rc != EMG_OK) return rc;
    }
    if (P->n_side > 0) EMG_HIP(hipStreamWaitEvent(main, sl->ready, 0));
    rc = compute(P, *sl, *cur, step, hyper6, main, nullptr, nullptr, (at_start || P->n_side == 0) ? nullptr : &between);
    if (rc != EMG_OK) return rc;
    if (P->n_side > 0) EMG_HIP(hipEventRecord(sl->done, main));
    // EMG_PLAN_KEEP (timing experiment: what does the preparation chain cost the compute kernels it runs beside?):
    // a consumed slot stays valid, so stepping the same few batches again skips their preparation
    static const bool keep = getenv("EMG_PLAN_KEEP") != nullptr;
    if (!keep) sl->has_key = false;
    return EMG_OK;
}

// ---- the step as a captured graph (small batches: the step is shorter than its dozen launches take to issue) -------------
// A graph node's arguments are fixed at capture; what changes from step to step — which rows, how many, the Philox draw
// counter, the optimizer step number and learning rates — lives in a device array of emg_step_ctl records that every kernel
// of step i reads at ctl + i.  emg_plan_run writes the records of the next <= 32 steps (one tiny kernel whose ARGUMENTS
// are the records) and replays the graph of that many steps: two launches per 32 steps from the host.
static bool graph_capable(const Plan* P) {
    const emg_plan_config& c = P->cfg;
    const bool cplx = c.model == EMG_COMPLEX || c.model == EMG_HOLE;
    const int n = cplx ? c.k_int / 2 : c.k_int;
    const int64_t et = (int64_t)c.eta * c.n_sides;
    return c.ctl_buf && c.ctl_bytes >= (int64_t)sizeof(CtlBlock) && !c.lr_t_hist && c.fused && (n % 4 == 0) && c.k_int / 4 > 16 && c.k_int % 4 == 0 &&
           c.ld_ent % 4 == 0 && c.ld_rel % 4 == 0 && c.ldc % 4 == 0 &&
           group_backend_counting((2 + et) * c.cap_B, c.n_ent) && group_backend_counting(c.cap_B, c.n_rel) &&
           !(getenv("EMG_APPLY") && strcmp(getenv("EMG_APPLY"), "window") == 0);
}

extern "C" int emg_plan_deferred_ok(int64_t cap_B, int32_t eta_total, int64_t n_ent, int64_t n_rel) {
    return cap_B > 0 && group_backend_counting((2 + (int64_t)eta_total) * cap_B, n_ent) && group_backend_counting(cap_B, n_rel) ? 1 : 0;
}

extern "C" int emg_plan_graph_ok(void* plan) { return plan && graph_capable((const Plan*)plan) ? 1 : 0; }

static int capture_steps(Plan* P, int len, const float* hyper6, hipGraphExec_t* out) {
    hipStream_t main = P->cap;
    const emg_plan_config& c = P->cfg;
    StepCtl* ctl = (StepCtl*)c.ctl_buf;
    for (int i = 0; i < c.n_slots; ++i) { P->slots[i].has_key = false; P->slots[i].ready_recorded = false; P->slots[i].done_in_capture = false; }
    P->side_joined[0] = P->side_joined[1] = false;
    P->side_rr = 0;
    const int saved_timing = P->timing_max;
    P->timing_max = 0;
    emg_plan_batch dummy{};
    dummy.B = c.cap_B;
    EMG_HIP(hipStreamBeginCapture(main, hipStreamCaptureModeThreadLocal));
    int rc = EMG_OK;
    // RIDER form (three slots): no side stream, no event — the preparation of later batches rides at the front of the
    // step's two launches (emg_group_kernels.hpp): fused(t) carries the id kernel of batch t + 2 and the scatter of batch
    // t + 1, apply(t) their scan / ordering.  A fork / join per step in a graph costs more than the step's small kernels
    // (tools/hbm_ceiling: 22 us per step against 10.6 for the same six kernels in a line).
    static const bool no_riders = getenv("EMG_RIDERS") && atoi(getenv("EMG_RIDERS")) == 0;   // A/B aid: everything in a line
    const bool ride = c.n_slots >= 3 && !no_riders;
    auto body = [&]() -> int {
        const int saved_side = P->n_side;
        P->n_side = 0;   // (prepare() on the capture's stream)
        auto stages = [&](int i, PrepStages& S) -> int {
            emg_prepare_args a;
            fill_prepare_args(P, P->slots[i % c.n_slots], dummy, ctl + i, a);
            return prepare_stages(&a, &S);
        };
        int r = EMG_OK;
        if (ride) {
            P->ctl = ctl;
            r = prepare(P, P->slots[0], dummy, main);                 // batch 0: all four stages
            if (r == EMG_OK && len > 1) {                              // batch 1: ids + scan
                PrepStages S;
                r = stages(1, S);
                if (r == EMG_OK) {
                    Riders R{};
                    add_rider(R, RIDE_IDS, S.nb_ids, S.G, &S.prep);
                    add_rider(R, RIDE_SCAN, S.nb_scan, S.G, nullptr);
                    r = launch_riders_alone(R, main);
                }
            }
        }
        for (int i = 0; i < len && r == EMG_OK; ++i) {
            Riders RA{}, RB{};
            if (ride) {
                PrepStages S;
                if (i + 2 < len) {
                    r = stages(i + 2, S);
                    if (r != EMG_OK) break;
                    add_rider(RA, RIDE_IDS, S.nb_ids, S.G, &S.prep);
                    add_rider(RB, RIDE_SCAN, S.nb_scan, S.G, nullptr);
                }
                if (i + 1 < len) {
                    r = stages(i + 1, S);
                    if (r != EMG_OK) break;
                    add_rider(RA, RIDE_SCATTER, S.nb_n, S.G, nullptr);
                    add_rider(RB, RIDE_ORDER, S.nb_n, S.G, nullptr);
                }
            } else {
                P->ctl = ctl + i;
                r = prepare(P, P->slots[i % c.n_slots], dummy, main);
                if (r != EMG_OK) break;
            }
            P->ctl = ctl + i;
            r = compute(P, P->slots[i % c.n_slots], dummy, 1, hyper6, main, ride ? &RA : nullptr, ride ? &RB : nullptr);
        }
        P->n_side = saved_side;
        return r;
    };
    rc = body();
    P->ctl = nullptr;
    P->timing_max = saved_timing;
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(main, &graph);
    for (int i = 0; i < c.n_slots; ++i) { P->slots[i].has_key = false; P->slots[i].ready_recorded = false; P->slots[i].done_in_capture = false; }
    if (rc != EMG_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess || !graph) return fail(EMG_EHIP, "emg_plan_run: stream capture failed: %s", hipGetErrorString(e));
    const hipError_t e2 = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e2 != hipSuccess) return fail(EMG_EHIP, "emg_plan_run: hipGraphInstantiate failed: %s", hipGetErrorString(e2));
    return EMG_OK;
}

extern "C" int emg_plan_run(void* plan, const emg_plan_batch* batches, int32_t n, int32_t first_step, const float* hyper6s,
                            void* stream) {
    EMG_REQUIRE(plan && (n == 0 || (batches && hyper6s)), "emg_plan_run: null pointer");
    Plan* P = (Plan*)plan;
    const emg_plan_config& c = P->cfg;
    EMG_REQUIRE(graph_capable(P), "emg_plan_run: this plan cannot run as a graph (see emg_plan_graph_ok)");
    hipStream_t main = (hipStream_t)stream;
    // eager preparations still in flight on the side streams (emg_plan_step's look-ahead) own slot buffers the graph uses
    for (int i = 0; i < c.n_slots; ++i) {
        SlotState& sl = P->slots[i];
        if (sl.has_key && sl.ready_recorded) EMG_HIP(hipStreamWaitEvent(main, sl.ready, 0));
        sl.has_key = false; sl.ready_recorded = false;
    }
    // Graph lengths are powers of two (32, 16, ..., 1): any number of steps is a sum of at most six replays, and ALL six
    // graphs are captured by the first call — a later call (the timed region of a benchmark, the second epoch of a fit)
    // never pays for a capture
    if (n > 0 && P->graphs.empty()) {
        for (int len = kGraphSteps; len >= 1; len >>= 1) {
            hipGraphExec_t exec = nullptr;
            int rc = capture_steps(P, len, hyper6s, &exec);
            if (rc != EMG_OK) return rc;
            P->graphs.push_back({len, exec});
        }
    }
    CtlBlock blk;
    for (int32_t at = 0; at < n;) {
        int want = kGraphSteps;
        while (want > n - at) want >>= 1;
        int len = 0;
        memset(&blk, 0, sizeof(blk));
        for (; len < want; ++len) {
            const emg_plan_batch& b = batches[at + len];
            EMG_REQUIRE(b.B > 0 && b.B <= c.cap_B && b.start >= 0 && b.start + b.B <= c.n_triples,
                        "emg_plan_run: batch [%lld, +%lld) outside the resident training set / scratch capacity", (long long)b.start, (long long)b.B);
            EMG_REQUIRE(!b.inj_repl && !b.inj_mask, "emg_plan_run: injected draws need emg_plan_step");
            StepCtl& r = blk.rec[len];
            r.start = b.start; r.B = b.B;
            r.draw_counter0 = (uint64_t)(((int64_t)(b.epoch - 1) * c.batches_count + (b.batch - 1)) * c.n_sides);
            r.n_choices = b.n_choices > 0 ? b.n_choices : 0; r.entities_list = b.n_choices > 0 ? b.entities_list : nullptr;
            r.step = first_step + at + len;
            const float* h = hyper6s + 6 * (size_t)(at + len);
            for (int k = 0; k < 6; ++k) r.hyper_ent[k] = r.hyper_rel[k] = h[k];
            r.hyper_ent[6] = c.lp_lambda_ent; r.hyper_rel[6] = c.lp_lambda_rel; r.hyper_ent[7] = r.hyper_rel[7] = (float)c.lp_p;
        }
        hipGraphExec_t exec = nullptr;
        for (auto& g : P->graphs) if (g.first == len) exec = g.second;
        EMG_REQUIRE(exec, "emg_plan_run: no graph of %d steps", len);
        hipLaunchKernelGGL(ctl_write_kernel, dim3(1), dim3(256), 0, main, blk, (StepCtl*)c.ctl_buf, len);
        EMG_LAUNCH_CHECK();
        EMG_HIP(hipGraphLaunch(exec, main));
        at += len;
    }
    // a later eager step prepares on the side streams: behind everything the graphs did to the slots
    if (P->n_side > 0)
        for (int i = 0; i < c.n_slots; ++i) EMG_HIP(hipEventRecord(P->slots[i].done, main));
    return EMG_OK;
}

extern "C" int emg_plan_timing(void* plan, int32_t max_samples) {
    EMG_REQUIRE(plan, "emg_plan_timing: null plan");
    Plan* P = (Plan*)plan;
    for (auto& v : P->ev) {
        for (auto& e : v) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        v.clear();
    }
    P->timing_max = max_samples > 0 ? max_samples : 0;
    return EMG_OK;
}

extern "C" int emg_plan_stage_ms(void* plan, float* avg_ms, int32_t* counts) {
    EMG_REQUIRE(plan && avg_ms && counts, "emg_plan_stage_ms: null pointer");
    Plan* P = (Plan*)plan;
    EMG_HIP(hipDeviceSynchronize());
    for (int s = 0; s < ST_COUNT; ++s) {
        double tot = 0.0;
        for (auto& e : P->ev[s]) {
            float ms = 0.f;
            EMG_HIP(hipEventElapsedTime(&ms, e.first, e.second));
            tot += ms;
        }
        counts[s] = (int32_t)P->ev[s].size();
        avg_ms[s] = counts[s] ? (float)(tot / counts[s]) : 0.f;
    }
    return EMG_OK;
}
