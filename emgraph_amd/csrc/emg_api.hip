// emg_api.hip — one-call forms of the hot path (the C-ABI SURVEY.md 8b sketches: emg_corrupt_fit, emg_train_step,
// emg_rank_1vsall).  Pure compositions of the fine-grained entry points in this library, on ONE stream: what a
// maintainer binding the library from the reference calls once per batch / per test set.  The Python host side
// of this repository uses the fine-grained calls instead because it overlaps batch preparation, the two applies
// and the host-side filter index on several streams (emgraph_amd/training.py, evaluation/ranking.py).
//
// Scratch: emg_train_step carves a caller-provided workspace (emg_train_step_workspace_bytes);
// emg_corrupt_fit / emg_rank_1vsall take stream-ordered scratch from the HIP memory pool (hipMallocAsync /
// hipFreeAsync on the call's stream: nothing persists after the call).
#include "emg_common.hpp"

namespace emg {

static inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

// rank assembly of EmbeddingModel.py:1966-1986 from the four counters (evaluation/ranking.py::ranks_from_counts)
__device__ __forceinline__ int cmp_strategy(int gt, int eq, int strategy) {
    return strategy == 0 ? gt + eq : (strategy == 1 ? gt : gt + (eq + 1) / 2);  // worst | best | middle (ceil)
}

__global__ void ranks_kernel(const int32_t* __restrict__ cnt, int64_t n_rows, int64_t n_q, int side_mode, int strategy,
                             int32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_q) return;
    const int32_t *gt = cnt, *eq = cnt + n_rows, *fgt = cnt + 2 * n_rows, *feq = cnt + 3 * n_rows;
    if (side_mode == EMG_EVAL_S || side_mode == EMG_EVAL_O) {
        out[i] = cmp_strategy(gt[i], eq[i], strategy) + 1 - cmp_strategy(fgt[i], feq[i], strategy);
        return;
    }
    const int64_t o = i, s = n_q + i;  // object-side rows first
    if (side_mode == EMG_EVAL_S_O) {    // [subject_rank, object_rank]
        out[2 * i + 0] = cmp_strategy(gt[s], eq[s], strategy) + 1 - cmp_strategy(fgt[s], feq[s], strategy);
        out[2 * i + 1] = cmp_strategy(gt[o], eq[o], strategy) + 1 - cmp_strategy(fgt[o], feq[o], strategy);
    } else {                            // 's+o': one rank against both blocks
        out[i] = cmp_strategy(gt[o] + gt[s], eq[o] + eq[s], strategy) + 1 - cmp_strategy(fgt[s], feq[s], strategy) -
                 cmp_strategy(fgt[o], feq[o], strategy);
    }
}

}  // namespace emg

using namespace emg;

extern "C" int emg_corrupt_fit(const int32_t* pos, int64_t B, int32_t eta, int side, int64_t entities_size,
                               const int32_t* entities_list, int64_t n_list, uint64_t seed, uint64_t counter,
                               int32_t* out_spo, void* stream) {
    EMG_REQUIRE(B >= 0 && eta >= 1, "emg_corrupt_fit: bad sizes");
    if (B == 0) return EMG_OK;
    EMG_REQUIRE(pos && out_spo, "emg_corrupt_fit: null pointer");
    // protocol.py:616-641: entities_size > 0 draws ids in [0, entities_size); otherwise from entities_list
    const int64_t n_choices = entities_size > 0 ? entities_size : n_list;
    EMG_REQUIRE(n_choices > 0 && (entities_size > 0 || entities_list), "emg_corrupt_fit: no corruption entities");
    hipStream_t st = (hipStream_t)stream;
    int32_t* codes = nullptr;
    EMG_HIP(hipMallocAsync((void**)&codes, (size_t)B * eta * sizeof(int32_t), st));
    int rc = emg_corrupt_codes(B, eta, side, n_choices, entities_size > 0 ? nullptr : entities_list, seed, counter, nullptr,
                               nullptr, codes, stream);
    if (rc == EMG_OK) rc = emg_corrupt_expand(pos, B, eta, codes, out_spo, stream);
    (void)hipFreeAsync(codes, st);
    return rc;
}

extern "C" int emg_rank_1vsall(int model, const float* ent, int64_t n_ent, int64_t ld_ent, const float* rel,
                               int64_t n_rel, int64_t ld_rel, int32_t k_int, float scale, const int32_t* test_spo,
                               int64_t n_q, int side_mode, const int32_t* cand, int64_t n_cand,
                               const int64_t* filt_ptr, const int32_t* filt_idx, int strategy, int precision_mode,
                               int32_t* rank_out, void* stream) {
    EMG_REQUIRE(side_mode >= EMG_EVAL_S && side_mode <= EMG_EVAL_S_O, "emg_rank_1vsall: bad side_mode %d", side_mode);
    EMG_REQUIRE(strategy >= 0 && strategy <= 2, "emg_rank_1vsall: strategy must be 0 worst, 1 best, 2 middle");
    EMG_REQUIRE(precision_mode >= 0 && precision_mode <= 2, "emg_rank_1vsall: precision_mode %d is not built", precision_mode);
    EMG_REQUIRE(precision_mode != 1 || (model >= EMG_DISTMULT && model <= EMG_HOLE),
                "emg_rank_1vsall: the bf16 mode needs a contraction model (DistMult, ComplEx, HolE)");
    // mode 2 = the SAME ranks as mode 0 through the half-precision prefilter: where its kernel does not apply (TransE, a
    // candidate list, an uncovered width, few rows) the exact kernel runs instead — same result either way
    const bool sad = precision_mode == 2 && model == EMG_TRANSE_L1 && cand == nullptr;   // TransE-L1: the fixed-point prefilter
    const bool l2 = precision_mode == 2 && model == EMG_TRANSE_L2 && cand == nullptr;    // TransE-L2: MFMA prefilter on augmented rows
    if (precision_mode == 2 && !sad && !l2 && (!(model >= EMG_DISTMULT && model <= EMG_HOLE) || cand != nullptr)) precision_mode = 0;
    if (n_q == 0) return EMG_OK;
    EMG_REQUIRE(ent && rel && test_spo && rank_out, "emg_rank_1vsall: null pointer");
    EMG_REQUIRE((filt_ptr == nullptr) == (filt_idx == nullptr) || filt_ptr, "emg_rank_1vsall: filter CSR needs both arrays");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n_rows = side_mode >= EMG_EVAL_SPO ? 2 * n_q : n_q;
    const int64_t ldq = (k_int + 3) / 4 * 4;
    const int64_t nc = cand ? n_cand : n_ent;
    const int64_t ldb = sad ? emg_eval_sad_ld(k_int)   // row stride of the 16-bit images
                            : (precision_mode == 2 ? emg_eval_prefilter_ld(k_int + (l2 ? 2 : 0)) : (k_int + 63) / 64 * 64);
    const size_t q_bytes = up256((size_t)n_rows * ldq * 4), p_bytes = up256((size_t)n_rows * 4);
    const size_t c_bytes = up256((size_t)4 * n_rows * 4);
    size_t total = q_bytes + p_bytes + c_bytes;
    size_t qb_off = 0, se_off = 0, eb_off = 0;
    if (precision_mode == 1) {
        qb_off = total; total += up256((size_t)n_rows * ldb * 2);
        se_off = total; total += p_bytes;
        eb_off = total; total += up256((size_t)n_ent * ldb * 2);
    }
    // precision 2: half copies of Q and the table, the band, the pair buffer (one segment per prefilter wave)
    size_t band_off = 0, bounds_off = 0, pairs_off = 0, pcount_off = 0, q2_off = 0, thr_off = 0;
    int64_t n_seg = 0, pair_cap = 0;
    if (precision_mode == 2) {
        const int32_t pre_cols = (k_int + (l2 ? 2 : 0) + 15) / 16 * 16;   // contraction width of the half-precision prefilter
        n_seg = sad ? emg_eval_sad_segments(n_rows, nc) : emg_eval_prefilter_segments_k(n_rows, nc, pre_cols);
        int64_t per = ((int64_t)1 << 27) / (n_seg > 0 ? n_seg : 1);   // <= 1 GiB of pairs in total
        per = per < 64 ? 64 : (per > 2048 ? 2048 : per);
        pair_cap = n_seg * per;
        qb_off = total; total += up256((size_t)n_rows * ldb * 2);
        eb_off = total; total += up256((size_t)n_ent * ldb * 2);
        band_off = total; total += 2 * p_bytes;   // (the fixed-point prefilter keeps two thresholds per row here)
        bounds_off = total; total += 256;
        pcount_off = total; total += up256((size_t)(n_seg + 1) * 4);
        pairs_off = total; total += up256((size_t)pair_cap * 8);
        if (l2) {
            q2_off = total; total += q_bytes;
            thr_off = total; total += up256((size_t)2 * n_rows * 4);
        }
    }
    char* ws = nullptr;
    EMG_HIP(hipMallocAsync((void**)&ws, total, st));
    float* Q = (float*)ws;
    int32_t* pos_int = (int32_t*)(ws + q_bytes);
    int32_t* cnt = (int32_t*)(ws + q_bytes + p_bytes);
    int rc = EMG_OK;
    auto step = [&](int r) { if (rc == EMG_OK) rc = r; };
    if (hipMemsetAsync(cnt, 0, (size_t)4 * n_rows * 4, st) != hipSuccess) rc = fail(EMG_EHIP, "emg_rank_1vsall: memset failed");
    step(emg_eval_build_queries(model, ent, n_ent, ld_ent, rel, n_rel, ld_rel, k_int, scale, test_spo, n_q, side_mode, Q,
                                ldq, pos_int, stream));
    if (precision_mode == 2 && nc > 0) {
        void* Qh = ws + qb_off;
        void* Eh = ws + eb_off;
        float* band = (float*)(ws + band_off);
        double* bounds = (double*)(ws + bounds_off);
        uint64_t* pairs = (uint64_t*)(ws + pairs_off);
        uint32_t* pcount = (uint32_t*)(ws + pcount_off);
        if (l2) {
            float* Q2 = (float*)(ws + q2_off);
            step(emg_to_f16_l2(ent, n_ent, ld_ent, k_int, 0, Eh, ldb, nullptr, bounds + 3, stream));
            step(emg_to_f16_l2(Q, n_rows, ldq, k_int, 1, Qh, ldb, Q2, nullptr, stream));
            step(emg_eval_prefilter_bounds(ent, n_ent, ld_ent, Eh, ldb, k_int, bounds, stream));
            step(emg_eval_prefilter_band(Q2, n_rows, ldq, Qh, ldb, k_int, bounds, band, stream));
            step(emg_eval_l2_thresholds(Q, n_rows, ldq, pos_int, band, bounds, k_int, (float*)(ws + thr_off), stream));
        } else if (sad) {
            step(emg_eval_sad_range(ent, n_ent, ld_ent, rel, n_rel, ld_rel, k_int, bounds, stream));
            step(emg_eval_sad_quantize(ent, n_ent, ld_ent, k_int, bounds, Eh, ldb, stream));
            step(emg_eval_sad_quantize(Q, n_rows, ldq, k_int, bounds, Qh, ldb, stream));
            step(emg_eval_sad_thresholds(pos_int, n_rows, k_int, bounds, (uint32_t*)band, (uint32_t*)((char*)band + p_bytes), stream));
        } else {
            step(emg_to_f16(ent, n_ent, ld_ent, k_int, Eh, ldb, stream));
            step(emg_to_f16(Q, n_rows, ldq, k_int, Qh, ldb, stream));
            step(emg_eval_prefilter_bounds(ent, n_ent, ld_ent, Eh, ldb, k_int, bounds, stream));
            step(emg_eval_prefilter_band(Q, n_rows, ldq, Qh, ldb, k_int, bounds, band, stream));
        }
        bool exact = rc != EMG_OK;
        if (l2 && rc == EMG_OK) {
            const int r = emg_eval_prefilter_f16_thr(Qh, ldb, (const float*)(ws + thr_off), n_rows, Eh, n_ent, ldb, 0,
                                                     (k_int + 2 + 15) / 16 * 16, cnt, pairs, pcount, pair_cap, stream);
            if (r == EMG_ENOSUP) exact = true;   // a width the register-stationary kernel does not cover
            else step(r);
        } else if (sad && rc == EMG_OK) {
            uint32_t* lo = (uint32_t*)band;
            uint32_t* hi = (uint32_t*)((char*)band + p_bytes);
            step(emg_eval_prefilter_sad(Qh, ldb, lo, hi, n_rows, Eh, n_ent, ldb, 0, k_int, cnt, pairs, pcount, pair_cap, stream));
        } else if (rc == EMG_OK) {
            const int r = emg_eval_prefilter_f16(model, Qh, ldb, pos_int, band, n_rows, Eh, n_ent, ldb, 0, (k_int + 15) / 16 * 16,
                                                 scale, cnt, pairs, pcount, pair_cap, stream);
            if (r == EMG_ENOSUP) exact = true;   // a shape the register-stationary kernel does not cover
            else step(r);
        }
        if (rc == EMG_OK && !exact) {
            step(emg_eval_rescore_pairs_rows(model, Q, ldq, pos_int, ent, ld_ent, 0, k_int, scale, pairs, pair_cap, pcount, n_seg,
                                             sad ? 4 : emg_eval_prefilter_waves((k_int + (l2 ? 2 : 0) + 15) / 16 * 16), sad ? 0 : 32,
                                             cnt, cnt + n_rows, stream));
            uint32_t over = 0;   // some wave ran out of pair room: these counters are void, the exact kernel redoes them
            if (rc == EMG_OK && (hipMemcpyAsync(&over, pcount + n_seg, 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
                                 hipStreamSynchronize(st) != hipSuccess))
                rc = fail(EMG_EHIP, "emg_rank_1vsall: reading the prefilter's overflow flag failed");
            exact = over != 0;
            if (rc == EMG_OK && exact && !l2 && !sad && pair_cap / n_seg >= 2048) {
                // too many undecided candidates: on a table of small scores they are TIES with the positive, which the prefilter's second
                // form proves (emg_eval_prefilter_f16_ties: the fresh model, the first epochs) — once more through it before the exact kernel
                if (hipMemsetAsync(cnt, 0, (size_t)2 * n_rows * 4, st) != hipSuccess) rc = fail(EMG_EHIP, "emg_rank_1vsall: memset failed");
                const int r = rc != EMG_OK ? rc : emg_eval_prefilter_f16_ties(model, Qh, ldb, pos_int, band, n_rows, Eh, n_ent, ldb, 0, (k_int + 15) / 16 * 16,
                                                                              scale, cnt, cnt + n_rows, pairs, pcount, pair_cap, stream);
                if (r == EMG_OK) {
                    step(emg_eval_rescore_pairs_rows(model, Q, ldq, pos_int, ent, ld_ent, 0, k_int, scale, pairs, pair_cap, pcount, n_seg,
                                                     emg_eval_prefilter_waves((k_int + 15) / 16 * 16), 32, cnt, cnt + n_rows, stream));
                    over = 0;
                    if (rc == EMG_OK && (hipMemcpyAsync(&over, pcount + n_seg, 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
                                         hipStreamSynchronize(st) != hipSuccess))
                        rc = fail(EMG_EHIP, "emg_rank_1vsall: reading the prefilter's overflow flag failed");
                    exact = over != 0;
                }   // (any error of the second form: the exact kernel, below)
            }
        }
        if (rc == EMG_OK && exact) {
            if (hipMemsetAsync(cnt, 0, (size_t)2 * n_rows * 4, st) != hipSuccess) rc = fail(EMG_EHIP, "emg_rank_1vsall: memset failed");
            step(emg_eval_count(model, Q, ldq, pos_int, n_rows, ent, nc, ld_ent, cand, k_int, scale, 0, nullptr, 0, cnt,
                                cnt + n_rows, stream));
        }
        if (filt_ptr)
            step(emg_eval_filter_count(model, Q, ldq, pos_int, n_rows, ent, n_ent, ld_ent, 0, k_int, scale, 0, filt_ptr,
                                       filt_idx, cnt + 2 * n_rows, cnt + 3 * n_rows, stream));
    } else if (precision_mode == 0 || precision_mode == 2) {
        if (nc > 0)
            step(emg_eval_count(model, Q, ldq, pos_int, n_rows, ent, nc, ld_ent, cand, k_int, scale, 0, nullptr, 0, cnt,
                                cnt + n_rows, stream));
        if (filt_ptr && nc > 0)
            step(emg_eval_filter_count(model, Q, ldq, pos_int, n_rows, ent, n_ent, ld_ent, 0, k_int, scale, 0, filt_ptr,
                                       filt_idx, cnt + 2 * n_rows, cnt + 3 * n_rows, stream));
    } else {
        void* Qb = ws + qb_off;
        int32_t* self_ent = (int32_t*)(ws + se_off);
        void* Eb = ws + eb_off;
        const int32_t k_pad = (k_int + 15) / 16 * 16;
        step(emg_to_bf16(ent, n_ent, ld_ent, k_int, Eb, ldb, stream));
        step(emg_to_bf16(Q, n_rows, ldq, k_int, Qb, ldb, stream));
        step(emg_eval_pos_int_bf16(model, Eb, ldb, k_int, scale, test_spo, n_q, side_mode, Qb, ldb, pos_int, self_ent, stream));
        if (nc > 0)
            // 'worst' reads only #(>=), 'best' only #(>): one comparison per score (see emg_eval_count_bf16)
            step(emg_eval_count_bf16(model, Qb, ldb, pos_int, self_ent, n_rows, Eb, nc, ldb, cand, 0, k_pad, scale, cnt,
                                     cnt + n_rows, strategy == 0 ? 1 : (strategy == 1 ? 2 : 0), stream));
        if (filt_ptr && nc > 0)
            step(emg_eval_filter_count_bf16(model, Qb, ldb, pos_int, self_ent, n_rows, Eb, n_ent, ldb, 0, k_int, scale,
                                            filt_ptr, filt_idx, cnt + 2 * n_rows, cnt + 3 * n_rows, stream));
    }
    if (rc == EMG_OK) {
        hipLaunchKernelGGL(ranks_kernel, dim3((unsigned)cdiv(n_q, 256)), dim3(256), 0, st, cnt, n_rows, n_q, side_mode,
                           strategy, rank_out);
        if (hipGetLastError() != hipSuccess) rc = fail(EMG_EHIP, "emg_rank_1vsall: launch failed");
    }
    (void)hipFreeAsync(ws, st);
    return rc;
}

// ---- one training batch ---------------------------------------------------------------------------------
struct StepLayout {
    size_t codes, dest_ent, dest_rel, single, contrib_ent, contrib_rel, scores, g, ws_ent, ws_rel, total;
    int64_t n_ce, n_neg, ldc, ws_ent_bytes, ws_rel_bytes;
};

static int step_layout(int64_t B, int32_t eta_total, int32_t k_int, int64_t n_ent, int64_t n_rel, StepLayout* L) {
    L->n_neg = B * (int64_t)eta_total;
    L->n_ce = 2 * B + L->n_neg;
    L->ldc = (k_int + 3) / 4 * 4;
    L->ws_ent_bytes = emg_apply_workspace_bytes_ex(L->n_ce, n_ent, k_int);
    L->ws_rel_bytes = emg_apply_workspace_bytes_ex(B, n_rel, k_int);
    EMG_REQUIRE(L->ws_ent_bytes >= 0 && L->ws_rel_bytes >= 0, "emg_train_step: workspace size query failed");
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += up256(bytes); return at; };
    L->codes = take((size_t)L->n_neg * 4);
    L->dest_ent = take((size_t)L->n_ce * 4);
    L->dest_rel = take((size_t)B * 4);
    L->single = take((size_t)L->n_ce);
    L->contrib_ent = take((size_t)L->n_ce * L->ldc * 4);
    L->contrib_rel = take((size_t)B * L->ldc * 4);
    L->scores = take((size_t)(B + L->n_neg) * 4);
    L->g = take((size_t)(B + L->n_neg) * 4);
    L->ws_ent = take((size_t)L->ws_ent_bytes);
    L->ws_rel = take((size_t)L->ws_rel_bytes);
    L->total = o;
    return EMG_OK;
}

extern "C" int64_t emg_train_step_workspace_bytes(int64_t B, int32_t eta_total, int32_t k_int, int64_t n_ent, int64_t n_rel) {
    if (B <= 0) return 256;
    StepLayout L;
    if (step_layout(B, eta_total, k_int, n_ent, n_rel, &L) != EMG_OK) return -1;
    return (int64_t)L.total;
}

extern "C" int emg_train_step(const emg_step_args* a, void* stream) {
    EMG_REQUIRE(a, "emg_train_step: null args");
    EMG_REQUIRE(a->B >= 0 && a->eta >= 1 && a->n_sides >= 1 && a->n_sides <= 4, "emg_train_step: bad sizes");
    if (a->B == 0) return EMG_OK;
    EMG_REQUIRE(a->ent && a->rel && a->pos && a->loss_accum && a->workspace, "emg_train_step: null pointer");
    EMG_REQUIRE(a->loss >= EMG_LOSS_PAIRWISE && a->loss <= EMG_LOSS_MULTICLASS_NLL, "emg_train_step: unknown loss %d", a->loss);
    EMG_REQUIRE(a->hyper[6] == 0.f, "emg_train_step: runs without the LP regulariser (hyper[6] must be 0)");
    const int32_t et = a->eta * a->n_sides;
    StepLayout L;
    int rc = step_layout(a->B, et, a->k_int, a->n_ent, a->n_rel, &L);
    if (rc != EMG_OK) return rc;
    EMG_REQUIRE((int64_t)L.total <= a->workspace_bytes, "emg_train_step: workspace too small (%lld < %lld)",
                (long long)a->workspace_bytes, (long long)L.total);
    char* ws = (char*)a->workspace;
    int32_t* codes = (int32_t*)(ws + L.codes);
    int32_t* dest_ent = (int32_t*)(ws + L.dest_ent);
    int32_t* dest_rel = (int32_t*)(ws + L.dest_rel);
    uint8_t* single = (uint8_t*)(ws + L.single);
    float* ce = (float*)(ws + L.contrib_ent);
    float* cr = (float*)(ws + L.contrib_rel);
    float* sp = (float*)(ws + L.scores);
    float* sn = sp + a->B;
    float* gp = (float*)(ws + L.g);
    float* gn = gp + a->B;
    const bool generic = a->model == EMG_TRANSE_P;   // TransE with an order of the norm other than 1 / 2: generic kernels, unfused, every row through the apply
    const bool inplace = a->inplace != 0 && !generic;

    emg_prepare_args pa{};
    pa.pos = a->pos; pa.B = a->B; pa.eta = a->eta; pa.n_sides = a->n_sides;
    for (int i = 0; i < a->n_sides; ++i) pa.sides[i] = a->sides[i];
    pa.n_choices = a->n_choices > 0 ? a->n_choices : a->n_ent; pa.entities_list = a->entities_list;
    pa.seed = a->seed; pa.draw_counter0 = a->draw_counter0; pa.inj_mask = a->inj_mask; pa.inj_repl = a->inj_repl;
    pa.codes = codes; pa.dest_ent = dest_ent; pa.n_extra_ent = 0; pa.n_ent = a->n_ent;
    pa.dest_rel = dest_rel; pa.n_extra_rel = 0; pa.n_rel = a->n_rel;
    pa.ws_ent = ws + L.ws_ent; pa.ws_ent_bytes = L.ws_ent_bytes; pa.ws_rel = ws + L.ws_rel; pa.ws_rel_bytes = L.ws_rel_bytes;
    pa.single_flags = inplace ? single : nullptr;
    // bilinear models: a negative's gradient row stays factored (one float x a query row of its group) between the
    // backward pass and the apply, as in emg_plan_step; rows wider than the register-tiled kernels take the separate
    // forward / loss / backward path (column blocks)
    const bool factored = !(a->model == EMG_TRANSE_L1 || a->model == EMG_TRANSE_L2 || generic);
    const bool cplx = a->model == EMG_COMPLEX || a->model == EMG_HOLE;
    const bool wide = (cplx ? a->k_int / 2 : a->k_int) > 512;
    pa.factored = factored;
    rc = emg_prepare_batch(&pa, stream);
    if (rc != EMG_OK) return rc;

    emg_backward_args ba{};
    ba.model = a->model; ba.k_int = a->k_int; ba.scale = a->scale; ba.eta = et;
    ba.ent = a->ent; ba.n_ent = a->n_ent; ba.ld_ent = a->ld_ent; ba.rel = a->rel; ba.n_rel = a->n_rel; ba.ld_rel = a->ld_rel;
    ba.pos = a->pos; ba.B = a->B; ba.codes = codes; ba.margin = a->margin; ba.loss_accum = a->loss_accum;
    ba.contrib_ent = ce; ba.contrib_rel = cr; ba.ldc = L.ldc;
    ba.single_ent = inplace ? single : nullptr; ba.opt = a->opt; ba.step = a->step;
    for (int i = 0; i < 8; ++i) ba.hyper[i] = a->hyper[i];
    ba.ent_state0 = a->ent_state0; ba.ent_state1 = a->ent_state1; ba.tag_ent = a->tag_ent;
    if (factored) { ba.fac_ws_ent = ws + L.ws_ent; ba.fac_ws_ent_bytes = L.ws_ent_bytes; }
    const bool pair_local = a->loss == EMG_LOSS_PAIRWISE || a->loss == EMG_LOSS_NLL || a->loss == EMG_LOSS_ABSOLUTE_MARGIN;
    if (pair_local && !wide && !generic) {
        ba.fused_loss = a->loss;
    } else {  // softmax-coupled losses: scores, then the loss kernel, then backward with external dL/dscore
        rc = emg_train_forward(a->model, a->ent, a->n_ent, a->ld_ent, a->rel, a->n_rel, a->ld_rel, a->k_int, a->scale, a->pos,
                               a->B, et, codes, EMG_SCORE_FINAL, sp, sn, stream);
        if (rc != EMG_OK) return rc;
        rc = emg_loss(a->loss, sp, sn, a->B, a->eta, a->n_sides, a->margin, a->alpha, a->loss_accum, gp, gn, stream);
        if (rc != EMG_OK) return rc;
        ba.fused_loss = -1; ba.g_pos = gp; ba.g_neg = gn;
        if (wide && a->model == EMG_TRANSE_L2) { ba.bw_scores_pos = sp; ba.bw_scores_neg = sn; }   // column blocks need the full norms
    }
    rc = emg_train_backward_ex(&ba, stream);
    if (rc != EMG_OK) return rc;
    emg_apply_args ae{}, ar{};   // both tables through shared launches
    ae.opt = ar.opt = a->opt; ae.k_int = ar.k_int = a->k_int; ae.step = ar.step = a->step; ae.ldc = ar.ldc = L.ldc;
    for (int i = 0; i < 8; ++i) ae.hyper[i] = ar.hyper[i] = a->hyper[i];
    ae.table = a->ent; ae.n_rows = a->n_ent; ae.ld = a->ld_ent; ae.state0 = a->ent_state0; ae.state1 = a->ent_state1;
    ae.tag = a->tag_ent; ae.skip_single = inplace ? 1 : 0; ae.contrib = ce; ae.n_contrib = L.n_ce;
    ae.workspace = ws + L.ws_ent; ae.workspace_bytes = L.ws_ent_bytes; ae.factored = factored;
    ar.table = a->rel; ar.n_rows = a->n_rel; ar.ld = a->ld_rel; ar.state0 = a->rel_state0; ar.state1 = a->rel_state1;
    ar.tag = a->tag_rel; ar.contrib = cr; ar.n_contrib = a->B; ar.workspace = ws + L.ws_rel; ar.workspace_bytes = L.ws_rel_bytes;
    ar.table_index = 1;
    return emg_apply_grouped_pair(&ae, &ar, stream);
}
