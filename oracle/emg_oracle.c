/* emg_oracle.c — plain-C CPU restatement of the Emgraph hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product
 * (emgraph_amd) never links or executes it.
 *
 * Two jobs:
 *  (1) the CANONICAL-ORDER oracle for ranks: every 1-vs-all score is the float32 chain
 *      acc = fmaf(q_k, e_k, acc), k ascending from +0 (TransE-L1: acc + |q_k - e_k|;
 *      L2: fmaf(d,d,acc) then -sqrtf), the query vectors use the same pinned arithmetic as
 *      emgraph_amd/csrc/emg_rank.hip, so ranks are comparable BIT-EXACTLY at sizes the numpy
 *      oracle cannot reach.  Semantics restated from the reference:
 *        generate_corruptions_for_eval   emgraph/evaluation/protocol.py:448-528
 *        eval scoring + side split       emgraph/models/EmbeddingModel.py:1856-1892
 *        perform_comparision             emgraph/models/EmbeddingModel.py:1989-2033
 *        filter correction               emgraph/models/EmbeddingModel.py:1894-1986
 *      It is itself validated against the literal numpy restatement (oracle/emgraph_oracle.py,
 *      pinned on the reference's goldens) in tests/test_host_logic.py::test_c_oracle_matches_numpy_oracle.
 *  (2) the timed CPU baseline ("port"): fused gather+score of a training batch
 *      (EmbeddingModel.py:675-677,788-799; TransE.py:208-216, DistMult.py:201, ComplEx.py:288-298,
 *      HolE.py:189) with OpenMP over triples.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -march=x86-64-v3 -fopenmp).
 * -ffp-contract=off + explicit fmaf() keeps the arithmetic exactly as written.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { TRANSE_L1 = 0, TRANSE_L2 = 1, DISTMULT = 2, COMPLEX_ = 3, HOLE = 4 };
enum { EVAL_S = 0, EVAL_O = 1, EVAL_SPO = 2, EVAL_S_O = 3 };

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ Philox4x32-10 */
static void philox(uint32_t c[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c[0], p1 = (uint64_t)M1 * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += W0; k1 += W1;
    }
}

/* codes[j] = replacement | keep_subject<<31   (protocol.py:598-641 with our Philox draws) */
void orc_corrupt_codes(int64_t B, int32_t eta, int side, int64_t n_choices, const int32_t* entities_list,
                       uint64_t seed, uint64_t counter, int32_t* codes) {
    for (int64_t j = 0; j < B * eta; ++j) {
        uint32_t c[4] = {(uint32_t)j, (uint32_t)((uint64_t)j >> 32), (uint32_t)counter, (uint32_t)(counter >> 32)};
        philox(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        uint32_t keep = c[0] & 1u;
        const uint64_t r64 = ((uint64_t)c[2] << 32) | c[1];
        const uint32_t idx = (uint32_t)(((unsigned __int128)r64 * (uint64_t)n_choices) >> 64);
        if (side == 1) keep = 1u; else if (side == 0) keep = 0u;
        const uint32_t repl = entities_list ? (uint32_t)entities_list[idx] : idx;
        codes[j] = (int32_t)((repl & 0x7fffffffu) | (keep << 31));
    }
}

/* ------------------------------------------------------------------ training-batch scoring (CPU baseline) */
static inline float score_rows(int model, const float* a, const float* p, const float* b, int k_int, float scale) {
    float acc = 0.f;
    if (model == TRANSE_L1) {
        for (int c = 0; c < k_int; ++c) acc += fabsf((a[c] + p[c]) - b[c]);
        return -acc;
    }
    if (model == TRANSE_L2) {
        for (int c = 0; c < k_int; ++c) { const float d = (a[c] + p[c]) - b[c]; acc += d * d; }
        return -sqrtf(acc);
    }
    if (model == DISTMULT) {
        for (int c = 0; c < k_int; ++c) acc += (a[c] * p[c]) * b[c];
        return acc;
    }
    const int k = k_int / 2;
    float s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f; /* ComplEx.py:293-297: four separate sums */
    for (int c = 0; c < k; ++c) {
        const float sr = a[c], si = a[k + c], pr = p[c], pi = p[k + c], orr = b[c], oi = b[k + c];
        s1 += (pr * sr) * orr; s2 += (pr * si) * oi; s3 += (pi * sr) * oi; s4 += (pi * si) * orr;
    }
    acc = ((s1 + s2) + s3) - s4;
    return model == HOLE ? scale * acc : acc;
}

/* scores of B positives and their eta*B code-defined negatives (eta-major) */
void orc_train_forward(int model, const float* ent, int64_t ld_ent, const float* rel, int64_t ld_rel, int32_t k_int,
                       float scale, const int32_t* pos, int64_t B, int32_t eta, const int32_t* codes,
                       float* scores_pos, float* scores_neg) {
#pragma omp parallel for schedule(static)
    for (int64_t g = 0; g < B; ++g) {
        const float* rs = ent + (int64_t)pos[3 * g] * ld_ent;
        const float* rp = rel + (int64_t)pos[3 * g + 1] * ld_rel;
        const float* ro = ent + (int64_t)pos[3 * g + 2] * ld_ent;
        scores_pos[g] = score_rows(model, rs, rp, ro, k_int, scale);
        for (int j = 0; j < eta; ++j) {
            const int32_t code = codes[(int64_t)j * B + g];
            const float* re = ent + (int64_t)(code & 0x7fffffff) * ld_ent;
            scores_neg[(int64_t)j * B + g] =
                code < 0 ? score_rows(model, rs, rp, re, k_int, scale) : score_rows(model, re, rp, ro, k_int, scale);
        }
    }
}

/* ------------------------------------------------------------------ canonical-order 1-vs-all */
static inline void row_to_query(int64_t r, int64_t n_q, int side_mode, int64_t* qi, int* obj) {
    if (side_mode == EVAL_S) { *qi = r; *obj = 0; }
    else if (side_mode == EVAL_O) { *qi = r; *obj = 1; }
    else { *obj = r < n_q; *qi = r < n_q ? r : r - n_q; }
}

void orc_build_queries(int model, const float* ent, int64_t ld_ent, const float* rel, int64_t ld_rel, int32_t k_int,
                       const int32_t* test, int64_t n_q, int side_mode, float* Q, int64_t ldq) {
    const int cplx = (model == COMPLEX_ || model == HOLE);
    const int n = cplx ? k_int / 2 : k_int;
    const int64_t n_rows = side_mode >= EVAL_SPO ? 2 * n_q : n_q;
    for (int64_t r = 0; r < n_rows; ++r) {
        int64_t qi; int obj;
        row_to_query(r, n_q, side_mode, &qi, &obj);
        const float* x = ent + (int64_t)(obj ? test[3 * qi] : test[3 * qi + 2]) * ld_ent;
        const float* pr = rel + (int64_t)test[3 * qi + 1] * ld_rel;
        float* q = Q + r * ldq;
        for (int c = 0; c < n; ++c) {
            if (model <= TRANSE_L2) q[c] = obj ? x[c] + pr[c] : x[c] - pr[c];
            else if (model == DISTMULT) q[c] = pr[c] * x[c];
            else {
                const float p_r = pr[c], p_i = pr[n + c], x_r = x[c], x_i = x[n + c];
                if (obj) { q[c] = fmaf(p_r, x_r, -(p_i * x_i)); q[n + c] = fmaf(p_r, x_i, p_i * x_r); }
                else     { q[c] = fmaf(p_r, x_r, p_i * x_i);    q[n + c] = fmaf(p_r, x_i, -(p_i * x_r)); }
            }
        }
    }
}

float orc_chain_score(int model, const float* q, const float* e, int32_t k_int, float scale) {
    float acc = 0.f;
    if (model == TRANSE_L1) { for (int k = 0; k < k_int; ++k) acc = acc + fabsf(q[k] - e[k]); return -acc; }
    if (model == TRANSE_L2) {
        for (int k = 0; k < k_int; ++k) { const float d = q[k] - e[k]; acc = fmaf(d, d, acc); }
        return -sqrtf(acc);
    }
    for (int k = 0; k < k_int; ++k) acc = fmaf(q[k], e[k], acc);
    return model == HOLE ? acc * scale : acc;
}

static inline int32_t cmp_int(float s) { return (int32_t)(s * 100000.0f); } /* EmbeddingModel.py:2010-2014 */

void orc_pos_int(int model, const float* ent, int64_t ld_ent, int32_t k_int, float scale, const int32_t* test,
                 int64_t n_q, int side_mode, const float* Q, int64_t ldq, int32_t* pos_int) {
    const int64_t n_rows = side_mode >= EVAL_SPO ? 2 * n_q : n_q;
    for (int64_t r = 0; r < n_rows; ++r) {
        int64_t qi; int obj;
        row_to_query(r, n_q, side_mode, &qi, &obj);
        const int32_t tgt = obj ? test[3 * qi + 2] : test[3 * qi];
        pos_int[r] = cmp_int(orc_chain_score(model, Q + r * ldq, ent + (int64_t)tgt * ld_ent, k_int, scale));
    }
}

void orc_scores_dense(int model, const float* Q, int64_t ldq, int64_t n_rows, const float* ent, int64_t n_cand,
                      int64_t ld_ent, const int32_t* cand, int32_t k_int, float scale, float* S, int64_t lds) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n_rows; ++r)
        for (int64_t e = 0; e < n_cand; ++e)
            S[r * lds + e] = orc_chain_score(model, Q + r * ldq, ent + (int64_t)(cand ? cand[e] : e) * ld_ent, k_int, scale);
}

/* counts of candidates with cmp_int > / == the positive's, per query row */
void orc_count(int model, const float* Q, int64_t ldq, const int32_t* pos_int, int64_t n_rows, const float* ent,
               int64_t n_cand, int64_t ld_ent, const int32_t* cand, int32_t k_int, float scale, int32_t* cnt_gt,
               int32_t* cnt_eq) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t r = 0; r < n_rows; ++r) {
        int32_t gt = 0, eq = 0;
        const int32_t p = pos_int[r];
        for (int64_t e = 0; e < n_cand; ++e) {
            const int32_t ci =
                cmp_int(orc_chain_score(model, Q + r * ldq, ent + (int64_t)(cand ? cand[e] : e) * ld_ent, k_int, scale));
            gt += ci > p; eq += ci == p;
        }
        cnt_gt[r] += gt; cnt_eq[r] += eq;
    }
}

void orc_filter_count(int model, const float* Q, int64_t ldq, const int32_t* pos_int, int64_t n_rows, const float* ent,
                      int64_t n_local, int64_t ld_ent, int64_t ent_offset, int32_t k_int, float scale,
                      const int64_t* fptr, const int32_t* fidx, int32_t* fgt, int32_t* feq) {
    for (int64_t r = 0; r < n_rows; ++r) {
        const int32_t p = pos_int[r];
        for (int64_t u = fptr[r]; u < fptr[r + 1]; ++u) {
            const int64_t e = (int64_t)fidx[u] - ent_offset;
            if (e < 0 || e >= n_local) continue;
            const int32_t ci = cmp_int(orc_chain_score(model, Q + r * ldq, ent + e * ld_ent, k_int, scale));
            fgt[r] += ci > p; feq[r] += ci == p;
        }
    }
}
