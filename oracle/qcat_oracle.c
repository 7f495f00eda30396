/*
 * qcat_oracle.c -- CPU restatement of qcat's barcode-demultiplexing hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; the product (qcat_amd/, libqcat_hip.so) never does.
 *
 * It follows the reference's Python line by line (citations are into /root/reference):
 *   qo_sg                      parasail.sg_striped_32 as called at qcat/scanner_base.py:111-117,
 *                              :214-218 (algorithm restated, see PARITY below)
 *   qo_sg_stats                parasail.sg_stats_striped_32 (scanner_base.py:168-172): qo_sg + matches / length
 *                              along one optimal path -- tie order of tests/golden/sg_independent.py, parity unpinned
 *   qo_window                  extract_align_sequence      qcat/scanner_base.py:223-244
 *                              + utils.revcomp             qcat/utils.py:20-21
 *   qo_find_best_template      find_best_adapter_template  qcat/scanner_base.py:313-359
 *                              (+ align_adapter :191-220, eval_adapter_template :258-296,
 *                               get_norm_socre :299-310)
 *   qo_region                  extract_barcode_region      qcat/scanner_base.py:29-60
 *   qo_best_barcode            find_highest_scoring_barcode qcat/scanner_base.py:63-141
 *   qo_scan_epi2me             BarcodeScannerEPI2ME.scan   qcat/scanner_epi2me.py:33-144
 *   qo_scan_dual               BarcodeScannerDual.scan     qcat/scanner_dual.py:35-146
 *   qo_detect_barcode          BarcodeScanner.detect_barcode qcat/scanner_base.py:521-604
 *   qo_detect_kit              scan_end/scan_ends/detect_kit qcat/scanner_base.py:618-678
 *
 * PARITY: the DP arithmetic of the reference lives in the third-party library parasail
 * (PyPI "parasail", un-pinned in setup.py:18-23, absent from this image).  qo_sg restates the
 * published semi-global affine recurrence and parasail's striped end-position rule
 * (SURVEY.md section 8a, R1).  It is pinned by the reference's own known answers
 * (tests/test_oracle_reference_vectors.py: test_barcode.py:300-304 end_query == 101, the
 * barcode names of the inline reads, truebc of the 34 shipped FASTQ reads); the end-position
 * TIE rule is not pinned by any reference test -> "parity unpinned" for ties.
 * Everything above the parasail call is pinned against the reference's own Python executed
 * in the authoring container (tests/golden/make_golden.py).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#include "../include/qcat_hip.h"

#define QO_NEG (-(1 << 28))
#define QO_MAXW (1 << 26)       /* longest query qo_sg accepts */

static __thread char qo_err[256];
const char* qo_last_error(void) { return qo_err; }

/* parasail's mapper: alphabet letters in either case -> index, the rest -> '*' */
static uint8_t qo_code_of[256];
static uint8_t qo_comp_of[256];      /* ASCII complement table of utils.revcomp */
static int qo_tables_ready = 0;

static void qo_init_tables(void) {
    if (qo_tables_ready) return;
    const char* alpha = "ATGCNX";
    for (int i = 0; i < 256; ++i) { qo_code_of[i] = QCAT_CODE_OTHER; qo_comp_of[i] = (uint8_t)i; }
    for (int i = 0; alpha[i]; ++i) {
        qo_code_of[(uint8_t)alpha[i]] = (uint8_t)i;
        qo_code_of[(uint8_t)(alpha[i] | 0x20)] = (uint8_t)i;
    }
    const char* from = "ACGTacgtRYMKrymkVBHDvbhd";      /* qcat/utils.py:21 */
    const char* to   = "TGCAtgcaYRKMyrkmBVDHbvdh";
    for (int i = 0; from[i]; ++i) qo_comp_of[(uint8_t)from[i]] = (uint8_t)to[i];
    qo_tables_ready = 1;
}

/* ------------------------------------------------------------------------------------------
 * R1: semi-global alignment, all four end gaps free, gap of length k costs open+(k-1)*extend.
 * q = s1 (read window / region), t = s2 (template / ctx+barcode+ctx); both ASCII.
 * mat[tc*7+qc].  Returns score, end_query, end_ref (0-based, parasail convention).
 * ---------------------------------------------------------------------------------------- */
typedef struct qo_align { int32_t score, end_query, end_ref; } qo_align;

/* rule: QCAT_R1_STRIPED / QCAT_R1_SCALAR (include/qcat_hip.h): which of the reference's two routines
 * (qcat/scanner_base.py:20-26: parasail.sg_striped_32 with SSE2, plain parasail.sg without) places the end */
void qo_sg_codes_rule(const uint8_t* q, int L, const uint8_t* t, int M, int open, int extend,
                      const int8_t* mat, int rule, qo_align* out) {
    int32_t Hrow[QCAT_MAX_TEMPLATE_LEN + 2], Frow[QCAT_MAX_TEMPLATE_LEN + 2];
    int32_t cmax = QO_NEG, cfirst = 0;      /* last-column maximum and the first row reaching it */
    for (int j = 0; j <= M; ++j) { Hrow[j] = 0; Frow[j] = QO_NEG; }
    for (int i = 1; i <= L; ++i) {
        const int8_t* wrow = mat;           /* indexed [tc*7 + qc] */
        int qc = q[i - 1];
        int32_t diag = Hrow[0];             /* H[i-1][0] = 0 */
        int32_t hleft = 0;                  /* H[i][0]   = 0 */
        int32_t e = QO_NEG;                 /* E[i][0] */
        for (int j = 1; j <= M; ++j) {
            int32_t up = Hrow[j];
            int32_t f = Frow[j] - extend;
            if (up - open > f) f = up - open;
            int32_t ee = e - extend;
            if (hleft - open > ee) ee = hleft - open;
            int32_t h = diag + wrow[t[j - 1] * 7 + qc];
            if (ee > h) h = ee;
            if (f > h) h = f;
            diag = up;
            Hrow[j] = h; Frow[j] = f; e = ee; hleft = h;
        }
        if (Hrow[M] > cmax) { cmax = Hrow[M]; cfirst = i; }
    }
    int32_t score = QO_NEG, end_q = L - 1, end_r = 0;
    if (rule == QCAT_R1_SCALAR) {
        /* plain parasail.sg as recalled: the last column was looked at while the rows went by (strict >: the first row
         * reaching its maximum -- cmax / cfirst above), then the last row, target index ascending, strict > */
        score = cmax; end_r = M - 1; end_q = cfirst - 1;
        for (int j = 1; j <= M; ++j) {
            if (Hrow[j] > score) { score = Hrow[j]; end_r = j - 1; end_q = L - 1; }
        }
    } else {
        /* end position: parasail sg_striped rule */
        for (int j = 1; j <= M; ++j) {          /* row "query fully consumed", strict > */
            if (Hrow[j] > score) { score = Hrow[j]; end_r = j - 1; end_q = L - 1; }
        }
        if (cmax > score || (cmax == score && end_r == M - 1)) {
            score = cmax; end_r = M - 1; end_q = cfirst - 1;
        }
    }
    out->score = score; out->end_query = end_q; out->end_ref = end_r;
}
void qo_sg_codes(const uint8_t* q, int L, const uint8_t* t, int M, int open, int extend,
                 const int8_t* mat, qo_align* out) {
    qo_sg_codes_rule(q, L, t, M, open, extend, mat, QCAT_R1_STRIPED, out);
}

/* ASCII front end (used by the parasail stand-in of tests/golden/make_golden.py) */
int qo_sg_rule(const char* s1, int L, const char* s2, int M, int open, int extend,
               const int8_t* mat, int rule, int32_t* score, int32_t* end_query, int32_t* end_ref);
int qo_sg(const char* s1, int L, const char* s2, int M, int open, int extend,
          const int8_t* mat, int32_t* score, int32_t* end_query, int32_t* end_ref) {
    return qo_sg_rule(s1, L, s2, M, open, extend, mat, QCAT_R1_STRIPED, score, end_query, end_ref);
}
int qo_sg_rule(const char* s1, int L, const char* s2, int M, int open, int extend,
               const int8_t* mat, int rule, int32_t* score, int32_t* end_query, int32_t* end_ref) {
    qo_init_tables();
    if (rule != QCAT_R1_STRIPED && rule != QCAT_R1_SCALAR) {
        snprintf(qo_err, sizeof qo_err, "qo_sg: unknown R1 rule %d", rule);
        return QCAT_ERR_ARG;
    }
    if (L <= 0 || M <= 0 || L > QO_MAXW || M > QCAT_MAX_TEMPLATE_LEN) {
        snprintf(qo_err, sizeof qo_err, "qo_sg: bad lengths %d x %d", L, M);
        return QCAT_ERR_ARG;
    }
    uint8_t* q = (uint8_t*)malloc((size_t)L + (size_t)M);
    uint8_t* t = q + L;
    for (int i = 0; i < L; ++i) q[i] = qo_code_of[(uint8_t)s1[i]];
    for (int j = 0; j < M; ++j) t[j] = qo_code_of[(uint8_t)s2[j]];
    qo_align a;
    qo_sg_codes_rule(q, L, t, M, open, extend, mat, rule, &a);
    free(q);
    *score = a.score; *end_query = a.end_query; *end_ref = a.end_ref;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * parasail.sg_stats_striped_32 as called at qcat/scanner_base.py:168-172 (align_adapter_identity) and :108-117
 * (find_highest_scoring_barcode with compute_identity): the alignment of qo_sg plus the number of exact matches and of
 * alignment columns along ONE optimal path, chosen by `rule` (include/qcat_hip.h QCAT_STATS_*; ONE switch shared with the
 * device kernel k_sg_align and tests/golden/sg_independent.py sg_stats):
 *   PARASAIL6 / PARASAIL5 -- parasail 2.x's *_stats_striped_* kernels as recalled: H = max(H_dag, E, F), then
 *       case1 = (H == H_dag), case2 = (H == F), HM = case1 ? H_dagM + match : (case2 ? FM : EM): on ties the diagonal, then
 *       F (the gap that consumes a QUERY letter), then E; `match` = equality of the MAPPED codes (alphabet ATGCNX + '*',
 *       or ATGCN + '*' where X maps to '*' as well: the barcode matrix), so two different letters outside the alphabet
 *       count as a match; a gap is opened only when strictly better than extended (case = opn > ext);
 *   ROUND3 -- diagonal, E, F and "the same letter" (what round 3 shipped).
 * PARITY UNPINNED for `matches` / `length` under every rule: parasail is absent here and no reference test holds the two
 * numbers.  Nothing on any scanner path consumes them (scanner_base.py:141 returns the score in their place).
 * ---------------------------------------------------------------------------------------- */
typedef struct qo_stats { int32_t score, end_query, end_ref, matches, length; } qo_stats;

int qo_sg_stats_rule(const char* s1, int L, const char* s2, int M, int open, int extend, const int8_t* mat, int rule_and_r1, qo_stats* out) {
    qo_init_tables();
    const int r1_scalar = (rule_and_r1 & QCAT_SG_R1_SCALAR) != 0;     /* as qcat_sg_align's with_stats: QCAT_STATS_* | QCAT_SG_R1_SCALAR */
    const int rule = rule_and_r1 & ~QCAT_SG_R1_SCALAR;
    if (rule != QCAT_STATS_PARASAIL6 && rule != QCAT_STATS_PARASAIL5 && rule != QCAT_STATS_ROUND3) {
        snprintf(qo_err, sizeof qo_err, "qo_sg_stats: unknown rule %d", rule);
        return QCAT_ERR_ARG;
    }
    if (L <= 0 || M <= 0 || L > QO_MAXW || M > QCAT_MAX_TEMPLATE_LEN) {
        snprintf(qo_err, sizeof qo_err, "qo_sg_stats: bad lengths %d x %d", L, M);
        return QCAT_ERR_ARG;
    }
    /* per target column j: H, the gap state that consumes query letters (from the row above) and their statistics */
    int32_t H[QCAT_MAX_TEMPLATE_LEN + 2], F[QCAT_MAX_TEMPLATE_LEN + 2];
    int32_t HM[QCAT_MAX_TEMPLATE_LEN + 2], HL[QCAT_MAX_TEMPLATE_LEN + 2], FM[QCAT_MAX_TEMPLATE_LEN + 2], FL[QCAT_MAX_TEMPLATE_LEN + 2];
    for (int j = 0; j <= M; ++j) { H[j] = 0; F[j] = QO_NEG; HM[j] = HL[j] = FM[j] = FL[j] = 0; }
    int32_t cmax = QO_NEG, cfirst = 0, cm = 0, cl = 0;
    for (int i = 1; i <= L; ++i) {
        const int qc = qo_code_of[(uint8_t)s1[i - 1]];
        int32_t diag = H[0], dm = HM[0], dl = HL[0];           /* cell (i-1, 0): 0 with empty statistics */
        int32_t hleft = 0, hlm = 0, hll = 0;                     /* cell (i, 0) */
        int32_t e = QO_NEG, em = 0, el = 0;                      /* gap that consumes target letters, along the row */
        for (int j = 1; j <= M; ++j) {
            const int tc = qo_code_of[(uint8_t)s2[j - 1]];
            /* gap consuming a target letter: from (i, j-1) */
            if (hleft - open > e - extend) { e = hleft - open; em = hlm; el = hll + 1; }
            else { e = e - extend; el = el + 1; }
            /* gap consuming a query letter: from (i-1, j) */
            int32_t f, fm, fl;
            if (H[j] - open > F[j] - extend) { f = H[j] - open; fm = HM[j]; fl = HL[j] + 1; }
            else { f = F[j] - extend; fm = FM[j]; fl = FL[j] + 1; }
            int32_t h = diag + mat[tc * 7 + qc];
            const int qm = (rule == QCAT_STATS_PARASAIL5 && qc == QCAT_CODE_X) ? QCAT_CODE_OTHER : qc;
            const int tm = (rule == QCAT_STATS_PARASAIL5 && tc == QCAT_CODE_X) ? QCAT_CODE_OTHER : tc;
            const int same = rule == QCAT_STATS_ROUND3 ? (qc == tc && ((s1[i - 1] | 0x20) == (s2[j - 1] | 0x20))) : (qm == tm);
            int32_t hm = dm + (same ? 1 : 0);
            int32_t hl = dl + 1;
            if (rule == QCAT_STATS_ROUND3) {
                if (e > h) { h = e; hm = em; hl = el; }
                if (f > h) { h = f; hm = fm; hl = fl; }
            } else {
                if (f > h) { h = f; hm = fm; hl = fl; }
                if (e > h) { h = e; hm = em; hl = el; }
            }
            diag = H[j]; dm = HM[j]; dl = HL[j];
            H[j] = h; HM[j] = hm; HL[j] = hl; F[j] = f; FM[j] = fm; FL[j] = fl;
            hleft = h; hlm = hm; hll = hl;
        }
        if (H[M] > cmax) { cmax = H[M]; cfirst = i; cm = HM[M]; cl = HL[M]; }
    }
    int32_t score = QO_NEG, end_q = L - 1, end_r = 0, mm = 0, ll = 0;
    for (int j = 1; j <= M; ++j)
        if (H[j] > score) { score = H[j]; end_r = j - 1; end_q = L - 1; mm = HM[j]; ll = HL[j]; }
    if (cmax > score || (cmax == score && (end_r == M - 1 || r1_scalar))) { score = cmax; end_r = M - 1; end_q = cfirst - 1; mm = cm; ll = cl; }
    out->score = score; out->end_query = end_q; out->end_ref = end_r; out->matches = mm; out->length = ll;
    return 0;
}

int qo_sg_stats(const char* s1, int L, const char* s2, int M, int open, int extend, const int8_t* mat, qo_stats* out) {
    return qo_sg_stats_rule(s1, L, s2, M, open, extend, mat, QCAT_STATS_PARASAIL6, out);
}

/* ------------------------------------------------------------------------------------------
 * Prepared kit: codes of templates, contexts, targets, denominators.
 * ---------------------------------------------------------------------------------------- */
typedef struct qo_set {
    int n, blen, tlen, uplen, downlen;
    uint8_t* targets;          /* n * tlen codes: up + barcode + down */
    const int32_t* ids;
    const int32_t* lens;       /* simple mode, a user FASTA: barcode b has lens[b] <= blen letters (NULL: all blen) */
} qo_set;

typedef struct qo_tpl {
    int len, trim_offset, is_double;
    uint8_t codes[QCAT_MAX_TEMPLATE_LEN];
    int bc_start[2], bc_end[2], bc_len[2];
    int den;                   /* get_norm_socre denominator, scanner_base.py:308-310 */
    qo_set sets[2];
} qo_tpl;

typedef struct qo_kit {
    qcat_kit_desc d;
    int nt;
    qo_tpl tpl[QCAT_MAX_TEMPLATES];
} qo_kit;

void qo_kit_free(qo_kit* k) {
    if (!k) return;
    for (int t = 0; t < k->nt; ++t)
        for (int s = 0; s < 2; ++s) free(k->tpl[t].sets[s].targets);
    free(k);
}

int qo_kit_prepare(const qcat_kit_desc* d, qo_kit** out) {
    qo_init_tables();
    if (d && d->mode == QCAT_MODE_SIMPLE && (d->n_templates != 1 || d->templates[0].length != 0)) {
        snprintf(qo_err, sizeof qo_err, "simple mode takes one empty template holding the barcode list");
        return QCAT_ERR_ARG;
    }
    if (!d || d->abi_version != QCAT_ABI_VERSION || d->n_templates < 0 ||
        d->n_templates > QCAT_MAX_TEMPLATES) {
        snprintf(qo_err, sizeof qo_err, "qo_kit_prepare: bad descriptor");
        return QCAT_ERR_ARG;
    }
    qo_kit* k = (qo_kit*)calloc(1, sizeof *k);
    k->d = *d; k->nt = d->n_templates;
    for (int t = 0; t < k->nt; ++t) {
        const qcat_template_desc* s = &d->templates[t];
        qo_tpl* p = &k->tpl[t];
        const int simple = d->mode == QCAT_MODE_SIMPLE;
        if ((simple ? s->length != 0 : s->length <= 0) || s->length > QCAT_MAX_TEMPLATE_LEN) {
            snprintf(qo_err, sizeof qo_err, "template %d: bad length %d", t, s->length);
            qo_kit_free(k); return QCAT_ERR_ARG;
        }
        p->len = s->length; p->trim_offset = s->trim_offset; p->is_double = s->is_double_barcode;
        for (int j = 0; j < p->len; ++j) p->codes[j] = qo_code_of[(uint8_t)s->sequence[j]];
        int bc_total = 0;
        for (int i = 0; i < 2; ++i) {
            p->bc_start[i] = s->bc_start[i]; p->bc_end[i] = s->bc_end[i]; p->bc_len[i] = s->bc_len[i];
            bc_total += s->bc_len[i];
        }
        p->den = simple ? 1 : (p->len - bc_total) * d->match + bc_total * d->nmatch;
        if (p->den == 0) {
            snprintf(qo_err, sizeof qo_err, "template %d: zero normalisation denominator", t);
            qo_kit_free(k); return QCAT_ERR_ARG;
        }
        for (int i = 0; i < 2; ++i) {
            const qcat_barcode_set_desc* bs = &s->sets[i];
            qo_set* q = &p->sets[i];
            q->n = bs->n; q->blen = bs->barcode_len; q->ids = bs->ids; q->lens = NULL;
            if (bs->n <= 0) { q->n = 0; continue; }
            if (bs->lengths) {                                   /* ABI 4: barcodes of unequal length, simple mode only */
                int ragged = 0;
                for (int b = 0; b < bs->n; ++b) {
                    if (bs->lengths[b] < 1 || bs->lengths[b] > bs->barcode_len) {
                        snprintf(qo_err, sizeof qo_err, "template %d set %d: lengths[%d] outside 1..barcode_len", t, i, b);
                        qo_kit_free(k); return QCAT_ERR_ARG;
                    }
                    ragged = ragged || bs->lengths[b] != bs->barcode_len;
                }
                if (ragged && d->mode != QCAT_MODE_SIMPLE) {
                    snprintf(qo_err, sizeof qo_err, "barcodes of unequal length are covered in simple mode only");
                    qo_kit_free(k); return QCAT_ERR_UNSUPPORTED;
                }
                if (ragged) q->lens = bs->lengths;
            }
            /* contexts: layout.py:191-238 */
            int n = d->barcode_context_length, up0 = 0, up1 = 0, dn0 = 0, dn1 = 0;
            if (p->bc_end[i] > -1) {
                up0 = p->bc_start[i] - n; if (up0 < 0) up0 = 0;
                up1 = p->bc_start[i]; if (up1 < up0) up1 = up0;   /* n < 0 would slice empty */
                dn0 = p->bc_end[i] + 1;
                dn1 = p->bc_end[i] + n + 1; if (dn1 > p->len) dn1 = p->len;
                if (dn1 < dn0) dn1 = dn0;
            }
            q->uplen = up1 - up0; q->downlen = dn1 - dn0;
            q->tlen = q->uplen + q->blen + q->downlen;
            if (q->tlen <= 0 || q->tlen > QCAT_MAX_TARGET_LEN) {
                snprintf(qo_err, sizeof qo_err, "template %d set %d: bad target length %d", t, i, q->tlen);
                qo_kit_free(k); return QCAT_ERR_ARG;
            }
            q->targets = (uint8_t*)malloc((size_t)q->n * q->tlen);
            for (int b = 0; b < q->n; ++b) {
                uint8_t* dst = q->targets + (size_t)b * q->tlen;
                for (int j = 0; j < q->uplen; ++j) dst[j] = p->codes[up0 + j];
                for (int j = 0; j < q->blen; ++j)
                    dst[q->uplen + j] = qo_code_of[(uint8_t)bs->sequences[(size_t)b * q->blen + j]];
                for (int j = 0; j < q->downlen; ++j) dst[q->uplen + q->blen + j] = p->codes[dn0 + j];
            }
        }
        if (p->sets[0].n == 0 || (d->mode == QCAT_MODE_DUAL && p->sets[1].n == 0)) {
            /* the reference would iterate over None (TypeError) */
            snprintf(qo_err, sizeof qo_err, "template %d lacks a barcode set required by the mode", t);
            qo_kit_free(k); return QCAT_ERR_ARG;
        }
    }
    *out = k;
    return 0;
}

/* extract_align_sequence (scanner_base.py:223-244): codes of read[:n] or revcomp(read[-n:]) */
static int qo_window(const uint8_t* read, int64_t len, int rev, int n, uint8_t* w) {
    int L = (len < n) ? (int)len : n;
    if (!rev) {
        for (int i = 0; i < L; ++i) w[i] = qo_code_of[read[i]];
    } else {
        const uint8_t* tail = read + (len - L);
        for (int i = 0; i < L; ++i) w[i] = qo_code_of[qo_comp_of[tail[L - 1 - i]]];
    }
    return L;
}

typedef struct qo_best_tpl { int idx, end; double score; int raw; } qo_best_tpl;

/* find_best_adapter_template, scanner_base.py:313-359 */
static qo_best_tpl qo_find_best_template(const qo_kit* k, const uint8_t* w, int L,
                                         qcat_end_trace* tr) {
    qo_best_tpl b = { -1, -1, -1.0, -1 };
    if (k->nt == 0 || L == 0) return b;
    for (int t = 0; t < k->nt; ++t) {
        const qo_tpl* p = &k->tpl[t];
        qo_align a;
        qo_sg_codes_rule(w, L, p->codes, p->len, k->d.gap_open, k->d.gap_extend, k->d.adapter_matrix, k->d.r1_rule, &a);
        if (tr) { tr->tpl_raw[t] = a.score; tr->tpl_end[t] = a.end_query; }
        double norm = a.score * 100.0 / (double)p->den;
        if (b.score < norm) { b.score = norm; b.idx = t; b.end = a.end_query; b.raw = a.score; }
    }
    return b;
}

/* extract_barcode_region, scanner_base.py:29-60, with Python slice semantics. */
static void qo_region(const qo_kit* k, const qo_tpl* p, int set, int stop, int L,
                      int* start_out, int* len_out) {
    int ext = k->d.extracted_barcode_extension;
    int end_ref = stop - (p->len - p->bc_end[set]) + 1;
    int start_ref = end_ref - p->bc_len[set];
    start_ref -= (ext < start_ref) ? ext : start_ref;
    end_ref += (ext < L - end_ref) ? ext : (L - end_ref);
    /* read_sequence[start_ref : end_ref + 1] */
    int a = start_ref, b = end_ref + 1;
    if (a < 0) { a += L; if (a < 0) a = 0; }
    if (b < 0) { b += L; if (b < 0) b = 0; }
    if (a > L) a = L;
    if (b > L) b = L;
    *start_out = a;
    *len_out = (b > a) ? (b - a) : 0;
}

/* find_highest_scoring_barcode, scanner_base.py:63-141.  Returns index or -1 (None);
 * *raw = raw score of the winner.  All targets of a set have one length, so the float
 * comparison `max_score < score` is the integer comparison of raw scores, and
 * `not max_score` is (None or raw == 0) -- rule R2. */
static int qo_best_barcode(const qo_kit* k, const qo_set* s, const uint8_t* region, int rl,
                           int* raw_out, int16_t* row) {
    *raw_out = 0;
    if (rl <= 0) return -1;
    int best = -1, best_raw = 0;
    for (int b = 0; b < s->n; ++b) {
        qo_align a;
        qo_sg_codes(region, rl, s->targets + (size_t)b * s->tlen, s->tlen, 1, 1,
                    k->d.barcode_matrix, &a);
        if (row) row[b] = (int16_t)a.score;
        if (best < 0 || best_raw == 0 || best_raw < a.score) { best = b; best_raw = a.score; }
    }
    *raw_out = best_raw;
    return best;
}

/* one scan() result; mirrors build_return_dict (scanner_base.py:362-390) */
typedef struct qo_scan {
    int has_barcode;        /* barcode is not None */
    int bc[2];              /* indices into set 0 / set 1 */
    int raw, den;           /* barcode_score = raw*100.0/den (den = 1, raw = 0 when score is 0) */
    double score;
    int adapter;            /* template index or -1 (None) */
    int adapter_end;
    int exit_status;
} qo_scan;

static qo_scan qo_empty(void) {          /* empty_return_dict, scanner_base.py:393-407 */
    qo_scan r; memset(&r, 0, sizeof r);
    r.bc[0] = r.bc[1] = -1; r.den = 1; r.adapter = -1; r.exit_status = 1;
    return r;
}

static qo_scan qo_scan_end(const qo_kit* k, const uint8_t* w, int L, qcat_end_trace* tr,
                           int16_t* rows, uint32_t stride) {
    qo_best_tpl bt = qo_find_best_template(k, w, L, tr);
    int used = bt.idx < 0 ? k->nt + bt.idx : bt.idx;          /* Python list[-1] */
    const qo_tpl* p = &k->tpl[used];
    int dual = k->d.mode == QCAT_MODE_DUAL;
    int start[2] = {0, 0}, len[2] = {0, 0}, bc[2] = {-1, -1}, raw[2] = {0, 0};
    int region_path = dual || bt.score > k->d.region_min_adapter_score || p->is_double;
    if (region_path) qo_region(k, p, 0, bt.end, L, &start[0], &len[0]);
    else { start[0] = 0; len[0] = L < k->d.max_align_length ? L : k->d.max_align_length; }
    bc[0] = qo_best_barcode(k, &p->sets[0], w + start[0], len[0], &raw[0], rows);
    if (dual) {
        qo_region(k, p, 1, bt.end, L, &start[1], &len[1]);
        bc[1] = qo_best_barcode(k, &p->sets[1], w + start[1], len[1], &raw[1],
                                rows ? rows + stride : NULL);
    }
    /* (epi2me also scans set 1 of a double-barcode template and discards the result,
     *  scanner_epi2me.py:104-131 -- no observable effect, not restated) */
    qo_scan r; memset(&r, 0, sizeof r);
    r.bc[0] = r.bc[1] = -1; r.den = 1;
    if (!dual) {
        int ae = bt.end + p->trim_offset;                      /* scanner_epi2me.py:135-137 */
        if (ae > L) ae = L;
        r.has_barcode = bc[0] >= 0;
        r.bc[0] = bc[0];
        if (bc[0] >= 0) { r.raw = raw[0]; r.den = p->sets[0].tlen; r.score = raw[0] * 100.0 / (1.0 * r.den); }
        r.adapter = used; r.adapter_end = ae; r.exit_status = 0;
    } else if (bc[0] >= 0 && bc[1] >= 0) {                     /* scanner_dual.py:130-144 */
        double s0 = raw[0] * 100.0 / (1.0 * p->sets[0].tlen);
        double s1 = raw[1] * 100.0 / (1.0 * p->sets[1].tlen);
        r.has_barcode = 1; r.bc[0] = bc[0]; r.bc[1] = bc[1];
        if (s1 < s0) { r.raw = raw[1]; r.den = p->sets[1].tlen; r.score = s1; }   /* min(a,b) */
        else { r.raw = raw[0]; r.den = p->sets[0].tlen; r.score = s0; }
        r.adapter = used; r.adapter_end = bt.end; r.exit_status = 0;
    } else {
        r = qo_empty();
    }
    if (tr) {
        tr->window_len = L; tr->best_tpl = bt.idx; tr->best_end = bt.end; tr->best_raw = bt.raw;
        tr->used_tpl = used; tr->region_path = region_path;
        for (int i = 0; i < 2; ++i) {
            tr->region_start[i] = start[i]; tr->region_len[i] = len[i];
            tr->bc_idx[i] = bc[i]; tr->bc_raw[i] = raw[i];
        }
        tr->adapter_end = r.adapter_end;
    }
    return r;
}

/* BarcodeScannerSimple.scan, qcat/scanner_simple.py:41-91: find_highest_scoring_barcode(window, self.barcodes,
 * compute_identity=True) -- every barcode against the WHOLE window, no contexts.  The function returns
 * (max_barcode, q_score, max_score, max_end) (scanner_base.py:141), so the value scan() calls `identity` and
 * compares with min_quality is the normalised SCORE; the matches/length statistics of sg_stats never leave
 * find_highest_scoring_barcode.  adapter is None, adapter_end = end_query of the winner's alignment. */
static qo_scan qo_scan_simple(const qo_kit* k, const uint8_t* w, int L, qcat_end_trace* tr, int16_t* rows) {
    const qo_set* s = &k->tpl[0].sets[0];
    int best = -1, best_raw = 0, best_end = -1, best_len = 1;
    double best_score = 0.0;
    if (L > 0) {
        for (int b = 0; b < s->n; ++b) {
            qo_align a;
            const int tl = s->lens ? s->lens[b] : s->tlen;       /* every barcode with its own length (:112-117) */
            qo_sg_codes(w, L, s->targets + (size_t)b * s->tlen, tl, 1, 1, k->d.barcode_matrix, &a);
            if (rows) rows[b] = (int16_t)a.score;
            const double sc = a.score * 100.0 / (1.0 * tl);      /* scanner_base.py:119 */
            /* `if not max_score or max_score < score` (:125) */
            if (best < 0 || best_score == 0.0 || best_score < sc) { best = b; best_raw = a.score; best_end = a.end_query; best_len = tl; best_score = sc; }
        }
    }
    qo_scan r = qo_empty();
    double score = best >= 0 ? best_raw * 100.0 / (1.0 * best_len) : 0.0;
    if (!(score < k->d.min_quality)) {                       /* `if identity < self.min_quality: return empty` */
        r.has_barcode = best >= 0; r.bc[0] = best; r.raw = best >= 0 ? best_raw : 0; r.den = best >= 0 ? best_len : 1;
        r.score = score; r.adapter = -1; r.adapter_end = best_end; r.exit_status = 0;
    }
    if (tr) {
        tr->window_len = L; tr->best_tpl = -1; tr->best_end = best_end; tr->best_raw = -1; tr->used_tpl = 0;
        tr->region_path = 0; tr->region_start[0] = 0; tr->region_len[0] = L;
        tr->bc_idx[0] = best; tr->bc_raw[0] = best_raw; tr->bc_idx[1] = -1;
        tr->adapter_end = r.adapter_end;
    }
    return r;
}

static int qo_same_id(const qo_kit* k, const qo_scan* a, const qo_scan* b) {
    /* barcode.id equality; dual ids are "id1/id2" strings (scanner_dual.py:131-134) */
    for (int s = 0; s < 2; ++s) {
        if (a->bc[s] < 0 && b->bc[s] < 0) continue;
        if (a->bc[s] < 0 || b->bc[s] < 0) return 0;
        int ia = k->tpl[a->adapter < 0 ? 0 : a->adapter].sets[s].ids[a->bc[s]];     /* (simple mode: adapter None) */
        int ib = k->tpl[b->adapter < 0 ? 0 : b->adapter].sets[s].ids[b->bc[s]];
        if (ia != ib) return 0;
    }
    return 1;
}

/* scan_middle, qcat/scanner_base.py:479-519: scan() (epi2me: scanner_epi2me.py:33-144, dual:
 * scanner_dual.py:35-146) of the read interior read[n:-n] and of its reverse complement with the
 * templates of one kit (get_adapters, :606-611); True as soon as one strand reaches barcode_score
 * >= middle_min_score. */
static double qo_scan_seq_score(const qo_kit* k, const uint8_t* w, int L, int kit_slot) {
    int sub[QCAT_MAX_TEMPLATES], ns = 0;
    for (int t = 0; t < k->nt; ++t) if (k->d.templates[t].kit_slot == kit_slot) sub[ns++] = t;
    qo_best_tpl b = { -1, -1, -1.0, -1 };
    if (ns > 0 && L > 0) {
        for (int i = 0; i < ns; ++i) {
            const qo_tpl* p = &k->tpl[sub[i]];
            qo_align a;
            qo_sg_codes_rule(w, L, p->codes, p->len, k->d.gap_open, k->d.gap_extend, k->d.adapter_matrix, k->d.r1_rule, &a);
            double norm = a.score * 100.0 / (double)p->den;
            if (b.score < norm) { b.score = norm; b.idx = i; b.end = a.end_query; b.raw = a.score; }
        }
    }
    const qo_tpl* p = &k->tpl[sub[b.idx < 0 ? ns + b.idx : b.idx]];
    const int dual = k->d.mode == QCAT_MODE_DUAL;
    int start, len, raw0 = 0, raw1 = 0, bc0, bc1 = -1;
    if (dual || b.score > k->d.region_min_adapter_score || p->is_double) qo_region(k, p, 0, b.end, L, &start, &len);
    else { start = 0; len = L < k->d.max_align_length ? L : k->d.max_align_length; }
    bc0 = qo_best_barcode(k, &p->sets[0], w + start, len, &raw0, NULL);
    if (!dual) return bc0 >= 0 ? raw0 * 100.0 / (1.0 * p->sets[0].tlen) : 0.0;
    qo_region(k, p, 1, b.end, L, &start, &len);
    bc1 = qo_best_barcode(k, &p->sets[1], w + start, len, &raw1, NULL);
    if (bc0 < 0 || bc1 < 0) return 0.0;
    double s0 = raw0 * 100.0 / (1.0 * p->sets[0].tlen), s1 = raw1 * 100.0 / (1.0 * p->sets[1].tlen);
    return s1 < s0 ? s1 : s0;
}

static int qo_scan_middle(const qo_kit* k, const uint8_t* read, int64_t len, int kit_slot) {
    const int n = k->d.max_align_length;
    int64_t m = len - 2 * (int64_t)n;                  /* len(sequence[n:-n]) */
    if (m < 0) m = 0;
    uint8_t* w = (uint8_t*)malloc((size_t)m + 1);
    int hit = 0;
    for (int strand = 0; strand < 2 && !hit; ++strand) {
        for (int64_t i = 0; i < m; ++i)
            w[i] = strand == 0 ? qo_code_of[read[n + i]] : qo_code_of[qo_comp_of[read[len - n - 1 - i]]];
        if (!(qo_scan_seq_score(k, w, (int)m, kit_slot) < k->d.middle_min_score)) hit = 1;
    }
    free(w);
    return hit;
}

static void qo_to_record(const qo_scan* s, int trim5, int64_t trim3, qcat_result* o) {
    o->barcode_idx = (int16_t)s->bc[0]; o->barcode2_idx = (int16_t)s->bc[1];
    o->adapter_idx = (int16_t)s->adapter; o->exit_status = (int16_t)s->exit_status;
    o->adapter_end = s->adapter_end; o->trim5p = trim5; o->trim3p = (int32_t)trim3;
    o->raw_score = (int16_t)s->raw; o->score_den = (int16_t)s->den;
}

/* detect_barcode, scanner_base.py:521-604 (R6) */
static void qo_detect_barcode(const qo_kit* k, const uint8_t* read, int64_t len,
                              qcat_result* o, qcat_end_trace* tr, int16_t* rows, uint32_t stride) {
    uint8_t w[QCAT_MAX_WINDOW];
    int n = k->d.max_align_length;
    const int simple = k->d.mode == QCAT_MODE_SIMPLE;
    int L = qo_window(read, len, 0, n, w);
    qo_scan r5 = simple ? qo_scan_simple(k, w, L, tr, rows) : qo_scan_end(k, w, L, tr, rows, stride);
    if (k->d.ends == QCAT_ENDS_5P) {           /* config 2: scan() of the 5' window only */
        qo_to_record(&r5, 0, 0, o);
        return;
    }
    int trim5 = r5.adapter_end > 0 ? r5.adapter_end : 0;
    if (r5.score < k->d.min_quality) r5 = qo_empty();
    L = qo_window(read, len, 1, n, w);
    qo_scan r3 = simple ? qo_scan_simple(k, w, L, tr ? tr + 1 : NULL, rows ? rows + 2 * stride : NULL)
                        : qo_scan_end(k, w, L, tr ? tr + 1 : NULL, rows ? rows + 2 * stride : NULL, stride);
    int64_t trim3 = len;
    if (r3.adapter >= 0 && r3.adapter_end > 0) trim3 -= r3.adapter_end;
    if (r3.score < k->d.min_quality) r3 = qo_empty();

    const qo_scan* best = NULL; double bs = 0.0;
    if (r5.score > bs) { bs = r5.score; best = &r5; }
    if (r3.score > bs) { bs = r3.score; best = &r3; }
    qo_scan res;
    if (!best) res = qo_empty();
    else {
        res = *best;
        if (r5.has_barcode && r3.has_barcode &&
            r5.score >= k->d.conflict_min_score && r3.score >= k->d.conflict_min_score &&
            !qo_same_id(k, &r5, &r3)) {
            res = qo_empty(); res.exit_status = 1002;
        }
    }
    if (k->d.scan_middle_adapter && res.adapter >= 0 &&
        qo_scan_middle(k, read, len, k->d.templates[res.adapter].kit_slot)) {
        res = qo_empty(); res.exit_status = 997;             /* scanner_base.py:593-595 */
    }
    if (trim3 < trim5) trim5 = 0;
    qo_to_record(&res, trim5, trim3, o);
}

static void qo_count(const qo_kit* k, const qcat_result* r, int64_t len, int64_t* counts) {
    int nb = k->d.n_barcode_slots, nk = k->d.n_kit_slots;
    int nbuckets = (k->d.mode == QCAT_MODE_DUAL) ? nb * nb : nb;
    if (k->d.min_read_length > 0) {                         /* the driver's filter, qcat/cli.py:521-534 */
        if (k->d.trim_reads && k->d.ends != QCAT_ENDS_5P) { /* sequence = sequence[trim_5p:trim_3p] */
            int64_t a = r->trim5p < len ? r->trim5p : len, b = r->trim3p < len ? r->trim3p : len;
            len = b > a ? b - a : 0;
        }
        if (len < k->d.min_read_length) { counts[nbuckets + 1 + nk + 1] += 1; return; }   /* skipped_reads += 1; continue */
    }
    int slot = nbuckets;                                    /* "none" */
    if (r->barcode_idx >= 0 && (r->adapter_idx >= 0 || k->d.mode == QCAT_MODE_SIMPLE)) {
        const qo_tpl* p = &k->tpl[r->adapter_idx < 0 ? 0 : r->adapter_idx];
        slot = p->sets[0].ids[r->barcode_idx];
        if (k->d.mode == QCAT_MODE_DUAL) slot = slot * nb + p->sets[1].ids[r->barcode2_idx];
    }
    counts[slot] += 1;
    int kslot = nk;
    if (r->adapter_idx >= 0) kslot = k->d.templates[r->adapter_idx].kit_slot;
    counts[nbuckets + 1 + kslot] += 1;
}

int qo_count_buckets(const qcat_kit_desc* d) {
    int nb = d->n_barcode_slots;
    return ((d->mode == QCAT_MODE_DUAL) ? nb * nb : nb) + 1 + d->n_kit_slots + 1 + 1;   /* .., [skipped] */
}

/* Batch entry point with the product's signature (minus the device context). */
int qo_scan_debug(const qcat_kit_desc* d, const uint8_t* bases, const uint64_t* offsets,
                  uint32_t n_reads, qcat_result* out, int64_t* counts,
                  qcat_end_trace* traces, int16_t* bc_rows, uint32_t row_stride, int threads) {
    qo_kit* k = NULL;
    int rc = qo_kit_prepare(d, &k);
    if (rc) return rc;
    if (d->max_align_length <= 0 || d->max_align_length > QCAT_MAX_WINDOW) {
        snprintf(qo_err, sizeof qo_err, "max_align_length %d outside 1..%d", d->max_align_length, QCAT_MAX_WINDOW);
        qo_kit_free(k); return QCAT_ERR_ARG;
    }
    int ends = d->ends == QCAT_ENDS_5P ? 1 : 2;
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads)
    for (int64_t r = 0; r < (int64_t)n_reads; ++r) {
        const uint8_t* read = bases + offsets[r];
        int64_t len = (int64_t)(offsets[r + 1] - offsets[r]);
        qcat_end_trace* tr = traces ? traces + (size_t)r * ends : NULL;
        int16_t* rows = bc_rows ? bc_rows + (size_t)r * ends * 2 * row_stride : NULL;
        if (tr) memset(tr, 0, sizeof(*tr) * ends);
        qo_detect_barcode(k, read, len, &out[r], tr, rows, row_stride);
    }
    if (counts) for (uint32_t r = 0; r < n_reads; ++r) qo_count(k, &out[r], (int64_t)(offsets[r + 1] - offsets[r]), counts);
    qo_kit_free(k);
    return 0;
}

int qo_scan_batch(const qcat_kit_desc* d, const uint8_t* bases, const uint64_t* offsets,
                  uint32_t n_reads, qcat_result* out, int64_t* counts, int threads) {
    return qo_scan_debug(d, bases, offsets, n_reads, out, counts, NULL, NULL, 0, threads);
}

/* BarcodeScanner.scan() of whole sequences of any length (scanner_epi2me.py:33-144 / scanner_dual.py:35-146):
 * the checker of qcat_scan_sequences.  One record per sequence, exactly what scan() returns. */
int qo_scan_sequences(const qcat_kit_desc* d, const uint8_t* bases, const uint64_t* offsets,
                      uint32_t n_seqs, qcat_result* out) {
    qo_kit* k = NULL;
    int rc = qo_kit_prepare(d, &k);
    if (rc) return rc;
    for (uint32_t r = 0; r < n_seqs; ++r) {
        const uint8_t* seq = bases + offsets[r];
        int64_t len = (int64_t)(offsets[r + 1] - offsets[r]);
        uint8_t* w = (uint8_t*)malloc((size_t)len + 1);
        for (int64_t i = 0; i < len; ++i) w[i] = qo_code_of[seq[i]];
        qo_scan s = d->mode == QCAT_MODE_SIMPLE ? qo_scan_simple(k, w, (int)len, NULL, NULL) : qo_scan_end(k, w, (int)len, NULL, NULL, 0);
        qo_to_record(&s, 0, 0, &out[r]);
        free(w);
    }
    qo_kit_free(k);
    return 0;
}

/* detect_kit (scanner_base.py:662-678) over the descriptor's templates: per read, scan_ends
 * (:632-642) -> the template of the higher-scoring end (3' on ties) gets one vote.
 * votes[n_templates] receives per-TEMPLATE votes (the host folds templates onto kit names),
 * votes[n_templates] itself is unused ("none" cannot occur: scan_end always indexes a layout). */
int qo_detect_kit_votes(const qcat_kit_desc* d, const uint8_t* bases, const uint64_t* offsets,
                        uint32_t n_reads, int64_t* votes, int32_t* per_read) {
    qo_kit* k = NULL;
    int rc = qo_kit_prepare(d, &k);
    if (rc) return rc;
    uint8_t w[QCAT_MAX_WINDOW];
    int n = d->max_align_length;
    for (uint32_t r = 0; r < n_reads; ++r) {
        const uint8_t* read = bases + offsets[r];
        int64_t len = (int64_t)(offsets[r + 1] - offsets[r]);
        int L = qo_window(read, len, 0, n, w);
        qo_best_tpl b5 = qo_find_best_template(k, w, L, NULL);
        L = qo_window(read, len, 1, n, w);
        qo_best_tpl b3 = qo_find_best_template(k, w, L, NULL);
        int i5 = b5.idx < 0 ? k->nt + b5.idx : b5.idx;
        int i3 = b3.idx < 0 ? k->nt + b3.idx : b3.idx;
        int pick = (b5.score > b3.score) ? i5 : i3;
        votes[pick] += 1;
        if (per_read) per_read[r] = pick;
    }
    qo_kit_free(k);
    return 0;
}
