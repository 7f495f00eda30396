/*
 * qcat_cpu_abi.c -- libqcat_cpu.so: the SAME C ABI as libqcat_hip.so (include/qcat_hip.h), implemented on the
 * CPU by the oracle (qcat_oracle.c).  SURVEY.md 8b: "the same ABI is implemented twice".
 *
 * TEST INFRASTRUCTURE ONLY, like the rest of oracle/: it lets one ctypes test body run unchanged against both
 * libraries (tests/test_abi_twin.py).  The product never loads it and has no CPU fallback.
 *
 * Only the host-buffer entry points exist here (kit handles, contexts as empty tokens, qcat_scan_batch /
 * _debug / _sequences, qcat_detect_kit, the count-bucket query); everything that names a device -- resident
 * batches, streams, timing, RCCL -- is absent on purpose.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#include "../include/qcat_hip.h"

/* the oracle's entry points (qcat_oracle.c, compiled into this library) */
const char* qo_last_error(void);
int qo_count_buckets(const qcat_kit_desc* d);
int qo_scan_debug(const qcat_kit_desc* d, const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads, qcat_result* out,
                  int64_t* counts, qcat_end_trace* traces, int16_t* bc_rows, uint32_t row_stride, int threads);
int qo_scan_sequences(const qcat_kit_desc* d, const uint8_t* bases, const uint64_t* offsets, uint32_t n_seqs, qcat_result* out);
int qo_detect_kit_votes(const qcat_kit_desc* d, const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads,
                        int64_t* votes, int32_t* per_read);

static __thread char cpu_err[256];

struct qcat_kit {                      /* a deep copy of the descriptor: the caller's buffers may go away */
    qcat_kit_desc d;
    qcat_template_desc tpl[QCAT_MAX_TEMPLATES];
    char* blob;                        /* template sequences, barcode sequences, id arrays */
};
struct qcat_ctx { int device; int threads; };

const char* qcat_last_error(void) { return cpu_err[0] ? cpu_err : qo_last_error(); }
int qcat_abi_version(void) { return QCAT_ABI_VERSION; }
const char* qcat_backend(void) { return "cpu-oracle"; }      /* (the product's loader refuses anything but "hip": qcat_amd/native.py) */
int qcat_device_count(void) { return 0; }

int qcat_kit_create(const qcat_kit_desc* desc, qcat_kit** out) {
    cpu_err[0] = 0;
    if (!desc || !out || desc->n_templates < 0 || desc->n_templates > QCAT_MAX_TEMPLATES) {
        snprintf(cpu_err, sizeof cpu_err, "qcat_kit_create: bad descriptor");
        return QCAT_ERR_ARG;
    }
    if (desc->abi_version != QCAT_ABI_VERSION) {
        snprintf(cpu_err, sizeof cpu_err, "kit descriptor ABI version mismatch");
        return QCAT_ERR_ARG;
    }
    size_t need = 0;
    for (int t = 0; t < desc->n_templates; ++t) {
        const qcat_template_desc* s = &desc->templates[t];
        need += (size_t)(s->length > 0 ? s->length : 0) + 8;
        for (int i = 0; i < 2; ++i)
            if (s->sets[i].n > 0) need += (size_t)s->sets[i].n * (size_t)s->sets[i].barcode_len + (size_t)s->sets[i].n * 8 + 24;
    }
    qcat_kit* k = (qcat_kit*)calloc(1, sizeof *k);
    k->blob = (char*)calloc(1, need + 16);
    k->d = *desc;
    k->d.templates = k->tpl;
    char* p = k->blob;
    for (int t = 0; t < desc->n_templates; ++t) {
        const qcat_template_desc* s = &desc->templates[t];
        k->tpl[t] = *s;
        if (s->length > 0) { memcpy(p, s->sequence, (size_t)s->length); }
        k->tpl[t].sequence = p; p += (s->length > 0 ? s->length : 0) + 1;
        for (int i = 0; i < 2; ++i) {
            if (s->sets[i].n <= 0) continue;
            const size_t nb = (size_t)s->sets[i].n * (size_t)s->sets[i].barcode_len;
            memcpy(p, s->sets[i].sequences, nb);
            k->tpl[t].sets[i].sequences = p; p += nb + 1;
            p = (char*)(((uintptr_t)p + 7) & ~(uintptr_t)7);
            memcpy(p, s->sets[i].ids, (size_t)s->sets[i].n * 4);
            k->tpl[t].sets[i].ids = (const int32_t*)p; p += (size_t)s->sets[i].n * 4;
            if (s->sets[i].lengths) {                                /* (ABI 4: per-barcode lengths travel with the kit) */
                memcpy(p, s->sets[i].lengths, (size_t)s->sets[i].n * 4);
                k->tpl[t].sets[i].lengths = (const int32_t*)p; p += (size_t)s->sets[i].n * 4;
            }
        }
    }
    /* validate by one empty scan: the oracle checks the descriptor when it prepares the kit */
    uint64_t off0 = 0;
    int rc = qo_scan_debug(&k->d, (const uint8_t*)"", &off0, 0, NULL, NULL, NULL, NULL, 0, 1);
    if (rc) { free(k->blob); free(k); return rc; }
    *out = k;
    return 0;
}

void qcat_kit_destroy(qcat_kit* k) { if (k) { free(k->blob); free(k); } }
int qcat_kit_count_buckets(const qcat_kit* k) { return k ? qo_count_buckets(&k->d) : QCAT_ERR_ARG; }

int qcat_ctx_create(int device, qcat_ctx** out) {
    cpu_err[0] = 0;
    if (!out) return QCAT_ERR_ARG;
    qcat_ctx* c = (qcat_ctx*)calloc(1, sizeof *c);
    c->device = device;
    const char* t = getenv("QCAT_CPU_THREADS");
    c->threads = t ? atoi(t) : 1;
    *out = c;
    return 0;
}
void qcat_ctx_destroy(qcat_ctx* c) { free(c); }

int qcat_scan_debug(qcat_ctx* c, const qcat_kit* k, const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads,
                    qcat_result* out, int64_t* counts, qcat_end_trace* traces, int16_t* bc_rows, uint32_t row_stride) {
    cpu_err[0] = 0;
    if (!c || !k || !offsets || !out) { snprintf(cpu_err, sizeof cpu_err, "null argument"); return QCAT_ERR_ARG; }
    return qo_scan_debug(&k->d, bases, offsets, n_reads, out, counts, traces, bc_rows, row_stride, c->threads);
}

int qcat_scan_batch(qcat_ctx* c, const qcat_kit* k, const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads,
                    qcat_result* out, int64_t* counts) {
    return qcat_scan_debug(c, k, bases, offsets, n_reads, out, counts, NULL, NULL, 0);
}

int qcat_scan_sequences(qcat_ctx* c, const qcat_kit* k, const uint8_t* bases, const uint64_t* offsets, uint32_t n_seqs,
                        qcat_result* out) {
    cpu_err[0] = 0;
    if (!c || !k || !offsets || !out) { snprintf(cpu_err, sizeof cpu_err, "null argument"); return QCAT_ERR_ARG; }
    return qo_scan_sequences(&k->d, bases, offsets, n_seqs, out);
}

int qcat_detect_kit(qcat_ctx* c, const qcat_kit* k, const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads,
                    int64_t* votes, int64_t* first_read) {
    cpu_err[0] = 0;
    if (!c || !k || !offsets || !votes || !first_read) { snprintf(cpu_err, sizeof cpu_err, "null argument"); return QCAT_ERR_ARG; }
    int32_t* per_read = (int32_t*)malloc(((size_t)n_reads + 1) * 4);
    int64_t tmp[QCAT_MAX_TEMPLATES + 1] = {0};
    int rc = qo_detect_kit_votes(&k->d, bases, offsets, n_reads, tmp, per_read);
    if (!rc) {
        for (int t = 0; t < k->d.n_templates; ++t) { votes[t] += tmp[t]; first_read[t] = n_reads; }
        for (uint32_t r = n_reads; r-- > 0;) first_read[per_read[r]] = r;
    }
    free(per_read);
    return rc;
}
