// synth.h -- stateless synthetic-read generator, shared by the device
// generator kernels and the host entry point qcat_synth_read().  tests/synth.py is the Python
// twin; tests/test_synth.py checks the three agree.
//
//   read = lead + M(fill(T5p)) + insert + M(revcomp(fill(T3p))) + tail
//
// Every read is a pure function of (seed, index, parameters, templates): SplitMix64 seeded with
// seed ^ (index * 0xD1342543DE82EF95).  The generator is written as a push-style state machine:
// `Sink::put(c)` either counts (length pass) or stores (write pass).
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define QS_HD __host__ __device__
#else
#define QS_HD
#endif

namespace qsynth {

struct Rng {
    uint64_t s;
    QS_HD Rng(uint64_t seed, uint64_t index) : s(seed ^ (index * 0xD1342543DE82EF95ull)) {}
    QS_HD uint64_t next() {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    QS_HD uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
    QS_HD uint32_t u24() { return (uint32_t)(next() >> 40); }
};

// filled template handed to the generator: ASCII, N-runs already replaced per barcode choice is
// done on the fly from these pieces
struct Tpl {
    const char* seq;        // template, upper-case ATGCNX
    int len;
    int bc_start[2], bc_len[2];
    const char* sets[2];    // barcode ASCII blobs (n * bc_len) or nullptr
    int n[2];
};

struct Params {
    uint64_t seed;
    uint32_t insert_len, lead_min, lead_max;
    uint32_t thr_err, thr_none;     // 24-bit thresholds: (uint32_t)(rate * 16777216.0f)
};

QS_HD inline char base_of(uint32_t k) { return "ACGT"[k & 3]; }
QS_HD inline int index_of(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }
QS_HD inline char comp_of(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c; }

// character p of fill(T): the template with N-run k replaced by barcode (b_k % n_k)
QS_HD inline char filled_at(const Tpl& t, int p, uint32_t b, uint32_t b2) {
    for (int k = 0; k < 2; ++k) {
        if (t.n[k] > 0 && t.bc_len[k] > 0 && p >= t.bc_start[k] && p < t.bc_start[k] + t.bc_len[k]) {
            uint32_t idx = (k == 0 ? b : b2) % (uint32_t)t.n[k];
            return t.sets[k][(size_t)idx * t.bc_len[k] + (p - t.bc_start[k])];
        }
    }
    return t.seq[p];
}

template <class Sink>
QS_HD inline void mutate_put(Rng& r, char c, uint32_t thr, Sink& out) {
    if (r.u24() < thr) {
        uint32_t kind = r.below(3);
        if (kind == 0) {
            int k = index_of(c);
            uint32_t x = r.below(3);
            out.put(k >= 0 ? base_of((uint32_t)k + 1 + x) : base_of(x));
        } else if (kind == 1) {
            // deletion
        } else {
            out.put(base_of(r.below(4)));
            out.put(c);
        }
    } else {
        out.put(c);
    }
}

template <class Sink>
QS_HD inline void generate(const Params& p, uint64_t index, const Tpl* t5, const Tpl* t3, Sink& out) {
    Rng r(p.seed, index);
    bool bare = r.u24() < p.thr_none;
    uint32_t span = p.lead_max - p.lead_min + 1;
    uint32_t lead = p.lead_min + r.below(span);
    uint32_t tail = p.lead_min + r.below(span);
    uint32_t b = r.below(1u << 16);
    uint32_t b2 = r.below(1u << 16);
    for (uint32_t i = 0; i < lead; ++i) out.put(base_of(r.below(4)));
    if (!bare && t5)
        for (int j = 0; j < t5->len; ++j) mutate_put(r, filled_at(*t5, j, b, b2), p.thr_err, out);
    for (uint32_t i = 0; i < p.insert_len; ++i) out.put(base_of(r.below(4)));
    if (!bare && t3)
        for (int j = t3->len - 1; j >= 0; --j) mutate_put(r, comp_of(filled_at(*t3, j, b, b2)), p.thr_err, out);
    for (uint32_t i = 0; i < tail; ++i) out.put(base_of(r.below(4)));
}

struct CountSink { uint64_t n = 0; QS_HD void put(char) { ++n; } };
struct StoreSink {
    uint8_t* dst; uint64_t n = 0, cap;
    QS_HD StoreSink(uint8_t* d, uint64_t c) : dst(d), cap(c) {}
    QS_HD void put(char c) { if (n < cap) dst[n] = (uint8_t)c; ++n; }
};

}  // namespace qsynth
