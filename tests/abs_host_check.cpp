// abs_host_check.cpp -- host check of the bit-sliced adapter arithmetic (test infrastructure, plain g++).
//
// qcat_amd/csrc/abs_core.h and the generated column programs (abs_generated.inc) are pure functions of 32-bit words,
// so the code the device kernels run is executed here 32 alignments at a time and compared, alignment by alignment,
// with the oracle's scalar DP (oracle/qcat_oracle.c: qo_sg, the restatement of parasail's semi-global alignment with
// the end-position rule R1).  Built and run by tests/test_abs_host.py:
//     g++ -O2 -std=c++17 -I qcat_amd/csrc tests/abs_host_check.cpp -o <tmp>/abs_host_check -L oracle -lqcat_oracle
//     abs_host_check <seed> <rounds>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "abs_core.h"
#include "abs_generated.inc"

extern "C" int qo_sg_rule(const char* s1, int L, const char* s2, int M, int open, int extend, const int8_t* mat, int rule,
                          int32_t* score, int32_t* end_query, int32_t* end_ref);
static int g_rule = 0;                              // rule R1: 0 = QCAT_R1_STRIPED, 1 = QCAT_R1_SCALAR (argv[3]; include/qcat_hip.h)
static int qo_sg(const char* s1, int L, const char* s2, int M, int open, int extend, const int8_t* mat,
                 int32_t* score, int32_t* end_query, int32_t* end_ref) {
    return qo_sg_rule(s1, L, s2, M, open, extend, mat, g_rule, score, end_query, end_ref);
}

using namespace qabs;

static uint64_t g_s;
static uint64_t rnd() { g_s += 0x9E3779B97F4A7C15ull; uint64_t z = g_s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static int below(int n) { return (int)((rnd() >> 33) % (uint64_t)n); }
static const char BASES[] = "ATGC";                 // plane codes 0..3 (qcat_amd/codes.py)

static void adapter_matrix(int8_t* m) {             // qcat/config.py:236-253, [target code * 7 + query code]; codes A T G C N X other
    for (int t = 0; t < 7; ++t)
        for (int q = 0; q < 7; ++q) {
            int v;
            if (t == 6 || q == 6) v = 0;
            else if (t == 5 || q == 5) v = 0;
            else if (t == 4 || q == 4) v = -1;
            else v = t == q ? 5 : -2;
            m[t * 7 + q] = (int8_t)v;
        }
}

static std::string make_window(const std::string* tpls, int nt, int L) {
    std::string w;
    const int kind = below(10);
    if (kind == 0) {                                 // homopolymer / short tandem repeat: ties everywhere
        const int p = 1 + below(3);
        char unit[4];
        for (int i = 0; i < p; ++i) unit[i] = BASES[below(4)];
        for (int i = 0; i < L; ++i) w.push_back(unit[i % p]);
        return w;
    }
    if (kind == 1) {                                 // adapter-free
        for (int i = 0; i < L; ++i) w.push_back(BASES[below(4)]);
        return w;
    }
    const std::string& t = tpls[below(nt)];
    const int lead = below(60);
    for (int i = 0; i < lead; ++i) w.push_back(BASES[below(4)]);
    const int err = below(25);                       // per cent
    const int from = kind == 2 ? below((int)t.size()) : 0;          // sometimes only a suffix of the adapter
    for (size_t j = (size_t)from; j < t.size(); ++j) {
        const char c = t[j] == 'N' ? BASES[below(4)] : t[j];
        if (below(100) < err) {
            const int k = below(3);
            if (k == 0) w.push_back(BASES[below(4)]);
            else if (k == 2) { w.push_back(BASES[below(4)]); w.push_back(c); }
        } else w.push_back(c);
    }
    while ((int)w.size() < L) w.push_back(BASES[below(4)]);
    w.resize((size_t)L);
    return w;
}

template <class P>
static int check_plan(const char* name, const std::string* tpls, int rounds, int L) {
    int8_t mat[49];
    adapter_matrix(mat);
    int bad = 0;
    for (int r = 0; r < rounds; ++r) {
        std::vector<std::string> win(32);
        for (auto& w : win) w = make_window(tpls, P::NT, L);
        // letter planes of the 32 alignments
        std::vector<u32> c1((size_t)L), c0((size_t)L);
        for (int i = 0; i < L; ++i)
            for (int b = 0; b < 32; ++b) {
                const int code = (int)(strchr(BASES, win[(size_t)b][(size_t)i]) - BASES);
                c1[(size_t)i] |= (u32)((code >> 1) & 1) << b;
                c0[(size_t)i] |= (u32)(code & 1) << b;
            }
        static u32 h0[P::NC0][4], h1[P::NC1][4];
        for (int j = 0; j < P::NC0; ++j) abs_set2(h0[j]);
        for (int j = 0; j < P::NC1; ++j) abs_set2(h1[j]);
        AbsBorder bd[P::NT];
        memset(bd, 0, sizeof bd);
        for (int i = 0; i < L; ++i) {
            u32 nq[4], ho[P::NH][4];
            abs_neq_masks(c1[(size_t)i], c0[(size_t)i], nq);
            P::row0(nq, h0, ho);
            P::row1(nq, h1, ho, bd, i == 0 ? 0xFFFFFFFFu : 0u, (unsigned)i);
        }
        AbsLastRow lo[P::NH], lr[P::NT];
        P::last0(h0, lo);
        P::last1(h1, lo, lr);
        for (int t = 0; t < P::NT; ++t) {
            u32 val[ABS_NF + 1], endq[ABS_NI];
            abs_decide(bd[t], lr[t], (unsigned)(L - 1), val, endq, g_rule != 0);
            const int M = (int)tpls[t].size();
            for (int b = 0; b < 32; ++b) {
                int v = 0, e = 0;
                for (int k = 0; k <= ABS_NF; ++k) v |= (int)((val[k] >> b) & 1u) << k;
                for (int k = 0; k < ABS_NI; ++k) e |= (int)((endq[k] >> b) & 1u) << k;
                const int score = v - 2 * M - 1;
                int32_t ws, wq, wr;
                qo_sg(win[(size_t)b].c_str(), L, tpls[t].c_str(), M, 2, 2, mat, &ws, &wq, &wr);
                if (score != ws || e != wq) {
                    if (bad < 10)
                        fprintf(stderr, "%s round %d template %d alignment %d: got (%d, %d), oracle (%d, %d)\n  %s\n", name, r, t, b,
                                score, e, ws, wq, win[(size_t)b].c_str());
                    ++bad;
                }
            }
        }
    }
    printf("%s: %d rounds x 32 alignments x %d template(s), L = %d: %d mismatches\n", name, rounds, P::NT, L, bad);
    return bad;
}

// the interior scan's form of a single-template plan (kernels_abs_mid.inc: abs_mid_body): 32 queries of unequal length, padded
// at the FRONT to the longest (L rows); an alignment that has not started is held at the boundary state after every row, its
// first row forces the border walk, its end position is the tile row minus its padding.  The same calls in the same order
// as the two pipeline stages make them.
template <class P>
static int check_padded(const char* name, const std::string* tpls, int rounds, int L) {
    static_assert(P::NT == 1 && P::NH == 1, "single-template plans");
    int8_t mat[49];
    adapter_matrix(mat);
    int bad = 0;
    for (int r = 0; r < rounds; ++r) {
        std::vector<std::string> win(32);
        int len[32], pz = 0;
        const int spread = r % 3 == 0 ? 0 : (r % 3 == 1 ? 70 : L - 1);          // equal lengths / a length class / anything down to one base
        for (int b = 0; b < 32; ++b) {
            len[b] = b == 0 ? L : L - (spread ? below(spread + 1) : 0);
            win[(size_t)b] = make_window(tpls, 1, len[b]);
            if (L - len[b] > pz) pz = L - len[b];
        }
        std::vector<u32> c1((size_t)L), c0((size_t)L), ns((size_t)L);
        for (int i = 0; i < L; ++i)
            for (int b = 0; b < 32; ++b) {
                const int pad = L - len[b];
                if (i < pad) { ns[(size_t)i] |= 1u << b; if (rnd() & 1) c1[(size_t)i] |= 1u << b; continue; }     // (whatever letters: the hold wipes them)
                const int code = (int)(strchr(BASES, win[(size_t)b][(size_t)(i - pad)]) - BASES);
                c1[(size_t)i] |= (u32)((code >> 1) & 1) << b;
                c0[(size_t)i] |= (u32)(code & 1) << b;
            }
        static u32 h0[P::NC0][4], h1[P::NC1][4];
        for (int j = 0; j < P::NC0; ++j) abs_set2(h0[j]);
        for (int j = 0; j < P::NC1; ++j) abs_set2(h1[j]);
        AbsBorder bd[1];
        memset(bd, 0, sizeof bd);
        u32 prev_ns = 0xFFFFFFFFu;
        for (int i = 0; i < L; ++i) {
            u32 nq[4], ho[1][4];
            const u32 nsm = i < pz ? ns[(size_t)i] : 0u;
            abs_neq_masks(c1[(size_t)i], c0[(size_t)i], nq);
            P::row0(nq, h0, ho);
            if (i < pz) { for (int j = 0; j < P::NC0; ++j) abs_hold2(h0[j], nsm); abs_hold2(ho[0], nsm); }
            const u32 first = prev_ns & ~nsm;
            prev_ns = nsm;
            P::row1(nq, h1, ho, bd, first, (unsigned)i);
            if (i < pz) for (int j = 0; j < P::NC1; ++j) abs_hold2(h1[j], nsm);
        }
        AbsLastRow lo[1], lr[1];
        P::last0(h0, lo);
        P::last1(h1, lo, lr);
        u32 val[ABS_NF + 1], endq[ABS_NI];
        abs_decide(bd[0], lr[0], (unsigned)(L - 1), val, endq, g_rule != 0);
        const int M = (int)tpls[0].size();
        for (int b = 0; b < 32; ++b) {
            int v = 0, e = 0;
            for (int k = 0; k <= ABS_NF; ++k) v |= (int)((val[k] >> b) & 1u) << k;
            for (int k = 0; k < ABS_NI; ++k) e |= (int)((endq[k] >> b) & 1u) << k;
            const int score = v - 2 * M - 1;
            e -= L - len[b];
            int32_t ws, wq, wr;
            qo_sg(win[(size_t)b].c_str(), len[b], tpls[0].c_str(), M, 2, 2, mat, &ws, &wq, &wr);
            if (score != ws || e != wq) {
                if (bad < 10)
                    fprintf(stderr, "%s padded round %d alignment %d (len %d of %d): got (%d, %d), oracle (%d, %d)\n  %s\n", name, r, b, len[b], L,
                            score, e, ws, wq, win[(size_t)b].c_str());
                ++bad;
            }
        }
    }
    printf("%s front-padded: %d rounds x 32 alignments of unequal length, L = %d: %d mismatches\n", name, rounds, L, bad);
    return bad;
}

// a plan of four stages (QAM_*): the stages of a row one after the other, each handing its differences to the next
template <class P>
static int check_multi(const char* name, const std::string* tpls, int rounds, int L) {
    static_assert(P::NS == 4, "four stages");
    typedef typename P::S0 A0; typedef typename P::S1 A1; typedef typename P::S2 A2; typedef typename P::S3 A3;
    int8_t mat[49];
    adapter_matrix(mat);
    int bad = 0;
    for (int r = 0; r < rounds; ++r) {
        std::vector<std::string> win(32);
        for (auto& w : win) w = make_window(tpls, P::NT, L);
        std::vector<u32> c1((size_t)L), c0((size_t)L);
        for (int i = 0; i < L; ++i)
            for (int b = 0; b < 32; ++b) {
                const int code = (int)(strchr(BASES, win[(size_t)b][(size_t)i]) - BASES);
                c1[(size_t)i] |= (u32)((code >> 1) & 1) << b;
                c0[(size_t)i] |= (u32)(code & 1) << b;
            }
        static u32 h0[A0::NC][4], h1[A1::NC][4], h2[A2::NC][4], h3[A3::NC][4];
        for (int j = 0; j < A0::NC; ++j) abs_set2(h0[j]);
        for (int j = 0; j < A1::NC; ++j) abs_set2(h1[j]);
        for (int j = 0; j < A2::NC; ++j) abs_set2(h2[j]);
        for (int j = 0; j < A3::NC; ++j) abs_set2(h3[j]);
        AbsBorder b0[A0::BD], b1[A1::BD], b2[A2::BD], b3[A3::BD];
        memset(b0, 0, sizeof b0); memset(b1, 0, sizeof b1); memset(b2, 0, sizeof b2); memset(b3, 0, sizeof b3);
        for (int i = 0; i < L; ++i) {
            u32 nq[4], x0[A0::HI][4], o0[A0::HO][4], o1[A1::HO][4], o2[A2::HO][4], o3[A3::HO][4];
            const u32 first = i == 0 ? 0xFFFFFFFFu : 0u;
            abs_neq_masks(c1[(size_t)i], c0[(size_t)i], nq);
            A0::row(nq, h0, x0, o0, b0, first, (unsigned)i);
            A1::row(nq, h1, o0, o1, b1, first, (unsigned)i);
            A2::row(nq, h2, o1, o2, b2, first, (unsigned)i);
            A3::row(nq, h3, o2, o3, b3, first, (unsigned)i);
        }
        AbsLastRow y0[A0::HI], l0[A0::HO], l1[A1::HO], l2[A2::HO], l3[A3::HO], r0[A0::BD], r1[A1::BD], r2[A2::BD], r3[A3::BD];
        A0::last(h0, y0, l0, r0);
        A1::last(h1, l0, l1, r1);
        A2::last(h2, l1, l2, r2);
        A3::last(h3, l2, l3, r3);
        AbsBorder bd[2]; AbsLastRow lr[2];
        int seen = 0;
        auto take = [&](int t, const AbsBorder& b, const AbsLastRow& l) { if (t >= 0) { bd[t] = b; lr[t] = l; ++seen; } };
        take(A0::BT0, b0[0], r0[0]); if (A0::NBD > 1) take(A0::BT1, b0[A0::BD - 1], r0[A0::BD - 1]);
        take(A1::BT0, b1[0], r1[0]); if (A1::NBD > 1) take(A1::BT1, b1[A1::BD - 1], r1[A1::BD - 1]);
        take(A2::BT0, b2[0], r2[0]); if (A2::NBD > 1) take(A2::BT1, b2[A2::BD - 1], r2[A2::BD - 1]);
        take(A3::BT0, b3[0], r3[0]); if (A3::NBD > 1) take(A3::BT1, b3[A3::BD - 1], r3[A3::BD - 1]);
        if (seen != P::NT) { fprintf(stderr, "%s: %d borders for %d templates\n", name, seen, P::NT); return 1000; }
        for (int t = 0; t < P::NT; ++t) {
            u32 val[ABS_NF + 1], endq[ABS_NI];
            abs_decide(bd[t], lr[t], (unsigned)(L - 1), val, endq, g_rule != 0);
            const int M = (int)tpls[t].size();
            for (int b = 0; b < 32; ++b) {
                int v = 0, e = 0;
                for (int k = 0; k <= ABS_NF; ++k) v |= (int)((val[k] >> b) & 1u) << k;
                for (int k = 0; k < ABS_NI; ++k) e |= (int)((endq[k] >> b) & 1u) << k;
                const int score = v - 2 * M - 1;
                int32_t ws, wq, wr;
                qo_sg(win[(size_t)b].c_str(), L, tpls[t].c_str(), M, 2, 2, mat, &ws, &wq, &wr);
                if (score != ws || e != wq) {
                    if (bad < 10)
                        fprintf(stderr, "%s round %d template %d alignment %d: got (%d, %d), oracle (%d, %d)\n  %s\n", name, r, t, b,
                                score, e, ws, wq, win[(size_t)b].c_str());
                    ++bad;
                }
            }
        }
    }
    printf("%s: %d rounds x 32 alignments x %d template(s), L = %d: %d mismatches\n", name, rounds, P::NT, L, bad);
    return bad;
}

// the searched cell networks against the reference forms, on every valid input (a, b in 0..9, both letter outcomes)
static int check_cells() {
    int bad = 0;
    for (int neq = 0; neq < 2; ++neq) {
        u32 a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
        // bit (av * 3 + bv / 4 ...) -- simply one (a, b) pair per bit, 100 pairs over four words' worth of passes
        for (int base = 0; base < 100; base += 32) {
            for (int k = 0; k < 4; ++k) a[k] = b[k] = 0;
            for (int bit = 0; bit < 32 && base + bit < 100; ++bit) {
                const int av = (base + bit) / 10, bv = (base + bit) % 10;
                for (int k = 0; k < 4; ++k) { a[k] |= (u32)((av >> k) & 1) << bit; b[k] |= (u32)((bv >> k) & 1) << bit; }
            }
            const u32 live = base + 32 <= 100 ? 0xFFFFFFFFu : ((1u << (100 - base)) - 1u);
            u32 a1[4], b1[4], a2[4], b2[4];
            for (int k = 0; k < 4; ++k) { a1[k] = a2[k] = a[k]; b1[k] = b2[k] = b[k]; }
            abs_cell_letter_ref(neq ? 0xFFFFFFFFu : 0u, a1, b1);
            abs_cell_letter(neq ? 0xFFFFFFFFu : 0u, a2, b2);
            for (int k = 0; k < 4; ++k) bad += ((a1[k] ^ a2[k]) & live) != 0 || ((b1[k] ^ b2[k]) & live) != 0;
            if (neq) {
                for (int k = 0; k < 4; ++k) { a1[k] = a2[k] = a[k]; b1[k] = b2[k] = b[k]; }
                abs_cell_n_ref(a1, b1);
                abs_cell_n(a2, b2);
                for (int k = 0; k < 4; ++k) bad += ((a1[k] ^ a2[k]) & live) != 0 || ((b1[k] ^ b2[k]) & live) != 0;
            }
        }
    }
    printf("cells: searched networks against the reference forms on all valid inputs: %d mismatches\n", bad);
    return bad;
}

struct Seqs { const char* a; const char* b; };

int main(int argc, char** argv) {
    g_s = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
    const int rounds = argc > 2 ? atoi(argv[2]) : 20;
    g_rule = argc > 3 ? atoi(argv[3]) : 0;
    int bad = check_cells();
#include "abs_host_cases.inc"
    return bad ? 1 : 0;
}
