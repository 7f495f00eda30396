// bs_host_check.cpp -- host check of the bit-sliced BARCODE arithmetic (test infrastructure, plain g++).
//
// qcat_amd/csrc/bs_core.h is a pure function of 32-bit words, so the cell, the counters and the deficit updates the
// device kernels run (kernels_bitslice.inc) are executed here 32 alignments at a time, in the order a k_bs_barcode unit
// walks them -- leading context once, the reversed DP of the trailing context once (round 5: both contexts shared), then
// every barcode's own columns -- and the score of every alignment is compared with the oracle's scalar DP
// (oracle/qcat_oracle.c: qo_sg, the restatement of parasail's semi-global alignment as
// qcat/scanner_base.py:111-117 calls it) on the ORIGINAL orientation of region and target.
// Built and run by tests/test_bs_host.py:
//     g++ -O2 -std=c++17 -I qcat_amd/csrc tests/bs_host_check.cpp -o <tmp>/bs_host_check -L oracle -lqcat_oracle
//     bs_host_check <families file> <seed> <rounds>
// families file: one line per target family  "<reversed> <shared leading columns> <trailing columns> <target> <target> ..."
// (targets in their original orientation: upstream context + barcode + downstream context).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "bs_core.h"

extern "C" int qo_sg(const char* s1, int L, const char* s2, int M, int open, int extend, const int8_t* mat,
                     int32_t* score, int32_t* end_query, int32_t* end_ref);

using namespace qk;

static uint64_t g_s;
static uint64_t rnd() { g_s += 0x9E3779B97F4A7C15ull; uint64_t z = g_s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static int below(int n) { return (int)((rnd() >> 33) % (uint64_t)n); }
static const char BASES[] = "ATGC";                 // plane codes 0..3 (qcat_amd/codes.py, the oracle's mapper)
static int code_of(char c) { return c == 'A' ? 0 : (c == 'T' ? 1 : (c == 'G' ? 2 : 3)); }

static void barcode_matrix(int8_t* m) {             // qcat/config.py:26, scanner_base.py:115: +1 / -1 over A, T, G, C
    for (int t = 0; t < 7; ++t)
        for (int q = 0; q < 7; ++q) m[t * 7 + q] = (int8_t)((t < 4 && t == q) ? 1 : -1);
}

// a barcode region as the scanners cut it: a target (or a piece of one) with errors somewhere in random flanks, a random
// string, or a low-complexity one (ties everywhere)
static std::string make_region(const std::vector<std::string>& targets, int L) {
    std::string w;
    const int kind = below(12);
    if (kind == 0) {
        const int p = 1 + below(3);
        char unit[4];
        for (int i = 0; i < p; ++i) unit[i] = BASES[below(4)];
        for (int i = 0; i < L; ++i) w.push_back(unit[i % p]);
        return w;
    }
    if (kind == 1) {
        for (int i = 0; i < L; ++i) w.push_back(BASES[below(4)]);
        return w;
    }
    const std::string& t = targets[(size_t)below((int)targets.size())];
    const int lead = kind == 2 ? 0 : below(std::max(1, L / 2));
    for (int i = 0; i < lead; ++i) w.push_back(BASES[below(4)]);
    const int err = below(30);                       // per cent
    const int from = kind == 3 ? below((int)t.size()) : 0;           // sometimes only a suffix of the target
    const int to = kind == 4 ? 1 + below((int)t.size()) : (int)t.size();   // ... or only a prefix
    for (int j = from; j < to; ++j) {
        if (below(100) < err) {
            const int k = below(3);
            if (k == 0) w.push_back(BASES[below(4)]);
            else if (k == 2) { w.push_back(BASES[below(4)]); w.push_back(t[(size_t)j]); }
        } else w.push_back(t[(size_t)j]);
    }
    while ((int)w.size() < L) w.push_back(BASES[below(4)]);
    w.resize((size_t)L);
    return w;
}

struct P2 { u32 p1, p0; };

// one unit: 32 regions of L rows against every target of a family.  split: the round-5 form (P shared leading columns, Q
// trailing columns through the reversed DP, C own); otherwise the unsplit form (P shared, C + Q own)
// maxpad > 0 (split form only): alignment k's region is pad[k] <= maxpad rows shorter than the unit and starts pad[k] rows late
// (front padding, bs_hold / bs_keep); the rows in front of it hold arbitrary letters
static int check_unit(const std::vector<std::string>& targets, bool rev, int P, int Q, int L, bool split, const int8_t* mat, int maxpad = 0) {
    const int M = (int)targets[0].size();
    const int C = split ? M - P - Q : M - P;
    std::vector<std::string> reg(32), real(32);
    int pad[32];
    for (int k = 0; k < 32; ++k) {
        pad[k] = maxpad ? below(std::min(maxpad, L - 1) + 1) : 0;
        real[(size_t)k] = make_region(targets, L - pad[k]);
        // the unit's rows in region order: forward sets are padded in front of the region, reversed sets behind it (their rows
        // run backwards, so that is the front of the walk as well)
        std::string junk;
        for (int i = 0; i < pad[k]; ++i) junk.push_back(BASES[below(4)]);
        reg[(size_t)k] = rev ? real[(size_t)k] + junk : junk + real[(size_t)k];
    }
    std::vector<u32> hold((size_t)std::max(1, maxpad), 0u);          // hold[i]: alignments that have not started in walk row i
    for (int i = 0; i < maxpad; ++i)
        for (int k = 0; k < 32; ++k) if (pad[k] > i) hold[(size_t)i] |= 1u << k;
    std::vector<P2> row((size_t)L), dvp((size_t)L), dvr((size_t)L);
    for (int i = 0; i < L; ++i) {                    // letter planes in the order the rows are walked
        u32 c1 = 0, c0 = 0;
        for (int k = 0; k < 32; ++k) {
            const int c = code_of(reg[(size_t)k][(size_t)(rev ? L - 1 - i : i)]);
            c1 |= (u32)((c >> 1) & 1) << k; c0 |= (u32)(c & 1) << k;
        }
        row[(size_t)i] = P2{c1, c0};
    }
    auto walk = [&](const std::string& t) { std::string w = t; if (rev) std::reverse(w.begin(), w.end()); return w; };
    const std::string w0 = walk(targets[0]);
    // leading context (bs_shared_columns): dv + 1 at column P of every row, the last row's counters after P columns
    u32 tail_r[BS_NB], tail_best[BS_NB];
    {
        std::vector<u32> h1((size_t)P, 0u), h0((size_t)P, 0xFFFFFFFFu);
        for (int i = 0; i < L; ++i) {
            u32 a1 = 0u, a0 = 0xFFFFFFFFu;
            for (int j = 0; j < P; ++j) bs_cell(bs_neq_letter(code_of(w0[(size_t)j]), row[(size_t)i].p1, row[(size_t)i].p0), a1, a0, h1[(size_t)j], h0[(size_t)j]);
            if (i < maxpad) for (int j = 0; j < P; ++j) bs_hold(h1[(size_t)j], h0[(size_t)j], hold[(size_t)i]);
            dvp[(size_t)i] = P2{a1, a0};
        }
        for (int q = 0; q < BS_NB; ++q) { tail_r[q] = (BS_OFF >> q) & 1 ? 0xFFFFFFFFu : 0u; tail_best[q] = 0u; }
        for (int j = 0; j < P; ++j) { bs_step(tail_r, h1[(size_t)j], h0[(size_t)j]); bs_max(tail_best, tail_r); }
    }
    // trailing context (bs_trailing_columns): the DP of the reversed region against the reversed context, rows L-1 .. 0
    u32 cmax[BS_NB];
    for (int q = 0; q < BS_NB; ++q) cmax[q] = 0u;
    if (split) {
        std::vector<u32> h1((size_t)Q, 0u), h0((size_t)Q, 0xFFFFFFFFu);
        for (int i = L - 1; i >= 0; --i) {
            u32 a1 = 0u, a0 = 0xFFFFFFFFu;
            const std::vector<u32> o1 = h1, o0 = h0;
            for (int j = 0; j < Q; ++j) bs_cell(bs_neq_letter(code_of(w0[(size_t)(M - 1 - j)]), row[(size_t)i].p1, row[(size_t)i].p0), a1, a0, h1[(size_t)j], h0[(size_t)j]);
            if (i < maxpad) {                          // rows in front of an alignment's first one: its state stays, r = 0
                const u32 hm = hold[(size_t)i];
                for (int j = 0; j < Q; ++j) { bs_keep(h1[(size_t)j], o1[(size_t)j], hm); bs_keep(h0[(size_t)j], o0[(size_t)j], hm); }
                a1 &= ~hm; a0 &= ~hm;
            }
            dvr[(size_t)i] = P2{a1, a0};
        }
        // no trailing column: G(first row - 1) = H(0, M) is no cell -- the first step of an alignment cannot keep it (r = 0 in ITS
        // first row: the row behind its padding)
        if (Q == 0) {
            if (!maxpad) dvr[0] = P2{0u, 0u};
            else for (int i = 0; i < L; ++i) {
                u32 first = 0;
                for (int k = 0; k < 32; ++k) if (pad[k] == i) first |= 1u << k;
                dvr[(size_t)i].p0 &= ~first;
            }
        }
        u32 r[BS_NB];
        for (int q = 0; q < BS_NB; ++q) r[q] = (BS_OFF >> q) & 1 ? 0xFFFFFFFFu : 0u;
        for (int j = 0; j < Q; ++j) { bs_step(r, h1[(size_t)j], h0[(size_t)j]); bs_max(cmax, r); }
    }
    int bad = 0;
    for (const std::string& tgt : targets) {
        const std::string w = walk(tgt);
        std::vector<u32> h1((size_t)C, 0u), h0((size_t)C, 0xFFFFFFFFu);
        u32 f[BS_NF], d[BS_ND];
        for (int q = 0; q < BS_NF; ++q) f[q] = 0u;
        for (int q = 0; q < BS_ND; ++q) d[q] = 0u;
        for (int i = 0; i < L; ++i) {
            u32 a1 = P ? dvp[(size_t)i].p1 : 0u, a0 = P ? dvp[(size_t)i].p0 : 0xFFFFFFFFu;
            for (int j = 0; j < C; ++j) bs_cell(bs_neq_letter(code_of(w[(size_t)(P + j)]), row[(size_t)i].p1, row[(size_t)i].p0), a1, a0, h1[(size_t)j], h0[(size_t)j]);
            if (i < maxpad) for (int j = 0; j < C; ++j) bs_hold(h1[(size_t)j], h0[(size_t)j], hold[(size_t)i]);
            if (split) bs_deficit_split(d, a1, a0, dvr[(size_t)i].p1, dvr[(size_t)i].p0);
            else bs_deficit(f, a1, a0);
        }
        u32 r[BS_NB], best[BS_NB];
        for (int q = 0; q < BS_NB; ++q) { r[q] = tail_r[q]; best[q] = tail_best[q]; }
        for (int j = 0; j < C; ++j) { bs_step(r, h1[(size_t)j], h0[(size_t)j]); bs_max(best, r); }
        if (split) bs_finish_split(best, r, d, cmax);
        else bs_finish(best, r, f);
        for (int k = 0; k < 32; ++k) {
            int got = 0;
            for (int q = 0; q < BS_NB; ++q) got |= (int)((best[q] >> k) & 1u) << q;
            got -= BS_OFF;
            int32_t ws = 0, wq = 0, wr = 0;
            qo_sg(real[(size_t)k].c_str(), L - pad[k], tgt.c_str(), M, 1, 1, mat, &ws, &wq, &wr);
            if (got != ws) {
                if (bad < 5) fprintf(stderr, "MISMATCH %s rev %d P %d Q %d L %d pad %d region %s target %s: got %d want %d\n",
                                     split ? "split" : "unsplit", (int)rev, P, Q, L, pad[k], real[(size_t)k].c_str(), tgt.c_str(), got, ws);
                ++bad;
            }
        }
    }
    return bad;
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: bs_host_check <families file> <seed> <rounds>\n"); return 2; }
    g_s = strtoull(argv[2], nullptr, 10);
    const int rounds = atoi(argv[3]);
    int8_t mat[49];
    barcode_matrix(mat);
    std::ifstream in(argv[1]);
    std::string line;
    int total = 0, fam = 0;
    static const int LENS[] = {1, 2, 3, 24, 46, 47, 48, 99, 150};
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        int rev = 0, P = 0, Q = 0;
        std::vector<std::string> targets;
        std::string t;
        ls >> rev >> P >> Q;
        while (ls >> t) targets.push_back(t);
        if (targets.empty()) continue;
        if (targets.size() > 16) {                    // a sample of a big family per round keeps the run short
            std::vector<std::string> pick;
            for (int i = 0; i < 16; ++i) pick.push_back(targets[(size_t)below((int)targets.size())]);
            targets.swap(pick);
        }
        int bad_split = 0, bad_plain = 0, bad_padded = 0;
        long n = 0;
        for (int r = 0; r < rounds; ++r)
            for (int L : LENS) {
                bad_split += check_unit(targets, rev != 0, P, Q, L, true, mat);
                bad_plain += check_unit(targets, rev != 0, P, 0, L, false, mat);
                if (L > 1) bad_padded += check_unit(targets, rev != 0, P, Q, L, true, mat, 5);
                n += 32 * (long)targets.size();
            }
        printf("family %d (%s, %d shared + %d own + %d trailing columns): %ld alignments each way, split: %d mismatches, unsplit: %d mismatches, front-padded: %d mismatches\n",
               fam, rev ? "reversed" : "forward", P, (int)targets[0].size() - P - Q, Q, n, bad_split, bad_plain, bad_padded);
        total += bad_split + bad_plain + bad_padded;
        ++fam;
    }
    return total ? 1 : 0;
}
