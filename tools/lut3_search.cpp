// lut3_search.cpp -- dev tool: stochastic search for a small network of three-input boolean functions (v_bitop3_b32)
// computing the bit-sliced DP cell of kernels_bitslice.inc:
//     inputs  a (2 bits) = dv + 1 of the left cell, b (2 bits) = dh + 1 of the upper cell, neq = letters differ
//     m = neq ? max(a, b, 1) : 3;   outputs a' = m - b, b' = m - a   (two bits each)
// over the three classes of two-bit encodings of a and b.  Simulated annealing over fan-ins and truth tables of N nodes
// (the four output nodes take the best table for their fan-ins); prints every network that computes all four outputs.
// Found: networks with N = 7 (binary encoding both sides) within seconds; none with N = 6 in 10 minutes on 8 threads.
// The first N = 7 network printed is bs_cell().   build: g++ -O2 -std=c++17 -pthread tools/lut3_search.cpp -o /tmp/lut3
// usage: /tmp/lut3 <nodes> [threads] [iterations per restart]     (runs until killed; wrap in `timeout`)
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include <thread>
#include <mutex>
#include <random>
#include <algorithm>
typedef uint32_t u32;
static const u32 IN[5] = {0xFFFF0000u, 0xFF00FF00u, 0xF0F0F0F0u, 0xCCCCCCCCu, 0xAAAAAAAAu}; // a_hi a_lo b_hi b_lo neq  (pattern index bits 4..0)
static int ENC[3][4] = {{0,1,2,3},{0,1,3,2},{0,3,1,2}};  // value -> code, class representatives
std::mutex mu;
struct Net { int N; int fan[16][3]; uint8_t tab[16]; };
static inline u32 evalnode(const u32* sig, const int* f, uint8_t tab) {
    u32 x = sig[f[0]], y = sig[f[1]], z = sig[f[2]], r = 0;
    for (int m = 0; m < 8; ++m) if (tab >> m & 1) {
        u32 t = (m & 4 ? x : ~x) & (m & 2 ? y : ~y) & (m & 1 ? z : ~z);
        r |= t;
    }
    return r;
}
// best table for target given fanins: returns mismatches, sets tab
static inline int fit(const u32* sig, const int* f, u32 T, uint8_t& tab) {
    u32 x = sig[f[0]], y = sig[f[1]], z = sig[f[2]]; int bad = 0; tab = 0;
    for (int m = 0; m < 8; ++m) {
        u32 t = (m & 4 ? x : ~x) & (m & 2 ? y : ~y) & (m & 1 ? z : ~z);
        int ones = __builtin_popcount(t & T), tot = __builtin_popcount(t);
        if (ones * 2 > tot) { tab |= 1 << m; bad += tot - ones; } else bad += ones;
    }
    return bad;
}
static void targets(int ea, int eb, u32 T[4]) {
    int deca[4], decb[4];
    for (int v = 0; v < 4; ++v) { deca[ENC[ea][v]] = v; decb[ENC[eb][v]] = v; }
    for (int k = 0; k < 4; ++k) T[k] = 0;
    for (int p = 0; p < 32; ++p) {
        int ca = (p >> 3) & 3, cb = (p >> 1) & 3, neq = p & 1;
        int a = deca[ca], b = decb[cb];
        int m = neq ? std::max(std::max(a, b), 1) : 3;
        int an = m - b, bn = m - a;
        int oa = ENC[ea][an], ob = ENC[eb][bn];
        if (oa & 2) T[0] |= 1u << p; if (oa & 1) T[1] |= 1u << p;
        if (ob & 2) T[2] |= 1u << p; if (ob & 1) T[3] |= 1u << p;
    }
}
static int cost(Net& n, const u32 T[4], const int perm[4], u32* sig) {
    for (int i = 0; i < 5; ++i) sig[i] = IN[i];
    int K = n.N - 4, c = 0;
    for (int i = 0; i < K; ++i) sig[5 + i] = evalnode(sig, n.fan[i], n.tab[i]);
    for (int o = 0; o < 4; ++o) {
        int i = K + o;
        c += fit(sig, n.fan[i], T[perm[o]], n.tab[i]);
        sig[5 + i] = evalnode(sig, n.fan[i], n.tab[i]);
    }
    return c;
}
int main(int argc, char** argv) {
    int N = atoi(argv[1]); int nthreads = argc > 2 ? atoi(argv[2]) : 16; long iters = argc > 3 ? atol(argv[3]) : 2000000;
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back([=] {
        std::mt19937_64 rng(1234567 + t * 7919);
        u32 sig[32];
        for (long round = 0;; ++round) {
            int ea = rng() % 3, eb = rng() % 3; u32 T[4]; targets(ea, eb, T);
            int perm[4] = {0,1,2,3}; std::shuffle(perm, perm + 4, rng);
            Net n; n.N = N;
            for (int i = 0; i < N; ++i) { for (int q = 0; q < 3; ++q) n.fan[i][q] = rng() % (5 + i); n.tab[i] = rng(); }
            int c = cost(n, T, perm, sig);
            double temp = 3.0;
            for (long it = 0; it < iters && c > 0; ++it) {
                Net old = n;
                int i = rng() % N, K = N - 4;
                int mv = rng() % 3;
                if (mv == 0 || i >= K) n.fan[i][rng() % 3] = rng() % (5 + i);
                else if (mv == 1) n.tab[i] ^= 1 << (rng() % 8);
                else n.tab[i] = rng();
                int c2 = cost(n, T, perm, sig);
                if (c2 <= c || exp((c - c2) / temp) > (rng() % 1000000) / 1e6) c = c2; else n = old;
                temp = 3.0 * (1.0 - (double)it / iters) + 0.15;
            }
            if (c == 0) {
                std::lock_guard<std::mutex> lk(mu);
                printf("SOLUTION N=%d enc a=%d b=%d perm %d%d%d%d\n", N, ea, eb, perm[0], perm[1], perm[2], perm[3]);
                for (int i = 0; i < N; ++i) printf("  n%d = LUT[%02x](s%d,s%d,s%d)\n", 5 + i, n.tab[i], n.fan[i][0], n.fan[i][1], n.fan[i][2]);
                fflush(stdout);
            }
        }
    });
    for (auto& x : th) x.join();
}
