// lut3_search_adapter.cpp -- dev tool: local search for a smaller network of three-input boolean functions
// (v_bitop3_b32) computing the bit-sliced ADAPTER cell of abs_core.h:
//     inputs  a (4 bits, 0..9), b (4 bits, 0..9), neq;   m = neq ? max(a, b, W) : 9   (W = 2 letter column; the N column
//     has no neq input and m = max(a, b, 3));   outputs a' = m - b, b' = m - a  (4 bits each)
// Unlike the barcode cell (tools/lut3_search.cpp: 5 inputs, found from random starts) this function has 9 inputs and
// the hand-made network 27 nodes: the search starts FROM that network, deletes one node (its readers are re-wired to one
// of its inputs) and anneals fan-ins and truth tables at a low temperature until the outputs are exact again on all
// valid input patterns (a, b <= 9), then goes on from the smaller network.  Every exact network found is printed.
// build: g++ -O2 -std=c++17 -pthread tools/lut3_search_adapter.cpp -o /tmp/lut3a
// usage: /tmp/lut3a <letter|n> [threads] [iterations per attempt] [log of an earlier run to continue from] [seed]
//        (runs until killed; wrap in `timeout`)
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

constexpr int NW = 32;                                  // up to 1024 patterns (adapter cells use the first 512: a << 5 | b << 1 | neq)
struct Sig { uint32_t w[NW]; };
struct Node { int f[3]; uint8_t tab; };
struct Net { std::vector<Node> nodes; };

static int g_nin = 9;                                   // 9 inputs (letter cell) or 8 (N cell: neq ignored)
static Sig g_in[10], g_valid, g_target[8];
static int g_mode = 0;                                  // 0 letter cell, 1 N cell, 2 the barcode kernels' deficit step
static std::mutex g_mu;

static inline void eval_node(const Sig* s, const Node& n, Sig& out) {
    const Sig &x = s[n.f[0]], &y = s[n.f[1]], &z = s[n.f[2]];
    for (int k = 0; k < NW; ++k) {
        uint32_t r = 0;
        for (int m = 0; m < 8; ++m)
            if (n.tab >> m & 1) r |= ((m & 4) ? x.w[k] : ~x.w[k]) & ((m & 2) ? y.w[k] : ~y.w[k]) & ((m & 1) ? z.w[k] : ~z.w[k]);
        out.w[k] = r;
    }
}

// cost: for every target the smallest number of wrong valid patterns over all nodes (0 = some node computes it exactly)
static int cost(const Net& n, std::vector<Sig>& s, int* out_nodes = nullptr) {
    const int N = (int)n.nodes.size();
    for (int i = 0; i < g_nin; ++i) s[i] = g_in[i];
    for (int i = 0; i < N; ++i) eval_node(s.data(), n.nodes[i], s[g_nin + i]);
    int total = 0;
    for (int t = 0; t < 8; ++t) {
        int best = 1 << 30, arg = -1;
        for (int i = 0; i < N; ++i) {
            int bad = 0;
            for (int k = 0; k < NW; ++k) bad += __builtin_popcount((s[g_nin + i].w[k] ^ g_target[t].w[k]) & g_valid.w[k]);
            if (bad < best) { best = bad; arg = i; }
        }
        if (out_nodes) out_nodes[t] = arg;
        total += best;
    }
    return total;
}

// mode 2: bs_deficit of kernels_bitslice.inc -- F (8 planes, 0..151: 1 + deficit of the last column, 0 before the first
// row) and a (2 planes, dv + 1 in 0..3) -> F' = max(F + 1 - a, 1).  signals: 0..7 = f0..f7, 8 = a1, 9 = a0.
static void setup_deficit() {
    g_nin = 10;
    memset(g_in, 0, sizeof g_in); memset(&g_valid, 0, sizeof g_valid); memset(g_target, 0, sizeof g_target);
    for (int p = 0; p < 1024; ++p) {
        const int F = p >> 2, a = p & 3;
        auto set = [&](Sig& sg) { sg.w[p >> 5] |= 1u << (p & 31); };
        for (int k = 0; k < 8; ++k) if (F >> k & 1) set(g_in[k]);
        if (a & 2) set(g_in[8]);
        if (a & 1) set(g_in[9]);
        if (F > 152) continue;
        set(g_valid);
        const int Fn = std::max(F + 1 - a, 1);
        for (int k = 0; k < 8; ++k) if (Fn >> k & 1) set(g_target[k]);
    }
}

static void setup(bool letter) {
    g_nin = letter ? 9 : 8;
    memset(g_in, 0, sizeof g_in); memset(&g_valid, 0, sizeof g_valid); memset(g_target, 0, sizeof g_target);
    for (int p = 0; p < 512; ++p) {
        const int a = p >> 5, b = (p >> 1) & 15, neq = p & 1;
        auto set = [&](Sig& sg) { sg.w[p >> 5] |= 1u << (p & 31); };
        for (int k = 0; k < 4; ++k) { if (a >> (3 - k) & 1) set(g_in[k]); if (b >> (3 - k) & 1) set(g_in[4 + k]); }
        if (neq) set(g_in[8]);
        if (a > 9 || b > 9) continue;
        if (!letter && !neq) continue;                  // N cell: the neq input does not exist; keep half the patterns
        set(g_valid);
        const int m = letter ? (neq ? std::max(std::max(a, b), 2) : 9) : std::max(std::max(a, b), 3);
        const int an = m - b, bn = m - a;
        for (int k = 0; k < 4; ++k) { if (an >> (3 - k) & 1) set(g_target[k]); if (bn >> (3 - k) & 1) set(g_target[4 + k]); }
    }
}

static uint8_t tab_of(unsigned (*f)(unsigned, unsigned, unsigned)) { return (uint8_t)f(0xF0u, 0xCCu, 0xAAu); }
#define LUT(F) tab_of([](unsigned x, unsigned y, unsigned z) -> unsigned { (void)z; return (F) & 0xFFu; })

// the hand-made network of abs_core.h (abs_cell_letter / abs_cell_n); signal ids: a3 a2 a1 a0 = 0..3, b3..b0 = 4..7, neq = 8
static Net seed(bool letter) {
    Net n;
    const int NI = letter ? 9 : 8;
    auto add = [&](int x, int y, int z, uint8_t t) { n.nodes.push_back(Node{{x, y, z}, t}); return NI + (int)n.nodes.size() - 1; };
    const int a[4] = {0, 1, 2, 3}, b[4] = {4, 5, 6, 7}, neq = 8;          // index 0 = bit 3
    const uint8_t BRW = LUT((~x & y) | ((~x | y) & z)), SEL = LUT((x & y) | (~x & z)), XOR3 = LUT(x ^ y ^ z);
    int k = add(a[3], b[3], b[3], LUT(~x & y));
    k = add(a[2], b[2], k, BRW); k = add(a[1], b[1], k, BRW);
    const int lt = add(a[0], b[0], k, BRW);
    int mx[4];
    for (int q = 0; q < 4; ++q) mx[q] = add(lt, b[q], a[q], SEL);
    int m[4];
    if (letter) {
        const int z = add(mx[0], mx[1], mx[2], LUT(x | y | z));
        m[0] = add(mx[0], neq, neq, LUT(x | ~y));
        m[1] = add(mx[1], neq, neq, LUT(x & y));
        m[2] = add(neq, mx[2], z, LUT(x & (y | ~z)));
        m[3] = add(neq, mx[3], z, LUT(~x | (y & z)));
    } else {
        const int z = add(mx[0], mx[1], mx[1], LUT(x | y));
        m[0] = mx[0]; m[1] = mx[1];
        m[2] = add(mx[2], z, z, LUT(x | ~y));
        m[3] = add(mx[3], z, z, LUT(x | ~y));
    }
    for (int side = 0; side < 2; ++side) {
        const int* s = side == 0 ? b : a;
        add(m[3], s[3], s[3], LUT(x ^ y));
        int c = add(m[3], s[3], s[3], LUT(~x & y));
        add(m[2], s[2], c, XOR3); c = add(m[2], s[2], c, BRW);
        add(m[1], s[1], c, XOR3); c = add(m[1], s[1], c, BRW);
        add(m[0], s[0], c, XOR3);
    }
    return n;
}

static Net seed_deficit() {
    Net n;
    auto add = [&](int x, int y, int z, uint8_t t) { n.nodes.push_back(Node{{x, y, z}, t}); return 10 + (int)n.nodes.size() - 1; };
    const int f[8] = {0, 1, 2, 3, 4, 5, 6, 7}, a1 = 8, a0 = 9;
    const uint8_t OR3 = LUT(x | y | z), XOR3 = LUT(x ^ y ^ z), MAJ = LUT((x & y) | (x & z) | (y & z));
    const int t1 = add(f[2], f[3], f[4], OR3), t2 = add(f[5], f[6], f[7], OR3);
    const int g1 = add(t1, t2, f[1], OR3), g0 = add(t1, t2, f[0], OR3);
    const int e1 = add(a1, g1, g1, LUT(x & y));
    const int same = add(a1, g1, g1, LUT(~(x ^ y)));
    const int pick = add(a1, g0, a0, LUT((x & y) | (~x & z)));
    const int ag = add(a0, g0, g0, LUT(x & y));
    const int e0 = add(same, ag, pick, LUT((x & y) | (~x & z)));
    int c = add(f[0], e0, e0, LUT(x & ~y));
    add(f[0], e0, e0, LUT(~(x ^ y)));
    add(f[1], e1, c, XOR3); c = add(f[1], e1, c, MAJ);
    for (int k = 2; k < 8; ++k) { add(f[k], e1, c, XOR3); if (k < 7) c = add(f[k], e1, c, MAJ); }
    return n;
}

static void print_net(const Net& n, std::vector<Sig>& s, bool letter) {
    int outs[8];
    const int c = cost(n, s, outs);
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_mode == 2) printf("EXACT=%d deficit step, %d nodes; signals 0..7 = f0..f7, 8 = a1, 9 = a0; outputs f'0..f'7 = ", c == 0, (int)n.nodes.size());
    else printf("EXACT=%d %s cell, %d nodes; signals 0..3 = a3..a0, 4..7 = b3..b0%s; outputs a'3..a'0 b'3..b'0 = ", c == 0, letter ? "letter" : "N",
           (int)n.nodes.size(), letter ? ", 8 = neq" : "");
    for (int t = 0; t < 8; ++t) printf("s%d ", g_nin + outs[t]);
    printf("\n");
    for (size_t i = 0; i < n.nodes.size(); ++i)
        printf("  s%d = LUT[0x%02x](s%d, s%d, s%d)\n", g_nin + (int)i, n.nodes[i].tab, n.nodes[i].f[0], n.nodes[i].f[1], n.nodes[i].f[2]);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const bool letter = argc < 2 || argv[1][0] == 'l';
    g_mode = (argc > 1 && argv[1][0] == 'd') ? 2 : (letter ? 0 : 1);
    const int nthreads = argc > 2 ? atoi(argv[2]) : 4;
    const long iters = argc > 3 ? atol(argv[3]) : 400000;
    if (g_mode == 2) setup_deficit(); else setup(letter);
    Net best = g_mode == 2 ? seed_deficit() : seed(letter);
    if (argc > 4) {                                      // continue from the last exact network of an earlier run's log
        FILE* fh = fopen(argv[4], "r");
        char line[512];
        Net cur; bool exact = false;
        while (fh && fgets(line, sizeof line, fh)) {
            if (!strncmp(line, "EXACT=", 6)) { if (exact && !cur.nodes.empty()) best = cur; cur.nodes.clear(); exact = line[6] == '1'; continue; }
            int sid, x, y, z; unsigned tab;
            if (sscanf(line, " s%d = LUT[0x%x](s%d, s%d, s%d)", &sid, &tab, &x, &y, &z) == 5) cur.nodes.push_back(Node{{x, y, z}, (uint8_t)tab});
        }
        if (exact && !cur.nodes.empty()) best = cur;
        if (fh) fclose(fh);
    }
    const unsigned long long seed0 = argc > 5 ? strtoull(argv[5], nullptr, 10) : 987654321ull;
    {
        std::vector<Sig> s(g_nin + best.nodes.size());
        if (cost(best, s) != 0) { fprintf(stderr, "the seed network is not exact\n"); print_net(best, s, letter); return 1; }
        print_net(best, s, letter);
    }
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back([=, &best] {
        std::mt19937_64 rng(seed0 + 7919ull * t);
        for (;;) {
            Net cur;
            { std::lock_guard<std::mutex> lk(g_mu); cur = best; }
            // delete one node; its readers take one of its inputs instead
            const int N0 = (int)cur.nodes.size(), del = (int)(rng() % N0);
            Net n;
            for (int i = 0; i < N0; ++i) {
                if (i == del) continue;
                Node nd = cur.nodes[i];
                for (int q = 0; q < 3; ++q) {
                    if (nd.f[q] == g_nin + del) nd.f[q] = cur.nodes[del].f[rng() % 3];
                    else if (nd.f[q] > g_nin + del) nd.f[q] -= 1;
                }
                n.nodes.push_back(nd);
            }
            const int N = N0 - 1;
            std::vector<Sig> s(g_nin + N);
            int c = cost(n, s);
            for (long it = 0; it < iters && c > 0; ++it) {
                const Net old = n;
                const int i = (int)(rng() % N), mv = (int)(rng() % 4);
                if (mv == 0) n.nodes[i].f[rng() % 3] = (int)(rng() % (g_nin + i));
                else if (mv == 1) n.nodes[i].tab ^= (uint8_t)(1u << (rng() % 8));
                else if (mv == 2) { n.nodes[i].f[rng() % 3] = (int)(rng() % (g_nin + i)); n.nodes[i].tab ^= (uint8_t)(1u << (rng() % 8)); }
                else n.nodes[i].tab = (uint8_t)rng();
                const int c2 = cost(n, s);
                const double temp = 1.2 * (1.0 - (double)it / iters) + 0.1;
                if (c2 <= c || exp((c - c2) / temp) > (double)(rng() % 1000000) / 1e6) c = c2; else n = old;
            }
            if (c == 0) {
                bool better = false;
                { std::lock_guard<std::mutex> lk(g_mu); if (n.nodes.size() < best.nodes.size()) { best = n; better = true; } }
                if (better) print_net(n, s, letter);
            }
        }
    });
    for (auto& x : th) x.join();
}
