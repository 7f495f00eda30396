// lut3_search_block.cpp -- dev tool: local search for a network of three-input boolean functions (v_bitop3_b32) that
// computes a BLOCK of R x C cells of the bit-sliced barcode DP (kernels_bitslice.inc: bs_cell) in fewer than 7 R C nodes.
//     inputs   a_r (2 bits each: dv + 1 entering row r from the left), b_c (2 bits each: dh + 1 entering column c from
//              above), neq_rc (letters differ in cell (r, c))
//     cell     m = neq ? max(a, b, 1) : 3;  a <- m - b (goes right), b <- m - a (goes down)
//     outputs  a_r after the last column of the block, b_c after its last row      (values inside the block are free)
// The search starts from R C copies of the seven-node cell, deletes a node (its readers take one of its inputs), anneals
// fan-ins and truth tables at a low temperature until every output is exact on all input patterns again, and goes on
// from the smaller network.  Every exact network found is printed in the format of tools/lut3_search_adapter.cpp.
// build: g++ -O2 -std=c++17 -pthread tools/lut3_search_block.cpp -o /tmp/lut3blk
// usage: /tmp/lut3blk <rows> <cols> [threads] [iterations per attempt] [log to continue from] [seed]   (runs until killed)
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

typedef uint64_t u64;
constexpr int MAXW = 1024;                              // up to 65536 patterns
static int g_nw = 4;                                    // 64-bit words per signal
struct Sig { std::vector<u64> w; Sig() : w((size_t)g_nw, 0) {} };
struct Node { int f[3]; uint8_t tab; };
struct Net { std::vector<Node> nodes; };

static int g_R = 1, g_C = 2, g_nin = 8, g_nt = 6;
static std::vector<Sig> g_in, g_target;
static std::mutex g_mu;
static double g_tscale = 1.0;                          // temperature in wrong patterns: scaled with the pattern count

static inline void eval_node(const std::vector<Sig>& s, const Node& n, Sig& out) {
    const u64 *x = s[(size_t)n.f[0]].w.data(), *y = s[(size_t)n.f[1]].w.data(), *z = s[(size_t)n.f[2]].w.data();
    u64 c[8];
    for (int m = 0; m < 8; ++m) c[m] = (n.tab >> m & 1) ? ~0ull : 0ull;
    for (int k = 0; k < g_nw; ++k) {
        const u64 zz = z[k], yy = y[k], xx = x[k];
        const u64 g0 = (c[1] & zz) | (c[0] & ~zz), g1 = (c[3] & zz) | (c[2] & ~zz), g2 = (c[5] & zz) | (c[4] & ~zz), g3 = (c[7] & zz) | (c[6] & ~zz);
        const u64 h0 = (yy & g1) | (~yy & g0), h1 = (yy & g3) | (~yy & g2);
        out.w[(size_t)k] = (xx & h1) | (~xx & h0);
    }
}

// evaluates nodes from `from` on (signals below are up to date); cost: per target the fewest wrong patterns over all nodes
static int cost(const Net& n, std::vector<Sig>& s, int from, int* out_nodes = nullptr) {
    const int N = (int)n.nodes.size();
    for (int i = from; i < N; ++i) eval_node(s, n.nodes[(size_t)i], s[(size_t)(g_nin + i)]);
    int total = 0;
    for (int t = 0; t < g_nt; ++t) {
        int best = 1 << 30, arg = -1;
        const u64* T = g_target[(size_t)t].w.data();
        for (int i = 0; i < N; ++i) {
            const u64* v = s[(size_t)(g_nin + i)].w.data();
            int bad = 0;
            for (int k = 0; k < g_nw && bad < best; ++k) bad += __builtin_popcountll(v[k] ^ T[k]);
            if (bad < best) { best = bad; arg = i; }
        }
        if (out_nodes) out_nodes[t] = arg;
        total += best;
    }
    return total;
}

// signal ids: a_r = (2 r, 2 r + 1) [bit 1, bit 0]; b_c = 2 R + (2 c, 2 c + 1); neq_rc = 2 R + 2 C + r C + c
static void setup() {
    g_nin = 2 * g_R + 2 * g_C + g_R * g_C;
    g_nt = 2 * g_R + 2 * g_C;
    const long np = 1L << g_nin;
    g_nw = (int)std::max(1L, np / 64);
    g_tscale = std::max(1.0, (double)np / 256.0);
    g_in.assign((size_t)g_nin, Sig()); g_target.assign((size_t)g_nt, Sig());
    for (long p = 0; p < np; ++p) {
        auto bit = [&](int sig) { return (int)((p >> sig) & 1); };
        auto set = [&](Sig& sg) { sg.w[(size_t)(p >> 6)] |= 1ull << (p & 63); };
        for (int i = 0; i < g_nin; ++i) if (bit(i)) set(g_in[(size_t)i]);
        int a[8], b[8];
        for (int r = 0; r < g_R; ++r) a[r] = bit(2 * r) * 2 + bit(2 * r + 1);
        for (int c = 0; c < g_C; ++c) b[c] = bit(2 * g_R + 2 * c) * 2 + bit(2 * g_R + 2 * c + 1);
        for (int r = 0; r < g_R; ++r)
            for (int c = 0; c < g_C; ++c) {
                const int neq = bit(2 * g_R + 2 * g_C + r * g_C + c);
                const int m = neq ? std::max(std::max(a[r], b[c]), 1) : 3;
                const int an = m - b[c], bn = m - a[r];
                a[r] = an; b[c] = bn;
            }
        for (int r = 0; r < g_R; ++r) { if (a[r] & 2) set(g_target[(size_t)(2 * r)]); if (a[r] & 1) set(g_target[(size_t)(2 * r + 1)]); }
        for (int c = 0; c < g_C; ++c) { if (b[c] & 2) set(g_target[(size_t)(2 * g_R + 2 * c)]); if (b[c] & 1) set(g_target[(size_t)(2 * g_R + 2 * c + 1)]); }
    }
}

// R C copies of bs_cell (kernels_bitslice.inc)
static Net seed() {
    Net n;
    auto add = [&](int x, int y, int z, unsigned t) { n.nodes.push_back(Node{{x, y, z}, (uint8_t)t}); return g_nin + (int)n.nodes.size() - 1; };
    int a1[8], a0[8], b1[8], b0[8];
    for (int r = 0; r < g_R; ++r) { a1[r] = 2 * r; a0[r] = 2 * r + 1; }
    for (int c = 0; c < g_C; ++c) { b1[c] = 2 * g_R + 2 * c; b0[c] = 2 * g_R + 2 * c + 1; }
    for (int r = 0; r < g_R; ++r)
        for (int c = 0; c < g_C; ++c) {
            const int neq = 2 * g_R + 2 * g_C + r * g_C + c;
            const int n5 = add(a1[r], a0[r], neq, 0xd5);
            const int n6 = add(b1[c], a1[r], n5, 0x5b);
            const int n7 = add(a0[r], b1[c], b0[c], 0x73);
            const int p0 = add(b0[c], b1[c], n6, 0x16);
            const int q1 = add(n5, a1[r], n7, 0x31);
            const int p1 = add(n5, b0[c], n6, 0xa1);
            const int q0 = add(b0[c], p0, a0[r], 0x16);
            a1[r] = p1; a0[r] = p0; b1[c] = q1; b0[c] = q0;
        }
    return n;
}

static void print_net(const Net& n, std::vector<Sig>& s) {
    std::vector<int> outs((size_t)g_nt);
    for (int i = 0; i < g_nin; ++i) s[(size_t)i] = g_in[(size_t)i];
    const int c = cost(n, s, 0, outs.data());
    std::lock_guard<std::mutex> lk(g_mu);
    printf("EXACT=%d block %d x %d, %d nodes; signals: a_r = (2r, 2r+1) [bit 1, bit 0], b_c = %d + (2c, 2c+1), neq_rc = %d + r * %d + c; outputs a'_r.. b'_c.. = ",
           c == 0, g_R, g_C, (int)n.nodes.size(), 2 * g_R, 2 * g_R + 2 * g_C, g_C);
    for (int t = 0; t < g_nt; ++t) printf("s%d ", g_nin + outs[(size_t)t]);
    printf("\n");
    for (size_t i = 0; i < n.nodes.size(); ++i)
        printf("  s%d = LUT[0x%02x](s%d, s%d, s%d)\n", g_nin + (int)i, n.nodes[i].tab, n.nodes[i].f[0], n.nodes[i].f[1], n.nodes[i].f[2]);
    fflush(stdout);
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s rows cols [threads] [iterations] [log] [seed]\n", argv[0]); return 2; }
    g_R = atoi(argv[1]); g_C = atoi(argv[2]);
    const int nthreads = argc > 3 ? atoi(argv[3]) : 4;
    const long iters = argc > 4 ? atol(argv[4]) : 400000;
    setup();
    Net best = seed();
    if (argc > 5) {
        FILE* fh = fopen(argv[5], "r");
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
    const unsigned long long seed0 = argc > 6 ? strtoull(argv[6], nullptr, 10) : 987654321ull;
    {
        std::vector<Sig> s((size_t)g_nin + best.nodes.size());
        for (int i = 0; i < g_nin; ++i) s[(size_t)i] = g_in[(size_t)i];
        if (cost(best, s, 0) != 0) { fprintf(stderr, "the seed network is not exact\n"); print_net(best, s); return 1; }
        print_net(best, s);
    }
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back([=, &best] {
        std::mt19937_64 rng(seed0 + 7919ull * (unsigned)t);
        for (;;) {
            Net cur;
            { std::lock_guard<std::mutex> lk(g_mu); cur = best; }
            const int N0 = (int)cur.nodes.size(), del = (int)(rng() % (unsigned)N0);
            Net n;
            for (int i = 0; i < N0; ++i) {
                if (i == del) continue;
                Node nd = cur.nodes[(size_t)i];
                for (int q = 0; q < 3; ++q) {
                    if (nd.f[q] == g_nin + del) nd.f[q] = cur.nodes[(size_t)del].f[rng() % 3];
                    else if (nd.f[q] > g_nin + del) nd.f[q] -= 1;
                }
                n.nodes.push_back(nd);
            }
            const int N = N0 - 1;
            std::vector<Sig> s((size_t)(g_nin + N)), keep((size_t)(g_nin + N));
            for (int i = 0; i < g_nin; ++i) s[(size_t)i] = g_in[(size_t)i];
            int c = cost(n, s, 0);
            for (long it = 0; it < iters && c > 0; ++it) {
                const int i = (int)(rng() % (unsigned)N), mv = (int)(rng() % 4);
                const Node old = n.nodes[(size_t)i];
                Node& nd = n.nodes[(size_t)i];
                if (mv == 0) nd.f[rng() % 3] = (int)(rng() % (unsigned)(g_nin + i));
                else if (mv == 1) nd.tab ^= (uint8_t)(1u << (rng() % 8));
                else if (mv == 2) { nd.f[rng() % 3] = (int)(rng() % (unsigned)(g_nin + i)); nd.tab ^= (uint8_t)(1u << (rng() % 8)); }
                else nd.tab = (uint8_t)rng();
                for (int q = i; q < N; ++q) keep[(size_t)(g_nin + q)].w.swap(s[(size_t)(g_nin + q)].w);     // park the signals that change
                for (int q = i; q < N; ++q) if (s[(size_t)(g_nin + q)].w.size() != (size_t)g_nw) s[(size_t)(g_nin + q)].w.resize((size_t)g_nw);
                const int c2 = cost(n, s, i);
                const double temp = (1.2 * (1.0 - (double)it / (double)iters) + 0.1) * g_tscale;
                if (c2 <= c || exp((c - c2) / temp) > (double)(rng() % 1000000) / 1e6) c = c2;
                else { n.nodes[(size_t)i] = old; for (int q = i; q < N; ++q) keep[(size_t)(g_nin + q)].w.swap(s[(size_t)(g_nin + q)].w); }
            }
            if (c == 0) {
                bool better = false;
                { std::lock_guard<std::mutex> lk(g_mu); if (n.nodes.size() < best.nodes.size()) { best = n; better = true; } }
                if (better) print_net(n, s);
            }
        }
    });
    for (auto& x : th) x.join();
}
