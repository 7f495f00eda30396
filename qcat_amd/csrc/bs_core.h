// bs_core.h -- the BARCODE DP in bit-sliced difference form: the arithmetic shared by the device kernels
// (kernels_bitslice.inc) and by the host check of the same arithmetic (tests/bs_host_check.cpp, plain g++).
//
// find_highest_scoring_barcode (qcat/scanner_base.py:63-141) aligns every barcode target to the barcode region with
// match +1 / mismatch -1 / gap open = extend = 1 (config.py:26, scanner_base.py:111-117) and consumes the SCORE only
// (:119, :141).  With those scores the differences between neighbouring DP cells take four values,
//       dv(i,j) = H(i,j) - H(i-1,j),   dh(i,j) = H(i,j) - H(i,j-1)      in {-1, 0, 1, 2},
// and one cell is a boolean function of five bits: with a = dv(i,j-1) + 1, b = dh(i-1,j) + 1 (two bits each) and
// eq = [query letter == target letter],
//       m = eq ? 3 : max(a, b, 1)          ( = H(i,j) - H(i-1,j-1) + 2 )
//       dv(i,j) + 1 = m - b,   dh(i,j) + 1 = m - a.
// Bit-slicing ACROSS ALIGNMENTS puts bit k of a quantity of 32 different alignments into one 32-bit word.
//
// Round 5 -- BOTH contexts of a target shared.  A target is  leading context (P columns) + barcode + trailing context
// (Q columns), M = P + C + Q.  The gap cost is linear, so every path that reaches a column beyond c = P + C passes
// through exactly one LAST lattice point (i, c) of the line in front of the trailing context, and what it scores after
// that point is a semi-global alignment of the rest of the region against the trailing context that STARTS in that
// point -- read backwards: the DP R of the reversed region against the reversed trailing context, ending in its last
// column.  With H the forward DP over the columns <= c:
//       score = max( max_{0<=i<=L} [ H(i,c) + R(L-i, Q) ],      paths that cross the line at row i
//                    max_{1<=j<=c} H(L,j),                      paths that end in the last row before it
//                    max_{1<=j<=Q} R(L,j) )                     paths that lie behind it altogether
// which is exact (every path of the full DP is one of the three kinds and every term is a path of the full DP).
// R depends on the region and the trailing context only -- not on the barcode: its last column is computed ONCE per
// 2048 alignments beside the leading context's columns, one difference dvR(k) = R(k,Q) - R(k-1,Q) per row, and a
// barcode walks  G(i) = H(i,c) + R(L-i,Q)  down its last own column as a deficit counter whose step is
// dv(i,c) - dvR(L-i+1) in [-3, 3] (bs_deficit_split), G(L) = H(L,c).  A barcode's row costs C cells instead of C + Q.
//
// Everything here is a pure function of 32-bit words, so the host check runs the very same code 32 alignments at a
// time against the oracle's scalar DP.
#ifndef QCAT_BS_CORE_H
#define QCAT_BS_CORE_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
#define BS_FN __host__ __device__ __forceinline__
#else
#define BS_FN inline
#endif

namespace qk {

typedef uint32_t u32;

// bit (x << 2 | y << 1 | z) of the table is f(x, y, z) -- the convention of v_bitop3_b32
BS_FN u32 bs_lut_host(u32 x, u32 y, u32 z, unsigned t) {
    u32 r = 0;
    for (int i = 0; i < 8; ++i)
        if ((t >> i) & 1u) r |= ((i & 4) ? x : ~x) & ((i & 2) ? y : ~y) & ((i & 1) ? z : ~z);
    return r;
}

}  // namespace qk

// gfx950 has v_bitop3_b32, an arbitrary boolean function of three registers given by its truth table.
// QB3(x, y, z, f) evaluates f on the constants 0xF0 / 0xCC / 0xAA at compile time to get the table, QT3 takes a table.
#if defined(__HIP_DEVICE_COMPILE__)
#define BS_LUT(X, Y, Z, T) __builtin_amdgcn_bitop3_b32((X), (Y), (Z), (unsigned)(T))
#else
#define BS_LUT(X, Y, Z, T) qk::bs_lut_host((X), (Y), (Z), (unsigned)(T))
#endif
#define QB3(X, Y, Z, ...) BS_LUT((X), (Y), (Z), ([](unsigned x, unsigned y, unsigned z) constexpr { return (unsigned)(__VA_ARGS__) & 0xFFu; }(0xF0u, 0xCCu, 0xAAu)))
#define QT3(X, Y, Z, TABLE) BS_LUT((X), (Y), (Z), (TABLE))

namespace qk {

constexpr int BS_NB = 7;                      // planes of a score counter: H + 64 in [0, 127]
constexpr int BS_OFF = 64;
constexpr int BS_NF = 8;                      // planes of the (unsplit) deficit counter
constexpr int BS_ND = 7;                      // planes of the split form's deficit: G stays within [-M, M], M <= 59
constexpr int BS_POST_MAX = 12;               // trailing columns the split form takes out of a barcode's rows

BS_FN u32 bs_bfi(u32 m, u32 x, u32 y) { return QB3(m, x, y, (x & y) | (~x & z)); }

// g += (p1 p0) - 1        (p in 0..3; never leaves [0, 127] by construction): one ripple pass over the planes with
// p - 1 in two's complement = (.., sg, sg, ~(p1 ^ p0), ~p0), sg = [p = 0]
BS_FN void bs_step(u32 (&g)[BS_NB], u32 p1, u32 p0) {
    u32 c = QB3(g[0], p0, p0, x & ~y);
    g[0] = QB3(g[0], p0, p0, ~(x ^ y));
    {
        const u32 x1 = QB3(p1, p0, p0, ~(x ^ y));
        const u32 n1 = QB3(g[1], x1, c, x ^ y ^ z);
        c = QB3(g[1], x1, c, (x & y) | (x & z) | (y & z));
        g[1] = n1;
    }
    const u32 sg = QB3(p1, p0, p0, ~x & ~y);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 2; k < BS_NB; ++k) {
        const u32 n = QB3(g[k], sg, c, x ^ y ^ z);
        c = QB3(g[k], sg, c, (x & y) | (x & z) | (y & z));
        g[k] = n;
    }
}

// mask of the alignments with x > y
BS_FN u32 bs_gt(const u32 (&x)[BS_NB], const u32 (&y)[BS_NB]) {
    u32 gt = QB3(x[BS_NB - 1], y[BS_NB - 1], y[BS_NB - 1], x & ~y);
    u32 eq = QB3(x[BS_NB - 1], y[BS_NB - 1], y[BS_NB - 1], ~(x ^ y));
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = BS_NB - 2; k >= 0; --k) {
        gt |= QB3(eq, x[k], y[k], x & y & ~z);
        if (k) eq = QB3(eq, x[k], y[k], x & ~(y ^ z));
    }
    return gt;
}

BS_FN void bs_max(u32 (&best)[BS_NB], const u32 (&x)[BS_NB]) {
    const u32 gt = bs_gt(x, best);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < BS_NB; ++k) best[k] = bs_bfi(gt, x[k], best[k]);
}

// one DP cell for 32 alignments per lane: SEVEN three-input boolean instructions (a minimal-size network found by
// tools/lut3_search.cpp; no six-node network turned up).  neq: letter mismatch mask; (a1 a0) = dv + 1 of the left cell
// (in: column j - 1, out: column j); (b1 b0) = dh + 1 of column j (in: row i - 1, out: row i).
// m = eq ? 3 : max(a, b, 1); out a = m - b, out b = m - a.
BS_FN void bs_cell(u32 neq, u32& a1, u32& a0, u32& b1, u32& b0) {
    const u32 n5 = QT3(a1, a0, neq, 0xd5);
    const u32 n6 = QT3(b1, a1, n5, 0x5b);
    const u32 n7 = QT3(a0, b1, b0, 0x73);
    const u32 p0 = QT3(b0, b1, n6, 0x16);                 // out a, bit 0
    const u32 q1 = QT3(n5, a1, n7, 0x31);                 // out b, bit 1
    const u32 p1 = QT3(n5, b0, n6, 0xa1);                 // out a, bit 1
    const u32 q0 = QT3(b0, p0, a0, 0x16);                 // out b, bit 0
    a1 = p1; a0 = p0; b1 = q1; b0 = q0;
}

// Front padding (round 5: regions a few bases shorter than the nominal length share the nominal units).  An alignment whose
// region has p rows fewer than the unit starts p rows late: semi-global alignment starts from H(0, j) = 0 whatever came
// before, so until its first row it is simply HELD at the boundary state -- after each of the unit's first rows the dh
// planes of the alignments that have not started (mask hm) go back to 1 (bs_hold), the trailing columns' reversed DP, which
// meets those rows LAST, keeps the state it had after the alignment's own last row and hands out r = 0 for them
// (bs_keep; with r = 0 the deficit stays 0: D' = max(0 + 0 - a, 0)), and every alignment ends in the unit's last row.
BS_FN void bs_hold(u32& h1, u32& h0, u32 hm) { h1 &= ~hm; h0 |= hm; }
BS_FN void bs_keep(u32& h, u32 old, u32 hm) { h = bs_bfi(hm, old, h); }

// letter mismatch mask: planes (c1 c0) of the alignments' letters against the wave-uniform letter (L1 L0), each 0 / ~0
BS_FN u32 bs_neq(u32 c1, u32 c0, u32 L1, u32 L0) {
    return QB3(c1 ^ L1, c0, L0, x | (y ^ z));
}

// mismatch mask against a compile-time letter: one of the four masks a row has (the optimiser forms each once per row)
template <int LETTER>
BS_FN u32 bs_neq_static(u32 c1, u32 c0) {
    return LETTER == 0 ? (c1 | c0) : (LETTER == 1 ? (u32)QB3(c1, c0, c0, x | ~y) : (LETTER == 2 ? (u32)QB3(c1, c0, c0, ~x | y) : (u32)QB3(c1, c0, c0, ~(x & y))));
}
BS_FN u32 bs_neq_letter(int letter, u32 c1, u32 c0) {
    return letter == 0 ? bs_neq_static<0>(c1, c0) : (letter == 1 ? bs_neq_static<1>(c1, c0) : (letter == 2 ? bs_neq_static<2>(c1, c0) : bs_neq_static<3>(c1, c0)));
}

// UNSPLIT form.  F = deficit + 1 of the last column (F = 0 before the first row): F = max(F + 1 - a, 1) = F + 1 - min(a, F)
BS_FN void bs_deficit(u32 (&f)[BS_NF], u32 a1, u32 a0) {
    const u32 t1 = QB3(f[2], f[3], f[4], x | y | z), t2 = QB3(f[5], f[6], f[7], x | y | z);   // F >= 4
    const u32 g1 = QB3(t1, t2, f[1], x | y | z), g0 = QB3(t1, t2, f[0], x | y | z);           // min(F, 3)
    const u32 e1 = a1 & g1;                                                                    // min(a, F): bit 1
    const u32 same = QB3(a1, g1, g1, ~(x ^ y));
    const u32 pick = bs_bfi(a1, g0, a0);                                                       // a > F in bit 1: F's bit 0; else a's
    const u32 e0 = QB3(same, a0 & g0, pick, (x & y) | (~x & z));
    // F += 1 - e:  x = 1 - e in two's complement = (e1, ~e0), sign e1
    u32 c = QB3(f[0], e0, e0, x & ~y);
    f[0] = QB3(f[0], e0, e0, ~(x ^ y));
    {
        const u32 s1 = QB3(f[1], e1, c, x ^ y ^ z);
        c = QB3(f[1], e1, c, (x & y) | (x & z) | (y & z));
        f[1] = s1;
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 2; k < BS_NF; ++k) {
        const u32 s = QB3(f[k], e1, c, x ^ y ^ z);
        if (k + 1 < BS_NF) c = QB3(f[k], e1, c, (x & y) | (x & z) | (y & z));
        f[k] = s;
    }
}

// SPLIT form.  D = (running maximum of G) - G(i) along the line in front of the trailing context, G(i) = H(i,c) + R(L-i,Q):
//       D' = max(D + r - a, 0),    a = dv(i,c) + 1 (this row's last own cell),  r = dvR(L-i+1) + 1 (the R row that pairs with it)
// D = 0 before the first row (G(0) = R(L,Q) is a path of its own).  t = r - a in [-3, 3] as (sign, t1, t0); x = D + t over
// seven planes and a sign; a negative x becomes 0.  23 instructions (the unsplit update is 25, with 7 more columns in front).
BS_FN void bs_deficit_split(u32 (&d)[BS_ND], u32 a1, u32 a0, u32 r1, u32 r0) {
    u32 s[BS_ND];
    s[0] = QB3(d[0], r0, a0, x ^ y ^ z);
    u32 c = QB3(d[0], r0, a0, x & (y ^ z));
    const u32 bw = QB3(r0, a0, a0, ~x & y);                                   // borrow of r0 - a0
    const u32 t1 = QB3(r1, a1, bw, x ^ y ^ z);
    const u32 sg = QB3(r1, a1, bw, (~x & y) | (~(x ^ y) & z));               // borrow out of bit 1 = sign of t
    s[1] = QB3(d[1], t1, c, x ^ y ^ z);
    c = QB3(d[1], t1, c, (x & y) | (x & z) | (y & z));
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 2; k < BS_ND - 1; ++k) {
        s[k] = QB3(d[k], sg, c, x ^ y ^ z);
        c = QB3(d[k], sg, c, (x & y) | (x & z) | (y & z));
    }
    // the top plane and the sign of x are functions of (d[6], sg, c): x6 = d ^ sg ^ c, carry = maj, negative = sg ^ carry
    const u32 neg = QB3(d[BS_ND - 1], sg, c, y ^ ((x & y) | (x & z) | (y & z)));
    d[BS_ND - 1] = QB3(d[BS_ND - 1], sg, c, (x ^ y ^ z) & ~(y ^ ((x & y) | (x & z) | (y & z))));
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < BS_ND - 1; ++k) d[k] = QB3(s[k], neg, neg, x & ~y);
}

// SPLIT form, the end of a barcode: r = H(L,c) + 64 (the last row summed up to the line), d = the deficit, rowbest = the
// last row's maximum over the columns <= c, cmax = max_j R(L,j): raw + 64 = max(r + d, rowbest, cmax) into rowbest
BS_FN void bs_finish_split(u32 (&rowbest)[BS_NB], u32 (&r)[BS_NB], const u32 (&d)[BS_ND], const u32 (&cmax)[BS_NB]) {
    static_assert(BS_ND == BS_NB, "one ripple over both");
    u32 c = 0u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int q = 0; q < BS_NB; ++q) {
        const u32 sum = QB3(r[q], d[q], c, x ^ y ^ z);
        c = QB3(r[q], d[q], c, (x & y) | (x & z) | (y & z));
        r[q] = sum;
    }
    bs_max(rowbest, r);
    bs_max(rowbest, cmax);
}

// UNSPLIT form, the end of a barcode: the last column's maximum = H(L,M) + deficit = r + F - 1 (mod 128: the value fits)
BS_FN void bs_finish(u32 (&rowbest)[BS_NB], u32 (&r)[BS_NB], const u32 (&f)[BS_NF]) {
    u32 c = 0u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int q = 0; q < BS_NB; ++q) {
        const u32 sum = QB3(r[q], f[q], c, x ^ y ^ z);
        c = QB3(r[q], f[q], c, (x & y) | (x & z) | (y & z));
        r[q] = sum;
    }
    u32 bw = ~r[0];
    r[0] = bw;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int q = 1; q < BS_NB; ++q) { const u32 t = QB3(r[q], bw, bw, ~x & y); r[q] ^= bw; bw = t; }
    bs_max(rowbest, r);
}

}  // namespace qk
#endif
