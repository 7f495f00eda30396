// abs_core.h -- the ADAPTER DP in bit-sliced difference form (round 3): arithmetic shared by the device kernels
// (kernels_abs.inc) and by the host check of the same arithmetic (tests/abs_host_check.cpp, plain g++).
//
// find_best_adapter_template (qcat/scanner_base.py:313-359) aligns every template to the read window with parasail's
// semi-global DP (:214-218) under qcatConfig's adapter scoring (qcat/config.py:12-21, :236-253): letter match +5,
// mismatch -2, anything against a template N -1, gap open = extend = 2.  In the gap-free form G(i,j) = H(i,j) + 2(i+j)
//       G(i,j) = max(G(i-1,j-1) + W', G(i-1,j), G(i,j-1)),     W' = W + 4  in {9 match, 2 mismatch, 3 N column},
// G is monotone in both directions and the differences
//       a(i,j) = G(i,j) - G(i-1,j),    b(i,j) = G(i,j) - G(i,j-1)          take the ten values 0..9:
// FOUR bit planes each, and a cell is
//       m = max(W', a(i,j-1), b(i-1,j));     a(i,j) = m - b(i-1,j);     b(i,j) = m - a(i,j-1).
// Bit-slicing ACROSS ALIGNMENTS (bit k of a quantity of 32 alignments per 32-bit word, 64 lanes: 2048 alignments per
// wave instruction) turns that into 24 three-input boolean instructions per column (27 / 25 in the hand-made reference form, 24 / 24 after the search)
// (v_bitop3_b32, any function of three registers), where the packed-binary16 form (kernels_static.inc) spends two
// half-rate packed instructions per 128 cells.
//
// The price is paid at the borders, which parasail's end-position rule (oracle/qcat_oracle.c:100-108)
// needs exactly: the maximum of the last column with the FIRST row that reaches it, the maximum of the last row, and
// whether the last row's first maximum sits in the last column.  Both borders are walked as "deficits":
//       F = 1 + (running maximum - current value),   F' = max(F - d, 1),   a strict new maximum  <=>  F - d <= 0
// with d = the step of H along the border (a - 2 down the last column, b - 2 along the last row), ten planes; the first
// row of a new maximum is latched into eight index planes.  H(L,M) itself is the sum of the last row's b planes.
//
// Everything here is a pure function of 32-bit words, so the host check runs the very same code 32 alignments at a
// time against the oracle's scalar DP.
#ifndef QCAT_ABS_CORE_H
#define QCAT_ABS_CORE_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
#define ABS_FN __host__ __device__ __forceinline__
#else
#define ABS_FN inline
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define ABS_LUT(X, Y, Z, T) __builtin_amdgcn_bitop3_b32((X), (Y), (Z), (unsigned)(T))
#else
#define ABS_LUT(X, Y, Z, T) qabs::abs_lut_host((X), (Y), (Z), (unsigned)(T))
#endif

namespace qabs {

typedef uint32_t u32;

// bit (x << 2 | y << 1 | z) of the table is f(x, y, z) -- the convention of v_bitop3_b32
ABS_FN u32 abs_lut_host(u32 x, u32 y, u32 z, unsigned t) {
    u32 r = 0;
    for (int i = 0; i < 8; ++i)
        if ((t >> i) & 1u) r |= ((i & 4) ? x : ~x) & ((i & 2) ? y : ~y) & ((i & 1) ? z : ~z);
    return r;
}
}  // namespace qabs

// AB3(x, y, z, f): f evaluated on the constants 0xF0 / 0xCC / 0xAA at compile time gives the truth table
#define AB3(X, Y, Z, ...) ABS_LUT((X), (Y), (Z), ([](unsigned x, unsigned y, unsigned z) constexpr { return (unsigned)(__VA_ARGS__) & 0xFFu; }(0xF0u, 0xCCu, 0xAAu)))

namespace qabs {

constexpr int ABS_NP = 4;        // planes of a difference (0..9)
constexpr int ABS_NF = 10;       // planes of a deficit counter / of the last-row sum: <= 7 * 128 < 1024
#ifndef QCAT_ABS_NI
#define QCAT_ABS_NI 8
#endif
constexpr int ABS_NI = QCAT_ABS_NI;   // planes of a row index: 8 for the read ends' windows (< 256 rows); the translation unit of the interior
                                      // scan (abs_mid_kernels.hip, --detect-middle) is compiled with 14 (interiors up to 16 384 rows)
constexpr int ABS_G = 2;         // the gap cost the form is built on (open == extend == 2)
constexpr int ABS_W_MATCH = 9, ABS_W_MISMATCH = 2, ABS_W_N = 3;       // W + 2g

// difference planes of the boundary: G(i,0) = 2i, G(0,j) = 2j -> every boundary difference is 2
ABS_FN void abs_set2(u32 (&v)[ABS_NP]) { v[0] = 0u; v[1] = 0xFFFFFFFFu; v[2] = 0u; v[3] = 0u; }

// v = hold ? 2 : v -- an alignment that has not started yet (front padding of the interior scan, kernels_abs_mid.inc) keeps the
// boundary state of row 0 whatever the row computed for it
ABS_FN void abs_hold2(u32 (&v)[ABS_NP], u32 hold) { v[0] &= ~hold; v[1] |= hold; v[2] &= ~hold; v[3] &= ~hold; }

// max(a, b) -> mx, through the borrow chain of a - b (lt = [a < b]) and four selects
ABS_FN void abs_max(const u32 (&a)[ABS_NP], const u32 (&b)[ABS_NP], u32 (&mx)[ABS_NP]) {
    const u32 k0 = AB3(a[0], b[0], b[0], ~x & y);
    const u32 k1 = AB3(a[1], b[1], k0, (~x & y) | ((~x | y) & z));
    const u32 k2 = AB3(a[2], b[2], k1, (~x & y) | ((~x | y) & z));
    const u32 lt = AB3(a[3], b[3], k2, (~x & y) | ((~x | y) & z));
#pragma unroll
    for (int k = 0; k < ABS_NP; ++k) mx[k] = AB3(lt, b[k], a[k], (x & y) | (~x & z));
}

// d = m - s for m >= s (four planes, the borrow out of the top plane is zero by construction)
ABS_FN void abs_sub(const u32 (&m)[ABS_NP], const u32 (&s)[ABS_NP], u32 (&d)[ABS_NP]) {
    d[0] = AB3(m[0], s[0], s[0], x ^ y);
    const u32 c0 = AB3(m[0], s[0], s[0], ~x & y);
    d[1] = AB3(m[1], s[1], c0, x ^ y ^ z);
    const u32 c1 = AB3(m[1], s[1], c0, (~x & y) | ((~x | y) & z));
    d[2] = AB3(m[2], s[2], c1, x ^ y ^ z);
    const u32 c2 = AB3(m[2], s[2], c1, (~x & y) | ((~x | y) & z));
    d[3] = AB3(m[3], s[3], c2, x ^ y ^ z);
}

// one cell of a LETTER column, reference form.  neq: mismatch mask of the alignments' query letter against the column's
// letter; a: difference down the column to the left (in) -> of this column (out); b: difference along the row above (in)
// -> of this row (out).  m = neq ? max(a, b, 2) : 9.   27 instructions: the hand-made network the search started from
// (tools/lut3_search_adapter.cpp) and what the searched networks below are checked against, exhaustively
// (tests/abs_host_check.cpp).
ABS_FN void abs_cell_letter_ref(u32 neq, u32 (&a)[ABS_NP], u32 (&b)[ABS_NP]) {
    u32 mx[ABS_NP], m[ABS_NP], na[ABS_NP], nb[ABS_NP];
    abs_max(a, b, mx);
    const u32 z = AB3(mx[3], mx[2], mx[1], x | y | z);           // max(a, b) >= 2
    m[3] = AB3(mx[3], neq, neq, x | ~y);                           // match: 9 = 1001
    m[2] = AB3(mx[2], neq, neq, x & y);
    m[1] = AB3(neq, mx[1], z, x & (y | ~z));                       // mismatch below 2: 2 = 0010
    m[0] = AB3(neq, mx[0], z, ~x | (y & z));
    abs_sub(m, b, na);
    abs_sub(m, a, nb);
#pragma unroll
    for (int k = 0; k < ABS_NP; ++k) { a[k] = na[k]; b[k] = nb[k]; }
}

// one cell of an N column (every query letter scores -1 against a template N), reference form: m = max(a, b, 3).  25.
ABS_FN void abs_cell_n_ref(u32 (&a)[ABS_NP], u32 (&b)[ABS_NP]) {
    u32 mx[ABS_NP], m[ABS_NP], na[ABS_NP], nb[ABS_NP];
    abs_max(a, b, mx);
    const u32 z = mx[3] | mx[2];                                   // max(a, b) >= 4
    m[3] = mx[3]; m[2] = mx[2];
    m[1] = AB3(mx[1], z, z, x | ~y);                               // below 4: at least 3 = 0011
    m[0] = AB3(mx[0], z, z, x | ~y);
    abs_sub(m, b, na);
    abs_sub(m, a, nb);
#pragma unroll
    for (int k = 0; k < ABS_NP; ++k) { a[k] = na[k]; b[k] = nb[k]; }
}

// The cells the kernels run: networks found by tools/lut3_search_adapter.cpp from the reference forms above (a node is
// deleted, fan-ins and truth tables are re-annealed until all eight outputs are exact again on every valid input: a, b in
// 0..9 -- values above 9 never occur, which is what the smaller networks exploit), pasted by tools/lut3_to_cpp.py.
ABS_FN void abs_cell_letter(u32 neq, u32 (&a)[ABS_NP], u32 (&b)[ABS_NP]) {
    // EXACT=1 letter cell, 24 nodes; signals 0..3 = a3..a0, 4..7 = b3..b0, 8 = neq; outputs a'3..a'0 b'3..b'0 = s25 s24 s22 s20 s32 s30 s28 s26
    const u32 s9 = ABS_LUT(a[1], b[1], b[0], 0x8e);
    const u32 s10 = ABS_LUT(a[2], b[2], s9, 0x8e);
    const u32 s11 = ABS_LUT(a[3], b[3], s10, 0x8e);
    const u32 s12 = ABS_LUT(b[3], neq, a[3], 0xfb);
    const u32 s13 = ABS_LUT(b[2], s12, a[2], 0x32);
    const u32 s14 = ABS_LUT(s10, b[1], a[1], 0xca);
    const u32 s15 = ABS_LUT(s11, b[0], a[0], 0xca);
    const u32 s16 = ABS_LUT(s12, s13, s14, 0x7e);
    const u32 s17 = ABS_LUT(s12, b[2], b[2], 0x34);
    const u32 s18 = ABS_LUT(s12, s14, s13, 0x2d);
    const u32 s19 = ABS_LUT(neq, s15, s16, 0x8b);
    const u32 s20 = ABS_LUT(s9, s19, b[0], 0x66);
    const u32 s21 = ABS_LUT(b[0], s19, b[0], 0x70);
    const u32 s22 = ABS_LUT(s18, b[1], s21, 0x96);
    const u32 s23 = ABS_LUT(s22, s21, b[1], 0xe8);
    const u32 s24 = ABS_LUT(s13, b[2], s23, 0x16);
    const u32 s25 = ABS_LUT(b[3], s23, s17, 0xc2);
    const u32 s26 = ABS_LUT(neq, s19, a[0], 0x64);
    const u32 s27 = ABS_LUT(s26, a[0], a[3], 0x40);
    const u32 s28 = ABS_LUT(s18, a[1], s27, 0x96);
    const u32 s29 = ABS_LUT(s28, a[1], s18, 0xd4);
    const u32 s30 = ABS_LUT(s13, a[2], s29, 0x96);
    const u32 s31 = ABS_LUT(s29, a[3], s12, 0x3d);
    const u32 s32 = ABS_LUT(s31, s30, b[3], 0x03);
    a[3] = s25; a[2] = s24; a[1] = s22; a[0] = s20;
    b[3] = s32; b[2] = s30; b[1] = s28; b[0] = s26;
}

ABS_FN void abs_cell_n(u32 (&a)[ABS_NP], u32 (&b)[ABS_NP]) {
    // EXACT=1 N cell, 24 nodes; signals 0..3 = a3..a0, 4..7 = b3..b0; outputs a'3..a'0 b'3..b'0 = s24 s22 s20 s18 s31 s29 s27 s25
    const u32 s8 = ABS_LUT(a[1], b[1], b[0], 0x8e);
    const u32 s9 = ABS_LUT(a[2], b[2], s8, 0x8e);
    const u32 s10 = ABS_LUT(a[3], b[3], s9, 0x8e);
    const u32 s11 = ABS_LUT(s10, b[3], a[3], 0xca);
    const u32 s12 = ABS_LUT(s10, b[2], a[2], 0xca);
    const u32 s13 = ABS_LUT(s10, b[1], a[1], 0xca);
    const u32 s14 = ABS_LUT(s10, b[0], a[0], 0xca);
    const u32 s15 = ABS_LUT(s11, s12, s12, 0xfc);
    const u32 s16 = ABS_LUT(s13, s15, s15, 0xf3);
    const u32 s17 = ABS_LUT(s14, s15, s15, 0xf3);
    const u32 s18 = ABS_LUT(s17, b[0], b[0], 0x3c);
    const u32 s19 = ABS_LUT(s17, b[0], b[0], 0x0c);
    const u32 s20 = ABS_LUT(s16, b[1], s19, 0x96);
    const u32 s21 = ABS_LUT(s16, b[1], s19, 0x8e);
    const u32 s22 = ABS_LUT(s12, b[2], s21, 0x96);
    const u32 s23 = ABS_LUT(s12, b[2], s21, 0x8e);
    const u32 s24 = ABS_LUT(s11, b[3], s23, 0x96);
    const u32 s25 = ABS_LUT(s17, a[0], a[0], 0x3c);
    const u32 s26 = ABS_LUT(s17, a[0], a[0], 0x0c);
    const u32 s27 = ABS_LUT(s16, a[1], s26, 0x96);
    const u32 s28 = ABS_LUT(s16, a[1], s26, 0x8e);
    const u32 s29 = ABS_LUT(s12, a[2], s28, 0x96);
    const u32 s30 = ABS_LUT(s12, a[2], s28, 0x8e);
    const u32 s31 = ABS_LUT(s11, a[3], s30, 0x96);
    a[3] = s24; a[2] = s22; a[1] = s20; a[0] = s18;
    b[3] = s31; b[2] = s29; b[1] = s27; b[0] = s25;
}

// mismatch masks of a row against the four letters (codes A, T, G, C = 0..3 in planes c1 c0)
ABS_FN void abs_neq_masks(u32 c1, u32 c0, u32 (&nq)[4]) {
    nq[0] = c1 | c0;
    nq[1] = AB3(c1, c0, c0, x | ~y);
    nq[2] = AB3(c1, c0, c0, ~x | y);
    nq[3] = AB3(c1, c0, c0, ~(x & y));
}

// one step along a border.  F = 1 + (running maximum - current value) of H along the border, d = the difference
// planes of the step (0..9; H moves by d - 2).  S = F + 2 - d; a strict new maximum <=> S <= 0; F' = max(S, 1).
// `force` (all ones or zero, wave-uniform): the first value of a border is a maximum by definition (F' = 1 whatever F
// held) -- a mask, not a branch, so that a row body is one basic block.  Returns the mask of the alignments with a new
// strict maximum.
ABS_FN u32 abs_border_step(u32 (&F)[ABS_NF], const u32 (&d)[ABS_NP], u32 force) {
    // y = 2 - d as five planes in two's complement (range -7 .. 2): the borrow chain of (0 1 0) - d
    const u32 y0 = d[0];
    const u32 r0 = d[0];                                           // borrow out of plane 0
    const u32 y1 = AB3(d[1], r0, r0, ~(x ^ y));                    // 1 ^ d1 ^ r0
    const u32 r1 = d[1] & r0;                                      // 1 - d1 - r0 < 0
    const u32 y2 = d[2] ^ r1;
    const u32 r2 = d[2] | r1;
    const u32 y3 = d[3] ^ r2;
    const u32 ys = d[3] | r2;                                      // sign = every higher plane
    // S = F + y
    u32 S[ABS_NF];
    S[0] = F[0] ^ y0;
    u32 c = F[0] & y0;
    S[1] = AB3(F[1], y1, c, x ^ y ^ z); c = AB3(F[1], y1, c, (x & y) | (x & z) | (y & z));
    S[2] = AB3(F[2], y2, c, x ^ y ^ z); c = AB3(F[2], y2, c, (x & y) | (x & z) | (y & z));
    S[3] = AB3(F[3], y3, c, x ^ y ^ z); c = AB3(F[3], y3, c, (x & y) | (x & z) | (y & z));
#pragma unroll
    for (int k = 4; k < ABS_NF; ++k) {
        S[k] = AB3(F[k], ys, c, x ^ y ^ z);
        c = AB3(F[k], ys, c, (x & y) | (x & z) | (y & z));
    }
    // F >= 0 and y >= -7: S < 0 <=> the sign plane of the (ABS_NF + 1)-bit sum = ys & ~carry
    const u32 neg = AB3(ys, c, c, x & ~y);
    const u32 nz0 = AB3(S[0], S[1], S[2], x | y | z), nz1 = AB3(S[3], S[4], S[5], x | y | z), nz2 = AB3(S[6], S[7], S[8], x | y | z);
    const u32 nz = AB3(nz0, nz1, nz2, x | y | z) | S[9];
    const u32 nm = AB3(neg, nz, force, x | ~y | z);                // S <= 0, or the first value
    F[0] = S[0] | nm;
#pragma unroll
    for (int k = 1; k < ABS_NF; ++k) F[k] = AB3(S[k], nm, nm, x & ~y);
    return nm;
}
static_assert(ABS_NF == 10, "abs_border_step's zero test is written for ten planes");

// idx = mask ? value : idx (value wave-uniform, eight planes)
ABS_FN void abs_latch_index(u32 (&idx)[ABS_NI], u32 mask, unsigned value) {
#pragma unroll
    for (int k = 0; k < ABS_NI; ++k) idx[k] = ((value >> k) & 1u) ? (idx[k] | mask) : AB3(idx[k], mask, mask, x & ~y);
}

// sum += d (four planes into ABS_NF planes; the sum of a last row's differences fits by construction)
ABS_FN void abs_accumulate(u32 (&sum)[ABS_NF], const u32 (&d)[ABS_NP]) {
    u32 c = sum[0] & d[0];
    sum[0] ^= d[0];
#pragma unroll
    for (int k = 1; k < ABS_NP; ++k) {
        const u32 s = AB3(sum[k], d[k], c, x ^ y ^ z);
        c = AB3(sum[k], d[k], c, (x & y) | (x & z) | (y & z));
        sum[k] = s;
    }
#pragma unroll
    for (int k = ABS_NP; k < ABS_NF; ++k) {
        const u32 s = sum[k] ^ c;
        c = sum[k] & c;
        sum[k] = s;
    }
}

// x > y over ABS_NF planes (borrow chain of y - x)
ABS_FN u32 abs_gt(const u32 (&xv)[ABS_NF], const u32 (&yv)[ABS_NF]) {
    u32 k = AB3(yv[0], xv[0], xv[0], ~x & y);
#pragma unroll
    for (int q = 1; q < ABS_NF; ++q) k = AB3(yv[q], xv[q], k, (~x & y) | ((~x | y) & z));
    return k;
}

// state of one template's borders (per 32 alignments)
struct AbsBorder {
    u32 Fc[ABS_NF];             // last column: deficit + 1 after the rows so far
    u32 ic[ABS_NI];             // ... first row (0-based) of its maximum
};
struct AbsLastRow {
    u32 Fr[ABS_NF];             // last row: deficit + 1 after the columns so far
    u32 sum[ABS_NF];            // sum of the differences b(L, 1..j) = H(L,j) + 2j
    u32 newmax;                 // did the latest column set a strict new maximum?
};

ABS_FN void abs_lastrow_init(AbsLastRow& r) {
#pragma unroll
    for (int k = 0; k < ABS_NF; ++k) { r.Fr[k] = 0u; r.sum[k] = 0u; }
    r.newmax = 0u;
}
ABS_FN void abs_lastrow_step(AbsLastRow& r, const u32 (&b)[ABS_NP], bool first) {
    r.newmax = abs_border_step(r.Fr, b, first ? 0xFFFFFFFFu : 0u);
    abs_accumulate(r.sum, b);
}

// The decision of oracle/qcat_oracle.c:100-108 for one template, still in planes:
//   s_row = H(L,M) + (Fr - 1),  s_col = H(L,M) + (Fc - 1),  jr == M <=> the last column of the last row was a strict
//   new maximum;  col = s_col > s_row || (s_col == s_row && jr == M)  -- and jr == M implies s_row = H(L,M) <= s_col,
//   so col = newmax | (Fc > Fr);   score = sum - 2M + (col ? Fc : Fr) - 1;   end_query = col ? ic : L - 1.
// Out: val = sum + (col ? Fc : Fr)  (the caller subtracts 2M + 1 after the planes are un-transposed), endq planes.
// r1_scalar (wave-uniform; QCAT_R1_SCALAR, plain parasail.sg's order): the last column wins every tie, col = s_col >= s_row
// = !(Fr > Fc)  (jr == M still implies it).
ABS_FN void abs_decide(const AbsBorder& bd, const AbsLastRow& lr, unsigned last_row_index, u32 (&val)[ABS_NF + 1], u32 (&endq)[ABS_NI],
                       bool r1_scalar = false) {
    const u32 col = r1_scalar ? ~abs_gt(lr.Fr, bd.Fc) : (lr.newmax | abs_gt(bd.Fc, lr.Fr));
    u32 c = 0u;
#pragma unroll
    for (int k = 0; k < ABS_NF; ++k) {
        const u32 f = AB3(col, bd.Fc[k], lr.Fr[k], (x & y) | (~x & z));
        val[k] = AB3(lr.sum[k], f, c, x ^ y ^ z);
        c = AB3(lr.sum[k], f, c, (x & y) | (x & z) | (y & z));
    }
    val[ABS_NF] = c;
#pragma unroll
    for (int k = 0; k < ABS_NI; ++k) endq[k] = ((last_row_index >> k) & 1u) ? (bd.ic[k] | ~col) : (bd.ic[k] & col);
}

}  // namespace qabs
#endif  // QCAT_ABS_CORE_H
