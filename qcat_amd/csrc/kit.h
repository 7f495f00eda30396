// kit.h -- prepared kit: what the kernels read.  Built on the host from a qcat_kit_desc
// (include/qcat_hip.h) by kit_prepare() and copied verbatim to device memory.
//
// Every float decision of the reference is turned into an integer threshold HERE, on the host,
// with the same IEEE double expression the reference evaluates (Python float == C double):
//   * get_norm_socre            score*100.0/den            (qcat/scanner_base.py:299-310)
//   * barcode score             raw*100.0/(1.0*len(target)) (qcat/scanner_base.py:119)
//   * `> 90.0`, `< min_quality`, `>= 60`                    (scanner_epi2me.py:74, scanner_base.py:546-548,:585-588)
// Both maps are monotone in the integer raw score (den > 0 is required), so "smallest raw that
// passes" is an exact replacement; comparisons BETWEEN two normalised scores are done by integer
// cross-multiplication (distinct rationals with these magnitudes never round to one double).
#pragma once
#include <stdint.h>
#ifndef QCAT_RTC            // (run-time compiled kit units see the device structs only: rtc_prelude.inc)
#include <string>
#include <vector>
#endif

#include "../../include/qcat_hip.h"
#include "options.h"

namespace qk {

constexpr int MAX_T = QCAT_MAX_TEMPLATES;
constexpr int MAX_TLEN = QCAT_MAX_TEMPLATE_LEN;
constexpr int MAX_TARGET = QCAT_MAX_TARGET_LEN;
constexpr int MAX_WIN = QCAT_MAX_WINDOW;
constexpr int WIN_STRIDE = 160;          // bytes per packed code window (16-B aligned rows)
constexpr int WIN2_WORDS = WIN_STRIDE / 16; // dwords of a window at two bits per code (k_pack_windows: win2)
constexpr int BS_PAD_ROWS = 12;             // rows a region of the "short of nominal" class may miss (front padding of the bit-sliced barcode units)
constexpr int WIN2_FRONT = 16;              // dwords of slack in front of the first window of the context's win2 buffer
constexpr int RAW_NEVER = 1 << 20;       // "no raw score passes"
constexpr int PADMAX = 8;                // most leading padding columns a width class may have

// width classes of the packed kernels (must match tools/gen_cols.py); 0 = not supported
inline int adapter_width_class(int len) {
    const int w[] = {40, 48, 56, 60, 64, 84, 92, 104, 112, 120, 128};
    for (int v : w) if (len <= v && v - len <= PADMAX) return v;
    return 0;
}
inline int barcode_width_class(int len) {
    const int w[] = {40, 42, 44, 46, 48, 56, 64};
    for (int v : w) if (len <= v && v - len <= PADMAX) return v;
    return 0;
}

struct DevSet {
    int32_t n, blen, tlen, uplen, downlen;
    int32_t tgt_off;            // codes blob: n * tlen target codes (up + barcode + down)
    int32_t ids_off;            // ids blob: n dense ids
    int32_t tbl_off;            // tables blob: n rows of `width` dwords, right-aligned (-1: not eligible)
    int32_t width;              // register-array width class of the packed barcode kernel
    int32_t min_raw_pass;       // smallest raw with raw*100.0/tlen >= min_quality
    int32_t min_raw_conflict;   // smallest raw with raw*100.0/tlen >= conflict_min_score
    int32_t min_raw_middle;     // smallest raw with raw*100.0/tlen >= middle_min_score (--detect-middle)
    int32_t static_kernel;      // generated static-letter kernel of this group (kernels_static.inc), -1: none
    int32_t case_off;           // ids blob: n_pairs entries (pair case, barcode of half 0 or -1, barcode of half 1 or -1)
    int32_t n_pairs;            // target pairs the static kernel runs for this group (quad mode: the pairs outside every quad)
    int32_t quad_off;           // ids blob: n_quads entries (quad case, barcode a, b, c, d) -- two pairs in one row pass
    int32_t n_quads;
    int32_t bs_off;             // ids blob: per barcode the target letters as bit words (letter bit 1 lo, hi, letter bit 0 lo, hi)
                                // for the bit-sliced kernels (kernels_bitslice.inc); -1: set not eligible
    int32_t hot_len;            // the region length almost every job of this set has: barcode + 2 * extension + 1
    int32_t bs_pre;             // bit-sliced kernels: leading columns every barcode of the set shares (0, 4, 8 or 11 context letters)
    int32_t bs_rev;             // ... rows and target letters run backwards (the downstream context is the longer one)
    int32_t bs_post;            // ... trailing columns (11 / 8 / 7 / 6 / 4 / 0 letters of the other context) that every barcode shares as well:
                                // computed once per super-tile by the reversed DP (bs_core.h, round 5); own columns = tlen - bs_pre - bs_post
    int32_t bs_short_min;       // ... regions of bs_short_min .. hot_len - 1 rows share the nominal units, padded at the front (round 5; hot_len: no such class)
    int32_t bs_kernel;          // bit-sliced kernel with this set's letters compiled in (static_generated.inc / run-time code), -1: none
    int32_t bs_case_off;        // ids blob: per barcode its case of that kernel
    int32_t len_off;            // simple mode, barcodes of unequal length: ids blob, per barcode (length, min_raw_pass, min_raw_conflict);
                                // -1: every barcode has blen letters (the set's own tlen / thresholds apply)
};

constexpr int BS_C_MIN = 20, BS_C_MAX = 48;    // own columns (tlen - bs_pre - bs_post) the bit-sliced kernels are instantiated for
// trailing columns of a set on the bit-sliced kernels (the rule is restated in tools/gen_static_kernels.py and qcat_amd/jit.py)
inline int bs_post_of(int trail, int tlen, int pre) {
    const int posts[5] = {11, 8, 7, 6, 4};
    for (int q : posts) if (q <= trail && tlen - pre - q >= BS_C_MIN) return q;
    return 0;
}

// work units of a static-letter barcode group per tile: chunks of quads first, then chunks of the pairs left over
inline int static_units(int n_quads, int n_pairs, int chunk_b) {
    const int cq = chunk_b / 4 > 0 ? chunk_b / 4 : 1, cp = chunk_b / 2 > 0 ? chunk_b / 2 : 1;
    return (n_quads + cq - 1) / cq + (n_pairs + cp - 1) / cp;
}

struct DevTpl {
    int32_t len, trim_offset, is_double, den, kit_slot;
    int32_t bc_end[2], bc_len[2];
    int32_t region_min_raw;     // smallest raw with raw*100.0/den > region_min_adapter_score
    int32_t code_off;           // codes blob: len template codes
    int32_t tbl_off;            // tables blob: `width` dwords, right-aligned (-1: not eligible)
    int32_t width;              // register-array width class of the packed adapter kernel
    int32_t static_kernel;      // generated static-letter adapter kernel (kernels_static.inc), -1: none
    int32_t fused_kernel;       // generated kernel that scans this template AND `fused_partner` in one pass, -1: none
    int32_t fused_partner;
    int32_t abs_jit;            // bit-sliced adapter plans in the kit's run-time generated code (qcat_kit_attach_code, template flag bits 1..3):
                                // 1 two stages (qj_abs_<t>), 2 four stages of <= 13 columns (qj_absm_<t>), 4 four wide stages (qj_absw_<t>)
    DevSet sets[2];
};

struct DevKit {
    int32_t mode, ends, nt;
    int32_t gap_open, gap_extend, max_align, ext;
    int32_t n_barcode_slots, n_kit_slots, n_buckets;
    int32_t scan_middle;        // --detect-middle enabled
    int32_t min_read_length, trim_reads;   // the driver's min-length filter of the histogram (qcat/cli.py:521-534)
    int32_t fast_ok;            // every template/set is eligible for the packed fast path
    int32_t barcode_f16;        // barcode tables hold binary16 high bytes (fp16-lane barcode kernels)
    int32_t bs_ok;              // the barcode scoring is +1 / -1 / gap 1 with N matching nothing: bit-sliced kernels allowed
    uint32_t special_adapter;   // v_perm pool bytes for query codes N, X, other, PAD (adapter)
    uint32_t special_barcode;   //   "    (barcode alignments)
    uint32_t letter_tbl_barcode[4];   // score dword of a column whose target letter is A, T, G, C (static kernels)
    // static adapter kernels: binary16 bits of the biased adapter score W'(query code, target letter x),
    // x = A, T, G, C, N; pool_lo/hi[x] = low / high bytes for query A, T, G, C
    uint32_t adapter_pool_lo[5], adapter_pool_hi[5];
    uint16_t adapter_w16[5 * 8];
    int32_t adapter_f16;        // the binary16 adapter DP is exact for this kit (static adapter kernels allowed)
    int32_t adapter_f16_headroom;   // 2047 - (largest value the window-sized biased adapter DP can reach)
    int32_t r1_scalar;          // rule R1 as plain parasail.sg orders it (include/qcat_hip.h QCAT_R1_SCALAR): on a tie of the two
                                // borders' maxima the last column wins; 0: sg_striped_32's order
    int32_t abs_ok;             // the adapter scoring is +5 / -2 / N -1 / gap 2 and every template is made of A, T, G, C, N:
                                // the bit-sliced adapter kernels (kernels_abs.inc) are exact for this kit
    int8_t amat[49], bmat[49];
    int8_t pad_[2];             // (sizeof stays a multiple of 4)
    DevTpl tpl[MAX_T];
};

// compact per-read-end record produced by the scan kernels and consumed by k_finalize
struct EndRec {
    int32_t window_len;
    int32_t best_tpl;           // -1: no template beat -1.0
    int32_t used_tpl;           // Python [-1] wrap applied
    int32_t best_end, best_raw;
    int32_t region_path;
    int32_t region_start[2], region_len[2];
    int32_t bc_idx[2], bc_raw[2];
};

#ifndef QCAT_RTC
struct HostKit {
    DevKit dk;
    std::vector<uint8_t> codes;
    std::vector<int32_t> ids;
    std::vector<uint32_t> tables;
    // ASCII copies for the synthetic generator (host + device)
    std::vector<char> ascii;                 // templates then barcode blobs
    int32_t ascii_tpl_off[MAX_T];
    int32_t ascii_set_off[MAX_T][2];
    int32_t bc_start[MAX_T][2];
};

int kit_prepare(const qcat_kit_desc* d, HostKit* out, std::string* err);
#endif

inline uint8_t code_of_ascii(uint8_t c) {
    switch (c & 0xDF) {
        case 'A': return QCAT_CODE_A;
        case 'T': return QCAT_CODE_T;
        case 'G': return QCAT_CODE_G;
        case 'C': return QCAT_CODE_C;
        case 'N': return QCAT_CODE_N;
        case 'X': return QCAT_CODE_X;
        default: return QCAT_CODE_OTHER;
    }
}

}  // namespace qk
