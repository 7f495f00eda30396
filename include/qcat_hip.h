/*
 * qcat_hip.h -- C ABI of the MI355X-native barcode-demultiplexing hot path.
 *
 * This is the drop-in boundary.  The reference crosses exactly one
 * process->native boundary on this path: one ctypes call into parasail per alignment
 *   qcat/scanner_base.py:214-218   parasail_sg(s1=window, s2=template, open, extend, matrix)
 *   qcat/scanner_base.py:111-117   parasail_sg(s1=region, s2=ctx+barcode+ctx, 1, 1, matrix_barcode)
 * 28..244 times per read, driven by the Python of
 *   qcat/scanner_base.py:521-604   BarcodeScanner.detect_barcode
 *   qcat/scanner_epi2me.py:33-144  BarcodeScannerEPI2ME.scan
 *   qcat/scanner_dual.py:35-146    BarcodeScannerDual.scan
 * The native library replaces that whole stack with ONE call per batch of reads: the caller
 * hands over the reads of a batch and a kit descriptor, and gets back one 24-byte record per
 * read holding everything `detect_barcode` returns (as indices into the kit).  Each entry
 * point below cites the reference interface it stands in for.
 *
 * Conventions: plain C, caller owns every buffer, the library keeps no caller pointer after a
 * call returns; every function returns 0 on success or a negative qcat_status, and
 * qcat_last_error() gives the message of the last failure on the calling thread.
 * The same descriptor/record structs are consumed by the CPU oracle (oracle/qcat_oracle.c),
 * which is test infrastructure and not part of this library.
 */
#ifndef QCAT_HIP_H
#define QCAT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QCAT_ABI_VERSION 6

/* Base code space shared by host and device (qcat_amd/codes.py): parasail's mapper sends the
 * alphabet letters (either case) to their index and everything else to the '*' row
 * (SURVEY.md 8a R1); both qcat matrices are over ATGCN(+X) -- qcat/config.py:26,245. */
enum { QCAT_CODE_A = 0, QCAT_CODE_T = 1, QCAT_CODE_G = 2, QCAT_CODE_C = 3,
       QCAT_CODE_N = 4, QCAT_CODE_X = 5, QCAT_CODE_OTHER = 6, QCAT_CODE_PAD = 7,
       QCAT_NCODES = 7 };

enum { QCAT_MAX_TEMPLATES = 16,      /* kit auto uses 12 (qcat/scanner_base.py:444-447)   */
       QCAT_MAX_TEMPLATE_LEN = 128,  /* longest shipped template: 102 (VMK001)            */
       QCAT_MAX_TARGET_LEN = 64,     /* ctx + barcode + ctx; shipped: 39..46              */
       QCAT_MAX_BARCODES = 1024,     /* per barcode set                                   */
       QCAT_MAX_WINDOW = 160 };      /* qcatConfig.max_align_length (default 150)         */

typedef enum qcat_status {
    QCAT_OK = 0,
    QCAT_ERR_ARG = -1,          /* bad descriptor / argument (RuntimeError on the Python side) */
    QCAT_ERR_UNSUPPORTED = -2,  /* valid qcat configuration the device path does not cover  */
    QCAT_ERR_DEVICE = -3,       /* HIP runtime failure / no device                           */
    QCAT_ERR_NOMEM = -4,
    QCAT_ERR_IO = -5            /* writing an output of qcat_fastq_demux failed (errno text in qcat_last_error) */
} qcat_status;

/* Scanner mode: which `scan()` the batch entry points reproduce. */
typedef enum qcat_mode {
    QCAT_MODE_EPI2ME = 0,   /* qcat/scanner_epi2me.py:33-144 */
    QCAT_MODE_DUAL = 1,     /* qcat/scanner_dual.py:35-146   */
    /* qcat/scanner_simple.py:41-91: no adapter templates -- every barcode of ONE list is aligned to the whole
     * window (find_highest_scoring_barcode without contexts, scanner_base.py:63-141), the winner is reported when
     * its score reaches min_quality, `adapter` is None and adapter_end is the winner's end_query.  The descriptor
     * carries one template of length 0 whose sets[0] is the barcode list (bc_len[0] = barcode length,
     * bc_start / bc_end = -1); general int32 kernel. */
    QCAT_MODE_SIMPLE = 2
} qcat_mode;

/* Which read ends are scanned. */
enum { QCAT_ENDS_5P = 1,          /* scan() on the 5' window only (BASELINE config 2)        */
       QCAT_ENDS_BOTH = 3 };      /* detect_barcode(): 5' + 3' (qcat/scanner_base.py:521-604) */

/* One barcode set of one template (qcat/layout.py:176-189 get_barcode_set).
 * `sequences` holds n * barcode_len ASCII characters (NUL-terminated: a shorter string is rejected with
 * QCAT_ERR_ARG), barcode b at sequences + b*barcode_len.
 * `lengths` (optional, ABI 4): barcode b has lengths[b] letters, 1 <= lengths[b] <= barcode_len, the rest of its row is
 * padding (any non-NUL character).  The reference aligns every barcode with its own length and normalises by it
 * (qcat/scanner_base.py:108-119); only a user FASTA in simple mode (scanner_simple.py:24-29) can hold barcodes of unequal
 * length -- a template's placeholder has ONE length (layout.py:55-61) -- so unequal lengths are accepted in
 * QCAT_MODE_SIMPLE and answered with QCAT_ERR_UNSUPPORTED elsewhere.  NULL: every barcode has barcode_len letters.
 * `ids[b]` is a dense integer standing for Barcode.id -- only equality is ever used
 * (qcat/scanner_base.py:589); the host maps YAML ids to ints. */
typedef struct qcat_barcode_set_desc {
    const char*    sequences;
    const int32_t* ids;
    int32_t        n;
    int32_t        barcode_len;
    const int32_t* lengths;
} qcat_barcode_set_desc;

/* One adapter template = one AdapterLayout (qcat/layout.py:15-70).  Placeholder geometry is
 * what AdapterLayout.get_placeholder_pos returns for the sets that exist (start = end = -1,
 * len = 0 otherwise, qcat/layout.py:48-49). */
typedef struct qcat_template_desc {
    const char* sequence;            /* upper-case ATGCNX, N-masked (layout.py:132-145)       */
    int32_t     length;
    int32_t     trim_offset;         /* YAML trim_offset (adapters.py:101)                    */
    int32_t     is_double_barcode;   /* barcode_set_2 is not None (layout.py:240-248)         */
    int32_t     kit_slot;            /* dense index of the template's kit name (for counts)   */
    int32_t     bc_start[2];
    int32_t     bc_end[2];
    int32_t     bc_len[2];
    qcat_barcode_set_desc sets[2];   /* n == 0 <=> get_barcode_set(i) is None/empty           */
} qcat_template_desc;

/* Everything `detect_barcode` depends on besides the read. */
typedef struct qcat_kit_desc {
    int32_t abi_version;             /* QCAT_ABI_VERSION */
    int32_t mode;                    /* qcat_mode */
    int32_t ends;                    /* QCAT_ENDS_* */
    int32_t n_templates;
    const qcat_template_desc* templates;   /* list order = tie-break order (scanner_base.py:354) */
    /* qcatConfig (qcat/config.py:12-21) */
    int32_t match, nmatch;           /* used by get_norm_socre (scanner_base.py:308-310)      */
    int32_t gap_open, gap_extend;    /* adapter alignments; barcode alignments use 1,1        */
    int32_t max_align_length;
    int32_t extracted_barcode_extension;
    int32_t barcode_context_length;
    int8_t  adapter_matrix[49];      /* [target_code*7 + query_code], config.py:236-253       */
    int8_t  barcode_matrix[49];      /* config.py:26 (X folded onto '*')                      */
    double  min_quality;             /* BarcodeScanner.min_quality (58 epi2me / 60 dual)      */
    double  conflict_min_score;      /* 60 (scanner_base.py:585)                              */
    double  region_min_adapter_score;/* 90.0 (scanner_epi2me.py:74)                           */
    int32_t n_barcode_slots;         /* number of distinct ids over all sets (count buckets)  */
    int32_t n_kit_slots;             /* number of distinct kit names                          */
    /* --detect-middle (qcat/scanner_base.py:479-519, :593-595): after the two end scans, the read
     * interior read[n:-n] and its reverse complement are scanned with the templates of the called
     * kit; a barcode score >= middle_min_score (50.0) there voids the call (exit_status 997). */
    int32_t scan_middle_adapter;
    double  middle_min_score;
    /* the driver's min-length filter for the count histogram (qcat/cli.py:521-534): a read whose sequence --
     * after trimming to [trim5p, trim3p) when trim_reads != 0 -- is shorter than min_read_length is counted in
     * the [skipped] bucket and in no barcode / kit bucket.  0 = no filter (update_barcode_count,
     * scanner_base.py:680-689, counts every read). */
    int32_t min_read_length;
    int32_t trim_reads;
    /* which of the reference's two alignment routines decides an alignment's END POSITION (QCAT_R1_*, below): the
     * reference binds `parasail.sg_striped_32` when parasail reports SSE2 and plain `parasail.sg` otherwise
     * (qcat/scanner_base.py:20-26).  Scores never differ; end_query -- and through it adapter_end, the barcode region
     * and the trims -- can, when the last row's and the last column's maxima tie. */
    int32_t r1_rule;
} qcat_kit_desc;

/* Rule R1 -- where a semi-global alignment ENDS when its score is reached on both borders (ABI 6).  Both restate parasail
 * 2.x as recalled (the library is absent from this image: PARITY UNPINNED beyond the reference's own known answers):
 *   QCAT_R1_STRIPED  `sg_striped_32`: the last row first (target index ascending, strictly greater replaces), then the last
 *                    column: a strictly greater cell replaces the result, an equal one only when the result already sits in
 *                    the last column (then the FIRST row reaching the maximum counts) -- what rounds 1-5 implemented;
 *   QCAT_R1_SCALAR   `sg`: the last column is looked at while the rows go by (strictly greater replaces: the first row
 *                    reaching its maximum), then the last row, target index ascending, strictly greater replaces -- on a
 *                    tie between the two borders the LAST COLUMN wins.
 * One switch, every implementation: this library's kernels (DevKit::r1_scalar), oracle/qcat_oracle.c (qo_sg_rule, the
 * descriptor's field), tests/golden/sg_independent.py sg(rule=...). */
enum { QCAT_R1_STRIPED = 0, QCAT_R1_SCALAR = 1 };
/* qcat_sg_align: OR into `with_stats` to align under QCAT_R1_SCALAR (the module-level helpers have no kit) */
#define QCAT_SG_R1_SCALAR 0x100

/* Result record: the dict of qcat/scanner_base.py:381-388 as indices (24 bytes, little endian).
 * barcode_score of the dict == raw_score * 100.0 / score_den (scanner_base.py:119), recomputed
 * on the host so it is the same IEEE double the reference produces. */
typedef struct qcat_result {
    int16_t barcode_idx;    /* index into templates[adapter_idx].sets[0]; -1 = None            */
    int16_t barcode2_idx;   /* dual mode: index into sets[1]; -1 otherwise                     */
    int16_t adapter_idx;    /* template index; -1 = None                                       */
    int16_t exit_status;    /* 0 / 1 / 1002 (ends disagree) / 997 (adapter in the read interior) */
    int32_t adapter_end;
    int32_t trim5p;
    int32_t trim3p;
    int16_t raw_score;      /* raw alignment score behind barcode_score (0 when None)          */
    int16_t score_den;      /* len(upstream + barcode + downstream) it is divided by (or 1)    */
} qcat_result;

/* Per read-end trace for parity tests (qcat_scan_debug).  All values are the reference's
 * intermediate quantities: per-template raw score / end_query of find_best_adapter_template
 * (scanner_base.py:341-357), the chosen template, the path of scanner_epi2me.py:74-82, the
 * slice taken by extract_barcode_region (scanner_base.py:29-60) and the arg-max of
 * find_highest_scoring_barcode (scanner_base.py:104-134) per barcode set. */
typedef struct qcat_end_trace {
    int32_t window_len;
    int32_t tpl_raw[QCAT_MAX_TEMPLATES];
    int32_t tpl_end[QCAT_MAX_TEMPLATES];
    int32_t best_tpl;          /* -1 if no template beat -1.0 (then the LAST template is used) */
    int32_t best_end;
    int32_t best_raw;
    int32_t used_tpl;          /* template actually indexed (Python [-1] wrap applied)         */
    int32_t region_path;       /* 1 = extract_barcode_region, 0 = whole window                 */
    int32_t region_start[2];
    int32_t region_len[2];
    int32_t bc_idx[2];         /* -1 = None */
    int32_t bc_raw[2];
    int32_t adapter_end;       /* after trim_offset / clamp (scanner_epi2me.py:135-137)        */
} qcat_end_trace;

typedef struct qcat_kit qcat_kit;      /* immutable, shareable between contexts/threads        */
typedef struct qcat_ctx qcat_ctx;      /* one device + stream + staging buffers; one per thread */
typedef struct qcat_batch qcat_batch;  /* reads resident in device memory                      */

const char* qcat_last_error(void);
int  qcat_abi_version(void);
int  qcat_device_count(void);
/* NUMA node of the device's PCI function (from sysfs), -1 when unknown.  No reference counterpart (the reference is
 * single-process): the rank launcher uses it to keep a rank's host threads beside its GPU. */
int  qcat_device_numa_node(int device);

/* replaces: BarcodeScanner.__init__ kit selection + qcatConfig (scanner_base.py:415-447). */
int  qcat_kit_create(const qcat_kit_desc* desc, qcat_kit** out);
void qcat_kit_destroy(qcat_kit* kit);
/* number of int64 count buckets: [barcode slots.., none][kit slots.., none][skipped]
 * (dual: barcode bucket = slot1 * n_barcode_slots + slot2; cli.py:366-383, scanner_base.py:680-689;
 * [skipped] = reads under the min-length filter, cli.py:527-530 -- see qcat_kit_desc.min_read_length) */
int  qcat_kit_count_buckets(const qcat_kit* kit);

/* Which kernels a prepared kit will run (no reference counterpart: the reference has one code
 * path; this reports the library's choice so callers and tests can see a fall-back).
 *   packed            1: packed inter-read DP kernels, 0: generic per-read kernel (affine gaps,
 *                     scores outside the packed range, templates / targets too long)
 *   barcode_f16       1: barcode DP in exact-integer binary16 lanes, 0: u16 lanes
 *   adapter_f16       1: the binary16 adapter DP is exact for this kit
 *   n_templates / n_static_templates   adapter templates / those bound to a generated
 *                     static-letter kernel (templates of the built-in kits)
 *   n_groups / n_static_groups         (template, barcode set) groups / those whose every
 *                     target is a case of a generated static-letter kernel */
typedef struct qcat_kit_info {
    int32_t packed, barcode_f16, adapter_f16;
    int32_t n_templates, n_static_templates;
    int32_t n_groups, n_static_groups;
    int32_t bitslice_groups;    /* low 16 bits: groups the bit-sliced barcode kernels take (big batches);
                                   high 16 bits: those of them with the target letters compiled in */
    int32_t bitslice_templates; /* (ABI 4) templates with a bit-sliced ADAPTER plan: low 8 bits two stages, bits 8-15 four
                                   stages (medium batches), bits 16-23 four wide stages (templates too long for two) */
} qcat_kit_info;
int  qcat_kit_describe(const qcat_kit* kit, qcat_kit_info* out);

/* Attach static-letter kernels compiled for THIS kit at run time (no reference counterpart).  `code`
 * is a gfx950 code object (hipcc --genco of the translation unit qcat_amd/jit.py generates from
 * qcat_amd/csrc/jit_prelude.inc plus the kit's column chains) exporting qj_ad_<t> / qj_am_<t> for every
 * template t with template_flags[t] != 0 and qj_bc_<t*2+s> for every (template, set) group with
 * group_flags[t*2+s] != 0 (arrays of QCAT_MAX_TEMPLATES and 2*QCAT_MAX_TEMPLATES entries).  The barcode
 * chains run two targets per row pass: pair_entries holds, for group g, the triples
 * (pair case in qj_bc_<g>, kit barcode of half 0 or -1, kit barcode of half 1 or -1) at
 * [pair_offsets[g], pair_offsets[g+1]) (pair_offsets has 2*QCAT_MAX_TEMPLATES + 1 entries).  Must be
 * called before the kit is first used on a device; templates / groups already bound to built-in
 * generated kernels keep those. */
int  qcat_kit_attach_code(qcat_kit* kit, const void* code, uint64_t size,
                          const int32_t* template_flags, const int32_t* group_flags,
                          const int32_t* pair_offsets, const int32_t* pair_entries);

/* The same with four-target chains: group g may also export cases of qj_bc_<g>'s run4 switch; quad_entries holds, at
 * [quad_offsets[g], quad_offsets[g+1]), the 5-tuples (quad case, kit barcodes a, b, c, d -- all present).  Every barcode
 * of a flagged group must appear exactly once over its pair and quad lists.  quad_offsets == NULL: no quads.
 * Bit 1 of group_flags[g] (value 2 or 3): the code object also exports qj_bs_<g>, the bit-sliced barcode kernel of the
 * group with its target letters compiled in (case = barcode index of the set; kernels_bitslice.inc); it is bound when the
 * kit's set takes the bit-sliced path at all (qcat_kit_info.bitslice_groups). */
int  qcat_kit_attach_code_quads(qcat_kit* kit, const void* code, uint64_t size,
                                const int32_t* template_flags, const int32_t* group_flags,
                                const int32_t* pair_offsets, const int32_t* pair_entries,
                                const int32_t* quad_offsets, const int32_t* quad_entries);

int  qcat_ctx_create(int device, qcat_ctx** out);
void qcat_ctx_destroy(qcat_ctx* ctx);

/* replaces: the per-read loop of detect_barcode_batch / cli.qcat_cli over detect_barcode
 * (scanner_base.py:723-726, cli.py:504-513) for a fixed kit.  Host buffers in, host buffers
 * out: `bases` = concatenated ASCII reads, read r = bases[offsets[r] .. offsets[r+1]).
 * `counts` (optional, qcat_kit_count_buckets() entries) is ADDED to. */
int  qcat_scan_batch(qcat_ctx* ctx, const qcat_kit* kit,
                     const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads,
                     qcat_result* out, int64_t* counts);

/* replaces: BarcodeScanner.detect_kit (qcat/scanner_base.py:662-678) = per read scan_ends (:632-642):
 * find_best_adapter_template over ALL templates of `kit` at both ends, the template of the
 * higher-scoring end (3' on ties) gets the read's vote.  votes[t] += reads voting for template t,
 * first_read[t] = smallest read index that voted for t (or n_reads): the host folds templates onto
 * kit names and breaks count ties by first appearance like the reference's dict + stable sort
 * (:657-660).  `kit` must have been created with QCAT_ENDS_BOTH. */
int  qcat_detect_kit(qcat_ctx* ctx, const qcat_kit* kit,
                     const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads,
                     int64_t* votes, int64_t* first_read);

/* replaces: BarcodeScanner.detect_barcode_batch with kit auto-detection (qcat/scanner_base.py:714-733): the
 * per-batch vote of qcat_detect_kit over ALL templates of `kit`, then detect_barcode of every read with the
 * templates of the voted kit only (override_kit_name, :527-528, :722-726).  The reference aligns the voted kit's
 * templates a second time; here the adapter alignments of the vote stay on the device and only their merge over
 * the voted kit's templates, the barcode phase and the finalisation follow.  out[i].adapter_idx indexes `kit`'s
 * template list; *chosen_kit_slot = the voted kit slot (-1: empty batch); votes / first_read as in
 * qcat_detect_kit (optional).  Returns QCAT_ERR_UNSUPPORTED for kits whose adapter pass cannot be resumed per
 * kit (templates outside the generated static-letter kernels): call qcat_detect_kit + qcat_scan_batch then. */
int  qcat_scan_batch_auto(qcat_ctx* ctx, const qcat_kit* kit,
                          const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads,
                          qcat_result* out, int64_t* counts, int32_t* chosen_kit_slot,
                          int64_t* votes, int64_t* first_read);

/* The same with the reads as ONE POINTER AND ONE LENGTH PER READ instead of the concatenated form: what a host language that
 * holds a string object per read has at hand -- the Python drop-in passes the buffers of the caller's str objects
 * (qcat/scanner_base.py:714-733 takes `read_sequences`, a list of str), so a batch is not joined and encoded first
 * (0.8 ms of a 3.3 ms call on 4000 reads).  The buffers must stay valid and unchanged during the call. */
int  qcat_scan_batch_auto_ptrs(qcat_ctx* ctx, const qcat_kit* kit,
                               const uint8_t* const* reads, const uint64_t* lengths, uint32_t n_reads,
                               qcat_result* out, int64_t* counts, int32_t* chosen_kit_slot,
                               int64_t* votes, int64_t* first_read);

/* Several consecutive kit-auto batches in ONE call (round 4; replaces: the driver's loop over detect_barcode_batch,
 * qcat/cli.py:500-513 -- the reference votes per batch of 4000 reads in file order): reads [q * batch_reads,
 * (q + 1) * batch_reads) are batch q; every batch votes for its own kit (scanner_base.py:662-678) and its reads are scanned
 * with that kit's templates (:714-733) -- one adapter pass over all reads, votes counted and decided per batch on the
 * device.  chosen_kit_slots: one entry per batch (ceil(n_reads / batch_reads), at most 65535).  Results are identical to
 * one qcat_scan_batch_auto_ptrs call per batch; a 4000-read batch alone keeps the device busy for a fraction of its
 * 0.9 ms call. */
int  qcat_scan_batches_auto_ptrs(qcat_ctx* ctx, const qcat_kit* kit, const uint8_t* const* reads, const uint64_t* lengths,
                                 uint32_t n_reads, uint32_t batch_reads, qcat_result* out, int64_t* counts,
                                 int32_t* chosen_kit_slots);

/* replaces: BarcodeScanner.scan(read_sequence, ...) on sequences of ANY length (qcat/scanner_base.py:466-477;
 * scanner_epi2me.py:33-144, scanner_dual.py:35-146) -- the form scan_middle uses on read interiors
 * (scanner_base.py:479-519) and qcat/eval_full.py:199-203 on whole reads.  Every sequence is one window: all
 * templates of `kit` are aligned to the whole sequence, the barcode region (or sequence[:max_align_length]) is
 * scanned, and out[i] is what scan() returns -- template index, barcode index/indices, adapter_end, raw score and
 * denominator, exit_status 0 (dual: 1 and no adapter when either barcode is missing); no thresholds, trims are 0.
 * Windows up to max_align_length can equally go through qcat_scan_batch with QCAT_ENDS_5P (the fast kernels). */
int  qcat_scan_sequences(qcat_ctx* ctx, const qcat_kit* kit,
                         const uint8_t* bases, const uint64_t* offsets, uint32_t n_seqs, qcat_result* out);

/* replaces: parasail_sg / parasail_sg_stat as the reference's module-level helpers call them -- align_adapter
 * (qcat/scanner_base.py:191-220), align_adapter_identity (:144-188), find_highest_scoring_barcode (:108-117): n independent
 * semi-global alignments of query i = queries[q_offsets[i] .. q_offsets[i+1]) against target i (ASCII, any case; targets
 * up to QCAT_MAX_TEMPLATE_LEN letters), affine gaps, `matrix`[target code * 7 + query code] over the codes above.
 * score / end_query / end_ref follow rule R1 (oracle/qcat_oracle.c:100-108); with_stats != 0 also fills `matches` and
 * `length` (alignment columns) along ONE optimal path, chosen by the rule `with_stats` names (QCAT_STATS_*):
 *   QCAT_STATS_PARASAIL6  ties: diagonal, then the gap that consumes a QUERY letter (parasail's F), then the gap that
 *                         consumes a target letter (E); a match = equal MAPPED codes over the alphabet ATGCNX + '*' (two
 *                         different letters outside the alphabet both map to '*' and count) -- the adapter matrix,
 *                         align_adapter_identity (qcat/scanner_base.py:168-172, config.py:245);
 *   QCAT_STATS_PARASAIL5  the same over ATGCN + '*' (X maps to '*' too) -- the barcode matrix, find_highest_scoring_barcode
 *                         with compute_identity (scanner_base.py:105-117, config.py:26);
 *   QCAT_STATS_ROUND3     round 3's order (diagonal, E, F; a match = the same letter) -- kept for comparison.
 * The PARASAIL rules restate parasail 2.x's *_stats_striped_* kernels as recalled (case1 = H == H_dag, case2 = H == F,
 * HM = case1 ? H_dagM + match : case2 ? FM : EM; a gap is opened only when strictly better than extended); PARITY WITH
 * PARASAIL IS UNPINNED for these two numbers -- parasail is absent from this image and no reference test holds them (no
 * scanner path consumes them, scanner_base.py:141).  One switch, three implementations: this kernel (k_sg_align),
 * oracle/qcat_oracle.c qo_sg_stats, tests/golden/sg_independent.py sg_stats.  An empty query or target gives
 * end_query = end_ref = -1. */
enum { QCAT_STATS_NONE = 0, QCAT_STATS_PARASAIL6 = 1, QCAT_STATS_PARASAIL5 = 3, QCAT_STATS_ROUND3 = 5 };
typedef struct qcat_alignment { int32_t score, end_query, end_ref, matches, length; } qcat_alignment;
int  qcat_sg_align(qcat_ctx* ctx, const uint8_t* queries, const uint64_t* q_offsets,
                   const uint8_t* targets, const uint64_t* t_offsets, uint32_t n,
                   int32_t gap_open, int32_t gap_extend, const int8_t* matrix /* 49 */, int32_t with_stats,
                   qcat_alignment* out);

/* Same, plus one qcat_end_trace per scanned read end (2*n_reads entries, 5' then 3' per read;
 * n_reads entries with QCAT_ENDS_5P) and, if bc_rows != NULL, the raw score of EVERY barcode
 * alignment: bc_rows[((end * 2 + set) * row_stride) + b], row_stride >= largest set. */
int  qcat_scan_debug(qcat_ctx* ctx, const qcat_kit* kit,
                     const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads,
                     qcat_result* out, int64_t* counts,
                     qcat_end_trace* traces, int16_t* bc_rows, uint32_t row_stride);

/* Device-resident batches (benchmarks, pipelines that keep reads in HBM). */
int  qcat_batch_upload(qcat_ctx* ctx, const uint8_t* bases, const uint64_t* offsets,
                       uint32_t n_reads, qcat_batch** out);
/* Synthetic reads generated ON the device with the stateless SplitMix64 generator of
 * SURVEY.md 8(d) (qcat_amd/csrc/synth.h; the host twin is qcat_synth_read()). */
typedef struct qcat_synth_params {
    uint64_t seed;
    uint32_t n_reads;
    uint32_t insert_len;         /* 600 */
    uint32_t lead_min, lead_max; /* 5..40 (also used for the tail) */
    float    error_rate;         /* per-base, split equally sub/del/ins */
    float    no_adapter_fraction;/* 0.05 */
    int32_t  tpl_5p, tpl_3p;     /* template indices used to build the ends (-1: none) */
} qcat_synth_params;
int  qcat_batch_synthesize(qcat_ctx* ctx, const qcat_kit* kit, const qcat_synth_params* p,
                           qcat_batch** out);
void qcat_batch_destroy(qcat_batch* batch);
int  qcat_batch_info(const qcat_batch* batch, uint32_t* n_reads, uint64_t* n_bases);
/* copy a resident batch back (bases may be NULL to fetch offsets only). */
int  qcat_batch_download(qcat_ctx* ctx, const qcat_batch* batch, uint8_t* bases, uint64_t* offsets);
/* host twin of the device generator: writes read `index` into buf (cap bytes), returns length
 * or a negative status. */
int64_t qcat_synth_read(const qcat_kit* kit, const qcat_synth_params* p, uint64_t index,
                        uint8_t* buf, uint64_t cap);

/* Scan a resident batch; results and counts stay in device memory owned by ctx until the next
 * scan.  Enqueues on the context's stream and returns without synchronising. */
int  qcat_scan_resident(qcat_ctx* ctx, const qcat_kit* kit, const qcat_batch* batch);
int  qcat_ctx_synchronize(qcat_ctx* ctx);
/* copy the last scan's records / counts to the host (synchronises). */
int  qcat_ctx_fetch_results(qcat_ctx* ctx, qcat_result* out, uint32_t n_reads);
int  qcat_ctx_fetch_counts(qcat_ctx* ctx, int64_t* counts, int32_t n_buckets);
/* device pointer of the last scan's int64 count vector (for an RCCL all-reduce by the caller,
 * SURVEY.md 8e) and of the records. */
void* qcat_ctx_counts_devptr(qcat_ctx* ctx);
void* qcat_ctx_results_devptr(qcat_ctx* ctx);

/* the context's HIP stream (hipStream_t): a caller that enqueues its own work on it -- e.g. the
 * RCCL all-reduce of the count vector -- is ordered after the scan and before the next one
 * without a host synchronisation. */
void* qcat_ctx_stream(qcat_ctx* ctx);

/* how many qcat_scan_batch_auto / _ptrs calls of this context replayed their device work as a captured graph (a call
 * shaped like the one before it -- same kit, read count and compacted size, no buffer reallocated -- is ONE
 * hipGraphLaunch instead of ~45 kernel launches on twelve streams; QCAT_HIP_NO_GRAPH=1 turns that off).  Diagnostics
 * and tests; -1: null context. */
int64_t qcat_ctx_graph_replays(const qcat_ctx* ctx);

/* Diagnostics of the context's latest scan of the read ends (find_highest_scoring_barcode, qcat/scanner_base.py:63-141, on the
 * bit-sliced kernels of csrc/kernels_bitslice.inc): super-tiles of 2048 barcode alignments taken per hot class, summed over the
 * (template, set) groups -- out[0] = regions 1..5 bases short of the nominal length (clipped by the window: units padded at the
 * front, ABI 5), out[1] = nominal regions, out[2] = full windows; all zero when the batch was too small for the path.
 * Synchronises the context's stream.  Tests and diagnostics. */
int qcat_ctx_barcode_bitslice_tiles(qcat_ctx* ctx, uint32_t out[3]);
/* Read ends the context's latest scan put on the one-wave-per-alignment kernels (csrc/kernels_tiny.inc: batches of a handful of
 * reads -- BarcodeScanner.detect_barcode on one read, qcat/scanner_base.py:521-604); 0: the scan took another path; -1: null
 * context.  Tests and diagnostics. */
int64_t qcat_ctx_tiny_ends(const qcat_ctx* ctx);
/* Reads of the context's latest --detect-middle scan (BarcodeScanner.scan_middle inside detect_barcode, qcat/scanner_base.py:479-519,
 * :593-595) whose interior ran on the one-wave-per-alignment kernels: interiors beyond what the packed interior scan takes (more
 * than 16 384 letters).  Synchronises the context's stream.  Tests and diagnostics. */
int64_t qcat_ctx_middle_wave_reads(qcat_ctx* ctx);

/* Diagnostics of the context's latest --detect-middle scan (detect_barcode's interior scan, qcat/scanner_base.py:479-519,
 * :593-595): out[0] = tiles of 2048 interiors whose adapter scan ran in bit-sliced form (csrc/kernels_abs_mid.inc), out[1] =
 * such tiles in all, out[2] = tiles of 128 interiors left to the binary16 kernel (a letter outside A, C, G, T, a kit without
 * plans, no room in the plane buffer), out[3] = tiles of 128 in all; all zero when the batch was too small for the path or
 * QCAT_HIP_MIDDLE_NO_ABS=1.  Synchronises the context's stream.  Tests and diagnostics. */
int qcat_ctx_middle_bitslice_tiles(qcat_ctx* ctx, uint32_t out[4]);

/* Kernel timing, measured with hipEvents recorded on the context's stream around each kernel
 * phase: the AVERAGE over the qcat_scan_resident calls since the previous qcat_ctx_last_timing /
 * qcat_ctx_set_timing (a ring of 64 scans; scans need no host synchronisation between them, this
 * call synchronises once and empties the ring).  names[i] points to a static string; returns the
 * number of entries written (<= cap). */
int  qcat_ctx_last_timing(qcat_ctx* ctx, const char** names, float* ms, int cap);
/* enable/disable per-kernel event timing (off by default: events serialise nothing but cost
 * a few microseconds per launch). */
int  qcat_ctx_set_timing(qcat_ctx* ctx, int enabled);

/* ---- options (ABI 5) ----
 * The library's tuning / diagnostic switches (csrc/options.h holds the table with one line of description each;
 * qcat_option_count / _name / _doc enumerate it).  They are process-wide.  None changes a result: they move work between
 * kernels that give identical records, size buffers or print diagnostics -- the tests flip them to cross-check the device
 * paths.  The environment variable QCAT_HIP_<NAME> is read ONCE, when the library is loaded; afterwards an option only
 * changes through these calls (`name` with or without the QCAT_HIP_ prefix).  A flag is on when it is set and not 0.
 * qcat_get_option returns 1 / 0 for set / not set (negative: no such option); qcat_reset_options goes back to the
 * environment's values. */
int  qcat_set_option(const char* name, int64_t value);
int  qcat_clear_option(const char* name);
int  qcat_get_option(const char* name, int64_t* value);
void qcat_reset_options(void);
int  qcat_option_count(void);
const char* qcat_option_name(int index);
const char* qcat_option_doc(int index);
/* "hip" for this library.  (The CPU oracle behind the same ABI -- test infrastructure, oracle/qcat_cpu_abi.c -- answers
 * "cpu-oracle": the Python host refuses to run a product path on it.) */
const char* qcat_backend(void);

/* ---- native FASTQ ingest and egress (SURVEY.md 8f rank 2 at speed) ----
 * replaces: the per-file loop of the reference driver, qcat/cli.py:445-563 -- iter_fastx over FastqGeneralIterator
 * (:235-306), detect_barcode_batch per batch of 4000 reads (:500-513), trimming (:521-526), the minimum-length filter
 * (:527-530), TSV rows (:408-442) and the per-barcode / annotated FASTQ writers (:309-358).  The histogram (:366-383)
 * stays with the caller: it gets the records and the skipped flags back.
 * Only plain four-line ASCII FASTQ files qualify (one title, one sequence, one '+', one quality line per read, no '\r',
 * no trailing blanks, no byte >= 0x80) and -- round 4 -- plain two-line FASTA files ('>' title, ONE sequence line per read;
 * the writers then produce FASTA, cli.py:340-346): qcat_fastq_open decides by the first byte like the driver
 * (cli.py:248-262) and answers QCAT_ERR_UNSUPPORTED for anything Biopython would read differently (wrapped sequences,
 * blank lines ...), and the caller's own parser takes the file. */
typedef struct qcat_fastq qcat_fastq;
int  qcat_fastq_open(const char* path, qcat_fastq** out, uint64_t* n_reads, uint64_t* n_bytes);
void qcat_fastq_close(qcat_fastq* f);
/* where read r lies in the file (offsets from the first byte): title without '@', sequence; for tests */
int  qcat_fastq_read_info(const qcat_fastq* f, uint64_t r, uint64_t* title_off, uint32_t* title_len,
                          uint64_t* seq_off, uint32_t* seq_len);
typedef struct qcat_demux_opts {
    int32_t batch_size;        /* reads per detect_barcode_batch call (qcat/cli.py:500: 4000); matters under kit_auto only */
    int32_t kit_auto;          /* 1: per-batch kit vote + detect_barcode with the voted kit (scanner_base.py:714-733) */
    int32_t trim;              /* --trim (cli.py:521-526) */
    int32_t min_read_length;   /* --min-read-length (cli.py:527-530) */
    int32_t tsv;               /* --tsv: rows to tsv_fd (the caller writes the header line, cli.py:486-487) */
    int32_t tsv_fd;
    int32_t out_fd;            /* the annotated stream (-o / stdout) when out_dir is NULL and reads are written */
    const char* out_dir;       /* -b: one <barcode name>.fastq per barcode (existing directory), or NULL */
    /* what the writers print, per template of the kit (n_templates entries): AdapterLayout.kit, and per barcode of
     * sets[0] Barcode.name and Barcode.id; dual mode: the ids of sets[1] as well */
    const char* const* kit_name;
    const char* const* const* bc_name;
    const int32_t* const* bc_id;
    const int32_t* const* bc2_id;
    /* ABI 5 */
    int32_t filter_barcodes;   /* --filter-barcodes (cli.py:165-171 -> scanner_base.py:690-712, :730-731): per batch of batch_size reads the
                                * calls are counted by Barcode.id ("0": no barcode), and a call whose barcode has at most
                                * int(0.05 * the largest count) reads becomes the empty result (no barcode, no adapter, exit status 1,
                                * trims 0).  0 when the driver runs without batches (--no-batch calls detect_barcode, cli.py:504-509) */
    int32_t stream_reader;     /* qcat_fastq_demux_stream: how a segment's bytes are fetched -- 0: the default, 1: pread() into reused buffers,
                                * 2: a mapped window of the file per segment */
    uint64_t segment_bytes;    /* qcat_fastq_demux_stream: bytes of the file per segment (0: 256 MiB) */
    /* ABI 6: a SHARD of the file -- qcat_fastq_demux_stream handles the records in the bytes [range_begin, range_end) only
     * (both record starts, e.g. from qcat_fastq_batch_offsets; range_end 0: the end of the file).  Batches are counted from
     * range_begin, so a shard that starts at a batch boundary of the whole file votes and filters exactly like the one-rank run
     * (qcat/cli.py:500-513: vote and filter are per batch); the outputs are the shard's own (the caller gives every rank its own
     * descriptors / directory and strings the shards together in rank order), stats->next_offset stays a file offset. */
    uint64_t range_begin, range_end;
    /* ABI 6: the input as a DESCRIPTOR instead of a path (`path` may be NULL) -- the driver's `cat *.fastq | qcat -b out`
     * (README.md:104; qcat/cli.py:256-257 reads sys.stdin, assumed FASTQ).  input_fd - 1 is the descriptor (0 keeps the field's
     * zero default meaning "the path"): a regular file is handled like a path; anything else (a pipe) is read sequentially in
     * segments.  A stream cannot be handed back by offset: when the loop ends in front of a segment it does not take (see
     * below) the bytes already read and not handled are written to rest_fd - 1 (a descriptor of the caller, e.g. an unlinked
     * temporary file; required with a stream), stats->incomplete = 1, and the caller's parser reads that file and then the
     * rest of the stream.  A stream never answers QCAT_ERR_UNSUPPORTED for its content (its first bytes are consumed too). */
    int32_t input_fd, rest_fd;
} qcat_demux_opts;
typedef struct qcat_demux_stats {
    uint64_t n_reads, n_skipped, file_bytes;
    double parse_s, scan_s, write_s;       /* record splitting (qcat_fastq_open); upload + kernels + download; formatting + write() */
    double total_s;                        /* the call: the writers run beside the scan, so total_s < scan_s + write_s */
    /* ABI 5, qcat_fastq_demux_stream */
    uint64_t next_offset;                  /* file offset behind the last read that was handled (the file size unless `incomplete`) */
    int32_t incomplete;                    /* 1: the loop ended in front of a segment holding a record that is not plain (see there) */
    int32_t segments;
} qcat_demux_stats;
/* scans every read of the file with `kit` (QCAT_ENDS_BOTH) and writes the outputs; recs[r] / skipped[r] (n_reads entries
 * each, caller-owned) receive the record of read r and whether the minimum-length filter dropped it.
 * QCAT_ERR_UNSUPPORTED: simple mode, or kit_auto with a kit whose adapter pass cannot be resumed per kit. */
int  qcat_fastq_demux(qcat_fastq* f, qcat_ctx* ctx, const qcat_kit* kit, const qcat_demux_opts* opts,
                      qcat_result* recs, uint8_t* skipped, qcat_demux_stats* stats);

/* ---- the same loop over a file of any size in bounded host memory (ABI 5) ----
 * replaces: the per-file loop of the reference driver as it STREAMS -- iter_fastx yields one batch at a time
 * (qcat/cli.py:235-306), the loop scans and writes it (:500-552) and keeps nothing but the two histograms (:366-383,
 * :531-534).  The file is taken in segments (opts->segment_bytes) through a three-stage pipeline: segment k + 1 is fetched
 * (opts->stream_reader; the default, 2, points a window of ONE read-only mapping of the file at the segment and gives a
 * written segment's pages back; 1 copies it with pread() into reused buffers: 2-3 x slower here) and split into records
 * while segment k is scanned and segment k - 1 is written.  The mapped reader checks the file's size again before every
 * window (QCAT_ERR_IO "the file changed size" as the pread reader's short read), but a file truncated WHILE a window of it
 * is being read raises SIGBUS in the calling process -- as any mapped read does; callers whose inputs are still being
 * written set stream_reader = 1.  A segment is
 * cut at a whole batch of batch_size reads counted from the start of the file whenever batches matter (kit_auto,
 * filter_barcodes).  Kits created with scan_middle_adapter (--detect-middle, scanner_base.py:593-595) upload whole reads.
 * Instead of one record per read the caller gets what the driver keeps: the histograms of the reads that passed the
 * minimum-length filter, as counts per (template, barcode of set 0[, barcode of set 1]) and per template.
 * Round 6: a segment whose records are not all plain -- wrapped sequence / quality lines, \r\n line ends, blank lines, trailing
 * blanks -- is rewritten as plain records by the reference's parsers' rules (Biopython's FastqGeneralIterator /
 * SimpleFastaParser, qcat/cli.py:235-306) on one thread and handled like any other; "not plain" below means what even those
 * rules reject or what the caller's parser has to report itself (quality and sequence of different lengths, captions that
 * differ, blanks inside a sequence, an empty sequence, bytes outside ASCII, a lone \r).
 * A record that is not a plain four-line FASTQ / two-line FASTA record: in the first segment QCAT_ERR_UNSUPPORTED before
 * anything is written (as qcat_fastq_open); later the call ends in front of that record's segment -- a batch boundary --
 * with stats->incomplete = 1, stats->next_offset = the file offset of the segment's first record and stats->n_reads = the
 * reads handled so far, and the caller's own parser carries on from there.  On ANY failure stats->n_reads / ->segments say
 * how much had been written when it happened (QCAT_ERR_UNSUPPORTED with both 0: nothing was -- the caller may redo the
 * file itself; anything else is an error behind written output). */
typedef struct qcat_demux_hist {
    int32_t w0, w1;            /* in: row widths -- w0 >= the largest set 0, w1 >= the largest set 1 (dual mode), else 1 */
    int64_t* barcode;          /* out [n_templates * w0 * w1]: kept reads per (template t, barcode b, second barcode b2) at (t * w0 + b) * w1 + b2 */
    int64_t* adapter;          /* out [n_templates]: kept reads per template of the call */
    int64_t n_none;            /* out: kept reads without a barcode call */
    int64_t n_adapter_none;    /* out: kept reads without an adapter */
} qcat_demux_hist;
/* replaces: nothing in the reference (one process, one file); what N ranks need to split ONE file into whole batches of the
 * driver's loop (qcat/cli.py:500: the unit of the kit vote and of --filter-barcodes, SURVEY.md 8e "what does not shard").
 * One pass of the streamed reader over the file: (*offsets)[i] = the file offset of record i * batch_size, for i <
 * *n_batches = ceil(n_reads / batch_size), and (*offsets)[*n_batches] = where the plain records end (*next_offset: the file
 * size, or the offset of the segment that holds the first record the native loop does not take -- the caller's own parser
 * carries on there, after all shards).  The array is malloc()ed: qcat_free() it.  QCAT_ERR_UNSUPPORTED: not a plain file. */
int  qcat_fastq_batch_offsets(const char* path, uint32_t batch_size, uint64_t segment_bytes, uint64_t** offsets,
                              uint64_t* n_batches, uint64_t* n_reads, uint64_t* next_offset);
void qcat_free(void* p);
int  qcat_fastq_demux_stream(const char* path, qcat_ctx* ctx, const qcat_kit* kit, const qcat_demux_opts* opts,
                             qcat_demux_hist* hist, qcat_demux_stats* stats);
/* the reader stage alone (no device): reads and sequence letters of the file, taken in the same segments (batch_size > 0: cut at
 * whole batches); *next_offset as stats->next_offset above (the file size when every record is plain). */
int  qcat_fastq_stream_count(const char* path, uint64_t segment_bytes, uint32_t batch_size, int32_t stream_reader, uint64_t* n_reads,
                             uint64_t* n_bases, uint64_t* next_offset, uint32_t* n_segments);

/* ---- multi-GPU: reads are sharded by rank, the count vector is the only exchange ----
 * One communicator per (context, rank); RCCL underneath (librccl is opened on the first call here).
 * Ranks may be processes (one per GPU, the id travels by any side channel: a file, a socket, MPI)
 * or host threads of one process (one context per device).
 * replaces: nothing in the reference -- it is single-process; the vector that is reduced is the
 * histogram its driver accumulates over all reads (qcat/cli.py:366-383, scanner_base.py:680-689). */
enum { QCAT_COMM_ID_BYTES = 128, QCAT_COMM_MAX_VALUES = 64 };
enum { QCAT_REDUCE_SUM = 0, QCAT_REDUCE_MAX = 1 };
typedef struct qcat_comm qcat_comm;
/* rank 0 calls this once and hands the 128 bytes to every rank. */
int  qcat_comm_unique_id(uint8_t* id /* QCAT_COMM_ID_BYTES */);
/* collective over the n_ranks holders of `id`; binds the communicator to ctx's device. */
int  qcat_comm_create(qcat_ctx* ctx, int n_ranks, int rank, const uint8_t* id, qcat_comm** out);
void qcat_comm_destroy(qcat_comm* comm);
int  qcat_comm_info(const qcat_comm* comm, int* n_ranks, int* rank, int* device);
/* ncclAllReduce(SUM, int64), in place on the count vector of ctx's last scan, enqueued on ctx's
 * stream (no host synchronisation): afterwards qcat_ctx_fetch_counts returns the global counts. */
int  qcat_counts_allreduce(qcat_ctx* ctx, qcat_comm* comm);
/* harness helpers over the same communicator: n <= QCAT_COMM_MAX_VALUES host doubles reduced in
 * place over all ranks (synchronises), and a barrier that also drains ctx's stream on every rank. */
int  qcat_comm_allreduce_f64(qcat_ctx* ctx, qcat_comm* comm, double* values, int n, int op);
int  qcat_comm_barrier(qcat_ctx* ctx, qcat_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* QCAT_HIP_H */
