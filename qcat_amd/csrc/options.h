// options.h -- the library's tuning / diagnostic switches as ONE table of process-wide options (round 5).
//
// Rounds 1-4 read 57 environment variables with `getenv` inside the scan paths, on every call: behaviour depended on
// undocumented environment state, and a setenv() from another thread raced the reads.  Now: every switch is an entry of
// the table below, an atomic 64-bit value; the environment variable QCAT_HIP_<NAME> is read ONCE, when the library is
// loaded (so a shell can still steer an A/B run), and afterwards the value only changes through the C ABI --
// qcat_set_option / qcat_clear_option / qcat_get_option / qcat_reset_options (include/qcat_hip.h), which is what the tests
// use.  A flag option is ON when it is set and not 0.  None of them changes a result: they move work between kernels that
// give identical records (the tests flip them to cross-check the kernels), size buffers, or print diagnostics.
#ifndef QCAT_OPTIONS_H
#define QCAT_OPTIONS_H
#ifndef QCAT_RTC
#include <atomic>
#include <stdint.h>
#include <stdlib.h>

// X(name, what it does)
#define QCAT_OPTION_LIST(X) \
    X(NO_STATIC, "1: no generated static-letter kernels (binary16 table kernels instead)") \
    X(NO_STATIC_ADAPTER, "1: the adapter templates on the table kernels") \
    X(NO_FUSED_ADAPTER, "1: the two templates of a kit as two launches instead of the fused kernel") \
    X(NO_BARCODE_MULTI, "1: the static-letter barcode kernels of a small batch as launches of their own instead of one (k_barcode_multi)") \
    X(NO_ADAPTER_MULTI, "1: the static-letter adapter kernels of a small batch as launches of their own instead of one (k_adapter_multi)") \
    X(BARCODE_U16, "1 at kit creation: u16 lanes instead of exact-integer binary16 for the barcode tables") \
    X(NO_BITSLICE, "1: every barcode alignment on the binary16 kernels") \
    X(NO_BS_STATIC, "1: bit-sliced barcode kernels with the letters from memory") \
    X(BITSLICE_MIN, "barcode alignments from which the bit-sliced path is taken (default: 40 000 + 1 900 000 / barcodes; dual kits 70 000 + 3 500 000 / barcodes)") \
    X(BITSLICE_PAD, "jobs from which the rest of a hot class becomes a padded super-tile (0 / unset: never)") \
    X(BS_NO_SOLO, "1: no producer waves for the shared columns") \
    X(BS_NO_SHORT, "1: regions a few bases short of nominal stay on the binary16 kernels (no front-padded units)") \
    X(BS_SIDE, "1: ... on side streams whatever the batch size") \
    X(LEFTOVER_SIDE, "0 / 1: the left-over binary16 tiles behind / beside the bit-sliced launches") \
    X(BS_DRAW, "0 / 1: the waves of a bit-sliced workgroup take a unit's barcodes round robin / draw them from a counter (default 1; sets of 32 barcodes or more)") \
    X(BS_TRACE, "1: print the phase boundaries of one workgroup and the unit log of the bit-sliced launches") \
    X(BS_TRACE_UNITS, "1: ... and every unit") \
    X(BS_TRACE_WG, "the workgroup whose phase boundaries BS_TRACE prints (default 0)") \
    X(BS_TRACE_SEQ, "... from its n-th unit on (eight units; default 0)") \
    X(CHUNK_BARCODES, "barcodes per work unit of the binary16 barcode kernels") \
    X(ONE_QUEUE, "1: one unit queue instead of XCD-local ones") \
    X(ONE_STREAM, "1: no side streams") \
    X(RAWS, "1: keep the per-barcode raw score array (no summary keys)") \
    X(SUMMARY, "1: summary keys for small barcode sets as well") \
    X(NO_ADAPTER_BITSLICE, "1: the adapter scan on the binary16 kernels") \
    X(ADAPTER_BITSLICE_MIN, "read ends from which the bit-sliced adapter scan is taken") \
    X(ABS_STAGES, "2 / 4: one form of the adapter plans for every batch size") \
    X(FORCE_GENERIC, "1 at context creation: the general int32 kernel for everything") \
    X(MIDDLE_GENERIC, "1: --detect-middle on the general kernel") \
    X(MIDDLE_NO_ABS, "1: the interior adapter scan on the binary16 kernel") \
    X(MIDDLE_ABS_MIN, "slots from which the bit-sliced interior adapter scan is taken") \
    X(MIDDLE_ABS_ROWS, "rows of the interior's plane buffer") \
    X(MIDDLE_ABS_ONE_WAVE, "0 / 1: two-wave pipeline / one wave per big tile") \
    X(NO_PIPELINE, "1: no chunked host pipeline") \
    X(PIPELINE_CHUNK, "reads per chunk of the host pipeline") \
    X(PIPELINE_TRACE, "1: print the split of a pipelined call") \
    X(NO_TINY, "1: batches of a handful of read ends take the throughput kernels like every other batch") \
    X(TINY_MAX_ENDS, "largest batch (read ends, at most 4096) on the one-wave-per-alignment kernels (default: by the number of alignments, 20000)") \
    X(AUTO_CHUNK, "batches per call of the kit-auto file loop") \
    X(AUTO_WORKERS, "contexts of the kit-auto file loop") \
    X(NO_GRAPH, "1: host-buffer calls never replay a captured graph") \
    X(DEBUG_VOTE, "1: print the kit vote") \
    X(DEBUG_BINS, "1: print the jobs per length class") \
    X(DEBUG_REDO, "1: print how many alignments took the sequential arg-max")

// switches that only exist to A/B a variant that was measured and dropped (records: profiles/, docs/DESIGN_rounds_1_to_4.md).
// A default build cannot set them -- they are not in the table the C ABI lists, the environment is not consulted, and every
// opt_on() on one of them folds to `false` at compile time, so the variant's branch is not in the library; build with
// -DQCAT_AB (QCAT_EXTRA_HIPFLAGS=-DQCAT_AB python -c "import __graft_entry__ as g; g.build(force=True)") to get them back.
#define QCAT_AB_OPTION_LIST(X) \
    X(NO_QUADS, "1 at kit creation: static barcode chains as pairs only") \
    X(BS_STATIC_MIN, "super-tiles from which the generated bit-sliced kernels are taken") \
    X(BS_FROM_TILES, "1: bit-sliced units read tile images instead of the two-bit windows") \
    X(BS_NO_TAIL_SPLIT, "1: the last round of long units is not cut into barcode chunks") \
    X(BS_SERIAL, "1: the bit-sliced launches of a scan in line on the context's stream") \
    X(NO_SLIM, "1: the 60-byte per-end records instead of the packed result arrays") \
    X(NO_FILL_MERGE, "1: one fill launch per buffer instead of k_fill_multi") \
    X(EAGER_BYTES, "1: byte windows for every read end, not only for those with a letter outside A, C, G, T") \
    X(PACK_PLANES, "1: letter planes built inside the window kernel (measured slower)") \
    X(FIN_BLOCKS, "blocks of k_finalize") \
    X(FINISH_BLOCKS, "blocks of k_adapter_finish") \
    X(ABS_NO_SPLIT, "1: medium batches keep the fused two-template plan") \
    X(ABS_PRIO, "issue priority rotation of the bit-sliced adapter kernels") \
    X(ADAPTER_REVERSE, "1: the side-by-side adapter launches in reverse order") \
    X(MIDDLE_NO_BITSLICE, "1: the interior's barcode jobs on the binary16 kernels") \
    X(MIDDLE_ABS_WINDOWS, "0: the M-ends' first windows from the reads instead of the packed batch") \
    X(MIDDLE_ABS_WGS, "workgroups per CU of the interior adapter kernels") \
    X(MIDDLE_ABS_EARLY, "1: the packed batch at the start of the scan") \
    X(MIDDLE_ABS_PRIO, "issue priority rotation of the interior adapter kernels") \
    X(FULL_UPLOAD, "1: host-buffer calls upload whole reads") \
    X(ABS_SERIAL, "1: the bit-sliced adapter launches of a scan one after the other instead of side by side (A/B)") \
    X(NO_ZERO_COPY, "1: host-buffer calls of a handful of reads copy their staging to the device like bigger ones (A/B)") \
    X(STREAM_SYNC_RELEASE, "1: the file loop's reader gives a written segment's pages back itself (A/B: a thread of its own)")

enum QcatOpt {
#define X(N, D) QO_##N,
    QCAT_OPTION_LIST(X)
    QO_PUBLIC_COUNT,
    QO_AB_BASE_ = QO_PUBLIC_COUNT - 1,
    QCAT_AB_OPTION_LIST(X)
#undef X
    QO_COUNT
};
#ifdef QCAT_AB
constexpr int QO_SETTABLE = QO_COUNT;
#else
constexpr int QO_SETTABLE = QO_PUBLIC_COUNT;          // the A/B switches are dead: never set, folded away where they are read
#endif

constexpr int64_t QOPT_UNSET = INT64_MIN;
extern std::atomic<int64_t> g_qcat_opt[QO_COUNT];          // (qcat_hip.hip)

struct QOptVal {                                            // an option's state; reads like the `const char*` of `getenv` did
    bool set;
    int64_t v;
    explicit operator bool() const { return set; }
};
inline QOptVal qopt_get(QcatOpt o) { if ((int)o >= QO_SETTABLE) return QOptVal{false, 0}; const int64_t v = g_qcat_opt[o].load(std::memory_order_relaxed); return QOptVal{v != QOPT_UNSET, v == QOPT_UNSET ? 0 : v}; }
inline bool opt_is_set(QcatOpt o) { return (int)o < QO_SETTABLE && g_qcat_opt[o].load(std::memory_order_relaxed) != QOPT_UNSET; }
inline bool opt_on(QcatOpt o) { if ((int)o >= QO_SETTABLE) return false; const int64_t v = g_qcat_opt[o].load(std::memory_order_relaxed); return v != QOPT_UNSET && v != 0; }
inline int64_t opt_val(QcatOpt o, int64_t dflt) { if ((int)o >= QO_SETTABLE) return dflt; const int64_t v = g_qcat_opt[o].load(std::memory_order_relaxed); return v == QOPT_UNSET ? dflt : v; }
inline int atoi(const QOptVal& o) { return (int)o.v; }
inline long long atoll(const QOptVal& o) { return (long long)o.v; }
#endif
#endif
