"""Host side of the drop-in boundary: ``BarcodeScanner`` with the reference's constructor
keywords and methods (``qcat/scanner_base.py:410-733``), executing on the MI355X through the
C ABI of ``include/qcat_hip.h``.

What stays in Python is only what the reference keeps per *batch* or per *object*: kit
selection (``:415-447``), the per-batch kit vote and low-abundance filter (``:662-733``) and
the mapping of native index records back to ``Barcode`` / ``AdapterLayout`` objects.  All
per-read work of ``detect_barcode`` (``:521-604``) -- windows, adapter and barcode alignments,
thresholds, conflict detection, trims -- happens inside one native call per batch.
"""
import logging
import operator

import numpy as np

from . import adapters, config, native
from .adapters import Barcode
import threading  # noqa: E402

_SHARED_CONTEXTS = {}            # (thread id, device) -> native.NativeContext, see BarcodeScanner._context


def release_contexts():
    """destroy the contexts the scanners of this process share (device staging, pinned buffers); scanners made afterwards
    start new ones"""
    _SHARED_CONTEXTS.clear()



def build_return_dict(best_barcode, best_barcode_score, best_adapter, best_adapter_end,
                      exit_status, trim5p=0, trim3p=0):
    """Result dictionary with the reference's keys (``qcat/scanner_base.py:381-388``)."""
    return {"barcode": best_barcode,
            "barcode_score": best_barcode_score,
            "adapter": best_adapter,
            "adapter_end": best_adapter_end,
            "trim5p": trim5p,
            "trim3p": trim3p,
            "exit_status": exit_status}


def empty_return_dict():
    """``qcat/scanner_base.py:393-407``."""
    return build_return_dict(None, 0.0, None, 0, 1, trim5p=0, trim3p=0)


def extract_align_sequence(read_sequence, rev_comp, length):
    """Window of the read that is scanned (``qcat/scanner_base.py:223-244``); kept on the host
    for API compatibility -- the device computes the same windows itself."""
    from .utils import revcomp
    seq = read_sequence if read_sequence else ""
    if length > 0:
        seq = revcomp(seq[-length:]) if rev_comp else seq[:length]
    return seq


# ---------------------------------------------------------------------------------------------------------------
# The reference's module-level helpers (``qcat/scanner_base.py:29-359``) with their signatures and return values, for
# callers that use them directly (``qcat/test/test_barcode.py:291-304`` calls ``find_best_adapter_template``; the eval
# tools call ``eval_adapter_template``).  The scanners above never go through them -- a batch is one native call -- but
# every alignment here runs on the device as well (``qcat_sg_align``, include/qcat_hip.h): there is no CPU path.
# ---------------------------------------------------------------------------------------------------------------
class Alignment(object):
    """What the helpers hand out where the reference hands out a parasail ``Result``: ``score``, ``end_query``,
    ``end_ref``, and from the ``_stat`` form ``matches`` / ``length``."""
    __slots__ = ("score", "end_query", "end_ref", "matches", "length")

    def __init__(self, rec):
        self.score, self.end_query, self.end_ref = int(rec["score"]), int(rec["end_query"]), int(rec["end_ref"])
        self.matches, self.length = int(rec["matches"]), int(rec["length"])


def _ctx(device=0):
    """the calling thread's context on `device` -- the one the scanner objects share (BarcodeScanner._context)"""
    key = (threading.get_ident(), device)
    ctx = _SHARED_CONTEXTS.get(key)
    if ctx is None:
        ctx = _SHARED_CONTEXTS[key] = native.NativeContext(device)
    return ctx


def _sg(queries, targets, gap_open, gap_extend, matrix, with_stats=False):
    return native.sg_align(_ctx(), queries, targets, gap_open, gap_extend, matrix.table, with_stats=with_stats)


def extract_barcode_region(read_sequence, adapter_template, barcode_set_index, alignment_stop_ref, qcat_config):
    """``qcat/scanner_base.py:29-60``: the slice of the window the barcode of the aligned adapter lies in, widened by
    ``extracted_barcode_extension`` on both sides (Python slice semantics, negative indices wrap)."""
    barcode_end = adapter_template.get_barcode_end(barcode_set_index)
    barcode_length = adapter_template.get_barcode_length(barcode_set_index)
    end_ref = alignment_stop_ref - (adapter_template.get_adapter_length() - barcode_end) + 1
    start_ref = end_ref - barcode_length
    start_ref -= min(qcat_config.extracted_barcode_extension, start_ref)
    end_ref += min(qcat_config.extracted_barcode_extension, len(read_sequence) - end_ref)
    return read_sequence[start_ref:end_ref + 1]


def find_highest_scoring_barcode(barcode_region_read, barcode_set, qcat_config, upstream_context="",
                                 downstream_context="", compute_identity=False):
    """``qcat/scanner_base.py:63-141``: every barcode of the list (with its contexts) aligned to the region, the
    highest normalised score wins with the reference's ``not max_score`` replacement rule; returns
    ``(barcode, score, score, end_query)`` -- the reference returns the score in the identity slot too (``:141``)."""
    max_barcode, q_score, max_identity, max_end = None, 0, 0.0, -1
    if not barcode_region_read:
        return max_barcode, q_score, max_identity, max_end
    targets = [upstream_context + b.sequence + downstream_context for b in barcode_set]
    al = _sg([barcode_region_read] * len(targets), targets, 1, 1, qcat_config.matrix_barcode,
             with_stats=native.STATS_PARASAIL5 if compute_identity else False)      # (the barcode matrix's alphabet is ATGCN)
    max_score = None
    for b, target, a in zip(barcode_set, targets, al):
        score = int(a["score"]) * 100.0 / (1.0 * len(target))
        if not max_score or max_score < score:
            max_score, max_barcode, max_end = score, b, int(a["end_query"])
    return max_barcode, max_score, max_score, max_end


def align_adapter_identity(adapter_sequence, adapter_length, read_sequence, barcode_length, qcat_config):
    """``qcat/scanner_base.py:144-188``: the adapter alignment with its identity = matches / (alignment columns - barcode
    length); an alignment that covers less than 85 % of the adapter is discarded ``(None, 0.0)``.  ``matches`` / ``length``
    follow one optimal path, chosen as parasail's stats kernels are recalled to choose it (include/qcat_hip.h,
    QCAT_STATS_PARASAIL6: diagonal, then the gap in the read, then the gap in the adapter; matches over mapped codes --
    parity with parasail is unpinned)."""
    if not read_sequence or not adapter_sequence:
        return None, 0.0
    a = Alignment(_sg([read_sequence], [adapter_sequence], qcat_config.gap_open, qcat_config.gap_extend, qcat_config.matrix,
                      with_stats=native.STATS_PARASAIL6)[0])
    if a.length < (adapter_length * 0.85):
        return None, 0.0
    return a, float(a.matches) / float(a.length - barcode_length)


def align_adapter(adapter_sequence, read_sequence, qcat_config):
    """``qcat/scanner_base.py:191-220``."""
    if not read_sequence or not adapter_sequence:
        return None, 0.0
    return Alignment(_sg([read_sequence], [adapter_sequence], qcat_config.gap_open, qcat_config.gap_extend, qcat_config.matrix)[0]), 0.0


def compute_adapter_identity(adapter_template, read_sequence, qcat_config):
    """``qcat/scanner_base.py:247-255``."""
    return align_adapter_identity(adapter_template.get_adapter_sequences(), adapter_template.get_adapter_length(), read_sequence,
                                  adapter_template.get_barcode_length(0) + adapter_template.get_barcode_length(1), qcat_config)[1]


def eval_adapter_template(adapter_template, read_sequence, qcat_config, identity=True):
    """``qcat/scanner_base.py:258-296``: (end_query, identity, raw score) of one template; (-1, 0.0, -1) without an
    alignment."""
    if identity:
        aligned, ident = align_adapter_identity(adapter_template.get_adapter_sequences(), adapter_template.get_adapter_length(),
                                                read_sequence,
                                                adapter_template.get_barcode_length(0) + adapter_template.get_barcode_length(1),
                                                qcat_config)
    else:
        aligned, ident = align_adapter(adapter_template.get_adapter_sequences(), read_sequence, qcat_config)
    if aligned is None:
        return -1, ident, -1
    return aligned.end_query, ident, aligned.score


def get_norm_socre(template, score, qcat_config):
    """``qcat/scanner_base.py:299-310`` (the reference's spelling)."""
    bc_len = template.get_barcode_length(0) + template.get_barcode_length(1)
    a_len = template.get_adapter_length()
    return score * 100.0 / ((a_len - bc_len) * qcat_config.match + bc_len * qcat_config.nmatch)


def find_best_adapter_template(adapter_templates, read_sequence, qcat_config):
    """``qcat/scanner_base.py:313-359``: (index, end_query, normalised score) of the best template, the first one on
    ties; (-1, -1, -1.0) for empty inputs.  All templates are aligned in one device call."""
    best_score, best_end, best_tpl = -1.0, -1, -1
    if not adapter_templates or not read_sequence:
        return best_tpl, best_end, best_score
    if not isinstance(adapter_templates, list):
        adapter_templates = [adapter_templates]
    idx = [i for i, t in enumerate(adapter_templates) if t.get_adapter_sequences()]
    al = _sg([read_sequence] * len(idx), [adapter_templates[i].get_adapter_sequences() for i in idx],
             qcat_config.gap_open, qcat_config.gap_extend, qcat_config.matrix)
    for i, a in zip(idx, al):
        score = get_norm_socre(adapter_templates[i], int(a["score"]), qcat_config)
        if best_score < score:
            best_score, best_tpl, best_end = score, i, int(a["end_query"])
    return best_tpl, best_end, best_score


class BarcodeScanner(object):
    """Abstract base class of the MI355X scanners (mirror of ``qcat.scanner_base.BarcodeScanner``)."""

    #: "epi2me" or "dual": which scan() the native library reproduces
    _native_mode = None

    def __init__(self, min_quality, kit_name, kit_folder=None,
                 enable_filter_barcodes=False, scan_middle_adapter=False, device=0):
        available_kits = adapters.populate_adapter_layouts(kit_folder)
        self.min_quality = min_quality
        self.layouts = []
        self.override_kit_name = None
        self.enable_filter_barcodes = enable_filter_barcodes
        self.scan_middle_adapter = scan_middle_adapter
        self.device = device
        if kit_name and kit_name.lower() != "auto":
            wanted = kit_name.lower()
            self.layouts = [l for l in available_kits if l.kit.lower() == wanted]
        else:
            self.layouts = [l for l in available_kits if l.auto_detect]
        self._kits = {}
        self._ctx = None

    # -- registry hooks -------------------------------------------------------------------------
    @staticmethod
    def get_name():
        raise NotImplementedError("Abstract class")

    def barcode_count(self):
        raise NotImplementedError("Abstract class")

    # -- native plumbing ------------------------------------------------------------------------
    def descriptor(self, layouts=None, qcat_config=None, ends=native.ENDS_BOTH, scan_middle=None,
                   min_read_length=0, trim=False):
        """KitDescriptor for ``layouts`` (default: this scanner's) -- also used by the tests to
        drive the CPU oracle with exactly the product's descriptor."""
        if qcat_config is None:
            qcat_config = config.qcatConfig()
        if layouts is None:
            layouts = self.layouts
        if scan_middle is None:
            scan_middle = self.scan_middle_adapter and ends == native.ENDS_BOTH
        return native.KitDescriptor(layouts, qcat_config, mode=self._native_mode,
                                    min_quality=self.min_quality, ends=ends, scan_middle=scan_middle,
                                    min_read_length=min_read_length, trim=trim)

    def _native_kit(self, layouts, qcat_config, ends):
        key = (tuple(id(l) for l in layouts), qcat_config.fingerprint(), ends, self.min_quality,
               bool(self.scan_middle_adapter), native.get_r1_rule())
        kit = self._kits.get(key)
        if kit is None:
            kit = native.NativeKit(self.descriptor(layouts, qcat_config, ends))
            self._kits[key] = kit
        return kit

    def _context(self):
        """the calling thread's context on this scanner's device.  Scanner objects SHARE it (round 6): a context owns the device
        staging and the pinned buffers of its calls, and making them anew for every scanner object -- the driver makes one per
        run -- was 80 ms in front of a run's first scan and 50 ms of hipFree behind it; `release_contexts()` gives them back."""
        if self._ctx is None:
            key = (threading.get_ident(), self.device)
            ctx = _SHARED_CONTEXTS.get(key)
            if ctx is None:
                ctx = _SHARED_CONTEXTS[key] = native.NativeContext(self.device)
            self._ctx = ctx
        return self._ctx

    def _record_to_dict(self, rec, layouts):
        adapter = layouts[rec["adapter_idx"]] if rec["adapter_idx"] >= 0 else None
        barcode, score = None, 0.0
        if rec["barcode_idx"] >= 0:
            owner = layouts[rec["adapter_idx"]]
            first = owner.get_barcode_set(0)[rec["barcode_idx"]]
            if rec["barcode2_idx"] >= 0:
                second = owner.get_barcode_set(1)[rec["barcode2_idx"]]
                barcode = Barcode("barcode{:02d}/{:02d}".format(first.id, second.id),
                                  "{}/{}".format(first.id, second.id), None, True)
            else:
                barcode = first
            score = int(rec["raw_score"]) * 100.0 / (1.0 * int(rec["score_den"]))
        return build_return_dict(barcode, score, adapter, int(rec["adapter_end"]),
                                 int(rec["exit_status"]), trim5p=int(rec["trim5p"]),
                                 trim3p=int(rec["trim3p"]))

    def _records_to_dicts(self, recs, layouts):
        """vectorised ``_record_to_dict``: the object lookups (barcode, adapter) are numpy takes from small object
        tables, the columns become plain Python lists once, and the loop only builds the dicts -- the host side of a
        4000-read batch is otherwise dominated by per-field numpy record access (1.4 ms -> 0.9 ms per batch; seven-key
        dicts cost ~0.2 us each whatever is done around them)"""
        n = len(recs)
        if n == 0:
            return []
        aidx = recs["adapter_idx"].astype(np.intp)
        bidx = recs["barcode_idx"].astype(np.intp)
        key = tuple(id(l) for l in layouts)
        tbl = getattr(self, "_dict_tables", None)
        if tbl is None or tbl[0] != key:         # (key, barcode table, adapter table[, the two as plain lists for the C helper])
            width = 1 + max([len(l.get_barcode_set(0) or ()) if l.barcode_set_1 is not None else 0 for l in layouts] + [0])
            bars = np.empty((len(layouts) + 1, width), dtype=object)          # row 0 / column 0: None (index -1)
            for t, lay in enumerate(layouts):
                if lay.barcode_set_1 is not None:
                    for j, b in enumerate(lay.get_barcode_set(0)):
                        bars[t + 1, j + 1] = b
            ads = np.empty(len(layouts) + 1, dtype=object)
            ads[1:] = layouts
            tbl = self._dict_tables = (key, bars, ads)
        bars, ads = tbl[1], tbl[2]
        if native._pyglue is not None:                                        # the same dicts built in C (csrc/pyglue.c)
            if len(tbl) < 5:
                tbl = self._dict_tables = tbl + (bars.tolist(), ads.tolist())
            got = native._pyglue.records_to_dicts(np.ascontiguousarray(recs), tbl[3], tbl[4])
            if got is not None:                                               # (None: dual records, a Barcode per pair below)
                return got
        adapters = ads[aidx + 1].tolist()
        # the same IEEE double expression as scanner_base.py:119: raw * 100.0 / (1.0 * den)
        den = np.maximum(recs["score_den"].astype(np.float64), 1.0)
        score = np.where(bidx >= 0, recs["raw_score"].astype(np.float64) * 100.0 / (1.0 * den), 0.0).tolist()
        b2idx = recs["barcode2_idx"]
        if (b2idx >= 0).any():                                                # dual kits: a name per pair of ids
            b2 = b2idx.tolist()
            bl = bidx.tolist()
            al = aidx.tolist()
            sets1 = [lay.get_barcode_set(1) if getattr(lay, "barcode_set_2", None) else None for lay in layouts]
            barcodes = []
            for i in range(n):
                if bl[i] < 0:
                    barcodes.append(None)
                    continue
                first = bars[al[i] + 1, bl[i] + 1]
                if b2[i] >= 0:
                    second = sets1[al[i]][b2[i]]
                    barcodes.append(Barcode("barcode{:02d}/{:02d}".format(first.id, second.id),
                                            "{}/{}".format(first.id, second.id), None, True))
                else:
                    barcodes.append(first)
        else:
            barcodes = bars[np.where(bidx >= 0, aidx + 1, 0), bidx + 1].tolist()
        return [{"barcode": b, "barcode_score": sc, "adapter": a, "adapter_end": e, "trim5p": t5, "trim3p": t3, "exit_status": st}
                for b, sc, a, e, t5, t3, st in zip(barcodes, score, adapters, recs["adapter_end"].tolist(), recs["trim5p"].tolist(),
                                                   recs["trim3p"].tolist(), recs["exit_status"].tolist())]

    def _run(self, read_sequences, layouts, qcat_config, ends=native.ENDS_BOTH):
        if not layouts:
            # the reference indexes an empty template list here (IndexError)
            raise IndexError("list index out of range")
        kit = self._native_kit(layouts, qcat_config, ends)
        bases, offsets = native.pack_reads(read_sequences)
        recs = self._context().scan(kit, bases, offsets)
        return self._records_to_dicts(recs, layouts)

    # -- reference API ----------------------------------------------------------------------------
    def scan(self, read_sequence, read_qualities, barcoding_kits, non_barocding_kits,
             qcat_config=None):
        """``scan()`` of one window (``qcat/scanner_epi2me.py:33``, ``scanner_dual.py:35``)."""
        if qcat_config is None:
            qcat_config = config.qcatConfig()
        if not isinstance(barcoding_kits, list):
            barcoding_kits = [barcoding_kits]
        window = (read_sequence or "")
        if len(window) > qcat_config.max_align_length:
            # a sequence longer than the end windows (scan_middle's read interiors, eval tools): every
            # template is aligned to the WHOLE sequence, as the reference does -- qcat_scan_sequences
            return self._scan_sequences([window], barcoding_kits, qcat_config)[0]
        return self._run([window], barcoding_kits, qcat_config, ends=native.ENDS_5P)[0]

    def _scan_sequences(self, sequences, layouts, qcat_config):
        """scan() of whole sequences of any length, one native call for the list."""
        if not layouts:
            raise IndexError("list index out of range")
        kit = self._native_kit(layouts, qcat_config, native.ENDS_5P)
        bases, offsets = native.pack_reads(sequences)
        recs = self._context().scan_sequences(kit, bases, offsets)
        return self._records_to_dicts(recs, layouts)

    def scan_middle(self, sequence, kit_name, qcat_config):
        """``qcat/scanner_base.py:479-519``: does the read interior ``sequence[n:-n]`` (or its reverse
        complement) carry a barcoded adapter of kit ``kit_name`` with barcode_score >= 50?  Both strands
        go to the device in one call; the reference's short-circuit only saves it work."""
        from .utils import revcomp
        detected_adapters = self.get_adapters(kit_name)
        n = qcat_config.max_align_length
        middle = (sequence or "")[n:-n]
        fwd, rev = self._scan_sequences([middle, revcomp(middle)], detected_adapters, qcat_config)
        for strand in (fwd, rev):
            if strand and not strand["barcode_score"] < 50.0:
                return True
        return False

    def detect_barcode(self, read_sequence, read_qualities=None, qcat_config=None):
        if qcat_config is None:
            qcat_config = config.qcatConfig()
        kits = self.layouts if not self.override_kit_name else self.get_adapters(self.override_kit_name)
        return self._run([read_sequence], kits, qcat_config)[0]

    def get_adapters(self, kit_name):
        return [l for l in self.layouts if kit_name.lower() == l.kit.lower()]

    def get_adapter(self, kit_name):
        for layout in self.layouts:
            if kit_name.lower() == layout.kit.lower():
                return layout

    @staticmethod
    def update_kit_count(adapter, adapter_counts):
        key = adapter.kit if adapter else "none"
        adapter_counts[key] = adapter_counts.get(key, 0) + 1

    @staticmethod
    def get_most_abundant_kits(adapter_counts):
        if not adapter_counts:
            return None
        return sorted(adapter_counts.items(), key=operator.itemgetter(1), reverse=True)[0][0]

    def detect_kit(self, read_sequences, qcat_config=None):
        """Per-batch kit vote (``qcat/scanner_base.py:662-678``): every read votes for the kit of
        the adapter template that scores best at its better end (``scan_ends``, ``:632-642``); the
        alignments run on the GPU (``qcat_detect_kit``), the fold onto kit names and the
        first-appearance tie-break of the reference's dict + stable sort stay here."""
        if qcat_config is None:
            qcat_config = config.qcatConfig()
        if not read_sequences:
            return None, []
        if not self.layouts:
            raise IndexError("list index out of range")
        if len(set(l.kit for l in self.layouts)) == 1:
            return self.layouts[0].kit, []          # one kit selected: every read can only vote for it
        kit = self._native_kit(self.layouts, qcat_config, native.ENDS_BOTH)
        bases, offsets = native.pack_reads(read_sequences)
        votes, first = self._context().detect_kit(kit, bases, offsets)
        counts, seen = {}, {}
        for t, lay in enumerate(self.layouts):
            if votes[t]:
                counts[lay.kit] = counts.get(lay.kit, 0) + int(votes[t])
                seen[lay.kit] = min(seen.get(lay.kit, len(read_sequences)), int(first[t]))
        if not counts:
            return None, []
        # sorted(..., reverse=True) is stable: equal counts keep dict insertion (= first vote) order
        best = sorted(counts, key=lambda kname: (-counts[kname], seen[kname]))[0]
        return best, []

    def _batch_auto(self, read_sequences, n, qcat_config):
        """Kit auto in batch mode as ONE native call (qcat_scan_batch_auto): the vote over all
        auto-detect templates and detect_barcode with the voted kit share one adapter pass.  None when
        it does not apply (one kit only, --detect-middle, or a kit the library cannot resume)."""
        if (not read_sequences or not self.layouts or len(set(l.kit for l in self.layouts)) == 1
                or self.scan_middle_adapter):
            return None
        kit = self._native_kit(self.layouts, qcat_config, native.ENDS_BOTH)
        views = native.read_views(read_sequences)         # the str objects' own buffers (csrc/pyglue.c), or None
        if views is not None:
            got = self._context().scan_auto_views(kit, views, len(read_sequences))
        else:
            bases, offsets = native.pack_reads(read_sequences)
            got = self._context().scan_auto(kit, bases, offsets)
        if got is None:
            return None
        recs, _slot = got
        return self._records_to_dicts(recs[:n], self.layouts)

    @staticmethod
    def update_barcode_count(result, barcode_count):
        key = result["barcode"].id if result and result["barcode"] else "0"
        barcode_count[key] = barcode_count.get(key, 0) + 1

    @staticmethod
    def get_valid(barcode_counts, min_perc=0.20):
        top = max(list(barcode_counts.values()) + [0])
        floor = int(top * min_perc)
        return [bc for bc, count in barcode_counts.items() if count > floor]

    def filter_barcodes(self, barcode_count, results):
        valid = self.get_valid(barcode_count, 0.05)
        for i, res in enumerate(results):
            if res and res["barcode"] and res["barcode"].id not in valid:
                results[i] = empty_return_dict()
        return results

    def detect_barcode_batch(self, read_sequences, read_qualities=[None], qcat_config=None):
        if qcat_config is None:
            qcat_config = config.qcatConfig()
        # zip() in the reference truncates to the shorter list (R7)
        n = min(len(read_sequences), len(read_qualities))
        results = self._batch_auto(read_sequences, n, qcat_config)
        if results is None:
            kit_name, _ = self.detect_kit(read_sequences, qcat_config)
            kits = self.layouts if not kit_name else self.get_adapters(kit_name)
            results = self._run(list(read_sequences[:n]), kits, qcat_config) if n else []
        # (the reference counts the barcodes of every batch and looks at the counts only under --filter-barcodes,
        # qcat/scanner_base.py:716-731; the count is local to the call, so without the filter the 4000 calls of
        # update_barcode_count -- 0.3-0.4 ms of a 2 ms batch call -- change nothing and are left out)
        if self.enable_filter_barcodes:
            barcode_count = {}
            for res in results:
                self.update_barcode_count(res, barcode_count)
            results = self.filter_barcodes(barcode_count, results)
        return results
