"""ctypes binding of the C ABI in ``include/qcat_hip.h`` (no PyTorch, no cffi build step).

``KitDescriptor`` flattens a list of :class:`~qcat_amd.layout.AdapterLayout` objects plus a
:class:`~qcat_amd.config.qcatConfig` into the ``qcat_kit_desc`` struct; ``HipLibrary`` loads
``libqcat_hip.so`` (built in-tree by ``__graft_entry__.build()``) and raises if it is missing --
there is no CPU fallback on the product path.
"""
import ctypes as C
import os
import sys
import threading

import numpy as np

ABI_VERSION = 6

# Rule R1 -- which of the reference's two alignment routines places an alignment's END (include/qcat_hip.h QCAT_R1_*).
# The reference binds `parasail.sg_striped_32` at import when parasail reports SSE2 and plain `parasail.sg` otherwise
# (qcat/scanner_base.py:20-26); the equivalent here is this module-level setting, read when a kit descriptor is built:
# "striped" (default: what every x86 host runs) or "scalar".  QCAT_R1_RULE in the environment sets the default.
R1_STRIPED, R1_SCALAR = 0, 1
SG_R1_SCALAR = 0x100            # OR-ed into qcat_sg_align's with_stats
_R1_NAMES = {"striped": R1_STRIPED, "sg_striped_32": R1_STRIPED, "scalar": R1_SCALAR, "sg": R1_SCALAR}
_r1_rule = [_R1_NAMES.get(os.environ.get("QCAT_R1_RULE", "striped").lower(), R1_STRIPED)]


def set_r1_rule(rule):
    """'striped' / 'scalar' (or R1_STRIPED / R1_SCALAR): the end-position rule of every kit descriptor built from now on and of
    the module-level alignment helpers.  Returns the previous rule."""
    old = _r1_rule[0]
    _r1_rule[0] = _R1_NAMES[rule.lower()] if isinstance(rule, str) else (R1_SCALAR if int(rule) else R1_STRIPED)
    return old


def get_r1_rule():
    return _r1_rule[0]
MODE_EPI2ME, MODE_DUAL, MODE_SIMPLE = 0, 1, 2
ENDS_5P, ENDS_BOTH = 1, 3
MAX_TEMPLATES = 16

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QCAT_HIP_LIBRARY") or os.path.join(_HERE, "csrc", "libqcat_hip.so")   # (the override: A/B runs of two builds)


class BarcodeSetDesc(C.Structure):
    _fields_ = [("sequences", C.c_char_p), ("ids", C.POINTER(C.c_int32)),
                ("n", C.c_int32), ("barcode_len", C.c_int32), ("lengths", C.POINTER(C.c_int32))]


class TemplateDesc(C.Structure):
    _fields_ = [("sequence", C.c_char_p), ("length", C.c_int32), ("trim_offset", C.c_int32),
                ("is_double_barcode", C.c_int32), ("kit_slot", C.c_int32),
                ("bc_start", C.c_int32 * 2), ("bc_end", C.c_int32 * 2), ("bc_len", C.c_int32 * 2),
                ("sets", BarcodeSetDesc * 2)]


class KitDesc(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("mode", C.c_int32), ("ends", C.c_int32),
                ("n_templates", C.c_int32), ("templates", C.POINTER(TemplateDesc)),
                ("match", C.c_int32), ("nmatch", C.c_int32),
                ("gap_open", C.c_int32), ("gap_extend", C.c_int32),
                ("max_align_length", C.c_int32), ("extracted_barcode_extension", C.c_int32),
                ("barcode_context_length", C.c_int32),
                ("adapter_matrix", C.c_int8 * 49), ("barcode_matrix", C.c_int8 * 49),
                ("min_quality", C.c_double), ("conflict_min_score", C.c_double),
                ("region_min_adapter_score", C.c_double),
                ("n_barcode_slots", C.c_int32), ("n_kit_slots", C.c_int32),
                ("scan_middle_adapter", C.c_int32), ("middle_min_score", C.c_double),
                ("min_read_length", C.c_int32), ("trim_reads", C.c_int32), ("r1_rule", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("barcode_idx", C.c_int16), ("barcode2_idx", C.c_int16),
                ("adapter_idx", C.c_int16), ("exit_status", C.c_int16),
                ("adapter_end", C.c_int32), ("trim5p", C.c_int32), ("trim3p", C.c_int32),
                ("raw_score", C.c_int16), ("score_den", C.c_int16)]


RESULT_DTYPE = np.dtype([("barcode_idx", "<i2"), ("barcode2_idx", "<i2"), ("adapter_idx", "<i2"),
                         ("exit_status", "<i2"), ("adapter_end", "<i4"), ("trim5p", "<i4"),
                         ("trim3p", "<i4"), ("raw_score", "<i2"), ("score_den", "<i2")])
assert RESULT_DTYPE.itemsize == 24 and C.sizeof(Result) == 24

TRACE_DTYPE = np.dtype([("window_len", "<i4"), ("tpl_raw", "<i4", (MAX_TEMPLATES,)),
                        ("tpl_end", "<i4", (MAX_TEMPLATES,)), ("best_tpl", "<i4"),
                        ("best_end", "<i4"), ("best_raw", "<i4"), ("used_tpl", "<i4"),
                        ("region_path", "<i4"), ("region_start", "<i4", (2,)),
                        ("region_len", "<i4", (2,)), ("bc_idx", "<i4", (2,)),
                        ("bc_raw", "<i4", (2,)), ("adapter_end", "<i4")])


class SynthParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_reads", C.c_uint32), ("insert_len", C.c_uint32),
                ("lead_min", C.c_uint32), ("lead_max", C.c_uint32),
                ("error_rate", C.c_float), ("no_adapter_fraction", C.c_float),
                ("tpl_5p", C.c_int32), ("tpl_3p", C.c_int32)]


class DemuxOpts(C.Structure):
    """qcat_demux_opts (include/qcat_hip.h)"""
    _fields_ = [("batch_size", C.c_int32), ("kit_auto", C.c_int32), ("trim", C.c_int32), ("min_read_length", C.c_int32),
                ("tsv", C.c_int32), ("tsv_fd", C.c_int32), ("out_fd", C.c_int32), ("out_dir", C.c_char_p),
                ("kit_name", C.POINTER(C.c_char_p)), ("bc_name", C.POINTER(C.POINTER(C.c_char_p))),
                ("bc_id", C.POINTER(C.POINTER(C.c_int32))), ("bc2_id", C.POINTER(C.POINTER(C.c_int32))),
                ("filter_barcodes", C.c_int32), ("stream_reader", C.c_int32), ("segment_bytes", C.c_uint64),
                ("range_begin", C.c_uint64), ("range_end", C.c_uint64), ("input_fd", C.c_int32), ("rest_fd", C.c_int32)]


class DemuxStats(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_skipped", C.c_uint64), ("file_bytes", C.c_uint64),
                ("parse_s", C.c_double), ("scan_s", C.c_double), ("write_s", C.c_double), ("total_s", C.c_double),
                ("next_offset", C.c_uint64), ("incomplete", C.c_int32), ("segments", C.c_int32)]


class DemuxHist(C.Structure):
    """qcat_demux_hist (include/qcat_hip.h)"""
    _fields_ = [("w0", C.c_int32), ("w1", C.c_int32), ("barcode", C.POINTER(C.c_int64)), ("adapter", C.POINTER(C.c_int64)),
                ("n_none", C.c_int64), ("n_adapter_none", C.c_int64)]


class KitDescriptor(object):
    """Owns a ``qcat_kit_desc`` and every buffer it points to.

    ``layouts``: list of AdapterLayout in tie-break order; ``mode``: "epi2me" | "dual";
    ``ends``: ENDS_BOTH (detect_barcode) or ENDS_5P (scan of the 5' window only).
    Barcode ids are mapped to dense "slots" in first-seen order (ids are only ever compared for
    equality, qcat/scanner_base.py:589, and used as count keys, :680-689).
    """

    def __init__(self, layouts, qcat_config, mode="epi2me", min_quality=None, ends=ENDS_BOTH,
                 scan_middle=False, min_read_length=0, trim=False, r1_rule=None):
        if mode not in ("epi2me", "dual", "simple"):
            raise RuntimeError("Invalid demultiplexing mode: {}".format(mode))
        if len(layouts) > MAX_TEMPLATES:
            raise RuntimeError("too many adapter templates: {} > {}".format(len(layouts), MAX_TEMPLATES))
        self.layouts = list(layouts)
        self.mode = mode
        self.ends = ends
        if min_quality is None:
            min_quality = 58 if mode == "epi2me" else 60       # (dual and simple: 60)
        self.min_quality = min_quality
        self._keep = []
        self.id_slots = {}          # Barcode.id -> slot
        self.slot_ids = []
        self.kit_slots = {}
        self.kit_names = []

        tarr = (TemplateDesc * max(1, len(self.layouts)))()
        for t, lay in enumerate(self.layouts):
            td = tarr[t]
            seq = lay.get_adapter_sequences().encode("ascii")
            self._keep.append(seq)
            td.sequence = seq
            td.length = len(seq)
            td.trim_offset = int(lay.trim_offset)
            td.is_double_barcode = 1 if lay.is_double_barcode() else 0
            if lay.kit not in self.kit_slots:
                self.kit_slots[lay.kit] = len(self.kit_names)
                self.kit_names.append(lay.kit)
            td.kit_slot = self.kit_slots[lay.kit]
            for i in (0, 1):
                td.bc_end[i] = lay.get_barcode_end(i)
                td.bc_len[i] = lay.get_barcode_length(i)
                td.bc_start[i] = (lay.barcode_pos_1, lay.barcode_pos_2)[i].start
                bset = lay.get_barcode_set(i)
                sd = td.sets[i]
                if not bset:
                    sd.n = 0
                    sd.barcode_len = 0
                    continue
                # a set travels as n rows of barcode_len letters; the reference aligns every barcode with its own length
                # (scanner_base.py:112-117), and a user FASTA in simple mode may hold barcodes of unequal length: rows padded
                # to the longest, the real lengths beside them (qcat_barcode_set_desc.lengths, ABI 4).  A template's
                # placeholder has ONE length (layout.py:55-61), so the other modes refuse such a list -- ValueError: the
                # driver logs it and exits cleanly (cli.main)
                lens = [len(b.sequence) for b in bset]
                if 0 in lens:
                    raise ValueError("barcode set %d of %s holds an empty barcode" % (i + 1, getattr(lay, "kit", "the barcode list")))
                width = max(lens)
                ragged = min(lens) != width
                if ragged and mode != "simple":
                    raise ValueError("barcode set %d of %s holds barcodes of different lengths (%s): only simple mode "
                                     "(a barcode FASTA) aligns every barcode with its own length"
                                     % (i + 1, getattr(lay, "kit", "the barcode list"),
                                        ", ".join(str(n) for n in sorted(set(lens)))))
                blob = "".join(b.sequence + "-" * (width - len(b.sequence)) for b in bset).encode("latin-1", "replace")
                ids = (C.c_int32 * len(bset))()
                for j, b in enumerate(bset):
                    if b.id not in self.id_slots:
                        self.id_slots[b.id] = len(self.slot_ids)
                        self.slot_ids.append(b.id)
                    ids[j] = self.id_slots[b.id]
                self._keep += [blob, ids]
                sd.sequences = blob
                sd.ids = ids
                sd.n = len(bset)
                sd.barcode_len = width
                if ragged:
                    larr = (C.c_int32 * len(bset))(*lens)
                    self._keep.append(larr)
                    sd.lengths = larr
        self._templates = tarr

        d = KitDesc()
        d.abi_version = ABI_VERSION
        d.mode = {"epi2me": MODE_EPI2ME, "dual": MODE_DUAL, "simple": MODE_SIMPLE}[mode]
        d.ends = ends
        d.n_templates = len(self.layouts)
        d.templates = C.cast(tarr, C.POINTER(TemplateDesc))
        d.match = int(qcat_config.match)
        d.nmatch = int(qcat_config.nmatch)
        d.gap_open = int(qcat_config.gap_open)
        d.gap_extend = int(qcat_config.gap_extend)
        d.max_align_length = int(qcat_config.max_align_length)
        d.extracted_barcode_extension = int(qcat_config.extracted_barcode_extension)
        d.barcode_context_length = int(qcat_config.barcode_context_length)
        d.adapter_matrix[:] = [int(v) for v in qcat_config.matrix.table.reshape(-1)]
        d.barcode_matrix[:] = [int(v) for v in qcat_config.matrix_barcode.table.reshape(-1)]
        d.min_quality = float(min_quality)
        d.conflict_min_score = 60.0
        d.region_min_adapter_score = 90.0
        d.n_barcode_slots = len(self.slot_ids)
        d.n_kit_slots = len(self.kit_names)
        d.scan_middle_adapter = 1 if scan_middle else 0
        d.middle_min_score = 50.0
        # the driver's min-length filter of the count histogram (qcat/cli.py:521-534); 0 = count every read
        d.min_read_length = max(0, int(min_read_length))
        d.trim_reads = 1 if trim else 0
        d.r1_rule = get_r1_rule() if r1_rule is None else int(r1_rule)
        self.scan_middle = bool(scan_middle)
        self.desc = d

    @property
    def n_count_buckets(self):
        nb = len(self.slot_ids)
        # [barcode slots.., none][kit slots.., none][skipped]
        return (nb * nb if self.mode == "dual" else nb) + 1 + len(self.kit_names) + 1 + 1

    def byref(self):
        return C.byref(self.desc)


class KitInfo(C.Structure):
    """qcat_kit_info (include/qcat_hip.h): which kernels a prepared kit runs."""
    _fields_ = [("packed", C.c_int32), ("barcode_f16", C.c_int32), ("adapter_f16", C.c_int32),
                ("n_templates", C.c_int32), ("n_static_templates", C.c_int32),
                ("n_groups", C.c_int32), ("n_static_groups", C.c_int32), ("bitslice_groups", C.c_int32),
                ("bitslice_templates", C.c_int32)]


try:                                   # optional C helper for the list <-> buffer conversions (csrc/pyglue.c, built by build())
    if os.environ.get("QCAT_AMD_NO_PYGLUE"):
        raise ImportError("switched off")
    from . import _pyglue
except ImportError:                    # pure-Python fall-backs below
    _pyglue = None


def read_views(read_sequences):
    """(pointer bytes, length bytes) over the caller's str / bytes objects -- no copy -- for the `_ptrs` entry points, or
    None (no helper module, or an element that is not an ASCII str / bytes / None: pack_reads then)."""
    if _pyglue is None or type(read_sequences) is not list:
        return None
    return _pyglue.read_views(read_sequences)


def pack_reads(read_sequences):
    """list of str/bytes/None -> (uint8 bases, uint64 offsets[n+1])."""
    n = len(read_sequences)
    offsets = np.zeros(n + 1, dtype=np.uint64)
    if n and all(type(s) is str for s in read_sequences):
        # the common case (FASTQ batches): one join + one encode instead of one per read
        joined = "".join(read_sequences)
        try:
            raw = joined.encode("ascii")
        except UnicodeEncodeError:
            raw = None                                    # non-ASCII characters: per-read path below
        if raw is not None:
            np.cumsum(np.fromiter(map(len, read_sequences), dtype=np.uint64, count=n), out=offsets[1:])
            bases = np.frombuffer(raw, dtype=np.uint8) if raw else np.zeros(1, dtype=np.uint8)
            return np.ascontiguousarray(bases), offsets
    chunks = []
    total = 0
    for i, s in enumerate(read_sequences):
        if s:
            b = s.encode("latin-1", "replace") if isinstance(s, str) else bytes(s)
            chunks.append(b)
            total += len(b)
        offsets[i + 1] = total
    bases = np.frombuffer(b"".join(chunks), dtype=np.uint8) if total else np.zeros(1, dtype=np.uint8)
    return np.ascontiguousarray(bases), offsets


def _ptr(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype))


class HipLibrary(object):
    """Loaded ``libqcat_hip.so`` with typed entry points.  Raises RuntimeError when the library
    is missing or a call fails (message from ``qcat_last_error``)."""

    _instance = None

    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise RuntimeError(
                "qcat_amd: native HIP library not found at {} -- run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback)"
                .format(path))
        lib = C.CDLL(path)
        vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int32
        sig = {
            "qcat_last_error": (C.c_char_p, []),
            "qcat_abi_version": (C.c_int, []),
            "qcat_backend": (C.c_char_p, []),
            "qcat_set_option": (C.c_int, [C.c_char_p, C.c_int64]),
            "qcat_clear_option": (C.c_int, [C.c_char_p]),
            "qcat_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_int64)]),
            "qcat_reset_options": (None, []),
            "qcat_option_count": (C.c_int, []),
            "qcat_option_name": (C.c_char_p, [C.c_int]),
            "qcat_option_doc": (C.c_char_p, [C.c_int]),
            "qcat_device_count": (C.c_int, []),
            "qcat_device_numa_node": (C.c_int, [C.c_int]),
            "qcat_kit_create": (C.c_int, [C.POINTER(KitDesc), C.POINTER(vp)]),
            "qcat_kit_destroy": (None, [vp]),
            "qcat_kit_count_buckets": (C.c_int, [vp]),
            "qcat_kit_describe": (C.c_int, [vp, C.POINTER(KitInfo)]),
            "qcat_kit_attach_code": (C.c_int, [vp, C.c_char_p, C.c_uint64, C.POINTER(i32), C.POINTER(i32),
                                              C.POINTER(i32), C.POINTER(i32)]),
            "qcat_kit_attach_code_quads": (C.c_int, [vp, C.c_char_p, C.c_uint64, C.POINTER(i32), C.POINTER(i32),
                                                    C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
            "qcat_ctx_stream": (vp, [vp]),
            "qcat_ctx_graph_replays": (C.c_int64, [vp]),
            "qcat_ctx_barcode_bitslice_tiles": (C.c_int, [vp, C.POINTER(C.c_uint32)]),
            "qcat_ctx_tiny_ends": (C.c_int64, [vp]),
            "qcat_ctx_middle_wave_reads": (C.c_int64, [vp]),
            "qcat_ctx_middle_bitslice_tiles": (C.c_int, [vp, C.POINTER(C.c_uint32)]),
            "qcat_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
            "qcat_ctx_destroy": (None, [vp]),
            "qcat_scan_batch": (C.c_int, [vp, vp, vp, vp, u32, vp, vp]),
            "qcat_scan_debug": (C.c_int, [vp, vp, vp, vp, u32, vp, vp, vp, vp, u32]),
            "qcat_detect_kit": (C.c_int, [vp, vp, vp, vp, u32, vp, vp]),
            "qcat_scan_sequences": (C.c_int, [vp, vp, vp, vp, u32, vp]),
            "qcat_scan_batch_auto": (C.c_int, [vp, vp, vp, vp, u32, vp, vp, C.POINTER(i32), vp, vp]),
            "qcat_scan_batch_auto_ptrs": (C.c_int, [vp, vp, C.c_char_p, C.c_char_p, u32, vp, vp, C.POINTER(i32), vp, vp]),
            "qcat_scan_batches_auto_ptrs": (C.c_int, [vp, vp, C.c_char_p, C.c_char_p, u32, u32, vp, vp, vp]),
            "qcat_batch_upload": (C.c_int, [vp, vp, vp, u32, C.POINTER(vp)]),
            "qcat_batch_synthesize": (C.c_int, [vp, vp, C.POINTER(SynthParams), C.POINTER(vp)]),
            "qcat_batch_destroy": (None, [vp]),
            "qcat_batch_info": (C.c_int, [vp, C.POINTER(u32), C.POINTER(C.c_uint64)]),
            "qcat_batch_download": (C.c_int, [vp, vp, vp, vp]),
            "qcat_synth_read": (C.c_int64, [vp, C.POINTER(SynthParams), C.c_uint64, vp, C.c_uint64]),
            "qcat_scan_resident": (C.c_int, [vp, vp, vp]),
            "qcat_ctx_synchronize": (C.c_int, [vp]),
            "qcat_ctx_fetch_results": (C.c_int, [vp, vp, u32]),
            "qcat_ctx_fetch_counts": (C.c_int, [vp, vp, i32]),
            "qcat_ctx_counts_devptr": (vp, [vp]),
            "qcat_ctx_results_devptr": (vp, [vp]),
            "qcat_ctx_last_timing": (C.c_int, [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]),
            "qcat_ctx_set_timing": (C.c_int, [vp, C.c_int]),
            "qcat_sg_align": (C.c_int, [vp, vp, vp, vp, vp, u32, i32, i32, vp, i32, vp]),
            "qcat_fastq_open": (C.c_int, [C.c_char_p, C.POINTER(vp), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
            "qcat_fastq_close": (None, [vp]),
            "qcat_fastq_read_info": (C.c_int, [vp, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(u32), C.POINTER(C.c_uint64), C.POINTER(u32)]),
            "qcat_fastq_demux": (C.c_int, [vp, vp, vp, C.POINTER(DemuxOpts), vp, vp, C.POINTER(DemuxStats)]),
            "qcat_fastq_demux_stream": (C.c_int, [C.c_char_p, vp, vp, C.POINTER(DemuxOpts), C.POINTER(DemuxHist), C.POINTER(DemuxStats)]),
            "qcat_fastq_batch_offsets": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint64, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_uint64),
                                                   C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
            "qcat_free": (None, [vp]),
            "qcat_fastq_stream_count": (C.c_int, [C.c_char_p, C.c_uint64, C.c_uint32, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                                  C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
            "qcat_comm_unique_id": (C.c_int, [vp]),
            "qcat_comm_create": (C.c_int, [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]),
            "qcat_comm_destroy": (None, [vp]),
            "qcat_comm_info": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
            "qcat_counts_allreduce": (C.c_int, [vp, vp]),
            "qcat_comm_allreduce_f64": (C.c_int, [vp, vp, C.POINTER(C.c_double), C.c_int, C.c_int]),
            "qcat_comm_barrier": (C.c_int, [vp, vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)          # AttributeError here = ABI symbol missing
            fn.restype = res
            fn.argtypes = args
        # QCAT_HIP_LIBRARY may name another BUILD of this library (A/B runs), never another implementation of the ABI: the CPU
        # oracle behind the same entry points (oracle/libqcat_cpu.so, test infrastructure) answers "cpu-oracle"
        backend = (lib.qcat_backend() or b"").decode()
        if backend != "hip" or lib.qcat_abi_version() != ABI_VERSION:
            raise RuntimeError("qcat_amd: {} is not the HIP library of ABI {} (backend {!r}, ABI {}); there is no CPU fallback"
                               .format(path, ABI_VERSION, backend, lib.qcat_abi_version()))
        self.lib = lib
        self.path = path
        self.symbols = sorted(sig)

    @classmethod
    def get(cls):
        if cls._instance is None:
            cls._instance = cls()
        return cls._instance

    def check(self, rc):
        if rc != 0:
            msg = self.lib.qcat_last_error()
            if rc == -5:                                  # QCAT_ERR_IO: an output of qcat_fastq_demux could not be written
                raise IOError("qcat_hip: {}".format((msg or b"").decode("utf-8", "replace")))
            raise RuntimeError("qcat_hip error {}: {}".format(rc, (msg or b"").decode("utf-8", "replace")))


def set_option(name, value=1):
    """qcat_set_option: one of the library's process-wide tuning / diagnostic switches (csrc/options.h; ``name`` with or
    without the QCAT_HIP_ prefix).  ``None`` clears it."""
    hip = HipLibrary.get()
    if value is None:
        hip.check(hip.lib.qcat_clear_option(name.encode()))
    else:
        hip.check(hip.lib.qcat_set_option(name.encode(), int(value)))


def get_option(name):
    """the option's value, or None when it is not set"""
    hip = HipLibrary.get()
    v = C.c_int64()
    rc = hip.lib.qcat_get_option(name.encode(), C.byref(v))
    if rc < 0:
        hip.check(rc)
    return int(v.value) if rc == 1 else None


def reset_options():
    """back to what the environment said when the library was loaded"""
    HipLibrary.get().lib.qcat_reset_options()


def options():
    """{name: (value or None, description)} of every option"""
    lib = HipLibrary.get().lib
    return {lib.qcat_option_name(i).decode(): (get_option(lib.qcat_option_name(i).decode()), lib.qcat_option_doc(i).decode())
            for i in range(lib.qcat_option_count())}


class NativeKit(object):
    """``qcat_kit*`` handle (immutable, shareable).

    Kits outside the built-in bundle get static-letter kernels generated at run time
    (``qcat_amd.jit``): ``jit=True`` compiles before first use, ``jit=False`` never, ``jit=None``
    follows QCAT_AMD_JIT (default "auto": usable at once on the table kernels, a background compile
    swaps in a second ``qcat_kit*`` with the code attached -- ``handle`` always names the best one)."""

    def __init__(self, descriptor, jit=None):
        self.hip = HipLibrary.get()
        self.descriptor = descriptor
        self._lock = threading.Lock()
        self._handles = [self.new_handle()]            # [-1] is current; older ones stay alive for scans in flight
        self.jit_thread = None
        from . import jit as jit_mod
        how = {True: "sync", False: "off", None: jit_mod.mode()}[jit]
        if how == "off":
            return
        if jit_mod.compiler() is None:
            if jit_mod.needs_code(self.describe()):
                if jit:
                    raise RuntimeError("qcat_amd.jit: neither libhiprtc nor hipcc found; cannot generate kernels for this kit")
                jit_mod.warn_once("qcat_amd: no hipRTC / hipcc on this machine -- custom kits run the table kernels "
                                  "(about 1.5x slower than generated static-letter kernels)")
            return
        if how == "sync":
            jit_mod.attach(self)
        else:
            self.jit_thread = jit_mod.attach_in_background(self)

    def new_handle(self):
        h = C.c_void_p()
        self.hip.check(self.hip.lib.qcat_kit_create(self.descriptor.byref(), C.byref(h)))
        return h

    @property
    def handle(self):
        return self._handles[-1]

    def upgrade(self, handle):
        with self._lock:
            self._handles.append(handle)

    def wait_for_code(self, timeout=None):
        """block until a background compile (if any) has finished; returns describe()."""
        if self.jit_thread is not None:
            self.jit_thread.join(timeout)
        return self.describe()

    def describe(self, handle=None):
        """dict of qcat_kit_info: packed / fp16 eligibility and how many templates and barcode groups
        are bound to generated static-letter kernels."""
        info = KitInfo()
        self.hip.check(self.hip.lib.qcat_kit_describe(handle if handle is not None else self.handle, C.byref(info)))
        return {name: int(getattr(info, name)) for name, _ in KitInfo._fields_ }

    def __del__(self):
        if sys.is_finalizing():
            return                                   # interpreter exit: the HIP runtime may be gone already, the process's memory goes with it
        th = getattr(self, "jit_thread", None)
        if th is not None and th.is_alive():
            return                                   # the compile thread still owns handles: leak rather than race
        for h in getattr(self, "_handles", []):
            try:
                self.hip.lib.qcat_kit_destroy(h)
            except Exception:            # interpreter shutdown: the library may already be gone
                pass
        self._handles = []


COMM_ID_BYTES = 128
REDUCE_SUM, REDUCE_MAX = 0, 1


def comm_unique_id():
    """128-byte RCCL unique id (bytes): rank 0 creates it and hands it to every rank."""
    hip = HipLibrary.get()
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    hip.check(hip.lib.qcat_comm_unique_id(buf))
    return bytes(buf)


class NativeComm(object):
    """``qcat_comm*``: one rank of the RCCL communicator that all-reduces the count vector
    (SURVEY.md 8e).  Creation is collective over the ``n_ranks`` holders of ``unique_id``."""

    def __init__(self, ctx, n_ranks, rank, unique_id):
        if len(unique_id) != COMM_ID_BYTES:
            raise RuntimeError("unique id must be {} bytes".format(COMM_ID_BYTES))
        self.hip = HipLibrary.get()
        self.ctx = ctx
        self.n_ranks, self.rank = int(n_ranks), int(rank)
        h = C.c_void_p()
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        self.hip.check(self.hip.lib.qcat_comm_create(ctx.handle, self.n_ranks, self.rank, buf, C.byref(h)))
        self.handle = h

    def allreduce_counts(self):
        """in place on the context's device-resident count vector, stream ordered, no host sync."""
        self.hip.check(self.hip.lib.qcat_counts_allreduce(self.ctx.handle, self.handle))

    def allreduce(self, values, op=REDUCE_SUM):
        """a few host doubles reduced over all ranks (synchronises); returns a list."""
        arr = (C.c_double * len(values))(*[float(v) for v in values])
        self.hip.check(self.hip.lib.qcat_comm_allreduce_f64(self.ctx.handle, self.handle, arr, len(values), op))
        return list(arr)

    def barrier(self):
        self.hip.check(self.hip.lib.qcat_comm_barrier(self.ctx.handle, self.handle))

    def close(self):
        h = getattr(self, "handle", None)
        if h:
            self.hip.lib.qcat_comm_destroy(h)
            self.handle = None

    def __del__(self):
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


class NativeContext(object):
    """``qcat_ctx*`` handle: one device, one stream; not re-entrant."""

    def __init__(self, device=0):
        self.hip = HipLibrary.get()
        h = C.c_void_p()
        self.hip.check(self.hip.lib.qcat_ctx_create(int(device), C.byref(h)))
        self.handle = h
        self.device = device

    def __del__(self):
        h = getattr(self, "handle", None)
        if h and not sys.is_finalizing():            # (at interpreter exit the HIP runtime may be gone already)
            try:
                self.hip.lib.qcat_ctx_destroy(h)
            except Exception:
                pass
            self.handle = None

    def detect_kit(self, kit, bases, offsets):
        """per-template (votes, first voting read) of qcat_detect_kit."""
        n = len(offsets) - 1
        nt = len(kit.descriptor.layouts)
        votes = np.zeros(nt, dtype=np.int64)
        first = np.zeros(nt, dtype=np.int64)
        self.hip.check(self.hip.lib.qcat_detect_kit(self.handle, kit.handle, bases.ctypes.data,
                                                    offsets.ctypes.data, n, votes.ctypes.data, first.ctypes.data))
        return votes, first

    def scan_auto(self, kit, bases, offsets):
        """qcat_scan_batch_auto: (records, voted kit slot or -1); None when the kit's adapter pass cannot be
        resumed per kit (QCAT_ERR_UNSUPPORTED) -- the caller then votes and scans in two calls."""
        n = len(offsets) - 1
        out = np.zeros(n, dtype=RESULT_DTYPE)
        slot = C.c_int32(-1)
        rc = self.hip.lib.qcat_scan_batch_auto(self.handle, kit.handle, bases.ctypes.data, offsets.ctypes.data, n,
                                               out.ctypes.data, None, C.byref(slot), None, None)
        if rc == -2:
            return None
        self.hip.check(rc)
        return out, int(slot.value)

    def scan_auto_views(self, kit, views, n):
        """qcat_scan_batch_auto_ptrs over :func:`read_views` of the caller's list (the list must stay alive and unchanged
        during the call: it does, the caller holds it); returns like :meth:`scan_auto`."""
        out = np.zeros(n, dtype=RESULT_DTYPE)
        slot = C.c_int32(-1)
        rc = self.hip.lib.qcat_scan_batch_auto_ptrs(self.handle, kit.handle, views[0], views[1], n,
                                                    out.ctypes.data, None, C.byref(slot), None, None)
        if rc == -2:
            return None
        self.hip.check(rc)
        return out, int(slot.value)

    def scan_batches_auto_views(self, kit, views, n, batch_reads):
        """qcat_scan_batches_auto_ptrs: consecutive kit-auto batches of ``batch_reads`` reads in one call;
        (records, voted kit slot per batch)."""
        out = np.zeros(n, dtype=RESULT_DTYPE)
        nb = max(1, (n + batch_reads - 1) // batch_reads)
        slots = np.full(nb, -1, dtype=np.int32)
        self.hip.check(self.hip.lib.qcat_scan_batches_auto_ptrs(self.handle, kit.handle, views[0], views[1], n, batch_reads,
                                                                out.ctypes.data, None, slots.ctypes.data))
        return out, slots

    def scan_sequences(self, kit, bases, offsets):
        """scan() of whole sequences of any length (qcat_scan_sequences): one record per sequence."""
        n = len(offsets) - 1
        out = np.zeros(n, dtype=RESULT_DTYPE)
        self.hip.check(self.hip.lib.qcat_scan_sequences(self.handle, kit.handle, bases.ctypes.data,
                                                        offsets.ctypes.data, n, out.ctypes.data))
        return out

    def scan(self, kit, bases, offsets, counts=None, trace=False, rows=False, out=None):
        """qcat_scan_batch / qcat_scan_debug; ``out``: a caller-owned record array to fill (reused across
        calls it saves the page faults of a fresh 24 B x n buffer)."""
        n = len(offsets) - 1
        if out is None:
            out = np.empty(n, dtype=RESULT_DTYPE)
        elif out.dtype != RESULT_DTYPE or len(out) != n or not out.flags.c_contiguous:
            raise RuntimeError("out must be a contiguous RESULT_DTYPE array of len(offsets) - 1 records")
        cptr = counts.ctypes.data if counts is not None else None
        if not trace:
            self.hip.check(self.hip.lib.qcat_scan_batch(
                self.handle, kit.handle, bases.ctypes.data, offsets.ctypes.data, n,
                out.ctypes.data, cptr))
            return out
        ends = 1 if kit.descriptor.ends == ENDS_5P else 2
        traces = np.zeros(n * ends, dtype=TRACE_DTYPE)
        stride = 0
        bc_rows = None
        if rows:
            stride = max(len(s) for lay in kit.descriptor.layouts
                         for s in (lay.barcode_set_1 or [], lay.barcode_set_2 or []))
            bc_rows = np.full((n * ends, 2, stride), -32768, dtype=np.int16)
        self.hip.check(self.hip.lib.qcat_scan_debug(
            self.handle, kit.handle, bases.ctypes.data, offsets.ctypes.data, n,
            out.ctypes.data, cptr, traces.ctypes.data,
            bc_rows.ctypes.data if bc_rows is not None else None, stride))
        return out, traces, bc_rows


ALIGN_DTYPE = np.dtype([("score", "<i4"), ("end_query", "<i4"), ("end_ref", "<i4"), ("matches", "<i4"), ("length", "<i4")])


# which optimal path `matches` / `length` follow (include/qcat_hip.h QCAT_STATS_*; parity with parasail unpinned)
STATS_NONE, STATS_PARASAIL6, STATS_PARASAIL5, STATS_ROUND3 = 0, 1, 3, 5


def sg_align(ctx, queries, targets, gap_open, gap_extend, table, with_stats=False):
    """qcat_sg_align: semi-global alignments of queries[i] against targets[i] on the device (parasail_sg /
    parasail_sg_stat of the reference's helpers); returns an ALIGN_DTYPE array.  ``with_stats``: False, True (the
    adapter matrix's alphabet, STATS_PARASAIL6) or one of STATS_*."""
    n = len(queries)
    out = np.zeros(n, dtype=ALIGN_DTYPE)
    if n == 0:
        return out
    qb, qo = pack_reads(queries)
    tb, to = pack_reads(targets)
    t = np.ascontiguousarray(np.asarray(table, dtype=np.int8).reshape(-1))
    hip = HipLibrary.get()
    hip.check(hip.lib.qcat_sg_align(ctx.handle, qb.ctypes.data, qo.ctypes.data, tb.ctypes.data, to.ctypes.data, n,
                                    int(gap_open), int(gap_extend), t.ctypes.data,
                                    (STATS_PARASAIL6 if with_stats is True else int(with_stats or 0))
                                    | (SG_R1_SCALAR if get_r1_rule() == R1_SCALAR else 0), out.ctypes.data))
    return out


class FastqFile(object):
    """A FASTQ file mapped and split into records by the native library (qcat_fastq_open): ``n_reads``, ``n_bytes``;
    raises ``Unsupported`` when the file is not a plain four-line ASCII FASTQ file (the caller's own parser takes it)."""

    class Unsupported(RuntimeError):
        pass

    def __init__(self, path):
        self.hip = HipLibrary.get()
        h, n, nb = C.c_void_p(), C.c_uint64(), C.c_uint64()
        rc = self.hip.lib.qcat_fastq_open(os.fsencode(path), C.byref(h), C.byref(n), C.byref(nb))
        if rc == -2:
            raise FastqFile.Unsupported((self.hip.lib.qcat_last_error() or b"").decode("utf-8", "replace"))
        self.hip.check(rc)
        self.handle, self.n_reads, self.n_bytes = h, int(n.value), int(nb.value)

    def read_info(self, r):
        to, tl, so, sl = C.c_uint64(), C.c_uint32(), C.c_uint64(), C.c_uint32()
        self.hip.check(self.hip.lib.qcat_fastq_read_info(self.handle, r, C.byref(to), C.byref(tl), C.byref(so), C.byref(sl)))
        return int(to.value), int(tl.value), int(so.value), int(sl.value)

    @staticmethod
    def _demux_opts(layouts, dual, batch_size, kit_auto, trim, min_read_length, tsv_fd, out_fd, out_dir, filter_barcodes, segment_bytes, reader=0,
                    byte_range=None, input_fd=None, rest_fd=None):
        """a qcat_demux_opts and the buffers it points to"""
        n_t = len(layouts)
        keep = []
        kit_names = (C.c_char_p * n_t)(*[str(l.kit).encode() for l in layouts])
        bc_name = (C.POINTER(C.c_char_p) * n_t)()
        bc_id = (C.POINTER(C.c_int32) * n_t)()
        bc2_id = (C.POINTER(C.c_int32) * n_t)()
        for t, lay in enumerate(layouts):
            s0 = lay.get_barcode_set(0) or []
            names = (C.c_char_p * max(1, len(s0)))(*[str(b.name).encode() for b in s0])
            ids = (C.c_int32 * max(1, len(s0)))(*[int(b.id) for b in s0])
            s1 = (lay.get_barcode_set(1) or []) if dual else []
            ids2 = (C.c_int32 * max(1, len(s1)))(*[int(b.id) for b in s1])
            keep += [names, ids, ids2]
            bc_name[t] = C.cast(names, C.POINTER(C.c_char_p))
            bc_id[t] = C.cast(ids, C.POINTER(C.c_int32))
            bc2_id[t] = C.cast(ids2, C.POINTER(C.c_int32))
        keep += [kit_names, bc_name, bc_id, bc2_id]
        o = DemuxOpts(batch_size=batch_size, kit_auto=1 if kit_auto else 0, trim=1 if trim else 0,
                      min_read_length=int(min_read_length), tsv=1 if tsv_fd is not None else 0,
                      tsv_fd=-1 if tsv_fd is None else tsv_fd, out_fd=-1 if out_fd is None else out_fd,
                      out_dir=os.fsencode(out_dir) if out_dir else None,
                      kit_name=C.cast(kit_names, C.POINTER(C.c_char_p)), bc_name=C.cast(bc_name, C.POINTER(C.POINTER(C.c_char_p))),
                      bc_id=C.cast(bc_id, C.POINTER(C.POINTER(C.c_int32))), bc2_id=C.cast(bc2_id, C.POINTER(C.POINTER(C.c_int32))),
                      filter_barcodes=1 if filter_barcodes else 0, stream_reader=int(reader), segment_bytes=int(segment_bytes or 0),
                      range_begin=int(byte_range[0]) if byte_range else 0, range_end=int(byte_range[1]) if byte_range else 0,
                      input_fd=0 if input_fd is None else int(input_fd) + 1, rest_fd=0 if rest_fd is None else int(rest_fd) + 1)
        return o, keep

    def demux(self, ctx, kit, layouts, dual, batch_size=4000, kit_auto=False, trim=False, min_read_length=0,
              tsv_fd=None, out_fd=None, out_dir=None, filter_barcodes=False):
        """qcat_fastq_demux: scan + write; returns (records, skipped flags, stats dict).  ``layouts``: the AdapterLayout
        list of ``kit`` (the names the writers print)."""
        o, _keep = self._demux_opts(layouts, dual, batch_size, kit_auto, trim, min_read_length, tsv_fd, out_fd, out_dir, filter_barcodes, 0)
        recs = np.zeros(self.n_reads, dtype=RESULT_DTYPE)
        skipped = np.zeros(self.n_reads, dtype=np.uint8)
        st = DemuxStats()
        rc = self.hip.lib.qcat_fastq_demux(self.handle, ctx.handle, kit.handle, C.byref(o), recs.ctypes.data, skipped.ctypes.data, C.byref(st))
        if rc == -2:
            raise FastqFile.Unsupported((self.hip.lib.qcat_last_error() or b"").decode("utf-8", "replace"))
        self.hip.check(rc)
        return recs, skipped, {"n_reads": int(st.n_reads), "n_skipped": int(st.n_skipped), "file_bytes": int(st.file_bytes),
                               "parse_s": st.parse_s, "scan_s": st.scan_s, "write_s": st.write_s, "total_s": st.total_s}

    @staticmethod
    def batch_offsets(path, batch_size=4000, segment_bytes=0):
        """qcat_fastq_batch_offsets: (offsets, n_reads, next_offset) -- offsets[i] = file offset of read i * batch_size (one per
        batch of the driver's loop, numpy uint64) and offsets[-1] = where the plain records end (no device needed)."""
        hip = HipLibrary.get()
        ptr = C.POINTER(C.c_uint64)()
        nb, nr, nxt = C.c_uint64(), C.c_uint64(), C.c_uint64()
        rc = hip.lib.qcat_fastq_batch_offsets(os.fsencode(path), int(batch_size), int(segment_bytes), C.byref(ptr), C.byref(nb), C.byref(nr), C.byref(nxt))
        if rc == -2:
            raise FastqFile.Unsupported((hip.lib.qcat_last_error() or b"").decode("utf-8", "replace"))
        hip.check(rc)
        try:
            offs = np.ctypeslib.as_array(ptr, shape=(int(nb.value) + 1,)).copy()
        finally:
            hip.lib.qcat_free(ptr)
        return offs, int(nr.value), int(nxt.value)

    @staticmethod
    def stream_count(path, segment_bytes=0, batch_size=0, reader=0):
        """qcat_fastq_stream_count: (reads, sequence letters, next offset, segments) of a file through the reader stage of
        qcat_fastq_demux_stream (no device needed)."""
        hip = HipLibrary.get()
        n, nb, off, segs = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint32()
        rc = hip.lib.qcat_fastq_stream_count(os.fsencode(path), int(segment_bytes), int(batch_size), int(reader), C.byref(n), C.byref(nb), C.byref(off), C.byref(segs))
        if rc == -2:
            raise FastqFile.Unsupported((hip.lib.qcat_last_error() or b"").decode("utf-8", "replace"))
        hip.check(rc)
        return int(n.value), int(nb.value), int(off.value), int(segs.value)

    @staticmethod
    def demux_stream(path, ctx, kit, layouts, dual, batch_size=4000, kit_auto=False, trim=False, min_read_length=0,
                     tsv_fd=None, out_fd=None, out_dir=None, filter_barcodes=False, segment_bytes=0, reader=0, byte_range=None,
                     input_fd=None, rest_fd=None):
        """qcat_fastq_demux_stream: the file in segments through read | scan | write, host memory independent of its size.
        ``byte_range`` = (begin, end): a rank's shard of the file, both record starts (``batch_offsets``; end 0: the file's end).
        ``input_fd`` (``path`` None): the input as a descriptor -- the driver's stdin; a pipe is read in order, and when the loop
        ends early (``stats["incomplete"]``) the bytes it had read and not handled are in ``rest_fd`` (a file of the caller's).
        Returns (barcode counts [template][barcode][second barcode], adapter counts [template], reads without a barcode,
        reads without an adapter, stats dict); ``stats["incomplete"]``: the loop ended at ``stats["next_offset"]`` in front of
        a record that is not plain -- the caller's own parser carries on from there.  Raises ``Unsupported`` when that is the
        case for the very first segment (nothing has been written)."""
        hip = HipLibrary.get()
        o, _keep = FastqFile._demux_opts(layouts, dual, batch_size, kit_auto, trim, min_read_length, tsv_fd, out_fd, out_dir,
                                         filter_barcodes, segment_bytes, reader, byte_range, input_fd, rest_fd)
        n_t = len(layouts)
        w0 = max(1, max(len(l.get_barcode_set(0) or ()) for l in layouts))
        w1 = max(1, max(len(l.get_barcode_set(1) or ()) for l in layouts)) if dual else 1
        barcode = np.zeros((n_t, w0, w1), dtype=np.int64)
        adapter = np.zeros(n_t, dtype=np.int64)
        h = DemuxHist(w0=w0, w1=w1, barcode=barcode.ctypes.data_as(C.POINTER(C.c_int64)), adapter=adapter.ctypes.data_as(C.POINTER(C.c_int64)))
        st = DemuxStats()
        rc = hip.lib.qcat_fastq_demux_stream(os.fsencode(path) if path is not None else None, ctx.handle, kit.handle, C.byref(o), C.byref(h), C.byref(st))
        if rc == -2 and int(st.segments) == 0 and int(st.n_reads) == 0:
            # "not for the native loop" is only an answer while NOTHING has been written: the caller then parses the file
            # itself from offset 0.  An UNSUPPORTED behind written segments (a scan call refused later in the file) is an
            # error like any other -- redoing the file would duplicate the rows already out
            raise FastqFile.Unsupported((hip.lib.qcat_last_error() or b"").decode("utf-8", "replace"))
        hip.check(rc)
        return barcode, adapter, int(h.n_none), int(h.n_adapter_none), {
            "n_reads": int(st.n_reads), "n_skipped": int(st.n_skipped), "file_bytes": int(st.file_bytes), "parse_s": st.parse_s,
            "scan_s": st.scan_s, "write_s": st.write_s, "total_s": st.total_s, "next_offset": int(st.next_offset),
            "incomplete": int(st.incomplete), "segments": int(st.segments)}

    def close(self):
        if self.handle:
            self.hip.lib.qcat_fastq_close(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:                     # noqa: BLE001 -- interpreter shutdown
            pass
