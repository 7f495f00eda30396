"""ctypes wrapper of the CPU oracle (oracle/libqcat_oracle.so) for the test-suite, smoke()
and bench.py's cpu_baseline leg.  Never imported by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

from qcat_amd import native

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "libqcat_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_ROOT, "oracle", "qcat_oracle.c")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")],
                                  stdout=subprocess.DEVNULL)
        l = C.CDLL(_SO)
        l.qo_last_error.restype = C.c_char_p
        vp, u32 = C.c_void_p, C.c_uint32
        l.qo_scan_debug.argtypes = [C.POINTER(native.KitDesc), vp, vp, u32, vp, vp, vp, vp, u32, C.c_int]
        l.qo_scan_batch.argtypes = [C.POINTER(native.KitDesc), vp, vp, u32, vp, vp, C.c_int]
        l.qo_detect_kit_votes.argtypes = [C.POINTER(native.KitDesc), vp, vp, u32, vp, vp]
        l.qo_sg.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, vp,
                            C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        l.qo_scan_sequences.argtypes = [C.POINTER(native.KitDesc), vp, vp, u32, vp]
        l.qo_count_buckets.argtypes = [C.POINTER(native.KitDesc)]
        _lib = l
    return _lib


def _check(rc):
    if rc != 0:
        raise RuntimeError("oracle error {}: {}".format(rc, lib().qo_last_error().decode()))


def sg(s1, s2, open_, extend, table, rule=native.R1_STRIPED):
    """(score, end_query, end_ref) of the oracle DP; ``table`` = int8 7x7 [target, query]; ``rule``: native.R1_STRIPED /
    native.R1_SCALAR (include/qcat_hip.h QCAT_R1_*: which of the reference's two routines places the end)."""
    t = np.ascontiguousarray(table, dtype=np.int8)
    sc, eq, er = C.c_int32(), C.c_int32(), C.c_int32()
    b1, b2 = s1.encode("latin-1", "replace"), s2.encode("latin-1", "replace")
    fn = lib().qo_sg_rule
    fn.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                   C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    _check(fn(b1, len(b1), b2, len(b2), open_, extend, t.ctypes.data, int(rule), C.byref(sc), C.byref(eq), C.byref(er)))
    return sc.value, eq.value, er.value


class _Stats(C.Structure):
    _fields_ = [("score", C.c_int32), ("end_query", C.c_int32), ("end_ref", C.c_int32), ("matches", C.c_int32), ("length", C.c_int32)]


def sg_stats(s1, s2, open_, extend, table, rule=native.STATS_PARASAIL6):
    """(score, end_query, end_ref, matches, length) of the oracle DP with statistics (qo_sg_stats_rule; `rule` is one of
    native.STATS_*, include/qcat_hip.h QCAT_STATS_*)."""
    t = np.ascontiguousarray(table, dtype=np.int8)
    st = _Stats()
    b1, b2 = s1.encode("latin-1", "replace"), s2.encode("latin-1", "replace")
    fn = lib().qo_sg_stats_rule
    fn.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(_Stats)]
    _check(fn(b1, len(b1), b2, len(b2), open_, extend, t.ctypes.data, int(rule), C.byref(st)))
    return st.score, st.end_query, st.end_ref, st.matches, st.length


def scan(descriptor, reads=None, packed=None, counts=False, trace=False, rows=False, threads=1):
    """Run the oracle over a batch.  ``descriptor`` is a native.KitDescriptor."""
    bases, offsets = packed if packed is not None else native.pack_reads(reads)
    n = len(offsets) - 1
    out = np.zeros(n, dtype=native.RESULT_DTYPE)
    cnt = np.zeros(descriptor.n_count_buckets, dtype=np.int64) if counts else None
    ends = 1 if descriptor.ends == native.ENDS_5P else 2
    traces = np.zeros(n * ends, dtype=native.TRACE_DTYPE) if trace else None
    stride, bc_rows = 0, None
    if rows:
        stride = max(len(s) for lay in descriptor.layouts
                     for s in (lay.barcode_set_1 or [], lay.barcode_set_2 or []))
        bc_rows = np.full((n * ends, 2, stride), -32768, dtype=np.int16)
    _check(lib().qo_scan_debug(descriptor.byref(), bases.ctypes.data, offsets.ctypes.data, n,
                               out.ctypes.data, cnt.ctypes.data if counts else None,
                               traces.ctypes.data if trace else None,
                               bc_rows.ctypes.data if rows else None, stride, threads))
    res = [out]
    if counts:
        res.append(cnt)
    if trace:
        res.append(traces)
    if rows:
        res.append(bc_rows)
    return res[0] if len(res) == 1 else tuple(res)


def detect_kit_votes(descriptor, reads):
    bases, offsets = native.pack_reads(reads)
    n = len(offsets) - 1
    votes = np.zeros(len(descriptor.layouts) + 1, dtype=np.int64)
    per_read = np.zeros(n, dtype=np.int32)
    _check(lib().qo_detect_kit_votes(descriptor.byref(), bases.ctypes.data, offsets.ctypes.data, n,
                                     votes.ctypes.data, per_read.ctypes.data))
    return votes, per_read


def scan_sequences(descriptor, seqs):
    """scan() of whole sequences of any length; one record per sequence."""
    bases, offsets = native.pack_reads(seqs)
    n = len(offsets) - 1
    out = np.zeros(n, dtype=native.RESULT_DTYPE)
    _check(lib().qo_scan_sequences(descriptor.byref(), bases.ctypes.data, offsets.ctypes.data, n, out.ctypes.data))
    return out
