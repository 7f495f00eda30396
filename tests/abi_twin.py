"""One ctypes driver for the host-buffer subset of the C ABI (include/qcat_hip.h) that works on ANY
library exporting it: libqcat_hip.so (the product) and oracle/libqcat_cpu.so (the oracle behind the
same entry points -- SURVEY.md 8b "the same ABI is implemented twice").  Test infrastructure."""
import ctypes as C
import os
import subprocess

import numpy as np

from qcat_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP = native.LIB_PATH
CPU = os.path.join(ROOT, "oracle", "libqcat_cpu.so")
COMMON = ["qcat_last_error", "qcat_abi_version", "qcat_device_count", "qcat_kit_create", "qcat_kit_destroy",
          "qcat_kit_count_buckets", "qcat_ctx_create", "qcat_ctx_destroy", "qcat_scan_batch", "qcat_scan_debug",
          "qcat_scan_sequences", "qcat_detect_kit"]


class Abi(object):
    def __init__(self, path):
        if path == CPU:
            src = [os.path.join(ROOT, "oracle", f) for f in ("qcat_cpu_abi.c", "qcat_oracle.c")]
            if not os.path.exists(CPU) or any(os.path.getmtime(CPU) < os.path.getmtime(s) for s in src):
                subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
        lib = C.CDLL(path)
        vp, u32 = C.c_void_p, C.c_uint32
        lib.qcat_last_error.restype = C.c_char_p
        lib.qcat_kit_create.argtypes = [C.POINTER(native.KitDesc), C.POINTER(vp)]
        lib.qcat_kit_destroy.argtypes = [vp]
        lib.qcat_kit_destroy.restype = None
        lib.qcat_kit_count_buckets.argtypes = [vp]
        lib.qcat_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
        lib.qcat_ctx_destroy.argtypes = [vp]
        lib.qcat_ctx_destroy.restype = None
        lib.qcat_scan_batch.argtypes = [vp, vp, vp, vp, u32, vp, vp]
        lib.qcat_scan_debug.argtypes = [vp, vp, vp, vp, u32, vp, vp, vp, vp, u32]
        lib.qcat_scan_sequences.argtypes = [vp, vp, vp, vp, u32, vp]
        lib.qcat_detect_kit.argtypes = [vp, vp, vp, vp, u32, vp, vp]
        self.lib, self.path = lib, path

    def check(self, rc):
        if rc:
            raise RuntimeError("%s: error %d: %s" % (os.path.basename(self.path), rc, (self.lib.qcat_last_error() or b"").decode()))

    def run(self, descriptor, reads, sequences=None, votes=False):
        """everything the common entry points return for one kit and one batch, as bytes / arrays"""
        lib = self.lib
        kit, ctx = C.c_void_p(), C.c_void_p()
        self.check(lib.qcat_kit_create(descriptor.byref(), C.byref(kit)))
        self.check(lib.qcat_ctx_create(0, C.byref(ctx)))
        try:
            out = {"buckets": lib.qcat_kit_count_buckets(kit)}
            bases, offsets = native.pack_reads(reads)
            n = len(reads)
            recs = np.zeros(n, dtype=native.RESULT_DTYPE)
            cnt = np.zeros(out["buckets"], dtype=np.int64)
            self.check(lib.qcat_scan_batch(ctx, kit, bases.ctypes.data, offsets.ctypes.data, n, recs.ctypes.data, cnt.ctypes.data))
            out["records"], out["counts"] = recs.tobytes(), cnt
            ends = 1 if descriptor.ends == native.ENDS_5P else 2
            stride = max(len(s) for lay in descriptor.layouts for s in (lay.barcode_set_1 or [], lay.barcode_set_2 or []))
            recs2 = np.zeros(n, dtype=native.RESULT_DTYPE)
            traces = np.zeros(n * ends, dtype=native.TRACE_DTYPE)
            rows = np.full((n * ends, 2, stride), -32768, dtype=np.int16)
            self.check(lib.qcat_scan_debug(ctx, kit, bases.ctypes.data, offsets.ctypes.data, n, recs2.ctypes.data, None,
                                           traces.ctypes.data, rows.ctypes.data, stride))
            out["debug_records"], out["traces"], out["rows"] = recs2.tobytes(), traces, rows
            if sequences is not None:
                b2, o2 = native.pack_reads(sequences)
                srec = np.zeros(len(sequences), dtype=native.RESULT_DTYPE)
                self.check(lib.qcat_scan_sequences(ctx, kit, b2.ctypes.data, o2.ctypes.data, len(sequences), srec.ctypes.data))
                out["sequences"] = srec.tobytes()
            if votes:
                nt = len(descriptor.layouts)
                v, f = np.zeros(nt, dtype=np.int64), np.zeros(nt, dtype=np.int64)
                self.check(lib.qcat_detect_kit(ctx, kit, bases.ctypes.data, offsets.ctypes.data, n, v.ctypes.data, f.ctypes.data))
                out["votes"], out["first"] = v, f
            return out
        finally:
            lib.qcat_ctx_destroy(ctx)
            lib.qcat_kit_destroy(kit)
