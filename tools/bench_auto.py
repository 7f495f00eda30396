#!/usr/bin/env python3
"""Throughput of the kit-auto batch path (detect_kit vote over the 12 auto-detect templates, then
the scan with the voted kit) through the host-buffer API, for one batch of synthetic PBC096 reads."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from qcat_amd import native, scanner
import ctypes as C

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
det = scanner.factory()                                   # kit auto
hip = native.HipLibrary.get(); lib = hip.lib
kit_all = native.NativeKit(det.descriptor())
ctx = native.NativeContext(0)
sp = native.SynthParams(seed=5, n_reads=n, insert_len=600, lead_min=5, lead_max=40, error_rate=0.08,
                        no_adapter_fraction=0.05, tpl_5p=3, tpl_3p=2)
b = C.c_void_p(); hip.check(lib.qcat_batch_synthesize(ctx.handle, kit_all.handle, C.byref(sp), C.byref(b)))
nb = C.c_uint64(); nr = C.c_uint32(); hip.check(lib.qcat_batch_info(b, C.byref(nr), C.byref(nb)))
bases = np.zeros(nb.value, dtype=np.uint8); offs = np.zeros(n + 1, dtype=np.uint64)
hip.check(lib.qcat_batch_download(ctx.handle, b, bases.ctypes.data, offs.ctypes.data))
for rep in range(2):
    t0 = time.perf_counter(); votes, first = ctx.detect_kit(kit_all, bases, offs); t1 = time.perf_counter()
    names = {}
    for t, lay in enumerate(det.layouts): names[lay.kit] = names.get(lay.kit, 0) + int(votes[t])
    best = max(names, key=names.get)
    sub = native.NativeKit(det.descriptor(layouts=det.get_adapters(best)))
    t2 = time.perf_counter(); recs = ctx.scan(sub, bases, offs); t3 = time.perf_counter()
    print("n=%d vote: %.1f ms (%.1f M reads/s) -> %s ; scan: %.1f ms ; called %.1f %%" %
          (n, (t1 - t0) * 1e3, n / (t1 - t0) / 1e6, best, (t3 - t2) * 1e3, 100.0 * (recs["barcode_idx"] >= 0).mean()))
