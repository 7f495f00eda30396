#!/usr/bin/env python3
"""Histogram of the barcode region lengths of a synthetic batch (dev tool): which lengths the jobs outside the two hot
classes have.  usage: python tools/region_len_hist.py [kit, default PBC096] [reads, default 200000]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from qcat_amd import native, scanner  # noqa: E402

kitname = sys.argv[1] if len(sys.argv) > 1 else "PBC096"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
mode = "dual" if kitname == "DUAL" else "epi2me"
det = scanner.factory(mode=mode, kit=None if kitname == "DUAL" else kitname)
ends = native.ENDS_5P if kitname.startswith("NBD") else native.ENDS_BOTH
desc = det.descriptor(ends=ends)
hip = native.HipLibrary.get(); lib = hip.lib
kit = native.NativeKit(desc); ctx = native.NativeContext(0)
t5 = len(det.layouts) - 1 if kitname != "DUAL" else 1
sp = native.SynthParams(seed=5, n_reads=n, insert_len=600, lead_min=5, lead_max=40, error_rate=0.08, no_adapter_fraction=0.05,
                        tpl_5p=t5 if kitname != "PBC096" else 1, tpl_3p=0 if ends == native.ENDS_BOTH else -1)
b = C.c_void_p()
hip.check(lib.qcat_batch_synthesize(ctx.handle, kit.handle, C.byref(sp), C.byref(b)))
nb, nr = C.c_uint64(), C.c_uint32()
hip.check(lib.qcat_batch_info(b, C.byref(nr), C.byref(nb)))
bases = np.zeros(nb.value, dtype=np.uint8); offs = np.zeros(n + 1, dtype=np.uint64)
hip.check(lib.qcat_batch_download(ctx.handle, b, bases.ctypes.data, offs.ctypes.data))
recs, traces, _rows = ctx.scan(kit, bases, offs, trace=True)
rl = np.asarray(traces["region_len"]).reshape(-1)
rl = rl[rl > 0]
h = np.bincount(rl, minlength=151)
tot = h.sum()
print(kitname, "jobs", tot)
for L in np.argsort(-h)[:25]:
    if h[L]:
        print("  L=%3d  %8d  %.3f %%" % (L, h[L], 100.0 * h[L] / tot))
