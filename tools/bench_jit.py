#!/usr/bin/env python3
"""Table kernels vs run-time generated static-letter kernels on a CUSTOM kit (a 96-barcode, two-template
kit that is not in the built-in bundle): resident-batch scan rate with and without qcat_amd.jit."""
import ctypes as C
import os
import random
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yaml                                      # noqa: E402
from qcat_amd import native, scanner              # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
tmp = tempfile.mkdtemp(prefix="qcat_jit_bench_")
rng = random.Random(1)
bcs = ["".join(rng.choice("ACGT") for _ in range(24)) for _ in range(96)]
for name, seq in (("X_5p", "GGTGCTGTA" + "N" * 24 + "TTAACCTTTCTGTTGGTGCTGATATTGCGT"), ("X_3p", "GGTGCTGTA" + "N" * 24 + "TTAACCTACTTGCCTGTCGCTCTATCTTCAC")):
    rows = [{"name": "barcode%02d" % (i + 1), "id": i + 1, "sequence": s, "fwd_strand": True} for i, s in enumerate(bcs)]
    with open(os.path.join(tmp, name + ".yml"), "w") as fh:
        yaml.safe_dump({"kit": "CUSTOM96", "auto_detect": False, "description": "bench", "sequence": seq, "trim_offset": 0,
                        "barcode_set_1": rows, "barcode_set_2": []}, fh)
det = scanner.factory(kit="CUSTOM96", kit_folder=tmp)
hip = native.HipLibrary.get(); lib = hip.lib
ctx = native.NativeContext(0)
for use_jit in (False, True):
    t0 = time.perf_counter()
    kit = native.NativeKit(det.descriptor(), jit=use_jit)
    t_kit = time.perf_counter() - t0
    sp = native.SynthParams(seed=5, n_reads=n, insert_len=600, lead_min=5, lead_max=40, error_rate=0.08,
                            no_adapter_fraction=0.05, tpl_5p=0, tpl_3p=1)
    b = C.c_void_p(); hip.check(lib.qcat_batch_synthesize(ctx.handle, kit.handle, C.byref(sp), C.byref(b)))
    for _ in range(2):
        hip.check(lib.qcat_scan_resident(ctx.handle, kit.handle, b))
    hip.check(lib.qcat_ctx_synchronize(ctx.handle))
    t0 = time.perf_counter()
    for _ in range(5):
        hip.check(lib.qcat_scan_resident(ctx.handle, kit.handle, b))
    hip.check(lib.qcat_ctx_synchronize(ctx.handle))
    dt = (time.perf_counter() - t0) / 5
    print("jit=%s kit creation %.2f s; %s; %d reads/step: %.2f ms = %.1f M reads/s" %
          (use_jit, t_kit, kit.describe(), n, dt * 1e3, n / dt / 1e6))
    lib.qcat_batch_destroy(b)
