#!/usr/bin/env python3
"""How many records change between the two end-position rules (include/qcat_hip.h QCAT_R1_STRIPED / QCAT_R1_SCALAR)?
BASELINE config 3 (10 M synthetic PBC096 reads, 8 % errors, 5 % adapter-free) and config 2 / the dual kit at 1 M reads, on the
device: python tools/r1_rule_diff.py [reads of config 3, default 10000000]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from qcat_amd import config, native, scanner  # noqa: E402

hip = native.HipLibrary.get()
lib = hip.lib
ctx = native.NativeContext(0)
cfg = config.qcatConfig()


def run(label, mode, kit_name, ends, n, seed, t5, t3):
    det = scanner.factory(mode=mode, kit=kit_name)
    out = {}
    batch = None
    for rule in ("striped", "scalar"):
        native.set_r1_rule(rule)
        kit = native.NativeKit(det.descriptor(qcat_config=cfg, ends=ends))
        if batch is None:
            sp = native.SynthParams(seed=seed, n_reads=n, insert_len=600, lead_min=5, lead_max=40, error_rate=0.08,
                                    no_adapter_fraction=0.05, tpl_5p=t5, tpl_3p=t3)
            batch = C.c_void_p()
            hip.check(lib.qcat_batch_synthesize(ctx.handle, kit.handle, C.byref(sp), C.byref(batch)))
        hip.check(lib.qcat_scan_resident(ctx.handle, kit.handle, batch))
        recs = np.zeros(n, dtype=native.RESULT_DTYPE)
        hip.check(lib.qcat_ctx_fetch_results(ctx.handle, recs.ctypes.data, n))
        out[rule] = recs
    native.set_r1_rule("striped")
    lib.qcat_batch_destroy(batch)
    a, b = out["striped"], out["scalar"]
    differ = a != b
    res = {"workload": label, "reads": n, "records_that_differ": int(differ.sum()), "fraction": float(differ.mean())}
    for f in native.RESULT_DTYPE.names:
        res["differ_in_" + f] = int((a[f] != b[f]).sum())
    res["calls_that_change_barcode_or_status"] = int(((a["barcode_idx"] != b["barcode_idx"]) | (a["exit_status"] != b["exit_status"])).sum())
    print(json.dumps(res))
    return res


n3 = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
run("config3 (PBC096, 5'+3')", "epi2me", "PBC096", native.ENDS_BOTH, n3, 20260930, 1, 0)
run("config2 (NBD104, 5' only)", "epi2me", "NBD103/NBD104", native.ENDS_5P, 1000000, 20260929, 1, 0)
run("dual kit", "dual", None, native.ENDS_BOTH, 1000000, 20260932, 1, 0)
