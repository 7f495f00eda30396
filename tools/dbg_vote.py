import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", ".")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests"))
import numpy as np, ctypes as C
import test_batch_auto_gpu as tb
from qcat_amd import scanner, config, native
det = scanner.factory(); cfg = config.qcatConfig()
reads = tb._mixed_batch(det, "PBC096", 3000, 77)
kit = det._native_kit(det.layouts, cfg, native.ENDS_BOTH)
bases, offsets = native.pack_reads(reads)
lib = native.HipLibrary.get().lib
for rep in range(3):
    ctx = native.NativeContext(0)
    v = np.zeros(16, dtype=np.int64); f = np.zeros(16, dtype=np.int64)
    rc = lib.qcat_detect_kit(ctx.handle, kit.handle, bases.ctypes.data, offsets.ctypes.data, len(reads), v.ctypes.data, f.ctypes.data)
    print("detect_kit rc", rc, v[:12].tolist())
    v2 = np.zeros(16, dtype=np.int64); f2 = np.zeros(16, dtype=np.int64); slot = C.c_int32(-9)
    out = np.zeros(len(reads), dtype=native.RESULT_DTYPE)
    rc = lib.qcat_scan_batch_auto(ctx.handle, kit.handle, bases.ctypes.data, offsets.ctypes.data, len(reads), out.ctypes.data, None, C.byref(slot), v2.ctypes.data, f2.ctypes.data)
    print("scan_auto  rc", rc, v2[:12].tolist(), "slot", slot.value, kit.descriptor.kit_names)
    v3 = np.zeros(16, dtype=np.int64); f3 = np.zeros(16, dtype=np.int64)
    rc = lib.qcat_detect_kit(ctx.handle, kit.handle, bases.ctypes.data, offsets.ctypes.data, len(reads), v3.ctypes.data, f3.ctypes.data)
    print("detect_kit again", v3[:12].tolist())
print("python detect_kit:", det.detect_kit(reads, cfg)[0])
