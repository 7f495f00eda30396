"""The synthetic-read generator: Python twin == host C twin (CPU) == device kernel (GPU)."""
import ctypes as C

import numpy as np
import pytest

import synth
from qcat_amd import native, scanner


def _params(seed, n, t5, t3, e):
    return native.SynthParams(seed=seed, n_reads=n, insert_len=600, lead_min=5, lead_max=40,
                              error_rate=e, no_adapter_fraction=0.05, tpl_5p=t5, tpl_3p=t3)


def _host_read(hip, kit, p, i):
    buf = np.zeros(4096, dtype=np.uint8)
    n = hip.lib.qcat_synth_read(kit.handle, C.byref(p), i, buf.ctypes.data, buf.size)
    assert 0 < n <= buf.size
    return buf[:n].tobytes().decode()


@pytest.mark.parametrize("mode,kit,e", [("epi2me", "PBC096", 0.0), ("epi2me", "PBC096", 0.08),
                                        ("dual", None, 0.08), ("epi2me", "NBD103/NBD104", 0.15)])
def test_python_twin_matches_host_c(mode, kit, e):
    det = scanner.factory(mode=mode, kit=kit)
    nk = native.NativeKit(det.descriptor())
    hip = native.HipLibrary.get()
    p = _params(4242, 40, 1, 0, e)
    for i in range(40):
        want = synth.synth_read(i, 4242, det.layouts, 1, 0, error_rate=e)
        assert _host_read(hip, nk, p, i) == want


@pytest.mark.gpu
def test_device_generator_matches_host():
    det = scanner.factory(kit="PBC096")
    nk = native.NativeKit(det.descriptor())
    hip = native.HipLibrary.get()
    ctx = native.NativeContext(0)
    p = _params(777, 3000, 1, 0, 0.08)
    b = C.c_void_p()
    hip.check(hip.lib.qcat_batch_synthesize(ctx.handle, nk.handle, C.byref(p), C.byref(b)))
    n, nb = C.c_uint32(), C.c_uint64()
    hip.check(hip.lib.qcat_batch_info(b, C.byref(n), C.byref(nb)))
    assert n.value == 3000
    bases = np.zeros(nb.value, dtype=np.uint8)
    offsets = np.zeros(n.value + 1, dtype=np.uint64)
    hip.check(hip.lib.qcat_batch_download(ctx.handle, b, bases.ctypes.data, offsets.ctypes.data))
    hip.lib.qcat_batch_destroy(b)
    for i in (0, 1, 2, 17, 999, 2999):
        got = bases[int(offsets[i]):int(offsets[i + 1])].tobytes().decode()
        assert got == _host_read(hip, nk, p, i)
