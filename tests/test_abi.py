"""CPU-only: the C-ABI library loads and exports every symbol include/qcat_hip.h declares, the
descriptor structs have the documented sizes, and the product fails loudly without a device."""
import ctypes as C
import os
import re

import pytest

from qcat_amd import native, scanner

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "qcat_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qcat_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    hip = native.HipLibrary.get()
    declared = _declared_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(hip.lib, name), name
    assert sorted(hip.symbols) == declared          # the Python binding covers the whole header
    assert hip.lib.qcat_abi_version() == native.ABI_VERSION


def test_struct_sizes():
    assert C.sizeof(native.Result) == 24
    assert native.TRACE_DTYPE.itemsize == 4 * (1 + 16 + 16 + 5 + 2 + 2 + 2 + 2 + 1)


def test_kit_create_validates_without_a_device():
    hip = native.HipLibrary.get()
    det = scanner.factory(kit="PBC096")
    kit = native.NativeKit(det.descriptor())         # host-side preparation only
    assert hip.lib.qcat_kit_count_buckets(kit.handle) == det.descriptor().n_count_buckets == 96 + 1 + 1 + 1 + 1
    bad = det.descriptor()
    bad.desc.abi_version = 99
    with pytest.raises(RuntimeError, match="ABI version"):
        native.NativeKit(bad)
    bad = det.descriptor()
    bad.desc.max_align_length = 100000
    with pytest.raises(RuntimeError, match="max_align_length"):
        native.NativeKit(bad)


def test_no_silent_cpu_fallback():
    """Without a GPU the scanners must raise, never compute on the host."""
    hip = native.HipLibrary.get()
    if hip.lib.qcat_device_count() > 0:
        pytest.skip("GPU present")
    det = scanner.factory(kit="PBC096")
    with pytest.raises(RuntimeError, match="no HIP device"):
        det.detect_barcode("ACGT" * 100)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "qcat_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".inc", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle_lib" not in text and "libqcat_oracle" not in text and "qo_" not in text, f
