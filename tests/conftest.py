import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present():
    try:
        from qcat_amd import native
        return native.HipLibrary.get().lib.qcat_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU should fail loudly, not skip: only auto-skip when the
    # marker expression was not given explicitly.
    if config.getoption("-m"):
        return
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# ---- the library's switches are options behind the C ABI, not environment variables (csrc/options.h, round 5) -----------------
# The library reads QCAT_HIP_<NAME> once, when it is loaded; afterwards a switch only changes through qcat_set_option /
# qcat_clear_option.  The tests flip switches with monkeypatch.setenv("QCAT_HIP_<NAME>", value) / delenv -- written when the
# library still called getenv() on every scan -- so MonkeyPatch forwards exactly those names to the option calls and puts the
# previous values back in undo() (every MonkeyPatch object: the fixture's and the ones of monkeypatch.context()).
def _forward_switches_to_the_option_abi():
    from _pytest.monkeypatch import MonkeyPatch
    if getattr(MonkeyPatch, "_qcat_forwarding", False):
        return
    orig_setenv, orig_delenv, orig_undo = MonkeyPatch.setenv, MonkeyPatch.delenv, MonkeyPatch.undo

    def option_of(name):
        if not name.startswith("QCAT_HIP_"):
            return None
        try:
            from qcat_amd import native
            lib = native.HipLibrary.get().lib
        except Exception:
            return None
        return name[9:] if lib.qcat_get_option(name.encode(), None) >= 0 else None      # (QCAT_HIP_LIBRARY, _RCCL_LIB: not options)

    def remember(self, opt):
        from qcat_amd import native
        saved = self.__dict__.setdefault("_qcat_saved_options", {})
        if opt not in saved:
            saved[opt] = native.get_option(opt)

    def setenv(self, name, value, prepend=None):
        opt = option_of(name)          # (first: this may LOAD the library, which imports QCAT_HIP_* from the environment as it is then)
        orig_setenv(self, name, value, prepend)
        if opt:
            from qcat_amd import native
            remember(self, opt)
            native.set_option(opt, int(value) if str(value).strip() else 1)

    def delenv(self, name, raising=True):
        opt = option_of(name)
        orig_delenv(self, name, raising)
        if opt:
            from qcat_amd import native
            remember(self, opt)
            native.set_option(opt, None)

    def undo(self):
        saved = self.__dict__.pop("_qcat_saved_options", {})
        if saved:
            from qcat_amd import native
            for opt, value in saved.items():
                native.set_option(opt, value)
        orig_undo(self)

    MonkeyPatch.setenv, MonkeyPatch.delenv, MonkeyPatch.undo = setenv, delenv, undo
    MonkeyPatch._qcat_forwarding = True


_forward_switches_to_the_option_abi()


@pytest.fixture
def hip_options():
    """set library options for one test: hip_options(NO_GRAPH=1, BITSLICE_MIN=2048); None clears; restored afterwards"""
    from qcat_amd import native
    saved = {}

    def set_(**kw):
        for name, value in kw.items():
            if name not in saved:
                saved[name] = native.get_option(name)
            native.set_option(name, value)
    yield set_
    for name, value in saved.items():
        native.set_option(name, value)


# ---- which kernels a small batch runs on ------------------------------------------------------------------------------------------
# Batches of up to 20000 alignments (a few hundred reads) take the one-wave-per-alignment kernels (csrc/kernels_tiny.inc, round 5).
# The GPU modules below were written to pin the THROUGHPUT kernels -- binary16, bit-sliced, table, general -- on golden cases
# and seeded batches of exactly such sizes; they keep doing that (QCAT_HIP_NO_TINY for the test), and tests/test_tiny_gpu.py
# runs the same golden cases and seeded batches on the other path.  Modules that test the product's default behaviour end to
# end (the scanner API, the driver, the file loop, the one-wave kernels themselves) are left alone.
_DEFAULT_PATH_MODULES = {"test_tiny_gpu", "test_scan_api_gpu", "test_cli_gpu", "test_stream_gpu", "test_fastq_native", "test_comm_gpu",
                         "test_synth", "test_sg_align_gpu"}


@pytest.fixture(autouse=True)
def _throughput_kernels_for_small_batches(request):
    if "gpu" not in request.keywords or request.module.__name__ in _DEFAULT_PATH_MODULES:
        yield
        return
    from qcat_amd import native
    before = native.get_option("NO_TINY")
    native.set_option("NO_TINY", 1)
    yield
    native.set_option("NO_TINY", before)
