"""The same C ABI implemented twice: oracle/libqcat_cpu.so exports the host-buffer entry
points of include/qcat_hip.h over the CPU oracle, so ONE ctypes driver (tests/abi_twin.py) runs against
both libraries.  CPU: the twin loads, exports the common subset with the header's signatures and
reproduces the reference's golden outputs through that ABI.  GPU: both libraries, same calls, same bytes."""
import re

import numpy as np
import pytest

import abi_twin
import helpers
import synth
from qcat_amd import native, scanner


def test_cpu_twin_exports_the_common_subset_with_the_headers_names():
    twin = abi_twin.Abi(abi_twin.CPU)
    for name in abi_twin.COMMON:
        assert hasattr(twin.lib, name), name
    header = open(abi_twin.ROOT + "/include/qcat_hip.h").read()
    for name in abi_twin.COMMON:
        assert re.search(r"\b%s\s*\(" % name, header), name
    assert twin.lib.qcat_abi_version() == native.ABI_VERSION and twin.lib.qcat_device_count() == 0
    det = scanner.factory(kit="PBC096")
    bad = det.descriptor()
    bad.desc.abi_version = 1
    with pytest.raises(RuntimeError, match="ABI version"):
        twin.run(bad, ["ACGT"])


def test_cpu_twin_reproduces_the_golden_case_through_the_abi():
    case = [c for c in helpers.golden()["cases"] if c["name"] == "synth:PBC096:e0.08"][0]
    det = helpers.make_scanner(case["mode"], case["kit"])
    reads = helpers.case_reads(case, det.layouts)
    got = abi_twin.Abi(abi_twin.CPU).run(det.descriptor(), reads, votes=True)
    recs = np.frombuffer(got["records"], dtype=native.RESULT_DTYPE)
    helpers.assert_case_matches(case, recs, got["traces"], got["rows"], det.layouts)
    assert got["debug_records"] == got["records"] and got["buckets"] == det.descriptor().n_count_buckets
    assert got["votes"].sum() == len(reads)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,kit,ends,t5,t3", [("epi2me", "PBC096", native.ENDS_BOTH, 1, 0), ("epi2me", "NBD103/NBD104", native.ENDS_5P, 1, 0),
                                                 ("epi2me", None, native.ENDS_BOTH, 3, 2), ("dual", None, native.ENDS_BOTH, 1, 0),
                                                 ("simple", "standard", native.ENDS_BOTH, 1, 0)])
def test_both_libraries_same_calls_same_bytes(mode, kit, ends, t5, t3):
    det = scanner.factory(mode=mode, kit=kit)
    lays = det.layouts if mode != "simple" else scanner.factory(kit="PBK004/LWB001").layouts
    reads = synth.synth_batch(600, 2718, lays, t5, t3, error_rate=0.09) + ["", "A", "N" * 200]
    seqs = [reads[0], reads[1][:200], reads[2][100:700], ""]
    d = det.descriptor(ends=ends, min_read_length=500, trim=True)
    votes = mode == "epi2me" and ends == native.ENDS_BOTH
    hip = abi_twin.Abi(abi_twin.HIP).run(d, reads, sequences=seqs, votes=votes)
    cpu = abi_twin.Abi(abi_twin.CPU).run(d, reads, sequences=seqs, votes=votes)
    assert hip["buckets"] == cpu["buckets"]
    for key in ("records", "debug_records", "sequences"):
        assert hip[key] == cpu[key], key
    assert np.array_equal(hip["counts"], cpu["counts"])
    skip = ("tpl_raw", "tpl_end", "best_tpl", "best_raw", "region_path", "region_start", "region_len", "used_tpl") if mode == "simple" else ()
    for name in native.TRACE_DTYPE.names:
        if name not in skip:
            assert np.array_equal(hip["traces"][name], cpu["traces"][name]), name
    assert np.array_equal(hip["rows"][:, 0 if mode != "dual" else slice(None), :], cpu["rows"][:, 0 if mode != "dual" else slice(None), :])
    if votes:
        assert np.array_equal(hip["votes"], cpu["votes"]) and np.array_equal(hip["first"], cpu["first"])
