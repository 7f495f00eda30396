"""Rule R1 as a switch (round 6): the reference binds `parasail.sg_striped_32` when parasail reports SSE2 and plain
`parasail.sg` otherwise (qcat/scanner_base.py:20-26); the two place an alignment's END differently when the last row's and
the last column's maxima tie (include/qcat_hip.h QCAT_R1_STRIPED / QCAT_R1_SCALAR).  Every kernel family under
QCAT_R1_SCALAR against
  (a) the unmodified reference run with `parasail.sg` bound (tests/golden/golden_r1_scalar.json, make_golden.py --r1-scalar),
  (b) the CPU oracle with the same rule,
and the default rule still bit-identical to the round-5 fixtures (the other modules)."""
import numpy as np
import pytest

import helpers
import oracle_lib
import synth
from qcat_amd import config, native, scanner

pytestmark = pytest.mark.gpu

R1_CASES = [c["name"] for c in helpers.golden_r1_scalar()["cases"]]
_ctx = {}


def ctx():
    if "c" not in _ctx:
        _ctx["c"] = native.NativeContext(0)
    return _ctx["c"]


def scan(desc, reads, trace=True):
    kit = native.NativeKit(desc)
    bases, offsets = native.pack_reads(reads)
    cnt = np.zeros(desc.n_count_buckets, dtype=np.int64)
    if trace:
        recs, traces, rows = ctx().scan(kit, bases, offsets, counts=cnt, trace=True, rows=True)
    else:
        recs, traces, rows = ctx().scan(kit, bases, offsets, counts=cnt), None, None
    return recs, traces, rows, cnt


def same_as_oracle(desc, reads, recs, traces, rows, cnt):
    o_recs, o_cnt, o_traces, o_rows = oracle_lib.scan(desc, reads, counts=True, trace=True, rows=True, threads=8)
    assert recs.tobytes() == o_recs.tobytes()
    assert np.array_equal(cnt, o_cnt)
    if traces is not None:
        for name in native.TRACE_DTYPE.names:
            assert np.array_equal(traces[name], o_traces[name]), name
        assert np.array_equal(rows, o_rows)


@pytest.mark.parametrize("family", ["throughput", "one wave per alignment", "general"])
@pytest.mark.parametrize("name", R1_CASES)
def test_reference_with_parasail_sg_bound(name, family, hip_options):
    """records, per-end traces (every template's raw score and end_query) and every per-barcode raw score of the reference's
    Python with the scalar routine bound, on the binary16 / table kernels, on the one-wave kernels and on the general kernel"""
    if family == "throughput":
        hip_options(NO_TINY=1)
    elif family == "one wave per alignment":
        hip_options(NO_TINY=None, TINY_MAX_ENDS=4096)
    case = [c for c in helpers.golden_r1_scalar()["cases"] if c["name"] == name][0]
    det = helpers.make_scanner(case["mode"], case["kit"])
    reads = helpers.case_reads(case, det.layouts)
    with helpers.r1_rule("scalar"):
        desc = det.descriptor()
    if family == "general":
        hip_options(FORCE_GENERIC=1)
        _ctx["g"] = native.NativeContext(0)              # (the switch is read when a context is made)
        kit = native.NativeKit(desc)
        bases, offsets = native.pack_reads(reads)
        cnt = np.zeros(desc.n_count_buckets, dtype=np.int64)
        recs, traces, rows = _ctx["g"].scan(kit, bases, offsets, counts=cnt, trace=True, rows=True)
    else:
        recs, traces, rows, cnt = scan(desc, reads)
    helpers.assert_case_matches(case, recs, traces, rows, det.layouts)
    same_as_oracle(desc, reads, recs, traces, rows, cnt)


@pytest.mark.parametrize("mode,kit,ends", [("epi2me", "PBC096", native.ENDS_BOTH), ("epi2me", "NBD103/NBD104", native.ENDS_5P),
                                           ("dual", None, native.ENDS_BOTH), ("epi2me", None, native.ENDS_BOTH)])
def test_bit_sliced_adapter_scan_under_both_rules(mode, kit, ends, hip_options):
    """a batch big enough for the bit-sliced adapter kernels (abs_core.h: abs_decide), half of it adapter-free reads -- where the
    borders' maxima tie -- under both rules against the oracle, every record; the rules must also part somewhere"""
    det = scanner.factory(mode=mode, kit=kit)
    t5 = len(det.layouts) - 1 if kit else (3 if mode == "epi2me" else 1)
    t3 = 0 if len(det.layouts) > 1 else -1
    if kit is None and mode == "epi2me":
        t3 = 2
    n = 24000
    reads = synth.synth_batch(n, 20260930, det.layouts, t5, t3, error_rate=0.08, no_adapter_fraction=0.5)
    reads[3], reads[4], reads[5] = "", "N" * 300, "ACGT" * 90
    bases, offsets = native.pack_reads(reads)
    hip_options(ADAPTER_BITSLICE_MIN=8192, BITSLICE_MIN=16384)
    got = {}
    for rule in ("striped", "scalar"):
        with helpers.r1_rule(rule):
            desc = det.descriptor(ends=ends)
        want, want_cnt = oracle_lib.scan(desc, reads, counts=True, threads=8)
        for variant in ("bit-sliced", "binary16"):
            hip_options(NO_ADAPTER_BITSLICE=None if variant == "bit-sliced" else 1)
            cnt = np.zeros(desc.n_count_buckets, dtype=np.int64)
            c = native.NativeContext(0)
            lib = native.HipLibrary.get().lib
            native.HipLibrary.get().check(lib.qcat_ctx_set_timing(c.handle, 1))
            recs = c.scan(native.NativeKit(desc), bases, offsets, counts=cnt)
            import ctypes as C
            names = (C.c_char_p * 16)()
            ms = (C.c_float * 16)()
            k = lib.qcat_ctx_last_timing(c.handle, names, ms, 16)
            ran = [names[i].decode() for i in range(k)]
            assert ("k_adapter_bitslice" in ran) == (variant == "bit-sliced"), (variant, ran)
            bad = np.nonzero(recs != want)[0]
            assert len(bad) == 0, (rule, variant, bad[:10], recs[bad[:3]], want[bad[:3]])
            assert np.array_equal(cnt, want_cnt)
        got[rule] = want
    assert np.count_nonzero(got["striped"] != got["scalar"]) > 0


def test_detect_middle_under_the_scalar_rule(hip_options):
    """--detect-middle: the interior scans (binary16 interior kernel and the bit-sliced interior adapter scan) place their
    adapter ends by the kit's rule as well"""
    det = scanner.factory(mode="epi2me", kit="PBC096", scan_middle_adapter=True)
    base = synth.synth_batch(6000, 777, det.layouts, 1, 0, error_rate=0.05, no_adapter_fraction=0.3)
    reads = [base[i] + base[i + 3000] if i % 3 == 0 else base[i] for i in range(3000)]
    bases, offsets = native.pack_reads(reads)
    with helpers.r1_rule("scalar"):
        desc = det.descriptor()
    want, want_cnt = oracle_lib.scan(desc, reads, counts=True, threads=8)
    for variant in ("default", "no interior bit-slice"):
        hip_options(MIDDLE_NO_ABS=None if variant == "default" else 1, MIDDLE_ABS_MIN=None if variant != "default" else 1024)
        cnt = np.zeros(desc.n_count_buckets, dtype=np.int64)
        recs = native.NativeContext(0).scan(native.NativeKit(desc), bases, offsets, counts=cnt)
        bad = np.nonzero(recs != want)[0]
        assert len(bad) == 0, (variant, bad[:10], recs[bad[:3]], want[bad[:3]])
        assert np.array_equal(cnt, want_cnt)
    assert np.count_nonzero(want["exit_status"] == 997) > 0


def test_module_level_helpers_follow_the_rule():
    """qcat_sg_align with QCAT_SG_R1_SCALAR (the module-level helpers have no kit) against the oracle's qo_sg_rule"""
    cfg = config.qcatConfig()
    rng = np.random.RandomState(7)
    qs = ["".join("ACGT"[i] for i in rng.randint(0, 3, size=int(rng.randint(1, 150)))) for _ in range(500)]
    ts = ["".join("ACGTN"[i] for i in rng.randint(0, 3, size=int(rng.randint(1, 60)))) for _ in range(500)]
    n_diff = 0
    res = {}
    for rule in ("striped", "scalar"):
        with helpers.r1_rule(rule):
            out = native.sg_align(ctx(), qs, ts, cfg.gap_open, cfg.gap_extend, cfg.matrix.table)
        res[rule] = out
        for q, t, o in zip(qs, ts, out):
            want = oracle_lib.sg(q, t, cfg.gap_open, cfg.gap_extend, cfg.matrix.table, rule=native.R1_SCALAR if rule == "scalar" else native.R1_STRIPED)
            assert (int(o["score"]), int(o["end_query"]), int(o["end_ref"])) == want, (rule, q, t)
    n_diff = np.count_nonzero(res["striped"]["end_query"] != res["scalar"]["end_query"])
    assert n_diff > 0 and np.array_equal(res["striped"]["score"], res["scalar"]["score"])
