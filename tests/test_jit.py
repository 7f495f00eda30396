"""Run-time generated static-letter kernels for custom kits (qcat_amd/jit.py, qcat_kit_attach_code).

CPU: the generated translation unit compiles for gfx950 and binds every template / barcode group of
a custom kit.  GPU: a custom epi2me kit and a custom dual kit give byte-identical records, traces and
per-barcode rows with and without the generated kernels, both equal to the CPU oracle; the packed
interior scan (--detect-middle) works on the generated adapter kernels too."""
import os
import random

import numpy as np
import pytest
import yaml

import oracle_lib
import synth
from qcat_amd import jit, native, scanner

needs_hipcc = pytest.mark.skipif(jit.hipcc_path() is None, reason="hipcc not available")


def _write_kit(folder, name, kit, seq, set1, set2=None, trim_offset=0):
    def rows(bcs):
        return [{"name": "barcode%02d" % (i + 1), "id": i + 1, "sequence": s, "fwd_strand": True} for i, s in enumerate(bcs)]
    data = {"kit": kit, "auto_detect": False, "description": "test kit", "sequence": seq, "trim_offset": trim_offset,
            "barcode_set_1": rows(set1), "barcode_set_2": rows(set2) if set2 else []}
    with open(os.path.join(folder, name + ".yml"), "w") as fh:
        yaml.safe_dump(data, fh)


def _custom_kits(folder):
    rng = random.Random(99)
    bcs = ["".join(rng.choice("ACGT") for _ in range(24)) for _ in range(20)]
    _write_kit(folder, "A_5p", "CUSTOM", "GGTGCTGAT" + "N" * 24 + "TTAACCTTTCTGTTGGTGCTGATATTGCAA", bcs, trim_offset=7)
    _write_kit(folder, "A_3p", "CUSTOM", "GGTGCTGAT" + "N" * 24 + "TTAACCTACTTGCCTGTCGCTCTATCTTCGG", bcs, trim_offset=7)
    s1 = ["".join(rng.choice("ACGT") for _ in range(24)) for _ in range(6)]
    s2 = ["".join(rng.choice("ACGT") for _ in range(24)) for _ in range(9)]
    # (the dual scanner only looks at layouts of the kit named DUAL, like the reference)
    os.makedirs(os.path.join(folder, "dual"), exist_ok=True)
    _write_kit(os.path.join(folder, "dual"), "DUAL_5p", "DUAL", "AGGTTAC" + "N" * 24 + "CAGCACCTGGTGATG" + "N" * 24 + "TTAACCTTTCTGTTGG", s1, s2)


@needs_hipcc
def test_generated_unit_compiles_and_binds_everything(tmp_path):
    _custom_kits(str(tmp_path))
    for mode, kit, nt, ng in (("epi2me", "CUSTOM", 2, 2), ("dual", None, 1, 2)):
        det = scanner.factory(mode=mode, kit=kit, kit_folder=str(tmp_path) if kit else os.path.join(str(tmp_path), "dual"))
        plain = native.NativeKit(det.descriptor(), jit=False).describe()
        assert plain["packed"] == 1 and plain["n_static_templates"] == 0 and plain["n_static_groups"] == 0
        info = native.NativeKit(det.descriptor(), jit=True).describe()
        assert info["n_static_templates"] == info["n_templates"] == nt
        assert info["n_static_groups"] == info["n_groups"] == ng


def test_builtin_kits_need_no_code_generation(monkeypatch):
    monkeypatch.setattr(jit, "compile_source", lambda source: pytest.fail("built-in kits must not compile anything"))
    det = scanner.factory(kit="PBC096")
    info = native.NativeKit(det.descriptor(), jit=True).describe()
    assert info["n_static_groups"] == info["n_groups"]


@pytest.mark.gpu
@needs_hipcc
@pytest.mark.parametrize("mode,kit,t5,t3,middle", [("epi2me", "CUSTOM", 0, 1, False), ("dual", None, 0, -1, False),
                                                   ("epi2me", "CUSTOM", 0, 1, True)])
def test_generated_kernels_match_table_kernels_and_oracle(tmp_path, mode, kit, t5, t3, middle):
    _custom_kits(str(tmp_path))
    det = scanner.factory(mode=mode, kit=kit, kit_folder=str(tmp_path) if kit else os.path.join(str(tmp_path), "dual"),
                          scan_middle_adapter=middle)
    reads = synth.synth_batch(900, 31, det.layouts, t5, t3, error_rate=0.09)
    if middle:
        reads = [r + r if i % 3 == 0 else r for i, r in enumerate(reads)]
    reads += ["", "ACGTN" * 40, reads[0][:77]]
    d = det.descriptor()
    ctx = native.NativeContext(0)
    bases, offsets = native.pack_reads(reads)
    out = []
    for use_jit in (False, True):
        kit_h = native.NativeKit(d, jit=use_jit)
        assert (kit_h.describe()["n_static_groups"] > 0) == use_jit
        cnt = np.zeros(d.n_count_buckets, dtype=np.int64)
        recs, traces, rows = ctx.scan(kit_h, bases, offsets, counts=cnt, trace=True, rows=True)
        out.append((recs.tobytes(), {n: traces[n].copy() for n in native.TRACE_DTYPE.names}, rows.copy(), cnt))
    assert out[0][0] == out[1][0]
    for n in out[0][1]:
        assert np.array_equal(out[0][1][n], out[1][1][n]), n
    assert np.array_equal(out[0][2], out[1][2]) and np.array_equal(out[0][3], out[1][3])
    o_recs, o_cnt, o_traces, o_rows = oracle_lib.scan(d, reads, counts=True, trace=True, rows=True, threads=8)
    assert out[1][0] == o_recs.tobytes()
    assert np.array_equal(out[1][3], o_cnt) and np.array_equal(out[1][2], o_rows)
    if middle:
        assert (np.frombuffer(out[1][0], dtype=native.RESULT_DTYPE)["exit_status"] == 997).sum() > 10
