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

needs_hipcc = pytest.mark.skipif(jit.compiler() is None, reason="neither libhiprtc nor hipcc available")


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
    # a template that two stages of <= 52 columns cannot hold (102 columns with the barcode near the front, like VMK001: the
    # first stage has to end before the barcode's border -- four wide stages) and a short one whose four stages hold <= 13
    # columns (40 columns: the medium-batch form)
    long_seq = "".join(rng.choice("ACGT") for _ in range(20)) + "N" * 24 + "".join(rng.choice("ACGT") for _ in range(58))
    _write_kit(folder, "L_5p", "LONGKIT", long_seq, bcs[:12])
    _write_kit(folder, "L_3p", "LONGKIT", "GGTGCTG" + "N" * 24 + "TTAACCTAC", bcs[:12])


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
        # ... and the bit-sliced kernels of every group have the kit's letters compiled in (qj_bs_<group>)
        assert plain["bitslice_groups"] == ng and info["bitslice_groups"] == ng * 0x10001
        # ... and every template got a bit-sliced ADAPTER plan of two stages (qj_abs_<t>, round 4)
        assert plain["bitslice_templates"] == 0 and info["bitslice_templates"] == nt
    det = scanner.factory(mode="epi2me", kit="LONGKIT", kit_folder=str(tmp_path))
    info = native.NativeKit(det.descriptor(), jit=True).describe()
    # sorted by file name: L_3p (40 columns: two stages AND four narrow ones), L_5p (102 columns: four wide stages only)
    assert [len(l.sequence) for l in det.layouts] == [40, 102]
    assert info["bitslice_templates"] == 1 + 0x100 + 0x10000


@pytest.mark.skipif(jit.hiprtc() is None or jit.hipcc_path() is None, reason="needs both libhiprtc and hipcc")
def test_hiprtc_and_hipcc_both_produce_loadable_units(tmp_path, monkeypatch):
    """in-process hipRTC (device-only rtc_prelude.inc) is the default; hipcc --genco (jit_prelude.inc) the
    fall-back: both must compile the same generated translation unit into a gfx950 code object"""
    _custom_kits(str(tmp_path))
    det = scanner.factory(mode="dual", kit_folder=os.path.join(str(tmp_path), "dual"))
    source = jit.generate(det.descriptor())[0]
    assert jit.compiler() == "hiprtc"
    a = jit._compile_hiprtc(source)
    b = jit._compile_hipcc(source)
    for blob in (a, b):
        assert blob[:4] in (b"\x7fELF", b"__CL") and len(blob) > 10000      # code object or clang offload bundle
        for sym in (b"qj_ad_0", b"qj_am_0", b"qj_bc_0", b"qj_bc_1", b"qj_bs_0", b"qj_bs_1", b"qj_abs_0"):
            assert sym in blob
    monkeypatch.setenv("QCAT_AMD_JIT_COMPILER", "hipcc")
    assert jit.compiler() == "hipcc"


@needs_hipcc
def test_cached_code_objects_are_verified_before_use(tmp_path, monkeypatch):
    monkeypatch.setenv("QCAT_AMD_JIT_CACHE", str(tmp_path / "cache"))
    _custom_kits(str(tmp_path))
    det = scanner.factory(mode="dual", kit_folder=os.path.join(str(tmp_path), "dual"))
    source = jit.generate(det.descriptor())[0]
    blob = jit.compile_source(source)
    files = sorted(os.listdir(str(tmp_path / "cache")))
    assert len(files) == 2 and files[1] == files[0] + ".sha256"
    path = os.path.join(str(tmp_path / "cache"), files[0])
    calls = []
    real = jit._compile_hiprtc if jit.compiler() == "hiprtc" else jit._compile_hipcc
    monkeypatch.setattr(jit, "_compile_hiprtc" if jit.compiler() == "hiprtc" else "_compile_hipcc",
                        lambda src: calls.append(1) or real(src))
    assert jit.compile_source(source) == blob and not calls            # served from the cache
    with open(path, "r+b") as fh:                                       # corrupt one byte of the cached object
        fh.seek(100)
        fh.write(b"\xff")
    assert jit.compile_source(source) == blob and calls == [1]         # digest mismatch: compiled again, cache repaired
    assert jit._cache_load(path) == blob
    os.remove(path + ".sha256")                                         # an object without its digest is not trusted either
    assert jit._cache_load(path) is None


@needs_hipcc
def test_auto_mode_compiles_in_the_background_and_upgrades_the_kit(tmp_path, monkeypatch):
    monkeypatch.setenv("QCAT_AMD_JIT_CACHE", str(tmp_path / "cache"))
    monkeypatch.setenv("QCAT_AMD_JIT", "auto")
    _custom_kits(str(tmp_path))
    det = scanner.factory(mode="epi2me", kit="CUSTOM", kit_folder=str(tmp_path))
    kit = native.NativeKit(det.descriptor())
    first = kit.handle
    assert kit.jit_thread is not None
    info = kit.wait_for_code(600)
    assert info["n_static_templates"] == info["n_templates"] == 2 and info["n_static_groups"] == info["n_groups"] == 2
    assert kit.handle is not first and kit.describe(first)["n_static_groups"] == 0   # the old handle stays valid
    monkeypatch.setenv("QCAT_AMD_JIT", "0")
    assert native.NativeKit(det.descriptor()).jit_thread is None
    # shipped kits never start a compile
    monkeypatch.setenv("QCAT_AMD_JIT", "auto")
    assert native.NativeKit(scanner.factory(kit="PBC096").descriptor()).jit_thread is None


def test_no_compiler_means_table_kernels_and_a_warning(tmp_path, monkeypatch, caplog):
    monkeypatch.setattr(jit, "compiler", lambda: None)
    monkeypatch.setattr(jit, "_warned", [])
    _custom_kits(str(tmp_path))
    det = scanner.factory(mode="epi2me", kit="CUSTOM", kit_folder=str(tmp_path))
    import logging
    with caplog.at_level(logging.WARNING):
        kit = native.NativeKit(det.descriptor())
    assert kit.describe()["n_static_groups"] == 0 and "table kernels" in caplog.text
    with pytest.raises(RuntimeError, match="cannot generate kernels"):
        native.NativeKit(det.descriptor(), jit=True)


def test_attach_code_rejects_malformed_pair_lists(tmp_path):
    """qcat_kit_attach_code validates what the barcode kernel will index with."""
    import ctypes as C
    _custom_kits(str(tmp_path))
    det = scanner.factory(mode="epi2me", kit="CUSTOM", kit_folder=str(tmp_path))
    hip = native.HipLibrary.get()
    n = 20

    def attach(entries, offs=None):
        kit = native.NativeKit(det.descriptor(), jit=False)
        tf = (C.c_int32 * 16)()
        gf = (C.c_int32 * 32)()
        gf[0] = 1
        flat = [v for e in entries for v in e]
        po = (C.c_int32 * 33)(*(offs or [0] + [len(entries)] * 32))
        pe = (C.c_int32 * max(1, len(flat)))(*flat)
        return hip.lib.qcat_kit_attach_code(kit.handle, b"\x7fELF" + b"\0" * 64, 68, tf, gf, po, pe), kit

    good = [(i, 2 * i, 2 * i + 1) for i in range(n // 2)]
    rc, kit = attach(good)
    assert rc == 0 and kit.describe()["n_static_groups"] == 1
    for bad in (good[:-1] + [(9, 18, 25)],            # barcode index outside the set
                good[:-1] + [(9, 18, 18)],            # the same barcode twice
                good[:-1],                            # a barcode is not covered
                good[:-1] + [(9, -1, 19)],            # half 0 must name a barcode
                good[:-1] + [(-1, 18, 19)]):          # negative pair case
        rc, kit = attach(bad)
        assert rc == -1, bad[-1]
        assert kit.describe()["n_static_groups"] == 0            # nothing was bound
    rc, _ = attach(good, offs=[0, 10, 5] + [5] * 30)
    assert rc == -1


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


@pytest.mark.gpu
@needs_hipcc
@pytest.mark.parametrize("mode,kit,t5,t3", [("epi2me", "CUSTOM", 0, 1), ("dual", None, 0, -1), ("epi2me", "LONGKIT", 1, 0)])
def test_generated_bit_sliced_adapter_plans_match_the_oracle(tmp_path, monkeypatch, mode, kit, t5, t3):
    """Round 4: a custom kit's templates get bit-sliced ADAPTER plans at run time (qcat_amd/abs_plan.py through jit.py:
    two stages, four narrow stages for medium batches, four wide stages for a template too long for two).  A debug
    scan of a mixed batch with the path forced (either pipeline form) against the oracle: per-template raw score and
    end_query of every window, every per-barcode row, the records, the counts."""
    _custom_kits(str(tmp_path))
    det = scanner.factory(mode=mode, kit=kit, kit_folder=str(tmp_path) if kit else os.path.join(str(tmp_path), "dual"))
    reads = synth.synth_batch(5000, 77, det.layouts, t5, t3, error_rate=0.08)
    for i in range(0, len(reads), 9):
        reads[i] = reads[i][:30 + (i % 200)] if i % 2 else reads[i][:50] + "N" + reads[i][51:]
    reads += ["", "ACGTN" * 40, ("ACG" * 80)[:170]]
    d = det.descriptor()
    o_recs, o_cnt, o_traces, o_rows = oracle_lib.scan(d, reads, counts=True, trace=True, rows=True, threads=8)
    kit_h = native.NativeKit(d, jit=True)
    assert kit_h.describe()["bitslice_templates"] > 0
    bases, offsets = native.pack_reads(reads)
    monkeypatch.setenv("QCAT_HIP_ADAPTER_BITSLICE_MIN", "1")
    for stages in ("2", "4"):
        monkeypatch.setenv("QCAT_HIP_ABS_STAGES", stages)
        ctx = native.NativeContext(0)
        lib = native.HipLibrary.get().lib
        native.HipLibrary.get().check(lib.qcat_ctx_set_timing(ctx.handle, 1))
        cnt = np.zeros(d.n_count_buckets, dtype=np.int64)
        recs, traces, rows = ctx.scan(kit_h, bases, offsets, counts=cnt, trace=True, rows=True)
        import ctypes as C
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        ran = [names[i].decode() for i in range(lib.qcat_ctx_last_timing(ctx.handle, names, ms, 16))]
        assert "k_adapter_bitslice" in ran, ran
        for n in ("tpl_raw", "tpl_end"):
            assert np.array_equal(traces[n], o_traces[n]), (stages, n)
        assert recs.tobytes() == o_recs.tobytes() and np.array_equal(rows, o_rows) and np.array_equal(cnt, o_cnt)
