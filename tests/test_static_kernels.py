"""Static-letter kernels (qcat_amd/csrc/kernels_static.inc + static_generated.inc).

CPU: the generated source is in sync with resources/kits.json, and every built-in kit selection
binds ALL of its templates and barcode groups to generated kernels (a silent fall-back to the table
kernels would only cost speed, so it has to be caught here).
GPU: the static, the table (fp16 and u16 lanes) and the single-stream paths produce byte-identical
records, traces and per-barcode rows, and equal the CPU oracle -- including reads with N / X / other
letters, which take the slow score path of the static adapter kernels."""
import importlib.util
import os

import numpy as np
import pytest

import oracle_lib
import synth
from qcat_amd import native, scanner

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _generator():
    spec = importlib.util.spec_from_file_location("gen_static_kernels", os.path.join(ROOT, "tools", "gen_static_kernels.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_generated_source_is_in_sync_with_the_kit_bundle():
    gen = _generator()
    text, n_kernels, n_targets, n_templates, bs_text = gen.render()
    assert open(gen.OUT).read() == text, "run python tools/gen_static_kernels.py"
    assert open(gen.BS_OUT).read() == bs_text, "run python tools/gen_static_kernels.py"
    assert n_kernels >= 16 and n_targets >= 600 and n_templates >= 14


@pytest.mark.parametrize("mode", ["epi2me", "dual"])
def test_every_builtin_kit_selection_is_fully_static(mode):
    for kit in [None] + sorted(scanner.get_kits()):
        det = scanner.factory(mode=mode, kit=kit)
        info = native.NativeKit(det.descriptor()).describe()
        assert info["packed"] == 1 and info["barcode_f16"] == 1 and info["adapter_f16"] == 1, (mode, kit, info)
        assert info["n_static_templates"] == info["n_templates"] > 0, (mode, kit, info)
        assert info["n_static_groups"] == info["n_groups"] > 0, (mode, kit, info)


def test_unknown_targets_keep_the_table_kernels(tmp_path):
    """a custom kit: PBC096's first template with a changed last letter and reversed barcodes -> no
    registry hit; the same kit with the shipped sequences binds to the generated kernels"""
    import yaml
    det = scanner.factory(kit="PBC096")
    lay = det.layouts[0]

    def write(name, kit, seq, barcodes):
        rows = [{"name": "barcode%02d" % (i + 1), "id": i + 1, "sequence": s, "fwd_strand": True} for i, s in enumerate(barcodes)]
        data = {"kit": kit, "auto_detect": False, "description": "test kit", "sequence": seq, "trim_offset": 0,
                "barcode_set_1": rows, "barcode_set_2": []}
        (tmp_path / (name + ".yml")).write_text(yaml.safe_dump(data))

    bcs = [b.sequence for b in lay.get_barcode_set(0)[:8]]
    write("a", "CUSTOMA", lay.sequence[:-1] + ("A" if lay.sequence[-1] != "A" else "C"), [b[::-1] for b in bcs])
    write("b", "CUSTOMB", lay.sequence, bcs)
    info = native.NativeKit(scanner.factory(kit="CUSTOMA", kit_folder=str(tmp_path)).descriptor()).describe()
    assert info["packed"] == 1 and info["n_static_templates"] == 0 and info["n_static_groups"] == 0
    info = native.NativeKit(scanner.factory(kit="CUSTOMB", kit_folder=str(tmp_path)).descriptor()).describe()
    assert info["n_static_templates"] == 1 and info["n_static_groups"] == 1


# ---------------------------------------------------------------------------------------------
gpu = pytest.mark.gpu
_ctx = {}


def ctx():
    if "c" not in _ctx:
        _ctx["c"] = native.NativeContext(0)
    return _ctx["c"]


def run(det, reads, ends=native.ENDS_BOTH):
    d = det.descriptor(ends=ends)
    kit = native.NativeKit(d)
    bases, offsets = native.pack_reads(reads)
    cnt = np.zeros(d.n_count_buckets, dtype=np.int64)
    recs, traces, rows = ctx().scan(kit, bases, offsets, counts=cnt, trace=True, rows=True)
    return d, recs.tobytes(), {n: traces[n].copy() for n in native.TRACE_DTYPE.names}, rows.copy(), cnt


def same(a, b):
    assert a[1] == b[1]
    for n in a[2]:
        assert np.array_equal(a[2][n], b[2][n]), n
    assert np.array_equal(a[3], b[3])
    assert np.array_equal(a[4], b[4])


VARIANTS = [{"QCAT_HIP_NO_STATIC": "1"}, {"QCAT_HIP_NO_STATIC_ADAPTER": "1"}, {"QCAT_HIP_ONE_STREAM": "1"},
            {"QCAT_HIP_NO_STATIC": "1", "QCAT_HIP_BARCODE_U16": "1"}, {"QCAT_HIP_CHUNK_BARCODES": "1"},
            {"QCAT_HIP_CHUNK_BARCODES": "1000"}, {"QCAT_HIP_ONE_QUEUE": "1"}, {"QCAT_HIP_NO_FUSED_ADAPTER": "1"}]


@gpu
@pytest.mark.parametrize("mode,kit,t5,t3,n", [
    ("epi2me", "NBD103/NBD104", 1, 0, 1500), ("epi2me", "PBC096", 1, 0, 1200), ("epi2me", None, 3, 2, 500),
    ("epi2me", "VMK001", 0, -1, 500), ("dual", None, 1, 0, 700)])
def test_static_and_table_paths_agree_and_match_the_oracle(monkeypatch, mode, kit, t5, t3, n):
    det = scanner.factory(mode=mode, kit=kit)
    reads = synth.synth_batch(n, 77, det.layouts, t5, t3, error_rate=0.1)
    reads += ["", "A", "ACGT" * 10, reads[0][:31], reads[1][:75]]
    base = run(det, reads)
    o_recs, o_cnt, o_traces, o_rows = oracle_lib.scan(base[0], reads, counts=True, trace=True, rows=True, threads=8)
    assert base[1] == o_recs.tobytes()
    assert np.array_equal(base[4], o_cnt) and np.array_equal(base[3], o_rows)
    for name in native.TRACE_DTYPE.names:
        assert np.array_equal(base[2][name], o_traces[name]), name
    for env in VARIANTS:
        with monkeypatch.context() as m:
            for k, v in env.items():
                m.setenv(k, v)
            same(base, run(det, reads))


@gpu
@pytest.mark.parametrize("kit", ["NBD103/NBD104", "RBK004", None])
def test_reads_with_n_x_and_other_letters(kit):
    """non-ACGT letters inside the windows: slow score path of the static adapter kernels and the
    shared `special` pool of the barcode kernels"""
    det = scanner.factory(kit=kit)
    rng = np.random.RandomState(5)
    reads = synth.synth_batch(600, 9, det.layouts, 1 if len(det.layouts) > 1 else 0, 0 if len(det.layouts) > 1 else -1,
                              error_rate=0.06)
    out = []
    for i, r in enumerate(reads):
        r = list(r)
        for _ in range(i % 7):                          # 0..6 substitutions, biased to both ends
            pos = rng.randint(0, min(len(r), 160)) if rng.rand() < 0.5 else len(r) - 1 - rng.randint(0, min(len(r), 160))
            r[pos] = "NXRnx-*"[rng.randint(0, 7)]
        out.append("".join(r))
    base = run(det, out)
    o_recs, o_cnt, o_traces, o_rows = oracle_lib.scan(base[0], out, counts=True, trace=True, rows=True, threads=8)
    assert base[1] == o_recs.tobytes()
    assert np.array_equal(base[3], o_rows)
    for name in native.TRACE_DTYPE.names:
        assert np.array_equal(base[2][name], o_traces[name]), name


def _scaled_barcode_cfg(scale):
    from qcat_amd import config
    cfg = config.qcatConfig()
    cfg._matrix_barcode = config.ScoreMatrix(cfg.matrix_barcode.table.astype(np.int64) * scale)   # not configurable in the reference
    return cfg


def test_fp16_lanes_are_only_used_when_exact():
    """binary16 barcode lanes need one-byte score encodings AND every DP value below 2048"""
    det = scanner.factory(kit="PBC096")
    assert native.NativeKit(det.descriptor(qcat_config=_scaled_barcode_cfg(3))).describe()["barcode_f16"] == 1    # W' = 5, 0
    assert native.NativeKit(det.descriptor(qcat_config=_scaled_barcode_cfg(30))).describe()["barcode_f16"] == 0   # 32*64 > 2047
    assert native.NativeKit(det.descriptor(qcat_config=_scaled_barcode_cfg(7))).describe()["barcode_f16"] == 0    # W' = 9 = 0x4880


@gpu
@pytest.mark.parametrize("scale", [3, 7, 30])
def test_scaled_barcode_matrix_matches_the_oracle(scale):
    det = scanner.factory(kit="PBC096")
    cfg = _scaled_barcode_cfg(scale)
    reads = synth.synth_batch(500, 3, det.layouts, 1, 0, error_rate=0.1)
    d = det.descriptor(qcat_config=cfg)
    kit = native.NativeKit(d)
    bases, offsets = native.pack_reads(reads)
    recs, traces, rows = ctx().scan(kit, bases, offsets, trace=True, rows=True)
    o_recs, o_traces, o_rows = oracle_lib.scan(d, reads, trace=True, rows=True, threads=8)
    assert recs.tobytes() == o_recs.tobytes()
    assert np.array_equal(rows, o_rows)


@gpu
@pytest.mark.parametrize("mode,kit,t5,t3", [("epi2me", "NBD103/NBD104", 1, 0), ("epi2me", "PBC096", 1, 0),
                                            ("epi2me", None, 3, 2), ("dual", None, 1, 0), ("epi2me", "RBK004", 0, -1)])
def test_packed_detect_middle_matches_generic_kernel_and_oracle(monkeypatch, mode, kit, t5, t3):
    """--detect-middle on the packed interior kernels (kernels_middle.inc): chimeric reads (adapters in
    the interior, either strand), interiors spanning several 152-row blocks and length classes, reads too
    short to have an interior, one interior beyond the packed length classes (general kernel), N runs."""
    det = scanner.factory(mode=mode, kit=kit, scan_middle_adapter=True)
    base = synth.synth_batch(260, 4242, det.layouts, t5, t3, error_rate=0.08)
    rng = np.random.RandomState(11)
    comp = {"A": "T", "T": "A", "G": "C", "C": "G"}
    reads = []
    for j, r in enumerate(base):
        kind = j % 6
        if kind == 0:
            reads.append(r + r)                                         # same barcode at both ends, adapters inside
        elif kind == 1:
            rc = "".join(comp.get(ch, "N") for ch in reversed(r))
            reads.append(r[:len(r) // 2] + rc + r[len(r) // 2:])       # reverse-strand adapter inside
        elif kind == 2:
            reads.append(r[:150 + rng.randint(0, 400)] + r[-170:])     # short interiors, many length classes
        elif kind == 3:
            reads.append(r + "".join("ACGT"[rng.randint(0, 4)] for _ in range(rng.randint(100, 2500))) + r)
        elif kind == 4:
            reads.append(r[:300] + "N" * rng.randint(1, 90) + r[300:])
        else:
            reads.append(r)
    reads += ["", "ACGT" * 70, base[0][:299], base[1][:300], base[2][:301], base[3][:364], base[4][:365],
              base[5] + "".join("ACGT"[rng.randint(0, 4)] for _ in range(17000)) + base[5]]
    d = det.descriptor()
    kit_h = native.NativeKit(d)
    bases, offsets = native.pack_reads(reads)
    cnt = np.zeros(d.n_count_buckets, dtype=np.int64)
    recs = ctx().scan(kit_h, bases, offsets, counts=cnt)
    with monkeypatch.context() as m:
        m.setenv("QCAT_HIP_MIDDLE_GENERIC", "1")
        cnt_g = np.zeros(d.n_count_buckets, dtype=np.int64)
        recs_g = ctx().scan(kit_h, bases, offsets, counts=cnt_g)
    assert recs.tobytes() == recs_g.tobytes()
    assert np.array_equal(cnt, cnt_g)
    o_recs, o_cnt = oracle_lib.scan(d, reads, counts=True, threads=8)
    assert recs.tobytes() == o_recs.tobytes()
    assert np.array_equal(cnt, o_cnt)
    assert (recs["exit_status"] == 997).sum() > 20
    # the bit-sliced interior adapter scan (kernels_abs_mid.inc; big batches only by default) forced onto this batch: one big
    # tile holds every length class (front padding of hundreds of rows), the N runs go back to the binary16 kernel per tile
    # of 128; then with a plane buffer that holds no tile / only the first ones (the rest falls back)
    import ctypes
    lib = native.HipLibrary.get().lib
    tiles = (ctypes.c_uint32 * 4)()
    assert lib.qcat_ctx_middle_bitslice_tiles(ctx().handle, tiles) == 0 and list(tiles) == [0, 0, 0, 0]
    # ... and both kernel forms: templates of up to 46 columns walk a tile on one wave (big batches, or QCAT_HIP_MIDDLE_ABS_ONE_WAVE=1),
    # the others (and medium batches, or everything with QCAT_HIP_MIDDLE_ABS_ONE_WAVE=0) on the two-wave pipeline
    for rows_cap, on_path, one_wave in ((None, True, "1"), (None, True, "0"), ("10", False, None), ("3000", None, "1"), ("3000", None, "0")):
        with monkeypatch.context() as m:
            m.setenv("QCAT_HIP_MIDDLE_ABS_MIN", "1")
            if rows_cap:
                m.setenv("QCAT_HIP_MIDDLE_ABS_ROWS", rows_cap)
            if one_wave:
                m.setenv("QCAT_HIP_MIDDLE_ABS_ONE_WAVE", one_wave)
            cnt_b = np.zeros(d.n_count_buckets, dtype=np.int64)
            recs_b = ctx().scan(kit_h, bases, offsets, counts=cnt_b)
            assert lib.qcat_ctx_middle_bitslice_tiles(ctx().handle, tiles) == 0
        assert recs_b.tobytes() == o_recs.tobytes(), (rows_cap, one_wave, list(tiles))
        assert np.array_equal(cnt_b, o_cnt)
        assert tiles[1] >= 1 and tiles[3] >= tiles[1]
        if on_path is True:
            assert tiles[0] >= 1 and 1 <= tiles[2] < tiles[3], list(tiles)       # (the N runs: some tiles of 128, not all)
        if on_path is False:
            assert tiles[0] == 0 and tiles[2] >= 1, list(tiles)


@gpu
@pytest.mark.parametrize("match,mismatch,gap", [(4, -3, 1), (6, -1, 3), (9, -4, 4)])
def test_packed_detect_middle_with_custom_scoring(match, mismatch, gap):
    """non-default adapter scoring: the static kernels take their scores from the kit at run time; where
    the binary16 bound does not hold (last case) the table / general kernels run -- same answers"""
    from qcat_amd import config
    cfg = config.qcatConfig()
    cfg.match, cfg.mismatch = match, mismatch
    cfg.gap_open = cfg.gap_extend = gap
    cfg.update_matrix()
    det = scanner.factory(kit="PBK004/LWB001", scan_middle_adapter=True)
    base = synth.synth_batch(240, 808, det.layouts, 1, 0, error_rate=0.07)
    reads = [r + r if i % 2 == 0 else r for i, r in enumerate(base)] + ["", "ACGT" * 90]
    d = det.descriptor(qcat_config=cfg)
    kit_h = native.NativeKit(d)
    info = kit_h.describe()
    assert info["adapter_f16"] == (1 if (match + 2 * gap) * 128 + gap * 288 <= 2047 else 0)
    bases, offsets = native.pack_reads(reads)
    cnt = np.zeros(d.n_count_buckets, dtype=np.int64)
    recs = ctx().scan(kit_h, bases, offsets, counts=cnt)
    o_recs, o_cnt = oracle_lib.scan(d, reads, counts=True, threads=8)
    assert recs.tobytes() == o_recs.tobytes()
    assert np.array_equal(cnt, o_cnt)


def _subset_kit(tmp_path, picks):
    import yaml
    lay = scanner.factory(kit="PBC096").layouts[0]
    bcs = lay.get_barcode_set(0)
    rows = [{"name": "barcode%02d" % (i + 1), "id": i + 1, "sequence": bcs[p].sequence, "fwd_strand": True} for i, p in enumerate(picks)]
    data = {"kit": "SUBSET", "auto_detect": False, "description": "subset", "sequence": lay.sequence, "trim_offset": 0,
            "barcode_set_1": rows, "barcode_set_2": []}
    (tmp_path / "s.yml").write_text(yaml.safe_dump(data))
    return scanner.factory(kit="SUBSET", kit_folder=str(tmp_path))


def test_any_subset_of_a_known_barcode_family_is_static(tmp_path):
    """the generated chains come in target pairs; a kit that uses only one target of a pair (or the
    targets in another order) still binds: the unused half is computed and dropped"""
    det = _subset_kit(tmp_path, [95, 2, 49, 10, 9])
    info = native.NativeKit(det.descriptor()).describe()
    assert info["n_static_templates"] == 1 and info["n_static_groups"] == 1


@gpu
def test_subset_kit_matches_the_oracle(tmp_path):
    det = _subset_kit(tmp_path, [95, 2, 49, 10, 9])
    reads = synth.synth_batch(800, 5, det.layouts, 0, -1, error_rate=0.08)
    base = run(det, reads)
    o_recs, o_cnt, o_traces, o_rows = oracle_lib.scan(base[0], reads, counts=True, trace=True, rows=True, threads=8)
    assert base[1] == o_recs.tobytes()
    assert np.array_equal(base[3], o_rows) and np.array_equal(base[4], o_cnt)
