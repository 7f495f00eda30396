"""Randomised configuration sweep on the GPU: random qcatConfig values (scores, linear and affine
gaps, context / extension / window lengths), shipped and custom kits (YAML kit folder, N and X inside
templates or barcodes, 96 x 96 dual), mixed reads -- HIP records, traces and per-barcode rows must
equal the oracle's.  Exercises both device paths (the packed kernels and the general fallback) and
the host-side eligibility checks that choose between them."""
import os
import random

import numpy as np
import pytest
import yaml

import custom_kits
import oracle_lib
import synth
from qcat_amd import adapters, config, native, scanner

pytestmark = pytest.mark.gpu
_ctx = {}


def ctx():
    if "c" not in _ctx:
        _ctx["c"] = native.NativeContext(0)
    return _ctx["c"]


def compare(det, cfg, reads, ends=native.ENDS_BOTH):
    d = det.descriptor(qcat_config=cfg, ends=ends)
    kit = native.NativeKit(d)
    bases, offsets = native.pack_reads(reads)
    cnt = np.zeros(d.n_count_buckets, dtype=np.int64)
    recs, traces, rows = ctx().scan(kit, bases, offsets, counts=cnt, trace=True, rows=True)
    o_recs, o_cnt, o_traces, o_rows = oracle_lib.scan(d, reads, counts=True, trace=True, rows=True, threads=8)
    assert recs.tobytes() == o_recs.tobytes()
    assert np.array_equal(cnt, o_cnt)
    for name in native.TRACE_DTYPE.names:
        assert np.array_equal(traces[name], o_traces[name]), name
    assert np.array_equal(rows, o_rows)


def random_reads(rng, layouts, n, t5, t3):
    reads = synth.synth_batch(n, rng.randrange(1 << 30), layouts, t5, t3, error_rate=rng.choice([0.0, 0.05, 0.12, 0.2]))
    out = []
    for r in reads:
        k = rng.random()
        if k < 0.08:
            r = r[:rng.randrange(0, 320)]
        elif k < 0.14:
            pos = rng.randrange(0, min(len(r), 140))
            r = r[:pos] + "N" * rng.randrange(1, 6) + r[pos:]
        elif k < 0.18:
            r = r.lower()
        elif k < 0.21:
            pos = rng.randrange(0, min(len(r), 140))
            r = r[:pos] + rng.choice("RYKMSWXU*-") + r[pos:]
        out.append(r)
    return out


@pytest.mark.parametrize("seed", range(16))
def test_random_config_shipped_kits(seed):
    rng = random.Random(1000 + seed)
    cfg = config.qcatConfig()
    cfg.match = rng.randint(1, 9)
    cfg.mismatch = -rng.randint(1, 6)
    if rng.random() < 0.3:
        cfg._nmatch = -rng.randint(0, 3)
        cfg.update_matrix()
    g = rng.randint(1, 4)
    cfg.gap_open = g
    cfg.gap_extend = g if rng.random() < 0.75 else rng.randint(1, 4)
    cfg.barcode_context_length = rng.choice([0, 3, 7, 11, 11, 11, 15])
    cfg.extracted_barcode_extension = rng.choice([0, 5, 11, 11, 11, 20])
    cfg.max_align_length = rng.choice([40, 64, 100, 150, 150, 150])
    mode, kit, t5, t3 = rng.choice([("epi2me", "PBC096", 1, 0), ("epi2me", "NBD103/NBD104", 1, 0),
                                    ("epi2me", "PBK004/LWB001", 1, 0), ("epi2me", "RBK004", 0, -1),
                                    ("epi2me", "RAB204", 1, 0), ("epi2me", "VMK001", 0, -1),
                                    ("dual", None, 1, 0), ("epi2me", None, 3, 2), ("epi2me", "DUAL", 1, 0)])
    det = scanner.factory(mode=mode, kit=kit, min_quality=rng.choice([None, 40, 58.5, 75]))
    try:
        det.descriptor(qcat_config=cfg)
        native.NativeKit(det.descriptor(qcat_config=cfg))
    except RuntimeError as e:       # e.g. non-positive normalisation denominator: rejected loudly
        assert "denominator" in str(e) or "target length" in str(e)
        return
    reads = random_reads(rng, det.layouts, 150, t5, t3)
    compare(det, cfg, reads, ends=rng.choice([native.ENDS_BOTH, native.ENDS_BOTH, native.ENDS_5P]))


_write_kit, _random_barcodes = custom_kits.write_kit, custom_kits.random_barcodes


def test_custom_kit_folder_with_n_and_x(tmp_path):
    """Custom kits from a YAML folder: an X in the template and an N inside a barcode are outside
    the packed path's tables -> generic device kernel; a plain custom kit stays on the packed path."""
    rng = random.Random(7)
    folder = str(tmp_path)
    bcs = _random_barcodes(rng, 10)
    _write_kit(folder, "A_plain_5p", "CUSTOM", "GGTGCTG" + "N" * 24 + "TTAACCTTTCTGTTGGTGCTGATATTGC", bcs, trim_offset=5)
    _write_kit(folder, "A_plain_3p", "CUSTOM", "GGTGCTG" + "N" * 24 + "TTAACCTACTTGCCTGTCGCTCTATCTTC", bcs, trim_offset=5)
    _write_kit(folder, "B_x", "CUSTOMX", "GGTGXTG" + "N" * 24 + "TTAACCTTTCTGTTGGTGCTGATATTGC", bcs)
    nb = list(bcs)
    nb[3] = nb[3][:10] + "N" + nb[3][11:]
    _write_kit(folder, "C_n", "CUSTOMN", "CCGTGAC" + "N" * 24 + "AGAGTTTGATCATGGCTCAG", nb)
    for kit in ("CUSTOM", "CUSTOMX", "CUSTOMN"):
        det = scanner.factory(kit=kit, kit_folder=folder)
        assert len(det.layouts) == (2 if kit == "CUSTOM" else 1)
        t5, t3 = (1, 0) if kit == "CUSTOM" else (0, -1)
        reads = random_reads(rng, det.layouts, 200, t5, t3)
        compare(det, config.qcatConfig(), reads)


def test_dual_96x96_custom_kit(tmp_path):
    """BASELINE config 5 variant: a custom dual kit whose first set also has 96 barcodes
    (9 217 barcode buckets)."""
    rng = random.Random(96)
    folder = custom_kits.write_dual_96x96(str(tmp_path), seed=96)
    det = scanner.factory(mode="dual", kit_folder=folder)
    assert [len(l.barcode_set_1) for l in det.layouts] == [96, 96]
    d = det.descriptor()
    assert d.n_count_buckets == 96 * 96 + 1 + 1 + 1 + 1
    reads = random_reads(rng, det.layouts, 300, 1, 0)
    compare(det, config.qcatConfig(), reads)
