"""Full BASELINE sizes on the GPU, checked through size-independent properties:
  * the count vector equals the histogram recomputed from the records (checksum of checksums)
    and sums to the number of reads;
  * idempotence: a second scan of the resident batch returns byte-identical records;
  * shard invariance: a contiguous slice scanned as its own batch equals the slice of the
    full-batch records (what multi-GPU sharding relies on);
  * a random sample of reads, regenerated with the host twin of the generator, goes through the
    CPU oracle and must match record for record;
  * error-free synthetic reads are assigned the barcode they were built from.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
import synth
from qcat_amd import native, scanner

pytestmark = pytest.mark.gpu


class Resident(object):
    def __init__(self, det, ends, n, seed, e, t5=1, t3=0, jit=None):
        self.det, self.n = det, n
        self.desc = det.descriptor(ends=ends)
        self.hip = native.HipLibrary.get()
        self.lib = self.hip.lib
        self.kit = native.NativeKit(self.desc, jit=jit)
        self.ctx = native.NativeContext(0)
        self.sp = native.SynthParams(seed=seed, n_reads=n, insert_len=600, lead_min=5, lead_max=40,
                                     error_rate=e, no_adapter_fraction=0.05, tpl_5p=t5, tpl_3p=t3)
        self.batch = C.c_void_p()
        self.hip.check(self.lib.qcat_batch_synthesize(self.ctx.handle, self.kit.handle, C.byref(self.sp), C.byref(self.batch)))

    def scan(self):
        self.hip.check(self.lib.qcat_scan_resident(self.ctx.handle, self.kit.handle, self.batch))
        recs = np.zeros(self.n, dtype=native.RESULT_DTYPE)
        self.hip.check(self.lib.qcat_ctx_fetch_results(self.ctx.handle, recs.ctypes.data, self.n))
        cnt = np.zeros(self.desc.n_count_buckets, dtype=np.int64)
        self.hip.check(self.lib.qcat_ctx_fetch_counts(self.ctx.handle, cnt.ctypes.data, len(cnt)))
        return recs, cnt

    def host_reads(self, indices):
        buf = np.zeros(4096, dtype=np.uint8)
        out = []
        for i in indices:
            ln = self.lib.qcat_synth_read(self.kit.handle, C.byref(self.sp), int(i), buf.ctypes.data, buf.size)
            out.append(buf[:ln].tobytes().decode())
        return out

    def close(self):
        self.lib.qcat_batch_destroy(self.batch)


def histogram_from_records(desc, layouts, recs):
    nb, nk = len(desc.slot_ids), len(desc.kit_names)
    cnt = np.zeros(desc.n_count_buckets, dtype=np.int64)
    slot_tables = [np.array([desc.id_slots[b.id] for b in lay.barcode_set_1]) for lay in layouts]
    kit_slot = np.array([desc.kit_slots[lay.kit] for lay in layouts])
    has = recs["barcode_idx"] >= 0
    slots = np.full(len(recs), nb, dtype=np.int64)
    for t in range(len(layouts)):
        m = has & (recs["adapter_idx"] == t)
        slots[m] = slot_tables[t][recs["barcode_idx"][m]]
    cnt[:nb + 1] = np.bincount(slots, minlength=nb + 1)
    ks = np.where(recs["adapter_idx"] >= 0, kit_slot[np.maximum(recs["adapter_idx"], 0)], nk)
    cnt[nb + 1:nb + 1 + nk + 1] = np.bincount(ks, minlength=nk + 1)       # (last bucket: [skipped], 0 without a filter)
    return cnt


def dual_histogram_from_records(desc, layouts, recs):
    """dual kits: barcode bucket = slot(set 1) * n_slots + slot(set 2) (include/qcat_hip.h)"""
    nb, nk = len(desc.slot_ids), len(desc.kit_names)
    cnt = np.zeros(desc.n_count_buckets, dtype=np.int64)
    has = recs["barcode_idx"] >= 0
    slots = np.full(len(recs), nb * nb, dtype=np.int64)
    for t, lay in enumerate(layouts):
        s1 = np.array([desc.id_slots[b.id] for b in lay.barcode_set_1])
        s2 = np.array([desc.id_slots[b.id] for b in lay.barcode_set_2])
        m = has & (recs["adapter_idx"] == t)
        slots[m] = s1[recs["barcode_idx"][m]] * nb + s2[recs["barcode2_idx"][m]]
    cnt[:nb * nb + 1] = np.bincount(slots, minlength=nb * nb + 1)
    kit_slot = np.array([desc.kit_slots[lay.kit] for lay in layouts])
    ks = np.where(recs["adapter_idx"] >= 0, kit_slot[np.maximum(recs["adapter_idx"], 0)], nk)
    cnt[nb * nb + 1:nb * nb + 1 + nk + 1] = np.bincount(ks, minlength=nk + 1)
    return cnt


def check_sample_against_oracle(r, recs, k, rng):
    idx = np.sort(rng.choice(r.n, size=k, replace=False))
    want = oracle_lib.scan(r.desc, r.host_reads(idx), threads=16)
    assert recs[idx].tobytes() == want.tobytes()


def check_all_records_across_device_paths(r, recs, cnt, monkeypatch, n_generic=200000):
    """Full-coverage self-check at size: the default path (bit-sliced barcode kernels + whatever adapter kernels the
    batch takes) against two INDEPENDENT device formulations of the same scan -- every record of the batch with the
    bit-sliced kernels switched off (packed binary16 DP: another algorithm, another data layout), and the first
    `n_generic` reads on the general int32 kernel (one thread per read end, affine DP in LDS).  The oracle pins a
    sample; this pins every read of the batch to a second and third implementation."""
    monkeypatch.setenv("QCAT_HIP_NO_BITSLICE", "1")
    monkeypatch.setenv("QCAT_HIP_NO_ADAPTER_BITSLICE", "1")
    recs_b16, cnt_b16 = r.scan()
    monkeypatch.delenv("QCAT_HIP_NO_BITSLICE")
    monkeypatch.delenv("QCAT_HIP_NO_ADAPTER_BITSLICE")
    diff = np.flatnonzero(recs_b16 != recs)
    assert diff.size == 0, "default path vs binary16 kernels: %d records differ, first at read %d" % (diff.size, diff[0])
    assert np.array_equal(cnt, cnt_b16)
    # the generator is stateless per read index: a batch of the first n reads of the same parameters is the head of r's
    n = min(n_generic, r.n)
    sp = native.SynthParams.from_buffer_copy(r.sp)
    sp.n_reads = n
    monkeypatch.setenv("QCAT_HIP_FORCE_GENERIC", "1")
    gctx = native.NativeContext(0)
    monkeypatch.delenv("QCAT_HIP_FORCE_GENERIC")
    head = C.c_void_p()
    r.hip.check(r.lib.qcat_batch_synthesize(gctx.handle, r.kit.handle, C.byref(sp), C.byref(head)))
    try:
        r.hip.check(r.lib.qcat_scan_resident(gctx.handle, r.kit.handle, head))
        recs_gen = np.zeros(n, dtype=native.RESULT_DTYPE)
        r.hip.check(r.lib.qcat_ctx_fetch_results(gctx.handle, recs_gen.ctypes.data, n))
    finally:
        r.lib.qcat_batch_destroy(head)
    diff = np.flatnonzero(recs_gen != recs[:n])
    assert diff.size == 0, "default path vs general int32 kernel: %d records differ, first at read %d" % (diff.size, diff[0])


def kernels_of_last_scan(r):
    names = (C.c_char_p * 16)()
    ms = (C.c_float * 16)()
    return [names[i].decode() for i in range(r.lib.qcat_ctx_last_timing(r.ctx.handle, names, ms, 16))]


def test_config2_one_million_reads_5p_only(monkeypatch):
    det = scanner.factory(kit="NBD103/NBD104")
    r = Resident(det, native.ENDS_5P, 1000000, 20260929, 0.08)
    try:
        r.hip.check(r.lib.qcat_ctx_set_timing(r.ctx.handle, 1))
        recs, cnt = r.scan()
        # 488 adapter tiles: the medium-batch form of the bit-sliced adapter scan (four-stage plans), 1 M jobs of 12-barcode
        # sets: the bit-sliced barcode kernels
        ran = kernels_of_last_scan(r)
        assert "k_adapter_bitslice" in ran and "k_barcode_bitslice" in ran, ran
        r.hip.check(r.lib.qcat_ctx_set_timing(r.ctx.handle, 0))
        assert cnt[:13].sum() == r.n and cnt[13:15].sum() == r.n and cnt[15] == 0
        assert np.array_equal(cnt, histogram_from_records(r.desc, det.layouts, recs))
        check_all_records_across_device_paths(r, recs, cnt, monkeypatch)
        recs2, cnt2 = r.scan()
        assert recs2.tobytes() == recs.tobytes() and np.array_equal(cnt, cnt2)          # idempotence
        check_sample_against_oracle(r, recs, 3000, np.random.default_rng(1))
        # shard invariance: reads [300000, 300000 + 50000) as their own batch
        nb = C.c_uint64(); nr = C.c_uint32()
        r.hip.check(r.lib.qcat_batch_info(r.batch, C.byref(nr), C.byref(nb)))
        bases = np.zeros(nb.value, dtype=np.uint8); offs = np.zeros(r.n + 1, dtype=np.uint64)
        r.hip.check(r.lib.qcat_batch_download(r.ctx.handle, r.batch, bases.ctypes.data, offs.ctypes.data))
        a, b = 300000, 350000
        sub_b = np.ascontiguousarray(bases[int(offs[a]):int(offs[b])])
        sub_o = np.ascontiguousarray(offs[a:b + 1] - offs[a])
        sub = r.ctx.scan(r.kit, sub_b, sub_o)
        assert sub.tobytes() == recs[a:b].tobytes()
    finally:
        r.close()


def test_medium_batch_takes_the_bit_sliced_barcode_kernels_by_default(monkeypatch):
    """120 k PBC096 reads = 240 k jobs: above the point where the bit-sliced barcode kernels pay for a 96-barcode set
    (round 6: 40 000 + 1 900 000 / 96 jobs), far below a super-tile per CU -- the batch size of the host pipeline's chunks.  The
    default path must take them, and every record must equal the binary16 and the general kernels' (+ an oracle sample)."""
    det = scanner.factory(kit="PBC096")
    r = Resident(det, native.ENDS_BOTH, 120000, 20261001, 0.08)
    try:
        r.hip.check(r.lib.qcat_ctx_set_timing(r.ctx.handle, 1))
        recs, cnt = r.scan()
        assert "k_barcode_bitslice" in kernels_of_last_scan(r)
        r.hip.check(r.lib.qcat_ctx_set_timing(r.ctx.handle, 0))
        assert np.array_equal(cnt, histogram_from_records(r.desc, det.layouts, recs))
        check_sample_against_oracle(r, recs, 3000, np.random.default_rng(7))
        check_all_records_across_device_paths(r, recs, cnt, monkeypatch, n_generic=120000)
    finally:
        r.close()


def test_config3_ten_million_reads_both_ends(monkeypatch):
    det = scanner.factory(kit="PBC096")
    r = Resident(det, native.ENDS_BOTH, 10000000, 20260930, 0.08)
    try:
        recs, cnt = r.scan()
        assert cnt[:97].sum() == r.n
        assert np.array_equal(cnt, histogram_from_records(r.desc, det.layouts, recs))
        check_sample_against_oracle(r, recs, 50000, np.random.default_rng(2))
        check_all_records_across_device_paths(r, recs, cnt, monkeypatch)
        # trims are always ordered and inside the read (qcat/test/test_barcode.py:599-603 style)
        assert (recs["trim5p"] <= recs["trim3p"]).all() and (recs["trim5p"] >= 0).all()
        called = recs["barcode_idx"] >= 0
        assert 0.80 < called.mean() < 0.96             # 5 % adapter-free reads, 8 % errors
        assert set(np.unique(recs["exit_status"])) <= {0, 1, 1002}
    finally:
        r.close()


def test_config4_shard_twelve_and_a_half_million_reads(monkeypatch):
    """BASELINE config 4 = 100 M PBC096 reads over 8 GPUs: ONE rank's shard (12.5 M reads, the seed rank 3 of
    bench.py --gpus 8 uses) at full size, every record cross-checked over the device paths, 50 k through the oracle."""
    det = scanner.factory(kit="PBC096")
    r = Resident(det, native.ENDS_BOTH, 12500000, 20260928 + 3 + 1000003 * 3, 0.08)
    try:
        recs, cnt = r.scan()
        assert cnt[:97].sum() == r.n and cnt[97:100].sum() == r.n
        assert np.array_equal(cnt, histogram_from_records(r.desc, det.layouts, recs))
        check_sample_against_oracle(r, recs, 50000, np.random.default_rng(4))
        check_all_records_across_device_paths(r, recs, cnt, monkeypatch)
        assert (recs["trim5p"] <= recs["trim3p"]).all() and (recs["trim5p"] >= 0).all()
    finally:
        r.close()


@pytest.mark.parametrize("n_shards", [2, 8])
def test_contiguous_shards_on_one_device_equal_the_whole_batch(n_shards):
    """The part of configs 4 / 5 that needs no RCCL: `parallel.shard_range` cuts a batch into contiguous
    shards, every shard is scanned through a context of its own (as a rank would, here all on device 0), the shard
    count vectors are summed on the host -- what the all-reduce does -- and counts and read-ordered records must equal
    the oracle's scan of the WHOLE batch."""
    from qcat_amd import parallel
    det = scanner.factory(kit="PBC096")
    n = 30011                                                    # (not a multiple of the shard count)
    reads = synth.synth_batch(n, 20260931, det.layouts, 1, 0, error_rate=0.08) + ["", "A", "N" * 200]
    desc = det.descriptor(ends=native.ENDS_BOTH)
    kit = native.NativeKit(desc)
    want, want_cnt = oracle_lib.scan(desc, reads, threads=16, counts=True)
    total = np.zeros(desc.n_count_buckets, dtype=np.int64)
    parts = []
    for rank in range(n_shards):
        lo, hi = parallel.shard_range(len(reads), rank, n_shards)
        ctx = native.NativeContext(0)
        bases, offsets = native.pack_reads(reads[lo:hi])
        cnt = np.zeros(desc.n_count_buckets, dtype=np.int64)
        parts.append(ctx.scan(kit, bases, offsets, counts=cnt))
        total += cnt
    got = np.concatenate(parts)
    assert got.tobytes() == want.tobytes()
    assert np.array_equal(total, want_cnt) and total[:97].sum() == len(reads)


def test_error_free_reads_recover_their_barcode():
    det = scanner.factory(kit="PBC096")
    n = 200000
    r = Resident(det, native.ENDS_BOTH, n, 424242, 0.0)
    try:
        recs, cnt = r.scan()
        idx = np.arange(0, n, 97)
        thr_none = synth.rate_threshold(0.05)
        for i in idx:
            rng = synth.SplitMix64(424242, int(i))
            bare = rng.u24() < thr_none
            rng.below(36); rng.below(36)
            b = rng.below(1 << 16) % 96
            if bare:
                assert recs[i]["barcode_idx"] == -1 or recs[i]["raw_score"] * 100.0 / recs[i]["score_den"] < 100.0
            else:
                assert recs[i]["barcode_idx"] == b and recs[i]["exit_status"] == 0
                assert recs[i]["raw_score"] == recs[i]["score_den"]        # perfect score 100.0
    finally:
        r.close()


@pytest.mark.parametrize("which", ["shipped 24x96", "custom 96x96"])
def test_config5_dual_one_million_reads(which, tmp_path):
    """BASELINE config 5 at size on one GPU: the shipped DUAL kit (24 x 96 pairs, 14 403 buckets) and a
    custom 96 x 96 kit (run-time generated kernels, 9 217 + ... buckets); 1 M reads each."""
    import custom_kits
    if which.startswith("custom"):
        det = scanner.factory(mode="dual", kit_folder=custom_kits.write_dual_96x96(str(tmp_path)))
        n1 = 96
    else:
        det = scanner.factory(mode="dual")
        n1 = 24
    assert [len(l.barcode_set_1) for l in det.layouts] == [n1, n1] and [len(l.barcode_set_2) for l in det.layouts] == [96, 96]
    r = Resident(det, native.ENDS_BOTH, 1000000, 20260932, 0.08, jit=True)
    try:
        info = r.kit.describe()
        assert info["packed"] == 1 and info["n_static_groups"] == info["n_groups"] == 4      # static-letter kernels either way
        assert info["bitslice_groups"] == 4 * 0x10001           # ... and bit-sliced ones with the letters compiled in
        recs, cnt = r.scan()
        nb = len(r.desc.slot_ids)
        assert cnt[:nb * nb + 1].sum() == r.n and cnt[nb * nb + 1:-1].sum() == r.n and cnt[-1] == 0            # histogram = records
        assert np.array_equal(cnt, dual_histogram_from_records(r.desc, det.layouts, recs))
        recs2, cnt2 = r.scan()
        assert recs2.tobytes() == recs.tobytes() and np.array_equal(cnt, cnt2)              # idempotence
        check_sample_against_oracle(r, recs, 1200, np.random.default_rng(5))
        called = recs["barcode_idx"] >= 0
        assert (recs["barcode2_idx"][called] >= 0).all() and (recs["barcode2_idx"][~called] == -1).all()
        assert 0.5 < called.mean() < 0.96
        assert set(np.unique(recs["exit_status"])) <= {0, 1, 1002}
    finally:
        r.close()


def test_detect_middle_one_million_reads_across_interior_paths(monkeypatch):
    """--detect-middle at the size of its bench workload: every record of the batch is the same with the interior adapter scan
    bit-sliced on one wave per tile (the default at this size, kernels_abs_mid.inc), on the two-wave pipeline and on the
    binary16 kernel (k_adapter_middle: another algorithm, another data layout); the tiles really ran bit-sliced; a sample
    goes through the oracle (detect_barcode's interior scan, qcat/scanner_base.py:479-519, :593-595)."""
    det = scanner.factory(kit="NBD103/NBD104", scan_middle_adapter=True)
    r = Resident(det, native.ENDS_BOTH, 1000000, 20260930, 0.08)
    try:
        assert r.desc.scan_middle
        recs, cnt = r.scan()
        tiles = (C.c_uint32 * 4)()
        r.hip.check(r.lib.qcat_ctx_middle_bitslice_tiles(r.ctx.handle, tiles))
        # (the slots have room for every length class of every kit slot: the big tiles beyond the last M-end are empty)
        assert 800 <= tiles[0] <= tiles[1] and tiles[2] <= tiles[3] // 50, list(tiles)
        assert cnt[:13].sum() == r.n
        for name, value in (("QCAT_HIP_MIDDLE_ABS_ONE_WAVE", "0"), ("QCAT_HIP_MIDDLE_NO_ABS", "1")):
            monkeypatch.setenv(name, value)
            other, cnt_o = r.scan()
            r.hip.check(r.lib.qcat_ctx_middle_bitslice_tiles(r.ctx.handle, tiles))
            monkeypatch.delenv(name)
            assert (tiles[0] == 0) == (name == "QCAT_HIP_MIDDLE_NO_ABS"), (name, list(tiles))
            diff = np.flatnonzero(other != recs)
            assert diff.size == 0, "%s=%s: %d records differ, first at read %d" % (name, value, diff.size, diff[0])
            assert np.array_equal(cnt, cnt_o)
        check_sample_against_oracle(r, recs, 2000, np.random.default_rng(5))
    finally:
        r.close()


def test_detect_middle_two_hundred_thousand_reads_an_eighth_of_them_chimeras(monkeypatch):
    """The synthetic reads of the 1 M-read test above are single molecules: almost none of them reaches exit status 997 (an
    adapter in the read's INTERIOR, qcat/scanner_base.py:593-595), which the small golden and fuzz cases carry.  Here every
    eighth read is two reads joined -- the second one's 5' adapter and barcode sit in the middle of the first one's insert side --
    so 25 000 interiors hold an adapter and some 2 800 reads come out as 997: every record the same on the three interior paths
    (bit-sliced on one wave per tile, the two-wave pipeline, the binary16 kernel); 1000 of the 997s, 1000 chimeras and 1000 single
    reads through the oracle."""
    det = scanner.factory(kit="NBD103/NBD104", scan_middle_adapter=True)
    n = 200000
    r = Resident(det, native.ENDS_BOTH, n + n // 8, 20260931, 0.08)
    try:
        single = r.host_reads(range(n + n // 8))
    finally:
        r.close()
    reads = [single[i] + single[n + i // 8] if i % 8 == 0 else single[i] for i in range(n)]
    desc = det.descriptor(ends=native.ENDS_BOTH)
    assert desc.scan_middle
    kit = native.NativeKit(desc)
    bases, offsets = native.pack_reads(reads)
    out = {}
    for name, value in ((None, None), ("QCAT_HIP_MIDDLE_ABS_ONE_WAVE", "0"), ("QCAT_HIP_MIDDLE_NO_ABS", "1")):
        if name:
            monkeypatch.setenv(name, value)
        cnt = np.zeros(desc.n_count_buckets, dtype=np.int64)
        out[name] = (native.NativeContext(0).scan(kit, bases, offsets, counts=cnt), cnt)
        if name:
            monkeypatch.delenv(name)
    recs, cnt = out[None]
    for name in list(out)[1:]:
        diff = np.flatnonzero(out[name][0] != recs)
        assert diff.size == 0, "%s: %d records differ, first at read %d" % (name, diff.size, diff[0])
        assert np.array_equal(out[name][1], cnt)
    is_997 = recs["exit_status"] == 997
    n_997 = int(np.count_nonzero(is_997))
    # (the interior scan wants the called kit's adapter AND a barcode at middle_min_score behind it: one chimera in nine passes)
    assert n_997 >= n // 100, n_997
    assert np.count_nonzero(is_997[np.arange(n) % 8 != 0]) < n // 200
    rng = np.random.default_rng(11)
    idx = np.unique(np.concatenate([rng.choice(np.flatnonzero(is_997), size=1000, replace=False),
                                    rng.choice(n // 8, size=1000, replace=False) * 8,
                                    rng.choice(n // 8, size=1000, replace=False) * 8 + 1 + rng.integers(0, 7, size=1000)]))
    want = oracle_lib.scan(desc, [reads[i] for i in idx], threads=16)
    assert recs[idx].tobytes() == want.tobytes()
