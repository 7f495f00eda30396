"""bench.py's replayed counter figures (valu_issue, roofline.traffic): no clamp, an impossible fraction or a stale
counter file prints null with the reason ("What's weak" 6)."""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _bench():
    import importlib
    return importlib.import_module("bench")


def _args(workload="config3", reads=1000000):
    return types.SimpleNamespace(workload=workload, reads=reads)


def _pmc(tmp_path, insts, mark_ms=None):
    rec = {"reads_per_launch": 1000000, "clock_ghz": 2.3, "commit": "abc1234",
           "marks": {"k_barcode_bitslice": {"insts_valu": insts, "kernels": ["k"]}}}
    if mark_ms is not None:
        rec["mark_ms"] = {"k_barcode_bitslice": mark_ms}
    p = tmp_path / "r99_pmc.json"
    p.write_text(json.dumps({"config3": rec}))
    return str(p)


def test_a_plausible_counter_file_gives_the_unclamped_fraction(tmp_path):
    b = _bench()
    # 1.0e9 instructions x 2 cycles / (1024 SIMDs x 2.3 GHz x 1.2 ms) = 0.7077
    out = b.valu_issue(_args(), {"k_barcode_bitslice": 1.2}, path=_pmc(tmp_path, 1000000000, mark_ms=1.25))
    m = out["marks"]["k_barcode_bitslice"]
    assert abs(m["issue_util"] - 2e9 / (1024 * 2.3e9 * 1.2e-3)) < 1e-4 and "reason" not in m
    assert "abc1234" in out["source"]


def test_a_fraction_above_one_is_null_with_a_reason_not_one(tmp_path):
    b = _bench()
    out = b.valu_issue(_args(), {"k_barcode_bitslice": 1.2}, path=_pmc(tmp_path, 5000000000))
    m = out["marks"]["k_barcode_bitslice"]
    assert m["issue_util"] is None and m["issue_util_at_2p4ghz"] is None and "impossible" in m["reason"]


def test_a_mark_that_moved_by_more_than_15_percent_drops_the_replayed_figure(tmp_path):
    b = _bench()
    out = b.valu_issue(_args(), {"k_barcode_bitslice": 1.2}, path=_pmc(tmp_path, 1000000000, mark_ms=1.6))
    m = out["marks"]["k_barcode_bitslice"]
    assert m["issue_util"] is None and m["reason"].startswith("stale") and "abc1234" in m["reason"]
    # the launch size scales the recorded duration: half the reads, half the reference duration
    out = b.valu_issue(_args(reads=500000), {"k_barcode_bitslice": 0.62}, path=_pmc(tmp_path, 1000000000, mark_ms=1.25))
    assert out["marks"]["k_barcode_bitslice"]["issue_util"] is not None


def test_traffic_replay_has_the_same_guard(tmp_path):
    b = _bench()
    p = tmp_path / "r99_traffic.json"
    rec = {"kernel": "k_barcode_bitslice (k_bs_barcode + k_bs_select + k_bs_plan)", "reads_per_launch": 1000000,
           "bytes": 1000, "commit": "abc1234", "mark_ms": 1.0}
    p.write_text(json.dumps({"config3": rec}))
    t, src = b.replayed_traffic(_args(), "k_barcode_bitslice", {"k_barcode_bitslice": 1.05}, path=str(p))
    assert t == 1000 and "abc1234" in src
    t, src = b.replayed_traffic(_args(), "k_barcode_bitslice", {"k_barcode_bitslice": 1.3}, path=str(p))
    assert t is None and "stale" in src
    t, src = b.replayed_traffic(_args(), "k_finalize", {"k_finalize": 1.0}, path=str(p))
    assert t is None and src is None


def test_the_committed_counter_files_parse_and_the_newest_round_is_used():
    b = _bench()
    for suffix in ("pmc.json", "traffic.json"):
        p = b.newest_profile(suffix)
        assert p and os.path.exists(p)
        rounds = sorted(f[:3] for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_" + suffix))
        assert os.path.basename(p).startswith(rounds[-1])
        json.load(open(p))
