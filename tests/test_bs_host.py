"""The bit-sliced BARCODE arithmetic on the CPU (no GPU needed): qcat_amd/csrc/bs_core.h is a pure function of 32-bit
words, so tests/bs_host_check.cpp runs the cell, the counters and both deficit updates of the device kernels
(kernels_bitslice.inc) 32 alignments at a time in the order a unit walks them -- round 5: BOTH contexts of a target
shared, the trailing one through the reversed DP -- against the oracle's scalar DP on the original orientation of region
and target: every target family of the shipped kits in the shape the generator gives it (both directions of bs_rev),
and custom shapes that cover 0 .. 12 trailing columns, 0 / 4 / 8 / 11 shared ones and both directions."""
import os
import random
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def host_check(tmp_path_factory):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    exe = str(tmp_path_factory.mktemp("bs") / "bs_host_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-w", "-I", os.path.join(ROOT, "qcat_amd", "csrc"),
                           os.path.join(ROOT, "tests", "bs_host_check.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "oracle"), "-lqcat_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    return exe


def shipped_families():
    import gen_static_kernels as g
    fams = g.collect()[0]
    lines = []
    for (up, dn, m), targets in fams.items():
        shape = g.bs_shape(len(up), len(dn), m)
        if shape is None:
            continue
        rev, pre, own, post = shape
        lines.append("%d %d %d %s" % (rev, pre, post, " ".join(targets)))
    return lines


def custom_families(seed):
    import gen_static_kernels as g
    rng = random.Random(seed)
    lines = []
    for uplen, dnlen in [(0, 0), (0, 5), (7, 0), (4, 4), (1, 11), (11, 1), (12, 12), (3, 9), (9, 3), (8, 12), (6, 11), (11, 11)]:
        up = "".join(rng.choice("ACGT") for _ in range(uplen))
        dn = "".join(rng.choice("ACGT") for _ in range(dnlen))
        blen = rng.choice([20, 24, 30])
        targets = [up + "".join(rng.choice("ACGT") for _ in range(blen)) + dn for _ in range(6)]
        shape = g.bs_shape(uplen, dnlen, len(targets[0]))
        assert shape is not None
        rev, pre, own, post = shape
        lines.append("%d %d %d %s" % (rev, pre, post, " ".join(targets)))
        if post:                                   # the same family with fewer trailing columns taken out (any split column is exact)
            lines.append("%d %d %d %s" % (rev, pre, post // 2, " ".join(targets)))
    return lines


@pytest.mark.parametrize("seed", [1, 20260930])
def test_bit_sliced_barcode_arithmetic_equals_the_oracle_dp(host_check, tmp_path, seed):
    lines = shipped_families()
    assert len(lines) >= 10 and any(l.startswith("1 ") for l in lines) and any(l.startswith("0 ") for l in lines)
    assert all(int(l.split()[2]) >= 6 for l in lines), "every shipped family has its trailing context taken out of the rows"
    lines += custom_families(seed)
    fam = tmp_path / "families.txt"
    fam.write_text("\n".join(lines) + "\n")
    p = subprocess.run([host_check, str(fam), str(seed), "3"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    out = p.stdout.decode()
    assert p.returncode == 0, out[-3000:] + p.stderr.decode()[-3000:]
    rows = [l for l in out.splitlines() if l.startswith("family")]
    assert len(rows) == len(lines) and all(l.endswith("split: 0 mismatches, unsplit: 0 mismatches, front-padded: 0 mismatches") for l in rows), out[-3000:]
