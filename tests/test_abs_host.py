"""The bit-sliced ADAPTER arithmetic on the CPU (no GPU needed): qcat_amd/csrc/abs_core.h and the generated column
programs (abs_generated.inc) are pure functions of 32-bit words, so tests/abs_host_check.cpp runs the code the device
kernels run, 32 alignments at a time, against the oracle's scalar DP (score AND end_query under rule R1) -- every plan,
full windows and an odd row count, adapter-carrying / adapter-free / tandem-repeat windows."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module", params=[8, 14], ids=["windows", "interiors"])
def host_check(request, tmp_path_factory):
    """8 row-index planes: the build of abs_kernels.hip (read-end windows); 14: that of abs_mid_kernels.hip (read interiors,
    --detect-middle), whose front-padded form then runs 611 rows"""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    exe = str(tmp_path_factory.mktemp("abs") / "abs_host_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-w", "-DQCAT_ABS_NI=%d" % request.param,
                           "-I", os.path.join(ROOT, "qcat_amd", "csrc"), "-I", os.path.join(ROOT, "tests"),
                           os.path.join(ROOT, "tests", "abs_host_check.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "oracle"), "-lqcat_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    return exe


@pytest.mark.parametrize("rule", [0, 1], ids=["r1-striped", "r1-scalar"])       # include/qcat_hip.h QCAT_R1_*: abs_decide under both end-position rules
@pytest.mark.parametrize("seed", [1, 20260929])
def test_bit_sliced_adapter_arithmetic_equals_the_oracle_dp(host_check, seed, rule):
    p = subprocess.run([host_check, str(seed), "40" if rule == 0 else "12", str(rule)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0, out[-2000:] + p.stderr.decode()[-2000:]
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) >= 1 + 2 * 15 and all(l.endswith(": 0 mismatches") for l in lines), out[-2000:]
    assert sum("front-padded" in l for l in lines) >= 10, out[-2000:]


def test_generated_plans_are_in_sync_with_the_kit_bundle():
    """abs_generated.inc / abs_host_cases.inc are what tools/gen_abs_kernels.py writes for resources/kits.json today,
    and the plans follow the fused / single-template ids of static_generated.inc."""
    import gen_abs_kernels as g
    text, nf, nt = g.render()
    with open(g.OUT) as fh:
        assert fh.read() == text, "run tools/gen_abs_kernels.py"
    with open(g.CASES_OUT) as fh:
        assert fh.read() == g.render.cases, "run tools/gen_abs_kernels.py"
    assert nf >= 3 and nt >= 10
    # every plan keeps both stages inside a wave's register budget and every border in stage 1
    for name, seqs in [("F", s) for s in g.gsk.collect()[2]]:
        ops = g.program(list(seqs))
        p = g.split_point(ops)
        assert not any(o[0] == "border" for o in ops[:p])
