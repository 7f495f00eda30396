#!/usr/bin/env python3
"""Per-kernel ISA statistics of the packed kernels (VGPRs, scratch, occupancy, instruction mix).
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only qcat_amd/csrc/qcat_hip.hip -o /tmp/q.s
       python tools/isa_stats.py /tmp/q.s [name-filter]"""
import re
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else "packed"
for m in re.finditer(r'^(_Z\w+):\s*; @\1\n(.*?)^\s*\.end_amdhsa_kernel(.*?)^; Occupancy: \d+', s, re.S | re.M):
    name, body, tail = m.group(1), m.group(2), m.group(3)
    if flt not in name:
        continue
    def g(pat):
        r = re.search(pat, body + tail)
        return r.group(1) if r else "?"
    short = re.sub(r'^_ZN2qk\d+', '', name)[:34]
    print("%-34s vgpr %3s scratch %3s occ %s  perm %4d max3 %4d pkmax %4d pkadd_f16 %4d add_u32 %4d" % (
        short, g(r'; NumVgprs: (\d+)'), g(r'; ScratchSize: (\d+)'), re.search(r"Occupancy: (\d+)", m.group(0)[-20:]).group(1),
        body.count('v_perm_b32'), body.count('v_pk_maximum3_f16'), body.count('v_pk_max_u16'),
        body.count('v_pk_add_f16'), body.count('v_add_u32')))
