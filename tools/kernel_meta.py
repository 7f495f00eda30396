#!/usr/bin/env python3
"""VGPRs / scratch / LDS of the kernels inside a built object or library (the gfx950 code object's metadata notes):
usage: tools/kernel_meta.py <file.o|.so> [name filter]"""
import re
import subprocess
import sys
import tempfile
import os

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
tmp = tempfile.mkdtemp()
# unbundle the device code object(s)
targets = []
if True:
    # the fat binary sits in the .hip_fatbin section of an object or a shared library
    fb = os.path.join(tmp, "fatbin")
    subprocess.check_call(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fb])
    path = fb
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o", "--input=" + path], stdout=subprocess.PIPE)
    targets = [t for t in out.stdout.decode().split() if "gfx950" in t]
for i, t in enumerate(targets):
    co = os.path.join(tmp, "co%d" % i)
    subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + path, "--targets=" + t, "--output=" + co])
    notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], stdout=subprocess.PIPE).stdout.decode()
    for blk in re.split(r"\n  - \.agpr_count:", notes)[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name:
            continue
        dem = subprocess.run(["c++filt", name.group(1)], stdout=subprocess.PIPE).stdout.decode().strip()
        if flt and flt not in dem:
            continue
        def g(k):
            r = re.search(r"\.%s:\s+(\d+)" % k, blk)
            return int(r.group(1)) if r else -1
        print("%-70s vgpr %3d sgpr %3d scratch %4d lds %6d spill_v %d" % (dem.split("(")[0][-70:], g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"), g("vgpr_spill_count")))
