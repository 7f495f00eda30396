#!/usr/bin/env python3
"""Turn the last exact network of a tools/lut3_search_adapter.cpp log into the body of abs_cell_letter / abs_cell_n
(abs_core.h): usage: tools/lut3_to_cpp.py <log> <letter|n>"""
import re
import sys

log, kind = sys.argv[1], sys.argv[2]
nets, cur = [], None
for line in open(log):
    if line.startswith("EXACT="):
        cur = {"head": line.strip(), "nodes": []}
        if line.startswith("EXACT=1"):
            nets.append(cur)
    else:
        m = re.match(r"\s+s(\d+) = LUT\[0x([0-9a-f]{2})\]\(s(\d+), s(\d+), s(\d+)\)", line)
        if m and cur is not None:
            cur["nodes"].append(tuple([int(m.group(1)), int(m.group(2), 16)] + [int(m.group(i)) for i in (3, 4, 5)]))
net = nets[-1]
outs = [int(x[1:]) for x in net["head"].split("=")[-1].split()]
nin = 9 if kind == "letter" else 8
names = {0: "a[3]", 1: "a[2]", 2: "a[1]", 3: "a[0]", 4: "b[3]", 5: "b[2]", 6: "b[1]", 7: "b[0]", 8: "neq"}
print("    // %s" % net["head"])
for sid, tab, x, y, z in net["nodes"]:
    names[sid] = "s%d" % sid
    print("    const u32 s%d = ABS_LUT(%s, %s, %s, 0x%02x);" % (sid, names[x], names[y], names[z], tab))
print("    a[3] = %s; a[2] = %s; a[1] = %s; a[0] = %s;" % tuple(names[o] for o in outs[:4]))
print("    b[3] = %s; b[2] = %s; b[1] = %s; b[0] = %s;" % tuple(names[o] for o in outs[4:]))
