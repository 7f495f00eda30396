#!/usr/bin/env python3
"""Generate qcat_amd/csrc/static_generated.inc: barcode column chains with compile-time target
letters for every barcode target of the built-in kits (see kernels_static.inc).

A *target* is upstream context + barcode + downstream context exactly as the scanners build it
(qcat_amd.layout.AdapterLayout, mirroring qcat/layout.py:191-238) with the default
barcode_context_length.  Targets that share both flanks form one *family* = one kernel; a kit group
(template, set) can use the kernel when all of its targets are cases of the same family, which the
library checks at kit creation through the registry (FNV-1a 64 of the target codes -> kernel, case).

Run from the repo root:  python tools/gen_static_kernels.py
"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from qcat_amd import config as qconfig          # noqa: E402
from qcat_amd import scanner                    # noqa: E402

OUT = os.path.join(ROOT, "qcat_amd", "csrc", "static_generated.inc")
QUAD_MIN_TARGETS = 48       # families this large also get four-target chains (the big sets dominate the run time)
BS_MIN_TARGETS = 12         # families this large get bit-sliced row loops with the letters compiled in (kernels_bitslice.inc)
BS_C_MIN, BS_C_MAX = 20, 48 # kit.h
BS_POSTS = (11, 8, 7, 6, 4) # trailing columns the reversed DP takes (kit_prepare.inc, the instantiations of kernels_bitslice.inc)
BS_PARTS = 6                # translation units the bit-sliced static-letter kernels are split over (__graft_entry__.build)
BS_OUT = os.path.join(ROOT, "qcat_amd", "csrc", "bs_static_generated.inc")
CODE = {"A": 0, "T": 1, "G": 2, "C": 3}
ACODE = {"A": 0, "T": 1, "G": 2, "C": 3, "N": 4}        # adapter templates also hold barcode placeholders


def fnv1a64(codes):
    h = 0xCBF29CE484222325
    for c in codes:
        h ^= c
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def chain(letters, code):
    """column chain in chunks of four.  Every read of the previous row happens before the chunk's
    first write and the diagonal term of the NEXT chunk's first column is formed inside this chunk,
    so h[] is updated in place: no old h[j] outlives its new value, no register copies at the loop
    back edge."""
    e = ["E[%d]" % code[c] for c in letters]
    out = ["QS_BEGIN(%s)" % e[0]]
    n = len(letters)
    for j in range(0, n, 4):
        k = min(4, n - j)
        inner = e[j + 1:j + k]
        if j + k < n:
            out.append("QS_CHUNK4(%d, %s)" % (j + 1, ", ".join(inner + [e[j + k]])))
        else:
            out.append("QS_LAST%d(%d%s)" % (k, j + 1, "".join(", " + x for x in inner)))
    return " ".join(out)


def bs_shape(uplen, downlen, m):
    """(reversed, shared columns, own columns, trailing columns) of a target family on the bit-sliced kernels, or None: the
    rule of kit_prepare.inc (the longer context leads; 11 / 8 / 4 / 0 of its columns are shared; round 5: the other context's
    columns -- 11 / 8 / 7 / 6 / 4 / 0 of them, as long as BS_C_MIN own columns remain -- are computed once per super-tile as well, by
    the reversed DP of bs_core.h)"""
    rev = downlen > uplen
    lead, trail = (downlen, uplen) if rev else (uplen, downlen)
    pre = 11 if lead >= 11 else (8 if lead >= 8 else (4 if lead >= 4 else 0))
    post = next((q for q in BS_POSTS if q <= trail and m - pre - q >= BS_C_MIN), 0)
    own = m - pre - post
    if not (BS_C_MIN <= own <= BS_C_MAX and m <= 64):
        return None
    return rev, pre, own, post


def bs_words(target, rev, pre, code, own=None):
    """letter bit words of the own columns in the order the kernel walks them (bit j = own column j)"""
    t = target[::-1] if rev else target
    w1 = w0 = 0
    for j, ch in enumerate(t[pre:pre + own] if own is not None else t[pre:]):
        c = code[ch]
        w1 |= ((c >> 1) & 1) << j
        w0 |= (c & 1) << j
    return w1, w0


def bs_trailing_words(target, rev, post, code):
    """letter bit words of the trailing context's columns in the order the REVERSED DP walks them (bit j = its column j =
    the target's last column but j, in walk order)"""
    t = target[::-1] if rev else target
    w1 = w0 = 0
    for j in range(post):
        c = code[t[len(t) - 1 - j]]
        w1 |= ((c >> 1) & 1) << j
        w0 |= (c & 1) << j
    return w1, w0


def bs_shared_words(target, rev, pre, code):
    """letter bit words of the shared (leading context) columns: bit j = shared column j"""
    t = target[::-1] if rev else target
    w1 = w0 = 0
    for j, ch in enumerate(t[:pre]):
        c = code[ch]
        w1 |= ((c >> 1) & 1) << j
        w0 |= (c & 1) << j
    return w1, w0


def pair_up(targets, flank):
    """greedy pairing by longest common prefix: [(ta, tb, shared columns)]; a leftover target is paired
    with itself.  Two targets that run in one row pass share their common prefix columns, so the
    longer the prefix the fewer columns a pair costs (static_barcode_rows2)."""
    rem = list(targets)
    cand = sorted(((len(os.path.commonprefix([a, b])), a, b) for i, a in enumerate(rem) for b in rem[i + 1:]),
                  key=lambda x: (-x[0], x[1], x[2]))
    used, pairs = set(), []
    for lcp, a, b in cand:
        if a in used or b in used:
            continue
        used.update((a, b))
        pairs.append((a, b, min(lcp, len(a) - 1)))
    for t in rem:
        if t not in used:
            pairs.append((t, t, flank))
    return sorted(pairs)


def collect():
    """family (up, down) -> ordered list of distinct targets over every mode / kit selection"""
    n = qconfig.qcatConfig().barcode_context_length
    fams = collections.OrderedDict()
    members = {}                                 # target -> the (mode, kit, template, set) groups that scan it
    templates = []
    fused = []                                   # (template A, template B): the two templates of a kit
    for mode in scanner.get_modes():
        kits = [None] + sorted(scanner.get_kits())
        for kit in kits:
            try:
                det = scanner.factory(mode=mode, kit=kit)
            except Exception:
                continue
            by_kit = collections.OrderedDict()
            for lay in det.layouts:
                by_kit.setdefault(lay.kit, []).append(lay.sequence.upper())
            for seqs in by_kit.values():
                if len(seqs) == 2 and all(c in ACODE for q in seqs for c in q):
                    u = len(os.path.commonprefix(seqs))
                    # both rows plus the shared columns must fit four waves per SIMD (128 VGPRs)
                    if len(seqs[0]) + len(seqs[1]) - u <= 100 and tuple(seqs) not in fused:
                        fused.append(tuple(seqs))
            for lay in det.layouts:
                seq = lay.sequence.upper()
                if seq not in templates and all(c in ACODE for c in seq):
                    templates.append(seq)
                for s in range(2 if mode == "dual" else 1):
                    bs = lay.get_barcode_set(s)
                    if not bs:
                        continue
                    up, dn = lay.get_upstream_context(n, s), lay.get_downstream_context(n, s)
                    for b in bs:
                        t = (up + b.sequence + dn).upper()
                        if any(c not in CODE for c in t):
                            break                       # non-ACGT target: table kernels only
                        lst = fams.setdefault((up, dn, len(t)), [])
                        if t not in lst:
                            lst.append(t)
                        members.setdefault(t, set()).add((mode, kit, lay.sequence, s))
    return fams, templates, fused, members


class _Buf(object):
    def __init__(self):
        self.parts = []

    def write(self, text):
        self.parts.append(text)


def render():
    """text of static_generated.inc plus (n kernels, n targets, n templates)"""
    fams, templates, fused, members = collect()
    reg = []
    bs_fams = []                                     # (kernel, upstream columns, downstream columns, has bit-sliced rows)
    bs_structs = []                                  # (kernel, targets, text of its QBS struct) -> bs_static_generated.inc
    quad_reg = []                                    # (kernel, quad case, pair a, pair b)
    fh = _Buf()
    if True:
        fh.write("// GENERATED by tools/gen_static_kernels.py -- do not edit.\n")
        fh.write("// %d kernels, %d targets (built-in kits, barcode_context_length = %d).\n\n"
                 % (len(fams), sum(len(v) for v in fams.values()), qconfig.qcatConfig().barcode_context_length))
        fh.write("namespace qk {\n\n")
        for kid, ((up, dn, m), targets) in enumerate(fams.items()):
            u = len(up)
            fh.write("// kernel %d: %s + barcode + %s (%d columns, %d-column flank, %d targets)\n" % (kid, up, dn, m, u, len(targets)))
            # targets that are always scanned together (same set of kit groups) may be paired with each other
            groups = collections.OrderedDict()
            for t in targets:
                groups.setdefault(frozenset(members[t]), []).append(t)
            pairs = []
            quads = []                                   # (pair index a, pair index b): consecutive pairs of one membership group
            for grp in groups.values():
                first = len(pairs)
                pairs.extend(pair_up(grp, u))
                if len(targets) >= QUAD_MIN_TARGETS:
                    quads.extend((i, i + 1) for i in range(first, len(pairs) - 1, 2)
                                 if pairs[i][0] != pairs[i][1] and pairs[i + 1][0] != pairs[i + 1][1])
            npairs = len(pairs)
            for pr, (ta, tb, up_) in enumerate(pairs):
                reg.append((fnv1a64([CODE[c] for c in ta]), kid, 2 * pr, ta))
                if tb != ta:
                    reg.append((fnv1a64([CODE[c] for c in tb]), kid, 2 * pr + 1, tb))
                fh.write("struct QSP_%d_%d {      // %d shared columns\n" % (kid, pr, up_))
                fh.write("    static __device__ __forceinline__ void pre(h2 (&h)[%d], h2& carry, h2& left, const h2 (&E)[4]) { %s }\n"
                         % (up_ + 1, chain(ta[:up_], CODE) if up_ else ""))
                for name, t in (("ta", ta), ("tb", tb)):
                    fh.write("    static __device__ __forceinline__ void %s(h2 (&h)[%d], h2& carry, h2& left, const h2 (&E)[4]) { %s }\n"
                             % (name, m - up_ + 1, chain(t[up_:], CODE)))
                fh.write("};\n")
            # quads: two pairs in ONE row pass -- the columns all four targets share (at least the flank) and the
            # per-row work (selector, score registers, boundary) are paid once per four targets (static_barcode_rows4)
            for q, (pa, pb) in enumerate(quads):
                (a1, a2, ua), (b1, b2, ub) = pairs[pa], pairs[pb]
                u0 = min(len(os.path.commonprefix([a1, a2, b1, b2])), ua, ub)
                quad_reg.append((kid, q, pa, pb))
                fh.write("struct QSQ_%d_%d {      // %d columns shared by all four, +%d / +%d inside the pairs\n" % (kid, q, u0, ua - u0, ub - u0))
                fh.write("    static __device__ __forceinline__ void pre0(h2 (&h)[%d], h2& carry, h2& left, const h2 (&E)[4]) { %s }\n"
                         % (u0 + 1, chain(a1[:u0], CODE) if u0 else ""))
                for name, t, lo, hi in (("prea", a1, u0, ua), ("ta", a1, ua, m), ("tb", a2, ua, m),
                                        ("preb", b1, u0, ub), ("tc", b1, ub, m), ("td", b2, ub, m)):
                    fh.write("    static __device__ __forceinline__ void %s(h2 (&h)[%d], h2& carry, h2& left, const h2 (&E)[4]) { %s }\n"
                             % (name, hi - lo + 1, chain(t[lo:hi], CODE) if hi > lo else ""))
                fh.write("};\n")
            nq = len(quads)
            fh.write("struct QSG_%d {\n    static constexpr int M = %d;\n    static constexpr int HAS_QUADS = %d;\n" % (kid, m, 1 if nq else 0))
            fh.write("    static __device__ __forceinline__ void run4(int quad, const uint8_t* qbuf, int lane, int Lmax, h2 gL2, "
                     "u32 special, const u32 (&ltr)[4], h2 rowoff, h2 coloff, u32& ra, u32& rb, u32& rc, u32& rd) {\n"
                     "        ra = 0; rb = 0; rc = 0; rd = 0;\n        switch (quad) {\n")
            for q, (pa, pb) in enumerate(quads):
                (a1, a2, ua), (b1, b2, ub) = pairs[pa], pairs[pb]
                u0 = min(len(os.path.commonprefix([a1, a2, b1, b2])), ua, ub)
                fh.write("        case %d: static_barcode_rows4<M, %d, %d, %d, QSQ_%d_%d>(qbuf, lane, Lmax, gL2, special, ltr, rowoff, coloff, ra, rb, rc, rd); break;\n"
                         % (q, u0, ua - u0, ub - u0, kid, q))
            fh.write("        default: break;\n        }\n    }\n")
            fh.write("    static __device__ __forceinline__ void run(int pair, const uint8_t* qbuf, int lane, int Lmax, h2 gL2, "
                     "u32 special, const u32 (&ltr)[4], h2 rowoff, h2 coloff, u32& ra, u32& rb) {\n        ra = 0; rb = 0;\n        switch (pair) {\n")
            for pr, (ta, tb, up_) in enumerate(pairs):
                fh.write("        case %d: static_barcode_rows2<M, %d, QSP_%d_%d>(qbuf, lane, Lmax, gL2, special, ltr, rowoff, coloff, ra, rb); break;\n"
                         % (pr, up_, kid, pr))
            fh.write("        default: break;\n        }\n    }\n};\n")
            shape = bs_shape(len(up), len(dn), m) if len(targets) >= BS_MIN_TARGETS else None
            bs_fams.append((kid, len(up), len(dn), shape is not None))
            if shape:
                rev, pre, own, post = shape
                bh = _Buf()
                bh.write("struct QBS_%d {      // %s + barcode + %s: %s, %d shared + %d own + %d trailing columns, %d targets\n"
                         % (kid, up, dn, "reversed" if rev else "forward", pre, own, post, len(targets)))
                s1, s0 = bs_shared_words(targets[0], rev, pre, CODE)
                t1, t0 = bs_trailing_words(targets[0], rev, post, CODE)
                bh.write("    static constexpr int C = %d, KERNEL = %d, PRE = %d, POST = %d;\n"
                         "    static constexpr unsigned S1 = 0x%Xu, S0 = 0x%Xu;      // letters of the shared columns\n"
                         "    static constexpr unsigned T1 = 0x%Xu, T0 = 0x%Xu;      // letters of the trailing columns, last column first\n"
                         % (own, kid, pre, post, s1, s0, t1, t0))
                bh.write("    static __device__ __forceinline__ void rows(int kase, const BsRowArgs& ra, "
                         "u32 (&h1)[C], u32 (&h0)[C], u32 (&f)[BS_ND]) {\n        switch (kase) {\n")
                for pr, (ta, tb, up_) in enumerate(pairs):
                    for half, t in ((0, ta), (1, tb)):
                        if half == 1 and tb == ta:
                            continue
                        w1, w0 = bs_words(t, rev, pre, CODE, own)
                        bh.write("        case %d: bs_rows_static<C, 0x%XULL, 0x%XULL>(ra, h1, h0, f); break;\n"
                                 % (2 * pr + half, w1, w0))
                bh.write("        default: break;\n        }\n    }\n};\n")
                bs_structs.append((kid, len(targets), "".join(bh.parts)))
            fh.write("\n")
        reg.sort()
        assert len(set(h for h, _, _, _ in reg)) == len(reg), "hash collision between targets"
        fh.write("#ifndef QCAT_STATIC_MULTI_TU      // (static_multi.hip takes the column chains above and the merged kernels only)\n")
        fh.write("// (the hash only finds the entry; static_match compares `seq` with the kit's target before binding)\n")
        fh.write("struct StaticTarget { uint64_t hash; int16_t kernel, kase; const char* seq; };\n")
        fh.write("static const StaticTarget g_static_targets[] = {\n")
        for h, kid, case, seq in reg:
            fh.write("    {0x%016XULL, %d, %d, \"%s\"},\n" % (h, kid, case, seq))
        fh.write("};\nstatic const int g_n_static_targets = %d;\n" % len(reg))
        fh.write("static const int g_static_kernel_M[] = {%s};\n" % ", ".join(str(m) for (_, _, m) in fams))
        fh.write("// per kernel: upstream / downstream context columns of the family, and whether it has bit-sliced rows (QBS_n)\n")
        fh.write("static const int g_static_kernel_up[] = {%s};\n" % ", ".join(str(u) for (_, u, _, _) in bs_fams))
        fh.write("static const int g_static_kernel_dn[] = {%s};\n" % ", ".join(str(d) for (_, _, d, _) in bs_fams))
        fh.write("static const int g_static_kernel_bs[] = {%s};\n" % ", ".join("1" if b else "0" for (_, _, _, b) in bs_fams))
        fh.write("// quads of a kernel: (kernel, quad case, pair case a, pair case b); a kit group runs them when it scans both pairs\n")
        fh.write("struct StaticQuad { int16_t kernel, quad, pair_a, pair_b; };\n")
        fh.write("static const StaticQuad g_static_quads[] = {\n")
        for kid, q, pa, pb in quad_reg:
            fh.write("    {%d, %d, %d, %d},\n" % (kid, q, pa, pb))
        fh.write("    {-1, -1, -1, -1}\n};\nstatic const int g_n_static_quads = %d;\n\n" % len(quad_reg))
        fh.write("static inline void launch_barcode_static(int kernel, dim3 grid, hipStream_t stream, const StaticArgs& a) {\n"
                 "    if (kernel >= QCAT_JIT_BASE) { jit_launch(QCAT_JIT_BARCODE, kernel - QCAT_JIT_BASE, grid, stream, &a); return; }\n"
                 "    switch (kernel) {\n")
        for kid in range(len(fams)):
            fh.write("    case %d: hipLaunchKernelGGL(k_barcode_static<QSG_%d>, grid, dim3(PK_WAVES * 64), 0, stream, a); break;\n" % (kid, kid))
        fh.write("    default: break;\n    }\n}\n#endif\n\n")
        fh.write("// every group of a SMALL batch in one launch (packed_host.inc: packed_barcode): blockIdx.x % n = the group.  A kit-auto batch launches\n"
                 "// one kernel per (template, set) group although only the voted kit's groups have jobs, and the runtime's four hardware\n"
                 "// queues serialise them around the two or three that do.  Compiled in a translation unit of its own (static_multi.hip).\n"
                 "#ifdef QCAT_STATIC_MULTI_TU\n"
                 "__global__ void __launch_bounds__(PK_WAVES * 64, 2)\n"
                 "k_barcode_multi(StaticBarcodeMulti m) {\n"
                 "    __shared__ uint8_t qbuf[PK_ROWS * 64];\n"
                 "    const int i = blockIdx.x % (uint32_t)m.n;      // (interleaved: the workgroups of the groups with jobs are resident side by side)\n"
                 "    StaticArgs a = m.common;\n"
                 "    a.gidx = m.gidx[i]; a.chunk_b = m.chunk_b[i];\n"
                 "    switch (m.kernel[i]) {\n")
        for kid in range(len(fams)):
            fh.write("    case %d: barcode_static_core<QSG_%d>(a, qbuf); break;\n" % (kid, kid))
        fh.write("    default: break;\n    }\n}\n"
                 "extern \"C\" void qcat_static_multi_barcode(unsigned grid, void* stream, const void* m) {\n"
                 "    hipLaunchKernelGGL(k_barcode_multi, dim3(grid), dim3(PK_WAVES * 64), 0, static_cast<hipStream_t>(stream), *static_cast<const StaticBarcodeMulti*>(m));\n}\n"
                 "#else\n"
                 "extern \"C\" void qcat_static_multi_barcode(unsigned grid, void* stream, const void* m);\n"
                 "static inline void launch_barcode_multi(dim3 grid, hipStream_t stream, const StaticBarcodeMulti& m) { qcat_static_multi_barcode(grid.x, stream, &m); }\n"
                 "#endif\n\n")
        # the bit-sliced static-letter kernels are compiled in translation units of their own (bs_static.hip with
        # QCAT_BS_PART = 0..BS_PARTS-1, in parallel with this one): greedy split by number of targets
        parts = [[] for _ in range(BS_PARTS)]
        for kid, nt, text in sorted(bs_structs, key=lambda x: -x[1]):
            min(parts, key=lambda p: sum(n for _, n, _ in p)).append((kid, nt, text))
        fh.write("}  // namespace qk\n#ifndef QCAT_STATIC_MULTI_TU\n")
        for p in range(BS_PARTS):
            fh.write('extern "C" void qcat_bs_launch_part%d(int kernel, unsigned grid, void* stream, const void* args);   // bs_static.hip\n' % p)
        fh.write("#endif\nnamespace qk {\n#ifndef QCAT_STATIC_MULTI_TU\n")
        fh.write("static inline void launch_bs_static(int kernel, dim3 grid, hipStream_t stream, const BsArgs& a) {\n"
                 "    if (kernel >= QCAT_JIT_BASE) { jit_launch(QCAT_JIT_BITSLICE, kernel - QCAT_JIT_BASE, grid, stream, &a); return; }\n"
                 "    switch (kernel) {\n")
        for p, part in enumerate(parts):
            if part:
                fh.write("    %s qcat_bs_launch_part%d(kernel, grid.x, stream, &a); break;\n"
                         % (" ".join("case %d:" % kid for kid, _, _ in sorted(part)), p))
        fh.write("    default: break;\n    }\n}\n#endif\n\n")
        bparts = ["// GENERATED by tools/gen_static_kernels.py -- do not edit.\n"
                  "// Bit-sliced barcode kernels with the target letters compiled in (kernels_bitslice.inc), one struct per target\n"
                  "// family; compiled by bs_static.hip in %d parts (QCAT_BS_PART), each in its own namespace.\n\n" % BS_PARTS]
        for p, part in enumerate(parts):
            bparts.append("#if QCAT_BS_PART == %d\nnamespace qk {\n" % p)
            for kid, _, text in sorted(part):
                bparts.append(text)
            bparts.append("}  // namespace qk\n")
            bparts.append('extern "C" void qcat_bs_launch_part%d(int kernel, unsigned grid, void* stream, const void* args) {\n'
                          "    const qk::BsArgs& a = *static_cast<const qk::BsArgs*>(args);\n    switch (kernel) {\n" % p)
            for kid, _, _ in sorted(part):
                bparts.append("    case %d: hipLaunchKernelGGL(qk::k_bs_barcode<qk::QBS_%d>, dim3(grid), dim3(qk::BS_WAVES * 64), 0, "
                              "static_cast<hipStream_t>(stream), a); break;\n" % (kid, kid))
            bparts.append("    default: break;\n    }\n}\n#endif\n\n")
        bs_text = "".join(bparts)
        # ---- adapter templates ------------------------------------------------------------------
        areg = []
        for tid, seq in enumerate(templates):
            cols = chain(seq, ACODE)
            fh.write("// adapter template %d: %s\n" % (tid, seq))
            fh.write("struct QAC_%d { static __device__ __forceinline__ void run(h2 (&h)[%d], h2& carry, h2& left, "
                     "const h2 (&E)[5]) { %s } };\n" % (tid, len(seq) + 1, cols))
            areg.append((fnv1a64([ACODE[c] for c in seq]), tid, len(seq), seq))
        areg.sort()
        assert len(set(h for h, _, _, _ in areg)) == len(areg), "hash collision between templates"
        fh.write("\n#ifndef QCAT_STATIC_MULTI_TU\nstruct StaticTemplate { uint64_t hash; int16_t kernel, len; const char* seq; };\n")
        fh.write("static const StaticTemplate g_static_templates[] = {\n")
        for h, tid, m, seq in areg:
            fh.write("    {0x%016XULL, %d, %d, \"%s\"},\n" % (h, tid, m, seq))
        fh.write("};\nstatic const int g_n_static_templates = %d;\n\n" % len(areg))
        fh.write("static inline void launch_adapter_static(int kernel, dim3 grid, hipStream_t stream, const StaticAdapterArgs& a) {\n"
                 "    if (kernel >= QCAT_JIT_BASE) { jit_launch(QCAT_JIT_ADAPTER, kernel - QCAT_JIT_BASE, grid, stream, &a); return; }\n"
                 "    switch (kernel) {\n")
        for tid, seq in enumerate(templates):
            fh.write("    case %d: hipLaunchKernelGGL((k_adapter_static<%d, QAC_%d>), grid, dim3(PK_WAVES * 64), 0, stream, a); break;\n"
                     % (tid, len(seq), tid))
        fh.write("    default: break;\n    }\n}\n#endif\n\n")
        # ---- two-template kits: one fused pass ----------------------------------------------------
        for fid, (sa, sb) in enumerate(fused):
            u = len(os.path.commonprefix([sa, sb]))
            fh.write("// fused adapter kernel %d: %d shared columns of\n//   %s\n//   %s\n" % (fid, u, sa, sb))
            fh.write("struct QAF_%d {\n" % fid)
            fh.write("    static __device__ __forceinline__ void pre(h2 (&h)[%d], h2& carry, h2& left, const h2 (&E)[5]) { %s }\n"
                     % (u + 1, chain(sa[:u], ACODE) if u else ""))
            for name, q in (("ta", sa), ("tb", sb)):
                fh.write("    static __device__ __forceinline__ void %s(h2 (&h)[%d], h2& carry, h2& left, const h2 (&E)[5]) { %s }\n"
                         % (name, len(q) - u + 1, chain(q[u:], ACODE)))
            fh.write("};\n")
        fh.write("\n#ifndef QCAT_STATIC_MULTI_TU\n// (tpl_a / tpl_b: the static adapter kernels of the two templates, which static_match has verified)\n")
        fh.write("struct StaticFused { int16_t tpl_a, tpl_b, kernel; };\n")
        fh.write("static const StaticFused g_static_fused[] = {\n")
        for fid, (sa, sb) in enumerate(fused):
            fh.write("    {%d, %d, %d},\n" % (templates.index(sa), templates.index(sb), fid))
        fh.write("};\nstatic const int g_n_static_fused = %d;\n\n" % len(fused))
        fh.write("static inline void launch_adapter_fused(int kernel, dim3 grid, hipStream_t stream, const StaticAdapterArgs& a) {\n"
                 "    switch (kernel) {\n")
        for fid, (sa, sb) in enumerate(fused):
            u = len(os.path.commonprefix([sa, sb]))
            fh.write("    case %d: hipLaunchKernelGGL((k_adapter_fused2<%d, %d, %d, QAF_%d>), grid, dim3(PK_WAVES * 64), 0, stream, a); break;\n"
                     % (fid, u, len(sa), len(sb), fid))
        fh.write("    default: break;\n    }\n}\n#endif\n\n")
        # ---- every chain of a small batch in one launch ------------------------------------------
        fh.write("// every static-letter adapter chain of a SMALL batch in one launch (packed_host.inc: packed_adapter): blockIdx.y = the unit --\n"
                 "// a template or a fused pair.  The units of such a batch are latency chains (one wave per SIMD, 50-110 us each for the\n"
                 "// 4000 reads of the reference driver's call), and launches of their own are serialised by the runtime's four hardware queues.\n"
                 "#ifdef QCAT_STATIC_MULTI_TU\n"
                 "__global__ void __launch_bounds__(PK_WAVES * 64, 2)\n"
                 "k_adapter_multi(StaticAdapterMulti m) {\n"
                 "    __shared__ uint8_t qbuf[PK_ROWS * 64];\n"
                 "    __shared__ uint16_t slow_tbl[5 * 16];\n"
                 "    const int i = blockIdx.y;\n"
                 "    StaticAdapterArgs a = m.common;\n"
                 "    a.bests = m.bests[i]; a.bests2 = m.bests2[i]; a.tpl = m.tpl[i]; a.tpl2 = m.tpl2[i];\n"
                 "    const int kernel = m.kernel[i];\n"
                 "    if (m.fused[i]) {\n"
                 "        switch (kernel) {\n")
        for fid, (sa, sb) in enumerate(fused):
            u = len(os.path.commonprefix([sa, sb]))
            fh.write("        case %d: adapter_fused2_core<%d, %d, %d, QAF_%d>(a, qbuf, slow_tbl); break;\n" % (fid, u, len(sa), len(sb), fid))
        fh.write("        default: break;\n        }\n    } else {\n        switch (kernel) {\n")
        for tid, seq in enumerate(templates):
            fh.write("        case %d: adapter_static_core<%d, QAC_%d>(a, qbuf, slow_tbl); break;\n" % (tid, len(seq), tid))
        fh.write("        default: break;\n        }\n    }\n}\n"
                 "extern \"C\" void qcat_static_multi_adapter(unsigned grid_x, unsigned grid_y, void* stream, const void* m) {\n"
                 "    hipLaunchKernelGGL(k_adapter_multi, dim3(grid_x, grid_y), dim3(PK_WAVES * 64), 0, static_cast<hipStream_t>(stream), *static_cast<const StaticAdapterMulti*>(m));\n}\n"
                 "#else\n"
                 "extern \"C\" void qcat_static_multi_adapter(unsigned grid_x, unsigned grid_y, void* stream, const void* m);\n"
                 "static inline void launch_adapter_multi(dim3 grid, hipStream_t stream, const StaticAdapterMulti& m) { qcat_static_multi_adapter(grid.x, grid.y, stream, &m); }\n"
                 "#endif\n\n")
        fh.write("// the same column chains over the read interior (--detect-middle, kernels_middle.inc)\n")
        fh.write("#ifdef QCAT_HAVE_MIDDLE_KERNELS\n")
        fh.write("static inline void launch_adapter_middle(int kernel, dim3 grid, hipStream_t stream, const MiddleAdapterArgs& a) {\n"
                 "    if (kernel >= QCAT_JIT_BASE) { jit_launch(QCAT_JIT_MIDDLE, kernel - QCAT_JIT_BASE, grid, stream, &a); return; }\n"
                 "    switch (kernel) {\n")
        for tid, seq in enumerate(templates):
            fh.write("    case %d: hipLaunchKernelGGL((k_adapter_middle<%d, QAC_%d>), grid, dim3(PK_WAVES * 64), 0, stream, a); break;\n"
                     % (tid, len(seq), tid))
        fh.write("    default: break;\n    }\n}\n#endif\n\n}  // namespace qk\n")
    return "".join(fh.parts), len(fams), len(reg), len(templates), bs_text


def main():
    text, nk, nt, na, bs_text = render()
    with open(OUT, "w") as out:
        out.write(text)
    with open(BS_OUT, "w") as out:
        out.write(bs_text)
    print("wrote %s: %d barcode kernels, %d targets, %d adapter templates" % (OUT, nk, nt, na))


if __name__ == "__main__":
    main()
