"""Column programs ("plans") of the bit-sliced ADAPTER kernels (csrc/kernels_abs.inc, csrc/abs_core.h): the generator.

A plan is what ONE pass over a read window computes: the two templates of a kit that has a fused binary16 kernel -- their
common prefix once, then each tail -- or a single template.  The pass is a linear program over the template columns

    start, col.. (shared prefix), fork, col.. (tail A), border A, resume, col.. (tail B), border B

cut into stages of about equal instruction count that run on the waves of a workgroup as a software pipeline over blocks of
rows; the differences `a` that cross a cut (the running one and, when the cut falls between fork and resume, the forked one)
travel through the LDS row by row.  Two forms: `emit_plan` -- TWO stages, every border (last column of a template) in stage
1, up to 52 columns per stage (256 VGPRs, k_adapter_bs); `emit_multi` -- FOUR stages, a stage may hold borders
(k_adapter_ms: up to 13 columns per stage at 128 VGPRs for medium batches; k_adapter_mw: up to 26 per stage at two waves per
SIMD for templates too long for two stages, VMK001's 102 columns).

Used twice: tools/gen_abs_kernels.py writes csrc/abs_generated.inc for the built-in kits at build time, qcat_amd/jit.py
emits the same structs for the templates of a custom kit at run time (hipRTC)."""
import os

LETTER = {"A": 0, "T": 1, "G": 2, "C": 3}          # qcat_amd/codes.py: the plane code of a letter
# instructions per row as compiled (profiles/r03_*): cells 24 / 24 (the searched networks of abs_core.h), a border step with its index latch 46, a hand-over
# set 5 LDS instructions on either side; stage 1 also pays the row's LDS reads, masks and loop overhead (+30)
COST = {"L": 24, "N": 24, "border": 46, "handover": 5, "stage1": 30}
MAX_STAGE_COLUMNS = 52                              # 4 planes per column + ~60 working registers <= 256 VGPRs (two waves per SIMD)


def program(seqs):
    """linear program of a plan: list of ops"""
    if len(seqs) == 1:
        return [("start",)] + [("col", c) for c in seqs[0]] + [("border", 0)]
    sa, sb = seqs
    u = len(os.path.commonprefix([sa, sb]))
    if u == 0:
        return ([("start",)] + [("col", c) for c in sa] + [("border", 0)] +
                [("start",)] + [("col", c) for c in sb] + [("border", 1)])
    return ([("start",)] + [("col", c) for c in sa[:u]] + [("fork",)] + [("col", c) for c in sa[u:]] + [("border", 0)] +
            [("resume",)] + [("col", c) for c in sb[u:]] + [("border", 1)])


def op_cost(op):
    if op[0] == "col":
        return COST["N"] if op[1] == "N" else COST["L"]
    if op[0] == "border":
        return COST["border"]
    return 0


def split_point(ops):
    """index p: ops[:p] = stage 0.  Balanced by the cost model, no border in stage 0, cut only after a column."""
    total = sum(op_cost(o) for o in ops)
    first_border = min(i for i, o in enumerate(ops) if o[0] == "border")
    best, best_p, run = None, None, 0
    for i, o in enumerate(ops):
        run += op_cost(o)
        if o[0] != "col" or i + 1 > first_border:
            continue
        nxt = ops[i + 1][0]
        if nxt in ("border",):
            continue                                   # the border belongs with its column's stage (stage 1)
        live = 1 + (1 if any(x[0] == "fork" for x in ops[:i + 1]) and any(x[0] == "resume" for x in ops[i + 1:]) else 0)
        if nxt == "fork":
            live = 1                                   # cut right before the fork: stage 1 forks itself
        s0 = run + COST["handover"] * live
        s1 = total - run + COST["handover"] * live + COST["stage1"]
        score = max(s0, s1)
        if best is None or score < best:
            best, best_p = score, i + 1
    return best_p


def emit_plan(name, seqs, comment):
    ops = program(seqs)
    p = split_point(ops)
    s0, s1 = ops[:p], ops[p:]
    fork_in_0 = any(o[0] == "fork" for o in s0)
    resume_in_1 = any(o[0] == "resume" for o in s1)
    fork_live = fork_in_0 and resume_in_1
    nh = 2 if fork_live else 1
    nc0 = sum(1 for o in s0 if o[0] == "col")
    nc1 = sum(1 for o in s1 if o[0] == "col")
    nt = len(seqs)
    if max(nc0, nc1) > MAX_STAGE_COLUMNS:
        return None                                   # the stage's difference planes would not fit a wave's registers
    cur_slot = nh - 1                                 # hand-over slot of the running difference; slot 0 = the forked one
    out = []
    out.append("// %s\n" % comment)
    for i, q in enumerate(seqs):
        out.append("//   template %d (%d columns): %s\n" % (i, len(q), q))
    out.append("//   stage 0: %d columns, stage 1: %d columns, %d hand-over set%s, cost model %d / %d instructions per row\n"
               % (nc0, nc1, nh, "s" if nh > 1 else "",
                  sum(op_cost(o) for o in s0), sum(op_cost(o) for o in s1)))
    out.append("struct %s {\n" % name)
    out.append("    static constexpr int NT = %d, NH = %d, NC0 = %d, NC1 = %d;\n" % (nt, nh, nc0, nc1))
    out.append("    static constexpr int M0 = %d, M1 = %d;\n" % (len(seqs[0]), len(seqs[1]) if nt > 1 else 0))

    def cell(j, c):
        if c == "N":
            return "abs_cell_n(a, h[%d]);" % j
        return "abs_cell_letter(nq[%d], a, h[%d]);" % (LETTER[c], j)

    # ---- row0 ----
    body, j = [], 0
    for o in s0:
        if o[0] == "start":
            body.append("abs_set2(a);")
        elif o[0] == "col":
            body.append(cell(j, o[1])); j += 1
        elif o[0] == "fork":
            body.append("ABS_COPY4(f, a);")
    if fork_live:
        body.append("ABS_COPY4(ho[0], f);")
    body.append("ABS_COPY4(ho[%d], a);" % cur_slot)
    out.append("    static ABS_FN void row0(const u32 (&nq)[4], u32 (&h)[NC0][4], u32 (&ho)[NH][4]) {\n"
               "        u32 a[4]%s;\n        %s\n    }\n" % (", f[4]" if fork_in_0 else "", "\n        ".join(body)))
    # ---- row1 ----
    body, j = ["ABS_COPY4(a, hi[%d]);" % cur_slot], 0
    if fork_live:
        body.append("ABS_COPY4(f, hi[0]);")
    for o in s1:
        if o[0] == "start":
            body.append("abs_set2(a);")
        elif o[0] == "col":
            body.append(cell(j, o[1])); j += 1
        elif o[0] == "fork":
            body.append("ABS_COPY4(f, a);")
        elif o[0] == "resume":
            body.append("ABS_COPY4(a, f);")
        elif o[0] == "border":
            body.append("{ const u32 nm = abs_border_step(bd[%d].Fc, a, first); abs_latch_index(bd[%d].ic, nm, row); }" % (o[1], o[1]))
    need_f1 = fork_live or any(o[0] == "fork" for o in s1)
    out.append("    static ABS_FN void row1(const u32 (&nq)[4], u32 (&h)[NC1][4], const u32 (&hi)[NH][4], AbsBorder (&bd)[NT], u32 first, unsigned row) {\n"
               "        u32 a[4]%s;\n        %s\n    }\n" % (", f[4]" if need_f1 else "", "\n        ".join(body)))
    # ---- last0 / last1: the walk along the last row, same program over the b planes ----
    body, j, first = [], 0, False
    for o in s0:
        if o[0] == "start":
            body.append("abs_lastrow_init(r);"); first = True
        elif o[0] == "col":
            body.append("abs_lastrow_step(r, h[%d], %s);" % (j, "true" if first else "false")); j += 1; first = False
        elif o[0] == "fork":
            body.append("rf = r;")
    if fork_live:
        body.append("lo[0] = rf;")
    body.append("lo[%d] = r;" % cur_slot)
    out.append("    static ABS_FN void last0(const u32 (&h)[NC0][4], AbsLastRow (&lo)[NH]) {\n"
               "        AbsLastRow r%s;\n        %s\n    }\n" % (", rf" if fork_in_0 else "", "\n        ".join(body)))
    body, j, first = ["r = li[%d];" % cur_slot], 0, False
    if fork_live:
        body.append("rf = li[0];")
    for o in s1:
        if o[0] == "start":
            body.append("abs_lastrow_init(r);"); first = True
        elif o[0] == "col":
            body.append("abs_lastrow_step(r, h[%d], %s);" % (j, "true" if first else "false")); j += 1; first = False
        elif o[0] == "fork":
            body.append("rf = r;")
        elif o[0] == "resume":
            body.append("r = rf;")
        elif o[0] == "border":
            body.append("lr[%d] = r;" % o[1])
    out.append("    static ABS_FN void last1(const u32 (&h)[NC1][4], const AbsLastRow (&li)[NH], AbsLastRow (&lr)[NT]) {\n"
               "        AbsLastRow r%s;\n        %s\n    }\n" % (", rf" if need_f1 else "", "\n        ".join(body)))
    out.append("};\n\n")
    return "".join(out)


# ---------------------------------------------------------------------------------------------------------------------
# plans of MS_STAGES stages (k_adapter_ms, kernels_abs.inc): the same linear program cut into four pieces, one wave each.
# A stage may hold borders (it then owns that template's end-position state and writes its result); every cut hands over
# the running difference and, between fork and resume, the forked one.  Four waves per tile: twice the waves of the
# two-stage plans for the same tiles -- for MEDIUM batches (1.5 to 3.5 tiles per CU), where a two-wave workgroup per tile
# leaves the SIMDs with one wave each.  Emitted for single templates whose stages hold at most MS_MAX_COLUMNS columns: those
# compile to 128 VGPRs (four waves per SIMD, two templates' kernels side by side fit the chip).  Wider stages (168 VGPRs)
# and the fused two-template plans were measured and lose (profiles/r03_ab_adapter_stages.txt).
# ---------------------------------------------------------------------------------------------------------------------
MS_STAGES = 4
MS_MAX_COLUMNS = 13                                  # 4 planes per column + ~70 working registers <= 128 VGPRs
MW_MAX_COLUMNS = 26                                  # ... <= 256 VGPRs (k_adapter_mw: templates too long for two stages)


def fork_live_at(ops, p):
    return any(o[0] == "fork" for o in ops[:p]) and any(o[0] == "resume" for o in ops[p:])


def multi_cuts(ops, ns, maxc):
    """cut positions (ns - 1 of them, ops[:c0] = stage 0, ...) minimising the largest stage cost; None if a stage cannot
    keep its columns in registers"""
    import itertools
    cand = [i + 1 for i, o in enumerate(ops[:-1]) if o[0] == "col" and ops[i + 1][0] != "border"]
    pre = [0]
    for o in ops:
        pre.append(pre[-1] + op_cost(o))
    colpre = [0]
    for o in ops:
        colpre.append(colpre[-1] + (1 if o[0] == "col" else 0))
    live = {c: (2 if fork_live_at(ops, c) else 1) for c in cand}
    best, best_cuts = None, None
    for cuts in itertools.combinations(cand, ns - 1):
        b = (0,) + cuts + (len(ops),)
        worst = 0
        ok = True
        for k in range(ns):
            lo, hi = b[k], b[k + 1]
            nc = colpre[hi] - colpre[lo]
            if nc < 1 or nc > maxc:
                ok = False
                break
            c = pre[hi] - pre[lo] + COST["stage1"]
            if k > 0:
                c += COST["handover"] * live[lo]
            if k < ns - 1:
                c += COST["handover"] * live[hi]
            worst = max(worst, c)
        if ok and (best is None or worst < best):
            best, best_cuts = worst, cuts
    return best_cuts


def emit_multi(name, seqs, comment, ns=MS_STAGES, maxc=MS_MAX_COLUMNS):
    ops = program(seqs)
    cuts = multi_cuts(ops, ns, maxc)
    if cuts is None:
        return None
    b = (0,) + tuple(cuts) + (len(ops),)
    nt = len(seqs)
    out = ["// %s, %d stages\n" % (comment, ns)]
    for i, q in enumerate(seqs):
        out.append("//   template %d (%d columns): %s\n" % (i, len(q), q))
    out.append("struct %s {\n" % name)
    out.append("    static constexpr int NS = %d, NT = %d;\n" % (ns, nt))
    out.append("    static constexpr int M0 = %d, M1 = %d;\n" % (len(seqs[0]), len(seqs[1]) if nt > 1 else 0))
    ring_off, off = [], 0
    for k in range(ns - 1):
        nho = 2 if fork_live_at(ops, b[k + 1]) else 1
        ring_off.append(off)
        off += 2 * 4 * nho                               # planes of the cut's two ring buffers per row (x ABS_R x 64 words)
    out.append("    static constexpr int RING_PLANES = %d;      // hand-over planes of all cuts, both buffers, per ring row\n" % off)

    def cell(j, c):
        if c == "N":
            return "abs_cell_n(a, h[%d]);" % j
        return "abs_cell_letter(nq[%d], a, h[%d]);" % (LETTER[c], j)

    for k in range(ns):
        sops = ops[b[k]:b[k + 1]]
        nhi = 0 if k == 0 else (2 if fork_live_at(ops, b[k]) else 1)
        nho = 0 if k == ns - 1 else (2 if fork_live_at(ops, b[k + 1]) else 1)
        nc = sum(1 for o in sops if o[0] == "col")
        borders = [o[1] for o in sops if o[0] == "border"]
        need_f = nhi == 2 or any(o[0] == "fork" for o in sops)
        out.append("    struct S%d {      // %d columns, cost model %d instructions per row\n"
                   % (k, nc, sum(op_cost(o) for o in sops)))
        out.append("        static constexpr int NC = %d, NHI = %d, NHO = %d, NBD = %d, HI = %d, HO = %d, BD = %d, BT0 = %d, BT1 = %d, RING_IN = %d, RING_OUT = %d;\n"
                   % (nc, nhi, nho, len(borders), max(1, nhi), max(1, nho), max(1, len(borders)),
                      borders[0] if borders else -1, borders[1] if len(borders) > 1 else -1,
                      ring_off[k - 1] if k > 0 else -1, ring_off[k] if k < ns - 1 else -1))
        # ---- one DP row ----
        body, j = [], 0
        if nhi:
            body.append("ABS_COPY4(a, hi[%d]);" % (nhi - 1))
            if nhi == 2:
                body.append("ABS_COPY4(f, hi[0]);")
        for o in sops:
            if o[0] == "start":
                body.append("abs_set2(a);")
            elif o[0] == "col":
                body.append(cell(j, o[1])); j += 1
            elif o[0] == "fork":
                body.append("ABS_COPY4(f, a);")
            elif o[0] == "resume":
                body.append("ABS_COPY4(a, f);")
            elif o[0] == "border":
                bi = borders.index(o[1])
                body.append("{ const u32 nm = abs_border_step(bd[%d].Fc, a, first); abs_latch_index(bd[%d].ic, nm, row); }" % (bi, bi))
        if nho:
            if nho == 2:
                body.append("ABS_COPY4(ho[0], f);")
            body.append("ABS_COPY4(ho[%d], a);" % (nho - 1))
        out.append("        static ABS_FN void row(const u32 (&nq)[4], u32 (&h)[NC][4], const u32 (&hi)[HI][4], u32 (&ho)[HO][4], AbsBorder (&bd)[BD], u32 first, unsigned row) {\n"
                   "            u32 a[4]%s;\n            %s\n        }\n" % (", f[4]" if need_f else "", "\n            ".join(body)))
        # ---- the walk along the last row ----
        body, j, first = [], 0, False
        if nhi:
            body.append("r = li[%d];" % (nhi - 1))
            if nhi == 2:
                body.append("rf = li[0];")
        for o in sops:
            if o[0] == "start":
                body.append("abs_lastrow_init(r);"); first = True
            elif o[0] == "col":
                body.append("abs_lastrow_step(r, h[%d], %s);" % (j, "true" if first else "false")); j += 1; first = False
            elif o[0] == "fork":
                body.append("rf = r;")
            elif o[0] == "resume":
                body.append("r = rf;")
            elif o[0] == "border":
                body.append("lr[%d] = r;" % borders.index(o[1]))
        if nho:
            if nho == 2:
                body.append("lo[0] = rf;")
            body.append("lo[%d] = r;" % (nho - 1))
        out.append("        static ABS_FN void last(const u32 (&h)[NC][4], const AbsLastRow (&li)[HI], AbsLastRow (&lo)[HO], AbsLastRow (&lr)[BD]) {\n"
                   "            AbsLastRow r%s;\n            %s\n        }\n    };\n" % (", rf" if need_f else "", "\n            ".join(body)))
    out.append("};\n\n")
    return "".join(out)


