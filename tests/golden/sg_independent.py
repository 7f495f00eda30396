"""Independent scalar semi-global aligner, written from the published recurrence -- NOT from
oracle/qcat_oracle.c and not calling it.  Dev tool of the fixture generators (make_golden.py,
make_sg_vectors.py): it is what stands in for the absent third-party ``parasail`` when the
reference's Python is executed here, so every ``tpl_raw`` / ``tpl_end`` / per-barcode row in the
committed fixtures is a value the oracle did NOT produce.

Definition (Gotoh, affine gaps, all four end gaps free; call sites
qcat/scanner_base.py:111-117 and :214-218 -- ``parasail.sg_striped_32(s1, s2, open, extend, matrix)``):

    H[i][0] = H[0][j] = 0
    E[i][j] = max(E[i][j-1] - extend, H[i][j-1] - open)      gap in the query  (consumes target)
    F[i][j] = max(F[i-1][j] - extend, H[i-1][j] - open)      gap in the target (consumes query)
    H[i][j] = max(H[i-1][j-1] + W(s1[i-1], s2[j-1]), E[i][j], F[i][j])

    a gap of length k costs open + (k-1)*extend.

Result: the best cell of the last row (query consumed) or last column (target consumed), with
parasail's striped end-position rule: the last row is scanned first, target index ascending, strict
``>``; then the last column: a strictly greater cell replaces the result, an equal cell only lowers
``end_query`` and only if ``end_ref`` already is the last target position.  Positions are 0-based
indices of the last aligned base.

The layout is deliberately different from the oracle's (which walks query rows and keeps one row):
this one fills whole matrices column by column (target-major), so a shared indexing slip would have
to be made twice in two different shapes to go unnoticed.
"""

NEG = -(1 << 40)


def make_scorer(alphabet, flat, size):
    """W(a, b) for a parasail-style matrix: ``flat`` holds size*size ints, row = mapper(b), column =
    mapper(a); the mapper sends the alphabet's letters (either case) to their index and everything
    else to size-1 (the '*' row/column)."""
    index = {}
    for k, ch in enumerate(alphabet):
        index[ch.upper()] = k
        index[ch.lower()] = k
    star = size - 1

    def score(a, b):
        return flat[index.get(b, star) * size + index.get(a, star)]
    return score


def scorer_from_table7(table7):
    """W(a, b) from the 7x7 [target, query] table over A T G C N X other used by the C ABI."""
    flat = [int(v) for row in table7 for v in row]
    return make_scorer("ATGCNX", flat, 7)


def sg(s1, s2, gap_open, gap_extend, score, rule="striped"):
    """-> (score, end_query, end_ref).  s1 = query (read window / region), s2 = target.
    rule: "striped" = parasail.sg_striped_32's end-position order as recalled (above); "scalar" = plain parasail.sg's, the
    routine the reference binds when parasail reports no SSE2 (qcat/scanner_base.py:20-26): the last column is examined while
    the rows go by (strictly greater replaces -> the first row reaching its maximum), then the last row, target index
    ascending, strictly greater replaces -- on a tie between the borders the last column keeps the result.  The switch is
    shared with oracle/qcat_oracle.c (qo_sg_rule) and the device kernels (include/qcat_hip.h QCAT_R1_*)."""
    assert rule in ("striped", "scalar")
    n, m = len(s1), len(s2)
    if n == 0 or m == 0:
        raise ValueError("empty sequence")
    # full matrices, indexed [j][i]: column j of the target, row i of the query
    H = [[0] * (n + 1) for _ in range(m + 1)]
    E = [[NEG] * (n + 1) for _ in range(m + 1)]
    F = [[NEG] * (n + 1) for _ in range(m + 1)]
    for j in range(1, m + 1):
        b = s2[j - 1]
        Hp, Hc = H[j - 1], H[j]
        Ep, Ec, Fc = E[j - 1], E[j], F[j]
        for i in range(1, n + 1):
            e = Ep[i] - gap_extend
            x = Hp[i] - gap_open
            if x > e:
                e = x
            f = Fc[i - 1] - gap_extend
            x = Hc[i - 1] - gap_open
            if x > f:
                f = x
            h = Hp[i - 1] + score(s1[i - 1], b)
            if e > h:
                h = e
            if f > h:
                h = f
            Ec[i], Fc[i], Hc[i] = e, f, h
    if rule == "scalar":
        last = H[m]
        best, end_query, end_ref = None, 0, m - 1
        for i in range(1, n + 1):
            if best is None or last[i] > best:
                best, end_query = last[i], i - 1
        for j in range(1, m + 1):
            if H[j][n] > best:
                best, end_query, end_ref = H[j][n], n - 1, j - 1
        return best, end_query, end_ref
    # last row: query fully consumed, scan the target positions in ascending order
    best, end_query, end_ref = None, n - 1, 0
    for j in range(1, m + 1):
        v = H[j][n]
        if best is None or v > best:
            best, end_query, end_ref = v, n - 1, j - 1
    # last column: target fully consumed
    last = H[m]
    for i in range(1, n + 1):
        v = last[i]
        if v > best:
            best, end_query, end_ref = v, i - 1, m - 1
        elif v == best and end_ref == m - 1 and i - 1 < end_query:
            end_query = i - 1
    return best, end_query, end_ref


def sg_stats(s1, s2, gap_open, gap_extend, score, alphabet="ATGCNX", rule="parasail", r1="striped"):
    """-> (score, end_query, end_ref, matches, length): the same alignment with the number of matches and of alignment
    columns along ONE optimal path.  `rule` is the switch shared with oracle/qcat_oracle.c qo_sg_stats and the device
    kernel k_sg_align (include/qcat_hip.h QCAT_STATS_*):
      "parasail" -- parasail 2.x's *_stats_striped_* kernels as recalled: on ties the diagonal, then F (the gap that
                    consumes a QUERY letter), then E (a target letter); a match = equal MAPPED codes over `alphabet` + '*'
                    (letters outside the alphabet all map to '*', case-insensitively); a gap is opened only when strictly
                    better than extended;
      "round3"   -- diagonal, E, F and "the same letter" (what round 3 shipped).
    Parity with parasail is UNPINNED for these two numbers (parasail is absent here) and NOTHING on the scanner paths
    consumes them: find_highest_scoring_barcode returns the score in their place, qcat/scanner_base.py:141."""
    assert rule in ("parasail", "round3")

    def code(ch):
        i = alphabet.find(ch.upper())
        return i if i >= 0 else len(alphabet)

    def same(a, b):
        if rule == "round3":
            return a.upper() == b.upper() and "ATGCNX".find(a.upper()) >= 0 or (a.upper() == b.upper())
        return code(a) == code(b)
    n, m = len(s1), len(s2)
    best, end_query, end_ref = sg(s1, s2, gap_open, gap_extend, score, rule=r1)
    # recompute with statistics carried along (small inputs only: the simple-mode fixtures)
    H = [[0] * (n + 1) for _ in range(m + 1)]
    E = [[NEG] * (n + 1) for _ in range(m + 1)]
    F = [[NEG] * (n + 1) for _ in range(m + 1)]
    MH = [[(0, 0)] * (n + 1) for _ in range(m + 1)]
    ME = [[(0, 0)] * (n + 1) for _ in range(m + 1)]
    MF = [[(0, 0)] * (n + 1) for _ in range(m + 1)]
    for j in range(1, m + 1):
        b = s2[j - 1]
        for i in range(1, n + 1):
            e_ext, e_open = E[j - 1][i] - gap_extend, H[j - 1][i] - gap_open
            if e_open > e_ext:
                E[j][i], ME[j][i] = e_open, (MH[j - 1][i][0], MH[j - 1][i][1] + 1)
            else:
                E[j][i], ME[j][i] = e_ext, (ME[j - 1][i][0], ME[j - 1][i][1] + 1)
            f_ext, f_open = F[j][i - 1] - gap_extend, H[j][i - 1] - gap_open
            if f_open > f_ext:
                F[j][i], MF[j][i] = f_open, (MH[j][i - 1][0], MH[j][i - 1][1] + 1)
            else:
                F[j][i], MF[j][i] = f_ext, (MF[j][i - 1][0], MF[j][i - 1][1] + 1)
            d = H[j - 1][i - 1] + score(s1[i - 1], b)
            dm = (MH[j - 1][i - 1][0] + (1 if same(s1[i - 1], b) else 0), MH[j - 1][i - 1][1] + 1)
            h, mh = d, dm
            if rule == "round3":
                if E[j][i] > h:
                    h, mh = E[j][i], ME[j][i]
                if F[j][i] > h:
                    h, mh = F[j][i], MF[j][i]
            else:
                if F[j][i] > h:
                    h, mh = F[j][i], MF[j][i]
                if E[j][i] > h:
                    h, mh = E[j][i], ME[j][i]
            H[j][i], MH[j][i] = h, mh
    matches, length = MH[end_ref + 1][end_query + 1]
    assert H[end_ref + 1][end_query + 1] == best
    return best, end_query, end_ref, matches, length


def sg_unique_path(s1, s2, gap_open, gap_extend, score):
    """-> (n_optimal, (score, end_query, end_ref, matches, length)): the number of DISTINCT optimal alignments (capped at 2)
    and -- meaningful when that number is 1 -- the statistics of the only one.

    An alignment is a sequence of columns (diagonal / gap in the target / gap in the query) from a cell of the first row or
    first column to a cell of the last row or last column.  Counted over three states per cell (ends in a diagonal step D, in
    a gap that consumes a target letter E, in a gap that consumes a query letter F) with E -> E and F -> F as the ONLY
    extension moves, so that every column sequence is one state path and is counted once (the H-based recurrence of sg()
    reaches "open from an H that came from E" as a second derivation of the same columns when open == extend).
    When the count is 1 there is no tie anywhere -- not among the border cells that end the alignment, not along the path --
    and (matches, length) are facts of the inputs: every correct implementation has to report them, whatever its tie order.
    A match here = the same letter; callers keep to A, C, G, T, where every rule of sg_stats agrees on what a match is."""
    assert gap_open >= gap_extend
    n, m = len(s1), len(s2)
    # per cell and state: (value, count, matches, length); the boundary is a start: value 0, one (empty) alignment
    start = (0, 1, 0, 0)
    none = (NEG, 0, 0, 0)

    def merge(cands):
        best = max(c[0] for c in cands)
        if best <= NEG // 2:
            return none
        tied = [c for c in cands if c[0] == best and c[1] > 0]
        cnt = min(2, sum(c[1] for c in tied))
        return (best, cnt, tied[0][2], tied[0][3])
    D = [[none] * (n + 1) for _ in range(m + 1)]
    E = [[none] * (n + 1) for _ in range(m + 1)]
    F = [[none] * (n + 1) for _ in range(m + 1)]
    B = [[start if (i == 0 or j == 0) else none for i in range(n + 1)] for j in range(m + 1)]

    def states(j, i):
        return (B[j][i], D[j][i], E[j][i], F[j][i])
    for j in range(1, m + 1):
        for i in range(1, n + 1):
            same = 1 if s1[i - 1].upper() == s2[j - 1].upper() else 0
            w = score(s1[i - 1], s2[j - 1])
            D[j][i] = merge([(v + w, c, a + same, l + 1) for (v, c, a, l) in states(j - 1, i - 1)])
            b, d, e, f = states(j - 1, i)
            E[j][i] = merge([(e[0] - gap_extend, e[1], e[2], e[3] + 1)] + [(v - gap_open, c, a, l + 1) for (v, c, a, l) in (b, d, f)])
            b, d, e, f = states(j, i - 1)
            F[j][i] = merge([(f[0] - gap_extend, f[1], f[2], f[3] + 1)] + [(v - gap_open, c, a, l + 1) for (v, c, a, l) in (b, d, e)])
    border = [(j, n) for j in range(1, m + 1)] + [(m, i) for i in range(1, n)]
    ends = []
    for (j, i) in border:
        h = merge(list(states(j, i)))
        ends.append((h, i - 1, j - 1))
    best = max(h[0] for h, _i, _j in ends)
    tied = [(h, i, j) for h, i, j in ends if h[0] == best]
    total = min(2, sum(h[1] for h, _i, _j in tied))
    h, i, j = tied[0]
    return total, (best, i, j, h[2], h[3])
