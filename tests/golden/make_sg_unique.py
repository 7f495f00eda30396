"""Generator of tests/golden/sg_unique_vectors.json: alignments whose optimal path is UNIQUE (sg_independent.sg_unique_path
counts exactly one optimal alignment -- no tie among the border cells, none along the path).  For these the five numbers
(score, end_query, end_ref, matches, length) follow from the inputs alone: they pin `matches` / `length` of the statistics
kernels (qcat_sg_align, oracle qo_sg_stats) to mathematics instead of to a recollection of parasail's tie order, which no
reference output holds.  No reference code runs here.   python tests/golden/make_sg_unique.py"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import sg_independent as si                       # noqa: E402
from qcat_amd import config as qconfig             # noqa: E402


def main():
    cfg = qconfig.qcatConfig()
    rng = random.Random(505)
    cases, tried, perfect = [], 0, 0
    mats = {"adapter": cfg.matrix.table, "barcode": cfg.matrix_barcode.table}
    while len(cases) < 400:
        tried += 1
        name = rng.choice(sorted(mats))
        score = si.scorer_from_table7(mats[name])
        L, M = rng.randrange(4, 120), rng.randrange(4, 60)
        s2 = "".join(rng.choice("ACGT") for _ in range(M))
        s1 = "".join(rng.choice("ACGT") for _ in range(L))
        kind = rng.random()
        if kind < 0.8 and L > M:                    # the target with substitutions / indels inside the query
            body = []
            for c in s2:
                r = rng.random()
                if r < 0.07:
                    continue
                if r < 0.14:
                    body.append(rng.choice("ACGT"))
                body.append(c if rng.random() > 0.10 else rng.choice("ACGT"))
            p = rng.randrange(0, L - M + 1)
            s1 = s1[:p] + "".join(body) + s1[p + M:]
        elif kind < 0.9:                            # the query ends inside the target (target's end free)
            s1 = s1[:rng.randrange(1, L)] + s2[:rng.randrange(3, M)]
        go, ge = rng.choice([(2, 2), (1, 1), (3, 1), (5, 2)])
        total, st = si.sg_unique_path(s1, s2, go, ge, score)
        if total != 1:
            continue
        if st[3] == st[4] and perfect >= 80:         # (at most 80 paths without a mismatch or a gap)
            continue
        perfect += st[3] == st[4]
        assert st[:3] == si.sg(s1, s2, go, ge, score)
        for rule in ("parasail", "round3"):
            assert tuple(si.sg_stats(s1, s2, go, ge, score, rule=rule)) == st, (s1, s2, go, ge, rule)
        cases.append({"matrix": name, "query": s1, "target": s2, "open": go, "extend": ge, "score": st[0], "end_query": st[1],
                      "end_ref": st[2], "matches": st[3], "length": st[4]})
    imperfect = sum(1 for c in cases if c["length"] != c["matches"])
    out = {"about": "alignments with exactly one optimal path (sg_independent.sg_unique_path): statistics pinned by uniqueness",
           "tried": tried, "cases": cases}
    with open(os.path.join(HERE, "sg_unique_vectors.json"), "w") as f:
        json.dump(out, f, indent=0)
    print(len(cases), "unique of", tried, "tried;", imperfect, "with a mismatch or a gap on the path")


if __name__ == "__main__":
    main()
