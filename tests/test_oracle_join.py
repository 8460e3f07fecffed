"""The oracle's hash join (orc_hash_join: single_keys_inner.rs / single_keys_left.rs restated) against the known answers the
reference's own join tests hold (tests/golden/reference_join_known_answers.json) and against a brute-force pair list."""
import json
import os

import numpy as np

from polars_amd.column import Column, column_from_list

HERE = os.path.dirname(os.path.abspath(__file__))
NP = {"i8": np.int8, "i16": np.int16, "i32": np.int32, "i64": np.int64}


def gold_cases():
    return json.load(open(os.path.join(HERE, "golden", "reference_join_known_answers.json")))["cases"]


def check_case(case, dt, join_fn):
    """join_fn(left key Column, right key Column, how) -> (left_idx, right_idx)"""
    on = case["on"]
    lk, rk = column_from_list(on, case["left"][on], NP[dt]), column_from_list(on, case["right"][on], NP[dt])
    li, ri = join_fn(lk, rk, case["how"])
    if "expected_rows" in case:
        assert len(li) == len(ri) == case["expected_rows"], case["name"]
        return
    got = {}
    for name, vals in case["left"].items():
        got[name] = [None if i == 0xFFFFFFFF else vals[i] for i in li]
        if name == on and case["how"] == "outer":  # coalesced key
            got[name] = [case["right"][on][j] if i == 0xFFFFFFFF else vals[i] for i, j in zip(li, ri)]
    for name, vals in case["right"].items():
        if name != on and case["how"] not in ("semi", "anti"):
            got[name] = [None if i == 0xFFFFFFFF else vals[i] for i in ri]
    assert got == case["expected"], (case["name"], dt, got)


def test_reference_known_answers(oracle):
    for case in gold_cases():
        for dt in case["dtypes"]:
            check_case(case, dt, oracle.hash_join)


def brute(lk, lm, rk, rm, how):
    """Pairs in the reference's order: the longer relation (inner; the right one on a tie) or the left one (left join) probes."""
    if how in ("semi", "anti"):
        keep = [any((lm[i] and rm[j] and lk[i] == rk[j]) or (not lm[i] and not rm[j]) for j in range(len(rk))) for i in range(len(lk))]
        return [i for i in range(len(lk)) if keep[i] == (how == "semi")]
    swapped = how == "inner" and not (len(lk) > len(rk))
    pk, pm, bk, bm = (rk, rm, lk, lm) if swapped else (lk, lm, rk, rm)
    out = []
    for i in range(len(pk)):
        hits = [j for j in range(len(bk)) if (pm[i] and bm[j] and pk[i] == bk[j]) or (not pm[i] and not bm[j])]
        if not hits and how == "left":
            out.append((i, 0xFFFFFFFF))
        out += [((j, i) if swapped else (i, j)) for j in hits]
    return out


def test_against_brute_force(oracle):
    rng = np.random.default_rng(2)
    for how in ("inner", "left", "semi", "anti"):
        for nl, nr in ((0, 5), (5, 0), (40, 40), (120, 70), (70, 120)):
            lk, rk = rng.integers(-6, 6, nl).astype(np.int16), rng.integers(-9, 3, nr).astype(np.int16)
            lm, rm = rng.random(nl) >= 0.15, rng.random(nr) >= 0.15
            li, ri = oracle.hash_join(Column("k", lk, lm), Column("k", rk, rm), how)
            if how in ("semi", "anti"):
                assert li.tolist() == brute(lk, lm, rk, rm, how), (how, nl, nr)
            else:
                assert list(zip(li.tolist(), ri.tolist())) == brute(lk, lm, rk, rm, how), (how, nl, nr)


def test_multi_key_against_brute_force(oracle):
    rng = np.random.default_rng(4)
    for how in ("inner", "left", "semi", "anti"):
        for nl, nr in ((0, 4), (30, 30), (90, 50), (50, 90)):
            def side(n):
                return rng.integers(0, 4, n).astype(np.int16), rng.integers(0, 3, n).astype(np.float32), rng.random(n) >= 0.2
            la, lb, lm = side(nl)
            ra, rb, rm = side(nr)
            lt = [(int(a), float(b) if m else None) for a, b, m in zip(la, lb, lm)]
            rt = [(int(a), float(b) if m else None) for a, b, m in zip(ra, rb, rm)]
            if how in ("semi", "anti"):
                want = [i for i in range(nl) if (lt[i] in rt) == (how == "semi")]
            else:
                swapped = how == "inner" and not (nl > nr)
                pt, bt = (rt, lt) if swapped else (lt, rt)
                want = []
                for i, t in enumerate(pt):
                    hits = [j for j, u in enumerate(bt) if u == t]
                    if not hits and how == "left":
                        want.append((i, 0xFFFFFFFF))
                    want += [((j, i) if swapped else (i, j)) for j in hits]
            li, ri = oracle.hash_join_multi([Column("a", la), Column("b", lb, lm)], [Column("a", ra), Column("b", rb, rm)], how)
            got = li.tolist() if how in ("semi", "anti") else list(zip(li.tolist(), ri.tolist()))
            assert got == want, (how, nl, nr)
