"""The oracle's restatement of the Utf8 group-by key (orc_utf8_group_codes, into_groups.rs:232-290) against the reference's
printed docstring results (tests/golden/reference_utf8_known_answers.json) and against a plain Python dict on random strings."""
import json
import os

import numpy as np

from polars_amd.column import Column
from polars_amd.strings import Utf8Column

HERE = os.path.dirname(os.path.abspath(__file__))


def _codes(oracle, vals):
    col = Utf8Column.from_list("s", vals)
    return oracle.utf8_group_codes(col.offsets, col.data, col.valid_mask())


def test_reference_docstring_answers(oracle):
    gold = json.load(open(os.path.join(HERE, "golden", "reference_utf8_known_answers.json")))
    fr = gold["frame"]
    for case in gold["cases"]:
        codes, first = _codes(oracle, fr["a"])
        keys = [Column("a", codes)] + [Column(k, np.array(fr[k], np.int64)) for k in case["by"][1:]]
        vcol, kind = case["agg"]
        want, _ = oracle.group_by_agg(keys, [Column(vcol, np.array(fr[vcol], np.int64))], [(0, kind)])
        rows = []
        for j in range(len(want[0][1])):
            row = [fr["a"][first[int(want[0][1][j])]]] + [int(want[i][1][j]) for i in range(1, len(want))]
            rows.append(row)
        assert sorted(rows) == sorted(case["rows"]), case


def test_codes_number_groups_by_first_occurrence(oracle):
    rng = np.random.default_rng(1)
    pool = ["", "a", "ab", "b" * 9, "b" * 8 + "c", "é", None]
    vals = [pool[i] for i in rng.integers(0, len(pool), 5000)]
    codes, first = _codes(oracle, vals)
    seen = {}
    for i, v in enumerate(vals):
        g = seen.setdefault(v, len(seen))
        assert codes[i] == g
        if first[g] == i:
            assert vals.index(v) == i
    assert len(first) == len(seen) and [vals[f] for f in first] == list(seen)
