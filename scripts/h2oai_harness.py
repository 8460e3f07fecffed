"""The reference's own benchmark harness (py-polars/tests/benchmark/run_h2oai_benchmark.py) written against the mirror API: the
group-by aggregation queries q1 .. q7 and q10 as they stand there (q8 -- top-2 per group behind a sort -- and q9 -- corr -- are not
aggregations of this path), on the harness's table shape (G1_1e7_1e2_5_0: 1e7 rows, K = 100, 5 % NAs, id1 .. id3 Categorical),
generated in HBM (polars_amd/h2oai.py: the random streams are not R's, so the harness's checksums do not apply -- every result is
checked against the oracle instead).  Like the harness: every query twice, wall time around `.collect()` (result frame on the
host), the second pass's total reported.  The reference's CI fails when ITS ten queries take more than 12 s on a GitHub runner
(run_h2oai_benchmark.py:316-320): a gate, not a published measurement.

    python scripts/h2oai_harness.py [--rows 10000000] [--no-check] > profiles/<name>.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import polars_amd as pa  # noqa: E402
from polars_amd import h2oai  # noqa: E402
from polars_amd.column import Column  # noqa: E402


def queries(x):
    c = pa.col
    return [
        ("q1", ["id1"], lambda: x.group_by("id1").agg(c("v1").sum().alias("v1_sum"))),
        ("q2", ["id1", "id2"], lambda: x.group_by(["id1", "id2"]).agg(c("v1").sum().alias("v1_sum"))),
        ("q3", ["id3"], lambda: x.group_by("id3").agg([c("v1").sum().alias("v1_sum"), c("v3").mean().alias("v3_mean")])),
        ("q4", ["id4"], lambda: x.group_by("id4").agg([c("v1").mean().alias("v1_mean"), c("v2").mean().alias("v2_mean"), c("v3").mean().alias("v3_mean")])),
        ("q5", ["id6"], lambda: x.group_by("id6").agg([c("v1").sum().alias("v1_sum"), c("v2").sum().alias("v2_sum"), c("v3").sum().alias("v3_sum")])),
        ("q6", ["id4", "id5"], lambda: x.group_by(["id4", "id5"]).agg([c("v3").median().alias("v3_median"), c("v3").std().alias("v3_std")])),
        # (pl.max("v1") - pl.min("v2")).alias("range_v1_v2"): the two aggregates come out of the query, the subtraction is the host's
        ("q7", ["id3"], lambda: x.group_by("id3").agg([c("v1").max().alias("v1_max"), c("v2").min().alias("v2_min")])),
        ("q10", ["id1", "id2", "id3", "id4", "id5", "id6"],
         lambda: x.group_by(["id1", "id2", "id3", "id4", "id5", "id6"]).agg([c("v3").sum().alias("v3"), c("v1").count().alias("count")])),
    ]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--no-check", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    tensors = h2oai.generate(a.rows, seed=108, device=dev, na=True)
    cols = h2oai.to_columns(tensors)
    x = pa.DataFrame(cols).lazy()
    qs = queries(x)
    passes = []
    frames = {}
    for p in range(2):
        times = {}
        t00 = time.perf_counter()
        for name, by, q in qs:
            t0 = time.perf_counter()
            out = q().collect()
            if name == "q7":
                v1, v2 = out.get_column("v1_max"), out.get_column("v2_min")
                m1, m2 = v1.valid_mask(), v2.valid_mask()
                rng_ = v1.to_numpy() - v2.to_numpy()
                out._range = (rng_, None if m1 is None and m2 is None else (np.ones(len(rng_), bool) if m1 is None else m1) & (np.ones(len(rng_), bool) if m2 is None else m2))
            times[name] = time.perf_counter() - t0
            frames[name] = out
        times["total"] = time.perf_counter() - t00
        passes.append(times)
        print(f"pass {p + 1}: " + ", ".join(f"{k} {v * 1e3:.2f} ms" for k, v in times.items()), file=sys.stderr, flush=True)
    line = {"what": "run_h2oai_benchmark.py's group-by aggregation queries through the mirror API, table resident in HBM, result frames on the host",
            "rows": a.rows, "na": True, "queries": [n for n, _, _ in qs], "not_run": {"q8": "sort + head(2) per group", "q9": "corr"},
            "first_pass_s": passes[0], "second_pass_s": passes[1],
            "shapes": {n: [frames[n].height, len(frames[n].columns)] for n in frames},
            "reference_gate": "its 10 queries' second pass must take <= 12 s on the CI runner (run_h2oai_benchmark.py:316-320)"}
    if not a.no_check:
        sys.path.insert(0, os.path.join(ROOT))
        from oracle import oracle as orc  # (checker only)
        from tests import util

        host = {}
        for n, t in tensors.items():
            if not n.endswith("__valid"):
                m = tensors.get(n + "__valid")
                dt = _ffi_dtype(cols[n])
                host[n] = Column(n, t.cpu().numpy().view(dt), None if m is None else m.cpu().numpy())
        checks = {}
        for name, by, _ in qs:
            got = frames[name]
            keys = [host[k] for k in by]
            t0 = time.perf_counter()
            if name == "q6":
                g = orc.Groups(keys, True, n_threads=8)
                want = [orc.take_keys(k, g) for k in keys] + [orc.agg_ext(host["v3"], "median", g), orc.agg_ext(host["v3"], "std", g)]
                want = [(v, None if m.all() else m) for v, m in want]
                got_np = [(got[c].to_numpy(), None if got[c].valid_mask() is None or got[c].valid_mask().all() else got[c].valid_mask()) for c in got.columns]
                util.assert_frames_match(got_np, want, len(by), rtol_f64=1e-11, what=name)
            else:
                spec = {"q1": [("v1", "sum")], "q2": [("v1", "sum")], "q3": [("v1", "sum"), ("v3", "mean")],
                        "q4": [("v1", "mean"), ("v2", "mean"), ("v3", "mean")], "q5": [("v1", "sum"), ("v2", "sum"), ("v3", "sum")],
                        "q7": [("v1", "max"), ("v2", "min")], "q10": [("v3", "sum"), ("v1", "count")]}[name]
                vnames = list(dict.fromkeys(v for v, _ in spec))
                want, _ = orc.group_by_agg(keys, [host[v] for v in vnames], [(vnames.index(v), k) for v, k in spec], n_threads=8)
                util.assert_frames_match(util.frame_to_np(got), util.oracle_to_np(want), len(by), rtol_f64=1e-11, what=name)
            checks[name] = {"ok": True, "oracle_s": round(time.perf_counter() - t0, 2)}
            print(f"{name}: matches the oracle ({checks[name]['oracle_s']} s)", file=sys.stderr, flush=True)
        line["parity"] = checks
    print(json.dumps(line))


def _ffi_dtype(col):
    from polars_amd import _ffi

    return _ffi.code_np_dtype(col.dtype)


if __name__ == "__main__":
    main()
