"""Synthetic h2oai db-benchmark group-by data, generated directly in HBM.

Shape and distributions follow the reference's benchmark harness
(py-polars/tests/benchmark/groupby-datagen.R:17-28 and run_h2oai_benchmark.py:21-35):
    id1,id2 (categorical -> u32 codes), id4,id5 (i32)  ~ U{1..K}
    id3 (categorical -> u32 code), id6 (i32)           ~ U{1..N/K}
    v1 ~ U{1..5} (i32), v2 ~ U{1..15} (i32), v3 = round(U(0,100), 6) (f64)
K = 100, seed 108 (same literal as the R script; the random streams differ, so the reference's checksums do
not apply).  `wide=True` gives the i64/f64 variant of BASELINE.json config 3.  torch is used for the
device RNG and memory only.
"""
from __future__ import annotations

from . import _ffi
from .column import Column


def generate(n: int, k: int = 100, seed: int = 108, device="cuda:0", wide: bool = False, columns=None, na: bool = False,
             n_for_ids: int | None = None) -> dict:
    """`na`: db-benchmark's NA variant (groupby-datagen.R:30-44): 5 % of the DISTINCT values of every id column are
    nulled (all rows that hold one of them), 5 % of the rows of every value column are NA; the nulls travel as bool
    masks under the key `<name>__valid` (True = valid).  `n_for_ids`: the N that sets the id cardinalities (N / K) when
    this call generates one shard of a larger job."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    idt = torch.int64 if wide else torch.int32
    want = set(columns) if columns else {"id1", "id2", "id3", "id4", "id5", "id6", "v1", "v2", "v3"}
    out = {}

    def ri(name, lo, hi):
        # every column consumes its stream whether wanted or not, so a subset is identical to the full table
        t = torch.randint(lo, hi + 1, (n,), generator=g, device=device, dtype=idt)
        if name in want:
            out[name] = t
        else:
            del t

    nk = max((n_for_ids if n_for_ids is not None else n) // k, 1)
    ri("id1", 1, k); ri("id2", 1, k); ri("id3", 1, nk); ri("id4", 1, k); ri("id5", 1, k); ri("id6", 1, nk)
    ri("v1", 1, 5); ri("v2", 1, 15)
    v3 = torch.rand((n,), generator=g, device=device, dtype=torch.float64)
    if "v3" in want:
        out["v3"] = torch.round(v3 * 1e8) / 1e6
    if na:
        # groupby-datagen.R:30-44: `nas = 5`: for every id column sample 5 % of its distinct values and set the rows that hold
        # them to NA; for every value column set a 5 % sample of the rows to NA.  Its own generator: the value streams
        # above are those of the variant without NAs.
        gn = torch.Generator(device=device)
        gn.manual_seed(seed + 7919)
        card = {"id1": k, "id2": k, "id3": nk, "id4": k, "id5": k, "id6": nk}
        for name in ("id1", "id2", "id3", "id4", "id5", "id6"):
            dead = torch.rand((card[name] + 1,), generator=gn, device=device) < 0.05  # indexed by the id value 1..card
            if name in out:
                out[name + "__valid"] = ~dead[out[name].to(torch.int64)]
        for name in ("v1", "v2", "v3"):
            m = torch.rand((n,), generator=gn, device=device) >= 0.05
            if name in out:
                out[name + "__valid"] = m
    return out


def to_columns(tensors: dict, wide: bool = False, k: int = 100, n: int | None = None) -> dict[str, Column]:
    """id1..id3 are Categorical in the reference harness (run_h2oai_benchmark.py:21-35): what reaches the group-by is
    their physical u32 code column plus the rev-map, whose length -- K, K, N/K categories -- the engine knows
    without touching the data.  The codes are 0-based (`id - 1`, rewritten IN PLACE in `tensors` so that whoever
    checks the result sees the same column; the i32 buffer is passed with a u32 dtype code, the bits are identical)."""
    cols = {}
    for name, t in tensors.items():
        if name.endswith("__valid"):
            continue
        mask = tensors.get(name + "__valid")
        if not wide and name in ("id1", "id2", "id3"):
            t.sub_(1)
            ncat = k if name in ("id1", "id2") else max((n if n is not None else t.numel()) // k, 1)
            cols[name] = Column(name, t, mask, dtype=_ffi.U32).set_categories(ncat)
        else:
            cols[name] = Column(name, t, mask)
    return cols
