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


def generate(n: int, k: int = 100, seed: int = 108, device="cuda:0", wide: bool = False, columns=None) -> dict:
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

    nk = max(n // k, 1)
    ri("id1", 1, k); ri("id2", 1, k); ri("id3", 1, nk); ri("id4", 1, k); ri("id5", 1, k); ri("id6", 1, nk)
    ri("v1", 1, 5); ri("v2", 1, 15)
    v3 = torch.rand((n,), generator=g, device=device, dtype=torch.float64)
    if "v3" in want:
        out["v3"] = torch.round(v3 * 1e8) / 1e6
    return out


def to_columns(tensors: dict, wide: bool = False, k: int = 100, n: int | None = None) -> dict[str, Column]:
    """id1..id3 are Categorical in the reference harness (run_h2oai_benchmark.py:21-35): what reaches the group-by is
    their physical u32 code column plus the rev-map, whose length -- K, K, N/K categories -- the engine knows
    without touching the data.  The codes are 0-based (`id - 1`, rewritten IN PLACE in `tensors` so that whoever
    checks the result sees the same column; the i32 buffer is passed with a u32 dtype code, the bits are identical)."""
    cols = {}
    for name, t in tensors.items():
        if not wide and name in ("id1", "id2", "id3"):
            t.sub_(1)
            ncat = k if name in ("id1", "id2") else max((n if n is not None else t.numel()) // k, 1)
            cols[name] = Column(name, t, dtype=_ffi.U32).set_categories(ncat)
        else:
            cols[name] = Column(name, t)
    return cols
