"""bench.py --gpus N starts its N ranks itself ("what's missing" #1): the parent never touches the GPU, every child
is one rank of plhip_group_by_agg_sharded, rank 0's single JSON line is relayed.  On a one-GPU box the ranks share device 0 and
the exchange goes through the host (gloo) transport (PLHIP_BENCH_SHARE_GPU=1); everything else -- local pass, pack, regroup,
finalize, the communicator's own rank count in "n_gpus", the parity leg against the oracle on the shards' concatenated
rows -- is what runs on N GPUs over RCCL (group_by_partitioned.rs:66-113, 320-360 is the reference counterpart)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _bench(*argv, env=None, timeout=420):
    e = dict(os.environ, PLHIP_BENCH_SHARE_GPU="1", **(env or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-u", str(ROOT / "bench.py"), *argv], env=e, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, f"bench.py {' '.join(argv)} -> {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"exactly ONE JSON line on stdout, got {len(lines)}: {r.stdout[:500]}"
    return json.loads(lines[0])


def test_bench_gpus_2_launches_two_ranks():
    line = _bench("--gpus", "2", "--rows", "1000000", "--steps", "3", "--warmup", "1")
    assert line["n_gpus"] == 2, line["n_gpus"]
    assert line["config"]["rows_per_gpu"] == 1_000_000 and line["scaling"] == "weak"
    assert line["parity"]["ok"], line["parity"]
    assert "oracle" in line["parity"]["against"]
    assert sum(line["parity"]["groups_per_rank"]) == line["config"]["groups"]
    assert line["shuffle"]["bytes_sent_per_rank"] > 0 and line["shuffle"]["ms"] is not None
    assert line["per_gpu"]["algorithmic_GBps"] > 0
    assert line["value"] == pytest.approx(2 * 1_000_000 / (line["ms_per_step"] * 1e-3), rel=1e-6)


def test_bench_gpus_3_strong_scaling_hashed_protocol():
    # config 4's shape in small: --total-rows splits the job's rows over the ranks; --no-dense takes the hashed (slot) protocol
    line = _bench("--gpus", "3", "--total-rows", "3000000", "--steps", "2", "--warmup", "1", "--no-dense")
    assert line["n_gpus"] == 3 and line["scaling"] == "strong" and line["config"]["rows_per_gpu"] == 1_000_000
    assert line["parity"]["ok"], line["parity"]
    assert line["shuffle"]["protocol"].startswith("slots")


def test_bench_gpus_2_many_codes_keep_the_dense_exchange():
    # config 4's regime in small: more ids than eight LDS ranges hold -> the local pass range-partitions (dense_part), the
    # exchange stays the dense one (one host wait)
    line = _bench("--gpus", "2", "--total-rows", "20000000", "--steps", "2", "--warmup", "1")
    assert line["n_gpus"] == 2 and line["config"]["groups"] == 200_000
    assert line["config"]["strategy"] == "dense_part", line["config"]["strategy"]
    assert line["shuffle"]["protocol"].startswith("dense"), line["shuffle"]
    assert line["parity"]["ok"], line["parity"]


def test_bench_rejects_a_world_size_that_contradicts_gpus():
    e = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4"], env=e, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
