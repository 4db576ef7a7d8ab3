"""bench.py's N > 1 code paths, functionally, on a 1-GPU box (-m gpu): `--gpus 2` makes bench.py start two ranks itself; with
`--dist-backend gloo` they share cuda:0 and stage their exchanges through the host (two RCCL ranks cannot share a device), so
what runs is the real lane layout, the anchor hand-off p2p, the CFG pair's per-step exchange with a hipGraph per forward, the
max-over-ranks timing and rank 0's single JSON line.  The driver's 8-GPU run uses the same code with the nccl backend."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--model", "small", "--res", "tiny",
           "--steps", "8", "--warmup", "4", "--no-cpu-baseline", "--no-vae"] + extra
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_two_chunk_lanes():
    r = _run([])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and "functional_only" in r
    assert "chunk-per-rank x2" in r["config"]["parallelism"] and "hand-off" in r["config"]["parallelism"]
    assert r["value"] > 0 and r["value_independent_chunks"] >= r["value"] and len(r["sec_per_denoise_step_by_stage"]) == 4
    assert r["roofline"]["achieved"] > 0


def test_cfg_pair_with_graphed_forwards():
    r = _run(["--cfg-split"])
    assert r["n_gpus"] == 2 and "CFG split" in r["config"]["parallelism"]
    assert "hipGraph replay per forward" in r["config"]["timed_path"]
    assert r["value"] > 0 and r["roofline"]["achieved"] > 0
    # each rank computes ONE branch per step
    assert r["achieved_pflops_per_gpu"] > 0
