"""CFG-split lanes on CPU (gloo, world_size 4 = 2 chunk lanes x (cond, uncond)): group layout, the per-step flow
exchange, role-0 broadcasts, and the anchor hand-off between lane heads through global-rank translation."""
import datetime
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mmpl_amd.handoff import CfgPair, ChunkHandoff, wavefront_layout

SHAPE = (1, 3, 16, 4, 6)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _branch(role, lat, t):
    """stand-in for the cond / uncond DiT forward: any deterministic function of (branch, latents, t)."""
    return (torch.sin(lat.float() * (1.0 + role)) * (0.5 + t)).to(torch.bfloat16)


def _denoise(lat, steps, flows):
    for i in range(steps):
        c, u = flows(lat, i)
        lat = (lat.float() - 0.1 * (u.float() + 5.0 * (c.float() - u.float()))).to(torch.bfloat16)
    return lat


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=240))
    try:
        pair, heads, lay = CfgPair.build(world, "cpu", cfg_split=True)
        lane = lay["lane_of"][rank]
        assert pair.role == lay["role_of"][rank] == rank % 2 and pair.ranks == [2 * lane, 2 * lane + 1]
        # role 0 draws the noise; role 1 must end up with the same tensor
        g = torch.Generator().manual_seed(100 + rank)
        noise = pair.broadcast(torch.randn(2, 16, 4, 6, generator=g).to(torch.bfloat16))

        def flows(lat, i):
            mine = _branch(pair.role, lat, i / 4)
            both = torch.empty((2,) + tuple(lat.shape), dtype=lat.dtype)
            pair.exchange(mine, both)
            return both[0], both[1]

        lat = _denoise(noise.clone(), 4, flows)
        # lane heads chain the anchor hand-off (lane 0 -> lane 1) over the heads group
        got = None
        if pair.role == 0:
            ho = ChunkHandoff(SHAPE, "cpu", group=heads)
            assert ho.world == world // 2 and ho.rank == lane
            if lane == 0:
                ho.send(0, torch.full(SHAPE, 7.0))
            else:
                got = ho.recv(1)
            ho.flush()
        if lane == 1:                                       # ... and the head forwards it to its uncond partner
            got = pair.broadcast(got if pair.role == 0 else torch.empty(SHAPE, dtype=torch.bfloat16))
        torch.save({"noise": noise, "lat": lat, "got": got}, f"{out_path}.{rank}")
    finally:
        dist.destroy_process_group()


def test_layout():
    lay = wavefront_layout(8, True)
    assert lay["lanes"] == [[0, 1], [2, 3], [4, 5], [6, 7]] and lay["heads"] == [0, 2, 4, 6]
    assert lay["lane_of"] == [0, 0, 1, 1, 2, 2, 3, 3] and lay["role_of"] == [0, 1] * 4
    lay = wavefront_layout(4, False)
    assert lay["lanes"] == [[0], [1], [2], [3]] and lay["role_of"] == [0] * 4
    with pytest.raises(ValueError):
        wavefront_layout(3, True)


def test_cfg_pairs_two_lanes(tmp_path):
    out = str(tmp_path / "o")
    mp.spawn(_worker, args=(4, _free_port(), out), nprocs=4, join=True)
    r = [torch.load(f"{out}.{k}") for k in range(4)]
    for lane in range(2):
        a, b = r[2 * lane], r[2 * lane + 1]
        assert torch.equal(a["noise"], b["noise"])
        assert torch.equal(a["lat"], b["lat"])              # both ranks of a pair hold bit-identical latents
        # == the single-process loop that runs both branches itself
        want = _denoise(a["noise"].clone(), 4, lambda lat, i: (_branch(0, lat, i / 4), _branch(1, lat, i / 4)))
        assert torch.equal(a["lat"], want)
    assert not torch.equal(r[0]["noise"], r[2]["noise"])
    assert r[0]["got"] is None and r[1]["got"] is None
    assert torch.equal(r[2]["got"], torch.full(SHAPE, 7.0, dtype=torch.bfloat16)) and torch.equal(r[3]["got"], r[2]["got"])


def _wavefront_worker(rank, world, port, out_path, n_chunks):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=240))
    try:
        from mmpl_amd.handoff import run_chunk_wavefront
        pair, heads, lay = CfgPair.build(world, "cpu", cfg_split=True)
        lane, n_lanes = lay["lane_of"][rank], world // 2
        ho = ChunkHandoff(SHAPE, "cpu", group=heads) if pair.role == 0 else None
        log = []

        def make_chunk(c, initial, sink):
            # what pipeline.inference does in a CfgPair: role 0's initial latent overwrites role 1's placeholder, both ranks
            # compute the same chunk, only role 0 delivers the hand-off
            if initial is not None:
                initial = pair.broadcast(initial)
            base = torch.full(SHAPE, float(c + 1), dtype=torch.bfloat16)
            if initial is not None:
                base = base + initial.float().mean().to(torch.bfloat16)
            sink(base)
            log.append((c, None if initial is None else float(initial.float().mean())))
            return base[:, :2].clone()

        res = run_chunk_wavefront(make_chunk, n_chunks, ho, to_initial=lambda t: t[:, :2], pair=pair, lane=lane, n_lanes=n_lanes,
                                  initial_like=torch.empty((1, 2) + SHAPE[2:], dtype=torch.bfloat16))
        torch.save({"log": log, "res": res}, f"{out_path}.{rank}")
    finally:
        dist.destroy_process_group()


def test_wavefront_with_cfg_pairs_four_ranks(tmp_path):
    """run_chunk_wavefront over 2 lanes x (cond, uncond): 5 chunks; the uncond ranks follow their lane head chunk by chunk
    (same chunk ids, same initial latents) and only rank 0 gathers the results."""
    out = str(tmp_path / "w")
    mp.spawn(_wavefront_worker, args=(4, _free_port(), out, 5), nprocs=4, join=True)
    r = [torch.load(f"{out}.{k}") for k in range(4)]
    assert [c for c, _ in r[0]["log"]] == [0, 2, 4] and [c for c, _ in r[2]["log"]] == [1, 3]
    assert r[1]["log"] == r[0]["log"] and r[3]["log"] == r[2]["log"]          # uncond partners saw the same chunks / initial latents
    assert r[1]["res"] is None and r[2]["res"] is None and r[3]["res"] is None and len(r[0]["res"]) == 5
    # the dependency chain ran through both lanes: chunk c's anchors = c + 1 + mean(initial from chunk c-1)
    want, prev = [], None
    for c in range(5):
        v = float(c + 1) + (0.0 if prev is None else prev)
        v = float(torch.tensor(v).to(torch.bfloat16))
        want.append(v)
        prev = v
    got = [float(t.float().mean()) for t in r[0]["res"]]
    assert got == want


@pytest.mark.parametrize("n_chunks", [8, 9])
def test_wavefront_with_cfg_pairs_eight_ranks(tmp_path, n_chunks):
    """BASELINE configs[4]'s layout on CPU: world 8 = 4 chunk lanes x (cond, uncond) -- 4 pair groups + the 4-rank heads group + the
    world group + their gloo control groups, all built by CfgPair.build in every process.  8 chunks (two per lane) and 9 (chunk 8 wraps
    around onto lane 0 while it may still be busy with chunk 4: the reference's round-robin with a busy flag,
    Wan_fps_inference_parallel_4gpu_5-60s.py:188-381).  Every rank exits (mp.spawn joins all eight), the uncond partner of every lane
    follows its head chunk by chunk, the hand-offs arrive in dependency order and only rank 0 gathers."""
    world, n_lanes = 8, 4
    lay = wavefront_layout(world, True)
    assert lay["lanes"] == [[0, 1], [2, 3], [4, 5], [6, 7]] and lay["heads"] == [0, 2, 4, 6]
    out = str(tmp_path / "w8")
    mp.spawn(_wavefront_worker, args=(world, _free_port(), out, n_chunks), nprocs=world, join=True)
    r = [torch.load(f"{out}.{k}") for k in range(world)]
    for lane in range(n_lanes):
        head, partner = r[2 * lane], r[2 * lane + 1]
        assert [c for c, _ in head["log"]] == list(range(lane, n_chunks, n_lanes))          # chunk c on lane c mod 4, in order
        assert partner["log"] == head["log"]                                                  # same chunks, same initial latents
        assert (head["res"] is not None) == (lane == 0) and partner["res"] is None
    # the dependency chain ran through all four lanes and wrapped around: chunk c's anchors = c + 1 + mean(initial from chunk c - 1)
    want, prev = [], None
    for c in range(n_chunks):
        v = float(torch.tensor(float(c + 1) + (0.0 if prev is None else prev)).to(torch.bfloat16))
        want.append(v)
        prev = v
    assert len(r[0]["res"]) == n_chunks and [float(t.float().mean()) for t in r[0]["res"]] == want
    # ... and what every consumer was handed is exactly its predecessor's value (None only for chunk 0)
    seen = {c: init for k in range(0, world, 2) for c, init in r[k]["log"]}
    assert seen[0] is None and all(seen[c] == want[c - 1] for c in range(1, n_chunks))
