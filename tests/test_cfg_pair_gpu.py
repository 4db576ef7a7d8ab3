"""CFG-split pair on the real pipeline (-m gpu): two processes share cuda:0, each runs ONE branch (cond | uncond) of a
tiny T2V chunk and they exchange flow predictions every step through a CfgPair (gloo here -- two RCCL ranks cannot
share one device; the class stages device tensors through the host for gloo).  Both ranks must end with latents that
are bit-identical to each other and to the ordinary single-process pipeline."""
import datetime
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs():
    from mmpl_amd.synthetic import philox_normal
    from tests.test_pipeline_gpu import LAT
    noise = philox_normal([1, 21, 16, *LAT], 23)
    init = philox_normal([1, 2, 16, *LAT], 55)
    return noise, init


def _worker(rank, port, out_path):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=2, timeout=datetime.timedelta(seconds=120))
    try:
        from mmpl_amd.handoff import CfgPair
        from tests.test_pipeline_gpu import _setup
        pipe, *_ = _setup("t2v")
        pair, _, _ = CfgPair.build(2, "cuda:0", cfg_split=True)
        pipe.cfg_pair = pair
        noise, init = _inputs()
        if rank == 1:                                       # the uncond rank's own inputs are overwritten by role 0's
            noise, init = torch.zeros_like(noise), torch.zeros_like(init)
        torch.manual_seed(1000 + rank)                      # re-noise draws differ per rank -> must be broadcast
        got = []
        pipe.handoff_sink = lambda t: got.append(t.clone())
        _, lat1 = pipe.inference(noise.cuda(), ["a cat"], return_latents=True, decode=False)
        _, lat2 = pipe.inference(noise.cuda(), ["a cat"], initial_latent=init.cuda(), return_latents=True, decode=False)
        torch.cuda.synchronize()
        assert (pipe.kv_cache_pos is None) == (rank == 1) and (pipe.kv_cache_neg is None) == (rank == 0)
        torch.save({"lat1": lat1.cpu(), "lat2": lat2.cpu(), "n_handoff": len(got),
                    "handoff": got[0].cpu() if got else None}, f"{out_path}.{rank}")
    finally:
        dist.destroy_process_group()


def test_pair_matches_each_other_and_single_process(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "o")
    mp.spawn(_worker, args=(_free_port(), out), nprocs=2, join=True)
    a, b = (torch.load(f"{out}.{r}") for r in range(2))
    assert torch.equal(a["lat1"], b["lat1"]) and torch.equal(a["lat2"], b["lat2"])
    assert a["n_handoff"] == 2 and b["n_handoff"] == 0      # only the cond rank delivers the anchor hand-off
    # single process, both branches, with the re-noise draws of the pair's role 0 (its RNG seed)
    from tests.test_pipeline_gpu import _setup
    pipe, *_ = _setup("t2v")
    noise, init = _inputs()
    torch.manual_seed(1000)
    got = []
    pipe.handoff_sink = lambda t: got.append(t.clone())
    _, lat1 = pipe.inference(noise.cuda(), ["a cat"], return_latents=True, decode=False)
    _, lat2 = pipe.inference(noise.cuda(), ["a cat"], initial_latent=init.cuda(), return_latents=True, decode=False)
    assert torch.equal(lat1.cpu(), a["lat1"]) and torch.equal(lat2.cpu(), a["lat2"])
    assert torch.equal(got[0].cpu(), a["handoff"])
