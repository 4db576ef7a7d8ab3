"""RCCL on the hand-off path with ONE rank (-m gpu): everything N > 1 has so far run over gloo, and RCCL only
through bench.py's 1-rank launch.  Here the `nccl` backend carries real hand-offs: `ChunkHandoff(loopback=True)` sends every
chunk's anchors to its own rank THROUGH the transport -- a grouped RCCL send + recv on the side stream behind the deferred-issue
bookkeeping of the ready handshake -- with the gloo control group created next to the RCCL group (the mixed-backend set-up an
8-GPU run has), the device all-gather of the chunk results, and the status header read back with .tolist().  gloo has no pair to
oneself, so the announcement itself is delivered in-process (the 2-rank gloo tests cover the message).  Runs in a child process
under a timeout: a transport that hangs must fail this test, not the session.
Reference: Wan_fps_inference_parallel_4gpu_5-60s.py:188-381 (wrap-around: chunk c + lanes runs on the rank chunk c ran on)."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent('''
    import datetime, json, os, sys
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    sys.path.insert(0, sys.argv[1])
    from mmpl_amd.handoff import ChunkHandoff, run_chunk_wavefront
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = sys.argv[2]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"), timeout=datetime.timedelta(seconds=120))
    shape = (1, 8, 16, 90, 160)                                    # the 720p anchor tensor: 3.7 MB
    ho = ChunkHandoff(shape, "cuda:0", loopback=True, control_timeout=datetime.timedelta(seconds=120))
    assert ho.backend == "nccl" and ho.handshake and ho.loopback
    assert dist.get_backend(ho._ctl) == "gloo"                      # the control group lives next to the RCCL group
    dist.barrier(group=ho._ctl)
    deferred_at_sink = {}

    def make_chunk(c, initial, sink):
        a = torch.full((512, 512), 0.5, device="cuda:0")
        for _ in range(20):
            a = (a @ a) * (1.0 / 256.0)                            # compute stream busy while the side stream carries the hand-off
        base = torch.full(shape, float(c + 1), dtype=torch.bfloat16, device="cuda:0")
        if initial is not None:
            base = base + initial.float().mean().to(torch.bfloat16)
        sink(base)
        base.zero_()                                               # the caller goes on writing its buffer: the hand-off must not alias it
        deferred_at_sink[c] = len(ho._deferred)
        return torch.full((1, 2, 16, 8, 8), float(c + 1), dtype=torch.bfloat16, device="cuda:0") + (0 if initial is None else initial.float().mean().to(torch.bfloat16))

    res = run_chunk_wavefront(make_chunk, 4, ho, to_initial=lambda t: t[:, :2].clone())
    torch.cuda.synchronize()
    vals = [float(t.float().mean()) for t in res]
    st = {c: {k: float(v) for k, v in s.items()} for c, s in ho.stats.items()}
    print("RESULT " + json.dumps({"vals": vals, "deferred_at_sink": deferred_at_sink, "stats": st}), flush=True)
    dist.destroy_process_group()
''')


def test_rccl_handoff_loopback_with_gloo_control_group(tmp_path):
    import json
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    p = subprocess.run([sys.executable, str(script), ROOT, str(port)], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    # the chained values of tests/test_handoff_gloo.py: v0 = 1, v1 = 2 + 1, v2 = 3 + 3, v3 = 4 + 6 -- every hand-off arrived intact
    assert r["vals"] == [1.0, 3.0, 6.0, 10.0]
    # each hand-off was deferred at its sink (its consumer -- this rank's NEXT chunk -- had not announced itself) ...
    assert r["deferred_at_sink"] == {"0": 1, "1": 1, "2": 1, "3": 0}
    # ... and went through the transport when it did: announced <= issued <= received
    for c in ("1", "2", "3"):
        st = r["stats"][c]
        assert st["t_sink"] <= st["t_ready"] + 1e-3 <= st["t_issued"] + 2e-3 <= st["t_recv_done"] + 3e-3, (c, st)
