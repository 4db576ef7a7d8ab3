"""Two host threads driving two engines concurrently (the reference's own model is one Python thread per pipeline,
Wan_fps_inference_parallel_4gpu_20s.py:229-256): the library keeps no process-global launch state -- kernel attributes
are cached per device under a mutex (csrc/device_state.hip), errors and the optional profiler are per thread -- so
concurrent first calls from two threads must work and give the single-threaded result bit for bit (-m gpu)."""
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


def _one_forward(seed, out, idx, barrier):
    from mmpl_amd.dit import DitEngine
    from mmpl_amd.synthetic import WAN_CONFIGS, dit_state_dict, philox_normal
    try:
        cfg = WAN_CONFIGS["tiny"]
        stream = torch.cuda.Stream(device="cuda:0")
        with torch.cuda.stream(stream):
            eng = DitEngine(cfg, 16, 24, "cuda:0")
            eng.load_state_dict(dit_state_dict(cfg, seed=seed))
            ctx = philox_normal([512, cfg["text_dim"]], seed + 1).cuda()
            x = philox_normal([7, 16, 16, 24], seed + 2).cuda()
            t = torch.full([7], 500.0, device="cuda:0")
            kc, vc = eng.new_kv_cache(15)
            ck, cv = eng.precompute_context(ctx)
            frames = [2, 3, 10, 11, 12, 19, 20]
            ws = [2, 3, 10, 11, 12, 13, 14]
            if barrier is not None:
                barrier.wait()                    # both threads make their first kernel launches at the same time
            ys = [eng.forward(x, t, frames, ws, ws, kc, vc, ck, cv).clone() for _ in range(3)]
        stream.synchronize()
        out[idx] = [y.cpu() for y in ys]
    except Exception as e:                        # surface the failure in the main thread
        out[idx] = e


def test_two_threads_two_engines_match_single_thread():
    ref = [None, None]
    _one_forward(11, ref, 0, None)
    _one_forward(23, ref, 1, None)
    assert not isinstance(ref[0], Exception) and not isinstance(ref[1], Exception), ref
    got = [None, None]
    b = threading.Barrier(2)
    th = [threading.Thread(target=_one_forward, args=(s, got, i, b)) for i, s in enumerate((11, 23))]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    for i in range(2):
        assert not isinstance(got[i], Exception), got[i]
        for y, r in zip(got[i], ref[i]):
            assert torch.equal(y, r)
