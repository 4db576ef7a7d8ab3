"""N>1 path on CPU: the chunk wavefront + anchor hand-off over torch.distributed (gloo, world_size 2)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mmpl_amd.handoff import ChunkHandoff, run_chunk_wavefront, stitch_chunks

SHAPE = (1, 8, 16, 6, 10)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_chunks, fail_at, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    ho = ChunkHandoff(SHAPE, "cpu")
    log = []

    def make_chunk(c, initial, sink):
        # chunk c's "anchors" are a deterministic function of c and of what it received (chains the dependency)
        base = torch.full(SHAPE, float(c + 1), dtype=torch.bfloat16)
        if initial is not None:
            base = base + initial.float().mean().to(torch.bfloat16)
        if c == fail_at:
            raise ValueError("boom")
        sink(base)
        log.append((c, None if initial is None else float(initial.float().mean())))
        return base[:, :2].clone()

    try:
        res = run_chunk_wavefront(make_chunk, n_chunks, ho, to_initial=lambda t: t[:, :2])
        torch.save({"rank": rank, "log": log, "res": res, "err": None}, f"{out_path}.{rank}")
    except Exception as e:
        torch.save({"rank": rank, "log": log, "res": None, "err": repr(e)}, f"{out_path}.{rank}")
        import time
        time.sleep(3)          # let the FAILED header drain before this rank tears its sockets down
    finally:
        dist.destroy_process_group()


def _run(tmp_path, n_chunks, fail_at=-1, world=2):
    out = str(tmp_path / "out")
    mp.spawn(_worker, args=(world, _free_port(), n_chunks, fail_at, out), nprocs=world, join=True)
    return [torch.load(f"{out}.{r}") for r in range(world)]


def test_wavefront_five_chunks_two_ranks(tmp_path):
    r0, r1 = _run(tmp_path, 5)
    assert r0["err"] is None and r1["err"] is None
    assert [c for c, _ in r0["log"]] == [0, 2, 4] and [c for c, _ in r1["log"]] == [1, 3]
    # chained values: v0 = 1, v1 = 2 + 1 = 3, v2 = 3 + 3 = 6, v3 = 4 + 6 = 10, v4 = 5 + 10 = 15
    assert [v for _, v in r0["log"]] == [None, 3.0, 10.0] and [v for _, v in r1["log"]] == [1.0, 6.0]
    res = r0["res"]
    assert r1["res"] is None and len(res) == 5
    assert [float(t.float().mean()) for t in res] == [1.0, 3.0, 6.0, 10.0, 15.0]


def test_producer_failure_propagates_instead_of_hanging(tmp_path):
    r0, r1 = _run(tmp_path, 4, fail_at=1)
    assert "boom" in r1["err"]                    # chunk 1 (rank 1) failed ...
    # ... and chunk 2's consumer (rank 0) raised instead of spinning: normally via the FAILED status header, or via the
    # transport error if the producer's sockets were already gone
    assert r0["err"] is not None and ("status -1" in r0["err"] or "rror" in r0["err"])


def test_stitch_chunks_drops_five_overlap_frames():
    v = [torch.arange(81).view(1, 81, 1).float() + 100 * i for i in range(3)]
    s = stitch_chunks(v)
    assert s.shape[1] == 81 + 76 + 76 and s[0, 81, 0] == 105 and s[0, 80, 0] == 80


def _hs_worker(rank, world, port, out_path):
    """5 chunks over 2 ranks with unequal chunk lengths: rank 0's chunks take long AFTER their sink (so its consumer announces
    ready while rank 0 still "computes" -> issued by a poll), rank 1's sinks come while rank 0 is still busy (deferred)."""
    import datetime
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    ho = ChunkHandoff(SHAPE, "cpu")
    ev = []
    issue = ho._issue

    def spy(chunk, hdr, payload):
        # the invariant the handshake exists for: a send is only ever issued against an announced consumer
        arrived = ho._ready_req[chunk + 1][2]
        ev.append(("issue", chunk, bool(arrived.is_set()), time.time()))
        issue(chunk, hdr, payload)
    ho._issue = spy

    def make_chunk(c, initial, sink):
        base = torch.full(SHAPE, float(c + 1), dtype=torch.bfloat16)
        if initial is not None:
            base = base + initial.float().mean().to(torch.bfloat16)
        time.sleep(0.2)                      # "anchor stage"
        sink(base)
        ev.append(("sink", c, len(ho._deferred), time.time()))
        for _ in range(6 if rank == 0 else 1):     # "in-fill stages", a poll at every stage boundary
            time.sleep(0.25)
            ho.poll()
        return base[:, :2].clone()

    res = run_chunk_wavefront(make_chunk, 5, ho, to_initial=lambda t: t[:, :2])
    torch.save({"ev": ev, "res": res, "stats": ho.stats}, f"{out_path}.{rank}")
    dist.destroy_process_group()


def test_ready_handshake_defers_sends_until_the_consumer_announced(tmp_path):
    out = str(tmp_path / "hs")
    mp.spawn(_hs_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = (torch.load(f"{out}.{r}") for r in range(2))
    assert [float(t.float().mean()) for t in r0["res"]] == [1.0, 3.0, 6.0, 10.0, 15.0]      # same chain as without the handshake
    for r in (r0, r1):
        issues = [e for e in r["ev"] if e[0] == "issue"]
        assert issues and all(e[2] for e in issues), issues            # never issued before the consumer's announcement had arrived
    # rank 1's chunks end their anchor stage while rank 0 is still inside its long chunk: deferred at the sink
    assert any(e[0] == "sink" and e[2] == 1 for e in r1["ev"]), r1["ev"]
    # rank 0 -> rank 1: the consumer is idle and announces early -- issued at the sink or by one of the polls of the long in-fill
    # part, in any case before rank 0's next chunk (not only by the blocking drain at the end of the wavefront)
    kinds = [(e[0], e[1]) for e in r0["ev"]]
    assert kinds.index(("issue", 0)) < kinds.index(("sink", 2)) and kinds.index(("issue", 2)) < kinds.index(("sink", 4))
    # ... and NOT by the drain: rank 1 is idle and has announced long before rank 0's sinks, so rank 0's hand-offs must leave AT the
    # sink (or the first poll, 0.25 s later), not 1.5 s later when its chunk ends.  (A gloo Work's is_completed() never turns true
    # without a wait: polling it deferred every hand-off to the end of the producer's chunk and serialised the wavefront -- which
    # bench.py --wavefront-chunks showed as stagger == chunk time.)
    t_of = {(e[0], e[1]): e[3] for e in r0["ev"]}
    for c in (0, 2):
        assert t_of[("issue", c)] - t_of[("sink", c)] < 0.6, (c, t_of[("issue", c)] - t_of[("sink", c)])
    assert all(e[2] == 0 for e in r0["ev"] if e[0] == "sink"), r0["ev"]                  # nothing of rank 0's was ever deferred
    # stamps of every hand-off on the consumer side: sink (producer clock, same host) <= recv done, ready announced before recv done
    for r, chunks in ((r0, (2, 4)), (r1, (1, 3))):
        for c in chunks:
            st = r["stats"][c]
            assert st["t_ready"] <= st["t_recv_done"] and st["t_sink"] <= st["t_recv_done"] + 1e-3, (c, st)


def _late_fail_worker(rank, world, port, out_path):
    """Rank 1's chunk 1 reaches its sink while rank 0 is still inside chunk 0 (hand-off DEFERRED by the ready handshake), then fails
    in a later stage: the deferred anchors must not be lost -- rank 0's recv for chunk 2 has to raise, not hang."""
    import datetime
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    ho = ChunkHandoff(SHAPE, "cpu")
    seen = {}

    def make_chunk(c, initial, sink):
        base = torch.full(SHAPE, float(c + 1), dtype=torch.bfloat16)
        time.sleep(0.1)
        sink(base)
        seen[c] = len(ho._deferred)
        if c == 0:
            time.sleep(1.5)                    # rank 0 stays busy long after its sink: chunk 1's hand-off finds it not ready
        if c == 1:
            time.sleep(0.2)
            raise ValueError("late boom")      # after the sink, with the hand-off still deferred
        return base[:, :2].clone()

    t0 = time.time()
    try:
        run_chunk_wavefront(make_chunk, 4, ho, to_initial=lambda t: t[:, :2])
        err = None
    except Exception as e:
        err = repr(e)
        time.sleep(2)
    torch.save({"err": err, "seen": seen, "secs": time.time() - t0}, f"{out_path}.{rank}")
    dist.destroy_process_group()


def test_deferred_handoff_of_a_producer_that_fails_later_is_not_lost(tmp_path):
    out = str(tmp_path / "lf")
    mp.spawn(_late_fail_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = (torch.load(f"{out}.{r}") for r in range(2))
    assert "late boom" in r1["err"] and r1["seen"].get(1) == 1, r1          # it WAS deferred when the producer failed
    assert r0["err"] is not None and "status -1" in r0["err"], r0           # the consumer got the FAILED header ...
    assert r0["secs"] < 30                                                   # ... promptly, not at the 60 s group timeout


def _six_chunk_worker(rank, world, port, out_path, stage_s):
    """n_chunks = 3 x lanes with equal chunks of 4 'stages' (sink after the 2nd, a poll at every stage boundary, as the pipeline does)."""
    import datetime
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    ho = ChunkHandoff(SHAPE, "cpu")
    marks = {}

    def make_chunk(c, initial, sink):
        base = torch.full(SHAPE, float(c + 1), dtype=torch.bfloat16)
        marks[c] = {"t_start": time.time()}
        for st in range(4):
            time.sleep(stage_s)
            if st == 1:
                sink(base)
            ho.poll()
        marks[c]["t_end"] = time.time()
        return base[:, :2].clone()

    run_chunk_wavefront(make_chunk, 3 * world, ho, to_initial=lambda t: t[:, :2], gather=False)
    torch.save({"stats": ho.stats, "marks": marks}, f"{out_path}.{rank}")
    dist.destroy_process_group()


def test_every_handoff_leaves_within_one_stage_of_sink_or_consumer_ready(tmp_path):
    """the regression the handshake bug was (every hand-off left at the END of the producer's chunk and the wavefront
    ran chunk after chunk), stated on ChunkHandoff.stats over 3 x lanes chunks: a hand-off is issued within one stage time of the
    later of {its sink, its consumer's announcement} -- at the sink when the consumer is idle, at the next stage-boundary poll when
    the consumer was still inside its previous chunk -- and the wavefront's wall clock shows the overlap."""
    out, stage_s, world = str(tmp_path / "six"), 0.3, 2
    mp.spawn(_six_chunk_worker, args=(world, _free_port(), out, stage_s), nprocs=world, join=True)
    rs = [torch.load(f"{out}.{r}") for r in range(world)]
    stats, marks = {}, {}
    for r in rs:
        for c, st in r["stats"].items():
            stats.setdefault(c, {}).update(st)          # producer side: t_sink, t_issued; consumer side: t_ready, t_recv_done
        marks.update(r["marks"])
    assert sorted(stats) == [1, 2, 3, 4, 5]
    for c, st in sorted(stats.items()):
        late = st["t_issued"] - max(st["t_sink"], st["t_ready"])
        assert late < stage_s + 0.15, (c, late, st)
        assert st["t_recv_done"] - st["t_issued"] < 0.5, (c, st)
    # and the chunks overlap: 6 chunks of 4 stages on 2 lanes take about 3 chunk times + the first stagger, not 6
    wall = max(m["t_end"] for m in marks.values()) - min(m["t_start"] for m in marks.values())
    assert wall < 4.2 * 4 * stage_s, wall
