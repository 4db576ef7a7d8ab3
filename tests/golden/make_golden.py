"""Generate golden fixtures by running the REAL reference (/root/reference, CPU, bf16) on seeded inputs.

Build-container only (the reference never travels to the GPU box).  Usage:
    python tests/golden/make_golden.py [dit] [chunk] [chunk50] [chunk50_gpu] [chunk50_deep:{cpu,gpu,perm,f32}] [chunk_i2v] [chunk_i2v50] [sched] [vae]
Writes tests/golden/*.pt.  Inputs are regenerated from seeds by the tests (mmpl_amd.synthetic), so the
fixtures hold expected OUTPUTS of the reference (full or strided + sha256), plus known-answer scalars.
While generating, the oracle restatement (oracle/) is run on the same inputs and its agreement with the
reference is printed -- tests/test_oracle_golden.py re-checks that against the committed files.
"""
import hashlib
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from _ref_import import load_reference  # noqa: E402
from mmpl_amd.synthetic import WAN_CONFIGS, dit_state_dict, philox_normal  # noqa: E402
from oracle import stage_ref, unipc_ref  # noqa: E402
from oracle import wan_dit_ref as W  # noqa: E402

torch.set_grad_enabled(False)
S480 = 1560
H, Wd = 60, 104


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()


def rel_l2(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def build_ref_model(fps, cfg_name="tiny", seed=1):
    cfg = WAN_CONFIGS[cfg_name]
    m = fps.CausalFPSWanModel(model_type="t2v", dim=cfg["dim"], ffn_dim=cfg["ffn_dim"], num_heads=cfg["num_heads"],
                              num_layers=cfg["num_layers"], text_dim=cfg["text_dim"], freq_dim=cfg["freq_dim"]).eval()
    sd = dit_state_dict(cfg, seed=seed)
    missing = m.load_state_dict(sd, strict=True)
    m = m.to(torch.bfloat16)
    return m, sd, cfg


def ref_caches(cfg, n_slots=15):
    kv = [{"k": torch.zeros(1, n_slots * S480, cfg["num_heads"], 128, dtype=torch.bfloat16),
           "v": torch.zeros(1, n_slots * S480, cfg["num_heads"], 128, dtype=torch.bfloat16),
           "global_end_index": torch.tensor([0]), "local_end_index": torch.tensor([0]),
           "attention_vis_index": []} for _ in range(cfg["num_layers"])]
    cross = [{"k": None, "v": None, "is_init": False} for _ in range(cfg["num_layers"])]
    return kv, cross


def ref_forward(m, x, tval, ctx, kv, cross, frames):
    """x: [1, nF, 16, h, w] (pipeline layout) -> flow [1, nF, 16, h, w]; mirrors WanFPSWrapper.forward."""
    t = torch.full([1, len(frames)], float(tval), dtype=torch.float32)
    y = m(x.permute(0, 2, 1, 3, 4), t=t, context=ctx, seq_len=32760, kv_cache=kv, crossattn_cache=cross,
          current_start=[f * S480 for f in frames], cache_start=[f * S480 for f in frames])
    return y.permute(0, 2, 1, 3, 4)


def make_context(cfg, seed, n_valid=48):
    c = philox_normal([1, 512, cfg["text_dim"]], seed)
    c[:, n_valid:] = 0                                # pad-zeroing, wan_wrapper.py:46-47
    return c


def gen_dit():
    fps, *_ = load_reference()
    m, sd, cfg = build_ref_model(fps, "tiny", seed=1)
    ocfg = W.DitCfg(**cfg)
    ctx = make_context(cfg, 11)
    noise = philox_normal([1, 21, 16, H, Wd], 7)
    kv, cross = ref_caches(cfg)
    okv = W.new_kv_cache(ocfg, 15, S480)
    ocross = [None] * cfg["num_layers"]
    vis = stage_ref.VisIndex()
    out = {}
    stages = stage_ref.stage_frames(stage_ref.T2V_CLEAN_STEPS)
    for si, frames in enumerate(stages):
        if si == 2:
            for blk in kv:
                for v in (31200, 29640):
                    if v in blk["attention_vis_index"]:
                        blk["attention_vis_index"].remove(v)
            vis.hide()
        if si == 3:
            for blk in kv:
                for v in (31200, 29640):
                    if v not in blk["attention_vis_index"]:
                        blk["attention_vis_index"].append(v)
            vis.show()
        x = noise[:, frames]
        tval = [999.0, 700.0, 301.0, 0.0][si]
        t0 = time.time()
        y = ref_forward(m, x, tval, ctx, kv, cross, frames)
        dt = time.time() - t0
        vis.on_forward(frames)
        # the reference gathers K/V in `list(set(...))` order (causal_fps_model.py:219); record it so the
        # oracle can be pinned bit-exactly (attention is permutation-invariant only up to fp32 summation order)
        ref_order = [stage_ref.slot_of(v // S480) for v in kv[0]["attention_vis_index"]]
        assert sorted(ref_order) == sorted(vis.slots()), (ref_order, vis.slots())
        out[f"s{si}_vis_order"] = ref_order
        t = torch.full([1, len(frames)], tval, dtype=torch.float32)
        yo = W.dit_forward(sd, ocfg, x[0].permute(1, 0, 2, 3), t, ctx[0], okv, ocross, frames,
                           stage_ref.write_slots_for(frames), ref_order).permute(1, 0, 2, 3).unsqueeze(0)
        print(f"[dit] stage {si} frames {frames} ref {dt:.2f}s  oracle-vs-ref rel_l2={rel_l2(yo, y):.3e} "
              f"max|d|={(yo.float() - y.float()).abs().max().item():.3e} rms={y.float().pow(2).mean().sqrt().item():.3f}")
        out[f"s{si}_sha"] = sha(y)
        out[f"s{si}_strided"] = y[..., ::2, ::2].clone()
        if si == 0:
            out["s0_full"] = y.clone()
    # cache content pin: K/V of layer 1, slot 13 (frame 19) strided
    out["kv_l1_slot13_k"] = kv[1]["k"][0, 13 * S480:14 * S480:13].clone()
    out["kv_l1_slot13_v"] = kv[1]["v"][0, 13 * S480:14 * S480:13].clone()
    ok = torch.equal(okv[1]["k"][0, 13 * S480:14 * S480:13], out["kv_l1_slot13_k"])
    print("[dit] oracle cache slot13 K equal to reference:", ok)
    out["meta"] = dict(cfg="tiny", weight_seed=1, ctx_seed=11, noise_seed=7, n_valid=48, tvals=[999.0, 700.0, 301.0, 0.0])
    torch.save(out, os.path.join(HERE, "dit_forward_tiny.pt"))


def gen_chunk(steps=2):
    """Re-enact casual_fps_inference.py:250-403 with the reference model + reference UniPC (steps reduced)."""
    fps, _, _, _, unipc, sched = load_reference()
    m, sd, cfg = build_ref_model(fps, "tiny", seed=2)
    ocfg = W.DitCfg(**cfg)
    ctx_c, ctx_u = make_context(cfg, 21, 40), make_context(cfg, 22, 12)
    noise = philox_normal([1, 21, 16, H, Wd], 23)
    renoise = {f: philox_normal([1, 16, H, Wd], 100 + f) for f in (4, 9, 13, 18)}
    t0 = time.time()
    output, handoff = _ref_stage_loop(m, fps, unipc, sched, cfg, noise.clone(), renoise, [ctx_c, ctx_u], steps)
    print(f"[chunk] reference stage loop ({steps} steps/stage): {time.time() - t0:.1f}s")
    t0 = time.time()
    o_out, o_hand, _ = stage_ref.run_chunk(sd, ocfg, noise, ctx_c[0], ctx_u[0], renoise, None, "t2v", 5.0, steps, 5.0)
    print(f"[chunk] oracle stage loop: {time.time() - t0:.1f}s  rel_l2 vs ref: out={rel_l2(o_out, output):.3e} "
          f"handoff={rel_l2(o_hand, handoff):.3e}  out rms={output.float().pow(2).mean().sqrt().item():.3f}")
    torch.save(dict(out_sha=sha(output), out_strided=output[..., ::2, ::2].clone(), handoff_sha=sha(handoff),
                    handoff_strided=handoff[..., ::3, ::3].clone(),
                    meta=dict(cfg="tiny", weight_seed=2, ctx_seeds=(21, 22), n_valid=(40, 12), noise_seed=23,
                              renoise_seed_base=100, steps=steps, guidance=5.0, shift=5.0)),
               os.path.join(HERE, "chunk_t2v_tiny.pt"))


class _GpuScalar(torch.Tensor):
    """The reference's UniPC step under the scalar semantics of its NATIVE platform, without editing the reference.

    `FlowUniPCMultistepScheduler` keeps `sigmas` on the CPU (fm_solvers_unipc.py:129,226) and writes its scalar-tensor products
    scalar-first (`sigma_t * model_output` :321, `sigma_t / sigma_s0 * x - alpha_t * h_phi_1 * m0` :460,600, `alpha_t * B_h * ...`
    :466,607).  PyTorch's GPU kernels take a 0-dim fp32 CPU tensor as an fp32 scalar on either side of a multiply; its CPU kernels
    round it to the tensor's dtype (bf16) when it is the FIRST operand and keep it in fp32 when it is the second.  Assigning
    `scheduler.sigmas = scheduler.sigmas.as_subclass(_GpuScalar)` makes every scalar derived from `sigmas` carry this class, and
    `__torch_function__` then does ONE thing: when a 0-dim fp32 tensor is the first operand of a multiply whose second operand is a
    non-fp32 tensor with dims, the operands are swapped.  Everything else runs the reference's own code on plain tensors.  fp32
    results of at most one dim are re-wrapped so that the property follows `sigma -> alpha, lambda, h, h_phi_1, B_h ...`."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        plain = lambda a: a.as_subclass(torch.Tensor) if isinstance(a, _GpuScalar) else a
        args = tuple(plain(a) for a in args)
        kwargs = {k: plain(v) for k, v in kwargs.items()}
        if (getattr(func, "__name__", "") in ("mul", "__mul__") and len(args) == 2 and isinstance(args[0], torch.Tensor)
                and args[0].dim() == 0 and args[0].dtype == torch.float32 and isinstance(args[1], torch.Tensor)
                and args[1].dim() > 0 and args[1].dtype != torch.float32):
            func, args = torch.mul, (args[1], args[0])
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **kwargs)
        if isinstance(out, torch.Tensor) and out.dtype == torch.float32 and out.dim() <= 1:
            out = out.as_subclass(_GpuScalar)
        return out


def _ref_stage_loop(m, fps, unipc, sched, cfg, noise, renoise, ctxs, steps, dtype=torch.bfloat16, progress=None,
                    mode="t2v", initial_latent=None, gpu_scalars=False):
    """The stage loop re-enacted on the reference model + the reference UniPC, `steps` per stage.
    t2v: MMPL_t2v/pipeline/casual_fps_inference.py:250-403 (first chunk: re-noise of frames 4/9 and 13/18, frames 19/20 hidden
    during stage 2 and re-added for stage 3, hand-off `cat([output[:, :1], latents])` after stage 1).
    i2v: MMPL_i2v/pipeline/casual_fps_inference.py:253-435: schedule [0],[1],anchors,[4..9],[13..18]; `initial_latent` of one
    frame (:403-435: refresh of stage 0, stage 1 is denoised) or two frames (:369-401: both refreshed, denoising starts at the
    anchors); no re-noise, nothing hidden; hand-off `cat([output[:, :1], output[:, -2:]])` after the anchor stage (:340-343).
    gpu_scalars: the scheduler's `sigmas` carry `_GpuScalar` (the reference's own step code under GPU scalar semantics)."""
    kvs = [ref_caches(cfg), ref_caches(cfg)]
    if dtype != torch.bfloat16:
        for kv, _ in kvs:
            for blk in kv:
                blk["k"], blk["v"] = blk["k"].to(dtype), blk["v"].to(dtype)
    noise = noise.to(dtype)
    stages = stage_ref.stage_frames(stage_ref.T2V_CLEAN_STEPS if mode == "t2v" else stage_ref.I2V_CLEAN_STEPS)
    output = torch.zeros_like(noise)
    handoff = None
    fm = sched.FlowMatchScheduler(shift=5.0, sigma_min=0.0, extra_one_step=True)
    fm.set_timesteps(1000, training=True)
    ddmp_t = torch.tensor([[1980.0]])                  # timesteps[idx]+1000 >= 1000 for any idx (pipeline :96-107)

    def refresh(latents, frames):
        for w in (0, 1):
            ref_forward(m, latents, 0.0, ctxs[w], kvs[w][0], kvs[w][1], frames)

    first = 0
    if mode == "i2v":
        assert initial_latent is not None and initial_latent.shape[1] in (1, 2)
        initial_latent = initial_latent.to(dtype)
        for j in range(initial_latent.shape[1]):       # i2v :369-435
            refresh(initial_latent[:, j:j + 1], stages[j])
            output[:, stages[j]] = initial_latent[:, j:j + 1]
        first = initial_latent.shape[1]
    for si in range(first, len(stages)):
        frames = stages[si]
        latents = noise[:, frames]
        if mode == "t2v" and si in (2, 3):
            src = (3, 10) if si == 2 else (12, 19)
            latents[:, 0:1] = fm.add_noise(output[:, src[0]:src[0] + 1].flatten(0, 1), renoise[frames[0]].to(dtype),
                                           ddmp_t.flatten(0, 1)).unflatten(0, (1, 1))
            latents[:, -1:] = fm.add_noise(output[:, src[1]:src[1] + 1].flatten(0, 1), renoise[frames[-1]].to(dtype),
                                           ddmp_t.flatten(0, 1)).unflatten(0, (1, 1))
            for kv, _ in kvs:
                for blk in kv:
                    for v in (31200, 29640):
                        if si == 2 and v in blk["attention_vis_index"]:
                            blk["attention_vis_index"].remove(v)
                        if si == 3 and v not in blk["attention_vis_index"]:
                            blk["attention_vis_index"].append(v)
        s = unipc.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
        s.set_timesteps(steps, device="cpu", shift=5.0)
        if gpu_scalars:
            s.sigmas = s.sigmas.as_subclass(_GpuScalar)
        for t in s.timesteps:
            fc = ref_forward(m, latents, t, ctxs[0], kvs[0][0], kvs[0][1], frames)
            fu = ref_forward(m, latents, t, ctxs[1], kvs[1][0], kvs[1][1], frames)
            flow = fu + 5.0 * (fc - fu)
            latents = s.step(flow, t, latents, return_dict=False)[0]
            assert type(latents) is torch.Tensor and latents.dtype == dtype
        output[:, frames] = latents
        if mode == "t2v" and si == 1:
            handoff = torch.cat([output[:, :1], latents], dim=1)
        if mode == "i2v" and si == 2:
            handoff = torch.cat([output[:, :1], output[:, -2:]], dim=1)
        refresh(latents, frames)
        if progress:
            progress(si)
    return output, handoff


def gen_chunk50(steps=50):
    """The REAL sampling length: 4 stages x (50 UniPC steps x 2 CFG forwards) + refresh = 408 reference forwards at
    60x104 (casual_fps_inference.py:338-403, fm_solvers_unipc.py:655-739), plus the reference's OWN noise floor over
    that trajectory: (i) the same run with only the K/V gather order of its self-attention changed (the reference's
    order is `list(set(...))`, i.e. unspecified, causal_fps_model.py:219) and (ii) the same weights run in fp32."""
    fps, model, attn, _, unipc, sched = load_reference()
    m, sd, cfg = build_ref_model(fps, "tiny", seed=2)
    ocfg = W.DitCfg(**cfg)
    ctx_c, ctx_u = make_context(cfg, 21, 40), make_context(cfg, 22, 12)
    noise = philox_normal([1, 21, 16, H, Wd], 23)
    renoise = {f: philox_normal([1, 16, H, Wd], 100 + f) for f in (4, 9, 13, 18)}
    tick = lambda tag: (lambda si: print(f"[chunk50] {tag}: stage {si} done at {time.time() - t0:.0f}s", flush=True))
    t0 = time.time()
    output, handoff = _ref_stage_loop(m, fps, unipc, sched, cfg, noise.clone(), renoise, [ctx_c, ctx_u], steps, progress=tick("ref bf16"))
    print(f"[chunk50] reference stage loop ({steps} steps/stage): {time.time() - t0:.1f}s", flush=True)

    # (i) the reference against itself with the gathered K/V rows in reverse FRAME order (softmax is permutation
    # invariant; only the fp32 summation order inside its SDPA changes)
    real_attention = fps.attention

    def permuted_attention(q, k, v, *a, **kw):
        n = k.shape[1] // S480
        idx = torch.arange(n * S480).view(n, S480).flip(0).reshape(-1)
        return real_attention(q, k[:, idx], v[:, idx], *a, **kw)

    fps.attention = permuted_attention
    t0 = time.time()
    out_perm, hand_perm = _ref_stage_loop(m, fps, unipc, sched, cfg, noise.clone(), renoise, [ctx_c, ctx_u], steps, progress=tick("ref permuted"))
    fps.attention = real_attention
    print(f"[chunk50] reference, K/V frame order reversed: {time.time() - t0:.1f}s  rel_l2 vs ref: out={rel_l2(out_perm, output):.3e} "
          f"handoff={rel_l2(hand_perm, handoff):.3e}", flush=True)

    # (ii) the same weights in fp32 (model, caches, latents, attention)
    m32 = fps.CausalFPSWanModel(model_type="t2v", dim=cfg["dim"], ffn_dim=cfg["ffn_dim"], num_heads=cfg["num_heads"],
                                num_layers=cfg["num_layers"], text_dim=cfg["text_dim"], freq_dim=cfg["freq_dim"]).eval()
    m32.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    f32_attention = lambda q, k, v, *a, **kw: real_attention(q, k, v, *a, **dict(kw, dtype=torch.float32))
    fps.attention, model.flash_attention = f32_attention, f32_attention
    t0 = time.time()
    out32, hand32 = _ref_stage_loop(m32, fps, unipc, sched, cfg, noise.clone(), renoise, [ctx_c.float(), ctx_u.float()], steps,
                                    dtype=torch.float32, progress=tick("ref fp32"))
    fps.attention, model.flash_attention = real_attention, attn.attention
    print(f"[chunk50] reference in fp32: {time.time() - t0:.1f}s  rel_l2 bf16-ref vs fp32-ref: out={rel_l2(output, out32):.3e} "
          f"handoff={rel_l2(handoff, hand32):.3e}", flush=True)

    t0 = time.time()
    o_out, o_hand, _ = stage_ref.run_chunk(sd, ocfg, noise, ctx_c[0], ctx_u[0], renoise, None, "t2v", 5.0, steps, 5.0)
    e_or = rel_l2(o_out, output)
    print(f"[chunk50] oracle stage loop: {time.time() - t0:.1f}s  rel_l2 vs ref: out={e_or:.3e} "
          f"handoff={rel_l2(o_hand, handoff):.3e}  out rms={output.float().pow(2).mean().sqrt().item():.3f}", flush=True)
    torch.save(dict(out_sha=sha(output), out_strided=output[..., ::2, ::2].clone(), handoff_sha=sha(handoff),
                    handoff_strided=handoff[..., ::3, ::3].clone(),
                    out_f32_strided=out32[..., ::2, ::2].to(torch.bfloat16).clone(),
                    noise_floor=dict(order_out=rel_l2(out_perm, output), order_handoff=rel_l2(hand_perm, handoff),
                                     f32_out=rel_l2(output, out32), f32_handoff=rel_l2(handoff, hand32),
                                     perm_vs_f32_out=rel_l2(out_perm, out32), oracle_out=e_or,
                                     oracle_handoff=rel_l2(o_hand, handoff), oracle_vs_f32_out=rel_l2(o_out, out32)),
                    meta=dict(cfg="tiny", weight_seed=2, ctx_seeds=(21, 22), n_valid=(40, 12), noise_seed=23,
                              renoise_seed_base=100, steps=steps, guidance=5.0, shift=5.0)),
               os.path.join(HERE, "chunk_t2v_tiny_50.pt"))


def _chunk_inputs(m):
    cfg = WAN_CONFIGS[m["cfg"]]
    ctxs = [make_context(cfg, s_, nv) for s_, nv in zip(m["ctx_seeds"], m["n_valid"])]
    noise = philox_normal([1, 21, 16, H, Wd], m["noise_seed"])
    renoise = {f: philox_normal([1, 16, H, Wd], m["renoise_seed_base"] + f) for f in (4, 9, 13, 18)}
    return cfg, ctxs, noise, renoise


def gen_chunk50_gpu_semantics(steps=50):
    """The same 408-forward chunk under the scalar semantics of the reference's NATIVE platform, produced by the REAL reference.
    The committed chunk50 fixture is the reference run on a CPU, where PyTorch rounds the UniPC step's 0-dim fp32 scalars to bf16
    when they are the first operand of a multiply (`sigma_t * x`, fm_solvers_unipc.py:315-331 and the predictor / corrector
    updates :460-476,600-618); its GPU kernels keep them in fp32.  Here the reference's own `FlowUniPCMultistepScheduler.step`
    runs with `sigmas` carrying `_GpuScalar` (above): unedited reference code, only the operand order of those products swapped at
    dispatch.  First the shim is checked on the toy trajectory of `gen_sched` against `FlowUniPCRef(gpu_scalar_semantics=True)`
    (bit-exact).  Stored with the reference's output: its distance to the CPU-semantics fixture (what the platform alone does to the
    trajectory), to the reference's fp32 run, and -- as a cross-check only -- the oracle's `gpu_scalar_semantics=True` run."""
    fps, _, _, _, unipc, sched = load_reference()
    _check_gpu_scalar_shim(unipc)
    fx = torch.load(os.path.join(HERE, "chunk_t2v_tiny_50.pt"))
    m = fx["meta"]
    assert m["steps"] == steps
    cfg, ctxs, noise, renoise = _chunk_inputs(m)
    mdl, sd, _ = build_ref_model(fps, m["cfg"], seed=m["weight_seed"])
    t0 = time.time()
    tick = lambda si: print(f"[chunk50_gpu] reference, GPU scalar semantics: stage {si} done at {time.time() - t0:.0f}s", flush=True)
    out, hand = _ref_stage_loop(mdl, fps, unipc, sched, cfg, noise.clone(), renoise, ctxs, steps, progress=tick, gpu_scalars=True)
    d = dict(vs_cpu_semantics_out=rel_l2(out[..., ::2, ::2], fx["out_strided"]), vs_cpu_semantics_handoff=rel_l2(hand[..., ::3, ::3], fx["handoff_strided"]),
             vs_f32_out=rel_l2(out[..., ::2, ::2], fx["out_f32_strided"]))
    print(f"[chunk50_gpu] REFERENCE under GPU scalar semantics: {time.time() - t0:.1f}s  vs its CPU-semantics run: out={d['vs_cpu_semantics_out']:.3e} "
          f"handoff={d['vs_cpu_semantics_handoff']:.3e}; vs its fp32 run: {d['vs_f32_out']:.3e}", flush=True)
    t0 = time.time()
    o_out, o_hand, _ = stage_ref.run_chunk(sd, W.DitCfg(**cfg), noise, ctxs[0][0], ctxs[1][0], renoise, None, "t2v", m["guidance"], steps, m["shift"],
                                           gpu_scalar_semantics=True)
    d.update(oracle_out=rel_l2(o_out, out), oracle_handoff=rel_l2(o_hand, hand))
    print(f"[chunk50_gpu] oracle(gpu_scalar_semantics=True): {time.time() - t0:.1f}s  vs the reference under the shim: out={d['oracle_out']:.3e} "
          f"handoff={d['oracle_handoff']:.3e} (the reference's own K/V-order noise: {fx['noise_floor']['order_out']:.3e})", flush=True)
    torch.save(dict(out_sha=sha(out), out_strided=out[..., ::2, ::2].clone(), handoff_sha=sha(hand), handoff_strided=hand[..., ::3, ::3].clone(),
                    distances=d, oracle_out_sha=sha(o_out),
                    produced_by="the REAL reference (CausalFPSWanModel + FlowUniPCMultistepScheduler.step, unedited) with scheduler.sigmas carrying "
                                "make_golden._GpuScalar; see make_golden.py gen_chunk50_gpu_semantics",
                    meta=dict(m)), os.path.join(HERE, "chunk_t2v_tiny_50_gpu_semantics.pt"))


def gen_chunk50_deep(which, steps=50, cfg_name="deep"):
    """The 408-forward T2V chunk of `gen_chunk50` / `gen_chunk50_gpu_semantics` on a DEEPER synthetic model (WAN_CONFIGS["deep"]:
    8 layers, dim 512, 4 heads; the tiny model has 2 / 256 / 2): how a per-forward difference compounds over 50 UniPC steps x CFG 5
    depends on the depth, and the per-forward HIP-vs-reference distance at 30 / 40 layers is ~5 x the tiny model's (DESIGN.md
    section 4).  Every run is the REAL reference (its model, its unedited UniPC step); one run is about an hour of 8 cores, so the
    runs are separate steps that each update chunk_t2v_deep_50.pt:
        cpu   the reference as is (CPU scalar semantics)                       -> out_strided / handoff_strided / sha
        gpu   scheduler.sigmas carrying `_GpuScalar` (the reference's native platform)  -> gpu_* entries: what the HIP pipeline is held to
        perm  the `gpu` run with only the K/V gather order of its self-attention reversed -> noise_floor order_* (the bound's unit)
        f32   the same weights in fp32 (CPU semantics; a scalar is fp32 either way)      -> noise_floor f32_*
    The oracle's own distance is NOT measured here (an hour more); tests/test_trajectory_gpu.py compares the HIP pipeline with the
    stored reference outputs directly."""
    fps, model, attn, _, unipc, sched = load_reference()
    path = os.path.join(HERE, "chunk_t2v_deep_50.pt")
    meta = dict(cfg=cfg_name, weight_seed=3, ctx_seeds=(31, 32), n_valid=(40, 12), noise_seed=33, renoise_seed_base=200, steps=steps,
                guidance=5.0, shift=5.0)
    fx = torch.load(path) if os.path.exists(path) else dict(meta=meta, noise_floor={}, seconds={},
                                                            produced_by="the REAL reference (CausalFPSWanModel + FlowUniPCMultistepScheduler.step, "
                                                                        "unedited); make_golden.py gen_chunk50_deep")
    assert fx["meta"] == meta, (fx["meta"], meta)
    cfg, ctxs, noise, renoise = _chunk_inputs(meta)
    mdl, sd, _ = build_ref_model(fps, cfg_name, seed=meta["weight_seed"])
    t0 = time.time()
    tick = lambda si: print(f"[chunk50_deep {which}] stage {si} done at {time.time() - t0:.0f}s", flush=True)
    real_attention = fps.attention
    if which == "cpu":
        out, hand = _ref_stage_loop(mdl, fps, unipc, sched, cfg, noise.clone(), renoise, ctxs, steps, progress=tick)
        fx.update(out_sha=sha(out), out_strided=out[..., ::2, ::2].clone(), handoff_sha=sha(hand), handoff_strided=hand[..., ::3, ::3].clone())
    elif which == "gpu":
        _check_gpu_scalar_shim(unipc)
        out, hand = _ref_stage_loop(mdl, fps, unipc, sched, cfg, noise.clone(), renoise, ctxs, steps, progress=tick, gpu_scalars=True)
        fx.update(gpu_out_sha=sha(out), gpu_out_strided=out[..., ::2, ::2].clone(), gpu_handoff_sha=sha(hand),
                  gpu_handoff_strided=hand[..., ::3, ::3].clone())
    elif which == "perm":
        def permuted_attention(q, k, v, *a, **kw):
            n = k.shape[1] // S480
            idx = torch.arange(n * S480).view(n, S480).flip(0).reshape(-1)
            return real_attention(q, k[:, idx], v[:, idx], *a, **kw)
        fps.attention = permuted_attention
        out, hand = _ref_stage_loop(mdl, fps, unipc, sched, cfg, noise.clone(), renoise, ctxs, steps, progress=tick, gpu_scalars=True)
        fps.attention = real_attention
        fx.update(gpu_perm_out_strided=out[..., ::2, ::2].clone(), gpu_perm_handoff_strided=hand[..., ::3, ::3].clone())
    elif which == "f32":
        m32 = fps.CausalFPSWanModel(model_type="t2v", dim=cfg["dim"], ffn_dim=cfg["ffn_dim"], num_heads=cfg["num_heads"],
                                    num_layers=cfg["num_layers"], text_dim=cfg["text_dim"], freq_dim=cfg["freq_dim"]).eval()
        m32.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
        f32_attention = lambda q, k, v, *a, **kw: real_attention(q, k, v, *a, **dict(kw, dtype=torch.float32))
        fps.attention, model.flash_attention = f32_attention, f32_attention
        out, hand = _ref_stage_loop(m32, fps, unipc, sched, cfg, noise.clone(), renoise, [c.float() for c in ctxs], steps, dtype=torch.float32,
                                    progress=tick)
        fps.attention, model.flash_attention = real_attention, attn.attention
        fx.update(out_f32_strided=out[..., ::2, ::2].to(torch.bfloat16).clone(), handoff_f32_strided=hand[..., ::3, ::3].to(torch.bfloat16).clone())
    else:
        raise SystemExit(f"chunk50_deep: unknown run {which!r}")
    fx["seconds"][which] = time.time() - t0
    nf = fx["noise_floor"]
    have = lambda *ks: all(k in fx for k in ks)
    if have("gpu_out_strided", "out_strided"):
        nf.update(gpu_vs_cpu_semantics_out=rel_l2(fx["gpu_out_strided"], fx["out_strided"]),
                  gpu_vs_cpu_semantics_handoff=rel_l2(fx["gpu_handoff_strided"], fx["handoff_strided"]))
    if have("gpu_out_strided", "gpu_perm_out_strided"):
        nf.update(order_out=rel_l2(fx["gpu_perm_out_strided"], fx["gpu_out_strided"]),
                  order_handoff=rel_l2(fx["gpu_perm_handoff_strided"], fx["gpu_handoff_strided"]))
    if have("out_f32_strided", "out_strided"):
        nf.update(f32_out=rel_l2(fx["out_strided"], fx["out_f32_strided"]), f32_handoff=rel_l2(fx["handoff_strided"], fx["handoff_f32_strided"]))
    if have("out_f32_strided", "gpu_out_strided"):
        nf.update(gpu_vs_f32_out=rel_l2(fx["gpu_out_strided"], fx["out_f32_strided"]))
    print(f"[chunk50_deep {which}] {fx['seconds'][which]:.0f}s  rms={out.float().pow(2).mean().sqrt().item():.3f}  noise floors so far: "
          + ", ".join(f"{k}={v:.3e}" for k, v in nf.items()), flush=True)
    torch.save(fx, path)


def _check_gpu_scalar_shim(unipc):
    """The shim on the toy trajectory: reference + `_GpuScalar` == FlowUniPCRef(gpu_scalar_semantics=True), bit for bit, and it differs
    from the plain CPU run (else the shim did nothing)."""
    dt = torch.bfloat16
    x, target = philox_normal([1, 3, 4, 6, 8], 5, dt), philox_normal([1, 3, 4, 6, 8], 6, dt)
    res = {}
    for shim in (False, True):
        s = unipc.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
        s.set_timesteps(50, device="cpu", shift=5.0)
        if shim:
            s.sigmas = s.sigmas.as_subclass(_GpuScalar)
        o = unipc_ref.FlowUniPCRef(1000, 2, 1.0, gpu_scalar_semantics=shim)
        o.set_timesteps(50, shift=5.0)
        xr, xo, traj = x.clone(), x.clone(), []
        for i, t in enumerate(s.timesteps):
            xr = s.step((xr - target) * (1.0 + 0.1 * torch.sin(xr.float() * 3 + i).to(dt)), t, xr, return_dict=False)[0]
            xo = o.step((xo - target) * (1.0 + 0.1 * torch.sin(xo.float() * 3 + i).to(dt)), xo)
            assert torch.equal(xr, xo), (shim, i)
            traj.append(xr.clone())
        res[shim] = torch.stack(traj)
    dmax = (res[True].float() - res[False].float()).abs().max().item()
    assert dmax > 1e-2, dmax
    print(f"[shim] reference + _GpuScalar == oracle(gpu_scalar_semantics=True) bit-exactly over 50 steps; vs the plain CPU run max|d| = {dmax:.3e}")
    return res[True]


def gen_chunk_i2v(steps=2, tag=""):
    """The I2V stage plan (BASELINE configs[4]) from the REAL reference: MMPL_i2v/pipeline/casual_fps_inference.py:253-435 re-enacted on
    the reference model (the I2V pipeline drives the same CausalFPSWanModel / UniPC sources: the two trees' files are identical up to
    blank lines) for a first chunk (`initial_latent` = the image latent, 1 frame: stage [1] and the anchors are denoised) and for a
    later chunk (2 frames, both refreshed, denoising starts at the anchors).  Under CPU scalar semantics (the reference as is) and,
    for the tight HIP bound, under `_GpuScalar`."""
    fps, _, _, _, unipc, sched = load_reference()
    _check_gpu_scalar_shim(unipc)
    m = dict(cfg="tiny", weight_seed=2, ctx_seeds=(21, 22), n_valid=(40, 12), noise_seed=24, renoise_seed_base=100, steps=steps,
             guidance=5.0, shift=5.0, initial_seed=56)
    cfg, ctxs, noise, _ = _chunk_inputs(m)
    mdl, sd, _ = build_ref_model(fps, m["cfg"], seed=m["weight_seed"])
    init2 = philox_normal([1, 2, 16, H, Wd], m["initial_seed"])
    out = dict(meta=m, produced_by="the REAL reference (CausalFPSWanModel + FlowUniPCMultistepScheduler) driven through the I2V stage plan; "
                                   "make_golden.py gen_chunk_i2v")
    for n_init in (1, 2):
        init = init2[:, :n_init]
        for gpu in (False, True):
            t0 = time.time()
            o, h = _ref_stage_loop(mdl, fps, unipc, sched, cfg, noise.clone(), None, ctxs, steps, mode="i2v", initial_latent=init, gpu_scalars=gpu)
            oo, oh, _ = stage_ref.run_chunk(sd, W.DitCfg(**cfg), noise, ctxs[0][0], ctxs[1][0], None, init, "i2v", m["guidance"], steps, m["shift"],
                                            gpu_scalar_semantics=gpu)
            key = f"init{n_init}_{'gpu' if gpu else 'cpu'}"
            print(f"[chunk_i2v{tag}] {key}: {time.time() - t0:.1f}s  oracle-vs-ref out={rel_l2(oo, o):.3e} handoff={rel_l2(oh, h):.3e} "
                  f"rms={o.float().pow(2).mean().sqrt().item():.3f}", flush=True)
            assert torch.equal(o[:, :n_init], init) and h.shape == (1, 3, 16, H, Wd)
            out[key] = dict(out_sha=sha(o), out_strided=o[..., ::4, ::2].clone(), handoff_sha=sha(h), handoff_strided=h[..., ::3, ::3].clone(),
                            oracle_out=rel_l2(oo, o), oracle_handoff=rel_l2(oh, h))      # out_strided: rows ::4, columns ::2
    torch.save(out, os.path.join(HERE, f"chunk_i2v_tiny{tag}.pt"))


def gen_sched():
    _, _, _, _, unipc, sched = load_reference()
    s = unipc.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
    s.set_timesteps(50, device="cpu", shift=5.0)
    o = unipc_ref.FlowUniPCRef(1000, 2, 1.0)
    o.set_timesteps(50, shift=5.0)
    assert torch.equal(s.timesteps, o.timesteps) and torch.equal(s.sigmas, o.sigmas)
    out = dict(timesteps=s.timesteps.clone(), sigmas=s.sigmas.clone())
    # toy velocity field trajectory in bf16 (what the pipeline feeds) and fp32
    for dt, name in ((torch.bfloat16, "bf16"), (torch.float32, "f32")):
        x = philox_normal([1, 3, 4, 6, 8], 5, dt)
        target = philox_normal([1, 3, 4, 6, 8], 6, dt)
        s = unipc.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
        s.set_timesteps(50, device="cpu", shift=5.0)
        o = unipc_ref.FlowUniPCRef(1000, 2, 1.0)
        o.set_timesteps(50, shift=5.0)
        xr, xo = x.clone(), x.clone()
        traj = []
        for i, t in enumerate(s.timesteps):
            vr = (xr - target) * (1.0 + 0.1 * torch.sin(xr.float() * 3 + i).to(dt))
            xr = s.step(vr, t, xr, return_dict=False)[0]
            vo = (xo - target) * (1.0 + 0.1 * torch.sin(xo.float() * 3 + i).to(dt))
            xo = o.step(vo, xo)
            traj.append(xr.clone())
        print(f"[sched] {name}: oracle-vs-ref max|d| = {(xo.float() - xr.float()).abs().max().item():.3e}")
        out[f"traj_{name}"] = torch.stack(traj)
    # the same bf16 toy trajectory from the REAL scheduler under the scalar semantics of its native platform (`_GpuScalar`)
    out["traj_bf16_gpu_semantics"] = _check_gpu_scalar_shim(unipc)
    fm = sched.FlowMatchScheduler(shift=5.0, sigma_min=0.0, extra_one_step=True)
    fm.set_timesteps(1000, training=True)
    of = unipc_ref.FlowMatchRef(5.0, 1000)
    assert torch.equal(fm.timesteps, of.timesteps)
    out["fm_timesteps_sample"] = fm.timesteps[[0, 980, 999]].clone()
    a, n = philox_normal([2, 4, 3, 5], 1, torch.bfloat16), philox_normal([2, 4, 3, 5], 2, torch.bfloat16)
    out["fm_add_noise_1980"] = fm.add_noise(a, n, torch.tensor([1980.0, 1000.0]))
    out["fm_add_noise_mid"] = fm.add_noise(a, n, torch.tensor([500.0, 92.59]))
    assert torch.equal(out["fm_add_noise_1980"], n)
    assert torch.equal(of.add_noise(a, n, torch.tensor([500.0, 92.59])), out["fm_add_noise_mid"])
    torch.save(out, os.path.join(HERE, "sched.pt"))


if __name__ == "__main__":
    what = sys.argv[1:] or ["dit", "chunk", "sched", "vae"]
    if "sched" in what:
        gen_sched()
    if "dit" in what:
        gen_dit()
    if "chunk" in what:
        gen_chunk()
    if "chunk50" in what:
        gen_chunk50()
    if "chunk50_gpu" in what:
        gen_chunk50_gpu_semantics()
    for w in what:                                   # chunk50_deep:cpu | :gpu | :perm | :f32  (about an hour of 8 cores each)
        if w.startswith("chunk50_deep:"):
            gen_chunk50_deep(w.split(":", 1)[1])
    if "chunk_i2v" in what:
        gen_chunk_i2v()
    if "chunk_i2v50" in what:
        gen_chunk_i2v(50, "_50")
    if "vae" in what:
        from make_golden_vae import gen_vae
        gen_vae()
