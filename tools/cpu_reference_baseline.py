"""The reference's OWN CPU path timed in the build container: BASELINE.json configs[0] (Wan2.1-T2V-1.3B shapes, 480p).

Build container only -- imports /root/reference through tests/golden/_ref_import.py, never travels.
What is timed: `CausalFPSWanModel._forward_inference` (MMPL_t2v/wan/modules/causal_fps_model.py:708-837) in bf16 on seeded
synthetic 1.3B-shaped weights, latent 60 x 104, through the reference's own SDPA fallback (wan/modules/attention.py:170-185):
  * ONE block (num_layers = 1, everything else as wan_t2v_1_3B.py:17-25) at the four T2V stage shapes s0 .. s3, against a live
    KV cache filled by the earlier stages, warm (best of `--reps`);
  * all 30 layers at s0 (`--full-s0`), the survey's "one forward per stage shape" for the cheapest shape.
Writes profiles/cpu_reference_1p3B_480p.json; BASELINE.md section 4 cites it.

    python tools/cpu_reference_baseline.py [--reps 3] [--full-s0] [--threads N]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from _ref_import import load_reference  # noqa: E402
from mmpl_amd.synthetic import WAN_CONFIGS, dit_state_dict, philox_normal  # noqa: E402
from oracle import stage_ref  # noqa: E402

S, H, W = 1560, 60, 104


def fwd_flops(cfg, lq, lkv):
    d, f, L = cfg["dim"], cfg["ffn_dim"], cfg["num_layers"]
    return L * (2 * lq * (6 * d * d + 2 * d * f) + 4 * lq * lkv * d + 4 * lq * 512 * d)


def build(fps, cfg, seed=1):
    m = fps.CausalFPSWanModel(model_type="t2v", dim=cfg["dim"], ffn_dim=cfg["ffn_dim"], num_heads=cfg["num_heads"],
                              num_layers=cfg["num_layers"], text_dim=cfg["text_dim"], freq_dim=cfg["freq_dim"]).eval()
    m.load_state_dict(dit_state_dict(cfg, seed=seed), strict=True)
    return m.to(torch.bfloat16)


def caches(cfg):
    kv = [{"k": torch.zeros(1, 15 * S, cfg["num_heads"], 128, dtype=torch.bfloat16), "v": torch.zeros(1, 15 * S, cfg["num_heads"], 128, dtype=torch.bfloat16),
           "global_end_index": torch.tensor([0]), "local_end_index": torch.tensor([0]), "attention_vis_index": []} for _ in range(cfg["num_layers"])]
    return kv, [{"k": None, "v": None, "is_init": False} for _ in range(cfg["num_layers"])]


def forward(m, x, tval, ctx, kv, cross, frames):
    t = torch.full([1, len(frames)], float(tval), dtype=torch.float32)
    return m(x.permute(0, 2, 1, 3, 4), t=t, context=ctx, seq_len=32760, kv_cache=kv, crossattn_cache=cross,
             current_start=[f * S for f in frames], cache_start=[f * S for f in frames])


def time_stages(fps, cfg, reps, only_s0=False):
    m = build(fps, cfg)
    kv, cross = caches(cfg)
    ctx = philox_normal([1, 512, cfg["text_dim"]], 11)
    ctx[:, 64:] = 0                                                         # pad-zeroing, utils/wan_wrapper.py:46-47
    noise = philox_normal([1, 21, 16, H, W], 7)
    out = []
    for si, frames in enumerate(stage_ref.stage_frames(stage_ref.T2V_CLEAN_STEPS)):
        for blk in kv:                                                      # casual_fps_inference.py:298-302, 321-325
            for v in (20 * S, 19 * S):
                if si == 2 and v in blk["attention_vis_index"]:
                    blk["attention_vis_index"].remove(v)
                if si == 3 and v not in blk["attention_vis_index"]:
                    blk["attention_vis_index"].append(v)
        best, cold = None, None
        for _ in range(reps + 1):                                           # the first call is the warm-up (and fills the cache)
            t0 = time.perf_counter()
            forward(m, noise[:, frames], 500.0, ctx, kv, cross, frames)
            dt = time.perf_counter() - t0
            if _ == 0:
                cold = dt
            if _ > 0 or reps == 0:
                best = dt if best is None else min(best, dt)
        lq = len(frames) * S
        lkv = {0: 2, 1: 9, 2: 13, 3: 21}[si] * S
        fl = fwd_flops(cfg, lq, lkv)
        out.append(dict(stage=f"s{si}", frames=len(frames), Lq=lq, Lkv=lkv, seconds=round(best, 3), first_call_seconds=round(cold, 3), tflop=round(fl / 1e12, 4),
                        tflops_per_s=round(fl / best / 1e12, 4)))
        print(f"[cpu-ref] {cfg['num_layers']:2d} layer(s) s{si}: {best:8.2f} s  {fl / 1e12:7.3f} TFLOP  {fl / best / 1e12:.3f} TFLOP/s", flush=True)
        if only_s0:
            break
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--full-s0", action="store_true")
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    if a.threads:
        torch.set_num_threads(a.threads)
    torch.set_grad_enabled(False)
    fps, *_ = load_reference()
    full = dict(WAN_CONFIGS["1.3B"])
    one = dict(full, num_layers=1)
    res = dict(what="the REFERENCE's CausalFPSWanModel._forward_inference (causal_fps_model.py:708-837), bf16, CPU, SDPA fallback (attention.py:170-185)",
               config="Wan2.1-T2V-1.3B shapes (dim 1536, ffn 8960, 12 heads), seeded synthetic weights, latent 60x104 (S = 1560), BASELINE.json configs[0]",
               host=dict(cores=os.cpu_count(), torch_threads=torch.get_num_threads(), torch=torch.__version__),
               one_block=time_stages(fps, one, a.reps))
    blk = {r["stage"]: r["seconds"] for r in res["one_block"]}
    # a T2V chunk = 4 stages x 102 forwards of 30 blocks (the embeddings / head are < 1 % of a 30-layer forward)
    res["extrapolated"] = dict(s_per_forward={k: round(30 * v, 1) for k, v in blk.items()}, chunk_hours=round(102 * 30 * sum(blk.values()) / 3600, 2),
                               latent_frames_per_s=round(21 / (102 * 30 * sum(blk.values())), 7), note="30 x the one-block time per stage shape x 102 forwards per stage")
    if a.full_s0:
        res["full_depth_s0"] = time_stages(fps, full, max(a.reps - 1, 1), only_s0=True)[0]
    path = os.path.join(ROOT, "profiles", "cpu_reference_1p3B_480p.json")
    json.dump(res, open(path, "w"), indent=1)
    print(json.dumps(res["extrapolated"]), "->", path)


if __name__ == "__main__":
    main()
