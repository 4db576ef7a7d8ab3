"""Time the HIP umT5-xxl encoder (24 layers, dim 4096, text_len 512) on one prompt; weights random on the device."""
import sys
import time

import torch

sys.path.insert(0, ".")
from mmpl_amd.synthetic import T5_CONFIGS  # noqa: E402
from mmpl_amd.t5 import T5Engine  # noqa: E402


def main():
    cfg = T5_CONFIGS["umt5-xxl"]
    dev = "cuda:0"
    d, da, df, n, nb = cfg["dim"], cfg["dim_attn"], cfg["dim_ffn"], cfg["num_heads"], cfg["num_buckets"]
    g = torch.Generator(device=dev).manual_seed(0)

    def rn(*s, std):
        return (torch.randn(*s, generator=g, device=dev) * std).bfloat16()
    sd = {"token_embedding.weight": rn(cfg["vocab"], d, std=1.0), "norm.weight": torch.ones(d, device=dev).bfloat16()}
    for i in range(cfg["num_layers"]):
        p = f"blocks.{i}."
        sd.update({p + "norm1.weight": torch.ones(d, device=dev).bfloat16(), p + "norm2.weight": torch.ones(d, device=dev).bfloat16(),
                   p + "attn.q.weight": rn(da, d, std=(d * da // n) ** -0.5), p + "attn.k.weight": rn(da, d, std=d ** -0.5),
                   p + "attn.v.weight": rn(da, d, std=d ** -0.5), p + "attn.o.weight": rn(d, da, std=(n * da // n) ** -0.5),
                   p + "pos_embedding.embedding.weight": rn(nb, n, std=(2 * nb * n) ** -0.5),
                   p + "ffn.gate.0.weight": rn(df, d, std=d ** -0.5), p + "ffn.fc1.weight": rn(df, d, std=d ** -0.5),
                   p + "ffn.fc2.weight": rn(d, df, std=df ** -0.5)})
    eng = T5Engine(cfg, text_len=512, device=dev)
    eng.load_state_dict(sd)
    del sd
    ids = torch.randint(2, cfg["vocab"], (1, 512))
    mask = torch.zeros(1, 512, dtype=torch.long)
    mask[:, :80] = 1
    out = eng.encode(ids, mask)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.encode(ids, mask)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    L = 512
    flops = cfg["num_layers"] * (2 * L * d * 3 * da + 2 * L * da * d + 3 * 2 * L * d * df + 4 * L * L * da)
    wbytes = cfg["num_layers"] * 2 * (4 * d * da + 3 * d * df)
    print(f"umT5-xxl encode (1 prompt, text_len 512): {dt * 1e3:.2f} ms  {flops / dt / 1e12:.1f} TFLOP/s  "
          f"weights {wbytes / 1e9:.2f} GB -> {wbytes / dt / 1e9:.0f} GB/s (HBM floor {wbytes / 8e12 * 1e3:.2f} ms)")


if __name__ == "__main__":
    main()
