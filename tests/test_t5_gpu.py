"""HIP umT5 encoder vs the reference's golden output and the oracle (-m gpu).  Tolerance: rel-L2 <= 2e-2 (bf16)."""
import pytest
import torch

from tests.util import GOLDEN, max_abs, rel_l2

pytestmark = pytest.mark.gpu


def test_t5_encoder_vs_reference_golden():
    from mmpl_amd.synthetic import T5_CONFIGS, t5_state_dict
    from mmpl_amd.t5 import T5Engine, relative_position_buckets
    from oracle import t5_ref
    fx = torch.load(f"{GOLDEN}/t5_tiny.pt")
    cfg = T5_CONFIGS[fx["meta"]["cfg"]]
    L = fx["meta"]["L"]
    sd = t5_state_dict(cfg, seed=fx["meta"]["weight_seed"])
    # host bucket table == the reference's bucket function
    rel = torch.arange(L).unsqueeze(0) - torch.arange(L).unsqueeze(1)
    tab = relative_position_buckets(L, cfg["num_buckets"])
    assert torch.equal(tab[(rel + L - 1)].long(), t5_ref.relative_position_bucket(rel, cfg["num_buckets"]))
    eng = T5Engine(cfg, text_len=L, device="cuda:0")
    eng.load_state_dict(sd)
    out = eng.encode(fx["ids"], fx["mask"])
    torch.cuda.synchronize()
    e = rel_l2(out, fx["out"])
    print(f"t5: rel_l2(HIP, reference golden) = {e:.3e} max|d| = {max_abs(out, fx['out']):.3e}")
    assert e < 2e-2
    assert out[0, 37:].abs().sum().item() == 0          # padding rows zeroed (wan_wrapper.py:46-47)
    e32 = rel_l2(out, fx["out_f32"])
    ref_gap = rel_l2(fx["out"], fx["out_f32"])
    print(f"t5: rel_l2(HIP, reference fp32 mode) = {e32:.3e}  (reference bf16 vs its own fp32: {ref_gap:.3e})")
    assert e32 < 1.5 * ref_gap + 5e-3                     # bf16 engine sits at the reference's own bf16-vs-fp32 distance


def test_t5_text_len_512_vs_oracle_and_padding_invariance():
    """Production text_len (512) on a 3-layer config against the CPU oracle; ids under the mask must not matter."""
    from mmpl_amd.synthetic import T5_CONFIGS, t5_state_dict
    from mmpl_amd.t5 import T5Engine
    from oracle import t5_ref
    cfg = T5_CONFIGS["small"]
    sd = t5_state_dict(cfg, seed=11)
    L = 512
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(2, cfg["vocab"], (3, L), generator=g)
    mask = torch.zeros(3, L, dtype=torch.long)
    for b, n in enumerate((1, 77, 512)):                  # one token / typical prompt / truncated-to-max prompt
        mask[b, :n] = 1
        ids[b, n:] = 0
    want = t5_ref.text_encoder_forward(sd, ids, mask, cfg["num_heads"], cfg["num_buckets"], cfg["num_layers"])
    eng = T5Engine(cfg, text_len=L, device="cuda:0")
    eng.load_state_dict(sd)
    out = eng.encode(ids, mask)
    for b in range(3):
        n = int(mask[b].sum())
        e = rel_l2(out[b, :n], want[b, :n])
        print(f"t5 small L=512 n_valid={n}: rel_l2(HIP, oracle) = {e:.3e}")
        assert e < 2e-2
        assert out[b, n:].abs().sum().item() == 0
    ids2 = ids.clone()
    ids2[1, 77:] = torch.randint(2, cfg["vocab"], (L - 77,), generator=g)
    assert torch.equal(eng.encode(ids2, mask)[1], out[1])


def test_wan_text_encoder_seam():
    """WanTextEncoder(text_prompts) -> {'prompt_embeds'} through tokenizer -> HIP engine (wan_wrapper.py:15-51)."""
    from mmpl_amd.synthetic import T5_CONFIGS, t5_state_dict
    from mmpl_amd.wan_wrapper import WanTextEncoder
    cfg = T5_CONFIGS["tiny"]

    class Tok:                                             # HF-tokenizer-shaped stand-in (no vocabulary files offline)
        def __call__(self, seqs, return_tensors, padding, truncation, max_length, add_special_tokens):
            ids = torch.zeros(len(seqs), max_length, dtype=torch.long)
            mask = torch.zeros_like(ids)
            for i, s in enumerate(seqs):
                t = [2 + (ord(c) % 900) for c in s][:max_length - 1] + [1]
                ids[i, :len(t)] = torch.tensor(t)
                mask[i, :len(t)] = 1
            import types
            return types.SimpleNamespace(input_ids=ids, attention_mask=mask)

    te = WanTextEncoder(state_dict=t5_state_dict(cfg, seed=4), tokenizer=Tok(), cfg=cfg, text_len=128)
    a = te(text_prompts=["a  cat \n on a   mat", "dog"])["prompt_embeds"]
    b = te(text_prompts=["a cat on a mat"])["prompt_embeds"]
    assert a.shape == (2, 128, cfg["dim"]) and a.dtype == torch.bfloat16
    assert torch.equal(a[0], b[0])                         # whitespace clean (tokenizers.py:18-21); per-prompt independence
    assert a[1, 4:].abs().sum().item() == 0 and a[1, :4].abs().sum().item() > 0
    with pytest.raises(RuntimeError):
        WanTextEncoder(pretrained_path="/nonexistent.pth")(text_prompts=["x"])
