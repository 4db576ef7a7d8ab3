"""HIP Wan-I2V image cross-attention (mmpl_i2v_*) vs the reference modules' golden outputs (-m gpu).
Stated tolerance: rel-L2 <= 2e-2 (bf16), like the other per-module checks."""
import pytest
import torch

from tests.test_oracle_golden import _i2v_inputs
from tests.util import GOLDEN, rel_l2

pytestmark = pytest.mark.gpu


def test_mlp_proj_and_i2v_cross_attention_vs_reference_golden():
    from mmpl_amd.i2v_clip import MLPProj, WanI2VCrossAttention
    from mmpl_amd.synthetic import i2v_cross_state_dict
    fx = torch.load(f"{GOLDEN}/i2v_cross_tiny.pt")
    m = fx["meta"]
    ca_sd, mp_sd = i2v_cross_state_dict(m["dim"], seed=m["weight_seed"])
    clip_fea, txt, x = _i2v_inputs(m)
    proj = MLPProj(1280, m["dim"])
    proj.load_state_dict(mp_sd)
    img = proj(clip_fea)
    e_img = rel_l2(img, fx["ctx_img"])
    ca = WanI2VCrossAttention(m["dim"], m["heads"])
    ca.load_state_dict(ca_sd)
    out = ca(x.cuda(), torch.cat([fx["ctx_img"], txt], dim=1).cuda(), None)
    torch.cuda.synchronize()
    e_out = rel_l2(out, fx["out"])
    print(f"i2v image cross-attention: rel_l2 MLPProj = {e_img:.3e}, WanI2VCrossAttention = {e_out:.3e}")
    assert img.shape == (1, 257, m["dim"]) and out.shape == (1, m["Lq"], m["dim"])
    assert e_img < 2e-2 and e_out < 2e-2
    # K/V prepared once per prompt are reused across forwards (the cached-context form the DiT uses for text)
    kv = ca.prepare(torch.cat([fx["ctx_img"], txt], dim=1)[0].cuda())
    assert torch.equal(ca(x.cuda(), None, None, kv=kv), out)
    with pytest.raises(RuntimeError, match="weights not loaded"):
        WanI2VCrossAttention(m["dim"], m["heads"])(x.cuda(), None, None, kv=kv)


def test_i2v_cross_attention_production_dims_vs_oracle():
    """dim 1536 (1.3B: 12 heads), 3 x 1560 query tokens; checker = the oracle on CPU."""
    from mmpl_amd.i2v_clip import WanI2VCrossAttention
    from mmpl_amd.synthetic import i2v_cross_state_dict, philox_normal
    from oracle import i2v_ref
    dim, heads, Lq = 1536, 12, 4680
    ca_sd, _ = i2v_cross_state_dict(dim, seed=9)
    ctx = philox_normal([1, 257 + 512, dim], 41)
    ctx[:, 257 + 77:] = 0
    x = philox_normal([1, Lq, dim], 42)
    want = i2v_ref.i2v_cross_attention(ca_sd, x, ctx, heads)
    ca = WanI2VCrossAttention(dim, heads)
    ca.load_state_dict(ca_sd)
    got = ca(x.cuda(), ctx.cuda(), None)
    torch.cuda.synchronize()
    e = rel_l2(got, want)
    print(f"i2v cross-attention dim 1536, Lq {Lq}: rel_l2(HIP, oracle) = {e:.3e}")
    assert e < 2e-2
