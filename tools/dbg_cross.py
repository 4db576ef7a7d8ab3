"""dev: where does attn_cross_kernel differ from the lock-step kernel?  python tools/dbg_cross.py Lq H S"""
import ctypes as C, math, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmpl_amd import _lib
lib = _lib.load()
Lq, H, S = (int(v) for v in sys.argv[1:4])
dev, BF, d = "cuda:0", torch.bfloat16, H * 128
torch.manual_seed(S)
q = torch.randn(Lq, d, device=dev).to(BF)
k = torch.randn(S, d, device=dev).to(BF)
v = torch.randn(S, d, device=dev).to(BF)
outs = []
for cross in (1, 0, 1):
    o = torch.zeros(Lq, d, device=dev, dtype=BF)
    kp, vp = (C.c_void_p * 1)(k.data_ptr()), (C.c_void_p * 1)(v.data_ptr())
    _lib.check(lib.mmpl_attn_fwd_variant(_lib.ptr(q), d, _lib.ptr(o), d, kp, vp, d, d, 1, S, Lq, H, 1.0 / math.sqrt(128), None, 0, 0, cross, _lib.stream_ptr()))
    torch.cuda.synchronize()
    outs.append(o)
ref = torch.softmax(torch.einsum("qhd,khd->hqk", q.float().view(Lq, H, 128), k.float().view(S, H, 128)) / math.sqrt(128), -1)
ref = torch.einsum("hqk,khd->qhd", ref, v.float().view(S, H, 128)).reshape(Lq, d)
for name, o in (("cross", outs[0]), ("lock", outs[1]), ("cross again", outs[2])):
    print(name, "rel_l2 vs fp32", ((o.float() - ref).norm() / ref.norm()).item())
bad = (outs[0] != outs[1])
print("mismatching elements", int(bad.sum()), "of", bad.numel(), " cross vs cross again", int((outs[0] != outs[2]).sum()))
if bad.any():
    rows = bad.any(1).nonzero().flatten()
    cols = bad.any(0).nonzero().flatten()
    print("rows", rows[:20].tolist(), "... n", len(rows), " q blocks", sorted(set((rows // 256).tolist()))[:40])
    print("heads", sorted(set((cols // 128).tolist())), " cols in head", sorted(set((cols % 128).tolist()))[:40])
    r, c = bad.nonzero()[0].tolist()
    print("first", r, c, outs[0][r, c].item(), outs[1][r, c].item(), ref[r, c].item())
