"""Host side of the fused CFG+UniPC step: known-answer schedule and a CPU emulation of the HIP kernel's arithmetic
(same scalars, same rounding points) against the oracle scheduler -- bit-exact, no GPU needed."""
import torch

from mmpl_amd.scheduler import FlowMatchScheduler, FlowUniPCMultistepScheduler
from mmpl_amd.synthetic import philox_normal
from oracle.unipc_ref import FlowMatchRef, FlowUniPCRef
from tests.util import GOLDEN

BF = torch.bfloat16
rbf = lambda t: t.to(BF).float()  # noqa: E731


def emulate_kernel(st, fc, fu, x, m0, m1, last):
    """mmpl_amd/csrc/elementwise.hip: unipc_kernel, in torch."""
    fc, fu, x, m0, m1, last = [t.float() for t in (fc, fu, x, m0, m1, last)]
    flow = rbf(fu + rbf(st.guidance * rbf(fc - fu)))
    m_conv = rbf(x - rbf(st.sigma_cur * flow))
    if st.use_corrector:
        xt_ = rbf(rbf(st.c_c1 * last) - rbf(st.c_c2 * m0))
        acc = rbf(st.c_rho_last * rbf(m_conv - m0))
        if st.corr_order == 2:
            acc = rbf(rbf(st.c_rho0 * rbf(rbf(m1 - m0) * st.c_inv_rk)) + acc)
        x = rbf(xt_ - rbf(st.c_c3 * acc))
    m1, m0, last = m0, m_conv, x
    xt = rbf(rbf(st.p_c1 * x) - rbf(st.p_c2 * m0))
    if st.pred_order == 2:
        xt = rbf(xt - rbf(st.p_c3 * rbf(0.5 * rbf(rbf(m1 - m0) * st.p_inv_rk))))
    return xt.to(BF), m0.to(BF), m1.to(BF), last.to(BF)


def test_known_answer_schedule():
    s = FlowUniPCMultistepScheduler(1000, 2, 1.0)
    s.set_timesteps(50, shift=5.0)
    assert s.timesteps[:6].tolist() == [999, 995, 991, 987, 982, 978]
    assert s.timesteps[-4:].tolist() == [302, 241, 172, 92]
    assert abs(s.sigmas[0].item() - 0.99980) < 1e-5 and abs(s.sigmas[-2].item() - 0.09251) < 1e-5 and s.sigmas[-1].item() == 0
    fx = torch.load(f"{GOLDEN}/sched.pt")
    assert torch.equal(s.timesteps, fx["timesteps"]) and torch.equal(s.sigmas, fx["sigmas"])


def test_flow_match_known_answers():
    fm = FlowMatchScheduler(shift=5.0, sigma_min=0.0, extra_one_step=True)
    fm.set_timesteps(1000, training=True)
    fx = torch.load(f"{GOLDEN}/sched.pt")
    assert torch.equal(fm.timesteps[[0, 980, 999]], fx["fm_timesteps_sample"])
    assert fm.timesteps[0].item() == 1000.0 and abs(fm.timesteps[980].item() - 92.59) < 0.01
    a, n = philox_normal([2, 4, 3, 5], 1, BF), philox_normal([2, 4, 3, 5], 2, BF)
    assert torch.equal(fm.add_noise(a, n, torch.tensor([1980.0, 1000.0])), n)          # t >= 1000 -> pure noise
    assert torch.equal(fm.add_noise(a, n, torch.tensor([500.0, 92.59])), fx["fm_add_noise_mid"])
    assert torch.equal(FlowMatchRef(5.0).add_noise(a, n, torch.tensor([500.0, 92.59])), fx["fm_add_noise_mid"])


def test_kernel_arithmetic_bit_exact_vs_oracle_all_50_steps():
    shape = [1, 3, 16, 6, 10]
    o = FlowUniPCRef(1000, 2, 1.0, gpu_scalar_semantics=True)
    o.set_timesteps(50, shift=5.0)
    s = FlowUniPCMultistepScheduler(1000, 2, 1.0)
    s.set_timesteps(50, shift=5.0)
    x = philox_normal(shape, 1, BF)
    xg = x.clone()
    m0 = m1 = last = torch.zeros_like(x)
    orders = []
    for i in range(50):
        fc, fu = philox_normal(shape, 100 + i, BF), philox_normal(shape, 200 + i, BF)
        x = o.step(fu + 5.0 * (fc - fu), x)
        st = s.step_scalars(5.0)
        orders.append((st.use_corrector, st.corr_order, st.pred_order))
        xg, m0, m1, last = emulate_kernel(st, fc, fu, xg, m0, m1, last)
        assert torch.equal(xg, x), i
    assert orders[0] == (0, 1, 1) and orders[1] == (1, 1, 2) and orders[2] == (1, 2, 2) and orders[-1] == (1, 2, 1)


def test_oracle_matches_reference_golden_trajectory():
    fx = torch.load(f"{GOLDEN}/sched.pt")
    for dt, name in ((BF, "bf16"), (torch.float32, "f32")):
        x, target = philox_normal([1, 3, 4, 6, 8], 5, dt), philox_normal([1, 3, 4, 6, 8], 6, dt)
        o = FlowUniPCRef(1000, 2, 1.0)
        o.set_timesteps(50, shift=5.0)
        for i in range(50):
            v = (x - target) * (1.0 + 0.1 * torch.sin(x.float() * 3 + i).to(dt))
            x = o.step(v, x)
            ref = fx[f"traj_{name}"][i]
            assert (x.float() - ref.float()).abs().max().item() <= (0.02 if dt == BF else 1e-4), (name, i)


def test_gpu_scalar_semantics_vs_the_reference_itself():
    """`sched.pt: traj_bf16_gpu_semantics` = the REAL FlowUniPCMultistepScheduler.step with its `sigmas` carrying
    make_golden._GpuScalar (operands of scalar-first products swapped at dispatch: what PyTorch's GPU kernels compute; unedited
    reference code).  (a) the oracle's `gpu_scalar_semantics=True` is that, bit for bit, over all 50 steps; (b) so is the HIP
    kernel's arithmetic (emulate_kernel above, with fc == fu so that the CFG combine is the identity) -- HIP step arithmetic against
    the reference with no oracle in between; (c) it is NOT the CPU-semantics trajectory."""
    fx = torch.load(f"{GOLDEN}/sched.pt")
    ref = fx["traj_bf16_gpu_semantics"]
    assert ref.shape[0] == 50 and (ref.float() - fx["traj_bf16"].float()).abs().max().item() > 1e-2
    x, target = philox_normal([1, 3, 4, 6, 8], 5, BF), philox_normal([1, 3, 4, 6, 8], 6, BF)
    o = FlowUniPCRef(1000, 2, 1.0, gpu_scalar_semantics=True)
    o.set_timesteps(50, shift=5.0)
    s = FlowUniPCMultistepScheduler(1000, 2, 1.0)
    s.set_timesteps(50, shift=5.0)
    xo, xk = x.clone(), x.clone()
    m0 = m1 = last = torch.zeros_like(x)
    for i in range(50):
        xo = o.step((xo - target) * (1.0 + 0.1 * torch.sin(xo.float() * 3 + i).to(BF)), xo)
        assert torch.equal(xo, ref[i]), ("oracle", i)
        v = (xk - target) * (1.0 + 0.1 * torch.sin(xk.float() * 3 + i).to(BF))
        xk, m0, m1, last = emulate_kernel(s.step_scalars(5.0), v, v, xk, m0, m1, last)
        assert torch.equal(xk, ref[i]), ("kernel arithmetic", i)
