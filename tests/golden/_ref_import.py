"""Import the REAL reference (Tele-AI/MMPL, /root/reference) on CPU.  Build-container only.

Used only by tests/golden/make_golden.py to generate fixtures; nothing that runs on the GPU box
imports this (the reference does not travel).  Recipe = SURVEY.md Appendix B:
  1. stub `diffusers` (not installed),
  2. bypass wan/__init__.py (needs easydict / torchvision),
  3. route WanT2VCrossAttention's flash_attention to the reference's own SDPA fallback (attention.py:170-185).
"""
import importlib
import importlib.util
import sys
import types

import torch

REF = "/root/reference/MMPL_t2v"


def _mk(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def load_reference():
    if "wan.modules.causal_fps_model" in sys.modules:
        m = sys.modules
        return (m["wan.modules.causal_fps_model"], m["wan.modules.model"], m["wan.modules.attention"],
                m["wan.modules.vae"], m["wan.utils.fm_solvers_unipc"], m["_ref_flow_sched"])
    for n in ["diffusers", "diffusers.configuration_utils", "diffusers.models", "diffusers.models.modeling_utils",
              "diffusers.schedulers", "diffusers.schedulers.scheduling_utils", "diffusers.utils"]:
        _mk(n)

    class _Cfg(dict):
        __getattr__ = dict.__getitem__

    class ConfigMixin:
        def register_to_config(self, **kw):
            if not hasattr(self, "config"):
                self.config = _Cfg()
            self.config.update(kw)

    def register_to_config(init):
        import functools
        import inspect

        @functools.wraps(init)
        def wrapper(self, *a, **kw):
            sig = inspect.signature(init)
            ba = sig.bind(self, *a, **kw)
            ba.apply_defaults()
            cfg = {k: v for k, v in ba.arguments.items() if k != "self"}
            self.config = _Cfg(cfg)
            init(self, *a, **kw)
        return wrapper

    cu = sys.modules["diffusers.configuration_utils"]
    cu.ConfigMixin, cu.register_to_config = ConfigMixin, register_to_config
    sys.modules["diffusers.models.modeling_utils"].ModelMixin = torch.nn.Module
    su = sys.modules["diffusers.schedulers.scheduling_utils"]

    class SchedulerMixin:
        pass

    class SchedulerOutput:
        def __init__(self, prev_sample):
            self.prev_sample = prev_sample

    class _K:
        name = "x"

    su.SchedulerMixin, su.SchedulerOutput, su.KarrasDiffusionSchedulers = SchedulerMixin, SchedulerOutput, [_K]
    du = sys.modules["diffusers.utils"]
    du.deprecate = lambda *a, **k: None
    du.is_scipy_available = lambda: False
    for pkg, path in [("wan", REF + "/wan"), ("wan.modules", REF + "/wan/modules"), ("wan.utils", REF + "/wan/utils")]:
        p = types.ModuleType(pkg)
        p.__path__ = [path]
        sys.modules[pkg] = p
    attn = importlib.import_module("wan.modules.attention")
    model = importlib.import_module("wan.modules.model")
    fps = importlib.import_module("wan.modules.causal_fps_model")
    vae = importlib.import_module("wan.modules.vae")
    unipc = importlib.import_module("wan.utils.fm_solvers_unipc")
    spec = importlib.util.spec_from_file_location("_ref_flow_sched", REF + "/utils/scheduler.py")
    sched = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sched)
    sys.modules["_ref_flow_sched"] = sched
    model.flash_attention = attn.attention
    return fps, model, attn, vae, unipc, sched
