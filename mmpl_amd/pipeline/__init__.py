from .casual_fps_inference import CausalFPSInferencePipeline  # noqa: F401
