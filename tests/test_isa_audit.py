"""Build-time proofs the hand-written asm relies on (CDNA guide 5.7 items 1 and 4), checked on the ISA hipcc actually emits.

* attn_w64.hip names the accumulator registers a[0:255] literally: valid only if the compiler never touches the accumulator file
  itself (a spill or a copy into it would silently corrupt O / Q / V fragments), spills nothing and uses no scratch.
* the LDS-DMA statements of attn_w64.hip, attention.hip and gemm.hip write M0 without declaring it (an "m0" clobber only draws a
  warning and a save / restore pair per DMA op costs issue slots in the hot loops): valid only if no compiler-generated
  instruction of those translation units reads or writes M0.
* attn_w64_sched.inc is what tools/gen_attn_w64.py generates.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
CSRC = os.path.join(ROOT, "mmpl_amd", "csrc")

hipcc = pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not installed")


@hipcc
def test_attn_w64_owns_the_accumulator_file(tmp_path):
    import audit_w64
    problems, info = audit_w64.audit(str(tmp_path))
    assert not problems, problems
    assert info["agpr_count"] == [256, 256] and all(v <= 512 for v in info["vgpr_count"])


@hipcc
@pytest.mark.parametrize("src", ["attention.hip", "gemm.hip"])
def test_compiler_never_touches_m0(tmp_path, src):
    import audit_w64
    problems, _ = audit_w64.audit(str(tmp_path), os.path.join(CSRC, src), own_agprs=False)
    assert not problems, problems


@hipcc
def test_gemm_v8_owns_the_accumulator_file(tmp_path):
    """gemm_bf16_v8_kernel (one wave per SIMD) keeps its 128 x 128 accumulators in a[0:255] by name, like attn_w64.hip: no
    compiler-generated access to the accumulator file, no spill, no scratch in any of its instantiations.  (hipcc DID park hoisted
    epilogue addresses in a[1..] until they were derived from an opaque copy of the lane id behind the k loop.)"""
    import audit_w64
    problems, info = audit_w64.audit(str(tmp_path), os.path.join(CSRC, "gemm.hip"), own_agprs=True, own_kernels="gemm_bf16_v8_kernel")
    assert not problems, problems
    assert len(info["own_kernels"]) == 7, info["own_kernels"]


@hipcc
def test_gemm_register_budget(tmp_path):
    """gemm_bf16_v6_kernel sits at 253-256 of 256 VGPRs; the persistent tile loop, the L2 prefetch shares and the epilogue's gate
    rows cost it 10-14 spilled registers, all of them OUTSIDE the k loop (tile bookkeeping saved before / restored after it).
    Two guards: the count (an attempt at overlapping the next tile's first fetch with the epilogue spilled 19-153 and ran at
    780 TFLOP/s), and -- the one that matters -- no scratch access between the loop's first fragment read and its last MFMA."""
    import re
    import audit_w64
    _, info = audit_w64.audit(str(tmp_path), os.path.join(CSRC, "gemm.hip"), own_agprs=False)
    assert max(info["vgpr_spill_count"]) <= 16, info["vgpr_spill_count"]
    asm = open(os.path.join(str(tmp_path), "gemm-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    kernels = re.findall(r"^(_ZN[^\n:]*gemm_bf16_v6_kernel[^\n:]*):[^\n]*\n(.*?)s_endpgm", asm, flags=re.S | re.M)
    assert len(kernels) >= 7
    for name, body in kernels:
        lines = body.split("\n")
        mf = [i for i, l in enumerate(lines) if "v_mfma" in l]
        # the k loop's body: from the loop header that dominates the MFMA block (the last label before the 24 fragment reads of
        # R_t, i.e. the nearest back-branch target before the first MFMA) to the last MFMA
        reads = [i for i, l in enumerate(lines[:mf[0]]) if "ds_read_b128" in l]
        start = reads[-24] if len(reads) >= 24 else reads[0]
        inside = [l.strip() for l in lines[start:mf[-1] + 1] if "scratch_" in l]
        assert not inside, (name, inside)


def test_schedule_include_is_current(tmp_path):
    import gen_attn_w64
    committed = open(gen_attn_w64.OUT).read()
    gen_attn_w64.OUT = str(tmp_path / "sched.inc")
    gen_attn_w64.emit()
    assert open(gen_attn_w64.OUT).read() == committed, "run python tools/gen_attn_w64.py"
