#!/usr/bin/env python3
"""Build attn_w64.hip with -save-temps in a scratch dir and audit the ISA (CDNA guide 5.7 item 4): the kernel owns a[0:255] by
name, so the build is only valid if the compiler never touches the accumulator file itself, spills nothing and uses no scratch.
    python tools/audit_w64.py [--src gemm.hip] [--keep DIR]"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "mmpl_amd", "csrc", "attn_w64.hip")


def audit(workdir, src=SRC, own_agprs=True, own_kernels=None):
    """own_agprs: the source names accumulator registers literally (attn_w64.hip) -> the compiler must not touch the file.
    own_kernels: a substring; only the kernels whose (mangled) name contains it own the accumulator file (gemm.hip: the
    one-wave-per-SIMD gemm_bf16_v8_kernel next to compiler-allocated ones); implies the per-kernel form of the check.
    For every source: m0 (written by the hand-issued LDS-DMA statements without a clobber) must not appear in compiler code."""
    stem = os.path.splitext(os.path.basename(src))[0]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-c", src, "-o", stem + ".o", "-save-temps"]
    cmd += os.environ.get("MMPL_EXTRA_HIPCC_FLAGS", "").split()
    r = subprocess.run(cmd, cwd=workdir, capture_output=True, text=True)
    if r.returncode:
        return [f"hipcc failed: {r.stderr[-2000:]}"], {}
    asm = open(os.path.join(workdir, stem + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    problems, info = [], {}
    # per kernel: compiler-generated accumulator-file accesses; and, for kernels whose hand-written statements write M0,
    # compiler-generated M0 accesses (a kernel that only uses the compiler's own LDS-DMA builtin may of course set M0)
    inasm, outside, m0 = False, [], []
    kernel, asm_m0, comp_m0 = None, {}, {}
    for i, ln in enumerate(asm.split("\n")):
        t = ln.strip()
        if re.match(r"^[A-Za-z_][\w$.]*:", t) and not t.startswith(".L"):
            kernel = t.split(":")[0]
        if ";;#ASMSTART" in ln:
            inasm = True
        elif ";;#ASMEND" in ln:
            inasm = False
        elif t.startswith((".", ";")) or not t:
            continue
        elif inasm:
            if re.search(r"\bm0\b", t.split(";")[0]):
                asm_m0[kernel] = asm_m0.get(kernel, 0) + 1
        else:
            owns = own_agprs if own_kernels is None else (kernel is not None and own_kernels in kernel)
            if owns and (t.startswith("v_accvgpr") or re.search(r"\ba\[?\d+", t.split(";")[0])):
                outside.append((i + 1, t))
            if re.search(r"\bm0\b", t.split(";")[0]):
                comp_m0.setdefault(kernel, []).append((i + 1, t))
    for kname, hits in comp_m0.items():
        if asm_m0.get(kname):
            m0 += hits
    if outside:
        problems.append(f"{len(outside)} compiler-generated accumulator-register accesses, first: {outside[:3]}")
    if m0:
        problems.append(f"compiler touches m0 in a kernel with hand-written M0 writes: {m0[:3]}")
    for key in ("vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size"):
        vals = [int(v) for v in re.findall(rf"\.{key}:\s+(\d+)", asm)]
        info[key] = vals
        if own_agprs and own_kernels is None and key != "sgpr_spill_count" and any(vals):
            problems.append(f"{key} = {vals}")
    if own_kernels is not None:      # per-kernel metadata of the owning kernels: no spills, no scratch
        for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", asm):
            if own_kernels in m.group(1) and (int(m.group(2)) or int(m.group(3))):
                problems.append(f"{m.group(1)}: scratch {m.group(2)} bytes, {m.group(3)} spilled VGPRs")
        info["own_kernels"] = sorted(set(k for k in re.findall(r"\.name:\s+(\S+)", asm) if own_kernels in k))
    info["vgpr_count"] = [int(v) for v in re.findall(r"\.vgpr_count:\s+(\d+)", asm)]
    info["agpr_count"] = [int(v) for v in re.findall(r"\.agpr_count:\s+(\d+)", asm)]
    info["sgpr_count"] = [int(v) for v in re.findall(r"\.sgpr_count:\s+(\d+)", asm)]
    return problems, info


if __name__ == "__main__":
    src = SRC
    if "--src" in sys.argv:
        src = os.path.join(ROOT, "mmpl_amd", "csrc", sys.argv[sys.argv.index("--src") + 1])
    if "--keep" in sys.argv:
        d = sys.argv[sys.argv.index("--keep") + 1]
        os.makedirs(d, exist_ok=True)
        problems, info = audit(d, src)
    else:
        with tempfile.TemporaryDirectory() as d:
            problems, info = audit(d, src)
    print(info)
    for p in problems:
        print("PROBLEM:", p)
    sys.exit(1 if problems else 0)
