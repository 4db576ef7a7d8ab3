#!/usr/bin/env python3
"""Dev: the schedule of a TIMING MOCK of attn_w64_kernel on v_mfma_f32_16x16x32_bf16 (round 4).

Writes mmpl_amd/csrc/attn_w64_sched16.inc, which attn_w64.hip includes instead of the shipping schedule when it is compiled with
-DW64_MOCK16=1.  Same tile, same registers, same filler multiset as the shipping kernel (64 v_exp_f32, 64 row-sum adds, 32 packs,
16 K + 16 x 2 V fragment reads, 8 LDS-DMA pieces, one barrier per KV tile and wave) -- but every 32x32x16 MFMA is issued as TWO
16x16x32 MFMAs (same FLOPs, same operand registers, accumulator chains of the length a native kernel has: 4 per S tile, 2 per O
tile and KV tile) and the fillers are list-scheduled into the 128 gaps of 16 cycles (3 issue slots besides the MFMA) instead of
64 gaps of 32 cycles.  Results are garbage (the S / P / V layouts of a native 16-wide kernel are not implemented -- they are
verified separately, tools/probe_attn16.hip); what the mock answers is the question the rewrite hinges on: how many shader
cycles per KV tile does this instruction mix cost at this issue density, and what does the chip clock it at?

    W64_BUDGET16=3.0 W64_WEXP=1.5 python tools/gen_attn_w64_mock16.py      # then build with MMPL_EXTRA_HIPCC_FLAGS="-DW64_MOCK16=1 [-DW64_ABL=16]"
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "mmpl_amd", "csrc", "attn_w64_sched16.inc")

W_EXP, W_VALU = float(os.environ.get("W64_WEXP", 1.5)), 1.0
BUDGET = float(os.environ.get("W64_BUDGET16", 3.0))
NG = 64                                       # gaps per phase: one behind every 16x16x32 MFMA
K_DMA = [int(x) for x in os.environ.get("W64_KDMA16", "8,16,24,32").split(",")]           # gaps 0..63 = phase A, 64..127 = phase B
V_DMA = [int(x) for x in os.environ.get("W64_VDMA16", "40,48,56,72").split(",")]
W_DMA = float(os.environ.get("W64_WDMA16", BUDGET))

A_FIXED = {g: [] for g in range(NG)}
B_FIXED = {g: [] for g in range(NG)}


def _fixed(g):
    return A_FIXED[g] if g < NG else B_FIXED[g - NG]


A_FIXED[0].append(("PV", "k.template lds_v<15>();", 2))
A_FIXED[6].append(("QK", "k.barrier();", BUDGET))
A_FIXED[6].append(("!QK", "k.wait_lgkm0();", 1))
for i in range(4):
    _fixed(K_DMA[i]).append(("QK", f"k.template dma_k<{i}>();", W_DMA))
    _fixed(V_DMA[i]).append(("QK", f"k.template dma_v<{i}>();", W_DMA))
_fixed(max(K_DMA) + 5).append(("QK", "k.advance_k();", 1))
_fixed(max(V_DMA) + 5).append(("QK", "k.advance_v();", 1))
B_FIXED[0].append(("QK", "k.addr_k();", 1))
for f in range(16):
    B_FIXED[2 + 2 * f].append(("QK", f"k.template lds_k<{f}>();", 1 if f & 1 or f == 0 else 2))
B_FIXED[33].append(("QK", "k.addr_v();", BUDGET))
for i in range(15):
    if i == 5:
        B_FIXED[34 + 2 * i].append(("QK", "k.template wait_lgkm<10>();", 0.5))
    B_FIXED[34 + 2 * i].append(("QK", f"k.template lds_v<{i}>();", 2))
B_FIXED[61].append(("QK", "k.rotate();", 1))


def capacity():
    cap = []
    for fixed in (A_FIXED, B_FIXED):
        for g in range(NG):
            used = sum(s for fl, _, s in fixed[g] if not fl.startswith("!"))
            cap.append(max(0.0, BUDGET - used))
    return cap


class Stream:
    def __init__(self, x, start, deadline, e_deadline, c_earliest):
        self.x, self.start, self.deadline, self.e_deadline, self.c_earliest = x, start, deadline, e_deadline, c_earliest
        self.e_done = self.ac_done = 0
        self.e_gap = {}

    def done(self):
        return self.ac_done == 48

    def candidate(self, gap):
        if gap < self.start:
            return None
        qa = self.ac_done // 3
        if qa < 16 and qa in self.e_gap and self.e_gap[qa] < gap and gap >= self.c_earliest(qa):
            return (("a0", "a1", "c")[self.ac_done % 3], qa, W_VALU)
        qe = self.e_done // 2
        if qe < 16 and qe - qa < 2:
            return (("e0", "e1")[self.e_done % 2], qe, W_EXP)
        return None

    def place(self, op, q, gap):
        if op[0] == "e":
            if gap > self.e_deadline:
                raise SystemExit(f"stream {self.x} pair {q} exp in gap {gap} > {self.e_deadline}")
            self.e_done += 1
            if op == "e1":
                self.e_gap[q] = gap
        else:
            self.ac_done += 1
            if op == "c" and gap > self.deadline(q):
                raise SystemExit(f"stream {self.x} pair {q} packed in gap {gap} > deadline {self.deadline(q)}")


def schedule():
    cap = capacity()
    s1 = Stream(1, 0, lambda q: NG - 1, NG - 1, lambda q: 0)
    s0 = Stream(0, 36, lambda q: 2 * NG - 1, 2 * NG - 1, lambda q: 40 + 8 * (q >> 2))
    placed = {g: [] for g in range(2 * NG)}
    for g in range(2 * NG):
        room = cap[g]
        for st in (s1, s0):
            while not st.done():
                cand = st.candidate(g)
                if cand is None or cand[2] > room + 1e-9:
                    break
                op, q, cost = cand
                room -= cost
                placed[g].append((st.x, op, q))
                st.place(op, q, g)
    for st in (s1, s0):
        if not st.done():
            raise SystemExit(f"stream {st.x} does not fit: e {st.e_done} ac {st.ac_done}")
    return placed, cap


def emit():
    placed, cap = schedule()
    last = {1: max(g for g in placed for o in placed[g] if o[0] == 1), 0: max(g for g in placed for o in placed[g] if o[0] == 0)}
    lines = ["// GENERATED by tools/gen_attn_w64_mock16.py (dev timing mock, -DW64_MOCK16=1) -- not part of the shipped library.", ""]
    for ph in ("A", "B"):
        x = 0 if ph == "A" else 1
        fixed = A_FIXED if ph == "A" else B_FIXED
        base = 0 if ph == "A" else NG
        lines.append(f"template <int MODE, bool QK, bool PV, bool S0, bool S1> MMPL_DEV void w64_phase_{ph.lower()}(Ctx& k) {{")
        for g in range(NG):
            load = BUDGET - cap[base + g] + sum((W_EXP if o[1][0] == "e" else W_VALU) for o in placed[base + g])
            lines.append(f"  // ---- gap {g}: {load:.1f} slots")
            if g < 32:
                lines.append(f"  if constexpr (QK) k.template mfma_qk_h<{x}, {g >> 1}, {g & 1}>();")
            else:
                lines.append(f"  if constexpr (PV) k.template mfma_pv_h<{x}, {(g - 32) >> 1}, {g & 1}>();")
            if g == 32:
                lines.append("  if constexpr (QK && !PV) k.mfma_write_pad();")
            sm = list(placed[base + g])
            body = []
            if sm:
                body.append(sm.pop(0))
            body += [("F",) + f for f in fixed[g]]
            body += sm
            for o in body:
                if o[0] == "F":
                    _, fl, stmt, _ = o
                    lines.append(f"  if constexpr ({fl}) {stmt}")
                else:
                    xs, op, q = o
                    lines.append(f"  if constexpr (S{xs}) k.template sm_{op}<MODE, {xs}, {q}>();")
            for xs in (1, 0):
                if last[xs] == base + g:
                    lines.append(f"  if constexpr (S{xs}) k.template finish<MODE, {xs}>();")
        lines.append("}")
        lines.append("")
    with open(OUT, "w") as fh:
        fh.write("\n".join(lines))
    tot = [BUDGET - cap[g] + sum((W_EXP if o[1][0] == "e" else W_VALU) for o in placed[g]) for g in range(2 * NG)]
    print("slots per gap A:", " ".join(f"{t:.1f}" for t in tot[:NG]))
    print("slots per gap B:", " ".join(f"{t:.1f}" for t in tot[NG:]))
    print("stream 1 last gap", last[1], " stream 0 last gap", last[0], " mean", sum(tot) / (2 * NG), " budget", BUDGET)


if __name__ == "__main__":
    emit()
