#!/usr/bin/env python3
"""Generates mmpl_amd/csrc/attn_w64_sched.inc: the hand-placed instruction stream of the two phases of attn_w64_kernel.

The kernel (attn_w64.hip) runs one wave per SIMD, so everything that is not an MFMA has to be issued in the 32-cycle shadow of
one: at most ~5 issue slots per MFMA gap (MI355X_MICROARCH.md, "one wave per SIMD").  This script owns WHERE every non-MFMA
instruction of a KV tile goes: per phase 32 gaps (gap g = what follows MFMA g), fixed fillers first (LDS fragment reads, the
LDS-DMA pieces, the barrier, cursor bookkeeping), then the two softmax streams are list-scheduled into what is left:

    stream 1 = softmax of S_B(j-1): lives entirely in phase A (B's first MFMA overwrites S_B, which the end-of-tile check's slow
               path may have to re-read)
    stream 0 = softmax of S_A(j)  : may start at A gap 18 (two MFMAs after S_A is complete), ends with phase B; P_A[ks] may only be
               overwritten after A's MFMA 19 + 4 ks has read the previous tile's

Per register pair q (16 per stream): e0 e1 (v_exp_f32; GENERAL mode: v_sub + v_exp) ... one gap later ... a0 a1 (row-sum adds)
c (v_cvt_pk_bf16_f32).  The output is straight-line C++ (calls of Ctx members with literal template arguments, each guarded
by the phase's compile-time flags), committed to the repo; re-run after changing CAP / placement:

    python tools/gen_attn_w64.py
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "mmpl_amd", "csrc", "attn_w64_sched.inc")

# Issue-slot weights and slots per gap besides the MFMA.  Measured with the timing build (tools/w64_gen_sweep.sh, shader cycles per
# KV tile of the steady loop, 14B/720p stage 3): BUDGET 5.5 -> 2480, 5.0 -> 2424, 6.0 -> 2545, 7.0 -> 2568 (floor: 64 MFMAs = 2114);
# 4.5 does not fit stream 1 into phase A.
W_EXP, W_VALU = float(os.environ.get("W64_WEXP", 1.5)), 1.0
BUDGET = float(os.environ.get("W64_BUDGET", 5.0))

# ---- fixed fillers: (flag, statement, slots)
A_FIXED = {g: [] for g in range(32)}
B_FIXED = {g: [] for g in range(32)}
# The LDS-DMA pieces of one event (4 K + 4 V per wave; gaps 0..31 = phase A, 32..63 = phase B).  All four waves of the block run
# the same stream in step (one barrier per tile), so pieces in consecutive gaps arrive at the CU's one address path 4 at a
# time, 8 times in a row: measured 145 cycles per tile of issue back-pressure; spread out they cost ~0.
K_DMA = [int(x) for x in os.environ.get("W64_KDMA", "4,8,12,16").split(",")]
V_DMA = [int(x) for x in os.environ.get("W64_VDMA", "20,24,28,36").split(",")]
W_DMA = float(os.environ.get("W64_WDMA", BUDGET - 1))


def _fixed(g):
    return A_FIXED[g] if g < 32 else B_FIXED[g - 32]


A_FIXED[0].append(("PV", "k.template lds_v<15>();", 2))
A_FIXED[3].append(("QK", "k.barrier();", BUDGET))
A_FIXED[3].append(("!QK", "k.wait_lgkm0();", 1))
for i in range(4):
    _fixed(K_DMA[i]).append(("QK", f"k.template dma_k<{i}>();", W_DMA))
    _fixed(V_DMA[i]).append(("QK", f"k.template dma_v<{i}>();", W_DMA))
_fixed(max(K_DMA) + 3).append(("QK", "k.advance_k();", 1))
_fixed(max(V_DMA) + 3).append(("QK", "k.advance_v();", 1))
B_FIXED[0].append(("QK", "k.addr_k();", 1))
for f in range(16):
    B_FIXED[1 + f].append(("QK", f"k.template lds_k<{f}>();", 1 if f & 1 or f == 0 else 2))
B_FIXED[16].append(("QK", "k.addr_v();", 4))
for i in range(15):
    if i == 5:
        B_FIXED[17 + i].append(("QK", "k.template wait_lgkm<10>();", 0.5))     # the 16 K reads are older than the 10 V reads since
    B_FIXED[17 + i].append(("QK", f"k.template lds_v<{i}>();", 2))
B_FIXED[30].append(("QK", "k.rotate();", 1))


def capacity():
    cap = []
    for ph, fixed in (("A", A_FIXED), ("B", B_FIXED)):
        for g in range(32):
            used = sum(s for fl, _, s in fixed[g] if not fl.startswith("!"))
            cap.append(max(0.0, BUDGET - used))
    return cap


class Stream:
    """op-level state of one softmax stream: per pair q the ops e0 e1, then (a later gap) a0 a1 c; <= 2 pairs in flight"""

    def __init__(self, x, start, deadline, e_deadline, c_earliest):
        self.x, self.start, self.deadline, self.e_deadline, self.c_earliest = x, start, deadline, e_deadline, c_earliest
        self.e_done = 0          # e-ops placed (2 per pair)
        self.ac_done = 0         # a/c ops placed (3 per pair)
        self.e_gap = {}          # pair -> gap its e1 was placed in

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
    s1 = Stream(1, 0, lambda q: 31, 31, lambda q: 0)
    s0 = Stream(0, 18, lambda q: 63, 63, lambda q: 20 + 4 * (q >> 2))
    placed = {g: [] for g in range(64)}
    for g in range(64):
        room = cap[g]
        for st in (s1, s0):            # stream 1 (tight window) has strict priority, stream 0 takes what is left
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


def interleave(ops):
    """order the softmax ops of one gap so that ops of the two streams alternate (no statement feeds its neighbour)"""
    a = [o for o in ops if o[0] == 1]
    b = [o for o in ops if o[0] == 0]
    out = []
    while a or b:
        if a:
            out.append(a.pop(0))
        if b:
            out.append(b.pop(0))
    return out


def emit():
    placed, cap = schedule()
    last = {1: max(g for g in placed for o in placed[g] if o[0] == 1), 0: max(g for g in placed for o in placed[g] if o[0] == 0)}
    lines = ["// GENERATED by tools/gen_attn_w64.py -- do not edit; see that script for the placement rules.",
             "// k: Ctx (attn_w64.hip).  MODE 0 = FAST (reference 0, p = exp2(s)), 1 = GENERAL (p = exp2(s - m_ref)).",
             "// Flags: QK / PV = this phase has the S = K.Q / O += V.P MFMAs; S0 / S1 = softmax stream of block A / B is live.", ""]
    for ph in ("A", "B"):
        x = 0 if ph == "A" else 1
        fixed = A_FIXED if ph == "A" else B_FIXED
        base = 0 if ph == "A" else 32
        lines.append(f"template <int MODE, bool QK, bool PV, bool S0, bool S1> MMPL_DEV void w64_phase_{ph.lower()}(Ctx& k) {{")
        for g in range(32):
            load = BUDGET - cap[base + g] + sum((W_EXP if o[1][0] == "e" else W_VALU) for o in placed[base + g])
            lines.append(f"  // ---- gap {g}: {load:.1f} slots")
            if g < 16:
                lines.append(f"  if constexpr (QK) k.template mfma_qk<{x}, {g}>();")
            else:
                lines.append(f"  if constexpr (PV) k.template mfma_pv<{x}, {g - 16}>();")
            if g == 16:
                # a phase without the PV MFMAs (the first tile): nothing covers the 19 wait states between the last QK MFMA's
                # write of S and the first VALU read of it (a software hazard, not interlocked)
                lines.append("  if constexpr (QK && !PV) k.mfma_write_pad();")
            sm = interleave(placed[base + g])
            fx = list(fixed[g])
            body = []
            # fixed fillers go after the first softmax op (so that an LDS / DMA op does not sit right behind the MFMA issue)
            if sm:
                body.append(sm.pop(0))
            body += [("F",) + f for f in fx]
            body += sm
            for o in body:
                if o[0] == "F":
                    _, fl, stmt, _ = o
                    cond = {"QK": "QK", "PV": "PV", "!QK": "!QK", "S0": "S0", "S1": "S1"}[fl]
                    lines.append(f"  if constexpr ({cond}) {stmt}")
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
    # report
    tot = [BUDGET - cap[g] + sum((W_EXP if o[1][0] == "e" else W_VALU) for o in placed[g]) for g in range(64)]
    print("slots per gap A:", " ".join(f"{t:.1f}" for t in tot[:32]))
    print("slots per gap B:", " ".join(f"{t:.1f}" for t in tot[32:]))
    print("stream 1 last gap", last[1], " stream 0 last gap", last[0], " mean", sum(tot) / 64)


if __name__ == "__main__":
    emit()
