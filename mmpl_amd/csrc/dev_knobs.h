// Every development-only build parameter of the kernel sources, in ONE place.
//
// A product build (mmpl_amd/build.py, __graft_entry__.build()) never defines MMPL_DEV_ABLATIONS: every knob below then has its
// shipped value, cannot be overridden from the command line (#error), and the timing / ablation bodies guarded by
// `#ifdef MMPL_DEV_ABLATIONS` are not compiled at all.  The sweep scripts under tools/ build with
// MMPL_EXTRA_HIPCC_FLAGS="-DMMPL_DEV_ABLATIONS -D<knob>=<value>"; results of ablation builds are garbage by design.
#pragma once

#ifndef MMPL_DEV_ABLATIONS
#if defined(W64_ABL) || defined(W64_LMIN_EXP) || defined(W64_LMAX_EXP) || defined(W64_REF_OFFSET) || defined(W64_REF_TILES) ||           \
    defined(GEMM_POLICY_A) || defined(GEMM_POLICY_W) || defined(GEMM6_ABL) || defined(GEMM6_STORE) || defined(GEMM6_RESLD) ||            \
    defined(GEMM6_TIMING) || defined(GEMM8_ABL) || defined(GEMM8_BAR1) || defined(GEMM8_DMAS) || defined(GEMM8_BAR2) || defined(GEMM8_RD) || \
    defined(GEMM8_FINEWAIT) || defined(GEMM8_PF)
#error "development knob on the command line without -DMMPL_DEV_ABLATIONS (mmpl_amd/csrc/dev_knobs.h)"
#endif
#endif

// ---- attn_w64.hip
// W64_ABL: timing ablations (bits: 1 no LDS-DMA in the loop, 2 no softmax, 4 no fragment reads, 8 no barrier / waits, 32 / 64 / 128
// no exp / row-sum adds / bf16 packs).  The FAST pass's window 2^-LMIN <= l <= 2^LMAX, its reference offset and sample size.
#ifndef W64_ABL
#define W64_ABL 0
#endif
#ifndef W64_LMIN_EXP
#define W64_LMIN_EXP 100
#endif
#ifndef W64_LMAX_EXP
#define W64_LMAX_EXP 100
#endif
#ifndef W64_REF_OFFSET
#define W64_REF_OFFSET 64
#endif
#ifndef W64_REF_TILES
#define W64_REF_TILES 4
#endif

// ---- gemm.hip
// GEMM_POLICY_A / _W: cache-policy modifier of the LDS-DMA loads of the activation / weight operand in the large-problem kernels:
// 0 none, 1 nt, 2 sc1, 3 sc0 sc1 (profiles/r05*_gemm_policy*.log: rejected).
#ifndef GEMM_POLICY_A
#define GEMM_POLICY_A 0
#endif
#ifndef GEMM_POLICY_W
#define GEMM_POLICY_W 0
#endif
// GEMM6_ABL (results are garbage): 1 = every block's LDS-DMA reads operand tile (0, 0): same instruction stream and LDS traffic, every
// fetch an L2 hit -> what the loop costs without fabric / HBM latency; 2 = no epilogue at all (what a perfectly overlapped epilogue
// would leave); 4 = no LDS-DMA in the k loop; 8 = the whole staged epilogue except its global stores; 16 = no residual loads in the
// staged epilogue (a constant instead: the upper bound of what issuing them under the last k-tile could hide)
#ifndef GEMM6_ABL
#define GEMM6_ABL 0
#endif
#ifndef GEMM6_STORE
#define GEMM6_STORE 1       // cache policy of the staged epilogue's C stores: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1 (write-through, no L2 allocate)
#endif
#ifndef GEMM6_RESLD
#define GEMM6_RESLD 0       // residual loads of the staged epilogue: 0 plain, 1 nt
#endif
#ifndef GEMM6_TIMING
#define GEMM6_TIMING 0      // 1 = every wave leaves { prologue, k loop, epilogue } shader cycles over the output (tools/bench_kernels.py gemmphases)
#endif
#ifndef GEMM8_ABL
#define GEMM8_ABL 0         // ablations of the v8 loop (results are garbage): 1 no LDS-DMA, 4 no fragment reads, 8 no barriers / waits
#endif
// gap placement of one v8 k-tile (profiles/r04d_gemm_v8_sweep.log)
#ifndef GEMM8_BAR1
#define GEMM8_BAR1 36        // phase 1: lgkmcnt(0) + barrier (behind the 16 second-half reads: gaps 0, RD, .. 15 RD)
#endif
#ifndef GEMM8_DMAS
#define GEMM8_DMAS 3         // gaps between DMA pieces (never back to back: the four waves run in step and the CU has one address path)
#endif
#ifndef GEMM8_BAR2
#define GEMM8_BAR2 30        // phase 2: vmcnt + barrier, then the next tile's first-half reads in gaps BAR2 + RD, + 2 RD, ..
#endif
#ifndef GEMM8_RD
#define GEMM8_RD 2           // gaps between fragment reads (16 per phase): phase 1 in gaps 0, RD, 2 RD ..; phase 2 in gaps BAR2 + RD, BAR2 + 2 RD ..
#endif
#ifndef GEMM8_FINEWAIT
#define GEMM8_FINEWAIT 1     // 1: the loop-top wait for the first-half fragments is split per activation fragment (counted lgkmcnt) instead of
#endif                       //    one lgkmcnt(0): only the 8 weight fragments + the first activation fragment gate the first MFMA
#ifndef GEMM8_PF
#define GEMM8_PF 0           // 1: two L2 prefetch ops per wave and k-tile behind the DMA pieces (G8::prefetch).  Measured on v8
#endif                       // (profiles/r04d_gemm_v8_sweep.log): ffn2 +4 % (= v6's level), but qkv -1 %, ffn0 -4 %: off, ffn2 stays on v6
