// Run-time switches of libmmpl_hip.so: ONE struct, filled ONCE from the environment on first use (device_state.hip), read by
// every launcher.  They exist for A/B measurements and for the parity tests of the non-default paths (each is exercised by a
// child-process test in tests/test_kernels_gpu.py); production runs set none of them.  Switches whose A/B is settled are removed
// with their logs as the record (round 5: MMPL_GEMM_DIRECT_EPILOGUE / _STATIC_TILES / _NO_SYNC_SWEEPS, MMPL_VAE_NO_HALO --
// profiles/r03d_*, r03C_*, r03_vae_decode_ladder.md; MMPL_CROSS_NO_COLLAPSE: the collapse is now the caller's explicit `cross_rows`).
//
//   MMPL_ATTN_V1=1               self-attention on the lock-step kernel (attn_fwd_kernel) instead of attn_w64_kernel
//   MMPL_ATTN_NOSPLIT=1          no split-KV tail round
//   MMPL_ATTN_NO_MERGE=1         do not merge contiguous KV pages into longer ones
//   MMPL_CROSS_W64=1             text / image cross-attention on attn_w64_kernel
//   MMPL_GEMM_V1=1 / _V2=1       every GEMM on the 128x128 register-staged / the 256x128 3-stage-DMA kernel
//   MMPL_GEMM_GROUP=n            M-tile group of v6's tile order (default: per shape)
//   MMPL_GEMM_PF=n               v6's L2 prefetch distance in k-tiles (default 2; 0 = off)
//   MMPL_GEMM_NO_SPLITK=1        v6 without the split-K launch for the partial last round of tiles
//   MMPL_GEMM_NO_SUBTILE=1       short-K GEMMs without the 128 x 128 sub-tile launch for the partial last round of tiles
//   MMPL_GEMM_V8=0|1             large GEMMs never / always on gemm_bf16_v8_kernel (default: the launcher's per-shape choice)
//   MMPL_VAE_NO_FUSE_NORM=1      RMS_norm + SiLU of the 96-channel layers as its own pass instead of the producing conv's epilogue
//   MMPL_LN_PIPELINE_MIN_ROWS=n  rows from which LayerNorm takes the pipelined kernel (default 16384; 0 = always, a huge n = never)
//   MMPL_CHECK_SHARE=1           debug guard of mmpl_dit_forward's share_in promise: the producer (share_out) and the consumer (share_in)
//                                fingerprint their visible layer-0 K / V on the device; a mismatch is an error (include/mmpl_hip.h)
#pragma once

struct MmplRuntimeConfig {
  bool attn_v1, attn_nosplit, attn_no_merge, cross_w64;
  bool gemm_v1, gemm_v2, gemm_no_splitk, gemm_no_subtile;
  int gemm_group;      // 0 = launcher's choice
  int gemm_pf;         // k-tiles
  int gemm_v8;         // -1 = launcher's choice, 0 / 1 = never / always the one-wave-per-SIMD kernel for the main launch
  bool vae_no_fuse_norm;
  int ln_pipeline_min_rows;
  bool check_share;
};
const MmplRuntimeConfig& mmpl_config();
