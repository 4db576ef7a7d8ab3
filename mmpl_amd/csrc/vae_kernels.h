// Launcher interface of the Wan 3D-VAE kernels (vae_kernels.hip), used by vae.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "common.h"

// implicit-GEMM convolution out of a padded channels-last volume (see vae_kernels.hip)
struct ConvArgs {
  const bf16_t* src;   // padded source volume [Tp, Hp, Wp, Cin]
  const bf16_t* frame[8];   // optional (frame[0] != null; conv_halo_kernel only): source frame j of the volume lives HERE instead of at
                            // src + j * Hp * Wp * Cin -- the temporal cache as a ring of frame slots instead of a shifted copy
  int Cin, Hp, Wp;
  int st, sy, sx;      // output -> source strides (time, y, x)
  int ntaps;
  int kt, kh, kw;      // the filter's extent (ntaps = kt * kh * kw, tap = (a * kh + b) * kw + d)
  int tap_off[27];     // pixel offset of each tap inside the padded source
  const bf16_t* W;     // [N, ntaps * Cin]
  const bf16_t* Wfrag; // optional: the same weights packed fragment-major for conv_halo_kernel:
                       // [Cin / 32][ntaps][ceil(N / 16)][64 lanes][8] with lane = 16 * (k chunk of 8) + (row within the 16)
  const bf16_t* bias;  // [N]
  int M, N, Ho, Wo;    // M = To * Ho * Wo output pixels
  bf16_t* dst;         // destination volume [.., Hd, Wd, ldd]; output pixel (t,y,x) -> (t+dt0, y+dy0, x+dx0), channel dc0+n
  int Hd, Wd, ldd, dt0, dy0, dx0, dc0;
  const bf16_t* res;   // optional residual, plain [M, ldres]
  int ldres;
  // optional (conv_halo_kernel, N == 96 only: one block owns every channel of its pixels): the CONSUMER's RMS_norm + SiLU
  // (vae.py:51-54 + nn.SiLU, the `residual.0/3` / `head.0` modules in front of the next conv) applied to this conv's bf16 output in
  // the epilogue and written into the consumer's padded frame slots -- output frame t, pixel (y, x) -> nframe[t] + ((y + 1) * (Wo + 2)
  // + x + 1) * N.  dst may then be null (nothing else reads the plain output).
  const bf16_t* ngamma; float nscale;
  bf16_t* nframe[8];
};
hipError_t vae_launch_conv(const ConvArgs& g, hipStream_t s);
bool vae_conv_uses_halo(const ConvArgs& g);   // will vae_launch_conv take conv_halo_kernel (the only one that understands ConvArgs.frame)?

struct NormArgs {
  const bf16_t* src;   // plain [npix, C]
  long npix;
  int C, H, W;         // npix = T*H*W
  const bf16_t* gamma; // null: plain copy
  float scale;         // sqrt(C)
  int silu;
  bf16_t* dst;         // destination volume [.., Hd, Wd, ldd]
  int Hd, Wd, ldd, dt0, dy0, dx0;
};
hipError_t vae_launch_norm(const NormArgs& a, hipStream_t s);

struct UpArgs {
  const bf16_t* src;   // [Ts, H, W, lds]  (lds = C, or 2C when interleave)
  int lds, C, H, W;
  int To;              // output frames (= Ts, or 2*Ts when interleave)
  int interleave;
  bf16_t* dst;         // [To, 2H+2, 2W+2, C] (spatially padded by 1)
  int Hd, Wd;
};
hipError_t vae_launch_upsample(const UpArgs& a, hipStream_t s);

struct ZPrepArgs {
  const bf16_t* z;     // [F, 16, h, w]
  int F, h, w;
  float mean[16], inv_std[16];
  const bf16_t* w2; const bf16_t* b2;   // conv2 1x1x1 [16,16], [16]
  bf16_t* dst;         // padded [.., h+2, w+2, 32]
  int dt0;
};
hipError_t vae_launch_zprep(const ZPrepArgs& a, hipStream_t s);
hipError_t vae_launch_px_out(const bf16_t* src, float* out, int T, int H, int W, int t_out, hipStream_t s);
hipError_t vae_launch_px_in(const bf16_t* px, bf16_t* dst, int Ttot, int t0, int T, int H, int W, int dt0, hipStream_t s);
struct MuArgs {
  const bf16_t* enc;   // [F*h*w, 32] encoder head output
  const bf16_t* w1; const bf16_t* b1;   // conv1 1x1x1 [32,32], [32] (only the first 16 outputs = mu are used)
  float mean[16], inv_std[16];
  float* out;          // float32 [.., 16, h, w], frames written at f_out..
  int F, f_out, h, w;
};
hipError_t vae_launch_mu_out(const MuArgs& a, hipStream_t s);
hipError_t vae_launch_softmax(const float* sc, int ld, bf16_t* p, int ldp, int rows, int cols, hipStream_t s);
hipError_t vae_launch_transpose(const bf16_t* v, int ld, bf16_t* vt, int ldt, int rows, int C, hipStream_t s);
