// Kernels of the Wan 3D causal VAE (decode + encode) on channels-last bf16 volumes.
//
// Every CausalConv3d / Conv2d is an implicit GEMM straight out of a zero-bordered ("padded") channels-last
// volume: output pixel m -> a base pixel in the padded source, each filter tap is a constant pixel offset from it, and
// the K axis is (tap, cin) with cin contiguous -- so the A operand rows are plain 64-byte runs of HBM and no im2col
// buffer ever exists.  The temporal "feat_cache" of the reference (vae.py:14, 207-216) is the first two time slots of
// each conv's persistent padded volume.  Tile 128x128x32, 4 waves, v_mfma_f32_16x16x32_bf16, XOR-swizzled LDS,
// register-staged double buffering (same structure as gemm.hip).  The norm / activation / resampling passes between
// convs are HBM-bound 16-byte-per-lane kernels that write directly into the next conv's padded volume.
#include "common.h"
#include "vae_kernels.h"

namespace {

constexpr int BM = 128, BK = 32;        // the N tile is a template parameter of conv_igemm_kernel
constexpr int TILE_BYTES = BM * BK * 2;  // 8 KiB

// swizzled byte offset of 16-B chunk c (0..3) of row r inside a [128][32] bf16 tile: chunk' = c ^ ((-(r>>2)) & 3)
MMPL_DEV int swz(int r, int c) { return r * 64 + ((c ^ ((0 - (r >> 2)) & 3)) << 4); }

// NF = 16-column fragments per wave along N: 4 -> a 128-wide N tile, 3 -> 96-wide (the C = 96 layers of the decoder's full-resolution
// stage, 40 % of a decode's conv time, would otherwise multiply 32 padding columns per tile)
template <int NF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4))) void conv_igemm_kernel(ConvArgs g) {
  constexpr int BN = 32 * NF;
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];
  char* As = smem;
  char* Ws = smem + 2 * TILE_BYTES;
  const int tiles_m = (g.M + BM - 1) / BM;
  const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;   // consecutive blocks share the weight panel
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // staging roles: 128 rows x 4 chunks = 512 chunks per operand -> 2 per thread
  const int srow = tid >> 2, schunk = tid & 3;
  const int K = g.ntaps * g.Cin;
  size_t a_base[2];
  const bf16_t* w_ptr[2];
  int lds_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = srow + 64 * j;
    const int m = min(m0 + row, g.M - 1);
    const int x = m % g.Wo, y = (m / g.Wo) % g.Ho, t = m / (g.Wo * g.Ho);
    a_base[j] = ((size_t)(t * g.st) * g.Hp + (size_t)y * g.sy) * g.Wp + (size_t)x * g.sx;   // pixel index
    w_ptr[j] = g.W + (size_t)min(n0 + row, g.N - 1) * K + schunk * 8;
    lds_off[j] = swz(row, schunk);
  }
  const int nt = K / BK;
  const int tiles_per_tap = g.Cin / BK;
  u32x4 ra[2], rw[2];
  auto load = [&](int t) {
    const int tap = t / tiles_per_tap, c0 = (t - tap * tiles_per_tap) * BK;
    const int toff = g.tap_off[tap];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      ra[j] = *reinterpret_cast<const u32x4*>(g.src + (a_base[j] + toff) * g.Cin + c0 + schunk * 8);
      rw[j] = *reinterpret_cast<const u32x4*>(w_ptr[j] + (size_t)t * BK);
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      *reinterpret_cast<u32x4*>(As + buf * TILE_BYTES + lds_off[j]) = ra[j];
      *reinterpret_cast<u32x4*>(Ws + buf * TILE_BYTES + lds_off[j]) = rw[j];
    }
  };
  load(0);
  store(0);
  __syncthreads();

  f32x4 acc[4][NF];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fchunk = lane >> 4;
  int a_off[4], w_off[NF];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_off[i] = swz(64 * wm + 16 * i + frow, fchunk);
#pragma unroll
  for (int j = 0; j < NF; ++j) w_off[j] = swz(16 * NF * wn + 16 * j + frow, fchunk);
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) load(t + 1);
    const char* Ac = As + cur * TILE_BYTES;
    const char* Wc = Ws + cur * TILE_BYTES;
    bf16x8 af[4], wf[NF];
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(Ac + a_off[i]);
#pragma unroll
    for (int j = 0; j < NF; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(Wc + w_off[j]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    if (t + 1 < nt) store(cur ^ 1);
    __syncthreads();
  }
  // epilogue: lane holds column m (one output pixel) x 4 consecutive output channels per fragment
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + 64 * wm + 16 * i + frow;
    if (m >= g.M) continue;
    const int x = m % g.Wo, y = (m / g.Wo) % g.Ho, t = m / (g.Wo * g.Ho);
    const size_t dpix = ((size_t)(t + g.dt0) * g.Hd + (y + g.dy0)) * g.Wd + (x + g.dx0);
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      const int n = n0 + 16 * NF * wn + 16 * j + 4 * fchunk;
      if (n >= g.N) continue;
      float v[4];
      const u32x2 bb = *reinterpret_cast<const u32x2*>(g.bias + n);
      v[0] = rbf(acc[i][j][0] + bf2f(bb.x & 0xffff)); v[1] = rbf(acc[i][j][1] + bf2f(bb.x >> 16));
      v[2] = rbf(acc[i][j][2] + bf2f(bb.y & 0xffff)); v[3] = rbf(acc[i][j][3] + bf2f(bb.y >> 16));
      if (g.res) {
        const u32x2 rr = *reinterpret_cast<const u32x2*>(g.res + (size_t)m * g.ldres + n);
        v[0] += bf2f(rr.x & 0xffff); v[1] += bf2f(rr.x >> 16); v[2] += bf2f(rr.y & 0xffff); v[3] += bf2f(rr.y >> 16);
      }
      u32x2 o;
      o.x = pack2bf(v[0], v[1]);
      o.y = pack2bf(v[2], v[3]);
      *reinterpret_cast<u32x2*>(g.dst + dpix * g.ldd + g.dc0 + n) = o;
    }
  }
}

// ---- RMS_norm (F.normalize over C * sqrt(C) * gamma, vae.py:51-54) [+ SiLU] -> padded destination.  LG lanes per pixel, 16 bytes
// (8 channels) per lane per pass: LG = 16 for C <= 128, 32 for C <= 256, else the whole wave with up to two passes (C <= 1024) --
// so a wave normalises 4 / 2 / 1 pixels and every lane moves data (with one pixel per wave the C = 96 layers of the decoder's
// full-resolution stage kept 12 of 64 lanes busy and the kernel ran at a quarter of the HBM rate: 27 % of a decode).
template <int LG>
__global__ __launch_bounds__(256) void norm_act_pad_kernel(NormArgs a) {
  constexpr int PPW = 64 / LG, NIT = LG == 64 ? 2 : 1;
  const int lane = threadIdx.x & 63, sub = lane % LG;
  const long pix = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * PPW + lane / LG;
  const bool live = pix < a.npix;
  const int nchunk = a.C >> 3;
  const bf16_t* sp = a.src + (size_t)(live ? pix : 0) * a.C;
  float v[NIT][8];
  float sq = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int ch = sub + LG * it;
    if (live && ch < nchunk) {
      const u32x4 u = *reinterpret_cast<const u32x4*>(sp + ch * 8);
      v[it][0] = bf2f(u.x & 0xffff); v[it][1] = bf2f(u.x >> 16); v[it][2] = bf2f(u.y & 0xffff); v[it][3] = bf2f(u.y >> 16);
      v[it][4] = bf2f(u.z & 0xffff); v[it][5] = bf2f(u.z >> 16); v[it][6] = bf2f(u.w & 0xffff); v[it][7] = bf2f(u.w >> 16);
#pragma unroll
      for (int j = 0; j < 8; ++j) sq += v[it][j] * v[it][j];
    }
  }
  float denom = 1.f;
  if (a.gamma) {
#pragma unroll
    for (int o = LG / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);          // sum over the pixel's LG lanes
    denom = fmaxf(rbf(sqrtf(sq)), 1e-12f);                                       // torch.norm output is a bf16 tensor, clamp_min(eps)
  }
  if (!live) return;
  const int x = (int)(pix % a.W), y = (int)((pix / a.W) % a.H), t = (int)(pix / ((long)a.W * a.H));
  bf16_t* dp = a.dst + (((size_t)(t + a.dt0) * a.Hd + (y + a.dy0)) * a.Wd + (x + a.dx0)) * a.ldd;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int ch = sub + LG * it;
    if (ch >= nchunk) continue;
    float o[8];
    if (a.gamma) {
      const u32x4 gu = *reinterpret_cast<const u32x4*>(a.gamma + ch * 8);
      const float gm[8] = {bf2f(gu.x & 0xffff), bf2f(gu.x >> 16), bf2f(gu.y & 0xffff), bf2f(gu.y >> 16),
                           bf2f(gu.z & 0xffff), bf2f(gu.z >> 16), bf2f(gu.w & 0xffff), bf2f(gu.w >> 16)};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float n = rbf(v[it][j] / denom);            // x / norm      (bf16 tensor)
        n = rbf(n * a.scale);                       // * sqrt(C)
        n = rbf(n * gm[j]);                         // * gamma
        o[j] = a.silu ? silu(n) : n;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = v[it][j];
    }
    u32x4 w;
    w.x = pack2bf(o[0], o[1]); w.y = pack2bf(o[2], o[3]); w.z = pack2bf(o[4], o[5]); w.w = pack2bf(o[6], o[7]);
    *reinterpret_cast<u32x4*>(dp + ch * 8) = w;
  }
}

// ---- nearest 2x spatial upsample (+ optional temporal de-interleave of a time_conv output [T,H,W,2C] -> frames
// 2t, 2t+1 taking channel halves, vae.py:134-137) into a padded destination.
__global__ void upsample_pad_kernel(UpArgs a) {
  const long nchunk_c = a.C >> 3;
  const long total = (long)a.To * (2 * a.H) * (2 * a.W) * nchunk_c;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % nchunk_c);
    long p = i / nchunk_c;
    const int X = (int)(p % (2 * a.W)); p /= (2 * a.W);
    const int Y = (int)(p % (2 * a.H));
    const int T = (int)(p / (2 * a.H));
    const int ts = a.interleave ? (T >> 1) : T, half = a.interleave ? (T & 1) : 0;
    const bf16_t* sp = a.src + (((size_t)ts * a.H + (Y >> 1)) * a.W + (X >> 1)) * a.lds + half * a.C + c * 8;
    bf16_t* dp = a.dst + (((size_t)T * a.Hd + (Y + 1)) * a.Wd + (X + 1)) * a.C + c * 8;
    *reinterpret_cast<u32x4*>(dp) = *reinterpret_cast<const u32x4*>(sp);
  }
}

// ---- z prep: latent [F,16,h,w] -> (z / inv_std + mean) -> conv2 (1x1x1, 16->16) -> decoder.conv1's padded volume
// (channels padded 16 -> 32 with zeros).   vae.py:548-554
__global__ void z_prep_kernel(ZPrepArgs a) {
  const long total = (long)a.F * a.h * a.w;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % a.w), y = (int)((i / a.w) % a.h), f = (int)(i / ((long)a.w * a.h));
    float z[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const float v = bf2f(a.z[(((size_t)f * 16 + c) * a.h + y) * a.w + x]);
      z[c] = rbf(rbf(v / a.inv_std[c]) + a.mean[c]);
    }
    bf16_t* dp = a.dst + (((size_t)(f + a.dt0) * (a.h + 2) + (y + 1)) * (a.w + 2) + (x + 1)) * 32;
#pragma unroll
    for (int o = 0; o < 16; ++o) {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) acc += z[c] * bf2f(a.w2[o * 16 + c]);
      dp[o] = f2bf(acc + bf2f(a.b2[o]));
    }
  }
}

// ---- head output [T,H,W,4] bf16 -> float32 [T,3,H,W] (frame offset t_out), clamp(-1,1)   wan_wrapper.py:108
__global__ void px_out_kernel(const bf16_t* src, float* out, int T, int H, int W, int t_out) {
  const long total = (long)T * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long hw = (long)H * W;
    const int t = (int)(i / hw);
    const long p = i - (long)t * hw;
    const u32x2 u = *reinterpret_cast<const u32x2*>(src + i * 4);
    const float v[3] = {bf2f(u.x & 0xffff), bf2f(u.x >> 16), bf2f(u.y & 0xffff)};
#pragma unroll
    for (int c = 0; c < 3; ++c) out[((size_t)(t + t_out) * 3 + c) * hw + p] = fminf(1.f, fmaxf(-1.f, v[c]));
  }
}

// ---- pixels [3,Ttot,H,W] bf16 (frames t0..t0+T) -> padded channels-last volume [.., H+2, W+2, 32] (c >= 3 zero)
__global__ void px_in_kernel(const bf16_t* px, bf16_t* dst, int Ttot, int t0, int T, int H, int W, int dt0) {
  const long total = (long)T * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long hw = (long)H * W;
    const int t = (int)(i / hw);
    const long p = i - (long)t * hw;
    const int y = (int)(p / W), x = (int)(p % W);
    bf16_t* dp = dst + (((size_t)(t + dt0) * (H + 2) + (y + 1)) * (W + 2) + (x + 1)) * 32;
#pragma unroll
    for (int c = 0; c < 3; ++c) dp[c] = px[((size_t)c * Ttot + (t0 + t)) * hw + p];
  }
}

// ---- mu = ((conv1_1x1x1(enc)[0:16]) - mean) * inv_std   vae.py:536-541;  enc: [F*h*w, 32] -> out float32 [F,16,h,w]
__global__ void mu_out_kernel(MuArgs a) {
  const bf16_t* enc = a.enc; const bf16_t* w1 = a.w1; const bf16_t* b1 = a.b1; float* out = a.out;
  const float* mean = a.mean; const float* inv_std = a.inv_std;
  const int F = a.F, f_out = a.f_out, h = a.h, w = a.w;
  const long total = (long)F * h * w;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long hw = (long)h * w;
    const int f = (int)(i / hw);
    const long p = i - (long)f * hw;
    float e[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) e[c] = bf2f(enc[i * 32 + c]);
#pragma unroll
    for (int o = 0; o < 16; ++o) {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < 32; ++c) acc += e[c] * bf2f(w1[o * 32 + c]);
      const float mu = rbf(acc + bf2f(b1[o]));
      out[((size_t)(f + f_out) * 16 + o) * hw + p] = rbf(rbf(mu - mean[o]) * inv_std[o]);
    }
  }
}

// ---- row softmax of fp32 scores [rows, ld] (cols valid, zero-fills the tail up to ldp) -> bf16 P [rows, ldp]
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* s, int ld, bf16_t* p, int ldp, int rows, int cols) {
  const int row = blockIdx.x;
  const float* sp = s + (size_t)row * ld;
  __shared__ float red[4];
  float mx = -INFINITY;
  for (int c = threadIdx.x; c < cols; c += 256) mx = fmaxf(mx, sp[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) sum += __expf(sp[c] - mx);
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
  bf16_t* pp = p + (size_t)row * ldp;
  for (int c = threadIdx.x; c < ldp; c += 256) pp[c] = c < cols ? f2bf(__expf(sp[c] - mx) * inv) : (bf16_t)0;
}

// ---- transpose v [rows, C] (row stride ld) -> vt [C, ldt] (zero tail)
__global__ void transpose_kernel(const bf16_t* v, int ld, bf16_t* vt, int ldt, int rows, int C) {
  __shared__ bf16_t tile[32][33];
  const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 8 rows per pass
  for (int k = ty; k < 32; k += 8) {
    const int r = r0 + k, c = c0 + tx;
    tile[k][tx] = (r < rows && c < C) ? v[(size_t)r * ld + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, r = r0 + tx;
    if (c < C && r < ldt) vt[(size_t)c * ldt + r] = tile[tx][k];
  }
}

inline int grid_for(long n, int block = 256) {
  long g = (n + block - 1) / block;
  return (int)(g > 8192 ? 8192 : (g == 0 ? 1 : g));
}

}  // namespace

hipError_t vae_launch_conv(const ConvArgs& g, hipStream_t s) {
  if (g.M <= 0) return hipSuccess;
  if (g.Cin % 32 || g.N % 4 || g.ntaps < 1 || g.ntaps > 27) return hipErrorInvalidValue;
  if (g.N % 96 == 0 && g.N % 128 != 0) {                     // 96, 288, ...: no padding columns with the 96-wide tile
    const int tiles = ((g.M + BM - 1) / BM) * (g.N / 96);
    hipLaunchKernelGGL(conv_igemm_kernel<3>, dim3(tiles), dim3(256), 0, s, g);
  } else {
    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + 127) / 128);
    hipLaunchKernelGGL(conv_igemm_kernel<4>, dim3(tiles), dim3(256), 0, s, g);
  }
  return hipGetLastError();
}
hipError_t vae_launch_norm(const NormArgs& a, hipStream_t s) {
  if (a.npix <= 0) return hipSuccess;
  if (a.C % 8 || a.C > 1024) return hipErrorInvalidValue;
  if (a.C <= 128) hipLaunchKernelGGL(norm_act_pad_kernel<16>, dim3((unsigned)((a.npix + 15) / 16)), dim3(256), 0, s, a);
  else if (a.C <= 256) hipLaunchKernelGGL(norm_act_pad_kernel<32>, dim3((unsigned)((a.npix + 7) / 8)), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(norm_act_pad_kernel<64>, dim3((unsigned)((a.npix + 3) / 4)), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t vae_launch_upsample(const UpArgs& a, hipStream_t s) {
  const long n = (long)a.To * 4 * a.H * a.W * (a.C / 8);
  hipLaunchKernelGGL(upsample_pad_kernel, dim3(grid_for(n)), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t vae_launch_zprep(const ZPrepArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(z_prep_kernel, dim3(grid_for((long)a.F * a.h * a.w)), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t vae_launch_px_out(const bf16_t* src, float* out, int T, int H, int W, int t_out, hipStream_t s) {
  hipLaunchKernelGGL(px_out_kernel, dim3(grid_for((long)T * H * W)), dim3(256), 0, s, src, out, T, H, W, t_out);
  return hipGetLastError();
}
hipError_t vae_launch_px_in(const bf16_t* px, bf16_t* dst, int Ttot, int t0, int T, int H, int W, int dt0, hipStream_t s) {
  hipLaunchKernelGGL(px_in_kernel, dim3(grid_for((long)T * H * W)), dim3(256), 0, s, px, dst, Ttot, t0, T, H, W, dt0);
  return hipGetLastError();
}
hipError_t vae_launch_mu_out(const MuArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(mu_out_kernel, dim3(grid_for((long)a.F * a.h * a.w)), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t vae_launch_softmax(const float* sc, int ld, bf16_t* p, int ldp, int rows, int cols, hipStream_t s) {
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, s, sc, ld, p, ldp, rows, cols);
  return hipGetLastError();
}
hipError_t vae_launch_transpose(const bf16_t* v, int ld, bf16_t* vt, int ldt, int rows, int C, hipStream_t s) {
  hipLaunchKernelGGL(transpose_kernel, dim3((ldt + 31) / 32, (C + 31) / 32), dim3(256), 0, s, v, ld, vt, ldt, rows, C);
  return hipGetLastError();
}
