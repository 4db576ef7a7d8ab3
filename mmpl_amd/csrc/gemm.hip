// bf16 GEMM with fused epilogues for the Wan DiT linears:  C[M,N] = epi( A[M,K] . W[N,K]^T + bias[N] ).
//
// A is token-major activations, W is an nn.Linear weight ([out, in], K contiguous) -- both operands are
// K-contiguous, which is exactly the MFMA A/B fragment shape, so no transposes anywhere.
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 fragments of
// v_mfma_f32_16x16x32_bf16.  Operands are staged global -> registers -> XOR-swizzled LDS (conflict-free
// ds_read_b128), double-buffered with the next tile's global loads in flight under the
// current tile's MFMAs; one barrier per K tile.  The MFMA is issued "swapped" (W fragment as the A
// operand) so each lane ends up with 4 consecutive output columns -> 8-byte epilogue loads/stores.
// Rounding points of the epilogues follow the reference's bf16 module boundaries (DESIGN.md "numerics").
#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

template <int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_bf16_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;                   // [2][128][64] bf16, swizzled
  char* Ws = smem + 2 * TILE_BYTES;  // [2][128][64]

  // ---- block -> tile mapping: XCD-aware (blocks b, b+8, ... share an L2) + grouped along M
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;  // bijective for any nwg
  }
  constexpr int GROUP = 8;
  const int per_group = GROUP * tiles_n;
  const int gid = bid / per_group;
  const int first_m = gid * GROUP;
  const int gsz = min(tiles_m - first_m, GROUP);
  const int tm = first_m + (bid % per_group) % gsz;
  const int tn = (bid % per_group) / gsz;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- staging: thread loads 4x16B of A and 4x16B of W per K tile
  const int srow = tid >> 3, schunk = tid & 7;
  const bf16_t* a_ptr[4];
  const bf16_t* w_ptr[4];
  int lds_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = srow + 32 * j;
    a_ptr[j] = g.A + (size_t)min(m0 + row, g.M - 1) * g.lda + schunk * 8;
    w_ptr[j] = g.W + (size_t)min(n0 + row, g.N - 1) * g.ldw + schunk * 8;
    lds_off[j] = row * 128 + ((schunk ^ (row & 7)) << 4);
  }
  u32x4 ra[4], rw[4];
  const int nt = g.K / BK;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    ra[j] = *reinterpret_cast<const u32x4*>(a_ptr[j]);
    rw[j] = *reinterpret_cast<const u32x4*>(w_ptr[j]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    *reinterpret_cast<u32x4*>(As + lds_off[j]) = ra[j];
    *reinterpret_cast<u32x4*>(Ws + lds_off[j]) = rw[j];
  }
  __syncthreads();

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (bytes within a tile): row = base + 16*i + (lane&15), 16-B chunk = 4*ks + (lane>>4)
  const int frow = lane & 15, fchunk = lane >> 4;
  int a_off[4], w_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ar = 64 * wm + 16 * i + frow, wr = 64 * wn + 16 * i + frow;
    a_off[i] = ar * 128 + ((fchunk ^ (ar & 7)) << 4);
    w_off[i] = wr * 128 + ((fchunk ^ (wr & 7)) << 4);
  }

  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) {
      const int koff = (t + 1) * BK;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ra[j] = *reinterpret_cast<const u32x4*>(a_ptr[j] + koff);
        rw[j] = *reinterpret_cast<const u32x4*>(w_ptr[j] + koff);
      }
    }
    const char* Ac = As + cur * TILE_BYTES;
    const char* Wc = Ws + cur * TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // chunk index for k-step ks is (4*ks + fchunk); XOR with (row&7) commutes with adding 4*ks (bit 2)
        af[i] = *reinterpret_cast<const bf16x8*>(Ac + (a_off[i] ^ (ks << 6)));
        wf[i] = *reinterpret_cast<const bf16x8*>(Wc + (w_off[i] ^ (ks << 6)));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < nt) {
      char* An = As + (cur ^ 1) * TILE_BYTES;
      char* Wn = Ws + (cur ^ 1) * TILE_BYTES;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        *reinterpret_cast<u32x4*>(An + lds_off[j]) = ra[j];
        *reinterpret_cast<u32x4*>(Wn + lds_off[j]) = rw[j];
      }
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds, per fragment (i,j), column m = ..+(lane&15) and rows n = ..+4*(lane>>4)+{0..3}
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + 64 * wm + 16 * i + frow;
    if (m >= g.M) continue;
    const int frame = (EPI == EPI_GATE_RES) ? (m / g.rows_per_frame) : 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + 64 * wn + 16 * j + 4 * fchunk;
      if (n >= g.N) continue;  // N is a multiple of 4 for every caller
      if (EPI == EPI_F32_SCALE) {
        f32x4 o4;
#pragma unroll
        for (int r = 0; r < 4; ++r) o4[r] = acc[i][j][r] * g.alpha;
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.C) + (size_t)m * g.ldc + n) = o4;
        continue;
      }
      float v[4];
      {
        float b[4] = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) {
          const uint2 bb = *reinterpret_cast<const uint2*>(g.bias + n);
          b[0] = bf2f(bb.x & 0xffff); b[1] = bf2f(bb.x >> 16); b[2] = bf2f(bb.y & 0xffff); b[3] = bf2f(bb.y >> 16);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = rbf(acc[i][j][r] + b[r]);  // Linear output rounds to bf16
      }
      if (EPI == EPI_BIAS_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_tanh(v[r]);
      } else if (EPI == EPI_BIAS_SILU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = silu(v[r]);
      } else if (EPI == EPI_GATE_RES || EPI == EPI_RES) {
        const uint2 xx = *reinterpret_cast<const uint2*>(g.res + (size_t)m * g.ldres + n);
        float x[4] = {bf2f(xx.x & 0xffff), bf2f(xx.x >> 16), bf2f(xx.y & 0xffff), bf2f(xx.y >> 16)};
        if (EPI == EPI_GATE_RES) {
          const uint2 ee = *reinterpret_cast<const uint2*>(g.gate + (size_t)frame * g.gate_frame_stride + n);
          float e[4] = {bf2f(ee.x & 0xffff), bf2f(ee.x >> 16), bf2f(ee.y & 0xffff), bf2f(ee.y >> 16)};
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = rbf(v[r] * e[r]);  // y * e rounds, then x + (.) rounds
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = x[r] + v[r];
      }
      uint2 o;
      o.x = pack2bf(v[0], v[1]);
      o.y = pack2bf(v[2], v[3]);
      *reinterpret_cast<uint2*>(g.C + (size_t)m * g.ldc + n) = o;
    }
  }
}

template <int EPI>
hipError_t launch(const GemmArgs& g, hipStream_t s) {
  static bool attr_set = false;
  constexpr int smem = 4 * TILE_BYTES;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<EPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  hipLaunchKernelGGL(gemm_bf16_kernel<EPI>, dim3(tiles), dim3(256), smem, s, g);
  return hipGetLastError();
}

}  // namespace

hipError_t mmpl_launch_gemm(const GemmArgs& g, hipStream_t s) {
  if (g.M <= 0 || g.N <= 0) return hipSuccess;
  if (g.K % BK != 0 || g.N % 4 != 0 || g.lda % 8 != 0 || g.ldw % 8 != 0 || g.ldc % 4 != 0) return hipErrorInvalidValue;
  switch (g.epi) {
    case EPI_BIAS: return launch<EPI_BIAS>(g, s);
    case EPI_BIAS_GELU: return launch<EPI_BIAS_GELU>(g, s);
    case EPI_BIAS_SILU: return launch<EPI_BIAS_SILU>(g, s);
    case EPI_GATE_RES: return launch<EPI_GATE_RES>(g, s);
    case EPI_RES: return launch<EPI_RES>(g, s);
    case EPI_F32_SCALE: return launch<EPI_F32_SCALE>(g, s);
  }
  return hipErrorInvalidValue;
}
