// Wan 3D causal VAE decode / encode: host orchestration + C ABI (include/mmpl_hip.h, mmpl_vae_*).
// Follows MMPL_t2v/wan/modules/vae.py (Decoder3d :369-472, Encoder3d :265-366, WanVAE_.decode/encode :517-569):
// latent frames are decoded one by one, each CausalConv3d keeps its last two input frames -- here the first two
// time slots of the conv's persistent padded channels-last volume inside the caller's workspace.
#include <math.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/mmpl_hip.h"
#include "kernels.h"
#include "mmpl_config.h"
#include "vae_kernels.h"

extern int mmpl_set_error(const char* where, const char* what);  // api.hip

namespace {

const int DIM = 96, ZD = 16;
const int DIM_MULT[4] = {1, 2, 4, 4};
const bool T_DOWN[3] = {false, true, true};

// ---- state_dict layout (same construction order as vae.py / mmpl_amd.synthetic.vae_layout)
std::vector<std::string> build_names() {
  std::vector<std::string> n;
  // frag: a 3x3(x3) stride-1 conv with 96 | N or N <= 16 -> also bound in the fragment-major packing of conv_halo_kernel
  auto conv = [&](const std::string& p, bool frag = false) {
    n.push_back(p + ".weight"); n.push_back(p + ".bias");
    if (frag) n.push_back(p + ".weight.frag");
  };
  auto res = [&](const std::string& p, int cin, int cout) {
    n.push_back(p + "residual.0.gamma"); conv(p + "residual.2", true); n.push_back(p + "residual.3.gamma"); conv(p + "residual.6", true);
    if (cin != cout) conv(p + "shortcut");
  };
  auto attn = [&](const std::string& p) { n.push_back(p + "norm.gamma"); conv(p + "to_qkv"); conv(p + "proj"); };
  int dims[5] = {DIM, DIM * DIM_MULT[0], DIM * DIM_MULT[1], DIM * DIM_MULT[2], DIM * DIM_MULT[3]};
  conv("encoder.conv1", true);
  int j = 0;
  for (int i = 0; i < 4; ++i) {
    int cin = dims[i], cout = dims[i + 1];
    for (int r = 0; r < 2; ++r) { res("encoder.downsamples." + std::to_string(j) + ".", cin, cout); cin = cout; ++j; }
    if (i != 3) {
      conv("encoder.downsamples." + std::to_string(j) + ".resample.1");
      if (T_DOWN[i]) conv("encoder.downsamples." + std::to_string(j) + ".time_conv");
      ++j;
    }
  }
  res("encoder.middle.0.", dims[4], dims[4]); attn("encoder.middle.1."); res("encoder.middle.2.", dims[4], dims[4]);
  n.push_back("encoder.head.0.gamma"); conv("encoder.head.2");
  conv("conv1"); conv("conv2");
  int dd[5] = {DIM * DIM_MULT[3], DIM * DIM_MULT[3], DIM * DIM_MULT[2], DIM * DIM_MULT[1], DIM * DIM_MULT[0]};
  const bool t_up[3] = {T_DOWN[2], T_DOWN[1], T_DOWN[0]};
  conv("decoder.conv1", true);
  res("decoder.middle.0.", dd[0], dd[0]); attn("decoder.middle.1."); res("decoder.middle.2.", dd[0], dd[0]);
  j = 0;
  for (int i = 0; i < 4; ++i) {
    int cin = dd[i], cout = dd[i + 1];
    if (i >= 1) cin /= 2;
    for (int r = 0; r < 3; ++r) { res("decoder.upsamples." + std::to_string(j) + ".", cin, cout); cin = cout; ++j; }
    if (i != 3) {
      conv("decoder.upsamples." + std::to_string(j) + ".resample.1", true);
      if (t_up[i]) conv("decoder.upsamples." + std::to_string(j) + ".time_conv");
      ++j;
    }
  }
  n.push_back("decoder.head.0.gamma"); conv("decoder.head.2", true);
  return n;
}

struct Arena {
  char* base = nullptr;
  size_t off = 0;
  std::map<std::string, std::pair<size_t, size_t>> named;
  bf16_t* get(const std::string& name, size_t elems, size_t elem_bytes = 2) {
    auto it = named.find(name);
    const size_t bytes = (elems * elem_bytes + 255) & ~(size_t)255;
    if (it == named.end()) {
      named[name] = {off, bytes};
      it = named.find(name);
      off += bytes;
    }
    return (bf16_t*)(base + it->second.first);
  }
};

}  // namespace

struct MmplVae {
  int lat_h, lat_w;
  std::vector<std::string> names;
  std::map<std::string, int> idx;
  std::vector<const bf16_t*> w;
  const bf16_t* W(const std::string& n) const { return w.empty() ? nullptr : w[idx.at(n)]; }
  const bf16_t* Wopt(const std::string& n) const { auto it = idx.find(n); return (w.empty() || it == idx.end()) ? nullptr : w[it->second]; }
};

namespace {

struct Ctx {
  MmplVae* v;
  Arena ar;
  hipStream_t s;
  bool dry;
  hipError_t err = hipSuccess;
  const char* where = "";
  std::map<std::string, int> ring_base;   // temporal-cache rings: first slot of the two cached frames, per conv volume
  std::map<std::string, bool> prefilled;  // convs whose new input frames the PRODUCING conv's epilogue already normalised into their ring slots
  size_t plain_elems;
  bf16_t* plain(int i) { return ar.get("plain" + std::to_string(i), plain_elems); }
  void chk(hipError_t e, const char* w) { if (e != hipSuccess && err == hipSuccess) { err = e; where = w; } }
};

void copy_slots(Ctx& c, bf16_t* vol, size_t slot_elems, int from, int to, int n) {
  if (c.dry) return;
  for (int i = 0; i < n; ++i)
    c.chk(hipMemcpyAsync(vol + (size_t)(to + i) * slot_elems, vol + (size_t)(from + i) * slot_elems, slot_elems * 2,
                         hipMemcpyDeviceToDevice, c.s), "cache shift");
}

void norm_into(Ctx& c, const bf16_t* x, int T, int H, int W, int C, const bf16_t* gamma, bool silu, bf16_t* dst, int Hd, int Wd, int ldd,
               int dt0, int dy0, int dx0) {
  if (c.dry) return;
  NormArgs a{x, (long)T * H * W, C, H, W, gamma, sqrtf((float)C), silu ? 1 : 0, dst, Hd, Wd, ldd, dt0, dy0, dx0};
  c.chk(vae_launch_norm(a, c.s), "norm");
}

// The consumer of a conv's output, when that consumer is a cached 3x3x3 conv behind RMS_norm + SiLU whose padded ring slots the
// producer's epilogue can fill directly (ConvArgs.ngamma): `name` / `gamma` / output channels `N` of the consumer; everything else (T, Tmax, H, W, channels)
// is the producer's output geometry.
struct Fuse { std::string name; const bf16_t* gamma; int N; };
struct NormDst { const bf16_t* gamma = nullptr; bf16_t* frame[8] = {}; };

// Resolve a Fuse request of a producer with output [T, H, W, C]: the consumer's ring slots for its T new frames, or nothing when
// either side cannot do it (the fused epilogue exists for the 96-channel halo kernel only; the consumer must read ring slots).
// Allocates the consumer's volume under the name cached_conv3 will ask for, in the dry pass too.
NormDst resolve_fuse(Ctx& c, const Fuse* f, bool producer_halo, int T, int Tmax, int H, int W, int C) {
  NormDst nd;
  if (!f || mmpl_config().vae_no_fuse_norm) return nd;
  const size_t slot = (size_t)(H + 2) * (W + 2) * C;
  const int Tp = Tmax + 2;
  bf16_t* vol = c.ar.get(f->name + ".pad", (size_t)Tp * slot);
  if (c.dry || !producer_halo || C != 96 || T > 8) return nd;
  ConvArgs probe = {};
  probe.Wfrag = c.v->Wopt(f->name + ".weight.frag"); probe.kt = probe.kh = probe.kw = 3; probe.st = probe.sy = probe.sx = 1;
  probe.Hp = H + 2; probe.Wp = W + 2; probe.Ho = H; probe.Wo = W; probe.M = T * H * W; probe.Cin = C;
  probe.N = f->N;
  if (!vae_conv_uses_halo(probe)) return nd;
  const int base = c.ring_base[f->name];
  for (int t = 0; t < T; ++t) nd.frame[t] = vol + (size_t)((base + t + 2) % Tp) * slot;
  nd.gamma = f->gamma;
  c.prefilled[f->name] = true;
  return nd;
}

void conv(Ctx& c, const bf16_t* src, int Cin, int Hp, int Wp, int st, int sy, int sx, int kt, int kh, int kw, const bf16_t* Wt,
          const bf16_t* bias, int To, int Ho, int Wo, int N, bf16_t* dst, int Hd, int Wd, int ldd, int dt0, int dy0, int dx0,
          const bf16_t* res, int ldres, const bf16_t* Wfrag = nullptr, const bf16_t* const* frames = nullptr, int n_frames = 0,
          const NormDst* nd = nullptr) {
  if (c.dry) return;
  ConvArgs g = {};
  if (nd && nd->gamma) {
    g.ngamma = nd->gamma; g.nscale = sqrtf((float)N);
    for (int j = 0; j < To && j < 8; ++j) g.nframe[j] = nd->frame[j];
  }
  g.Wfrag = Wfrag;
  for (int j = 0; j < n_frames && j < 8; ++j) g.frame[j] = frames[j];
  g.src = src; g.Cin = Cin; g.Hp = Hp; g.Wp = Wp; g.st = st; g.sy = sy; g.sx = sx; g.ntaps = kt * kh * kw;
  g.kt = kt; g.kh = kh; g.kw = kw;
  int k = 0;
  for (int a = 0; a < kt; ++a)
    for (int b = 0; b < kh; ++b)
      for (int d = 0; d < kw; ++d) g.tap_off[k++] = (a * Hp + b) * Wp + d;
  g.W = Wt; g.bias = bias; g.M = To * Ho * Wo; g.N = N; g.Ho = Ho; g.Wo = Wo;
  g.dst = dst; g.Hd = Hd; g.Wd = Wd; g.ldd = ldd; g.dt0 = dt0; g.dy0 = dy0; g.dx0 = dx0; g.dc0 = 0; g.res = res; g.ldres = ldres;
  c.chk(vae_launch_conv(g, c.s), "conv");
}

void gemm(Ctx& c, const bf16_t* A, int lda, const bf16_t* Wt, int ldw, const bf16_t* bias, void* C, int ldc, int M, int N, int K, int epi,
          const bf16_t* res, int ldres, float alpha = 1.f) {
  if (c.dry) return;
  GemmArgs g{A, lda, Wt, ldw, bias, (bf16_t*)C, ldc, M, N, K, epi, res, ldres, nullptr, 0, 1, alpha, 0, 0, 0, 0, 0};
  c.chk(mmpl_launch_gemm(g, c.s), "gemm");
}

// CausalConv3d 3x3x3 with temporal cache; input = norm+SiLU(x) (gamma != null) or x as is.
// The cache (vae.py:14, 207-216: the conv's last two input frames) is the two frame slots in front of the T new ones.  With the
// halo-tile kernel the Tmax + 2 slots of the volume form a RING -- logical frame j lives in slot (base + j) mod (Tmax + 2), the norm
// pass writes the new frames straight into their slots, the kernel is handed the slot pointers, and base advances by T: no copy.
// (The plain kernel addresses frames linearly, so there the last two frames are copied to the front after every call: 4 % of a decode.)
void cached_conv3(Ctx& c, const std::string& name, const bf16_t* x, const bf16_t* gamma, int T, int Tmax, int H, int W, int Cin,
                  int N, bf16_t* out, const bf16_t* res, const Fuse* fuse = nullptr, bool plain_needed = true) {
  const size_t slot = (size_t)(H + 2) * (W + 2) * Cin;
  const int Tp = Tmax + 2;
  bf16_t* vol = c.ar.get(name + ".pad", (size_t)Tp * slot);
  const bf16_t* Wfrag = c.v->Wopt(name + ".weight.frag");
  bool ring = false;
  if (x && !c.dry) {
    ConvArgs probe = {};
    probe.Wfrag = Wfrag; probe.kt = probe.kh = probe.kw = 3; probe.st = probe.sy = probe.sx = 1; probe.Hp = H + 2; probe.Wp = W + 2;
    probe.Ho = H; probe.Wo = W; probe.M = T * H * W; probe.N = N; probe.Cin = Cin;
    ring = vae_conv_uses_halo(probe);
  }
  const NormDst nd = resolve_fuse(c, fuse, ring, T, Tmax, H, W, N);
  if (!ring) {
    if (x) norm_into(c, x, T, H, W, Cin, gamma, gamma != nullptr, vol, H + 2, W + 2, Cin, 2, 1, 1);
    conv(c, vol, Cin, H + 2, W + 2, 1, 1, 1, 3, 3, 3, c.v->W(name + ".weight"), c.v->W(name + ".bias"), T, H, W, N, out, H, W, N, 0, 0, 0,
         res, N, Wfrag);
    copy_slots(c, vol, slot, T, 0, 2);
    return;
  }
  int& base = c.ring_base[name];
  const bf16_t* frames[8];
  for (int j = 0; j < T + 2; ++j) frames[j] = vol + (size_t)((base + j) % Tp) * slot;
  auto pre = c.prefilled.find(name);
  if (pre != c.prefilled.end() && pre->second) {
    pre->second = false;               // the producer's epilogue wrote frames[2 .. T + 1]
  } else {
    for (int t = 0; t < T; ++t)
      norm_into(c, x + (size_t)t * H * W * Cin, 1, H, W, Cin, gamma, gamma != nullptr, const_cast<bf16_t*>(frames[t + 2]), H + 2, W + 2, Cin, 0, 1, 1);
  }
  // (a fused consumer that nothing else reads -- a ResidualBlock's first conv -- needs no plain output at all)
  conv(c, vol, Cin, H + 2, W + 2, 1, 1, 1, 3, 3, 3, c.v->W(name + ".weight"), c.v->W(name + ".bias"), T, H, W, N,
       (nd.gamma && !plain_needed) ? nullptr : out, H, W, N, 0, 0, 0, res, N, Wfrag, frames, T + 2, &nd);
  base = (base + T) % Tp;
}

// ResidualBlock (vae.py:186-220): x [T,H,W,cin] -> out [T,H,W,cout]; x, out, tmp are distinct plain buffers
void res_block(Ctx& c, const std::string& pre, const bf16_t* x, bf16_t* out, bf16_t* tmp, bf16_t* tmp2, int T, int Tmax, int H, int W,
               int cin, int cout, const Fuse* next = nullptr) {
  const bf16_t* h = x;
  if (cin != cout) {  // CausalConv3d(in, out, 1): a 1-tap implicit GEMM straight from x
    conv(c, x, cin, H, W, 1, 1, 1, 1, 1, 1, c.v->W(pre + "shortcut.weight"), c.v->W(pre + "shortcut.bias"), T, H, W, cout, tmp2, H, W, cout,
         0, 0, 0, nullptr, 0);
    h = tmp2;
  }
  const Fuse inner{pre + "residual.6", c.v->W(pre + "residual.3.gamma"), cout};
  cached_conv3(c, pre + "residual.2", x, c.v->W(pre + "residual.0.gamma"), T, Tmax, H, W, cin, cout, tmp, nullptr, &inner, false);
  cached_conv3(c, pre + "residual.6", tmp, c.v->W(pre + "residual.3.gamma"), T, Tmax, H, W, cout, cout, out, h, next, true);
}

// AttentionBlock (vae.py:223-262), per frame, single head of dim C over H*W tokens
void attn_block(Ctx& c, const std::string& pre, const bf16_t* x, bf16_t* out, int T, int H, int W, int C) {
  const int HW = H * W, HWp = (HW + 63) / 64 * 64;
  bf16_t* xn = c.ar.get(pre + "xn", (size_t)HW * C);
  bf16_t* qkv = c.ar.get(pre + "qkv", (size_t)HW * 3 * C);
  float* sc = (float*)c.ar.get(pre + "scores", (size_t)HW * HWp, 4);
  bf16_t* p = c.ar.get(pre + "p", (size_t)HW * HWp);
  bf16_t* vt = c.ar.get(pre + "vt", (size_t)C * HWp);
  bf16_t* o = c.ar.get(pre + "o", (size_t)HW * C);
  for (int t = 0; t < T; ++t) {
    const bf16_t* xt = x + (size_t)t * HW * C;
    norm_into(c, xt, 1, H, W, C, c.v->W(pre + "norm.gamma"), false, xn, H, W, C, 0, 0, 0);
    gemm(c, xn, C, c.v->W(pre + "to_qkv.weight"), C, c.v->W(pre + "to_qkv.bias"), qkv, 3 * C, HW, 3 * C, C, EPI_BIAS, nullptr, 0);
    gemm(c, qkv, 3 * C, qkv + C, 3 * C, nullptr, sc, HWp, HW, HW, C, EPI_F32_SCALE, nullptr, 0, 1.0f / sqrtf((float)C));
    if (!c.dry) {
      c.chk(vae_launch_softmax(sc, HWp, p, HWp, HW, HW, c.s), "softmax");
      c.chk(vae_launch_transpose(qkv + 2 * C, 3 * C, vt, HWp, HW, C, c.s), "transpose");
    }
    gemm(c, p, HWp, vt, HWp, nullptr, o, C, HW, C, HWp, EPI_BIAS, nullptr, 0);
    gemm(c, o, C, c.v->W(pre + "proj.weight"), C, c.v->W(pre + "proj.bias"), out + (size_t)t * HW * C, C, HW, C, C, EPI_RES, xt, C);
  }
}

// ------------------------------------------------------------------------------------------------ decoder
// one latent frame (vae.py:423-472).  Returns the number of pixel frames produced (1 for the first latent, else 4).
int decoder_frame(Ctx& c, const bf16_t* z_all, int F, int fi, const float* mean, const float* inv_std, float* out, int t_out) {
  MmplVae* v = c.v;
  const bool first = fi == 0;
  int H = v->lat_h, W = v->lat_w, T = 1;
  bf16_t *a = c.plain(0), *b = c.plain(1), *t1 = c.plain(2), *t2 = c.plain(3);
  {  // z prep + conv2 -> decoder.conv1's padded volume, then conv1
    const size_t slot = (size_t)(H + 2) * (W + 2) * 32;
    bf16_t* vol = c.ar.get("decoder.conv1.pad", 3 * slot);
    if (!c.dry) {
      ZPrepArgs zp = {};
      zp.z = z_all + (size_t)fi * 16 * H * W; zp.F = 1; zp.h = H; zp.w = W;
      for (int i = 0; i < 16; ++i) { zp.mean[i] = mean[i]; zp.inv_std[i] = inv_std[i]; }
      zp.w2 = v->W("conv2.weight"); zp.b2 = v->W("conv2.bias"); zp.dst = vol; zp.dt0 = 2;
      c.chk(vae_launch_zprep(zp, c.s), "zprep");
    }
    cached_conv3(c, "decoder.conv1", nullptr, nullptr, 1, 1, H, W, 32, 384, a, nullptr);
  }
  res_block(c, "decoder.middle.0.", a, b, t1, t2, 1, 1, H, W, 384, 384);
  attn_block(c, "decoder.middle.1.", b, a, 1, H, W, 384);
  res_block(c, "decoder.middle.2.", a, b, t1, t2, 1, 1, H, W, 384, 384);
  bf16_t *x = b, *y = a;
  const int dd[5] = {384, 384, 384, 192, 96};
  const bool t_up[3] = {true, true, false};
  int j = 0, tmax = 1;
  for (int i = 0; i < 4; ++i) {
    int cin = dd[i], cout = dd[i + 1];
    if (i >= 1) cin /= 2;
    for (int r = 0; r < 3; ++r) {
      // who reads this block's output behind a norm: the next block's first conv (same width only -- a widening block's 1x1
      // shortcut is not a ring conv), after the last block of the last stage the head; an upsampler reads it raw
      Fuse next;
      const Fuse* nextp = nullptr;
      if (r < 2) {
        const std::string np = "decoder.upsamples." + std::to_string(j + 1) + ".";
        next = Fuse{np + "residual.2", v->W(np + "residual.0.gamma"), cout}; nextp = &next;
      } else if (i == 3) {
        next = Fuse{"decoder.head.2", v->W("decoder.head.0.gamma"), 4}; nextp = &next;
      }
      res_block(c, "decoder.upsamples." + std::to_string(j) + ".", x, y, t1, t2, T, tmax, H, W, cin, cout, nextp);
      std::swap(x, y);
      cin = cout;
      ++j;
    }
    if (i != 3) {
      const std::string pre = "decoder.upsamples." + std::to_string(j) + ".";
      const int C = cout;
      const bf16_t* up_src = x;
      int lds = C, To = T, inter = 0;
      if (t_up[i]) {
        if (!first) {  // time_conv (3,1,1): C -> 2C, then frames interleave (vae.py:103-137); first latent frame: 'Rep'
          const size_t slot = (size_t)H * W * C;
          bf16_t* vol = c.ar.get(pre + "time_conv.pad", (size_t)(tmax + 2) * slot);
          norm_into(c, x, T, H, W, C, nullptr, false, vol, H, W, C, 2, 0, 0);
          conv(c, vol, C, H, W, 1, 1, 1, 3, 1, 1, v->W(pre + "time_conv.weight"), v->W(pre + "time_conv.bias"), T, H, W, 2 * C, t1, H, W,
               2 * C, 0, 0, 0, nullptr, 0);
          copy_slots(c, vol, slot, T, 0, 2);
          up_src = t1; lds = 2 * C; To = 2 * T; inter = 1;
        }
        tmax *= 2;
      }
      bf16_t* pu = c.ar.get(pre + "up.pad", (size_t)tmax * (2 * H + 2) * (2 * W + 2) * C);
      if (!c.dry) {
        UpArgs u{up_src, lds, C, H, W, To, inter, pu, 2 * H + 2, 2 * W + 2};
        c.chk(vae_launch_upsample(u, c.s), "upsample");
      }
      T = To; H *= 2; W *= 2;
      {  // the upsampler's Conv2d feeds the next stage's first block: its norm goes into this conv's epilogue where it can (C / 2 == 96)
        const std::string np = "decoder.upsamples." + std::to_string(j + 1) + ".";
        const Fuse next{np + "residual.2", v->W(np + "residual.0.gamma"), dd[i + 2]};
        ConvArgs probe = {};
        probe.Wfrag = v->Wopt(pre + "resample.1.weight.frag"); probe.kt = 1; probe.kh = probe.kw = 3; probe.st = probe.sy = probe.sx = 1;
        probe.Hp = H + 2; probe.Wp = W + 2; probe.Ho = H; probe.Wo = W; probe.M = T * H * W; probe.Cin = C; probe.N = C / 2;
        const NormDst nd = resolve_fuse(c, &next, !c.dry && vae_conv_uses_halo(probe), T, tmax, H, W, C / 2);
        conv(c, pu, C, H + 2, W + 2, 1, 1, 1, 1, 3, 3, v->W(pre + "resample.1.weight"), v->W(pre + "resample.1.bias"), T, H, W, C / 2, y, H, W,
             C / 2, 0, 0, 0, nullptr, 0, v->Wopt(pre + "resample.1.weight.frag"), nullptr, 0, &nd);
      }
      std::swap(x, y);
      ++j;
    }
  }
  // head: RMS_norm + SiLU + conv(96 -> 3 padded to 4)
  cached_conv3(c, "decoder.head.2", x, v->W("decoder.head.0.gamma"), T, tmax, H, W, 96, 4, y, nullptr);
  if (!c.dry) c.chk(vae_launch_px_out(y, out, T, H, W, t_out, c.s), "px_out");
  return T;
}

// ------------------------------------------------------------------------------------------------ encoder
// one pixel chunk (1 frame for the first call, 4 afterwards) -> one latent frame (vae.py:318-366, 517-543)
void encoder_chunk(Ctx& c, const bf16_t* px, int Ttot, int t0, int T, bool first, const float* mean, const float* inv_std, float* out,
                   int f_out) {
  MmplVae* v = c.v;
  int H = v->lat_h * 8, W = v->lat_w * 8;
  bf16_t *x = c.plain(0), *y = c.plain(1), *t1 = c.plain(2), *t2 = c.plain(3);
  int tmax = 4;
  {
    const size_t slot = (size_t)(H + 2) * (W + 2) * 32;
    bf16_t* vol = c.ar.get("encoder.conv1.pad", (size_t)(tmax + 2) * slot);
    if (!c.dry) c.chk(vae_launch_px_in(px, vol, Ttot, t0, T, H, W, 2, c.s), "px_in");
    cached_conv3(c, "encoder.conv1", nullptr, nullptr, T, tmax, H, W, 32, 96, x, nullptr);
  }
  const int dims[5] = {96, 96, 192, 384, 384};
  int j = 0;
  for (int i = 0; i < 4; ++i) {
    int cin = dims[i], cout = dims[i + 1];
    for (int r = 0; r < 2; ++r) {
      res_block(c, "encoder.downsamples." + std::to_string(j) + ".", x, y, t1, t2, T, tmax, H, W, cin, cout);
      std::swap(x, y);
      cin = cout;
      ++j;
    }
    if (i != 3) {
      const std::string pre = "encoder.downsamples." + std::to_string(j) + ".";
      const int C = cout;
      // ZeroPad2d((0,1,0,1)) + Conv2d(3, stride 2)  (vae.py:87-90)
      bf16_t* pd = c.ar.get(pre + "down.pad", (size_t)tmax * (H + 1) * (W + 1) * C);
      norm_into(c, x, T, H, W, C, nullptr, false, pd, H + 1, W + 1, C, 0, 0, 0);
      conv(c, pd, C, H + 1, W + 1, 1, 2, 2, 1, 3, 3, v->W(pre + "resample.1.weight"), v->W(pre + "resample.1.bias"), T, H / 2, W / 2, C, y,
           H / 2, W / 2, C, 0, 0, 0, nullptr, 0);
      H /= 2; W /= 2;
      std::swap(x, y);
      if (T_DOWN[i]) {  // time_conv (3,1,1) stride (2,1,1) over [cache(1) | x]  (vae.py:143-159)
        const size_t slot = (size_t)H * W * C;
        bf16_t* vol = c.ar.get(pre + "time_conv.pad", (size_t)(tmax + 1) * slot);
        if (first) {
          norm_into(c, x, 1, H, W, C, nullptr, false, vol, H, W, C, 0, 0, 0);   // feat_cache = x.clone(), no conv
        } else {
          norm_into(c, x, T, H, W, C, nullptr, false, vol, H, W, C, 1, 0, 0);
          const int To = (T + 1 - 3) / 2 + 1;
          conv(c, vol, C, H, W, 2, 1, 1, 3, 1, 1, v->W(pre + "time_conv.weight"), v->W(pre + "time_conv.bias"), To, H, W, C, y, H, W, C, 0, 0,
               0, nullptr, 0);
          copy_slots(c, vol, slot, T, 0, 1);
          T = To;
          std::swap(x, y);
        }
        tmax /= 2;
      }
      ++j;
    }
  }
  res_block(c, "encoder.middle.0.", x, y, t1, t2, T, tmax, H, W, 384, 384);
  attn_block(c, "encoder.middle.1.", y, x, T, H, W, 384);
  res_block(c, "encoder.middle.2.", x, y, t1, t2, T, tmax, H, W, 384, 384);
  cached_conv3(c, "encoder.head.2", y, v->W("encoder.head.0.gamma"), T, tmax, H, W, 384, 32, x, nullptr);
  if (!c.dry) {
    MuArgs m = {};
    m.enc = x; m.w1 = v->W("conv1.weight"); m.b1 = v->W("conv1.bias"); m.out = out; m.F = T; m.f_out = f_out; m.h = H; m.w = W;
    for (int i = 0; i < 16; ++i) { m.mean[i] = mean[i]; m.inv_std[i] = inv_std[i]; }
    c.chk(vae_launch_mu_out(m, c.s), "mu_out");
  }
}

size_t plain_elems_for(const MmplVae* v) { return (size_t)4 * (v->lat_h * 8) * (v->lat_w * 8) * 96; }

}  // namespace

extern "C" {

int mmpl_vae_num_weights(void) { return (int)build_names().size(); }

const char* mmpl_vae_weight_name(int i) {
  static std::vector<std::string> names = build_names();
  return (i >= 0 && i < (int)names.size()) ? names[i].c_str() : nullptr;
}

int mmpl_vae_create(int lat_h, int lat_w, MmplVae** out) {
  if (!out || lat_h < 2 || lat_w < 2) return mmpl_set_error("mmpl_vae_create", "bad arguments");
  MmplVae* v = new MmplVae();
  v->lat_h = lat_h;
  v->lat_w = lat_w;
  v->names = build_names();
  for (size_t i = 0; i < v->names.size(); ++i) v->idx[v->names[i]] = (int)i;
  *out = v;
  return 0;
}

void mmpl_vae_destroy(MmplVae* v) { delete v; }

int mmpl_vae_bind_weights(MmplVae* v, const void* const* ptrs, int n) {
  if (!v || !ptrs || n != (int)v->names.size()) return mmpl_set_error("mmpl_vae_bind_weights", "wrong pointer count");
  v->w.resize(n);
  for (int i = 0; i < n; ++i) {
    if (!ptrs[i]) return mmpl_set_error("mmpl_vae_bind_weights", "null weight pointer");
    v->w[i] = (const bf16_t*)ptrs[i];
  }
  return 0;
}

size_t mmpl_vae_workspace_bytes(MmplVae* v, int mode) {
  Ctx c{v, Arena(), nullptr, true};
  c.plain_elems = plain_elems_for(v);
  float dummy[16] = {0};
  if (mode == 0) {
    decoder_frame(c, nullptr, 2, 0, dummy, dummy, nullptr, 0);
    decoder_frame(c, nullptr, 2, 1, dummy, dummy, nullptr, 1);
  } else {
    encoder_chunk(c, nullptr, 5, 0, 1, true, dummy, dummy, nullptr, 0);
    encoder_chunk(c, nullptr, 5, 1, 4, false, dummy, dummy, nullptr, 1);
  }
  return c.ar.off;
}

int mmpl_vae_decode(MmplVae* v, const void* z, int n_frames, const float* mean, const float* inv_std, void* out, void* ws,
                    size_t ws_bytes, mmpl_stream_t stream) {
  if (!v || v->w.empty()) return mmpl_set_error("mmpl_vae_decode", "weights not bound");
  if (n_frames < 1) return mmpl_set_error("mmpl_vae_decode", "n_frames < 1");
  const size_t need = mmpl_vae_workspace_bytes(v, 0);
  if (ws_bytes < need) return mmpl_set_error("mmpl_vae_decode", "workspace too small");
  Ctx c{v, Arena(), (hipStream_t)stream, false};
  c.ar.base = (char*)ws;
  c.plain_elems = plain_elems_for(v);
  if (hipMemsetAsync(ws, 0, need, c.s) != hipSuccess) return mmpl_set_error("mmpl_vae_decode", "memset failed");  // clear_cache()
  int t_out = 0;
  for (int i = 0; i < n_frames; ++i) t_out += decoder_frame(c, (const bf16_t*)z, n_frames, i, mean, inv_std, (float*)out, t_out);
  if (c.err != hipSuccess) return mmpl_set_error(c.where, hipGetErrorString(c.err));
  return 0;
}

int mmpl_vae_encode(MmplVae* v, const void* px, int n_px_frames, const float* mean, const float* inv_std, void* out, void* ws,
                    size_t ws_bytes, mmpl_stream_t stream) {
  if (!v || v->w.empty()) return mmpl_set_error("mmpl_vae_encode", "weights not bound");
  if (n_px_frames < 1 || (n_px_frames - 1) % 4) return mmpl_set_error("mmpl_vae_encode", "pixel frames must be 1 + 4k");
  const size_t need = mmpl_vae_workspace_bytes(v, 1);
  if (ws_bytes < need) return mmpl_set_error("mmpl_vae_encode", "workspace too small");
  Ctx c{v, Arena(), (hipStream_t)stream, false};
  c.ar.base = (char*)ws;
  c.plain_elems = plain_elems_for(v);
  if (hipMemsetAsync(ws, 0, need, c.s) != hipSuccess) return mmpl_set_error("mmpl_vae_encode", "memset failed");
  const int iters = 1 + (n_px_frames - 1) / 4;
  for (int i = 0; i < iters; ++i)
    encoder_chunk(c, (const bf16_t*)px, n_px_frames, i == 0 ? 0 : 1 + 4 * (i - 1), i == 0 ? 1 : 4, i == 0, mean, inv_std, (float*)out, i);
  if (c.err != hipSuccess) return mmpl_set_error(c.where, hipGetErrorString(c.err));
  return 0;
}

}  // extern "C"
