/* libmmpl_hip.so -- C ABI of the MI355X-native MMPL denoising hot path.
 *
 * The reference (Tele-AI/MMPL) has no FFI: its seams are Python signatures.  Each entry point
 * below names the reference interface it replaces (paths relative to the reference root); the Python mirror
 * in mmpl_amd/ binds them with ctypes (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions: every pointer marked "dev" is a borrowed device pointer (e.g. torch.Tensor.data_ptr()); nothing
 * is retained after the call returns except by mmpl_dit_bind_weights (which stores the weight pointers) and
 * nothing is allocated on the device except two 256 KiB RoPE tables owned by the handle (and, under MMPL_CHECK_SHARE=1, 64 bytes of
 * check state).  All tensors are
 * bfloat16 unless stated.  Every call takes the HIP stream to enqueue on and is asynchronous with respect to
 * the host.  Return value: 0 = ok, non-zero = error (text via mmpl_last_error(), thread-local).
 */
#ifndef MMPL_HIP_H
#define MMPL_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* mmpl_stream_t; /* hipStream_t */

typedef struct MmplDitConfig {
  int dim, ffn_dim, num_heads, num_layers; /* wan/configs/wan_t2v_14B.py:17-25 */
  int text_dim, freq_dim, in_dim, out_dim, text_len;
  float eps;
  int lat_h, lat_w;   /* latent frame size: 60x104 (480p, the reference's literal 1560 tokens) or 90x160 (720p) */
  int max_frames;     /* largest stage (7) */
} MmplDitConfig;

typedef struct MmplDit MmplDit;

/* Number of weight pointers mmpl_dit_bind_weights expects, and the name of slot i (reference state_dict key,
 * "blocks.N." prefix for per-layer slots; packed slots say how they are packed). */
int mmpl_dit_num_weights(const MmplDitConfig* cfg);
const char* mmpl_dit_weight_name(int slot_in_layer_or_global, int per_layer);

/* CausalFPSWanModel.__init__ (wan/modules/causal_fps_model.py:409-530): geometry + RoPE tables. */
int mmpl_dit_create(const MmplDitConfig* cfg, MmplDit** out);
void mmpl_dit_destroy(MmplDit* h);
/* load_state_dict seam (Wan_fps_inference_1gpu.py:66-68): stores n borrowed dev pointers, order = weight slots. */
int mmpl_dit_bind_weights(MmplDit* h, const void* const* dev_ptrs, int n);

size_t mmpl_dit_workspace_bytes(const MmplDit* h, int n_frames);
size_t mmpl_dit_context_workspace_bytes(const MmplDit* h);

/* text_embedding + per-layer cross-attn K/V (causal_fps_model.py:780-786, model.py:175-180), once per prompt.
 * context: dev [text_len, text_dim] zero-padded; cross_k/cross_v: dev out [num_layers, text_len, dim].
 * distinct_rows (host out, may be NULL): n such that rows n .. text_len-1 of the EMBEDDED context are bitwise identical (the
 * reference zero-pads the T5 output and attends over the padding unmasked, utils/wan_wrapper.py:46-47, model.py:189, so after
 * the text embedding the padded tail is one repeated row, and so are its K and V rows in every block); text_len when the tail
 * does not repeat.  Determined on the device with one 2 KiB read-back (synchronises `stream`); inside a stream capture nothing
 * is read back and text_len is reported.  It is a property of the K/V CONTENTS: it stays valid for any copy of cross_k/cross_v
 * and is what mmpl_dit_forward takes as `cross_rows`. */
int mmpl_dit_precompute_context(MmplDit* h, const void* context, void* cross_k, void* cross_v, void* workspace,
                                size_t workspace_bytes, int* distinct_rows, mmpl_stream_t stream);

/* Wan-I2V model type (WanModel(model_type='i2v'), wan/modules/model.py:563-616,672-712): in_dim = 36 (x and the
 * conditioning video y concatenated on the channel axis, model.py:680-681 -- the caller concatenates) and every block's
 * cross-attention also attends to the 257 projected CLIP tokens (WanI2VCrossAttention, model.py:224-266).  The image
 * K / V depend only on the image: img_k[l] = norm_k_img(k_img(img_emb(clip_fea))), img_v[l] = v_img(...), built with
 * mmpl_i2v_img_proj + mmpl_i2v_img_kv per layer; dev [num_layers, n_img_tokens, dim], borrowed until replaced.
 * NULL, NULL restores the text-only cross-attention. */
int mmpl_dit_set_image_kv(MmplDit* h, const void* img_k, const void* img_v, int n_img_tokens);

/* Optional diagnostics of the self-attention kernel's data dependence.  attn_w64_kernel runs a max-free FAST softmax pass per
 * 256-row query block and redoes the block with the GENERAL (running-reference) pass if any row sum left [2^-100, 2^100].
 * stats_dev: 5 x uint64 in device memory (borrowed; zero them yourself), incremented by every self-attention launch of
 * mmpl_dit_forward on this handle, inside hipGraph replays too: [0] += blocks run, [1] += blocks whose FAST pass failed and that were
 * redone (both passes paid), [2] += waves (64 of a block's 256 query rows) that held a failing row themselves -- the unit a
 * finer-grained redo would pay for, [3] += blocks their history sent straight to the GENERAL pass, [4] += blocks whose FAST pass held
 * on the references their history remembered (`attn_history`).  NULL switches it off. */
int mmpl_dit_set_attn_stats(MmplDit* h, void* stats_dev);

/* CausalFPSWanModel._forward_inference (causal_fps_model.py:708-837) behind WanFPSWrapper.forward
 * (utils/wan_wrapper.py:422-493).
 *   x_in / out : dev [n_frames, in_dim, lat_h, lat_w] / [n_frames, 16, lat_h, lat_w]  (the pipeline's [B=1, F, C, H, W] layout)
 *   t_dev      : dev float32 [n_frames]
 *   frame_ids  : host, RoPE temporal index per frame (= current_start / frame_seqlen)
 *   write_slots: host, KV slot each frame's K/V is written to before attending; all -1 = do not persist
 *                (the reference's [13..18] stage, causal_fps_model.py:254-264)
 *   visible_slots: host, cache slots attended (attention_vis_index after the 19,20 -> 13,14 remap)
 *   k_cache / v_cache: dev [num_layers, n_slots * S, dim], mutated in place like the reference's kv_cache
 *   cross_k / cross_v: from mmpl_dit_precompute_context (or any copy of them)
 *   cross_rows : the caller's statement that rows cross_rows .. text_len-1 of every layer of cross_k (and of cross_v) are copies
 *                of row cross_rows -- the `distinct_rows` mmpl_dit_precompute_context reported for these contents.  The text
 *                cross-attention then attends over rows 0 .. cross_rows with the last one weighted text_len - cross_rows times:
 *                the same softmax, (text_len - cross_rows - 1) fewer keys.  text_len (or any value outside [0, text_len - 2])
 *                = attend over all text_len rows, which is always correct.
 *   share_out / share_in (both may be NULL, at most one non-NULL): dev [n_frames * S, dim] bf16.  Block 0's self-attention sees
 *                nothing that differs between the two branches of classifier-free guidance (same latents, same timestep, and -- by
 *                induction -- the same layer-0 K / V in both caches), only the text context differs and that enters after it.
 *                share_out: this forward also leaves x as it is after block 0's self-attention residual there.  share_in: the
 *                caller's statement that ANOTHER forward on the same x_in / t / frame_ids / write_slots / visible_slots, whose
 *                layer-0 cache contents equal this one's, produced it: block 0's attention and output projection are skipped and
 *                x continues from share_in (this forward's own layer-0 K / V slots are still written).  Same kernels on the same
 *                inputs: the result is bit-identical to computing it.  MMPL_CHECK_SHARE=1 (environment, read when the library is
 *                first used; a debug switch) turns the statement into a check: the share_out and the share_in forward each
 *                fingerprint the layer-0 K / V slots they attend to on the device (one pass over them), the share_in forward
 *                compares; outside a stream capture it then fails with an error on a mismatch, inside one the mismatch is counted
 *                on the device (mmpl_dit_share_check_failures).
 *   attn_history (may be NULL): dev, mmpl_dit_attn_history_bytes(h, n_frames) bytes owned by the caller: per (layer, head, 256-row query
 *                block, split part) of the self-attention one state byte and 128 int16 lane references -- what this attention's
 *                previous launch learned.  Hand the same buffer to every forward of one (CFG branch, stage) -- consecutive denoise
 *                steps see the same K / V and nearly the same q -- and zero it when the stage changes.  A query block whose max-free
 *                FAST softmax pass failed (heavy-tailed scores) then stops paying for both passes: from its first failure on every pass
 *                leaves each lane's mean log-sum-exp, the next FAST pass takes that as its reference and holds wherever the scores'
 *                range is; a block that fails even so goes straight to the GENERAL pass for 7, then 15, then 30 launches between
 *                retries.  Both passes are the exact softmax up to rounding, so any contents give a correct result -- but which pass
 *                runs, against which reference, decides the rounding: with a history the output bits depend on the launches before
 *                (two identical sequences of launches from a zeroed history are bit-identical, eager or replayed from a hipGraph; a
 *                block that never fails never leaves the zero state and computes the stateless kernel's bits); NULL = stateless, every
 *                launch bit-reproducible by itself. */
int mmpl_dit_forward(MmplDit* h, const void* x_in, const float* t_dev, int n_frames, const int* frame_ids,
                     const int* write_slots, const int* visible_slots, int n_visible, void* k_cache, void* v_cache,
                     int n_slots, const void* cross_k, const void* cross_v, int cross_rows, void* share_out, const void* share_in,
                     void* attn_history, void* out, void* workspace, size_t workspace_bytes, mmpl_stream_t stream);
size_t mmpl_dit_attn_history_bytes(const MmplDit* h, int n_frames);
/* MMPL_CHECK_SHARE=1: number of share_in forwards (eager or replayed) since the last call whose layer-0 K / V fingerprint differed
 * from their share_out forward's; synchronises `stream` and resets the count.  0 when the switch is off. */
int mmpl_dit_share_check_failures(MmplDit* h, long long* count, mmpl_stream_t stream);

/* attention() seam (wan/modules/attention.py:139-185) over paged K/V.  q/o: row r, head h at base + r*ld + h*128; every base
 * 16-byte aligned, every leading dimension a multiple of 8 elements (rows are read and written 16 bytes per lane).
 * k_pages/v_pages: host arrays of n_pages dev pointers, each page = page_rows rows of stride ldk/ldv.
 * Softmax does not depend on the order of the keys, the fp32 accumulation does: the 64-rows-per-wave kernel visits the pages in
 * ADDRESS order (back-to-back pages are merged), so a result is bit-reproducible for a given relative placement of the pages.
 * (mmpl_dit_forward keeps its two allocations -- cache slots, scratch pages -- apart, so ITS bits do not depend on placement.) */
int mmpl_attn_fwd(const void* q, int ldq, void* o, int ldo, const void* const* k_pages, const void* const* v_pages,
                  int ldk, int ldv, int n_pages, int page_rows, int Lq, int num_heads, float softmax_scale,
                  mmpl_stream_t stream);
/* Same, with scratch for the split-KV tail round (mmpl_attn_workspace_bytes() is always enough; NULL = mmpl_attn_fwd):
 * when the query blocks do not fill the last round of one-block-per-CU evenly, the leftover blocks are run as 2..4 blocks
 * over disjoint KV ranges plus a merge, which shortens the launch by up to one block time. */
size_t mmpl_attn_workspace_bytes(void);
int mmpl_attn_fwd_ws(const void* q, int ldq, void* o, int ldo, const void* const* k_pages, const void* const* v_pages,
                     int ldk, int ldv, int n_pages, int page_rows, int Lq, int num_heads, float softmax_scale,
                     void* workspace, size_t workspace_bytes, mmpl_stream_t stream);

/* Same, with the kernel chosen by the caller (tests and A/B runs): variant 0 = what mmpl_attn_fwd_ws picks (the lock-step
 * kernel for a raw q), 1 = lock-step 8 x 32 rows (the kernel the text cross-attention uses), 2 = removed (round 1's ping-pong
 * kernel: error), 3 = 4 waves x 64
 * rows (the DiT forward's self-attention kernel) on a raw q, 4 = the same on a q its producer already multiplied by
 * softmax_scale * log2(e) before rounding it to bf16 (what mmpl_dit_forward does: one rounding of q instead of two).
 * cross != 0 tags the launch as a text cross-attention launch (kernel symbol of variant 1 only).  Unknown variant: error. */
int mmpl_attn_fwd_variant(const void* q, int ldq, void* o, int ldo, const void* const* k_pages, const void* const* v_pages,
                          int ldk, int ldv, int n_pages, int page_rows, int Lq, int num_heads, float softmax_scale,
                          void* workspace, size_t workspace_bytes, int variant, int cross, mmpl_stream_t stream);

/* The DiT forward's self-attention launch by itself (variant 4 above: attn_w64_kernel on a q its producer multiplied by
 * softmax_scale * log2(e) before rounding it to bf16), with the two optional pieces of state mmpl_dit_forward threads through it:
 * history = mmpl_attn_history_bytes(Lq, num_heads) device bytes (see mmpl_dit_forward's attn_history; NULL = stateless) and
 * stats_dev = 5 x uint64 (see mmpl_dit_set_attn_stats; NULL = off). */
size_t mmpl_attn_history_bytes(int Lq, int num_heads);
int mmpl_attn_fwd_history(const void* q, int ldq, void* o, int ldo, const void* const* k_pages, const void* const* v_pages,
                          int ldk, int ldv, int n_pages, int page_rows, int Lq, int num_heads, float softmax_scale,
                          void* workspace, size_t workspace_bytes, void* history, void* stats_dev, mmpl_stream_t stream);

/* nn.Linear (+ fused epilogue). epi: 0 bias, 1 bias+GELU(tanh), 2 bias+SiLU, 3 x + (y*gate[frame]) , 4 x + y */
int mmpl_gemm(const void* A, int lda, const void* W, int ldw, const void* bias, void* C, int ldc, int M, int N, int K,
              int epi, const void* res, int ldres, const void* gate, int gate_frame_stride, int rows_per_frame,
              mmpl_stream_t stream);
/* The same GEMM with dynamic tile scheduling for the large-problem kernel: tile_counter = 8 device ints (one per XCD) that are
 * zero when the launch starts (the launch leaves them zero); the kernel is then launched once per CU and its blocks draw
 * tiles of their XCD's share until none is left, instead of one block per tile in lock-step rounds.  Launches sharing a
 * counter must be stream-ordered.  NULL = mmpl_gemm.  (mmpl_dit_forward keeps such a counter in its workspace.) */
int mmpl_gemm_tickets(const void* A, int lda, const void* W, int ldw, const void* bias, void* C, int ldc, int M, int N, int K,
                      int epi, const void* res, int ldres, const void* gate, int gate_frame_stride, int rows_per_frame,
                      void* tile_counter, mmpl_stream_t stream);
/* mmpl_gemm_tickets + a split-K launch for the partial last round of tiles (a GEMM of R * 256 + t tiles on 256 CUs otherwise takes
 * R + 1 rounds however small t is: the Wan 1.3B block GEMMs, every model's 2-frame stage): when the leftover tiles fit one round in
 * 2-4 parts each, they are computed as that many blocks over a share of K each, fp32 partials in `scratch`, summed in part order by
 * the part that finishes last (deterministic), which then runs the ordinary epilogue.  scratch: mmpl_gemm_scratch_bytes() device
 * bytes whose first 2048 are zero when a launch starts (every launch leaves them zero): [8 tile tickets | pad to 256 B | 256 tile
 * counters | pad to 2048 B | partials].  Launches sharing a scratch must be stream-ordered.  (mmpl_dit_forward keeps one in its
 * workspace.) */
size_t mmpl_gemm_scratch_bytes(void);
/* Diagnostic: 1 if workgroup b of a launch runs on XCD b & 7 on the current device (checked on the hardware once per device), else
 * 0.  The tile / head orders use that mapping for L2 locality only; since round 4 no kernel depends on it for correctness (the
 * split-K launch exchanges its partials with system-scope accesses).  Synchronises on first use: not inside a stream capture. */
int mmpl_device_xcd_round_robin(void);
/* Calibration of the box a measurement runs on: the rate (TFLOP/s, dense bf16) the current device sustains on nothing but
 * back-to-back MFMAs on random operands, one wave per SIMD, accumulators in the accumulator file; shape 32 =
 * v_mfma_f32_32x32x16_bf16 (the attention kernels), 16 = v_mfma_f32_16x16x32_bf16 (GEMM, VAE).  Runs for about `seconds` on the
 * null stream and synchronises; reports the second half of the run.  MI355X is power-limited under dense MFMA streams, so this --
 * not the nominal 2.5 PFLOP/s -- is the ceiling wall clock can be priced against on THIS box (bench.py `roofline.sustained_probe_tflops`). */
int mmpl_probe_mfma_tflops(int shape, double seconds, double* tflops);
int mmpl_gemm_scratch(const void* A, int lda, const void* W, int ldw, const void* bias, void* C, int ldc, int M, int N, int K,
                      int epi, const void* res, int ldres, const void* gate, int gate_frame_stride, int rows_per_frame,
                      void* scratch, size_t scratch_bytes, mmpl_stream_t stream);

/* WanLayerNorm (+ per-frame modulation or affine) (wan/modules/model.py:89-99, causal_fps_model.py:343,352,355) */
int mmpl_layernorm(const void* x, int ldx, void* y, int ldy, int rows, int d, float eps, const void* scale,
                   const void* shift, int mod_frame_stride, int rows_per_frame, const void* w, const void* b,
                   mmpl_stream_t stream);

/* WanRMSNorm over the full dim on q (in place) and k, causal_fps_rope_apply, KV slot write
 * (model.py:70-86, causal_fps_model.py:27-55, 211-217).  k_dst/v_dst: host arrays of n_frames dev page pointers. */
int mmpl_qknorm_rope(MmplDit* h, void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* wq,
                     const void* wk, int n_frames, const int* frame_ids, void* const* k_dst, void* const* v_dst,
                     mmpl_stream_t stream);

/* CFG combine + FlowUniPCMultistepScheduler.step (casual_fps_inference.py:366-374, fm_solvers_unipc.py:655-739) */
typedef struct MmplUniPCStep {
  float guidance, sigma_cur;
  int use_corrector, corr_order;
  float c_c1, c_c2, c_c3, c_inv_rk, c_rho0, c_rho_last;
  int pred_order;
  float p_c1, p_c2, p_c3, p_inv_rk;
} MmplUniPCStep;
int mmpl_cfg_unipc_step(const void* flow_cond, const void* flow_uncond, void* x, void* m0, void* m1, void* last_sample,
                        size_t n, const MmplUniPCStep* s, mmpl_stream_t stream);
/* Device-resident form, for ONE hipGraph per denoise step (2 DiT forwards + this) replayed sampling_steps times with no
 * host work in between: the scalars of step *step_dev are read from table_dev[*step_dev] (n_steps entries, device memory),
 * then *step_dev += 1 and the next step's timestep (timestep_table_dev[*step_dev]) is written to timestep_dev[0..n_timestep)
 * -- the tensor the forwards of the next replay read. */
int mmpl_cfg_unipc_step_table(const void* flow_cond, const void* flow_uncond, void* x, void* m0, void* m1, void* last_sample,
                              size_t n, const MmplUniPCStep* table_dev, int* step_dev, float* timestep_dev,
                              const float* timestep_table_dev, int n_timestep, int n_steps, mmpl_stream_t stream);

/* ---- Wan 3D causal VAE (wan/modules/vae.py:483-569 behind WanVAEWrapper, utils/wan_wrapper.py:54-113) ----
 * Weights: mmpl_vae_num_weights() dev pointers in the order of mmpl_vae_weight_name(i): the reference's state_dict keys, conv
 * weights repacked host-side to [Cout, taps * Cin] (tap-major, Cin contiguous; Cin 3/16 zero-padded to 32, decoder.head Cout
 * 3 -> 4).  Every 3x3(x3) stride-1 conv is listed TWICE: "<conv>.weight" as above and a synthetic entry "<conv>.weight.frag" =
 * the same [Cout, taps * Cin] matrix in the fragment-major packing conv_halo_kernel loads (Cout zero-padded to a multiple of 16):
 * element (n, tap, c) at ((((c / 32) * taps + tap) * (Cout_pad / 16) + n / 16) * 4 + (c % 32) / 8) * 16 + n % 16) * 8 + c % 8,
 * i.e. per (32-channel chunk, tap, 16-row group) the 64 lanes' 16 bytes back to back, lane = 16 * (8-channel k chunk) + row
 * (mmpl_amd/vae.py VaeEngine._frag_pack is the reference packer).  bind_weights rejects any other count: a caller that binds
 * by state-dict name must produce the .frag entries too.
 * mode 0 = decode, 1 = encode.  mean / inv_std: host float[16] (bf16-rounded values of the wrapper's scale tensors). */
typedef struct MmplVae MmplVae;
int mmpl_vae_num_weights(void);
const char* mmpl_vae_weight_name(int i);
int mmpl_vae_create(int lat_h, int lat_w, MmplVae** out);
void mmpl_vae_destroy(MmplVae* v);
int mmpl_vae_bind_weights(MmplVae* v, const void* const* dev_ptrs, int n);
size_t mmpl_vae_workspace_bytes(MmplVae* v, int mode);
/* WanVAE_.decode: z dev bf16 [n_frames, 16, lat_h, lat_w] -> out dev float32 [1 + 4(n_frames-1), 3, 8 lat_h, 8 lat_w], clamped */
int mmpl_vae_decode(MmplVae* v, const void* z, int n_frames, const float* mean, const float* inv_std, void* out, void* workspace,
                    size_t workspace_bytes, mmpl_stream_t stream);
/* WanVAE_.encode: px dev bf16 [3, n_px_frames = 1 + 4k, 8 lat_h, 8 lat_w] -> out dev float32 [1 + k, 16, lat_h, lat_w] (normalised mu) */
int mmpl_vae_encode(MmplVae* v, const void* px, int n_px_frames, const float* mean, const float* inv_std, void* out, void* workspace,
                    size_t workspace_bytes, mmpl_stream_t stream);

/* ---- umT5 text encoder (wan/modules/t5.py:267-312 behind WanTextEncoder, utils/wan_wrapper.py:15-51) ----
 * Weights (bf16 dev pointers): [token_embedding.weight, norm.weight] then per block
 * [norm1.weight, pack:attn.{q,k,v}.weight[3*dim_attn,dim], attn.o.weight, pos_embedding.embedding.weight[num_buckets,heads],
 *  norm2.weight, ffn.gate.0.weight, ffn.fc1.weight, ffn.fc2.weight].
 * mmpl_t5_encode: ids / mask dev int32 [text_len]; bucket dev int32 [2*text_len-1] = relative-position bucket of (j - i)
 * at index j - i + text_len - 1 (t5.py:240-264, computed host-side); out dev bf16 [text_len, dim], padding rows zeroed. */
typedef struct MmplT5Config {
  int vocab, dim, dim_attn, dim_ffn, num_heads, num_layers, num_buckets, text_len;
  float eps;
} MmplT5Config;
typedef struct MmplT5 MmplT5;
int mmpl_t5_num_weights(const MmplT5Config* cfg);
int mmpl_t5_create(const MmplT5Config* cfg, MmplT5** out);
void mmpl_t5_destroy(MmplT5* h);
int mmpl_t5_bind_weights(MmplT5* h, const void* const* dev_ptrs, int n);
size_t mmpl_t5_workspace_bytes(const MmplT5* h);
int mmpl_t5_encode(MmplT5* h, const int* ids, const int* mask, const int* bucket, void* out, void* workspace,
                   size_t workspace_bytes, mmpl_stream_t stream);

/* ---- Wan-I2V image cross-attention (wan/modules/model.py:224-266 WanI2VCrossAttention, :469-481 MLPProj).
 * mmpl_i2v_img_proj : clip_fea [n_tok=257, clip_dim=1280] -> context_img [n_tok, dim]; w = {proj.0.weight, proj.0.bias,
 *                     proj.1.weight, proj.1.bias, proj.3.weight, proj.3.bias, proj.4.weight, proj.4.bias} (LayerNorm eps 1e-5,
 *                     erf-GELU -- torch defaults).
 * mmpl_i2v_img_kv   : k_img = norm_k_img(k_img(context_img)), v_img = v_img(context_img)  (once per prompt and layer).
 * mmpl_i2v_cross_attn: out = o(attention(q, k_txt, v_txt) + attention(q, k_img, v_img)), q = norm_q(q(x)); x, out [Lq, dim]. */
size_t mmpl_i2v_img_proj_workspace_bytes(int n_tok, int clip_dim, int dim);
int mmpl_i2v_img_proj(const void* clip_fea, int n_tok, int clip_dim, int dim, const void* const* w, void* out, void* workspace,
                      size_t workspace_bytes, mmpl_stream_t stream);
int mmpl_i2v_img_kv(const void* ctx_img, int n_tok, int dim, const void* wk, const void* bk, const void* wv, const void* bv,
                    const void* norm_k_img_w, float eps, void* k_out, void* v_out, mmpl_stream_t stream);
size_t mmpl_i2v_cross_attn_workspace_bytes(int Lq, int dim);
int mmpl_i2v_cross_attn(const void* x, int Lq, int dim, const void* wq, const void* bq, const void* norm_q_w, float eps,
                        const void* k_txt, const void* v_txt, int n_txt, const void* k_img, const void* v_img, int n_img,
                        const void* wo, const void* bo, void* out, void* workspace, size_t workspace_bytes, mmpl_stream_t stream);

/* CLIP ViT-H/14 vision tower of Wan-I2V: VisionTransformer.forward(x, use_31_block=True) (wan/modules/clip.py:209-327) behind
 * CLIPModel.visual (clip.py:527-542; the bicubic resize / normalisation / im2col are the caller's).
 *   patches : dev [n_patch, pk] bf16, im2col of the normalised image in (c, ky, kx) order, K zero-padded to a multiple of 64
 *   gw      : 5 dev pointers  patch_embedding.weight [dim, pk], cls_embedding [dim], pos_embedding [n_patch+1, dim], pre_norm.{weight,bias}
 *   lw      : 12 per block    norm1.{weight,bias}, to_qkv.{weight [3*heads*128, dim], bias}, proj.{weight [dim, heads*128], bias},
 *                             norm2.{weight,bias}, mlp.0.{weight,bias}, mlp.2.{weight,bias} -- heads padded from head_dim to 128
 *   out     : dev [n_patch+1, dim] bf16, the tokens after n_blocks blocks (31 of the 32 for Wan-I2V) */
size_t mmpl_clip_visual_workspace_bytes(int n_tok, int dim, int mlp_dim, int heads);
int mmpl_clip_visual(const void* patches, int n_patch, int pk, int dim, int mlp_dim, int heads, int head_dim, int n_blocks,
                     const void* const* gw, const void* const* lw, float eps, void* out, void* workspace, size_t workspace_bytes,
                     mmpl_stream_t stream);

/* Optional per-kernel-class hipEvent timing (bench.py's live roofline numbers; off by default, not thread-safe).
 * kinds: 0 gemm, 1 self-attention, 2 cross-attention, 3 layernorm, 4 qk-norm/rope/kv-write, 5 elementwise, 6 cfg+unipc,
 * 7 vae.  on = 0 off, 1 every kind, > 1: only the kinds in the bit mask (on >> 1) (e.g. 2 << 1 | ... ; bench.py times the
 * self-attention only unless --profile-all, an event pair per op costs ~0.5 % of a step when every op is timed).
 * mmpl_profile_read synchronises the device, sums the event pairs recorded since enable/last read. */
int mmpl_profile_enable(int on);
int mmpl_profile_read(int n_kinds, double* ms, double* flops, long long* launches);

const char* mmpl_last_error(void);
const char* mmpl_version(void);

#ifdef __cplusplus
}
#endif
#endif
