/* libds6g.so - C ABI of the MI355X (gfx950) kernel library for the DeepSense6G fusion hot path.
 *
 * The reference (szy4017/DeepSense6G_TII) has no FFI: its boundary is the Python class
 * model2_seq.TransFuser (model2_seq.py:850-894).  deepsense6g_tii_amd/model.py mirrors that class
 * and binds these entry points with ctypes; every function below states the reference op
 * (file:line under /root/reference) whose arithmetic it replaces.
 *
 * Conventions: all pointers are DEVICE pointers (fp32 unless noted); activations are NHWC,
 * conv weights OHWI (= torch channels_last storage of an OIHW parameter), Linear weights [N][K]
 * (torch layout); `stream` is a hipStream_t; every call is asynchronous on that stream, performs
 * no allocation and no synchronisation (hipGraph-capturable).  Return 0 on success, non-zero on
 * bad arguments / launch failure (message on stderr).  `ws` arguments are caller-owned scratch.
 * Dropout masks are a pure function of (seed, seed_off + element index): backward regenerates
 * them, nothing is stored.
 */
#ifndef DS6G_H
#define DS6G_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int ds6g_version(void);
/* which implicit-GEMM instantiation the last conv/linear call launched: 10000*wide + 1000*epilogue + 100*walk +
 * 10*mode + tile (wide 1 = 32-column k-tiles, 0 = 16; epilogue 1 = fused bias/ReLU/dropout/residual epilogue, 0 = plain
 * store; walk 1 = wave-uniform k walk, 0 = general walk; mode 0 fwd, 1 dgrad, 2 wgrad; tile 0 128x128, 1 128x64,
 * 2 64x64), i.e. the template arguments of igemm_kernel<mode, BM, BN, epilogue, BK, bf16, walk> as rocprofv3 prints
 * them.  Bench instrumentation only. */
int ds6g_last_igemm_variant(void);
/* bench instrumentation: between profile_begin and profile_end every implicit-GEMM KERNEL launch (not the split-K
 * reduction that may follow it) is bracketed by HIP events on its launch stream.  profile_end synchronises and returns
 * the number of records written to the three arrays (variant code as above, GEMM flops 2*M*N*K as launched, elapsed
 * milliseconds).  Not thread-safe; bench.py only. */
int ds6g_profile_begin(int max_records);
int ds6g_profile_end(int* variants, double* flops, float* ms, int cap);
/* ablation switches for kernel timing experiments (results become wrong); 0 = normal operation */
int ds6g_set_debug_flags(int flags);
/* multi-GPU rehearsal on one GPU: keep `workgroups` (<= 256) workgroups of 256 threads with `lds_bytes` (<= 64 KiB) of LDS
 * each resident for `microseconds` (<= 2 s; every wave leaves on time) on `stream` - a stand-in for the channel workgroups an
 * RCCL all-reduce keeps on a few CUs while the backward pass runs (train2_seq.py:538 -> dist.py). */
int ds6g_debug_occupy_cus(int workgroups, int lds_bytes, int microseconds, void* stream);
/* matrix-core mode of the conv / linear / attention kernels (process-wide).  Everything else (BN, LN, softmax, loss,
 * AdamW, every accumulator) stays fp32 in every mode.  Returns DS6G_ERR_ARG for any other value.
 *   0 (default) "f32"   exact fp32 MFMA (v_mfma_f32_32x32x2_f32) - the parity path, 1e-3 bar against the reference.
 *   1           "bf16"  the "bf16 forward/backward" throughput configuration of BASELINE.json configs[1] / [4]: the
 *                       fp32-storage entry points round their operands to bf16 (RNE) between the LDS fragment and the
 *                       matrix core; the host (model.py) additionally routes every conv / linear but the 4-channel
 *                       stems, and attention, to the bf16-STORAGE entry points (ds6g_bf16_*: activations, their
 *                       gradients and a per-step shadow of the weights are bf16 in HBM; fp32 master weights).
 *   2           "f32x3" split bf16: each fp32 operand a = hi + lo with hi = bf16(a), lo = bf16(a - hi); a*b is
 *                       evaluated as hi*hi + hi*lo + lo*hi on the bf16 matrix cores with fp32 accumulation - relative
 *                       product error <= ~2^-16 (between fp32 and TF32), fp32 storage.
 *   3           "f32x6" three-way truncating split (a == hi + mid + lo exactly), the six products above 2^-24 summed
 *                       smallest first: fp32-grade products on the bf16 matrix cores, fp32 storage; the 3x3 / stride-1
 *                       convs keep the fp32 Winograd kernels. */
int ds6g_set_compute_mode(int mode);
int ds6g_get_compute_mode(void);

/* Dropout masks are a pure function of (seed, counter).  Elementwise sites (embd_drop, resid_drop: ds6g_dropout, the
 * linear epilogues, ds6g_layernorm_bwd's dx_drop, the token-pool kernels) use counter = seed_off + element index, one
 * hash (lowbias32 finalizer on the keyed counter) per element, dropped when hash < floor(p * 2^32).  The attention kernels
 * (attn_drop on the [B * nh][T][T] probabilities) draw FOUR decisions per hash: element (row, key), row = (b * nh + h) * T +
 * query, uses counter = seed_off + row * ceil(T / 4) + (key >> 2) and the 16-bit half (key & 3) of the two words derived
 * from that counter, dropped when the half < floor(p * 2^16) (csrc/common.h Ds6gKeep4Base; numpy restatement in
 * tests/test_bench_shapes_gpu.py::_keep_bits_attn); kept probabilities are scaled by 1 / (1 - floor(p * 2^16) / 2^16), the
 * probability the 16-bit decisions realise (p < 2^-16: no dropout); the elementwise sites scale by 1 / (1 - p).  seed_off is a launch argument, frozen when a
 * training step is captured into a hipGraph; ds6g_set_dropout_salt(dev_ptr) makes every kernel that draws a mask add the
 * uint64 at dev_ptr to seed_off when it RUNS (the caller advances that value on the device once per step), so replays draw
 * fresh masks.  Applies to the launches of the calling thread from now on (thread-local); NULL switches it off. */
int ds6g_set_dropout_salt(const uint64_t* dev_ptr);

/* ---- igemm.hip : Conv2d / Linear as implicit GEMM on v_mfma_f32_32x32x2_f32 -------------------
 * Conv2d(bias=False) of the ResNet trunks: model2_seq.py:495,500,505 (7x7/2 stems), :510-512,
 * 528-530,546-548,565-567 (BasicBlock 3x3 and 1x1/2 downsample).  C and K multiples of 4. */
int ds6g_conv2d_fwd(const float* x, const float* w, float* y, int N, int H, int W, int C, int K, int R, int S,
                    int stride, int pad, void* stream);
int ds6g_conv2d_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int K, int R, int S,
                      int stride, int pad, int accumulate, void* stream);
int ds6g_conv2d_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K, int R, int S,
                      int stride, int pad, int accumulate, float* ws, size_t ws_bytes, void* stream);
/* inference form of Conv2d + eval-mode BatchNorm2d (+ identity) (+ ReLU) of the torchvision BasicBlock / stem
 * (call sites model2_seq.py:495-512,528-530,546-548,565-567 under model.eval(), train2_seq.py:161): the BN is folded
 * into w / bias by ds6g_bn_fold, y = act(conv(x, w) + bias [+ residual]); relu: 0 none, 1 before the residual add,
 * 2 after it. */
int ds6g_conv2d_bias_act_fwd(const float* x, const float* w, const float* bias, const float* residual, float* y, int N,
                             int H, int W, int C, int K, int R, int S, int stride, int pad, int relu, void* stream);
/* w_out[o][tap][c] = w[o][tap][c] * gamma[o]/sqrt(running_var[o]+eps) (c >= cin: zero padding up to cpad channels),
 * bias_out[o] = beta[o] - running_mean[o] * gamma[o]/sqrt(running_var[o]+eps);  w: [K][taps][cin] (OHWI). */
int ds6g_bn_fold(const float* w, const float* gamma, const float* beta, const float* running_mean,
                 const float* running_var, float eps, float* w_out, float* bias_out, int K, int taps, int cin, int cpad,
                 void* stream);
/* ---- winograd.hip : Winograd F(2x2, 3x3) for the 3x3 / stride 1 / pad 1 convolutions of the BasicBlocks
 * (model2_seq.py:510-512,528-530,546-548,565-567): 16 GEMMs on transformed 4x4 tiles, 2.25x fewer MFMA FLOPs.
 * winograd_weights builds U[16][K][C] = G g G^T from the OHWI filter (transpose_flip = 1: the dgrad filter, i.e.
 * U[16][C][K] of the channel-swapped, 180-degree-rotated filter; 2: both in one launch, forward first, into 2x the
 * floats), stored in MFMA fragment order; conv3x3_winograd_fwd computes
 * y[N][H][W][K] (+)= conv(x[N][H][W][C]) from it (call it with dy and the dgrad filter to obtain dx).
 * winograd_supported: H, W even, C % 16 == 0, K % 32 == 0, W/2 a multiple of 8 or a divisor of 32. */
size_t ds6g_winograd_weight_floats(int K, int C);
int ds6g_winograd_weights(const float* w, float* u, int K, int C, int transpose_flip, void* stream);
int ds6g_winograd_supported(int N, int H, int W, int C, int K);
int ds6g_conv3x3_winograd_fwd(const float* x, const float* u, float* y, int N, int H, int W, int C, int K, int accumulate,
                              void* stream);
/* inference form: eval-mode BN folded into the filter (ds6g_bn_fold, then ds6g_winograd_weights), epilogue fused into the
 * output transform: y = act(conv(x) + bias [+ residual]); relu 0 none / 1 before / 2 after the residual add. */
int ds6g_conv3x3_winograd_bias_act_fwd(const float* x, const float* u, const float* bias, const float* residual, float* y,
                                       int N, int H, int W, int C, int K, int relu, void* stream);
/* weight gradient of the same convs in the Winograd domain (dU = sum_tiles (A dY A^T) (B^T d B), dW = G^T dU G): 4 MFMA
 * FLOPs per pixel / channel pair / filter instead of 9.  ws: scratch for the per-split dU slabs (>= 16*K*C floats; more
 * allows more splits).  Supported: H, W even, C and K multiples of 64. */
int ds6g_winograd_wgrad_supported(int N, int H, int W, int C, int K);
int ds6g_conv3x3_winograd_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K,
                                int accumulate, float* ws, size_t ws_bytes, void* stream);
/* nn.Linear of the GPT blocks with fused epilogue y = residual + dropout(act(x w^T + b)):
 * model2_seq.py:97-99 (q,k,v), :109 (proj + resid_drop), :121-126 (MLP, ReLU), :131-132 (residuals). */
int ds6g_linear_fwd(const float* x, const float* w, const float* bias, float* y, int M, int N, int K, int relu,
                    const float* residual, float drop_p, uint64_t seed, uint64_t seed_off, void* stream);
int ds6g_linear_dgrad(const float* dy, const float* w, float* dx, int M, int N, int K, const float* relu_mask_src,
                      int accumulate, void* stream);
int ds6g_linear_wgrad(const float* x, const float* dy, float* dw, float* dbias, int M, int N, int K, int accumulate,
                      float* ws, size_t ws_bytes, void* stream);

/* ---- bgemm.hip : the same Conv2d / Linear products on bf16-STORED operands (the bf16 configuration of BASELINE configs[1]
 * and [4]: bf16 activations, a bf16 shadow of the weights refreshed by ds6g_adamw_step, fp32 accumulation; the reference has
 * no mixed precision, train2_seq.py:111-116).  x / dy / w: bf16; y / dx: bf16 when out16 else fp32; dw / dbias / bias /
 * residual: fp32.  Tiles travel HBM -> LDS as bf16 by LDS-DMA and feed v_mfma_f32_32x32x16_bf16 without conversion.
 * Shape limits (every layer of the model but the 4-channel stems): the reduction channel count is a multiple of 64, the
 * other a multiple of 8; dgrad: stride 1, or 2 with even H, W; wgrad: Wo % 64 == 0, or 64 % Wo == 0 with Ho % (64 / Wo) == 0, or a Linear. */
int ds6g_bf16_conv2d_fwd(const void* x, const void* w, void* y, int out16, int N, int H, int W, int C, int K, int R, int S,
                         int stride, int pad, void* stream);
/* conv (bf16 output) + the train-mode BatchNorm statistics of its output in one call (the BasicBlock pairs conv1 / bn1,
 * conv2 / bn2, downsample.0 / .1 of torchvision's ResNet, model2_seq.py:510-512,528-530,546-548,565-567): the conv's epilogue
 * writes per-tile column sums / sums of squares of the STORED bf16 tile, a small finalize kernel turns them into mean /
 * invstd and updates the running statistics - what ds6g_bf16_bn_stats(y) computes, without its pass over y. */
size_t ds6g_bf16_conv_bnstats_workspace_bytes(long M, int K);
int ds6g_bf16_conv2d_fwd_bnstats(const void* x, const void* w, void* y, int N, int H, int W, int C, int K, int R, int S,
                                 int stride, int pad, float eps, float momentum, float* mean, float* invstd,
                                 float* running_mean, float* running_var, void* ws, size_t ws_bytes, void* stream);
int ds6g_bf16_conv2d_dgrad(const void* dy, const void* w, void* dx, int out16, int N, int H, int W, int C, int K, int R,
                           int S, int stride, int pad, int accumulate, void* stream);
int ds6g_bf16_conv2d_wgrad(const void* x, const void* dy, float* dw, int N, int H, int W, int C, int K, int R, int S,
                           int stride, int pad, int accumulate, float* ws, size_t ws_bytes, void* stream);
/* nn.Linear of the GPT blocks (model2_seq.py:97-99,109,121-126,131-132) on bf16 operands; a fused residual add writes the
 * fp32 residual stream (out16 must be 0 then); mask_src [M][K]: bf16 (mask16) or fp32, with a bf16 dx only. */
int ds6g_bf16_linear_fwd(const void* x, const void* w, const float* bias, void* y, int out16, int M, int N, int K, int relu,
                         const float* residual, float drop_p, uint64_t seed, uint64_t seed_off, void* stream);
int ds6g_bf16_linear_dgrad(const void* dy, const void* w, void* dx, int out16, int M, int N, int K, const void* mask_src,
                           int mask16, int accumulate, void* stream);
int ds6g_bf16_linear_wgrad(const void* x, const void* dy, float* dw, float* dbias, int M, int N, int K, int accumulate,
                           float* ws, size_t ws_bytes, void* stream);

/* ---- stem.hip : the 7x7 / stride 2 / pad 3 stem convs (torchvision ResNet conv1 via model2_seq.py:495,500,505) of the
 * bf16 configuration.  x [N][H][W][4] bf16 (ds6g_pack_input_bf16; channels >= cin zero), w: the fp32 master filter
 * [64][7][7][cin] (OHWI), y / dy [N][H/2][W/2][64] bf16; H % 16 == 0, W % 32 == 0.  The forward also delivers the train-mode
 * BatchNorm statistics of y (mean == NULL: convolution only); ws >= ds6g_bf16_stem_workspace_bytes().  The BN -> ReLU ->
 * MaxPool pass over the bf16 conv output and its backward: ds6g_bf16_stem_bn_relu_maxpool_fwd / ds6g_bf16_stem_bn_bwd_maxpool. */
size_t ds6g_bf16_stem_workspace_bytes(void);
int ds6g_bf16_stem_fwd(const void* x, const float* w, int cin, void* y, int N, int H, int W, float eps, float momentum,
                       float* mean, float* invstd, float* running_mean, float* running_var, void* ws, size_t ws_bytes,
                       void* stream);
int ds6g_bf16_stem_wgrad(const void* x, const void* dy, float* dw, int cin, int N, int H, int W, int accumulate, void* ws,
                         size_t ws_bytes, void* stream);
int ds6g_bf16_stem_bn_relu_maxpool_fwd(const void* x, const float* mean, const float* invstd, const float* gamma,
                                       const float* beta, void* y, uint8_t* idx, int N, int H, int W, int C, void* stream);
int ds6g_bf16_stem_bn_bwd_maxpool(const void* dpool, const uint8_t* idx, const void* x, const float* mean,
                                  const float* invstd, const float* gamma, const float* relu_beta, void* dx, float* dgamma,
                                  float* dbeta, int N, int H, int W, int C, int accumulate_param_grads, void* ws,
                                  size_t ws_bytes, void* stream);

/* ---- norm.hip ----------------------------------------------------------------------------------
 * BatchNorm2d in train mode (+ReLU, +residual add of BasicBlock): torchvision BasicBlock via
 * model2_seq.py:496-497,501-502,506-507 and the layer calls above; eval mode uses running stats. */
size_t ds6g_bn_workspace_bytes(long M, int C);
int ds6g_bn_stats(const float* x, long M, int C, float eps, float momentum, float* mean, float* invstd,
                  float* running_mean, float* running_var, void* ws, size_t ws_bytes, void* stream);
int ds6g_bn_eval_prepare(const float* running_mean, const float* running_var, int C, float eps, float* mean,
                         float* invstd, void* stream);
int ds6g_bn_apply(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                  const float* residual, float* y, long M, int C, int relu, void* stream);
/* relu_beta (nullable, y_mask then NULL): the BN fed a ReLU with no residual in between (bn1 of a BasicBlock, stem
 * bn1): the ReLU mask is re-derived from x as (gamma * xhat + relu_beta > 0) with the forward's own arithmetic instead of
 * reading the activation tensor - one tensor read less in each of the two backward passes. */
int ds6g_bn_bwd(const float* dy, const float* y_mask, const float* x, const float* mean, const float* invstd,
                const float* gamma, const float* relu_beta, float* dx, float* dgamma, float* dbeta, float* dres, long M,
                int C, int accumulate_param_grads, void* ws, size_t ws_bytes, void* stream);
/* nn.LayerNorm(C), eps 1e-5: model2_seq.py:118-119,131-132,199,274.  C in {64,128,256,512}. */
int ds6g_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                       int M, int C, float eps, void* stream);
size_t ds6g_layernorm_bwd_workspace_bytes(int M, int C);
/* dx = add? + LN'(dy); dgamma / dbeta (+)=.  dx_drop (nullable): also writes dropout(dx; drop_p, seed, seed_off), the
 * gradient the resid_drop branch of the block below consumes (model2_seq.py:109,126), saving a separate pass. */
int ds6g_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                       const float* add, float* dx, float* dgamma, float* dbeta, int M, int C,
                       int accumulate_param_grads, float* dx_drop, float drop_p, uint64_t seed, uint64_t seed_off,
                       void* ws, size_t ws_bytes, void* stream);
/* bf16-storage path of BatchNorm2d: conv outputs, activations and their gradients are bf16 in HBM (x / residual / y /
 * dy / y_mask / dx / dres), the statistics, running stats, parameter gradients and every reduction stay fp32 / fp64. */
int ds6g_bf16_bn_stats(const void* x, long M, int C, float eps, float momentum, float* mean, float* invstd,
                       float* running_mean, float* running_var, void* ws, size_t ws_bytes, void* stream);
int ds6g_bf16_bn_apply(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                       const void* residual, void* y, long M, int C, int relu, void* stream);
int ds6g_bf16_bn_bwd(const void* dy, const void* y_mask, const void* x, const float* mean, const float* invstd,
                     const float* gamma, const float* relu_beta, void* dx, float* dgamma, float* dbeta, void* dres, long M,
                     int C, int accumulate_param_grads, void* ws, size_t ws_bytes, void* stream);
/* bf16-storage path (GEMM-facing tensors are bf16; statistics, the residual stream and all arithmetic stay fp32):
 * layernorm_fwd_bf16out writes y as bf16; layernorm_bwd_bf16 reads dy as bf16 (dy16) or fp32, writes dx fp32 and dx_drop
 * (nullable) = dropout(dx) as bf16 (drop_p = 0: a bf16 copy of dx, the next GEMM's operand). */
int ds6g_layernorm_fwd_bf16out(const float* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                               int M, int C, float eps, void* stream);
int ds6g_layernorm_bwd_bf16(const void* dy, int dy16, const float* x, const float* mean, const float* rstd,
                            const float* gamma, const float* add, float* dx, float* dgamma, float* dbeta, int M, int C,
                            int accumulate_param_grads, void* dx_drop, float drop_p, uint64_t seed, uint64_t seed_off,
                            void* ws, size_t ws_bytes, void* stream);
/* bias gradients: out[c] (+)= sum_r x[r][c] */
size_t ds6g_colsum_workspace_bytes(long M, int C);
int ds6g_colsum(const float* x, long M, int C, float* out, int accumulate, void* ws, size_t ws_bytes, void* stream);

/* ---- attention.hip : SelfAttention core, model2_seq.py:101-106 (softmax(q k^T/sqrt(hd)), attn_drop,
 * att @ v, head merge).  Head h lives at columns h*hd of every operand.  q/k/v: [B*T][ld_qkv] (ld_qkv = 3C when
 * they are column blocks of ONE fused key|query|value projection output, model2_seq.py:97-99); o / d_o: [B*T][ld];
 * dq/dk/dv: [B*T][ld_dqkv] (again 3C to write one fused gradient matrix).  hd in {16,32,64,128}.
 * ws: scratch for the split-loop partial results (any size; more allows more splits, see *_workspace_bytes).  With the
 * full *_workspace_bytes the backward hands its dS (and, hd = 128, dropped-P) tiles from the dK/dV kernel to the dQ / dV
 * kernels through ws (5 matrix products); with less it recomputes the scores in every kernel (7-8 products, same results
 * up to summation order). */
size_t ds6g_attention_workspace_bytes(int B, int T, int nh, int hd, int ld);
int ds6g_attention_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int B, int T, int nh,
                       int hd, int ld_qkv, int ld, float drop_p, uint64_t seed, uint64_t seed_off, void* ws,
                       size_t ws_bytes, void* stream);
int ds6g_attention_bwd(const float* q, const float* k, const float* v, const float* o, const float* d_o,
                       const float* lse, float* delta, float* dq, float* dk, float* dv, int B, int T, int nh, int hd,
                       int ld_qkv, int ld, int ld_dqkv, float drop_p, uint64_t seed, uint64_t seed_off, void* ws,
                       size_t ws_bytes, void* stream);

/* bf16-storage path: q / k / v / d_o stay fp32 (column blocks of fp32 GEMM outputs); the forward output o is written -
 * and read back by the backward - as bf16 [B*T][ld], dq / dk / dv are written as bf16 [B*T][ld_dqkv] (operands of the
 * projection GEMMs).  The backward needs the full ds6g_attention_workspace_bytes (hand-over form). */
int ds6g_attention_fwd_bf16out(const float* q, const float* k, const float* v, void* o, float* lse, int B, int T, int nh,
                               int hd, int ld_qkv, int ld, float drop_p, uint64_t seed, uint64_t seed_off, void* ws,
                               size_t ws_bytes, void* stream);
int ds6g_attention_bwd_bf16(const float* q, const float* k, const float* v, const void* o, const float* d_o,
                            const float* lse, float* delta, void* dq, void* dk, void* dv, int B, int T, int nh, int hd,
                            int ld_qkv, int ld, int ld_dqkv, float drop_p, uint64_t seed, uint64_t seed_off, void* ws,
                            size_t ws_bytes, void* stream);

/* all operands bf16 in HBM (q / k / v [B*T][ld_qkv], o / d_o [B*T][ld], dq / dk / dv [B*T][ld_dqkv]): K / V / Q / dO tiles are
 * staged as bf16 LDS images by LDS-DMA and reach v_mfma_f32_32x32x16_bf16 unconverted (row reads: ds_read_b128, transposed
 * reads for the P.V / dS^T.Q / P^T.dO / dS.K products: ds_read_b64_tr_b16); lse, delta, the split slabs and the dS / P
 * hand-over tiles stay fp32.  ld_qkv and ld multiples of 8. */
int ds6g_attention_fwd_bf16(const void* q, const void* k, const void* v, void* o, float* lse, int B, int T, int nh, int hd,
                            int ld_qkv, int ld, float drop_p, uint64_t seed, uint64_t seed_off, void* ws, size_t ws_bytes,
                            void* stream);
int ds6g_attention_bwd_bf16io(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                              float* delta, void* dq, void* dk, void* dv, int B, int T, int nh, int hd, int ld_qkv, int ld,
                              int ld_dqkv, float drop_p, uint64_t seed, uint64_t seed_off, void* ws, size_t ws_bytes,
                              void* stream);

/* ---- input.hip : device-side counterpart of CARLA_Data.__getitem__, data2_seq.py:42-173 --------
 * pack_image_u8: decoded RGB frame batch [B][H][W][3] uint8 (data2_seq.py:110-141, before its HWC->CHW transpose) ->
 *   frame slot t of the NHWC x4 stem input [(b*frames_per_sample + t)][H][W][4], cast + normalize_imagenet
 *   (model2_seq.py:36-45) fused; flip mirrors W (data2_seq.py:144-146).
 * lidar_bev_count / lidar_bev_finish: lidar_to_histogram_features (data2_seq.py:177-211).  points: float64
 *   [npoints][point_stride] (x, y first) of nclouds clouds back to back, cloud c = [cloud_offsets[c],
 *   cloud_offsets[c+1]); xedges / yedges: nbins+1 float64 bin edges - one set, or one per cloud when edges_per_cloud
 *   (the per-scenario custom field of view, data2_seq.py:190-202) - (np.linspace in the reference, np.histogramdd
 *   semantics: half-open bins, last bin closed, outliers dropped); counts: [nclouds][nbins][nbins] uint32, must be zero
 *   on entry.  finish writes min(count, cap)/cap into channel 0 of frame slot t of an NHWC xCd tensor (flip mirrors
 *   the y axis, data2_seq.py:157-158) and re-zeroes counts.
 * soft_beam_target: target[b][k] = 1.25 * N(k; beamidx[b], 0.5) for |k - beamidx[b]| <= 5 else 0 (data2_seq.py:160-170);
 *   flip mirrors the beam axis and writes beamidx_out[b] = nbeams-1-beamidx[b] (nullable). */
int ds6g_pack_image_u8(const uint8_t* src_hwc, float* dst, int B, int H, int W, int frames_per_sample, int t, int flip,
                       void* stream);
int ds6g_lidar_bev_count(const double* points, int point_stride, const long* cloud_offsets, int nclouds,
                         long npoints, const double* xedges, const double* yedges, int edges_per_cloud, int nbins,
                         unsigned* counts, void* stream);
int ds6g_lidar_bev_finish(unsigned* counts, float* dst, int B, int nbins, int Cd, int frames_per_sample, int t, int flip,
                          int cap, void* stream);
int ds6g_soft_beam_target(const int* beamidx, float* target, int* beamidx_out, int B, int nbeams, int flip,
                          void* stream);

/* ---- gru.hip : autoregressive GRU beam-sequence head of the 30->5 variant, model2_seq_30to5.py:842-862:
 * x = 0, h = h0 (the join output); T times: h = GRUCell(x, h); x = x + Linear(h); pred[:, t] = x.  H must be 64.
 * w_ih / w_hh: [3H][H] (gate rows r, z, n as in nn.GRUCell), b_ih / b_hh: [3H], w_out: [H][H], b_out: [H].
 * saved: B*T*6*H floats (gru_head_saved_floats) written by fwd for bwd, or NULL for inference.
 * bwd writes dh0 [B][H] and one parameter-gradient slab per sample (gru_head_slab_floats each, laid out dW_ih, dW_hh,
 * db_ih, db_hh, dW_out, db_out); sum the slabs over the batch with ds6g_batch_sum. */
size_t ds6g_gru_head_saved_floats(int B, int T);
size_t ds6g_gru_head_slab_floats(void);
int ds6g_gru_head_fwd(const float* h0, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                      const float* w_out, const float* b_out, float* pred, float* saved, int B, int T, int H,
                      void* stream);
int ds6g_gru_head_bwd(const float* dpred, const float* h0, const float* saved, const float* w_ih, const float* w_hh,
                      const float* w_out, float* dh0, float* slabs, int B, int T, int H, void* stream);

/* ---- spatial.hip -------------------------------------------------------------------------------*/
/* normalize_imagenet + stack + NCHW->NHWC: model2_seq.py:36-45,481-482,491-493 */
int ds6g_pack_input(const float* src, float* dst, int B, int Cs, int H, int W, int Cd, int frames_per_sample, int t,
                    int normalize_imagenet, void* stream);
int ds6g_pack_input_bf16(const float* src, void* dst, int B, int Cs, int H, int W, int frames_per_sample, int t,
                         int normalize_imagenet, void* stream);
int ds6g_pad_channels(const float* src, float* dst, long rows, int cin, int cout, int unpad, int accumulate,
                      void* stream);
/* MaxPool2d(3,2,1) of the stems: model2_seq.py:498,503,508 */
int ds6g_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int N, int H, int W, int C, void* stream);
int ds6g_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, float* dx, int N, int H, int W, int C, void* stream);
/* the ResNet stem's BN -> ReLU -> MaxPool(3, 2, 1) (torchvision resnet.py forward, used at model2_seq.py:495-500) without
 * materialising the activation: forward = ds6g_bn_apply(relu) + ds6g_maxpool3x3s2_fwd in one pass (bit-identical);
 * backward = ds6g_maxpool3x3s2_bwd + ds6g_bn_bwd(relu_beta) with the pool gradient gathered from (dpool, idx) inside the
 * BN kernels.  x: the conv output [N][H][W][C]; mean / invstd from ds6g_bn_stats or ds6g_bn_eval_prepare. */
int ds6g_bn_relu_maxpool3x3s2_fwd(const float* x, const float* mean, const float* invstd, const float* gamma,
                                  const float* beta, float* y, uint8_t* idx, int N, int H, int W, int C, void* stream);
int ds6g_bn_bwd_maxpool(const float* dpool, const uint8_t* idx, const float* x, const float* mean, const float* invstd,
                        const float* gamma, const float* relu_beta, float* dx, float* dgamma, float* dbeta, int N, int H,
                        int W, int C, int accumulate_param_grads, void* ws, size_t ws_bytes, void* stream);
/* bf16-storage path: the 4-channel stem keeps the fp32-storage conv kernels (x, dx fp32); its pooled output is written as
 * bf16 (first tensor of the bf16 trunk) and the gradient of the pooled tensor arrives as bf16 */
int ds6g_bn_relu_maxpool3x3s2_fwd_bf16out(const float* x, const float* mean, const float* invstd, const float* gamma,
                                          const float* beta, void* y, uint8_t* idx, int N, int H, int W, int C,
                                          void* stream);
int ds6g_bn_bwd_maxpool_bf16in(const void* dpool, const uint8_t* idx, const float* x, const float* mean,
                               const float* invstd, const float* gamma, const float* relu_beta, float* dx, float* dgamma,
                               float* dbeta, int N, int H, int W, int C, int accumulate_param_grads, void* ws,
                               size_t ws_bytes, void* stream);
/* bf16-storage path of the pooling / resampling / head kernels below: feature maps (feat, out, dfeat, dout) bf16, tokens /
 * pos_emb / pooled vectors fp32 */
int ds6g_bf16_avgpool_tokens_fwd(const void* feat, const float* pos_emb, float* tokens, int N, int H, int C,
                                 int frames_per_sample, int mod_off, int T, float drop_p, uint64_t seed,
                                 uint64_t seed_off, void* stream);
int ds6g_bf16_avgpool_tokens_bwd(const float* dtok, const void* dfeat_in, void* dfeat, int N, int H, int C,
                                 int frames_per_sample, int mod_off, int T, void* stream);
int ds6g_bf16_upsample_add_fwd(const void* feat, const float* tokens, void* out, int N, int H, int C,
                               int frames_per_sample, int mod_off, int T, void* stream);
int ds6g_bf16_upsample_add_bwd(const void* dout, float* dtok, int N, int H, int C, int frames_per_sample, int mod_off,
                               int T, void* stream);
int ds6g_bf16_global_pool(const void* feat, float* pooled, int N, int C, void* stream);
int ds6g_bf16_head_bwd(const float* dfused, void* dfeat, int N, int C, int frames_per_sample, void* stream);
/* AdaptiveAvgPool2d((8,8)) + token pack + pos_emb + embd dropout: model2_seq.py:414,515-517,261-272 */
int ds6g_avgpool_tokens_fwd(const float* feat, const float* pos_emb, float* tokens, int N, int H, int C,
                            int frames_per_sample, int mod_off, int T, float drop_p, uint64_t seed,
                            uint64_t seed_off, void* stream);
int ds6g_gps_tokens_fwd(const float* emb, const float* pos_emb, float* tokens, int B, int C, int T, float drop_p,
                        uint64_t seed, uint64_t seed_off, void* stream);
int ds6g_dropout(const float* src, float* dst, long n, float drop_p, uint64_t seed, uint64_t seed_off, void* stream);
int ds6g_avgpool_tokens_bwd(const float* dtok, const float* dfeat_in, float* dfeat, int N, int H, int C,
                            int frames_per_sample, int mod_off, int T, void* stream);
/* token unpack + F.interpolate(bilinear, align_corners=False) + residual: model2_seq.py:275-287,521-526,
 * 539-544,558-563,577-579 */
int ds6g_upsample_add_fwd(const float* feat, const float* tokens, float* out, int N, int H, int C,
                          int frames_per_sample, int mod_off, int T, void* stream);
int ds6g_upsample_add_bwd(const float* dout, float* dtok, int N, int H, int C, int frames_per_sample, int mod_off,
                          int T, void* stream);
/* features.avgpool + flatten + cat + sum(dim=1): model2_seq.py:581-595 */
int ds6g_global_pool(const float* feat, float* pooled, int N, int C, void* stream);
int ds6g_head_sum(const float* pooled_img, const float* pooled_lidar, const float* pooled_radar, const float* tokens,
                  float* fused, int B, int C, int fps_img, int fps_other, int T, void* stream);
int ds6g_head_bwd(const float* dfused, float* dfeat, int N, int C, int frames_per_sample, void* stream);
int ds6g_feat_to_tokens(const float* dfeat, float* dtok, int N, int C, int frames_per_sample, int mod_off, int T,
                        void* stream);
int ds6g_gps_rows(const float* src, float* dst, int B, int C, int T, int to_tokens, int accumulate, int src_bcast,
                  void* stream);
int ds6g_axpby(const float* a, const float* b, float* out, long n, float alpha, float beta, void* stream);
int ds6g_batch_sum(const float* src, float* out, long n, int count, long stride, int accumulate, void* stream);

/* ---- step.hip ----------------------------------------------------------------------------------*/
/* FocalLoss -> torchvision.ops.sigmoid_focal_loss(alpha .25, gamma 2, mean): train2_seq.py:291-301 */
int ds6g_focal_loss(const float* logits, const float* target, float* loss, float* dlogits, int n, float alpha,
                    float gamma, float upstream, void* stream);
/* optim.AdamW step (train2_seq.py:131,539) fused with EMA.update (train2_seq.py:315-320) */
/* grad_scale_dev (nullable): one device float multiplied into grad_scale at run time - the clip coefficient below. */
int ds6g_adamw_step(float* p, const float* g, float* m, float* v, float* shadow, long n, int step, float lr,
                    float beta1, float beta2, float eps, float wd, float ema_decay, float grad_scale,
                    const float* grad_scale_dev, void* stream);
/* the same step with its per-step scalars in DEVICE memory, so that a hipGraph capture of the whole training step can be
 * replayed: state = 16 bytes {float lr, float bc1, float bc2_sqrt, int step}, zero-initialised; the host writes lr when the
 * schedule changes it; ds6g_adamw_state_advance (one thread) increments step and recomputes the bias corrections. */
int ds6g_adamw_state_advance(void* state, float beta1, float beta2, void* stream);
int ds6g_adamw_step_dev(float* p, const float* g, float* m, float* v, float* shadow, long n, const void* state, float beta1,
                        float beta2, float eps, float wd, float ema_decay, float grad_scale, const float* grad_scale_dev,
                        void* stream);
/* torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm) of the 30->5 training step (train2_seq_30to5.py:120) on
 * the flat gradient arena: out[0] = pre_scale * ||g||_2, out[1] = min(1, max_norm / (out[0] + 1e-6)); the gradients are
 * not rewritten - pass out + 1 to ds6g_adamw_step as grad_scale_dev.  One launch (last-block-done reduction in index
 * order: deterministic).  ws: >= 8 KiB + 8 bytes, its first 4 bytes zero on first use. */
int ds6g_grad_norm_clip(const float* g, long n, float max_norm, float pre_scale, float* out, void* ws, size_t ws_bytes,
                        void* stream);
/* dst (bf16, RNE) = src (fp32): refreshes the bf16 shadow of the parameter arena for the bf16-storage path */
int ds6g_cast_f32_bf16(const float* src, void* dst, long n, void* stream);
/* vel_emb1..4 and the join MLP: model2_seq.py:422-425,518,536,555,574,863-869 */
int ds6g_small_linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int N, int K,
                          int rows_per_group, long group_stride, int relu, void* stream);
int ds6g_small_linear_bwd(const float* dy, const float* y_mask, const float* x, const float* w, float* dx, float* dw,
                          float* db, int M, int N, int K, int rows_per_group, long group_stride,
                          int dx_rows_per_group, long dx_group_stride, int accumulate_dx, int accumulate_params,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DS6G_H */
