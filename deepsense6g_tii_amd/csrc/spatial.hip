// HBM-bound spatial / token-layout kernels of the fusion path (fp32, NHWC, 16-B per lane):
//   input pack (normalize_imagenet + NCHW->NHWC frame interleave)   model2_seq.py:36-45,481-493
//   3x3/2 max-pool fwd/bwd (stem)                                   model2_seq.py:498,503,508
//   adaptive 8x8 avg-pool straight into the GPT token buffer (+pos_emb, dropout)  :515-517,261-272
//   bilinear upsample (align_corners=False) + residual add, fwd/bwd :521-526,539-544,558-563,577-579
//   global avg-pool + 17-token sum (head)                           :581-595
// Token buffer layout: x[b][tok][c], tok = mod_off + (f % fps)*64 + ph*8 + pw for frame f of a
// modality with fps frames per sample (image: n_views*seq_len, LiDAR/radar: seq_len); the two GPS
// tokens are the last two rows of each sample.  Because features are NHWC the pooled 8x8 maps ARE
// token rows: pack/unpack (model2_seq.py:261-270, 275-287) costs no copy.
#include "common.h"

namespace {

__device__ __forceinline__ long tok_row(int f, int fps, int mod_off, int T, int hw) {
    const int b = f / fps;
    return (long)b * T + mod_off + (f - b * fps) * 64 + hw;
}

// ------------------------------------------------------------------------------------------------
template <typename TD>
__global__ __launch_bounds__(256) void pack_input_kernel(const float* __restrict__ src, TD* __restrict__ dst,
                                                         int B, int Cs, int HW, int Cd, int fps, int t,
                                                         int normalize) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * HW) return;
    const int b = (int)(i / HW);
    const int pix = (int)(i - (long)b * HW);
    const float mean[3] = {0.485f, 0.456f, 0.406f};
    const float stdv[3] = {0.229f, 0.224f, 0.225f};
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < Cs; ++c) {
        float x = src[((long)b * Cs + c) * HW + pix];
        if (normalize) x = (x / 255.0f - mean[c]) / stdv[c];
        v[c] = x;
    }
    TD* o = dst + ((long)(b * fps + t) * HW + pix) * Cd;
    if (Cd == 4) {
        st4(o, f32x4{v[0], v[1], v[2], v[3]});
    } else {
        for (int c = 0; c < Cd; ++c) o[c] = (TD)v[c];
    }
}

// rows x cin  ->  rows x cout (zero padded) and back
__global__ void pad_channels_kernel(const float* __restrict__ src, float* __restrict__ dst, long rows, int cin,
                                    int cout, int unpad, int accumulate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (unpad) {
        if (i >= rows * cin) return;
        const long r = i / cin;
        const int c = (int)(i - r * cin);
        const float v = src[r * cout + c];
        dst[i] = accumulate ? dst[i] + v : v;
    } else {
        if (i >= rows * cout) return;
        const long r = i / cout;
        const int c = (int)(i - r * cout);
        dst[i] = c < cin ? src[r * cin + c] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------
// bn_mean != NULL: the pooled tensor is relu(BN(x)), evaluated on the fly (the activation is never materialised)
template <typename TY, typename TX = float>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const TX* __restrict__ x, TY* __restrict__ y,
                                                          uint8_t* __restrict__ idx, int N, int H, int W, int C,
                                                          int Ho, int Wo, const float* __restrict__ bn_mean,
                                                          const float* __restrict__ bn_invstd,
                                                          const float* __restrict__ bn_gamma,
                                                          const float* __restrict__ bn_beta) {
    const int cg = C >> 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * Ho * Wo * cg) return;
    const int c4 = (int)(i % cg) * 4;
    long t = i / cg;
    const int ow = (int)(t % Wo); t /= Wo;
    const int oh = (int)(t % Ho);
    const int n = (int)(t / Ho);
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {0, 0, 0, 0};
    f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = mu, gam = mu, bet = mu;
    if (bn_mean) {
        mu = *reinterpret_cast<const f32x4*>(bn_mean + c4);
        is = *reinterpret_cast<const f32x4*>(bn_invstd + c4);
        gam = *reinterpret_cast<const f32x4*>(bn_gamma + c4);
        bet = *reinterpret_cast<const f32x4*>(bn_beta + c4);
    }
    for (int r = 0; r < 3; ++r) {
        const int ih = oh * 2 - 1 + r;
        if (ih < 0 || ih >= H) continue;
        for (int s = 0; s < 3; ++s) {
            const int iw = ow * 2 - 1 + s;
            if (iw < 0 || iw >= W) continue;
            f32x4 v = ld4(x + (((long)n * H + ih) * W + iw) * C + c4);
            if (bn_mean) {
                v = bn_affine(v, mu, is, gam, bet);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (v[j] > best[j] || v[j] != v[j]) { best[j] = v[j]; bi[j] = r * 3 + s; }
        }
    }
    const long o = (((long)n * Ho + oh) * Wo + ow) * C + c4;
    st4(y + o, best);
    *reinterpret_cast<uint32_t*>(idx + o) = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy,
                                                          const uint8_t* __restrict__ idx, float* __restrict__ dx,
                                                          int N, int H, int W, int C, int Ho, int Wo) {
    const int cg = C >> 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * H * W * cg) return;
    const int c4 = (int)(i % cg) * 4;
    long t = i / cg;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int oh = h >> 1; oh <= (h + 1) >> 1; ++oh) {
        if (oh >= Ho) continue;
        const int r = h - (oh * 2 - 1);
        for (int ow = w >> 1; ow <= (w + 1) >> 1; ++ow) {
            if (ow >= Wo) continue;
            const int s = w - (ow * 2 - 1);
            const long o = (((long)n * Ho + oh) * Wo + ow) * C + c4;
            const uint32_t pk = *reinterpret_cast<const uint32_t*>(idx + o);
            const f32x4 g = *reinterpret_cast<const f32x4*>(dy + o);
            const uint32_t me = (uint32_t)(r * 3 + s);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (((pk >> (8 * j)) & 0xff) == me) acc[j] += g[j];
        }
    }
    *reinterpret_cast<f32x4*>(dx + i * 4) = acc;
}

// ------------------------------------------------------------------------------------------------
// tokens[row(f,ph,pw)][c] = dropout( mean_{k x k window} feat + pos_emb[tok][c] )
template <typename TF>
__global__ __launch_bounds__(256) void avgpool_tokens_fwd_kernel(const TF* __restrict__ feat,
                                                                 const float* __restrict__ pos_emb,
                                                                 float* __restrict__ tokens, int N, int H, int C,
                                                                 int fps, int mod_off, int T, uint32_t thr,
                                                                 float scale, uint64_t seed, uint64_t seed_off_in,
                                                                 const uint64_t* __restrict__ salt) {
    const uint64_t seed_off = seed_off_in + ((salt && thr) ? *salt : (uint64_t)0);
    const int cg = C >> 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * 64 * cg) return;
    const int c4 = (int)(i % cg) * 4;
    long t = i / cg;
    const int hw = (int)(t & 63);
    const int f = (int)(t >> 6);
    const int k = H >> 3;
    const int ph = hw >> 3, pw = hw & 7;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < k; ++r)
        for (int q = 0; q < k; ++q)
            s += ld4(feat + (((long)f * H + ph * k + r) * H + pw * k + q) * C + c4);
    s *= 1.0f / (float)(k * k);
    const long row = tok_row(f, fps, mod_off, T, hw);
    const int tok = (int)(row % T);
    s += *reinterpret_cast<const f32x4*>(pos_emb + (long)tok * C + c4);
    const long o = row * C + c4;
    if (thr) {
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] = ds6g_keep(seed, seed_off + o + j, thr) ? s[j] * scale : 0.f;
    }
    *reinterpret_cast<f32x4*>(tokens + o) = s;
}

// GPS rows: tokens[b][T-2+j][c] = dropout(emb[b][j][c] + pos_emb[T-2+j][c])
__global__ void gps_tokens_fwd_kernel(const float* __restrict__ emb, const float* __restrict__ pos_emb,
                                      float* __restrict__ tokens, int B, int C, int T, uint32_t thr, float scale,
                                      uint64_t seed, uint64_t seed_off_in, const uint64_t* __restrict__ salt) {
    const uint64_t seed_off = seed_off_in + ((salt && thr) ? *salt : (uint64_t)0);
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * 2 * C) return;
    const int c = (int)(i % C);
    const int j = (int)((i / C) & 1);
    const int b = (int)(i / (2 * C));
    const int tok = T - 2 + j;
    const long o = ((long)b * T + tok) * C + c;
    float v = emb[i] + pos_emb[(long)tok * C + c];
    if (thr) v = ds6g_keep(seed, seed_off + o, thr) ? v * scale : 0.f;
    tokens[o] = v;
}

// dst[i] = keep(i) ? src[i]*scale : 0    (dropout forward on a buffer, or its backward on a gradient)
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ src, float* __restrict__ dst, long n4,
                                                      uint32_t thr, float scale, uint64_t seed, uint64_t seed_off_in,
                                                      const uint64_t* __restrict__ salt) {
    const uint64_t seed_off = seed_off_in + ((salt && thr) ? *salt : (uint64_t)0);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 v = *reinterpret_cast<const f32x4*>(src + i * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ds6g_keep(seed, seed_off + i * 4 + j, thr) ? v[j] * scale : 0.f;
        *reinterpret_cast<f32x4*>(dst + i * 4) = v;
    }
}

// dfeat[f,h,w,c] = dfeat_in[f,h,w,c] + dtok[row(f,h/k,w/k)][c] / k^2
template <typename TF>
__global__ __launch_bounds__(256) void avgpool_tokens_bwd_kernel(const float* __restrict__ dtok,
                                                                 const TF* __restrict__ dfeat_in,
                                                                 TF* __restrict__ dfeat, int N, int H, int C,
                                                                 int fps, int mod_off, int T) {
    const int cg = C >> 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * H * H * cg) return;
    const int c4 = (int)(i % cg) * 4;
    long t = i / cg;
    const int w = (int)(t % H); t /= H;
    const int h = (int)(t % H);
    const int f = (int)(t / H);
    const int k = H >> 3;
    const long row = tok_row(f, fps, mod_off, T, (h / k) * 8 + (w / k));
    f32x4 g = *reinterpret_cast<const f32x4*>(dtok + row * C + c4) * (1.0f / (float)(k * k));
    if (dfeat_in) g += ld4(dfeat_in + i * 4);
    st4(dfeat + i * 4, g);
}

// PyTorch bilinear, align_corners=False, scale_factor given: src = max((dst+0.5)/scale - 0.5, 0)
__device__ __forceinline__ void bilin_src(int d, float inv_scale, int in_size, int& i0, int& i1, float& lam) {
    float s = ((float)d + 0.5f) * inv_scale - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    lam = s - (float)i0;
}

// out[f,h,w,c] = feat[f,h,w,c] + bilinear_up(tokmap[f])[h,w,c]; tokmap rows live in the token buffer
template <typename TF>
__global__ __launch_bounds__(256) void upsample_add_fwd_kernel(const TF* __restrict__ feat,
                                                               const float* __restrict__ tokens,
                                                               TF* __restrict__ out, int N, int H, int C, int fps,
                                                               int mod_off, int T) {
    const int cg = C >> 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * H * H * cg) return;
    const int c4 = (int)(i % cg) * 4;
    long t = i / cg;
    const int w = (int)(t % H); t /= H;
    const int h = (int)(t % H);
    const int f = (int)(t / H);
    const float inv = 8.0f / (float)H;
    int h0, h1, w0, w1;
    float lh, lw;
    bilin_src(h, inv, 8, h0, h1, lh);
    bilin_src(w, inv, 8, w0, w1, lw);
    const long base = tok_row(f, fps, mod_off, T, 0);
    const f32x4 v00 = *reinterpret_cast<const f32x4*>(tokens + (base + h0 * 8 + w0) * C + c4);
    const f32x4 v01 = *reinterpret_cast<const f32x4*>(tokens + (base + h0 * 8 + w1) * C + c4);
    const f32x4 v10 = *reinterpret_cast<const f32x4*>(tokens + (base + h1 * 8 + w0) * C + c4);
    const f32x4 v11 = *reinterpret_cast<const f32x4*>(tokens + (base + h1 * 8 + w1) * C + c4);
    const f32x4 up = (1.f - lh) * ((1.f - lw) * v00 + lw * v01) + lh * ((1.f - lw) * v10 + lw * v11);
    st4(out + i * 4, ld4(feat + i * 4) + up);
}

// dtok[row(f,ph,pw)][c] = sum_{h,w} wh(h,ph) ww(w,pw) dout[f,h,w,c]   (adjoint of the upsample).
// One workgroup per token cell (f, ph, pw): the <= 4k x 4k window of contributing pixels (k = H/8) is split over
// 256 / (C/4) row slices, the 1-D bilinear weights are tabulated once per workgroup, partial sums meet in LDS.
constexpr int UPB_MAXWIN = 64;  // window rows / columns per cell (4 * H/8, H <= 128)
template <typename TF>
__global__ __launch_bounds__(256) void upsample_add_bwd_kernel(const TF* __restrict__ dout,
                                                               float* __restrict__ dtok, int N, int H, int C, int fps,
                                                               int mod_off, int T) {
    __shared__ float whs[UPB_MAXWIN], wws[UPB_MAXWIN];
    __shared__ f32x4 part[256];
    const int cg = C >> 2;                      // float4 channel groups
    const int lanes_c = cg < 256 ? cg : 256;    // threads along channels
    const int slices = 256 / lanes_c;           // row slices of the window
    const int f = blockIdx.x >> 6, hw = blockIdx.x & 63;
    const int ph = hw >> 3, pw = hw & 7;
    const int k = H >> 3;
    const float inv = 8.0f / (float)H;
    const int hlo = max(0, (ph - 2) * k), hhi = min(H, (ph + 3) * k);
    const int wlo = max(0, (pw - 2) * k), whi = min(H, (pw + 3) * k);
    const int tid = threadIdx.x;
    if (tid < UPB_MAXWIN) {
        int i0, i1;
        float l;
        float a = 0.f, b2 = 0.f;
        if (hlo + tid < hhi) {
            bilin_src(hlo + tid, inv, 8, i0, i1, l);
            a = (i0 == ph ? 1.f - l : 0.f) + (i1 == ph ? l : 0.f);
        }
        if (wlo + tid < whi) {
            bilin_src(wlo + tid, inv, 8, i0, i1, l);
            b2 = (i0 == pw ? 1.f - l : 0.f) + (i1 == pw ? l : 0.f);
        }
        whs[tid] = a;
        wws[tid] = b2;
    }
    __syncthreads();
    const int lc = tid % lanes_c, sl = tid / lanes_c;
    for (int c0 = 0; c0 < cg; c0 += lanes_c) {  // one pass unless C > 1024
        const int c4 = (c0 + lc) * 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (sl < slices && c0 + lc < cg) {
            for (int h = hlo + sl; h < hhi; h += slices) {
                const float wh = whs[h - hlo];
                if (wh == 0.f) continue;
                const TF* row = dout + (((long)f * H + h) * H) * C + c4;
                for (int w = wlo; w < whi; ++w) {
                    const float ww = wws[w - wlo];
                    if (ww != 0.f) acc += (wh * ww) * ld4(row + (long)w * C);
                }
            }
        }
        part[tid] = acc;
        __syncthreads();
        if (sl == 0 && c0 + lc < cg) {
            for (int s2 = 1; s2 < slices; ++s2) acc += part[s2 * lanes_c + lc];
            *reinterpret_cast<f32x4*>(dtok + tok_row(f, fps, mod_off, T, hw) * C + c4) = acc;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// pooled[f][c] = mean over the 64 pixels of the 8x8 map (features.avgpool + flatten)
template <typename TF>
__global__ __launch_bounds__(256) void global_pool_kernel(const TF* __restrict__ feat, float* __restrict__ pooled,
                                                          int N, int C) {
    const int cg = C >> 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * cg) return;
    const int c4 = (int)(i % cg) * 4;
    const int f = (int)(i / cg);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < 64; ++p) s += ld4(feat + ((long)f * 64 + p) * C + c4);
    *reinterpret_cast<f32x4*>(pooled + (long)f * C + c4) = s * (1.0f / 64.0f);
}

// fused[b][c] = sum_{frames of b, 3 modalities} pooled + gps[b][0][c] + gps[b][1][c]
// gps rows are the last two token rows of sample b in the (B,T,C) buffer `tokens`
__global__ void head_sum_kernel(const float* __restrict__ p0, const float* __restrict__ p1,
                                const float* __restrict__ p2, const float* __restrict__ tokens,
                                float* __restrict__ fused, int B, int C, int fps0, int fps1, int T) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * C) return;
    const int c = (int)(i % C);
    const int b = (int)(i / C);
    float s = 0.f;
    for (int t = 0; t < fps0; ++t) s += p0[((long)b * fps0 + t) * C + c];
    for (int t = 0; t < fps1; ++t) s += p1[((long)b * fps1 + t) * C + c];
    for (int t = 0; t < fps1; ++t) s += p2[((long)b * fps1 + t) * C + c];
    s += tokens[((long)b * T + T - 2) * C + c];
    s += tokens[((long)b * T + T - 1) * C + c];
    fused[i] = s;
}

// backward of head: dfeat[f][p][c] = dfused[b][c]/64 for every frame; dtok rows: same value for
// the spatial rows (scale-1 residual add) and dfused[b][c] for the two GPS rows
template <typename TF>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dfused, TF* __restrict__ dfeat,
                                                       int N, int C, int fps) {
    const int cg = C >> 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * 64 * cg) return;
    const int c4 = (int)(i % cg) * 4;
    const int f = (int)(i / ((long)64 * cg));
    const int b = f / fps;
    st4(dfeat + i * 4, *reinterpret_cast<const f32x4*>(dfused + (long)b * C + c4) * (1.0f / 64.0f));
}

// dtok[row(f,hw)][c] = dfeat[f][hw][c] (scale-1 stage: feature and token gradients coincide)
__global__ __launch_bounds__(256) void feat_to_tok_kernel(const float* __restrict__ dfeat, float* __restrict__ dtok,
                                                          int N, int C, int fps, int mod_off, int T) {
    const int cg = C >> 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * 64 * cg) return;
    const int c4 = (int)(i % cg) * 4;
    long t = i / cg;
    const int hw = (int)(t & 63);
    const int f = (int)(t >> 6);
    *reinterpret_cast<f32x4*>(dtok + tok_row(f, fps, mod_off, T, hw) * C + c4) = *reinterpret_cast<const f32x4*>(dfeat + i * 4);
}

// rows [T-2, T) of each sample: dst[b][j][c] (+)= src rows, or the reverse scatter
__global__ void gps_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int C, int T,
                                int to_tokens, int accumulate, int src_bcast) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * 2 * C) return;
    const int c = (int)(i % C);
    const int j = (int)((i / C) & 1);
    const int b = (int)(i / (2 * C));
    const long o = ((long)b * T + T - 2 + j) * C + c;
    if (to_tokens) {
        const float v = src_bcast ? src[(long)b * C + c] : src[i];
        dst[o] = accumulate ? dst[o] + v : v;
    } else {
        dst[i] = accumulate ? dst[i] + src[o] : src[o];
    }
}

__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                    float* __restrict__ out, long n4, float alpha, float beta) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 v = alpha * *reinterpret_cast<const f32x4*>(a + i * 4);
        if (b) v += beta * *reinterpret_cast<const f32x4*>(b + i * 4);
        *reinterpret_cast<f32x4*>(out + i * 4) = v;
    }
}

// out[i] = (accumulate ? out[i] : 0) + sum_k src[k*stride + i]
__global__ void batch_sum_kernel(const float* __restrict__ src, float* __restrict__ out, long n4, int count,
                                 size_t stride, int accumulate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < count; ++k) s += *reinterpret_cast<const f32x4*>(src + (size_t)k * stride + i * 4);
    f32x4* o = reinterpret_cast<f32x4*>(out + i * 4);
    if (accumulate) s += *o;
    *o = s;
}

inline int grid1(long n) { return cdiv(n, 256); }
inline int grid_stride(long n) { return (int)min((long)8192, (n + 255) / 256); }

}  // namespace

extern "C" {

int ds6g_pack_input(const float* src, float* dst, int B, int Cs, int H, int W, int Cd, int frames_per_sample,
                    int t, int normalize_imagenet, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(src && dst && Cs >= 1 && Cs <= 4 && Cd >= Cs && Cd <= 4 && t >= 0 && t < frames_per_sample);
    DS6G_CHECK_ARG(!normalize_imagenet || Cs == 3);
    hipLaunchKernelGGL(pack_input_kernel<float>, dim3(grid1((long)B * H * W)), dim3(256), 0, (hipStream_t)stream, src, dst, B,
                       Cs, H * W, Cd, frames_per_sample, t, normalize_imagenet);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

// the same with a bf16 destination [B * frames][H][W][4] (the bf16 stem's input: csrc/stem.hip)
int ds6g_pack_input_bf16(const float* src, void* dst, int B, int Cs, int H, int W, int frames_per_sample, int t,
                         int normalize_imagenet, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(src && dst && Cs >= 1 && Cs <= 4 && t >= 0 && t < frames_per_sample);
    DS6G_CHECK_ARG(!normalize_imagenet || Cs == 3);
    hipLaunchKernelGGL(pack_input_kernel<__bf16>, dim3(grid1((long)B * H * W)), dim3(256), 0, (hipStream_t)stream, src,
                       (__bf16*)dst, B, Cs, H * W, 4, frames_per_sample, t, normalize_imagenet);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

int ds6g_pad_channels(const float* src, float* dst, long rows, int cin, int cout, int unpad, int accumulate,
                      void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(src && dst && cin <= cout);
    hipLaunchKernelGGL(pad_channels_kernel, dim3(grid1(rows * (unpad ? cin : cout))), dim3(256), 0,
                       (hipStream_t)stream, src, dst, rows, cin, cout, unpad, accumulate);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

int ds6g_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int N, int H, int W, int C, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(x && y && idx && C % 4 == 0);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL((maxpool_fwd_kernel<float>), dim3(grid1((long)N * Ho * Wo * (C / 4))), dim3(256), 0,
                       (hipStream_t)stream, x, y, idx, N, H, W, C, Ho, Wo, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

// y = maxpool3x3/2(relu(BN(x))) with the per-channel (mean, invstd) of ds6g_bn_stats / ds6g_bn_eval_prepare: the stem's
// BN -> ReLU -> MaxPool in one pass, bit-identical to ds6g_bn_apply(relu) followed by ds6g_maxpool3x3s2_fwd
int ds6g_bn_relu_maxpool3x3s2_fwd(const float* x, const float* mean, const float* invstd, const float* gamma,
                                  const float* beta, float* y, uint8_t* idx, int N, int H, int W, int C, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(x && mean && invstd && gamma && beta && y && idx && C % 4 == 0 && N > 0 && H > 0 && W > 0);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL((maxpool_fwd_kernel<float>), dim3(grid1((long)N * Ho * Wo * (C / 4))), dim3(256), 0,
                       (hipStream_t)stream, x, y, idx, N, H, W, C, Ho, Wo, mean, invstd, gamma, beta);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

int ds6g_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, float* dx, int N, int H, int W, int C, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(dy && dx && idx && C % 4 == 0);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid1((long)N * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                       dy, idx, dx, N, H, W, C, Ho, Wo);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

int ds6g_avgpool_tokens_fwd(const float* feat, const float* pos_emb, float* tokens, int N, int H, int C,
                            int frames_per_sample, int mod_off, int T, float drop_p, uint64_t seed,
                            uint64_t seed_off, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(feat && pos_emb && tokens && C % 4 == 0 && H % 8 == 0 && N % frames_per_sample == 0);
    hipLaunchKernelGGL((avgpool_tokens_fwd_kernel<float>), dim3(grid1((long)N * 64 * (C / 4))), dim3(256), 0,
                       (hipStream_t)stream, feat, pos_emb, tokens, N, H, C, frames_per_sample, mod_off, T,
                       ds6g_drop_threshold(drop_p), 1.f / (1.f - drop_p), seed, seed_off, g_ds6g_salt);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

int ds6g_gps_tokens_fwd(const float* emb, const float* pos_emb, float* tokens, int B, int C, int T, float drop_p,
                        uint64_t seed, uint64_t seed_off, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(emb && pos_emb && tokens);
    hipLaunchKernelGGL(gps_tokens_fwd_kernel, dim3(grid1((long)B * 2 * C)), dim3(256), 0, (hipStream_t)stream, emb,
                       pos_emb, tokens, B, C, T, ds6g_drop_threshold(drop_p), 1.f / (1.f - drop_p), seed, seed_off, g_ds6g_salt);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

// dst = dropout-mask(seed, seed_off + i) * src / (1-p); p == 0 is a copy
int ds6g_dropout(const float* src, float* dst, long n, float drop_p, uint64_t seed, uint64_t seed_off, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(src && dst && n % 4 == 0 && drop_p >= 0.f && drop_p < 1.f);
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_stride(n / 4)), dim3(256), 0, (hipStream_t)stream, src, dst, n / 4,
                       ds6g_drop_threshold(drop_p), 1.f / (1.f - drop_p), seed, seed_off, g_ds6g_salt);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

int ds6g_avgpool_tokens_bwd(const float* dtok, const float* dfeat_in, float* dfeat, int N, int H, int C,
                            int frames_per_sample, int mod_off, int T, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(dtok && dfeat && C % 4 == 0 && H % 8 == 0);
    hipLaunchKernelGGL((avgpool_tokens_bwd_kernel<float>), dim3(grid1((long)N * H * H * (C / 4))), dim3(256), 0,
                       (hipStream_t)stream, dtok, dfeat_in, dfeat, N, H, C, frames_per_sample, mod_off, T);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

int ds6g_upsample_add_fwd(const float* feat, const float* tokens, float* out, int N, int H, int C,
                          int frames_per_sample, int mod_off, int T, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(feat && tokens && out && C % 4 == 0 && H % 8 == 0);
    hipLaunchKernelGGL((upsample_add_fwd_kernel<float>), dim3(grid1((long)N * H * H * (C / 4))), dim3(256), 0,
                       (hipStream_t)stream, feat, tokens, out, N, H, C, frames_per_sample, mod_off, T);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

int ds6g_upsample_add_bwd(const float* dout, float* dtok, int N, int H, int C, int frames_per_sample, int mod_off,
                          int T, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(dout && dtok && C % 4 == 0 && H % 8 == 0 && 5 * (H / 8) <= UPB_MAXWIN && N > 0);
    hipLaunchKernelGGL((upsample_add_bwd_kernel<float>), dim3(N * 64), dim3(256), 0, (hipStream_t)stream, dout, dtok, N, H, C,
                       frames_per_sample, mod_off, T);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

int ds6g_global_pool(const float* feat, float* pooled, int N, int C, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(feat && pooled && C % 4 == 0);
    hipLaunchKernelGGL((global_pool_kernel<float>), dim3(grid1((long)N * (C / 4))), dim3(256), 0, (hipStream_t)stream, feat,
                       pooled, N, C);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

int ds6g_head_sum(const float* pooled_img, const float* pooled_lidar, const float* pooled_radar, const float* tokens,
                  float* fused, int B, int C, int fps_img, int fps_other, int T, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(pooled_img && pooled_lidar && pooled_radar && tokens && fused);
    hipLaunchKernelGGL(head_sum_kernel, dim3(grid1((long)B * C)), dim3(256), 0, (hipStream_t)stream, pooled_img,
                       pooled_lidar, pooled_radar, tokens, fused, B, C, fps_img, fps_other, T);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

int ds6g_head_bwd(const float* dfused, float* dfeat, int N, int C, int frames_per_sample, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(dfused && dfeat && C % 4 == 0);
    hipLaunchKernelGGL((head_bwd_kernel<float>), dim3(grid1((long)N * 64 * (C / 4))), dim3(256), 0, (hipStream_t)stream, dfused,
                       dfeat, N, C, frames_per_sample);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

int ds6g_feat_to_tokens(const float* dfeat, float* dtok, int N, int C, int frames_per_sample, int mod_off, int T,
                        void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(dfeat && dtok && C % 4 == 0);
    hipLaunchKernelGGL(feat_to_tok_kernel, dim3(grid1((long)N * 64 * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                       dfeat, dtok, N, C, frames_per_sample, mod_off, T);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

// to_tokens=1: tokens[b][T-2+j][:] (+)= src[b][j][:] (src_bcast: src[b][:] for both j);
// to_tokens=0: dst[b][j][:] (+)= tokens rows
int ds6g_gps_rows(const float* src, float* dst, int B, int C, int T, int to_tokens, int accumulate, int src_bcast,
                  void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(src && dst);
    hipLaunchKernelGGL(gps_rows_kernel, dim3(grid1((long)B * 2 * C)), dim3(256), 0, (hipStream_t)stream, src, dst, B, C,
                       T, to_tokens, accumulate, src_bcast);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

// out = alpha*a + beta*b   (b nullable)
int ds6g_axpby(const float* a, const float* b, float* out, long n, float alpha, float beta, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(a && out && n % 4 == 0);
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_stride(n / 4)), dim3(256), 0, (hipStream_t)stream, a, b, out, n / 4,
                       alpha, beta);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

// out[i] (+)= sum_{k<count} src[k*stride + i], i < n   (pos_emb gradient: sum over the batch)
int ds6g_batch_sum(const float* src, float* out, long n, int count, long stride, int accumulate, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(src && out && n % 4 == 0 && stride % 4 == 0);
    hipLaunchKernelGGL(batch_sum_kernel, dim3(grid1(n / 4)), dim3(256), 0, (hipStream_t)stream, src, out, n / 4, count,
                       (size_t)stride, accumulate);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

// ---- bf16-storage path: the same kernels on bf16 feature maps (tokens, pos_emb, pooled vectors and all arithmetic fp32) ----
int ds6g_bn_relu_maxpool3x3s2_fwd_bf16out(const float* x, const float* mean, const float* invstd, const float* gamma,
                                          const float* beta, void* y, uint8_t* idx, int N, int H, int W, int C,
                                          void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(x && mean && invstd && gamma && beta && y && idx && C % 4 == 0 && N > 0 && H > 0 && W > 0);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL((maxpool_fwd_kernel<__bf16>), dim3(grid1((long)N * Ho * Wo * (C / 4))), dim3(256), 0,
                       (hipStream_t)stream, x, (__bf16*)y, idx, N, H, W, C, Ho, Wo, mean, invstd, gamma, beta);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

// bf16 stem (csrc/stem.hip): the conv output x is bf16 as well
int ds6g_bf16_stem_bn_relu_maxpool_fwd(const void* x, const float* mean, const float* invstd, const float* gamma,
                                       const float* beta, void* y, uint8_t* idx, int N, int H, int W, int C, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(x && mean && invstd && gamma && beta && y && idx && C % 4 == 0 && N > 0 && H > 0 && W > 0);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL((maxpool_fwd_kernel<__bf16, __bf16>), dim3(grid1((long)N * Ho * Wo * (C / 4))), dim3(256), 0,
                       (hipStream_t)stream, (const __bf16*)x, (__bf16*)y, idx, N, H, W, C, Ho, Wo, mean, invstd, gamma, beta);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

int ds6g_bf16_avgpool_tokens_fwd(const void* feat, const float* pos_emb, float* tokens, int N, int H, int C,
                                 int frames_per_sample, int mod_off, int T, float drop_p, uint64_t seed,
                                 uint64_t seed_off, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(feat && pos_emb && tokens && C % 4 == 0 && H % 8 == 0 && N % frames_per_sample == 0);
    hipLaunchKernelGGL((avgpool_tokens_fwd_kernel<__bf16>), dim3(grid1((long)N * 64 * (C / 4))), dim3(256), 0,
                       (hipStream_t)stream, (const __bf16*)feat, pos_emb, tokens, N, H, C, frames_per_sample, mod_off, T,
                       ds6g_drop_threshold(drop_p), 1.f / (1.f - drop_p), seed, seed_off, g_ds6g_salt);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

int ds6g_bf16_avgpool_tokens_bwd(const float* dtok, const void* dfeat_in, void* dfeat, int N, int H, int C,
                                 int frames_per_sample, int mod_off, int T, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(dtok && dfeat && C % 4 == 0 && H % 8 == 0);
    hipLaunchKernelGGL((avgpool_tokens_bwd_kernel<__bf16>), dim3(grid1((long)N * H * H * (C / 4))), dim3(256), 0,
                       (hipStream_t)stream, dtok, (const __bf16*)dfeat_in, (__bf16*)dfeat, N, H, C, frames_per_sample, mod_off, T);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

int ds6g_bf16_upsample_add_fwd(const void* feat, const float* tokens, void* out, int N, int H, int C,
                               int frames_per_sample, int mod_off, int T, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(feat && tokens && out && C % 4 == 0 && H % 8 == 0);
    hipLaunchKernelGGL((upsample_add_fwd_kernel<__bf16>), dim3(grid1((long)N * H * H * (C / 4))), dim3(256), 0,
                       (hipStream_t)stream, (const __bf16*)feat, tokens, (__bf16*)out, N, H, C, frames_per_sample, mod_off, T);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

int ds6g_bf16_upsample_add_bwd(const void* dout, float* dtok, int N, int H, int C, int frames_per_sample, int mod_off,
                               int T, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(dout && dtok && C % 4 == 0 && H % 8 == 0 && 5 * (H / 8) <= UPB_MAXWIN && N > 0);
    hipLaunchKernelGGL((upsample_add_bwd_kernel<__bf16>), dim3(N * 64), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)dout, dtok, N, H, C, frames_per_sample, mod_off, T);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

int ds6g_bf16_global_pool(const void* feat, float* pooled, int N, int C, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(feat && pooled && C % 4 == 0);
    hipLaunchKernelGGL((global_pool_kernel<__bf16>), dim3(grid1((long)N * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)feat, pooled, N, C);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

int ds6g_bf16_head_bwd(const float* dfused, void* dfeat, int N, int C, int frames_per_sample, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(dfused && dfeat && C % 4 == 0);
    hipLaunchKernelGGL((head_bwd_kernel<__bf16>), dim3(grid1((long)N * 64 * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                       dfused, (__bf16*)dfeat, N, C, frames_per_sample);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

}  // extern "C"
