// Implicit-GEMM convolution / linear kernels on the exact-fp32 matrix pipe of gfx950
// (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate).  One kernel template covers
//   FWD   : y[m][o]      = sum_{r,s,c} x[n, oh*st-p+r, ow*st-p+s, c] * w[o][r][s][c]
//   DGRAD : dx[m][c]     = sum_{r,s,o} dy[n, (h+p-r)/st, (w+p-s)/st, o] * w[o][r][s][c]
//   WGRAD : dw[o][r,s,c] = sum_{pixels} dy[pixel][o] * x[n, oh*st-p+r, ow*st-p+s, c]   (split-K)
// on NHWC activations and OHWI weights; a Linear layer is the 1x1 case over an M x 1 x 1 "image".
// Covers the reference's Conv2d / Linear call sites: model2_seq.py:495-512,528-530,546-548,
// 565-567 (ResNet trunks), :83-90,:97-99,:109,:121-126 (GPT linears), :422-425,:863-869.
//
// Tiling: 256 threads = 4 waves (2x2); block tile BM x BN, BK = 16; operands are staged in LDS
// k-major ([k][row]) so that a wave's MFMA operand read (lane l -> row l&31, k = l>>5) is a
// conflict-free ds_read_b32 of 32 consecutive floats; global loads are 16-B per lane and
// register-prefetched one k-tile ahead (single barrier per k-tile, two LDS buffers).
#include "common.h"

namespace {

enum { MODE_FWD = 0, MODE_DGRAD = 1, MODE_WGRAD = 2 };

struct IgemmParams {
    const float* a_src;
    const float* b_src;
    float* out;
    int N, H, W, C;      // input-side tensor (x / dx), NHWC
    int Ho, Wo, K;       // output-side tensor (y / dy), NHWC, K = out channels
    int R, S, stride, pad;
    int Mg, Ng, Kg;      // GEMM extents
    const float* bias;
    const float* residual;
    const float* mask_src;
    int relu;
    int accumulate;
    uint32_t drop_thr;
    float drop_scale;
    uint64_t seed;
    uint64_t seed_off;
    int k_per_split;     // multiple of 16
    size_t split_stride; // elements between split-K slabs
    int tiles_n;
};

constexpr int BK = 16;

template <int MODE, int BM, int BN>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmParams p) {
    constexpr int LDA = BM + 4;
    constexpr int LDB = BN + 4;
    constexpr int TM = BM / 64;  // 32x32 MFMA tiles per wave along M (wave grid 2x2)
    constexpr int TN = BN / 64;
    constexpr int A_LD = BM * 4 / 256;  // float4 loads per thread per k-tile
    constexpr int B_LD = BN * 4 / 256;
    __shared__ __attribute__((aligned(16))) float As[2][BK][LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDB];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile_m = blockIdx.x / p.tiles_n;
    const int tile_n = blockIdx.x - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int split = blockIdx.z;
    const int kbegin = (MODE == MODE_WGRAD) ? split * p.k_per_split : 0;
    const int kend = (MODE == MODE_WGRAD) ? min(p.Kg, kbegin + p.k_per_split) : p.Kg;
    const int nk = (kend - kbegin + BK - 1) / BK;

    // ---- per-thread, k-independent load coordinates ------------------------------------------
    // A operand
    int a_n[A_LD], a_y[A_LD], a_x[A_LD];
    bool a_ok[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int idx = tid + i * 256;
        if (MODE == MODE_FWD) {
            const int m = m0 + (idx >> 2);
            a_ok[i] = m < p.Mg;
            const int mm = a_ok[i] ? m : 0;
            const int ow = mm % p.Wo;
            const int t = mm / p.Wo;
            const int oh = t % p.Ho;
            a_n[i] = t / p.Ho;
            a_y[i] = oh * p.stride - p.pad;
            a_x[i] = ow * p.stride - p.pad;
        } else if (MODE == MODE_DGRAD) {
            const int m = m0 + (idx >> 2);
            a_ok[i] = m < p.Mg;
            const int mm = a_ok[i] ? m : 0;
            const int w_ = mm % p.W;
            const int t = mm / p.W;
            const int h_ = t % p.H;
            a_n[i] = t / p.H;
            a_y[i] = h_ + p.pad;
            a_x[i] = w_ + p.pad;
        } else {
            const int mc = idx % (BM / 4);
            a_ok[i] = (m0 + mc * 4) < p.Mg;
            a_n[i] = 0; a_y[i] = 0; a_x[i] = 0;
        }
    }
    // B operand (WGRAD: fixed (r,s,c) per thread)
    int b_r[B_LD], b_s[B_LD], b_c[B_LD];
    bool b_ok[B_LD];
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
        const int idx = tid + i * 256;
        if (MODE == MODE_WGRAD) {
            const int ncol = n0 + (idx % (BN / 4)) * 4;
            b_ok[i] = ncol < p.Ng;
            const int nn = b_ok[i] ? ncol : 0;
            const int tap = nn / p.C;
            b_c[i] = nn - tap * p.C;
            b_r[i] = tap / p.S;
            b_s[i] = tap - b_r[i] * p.S;
        } else if (MODE == MODE_FWD) {
            b_ok[i] = (n0 + (idx >> 2)) < p.Ng;
            b_r[i] = b_s[i] = b_c[i] = 0;
        } else {
            b_ok[i] = (n0 + (idx % (BN / 4)) * 4) < p.Ng;
            b_r[i] = b_s[i] = b_c[i] = 0;
        }
    }

    f32x4 a_reg[A_LD], b_reg[B_LD];

    auto load_tiles = [&](int kt) {
        const int kb = kbegin + kt * BK;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int idx = tid + i * 256;
            const float* ptr = nullptr;
            bool ok = a_ok[i];
            if (MODE == MODE_FWD) {
                const int kg = kb + (idx & 3) * 4;
                const int tap = kg / p.C;
                const int c = kg - tap * p.C;
                const int r = tap / p.S;
                const int s = tap - r * p.S;
                const int ih = a_y[i] + r, iw = a_x[i] + s;
                ok = ok && kg < kend && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
                ptr = p.a_src + (((long)a_n[i] * p.H + ih) * p.W + iw) * p.C + c;
            } else if (MODE == MODE_DGRAD) {
                const int kg = kb + (idx & 3) * 4;
                const int tap = kg / p.K;
                const int o = kg - tap * p.K;
                const int r = tap / p.S;
                const int s = tap - r * p.S;
                const int th = a_y[i] - r, tw = a_x[i] - s;
                int oh = th, ow = tw;
                bool okk = th >= 0 && tw >= 0;
                if (p.stride != 1) {
                    oh = th / p.stride;
                    ow = tw / p.stride;
                    okk = okk && (oh * p.stride == th) && (ow * p.stride == tw);
                }
                ok = ok && okk && kg < kend && oh < p.Ho && ow < p.Wo;
                ptr = p.a_src + (((long)a_n[i] * p.Ho + oh) * p.Wo + ow) * p.K + o;
            } else {
                const int krow = idx / (BM / 4);
                const int mc = idx % (BM / 4);
                const int pix = kb + krow;
                ok = ok && pix < kend;
                ptr = p.a_src + (long)pix * p.K + m0 + mc * 4;
            }
            a_reg[i] = ok ? *reinterpret_cast<const f32x4*>(ptr) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const int idx = tid + i * 256;
            const float* ptr = nullptr;
            bool ok = b_ok[i];
            if (MODE == MODE_FWD) {
                const int kg = kb + (idx & 3) * 4;
                ok = ok && kg < kend;
                ptr = p.b_src + (long)(n0 + (idx >> 2)) * p.Kg + kg;
            } else if (MODE == MODE_DGRAD) {
                const int krow = idx / (BN / 4);
                const int nc = idx % (BN / 4);
                const int kg = kb + krow;
                const int tap = kg / p.K;
                const int o = kg - tap * p.K;
                ok = ok && kg < kend;
                ptr = p.b_src + ((long)o * (p.R * p.S) + tap) * p.C + n0 + nc * 4;
            } else {
                const int krow = idx / (BN / 4);
                const int pix = kb + krow;
                const int pp = pix < kend ? pix : 0;
                const int ow = pp % p.Wo;
                const int t = pp / p.Wo;
                const int oh = t % p.Ho;
                const int n = t / p.Ho;
                const int ih = oh * p.stride - p.pad + b_r[i];
                const int iw = ow * p.stride - p.pad + b_s[i];
                ok = ok && pix < kend && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
                ptr = p.b_src + (((long)n * p.H + ih) * p.W + iw) * p.C + b_c[i];
            }
            b_reg[i] = ok ? *reinterpret_cast<const f32x4*>(ptr) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };

    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int idx = tid + i * 256;
            if (MODE == MODE_WGRAD) {
                const int krow = idx / (BM / 4);
                const int mc = idx % (BM / 4);
                *reinterpret_cast<f32x4*>(&As[buf][krow][mc * 4]) = a_reg[i];
            } else {
                const int row = idx >> 2, kc = (idx & 3) * 4;
                As[buf][kc + 0][row] = a_reg[i][0];
                As[buf][kc + 1][row] = a_reg[i][1];
                As[buf][kc + 2][row] = a_reg[i][2];
                As[buf][kc + 3][row] = a_reg[i][3];
            }
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const int idx = tid + i * 256;
            if (MODE == MODE_FWD) {
                const int row = idx >> 2, kc = (idx & 3) * 4;
                Bs[buf][kc + 0][row] = b_reg[i][0];
                Bs[buf][kc + 1][row] = b_reg[i][1];
                Bs[buf][kc + 2][row] = b_reg[i][2];
                Bs[buf][kc + 3][row] = b_reg[i][3];
            } else {
                const int krow = idx / (BN / 4);
                const int nc = idx % (BN / 4);
                *reinterpret_cast<f32x4*>(&Bs[buf][krow][nc * 4]) = b_reg[i];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_off = wm * (BM / 2) + (lane & 31);
    const int b_off = wn * (BN / 2) + (lane & 31);
    const int khalf = lane >> 5;

    if (nk > 0) {
        load_tiles(0);
        store_tiles(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tiles(kt + 1);
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[buf][kk * 2 + khalf][a_off + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[buf][kk * 2 + khalf][b_off + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: D[row][col], col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ---------
    float* outp = p.out + ((MODE == MODE_WGRAD) ? (size_t)split * p.split_stride : (size_t)0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
            if (col >= p.Ng) continue;
            const float bias = (MODE != MODE_WGRAD && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (row >= p.Mg) continue;
                const size_t o = (size_t)row * p.Ng + col;
                float v = acc[i][j][r];
                if (MODE != MODE_WGRAD) {
                    v += bias;
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (p.mask_src) v = (p.mask_src[o] > 0.f) ? v : 0.f;
                    if (p.drop_thr) v = ds6g_keep(p.seed, p.seed_off + o, p.drop_thr) ? v * p.drop_scale : 0.f;
                    if (p.residual) v += p.residual[o];
                    if (p.accumulate) v += outp[o];
                }
                outp[o] = v;
            }
        }
    }
}

// out[i] = (accumulate ? out[i] : 0) + sum_s part[s*stride + i]
__global__ void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, long n,
                                     int splits, size_t stride, int accumulate) {
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < splits; ++k) s += *reinterpret_cast<const f32x4*>(part + (size_t)k * stride + i);
    f32x4* o = reinterpret_cast<f32x4*>(out + i);
    if (accumulate) s += *o;
    *o = s;
}

// last launched variant, for the bench's live per-kernel timing: mode * 10 + {0: 128x128, 1: 128x64, 2: 64x64}
int g_last_variant = -1;

template <int MODE>
int launch_igemm(IgemmParams& p, int splits, hipStream_t st) {
    const long t128 = (long)cdiv(p.Mg, 128) * cdiv(p.Ng, 128) * splits;
    const long t12864 = (long)cdiv(p.Mg, 128) * cdiv(p.Ng, 64) * splits;
    dim3 block(256);
    if (p.Ng > 64 && p.Mg > 64 && t128 >= 384) {
        p.tiles_n = cdiv(p.Ng, 128);
        dim3 grid(cdiv(p.Mg, 128) * p.tiles_n, 1, splits);
        hipLaunchKernelGGL((igemm_kernel<MODE, 128, 128>), grid, block, 0, st, p);
        g_last_variant = MODE * 10 + 0;
    } else if (p.Mg > 64 && t12864 >= 384) {
        p.tiles_n = cdiv(p.Ng, 64);
        dim3 grid(cdiv(p.Mg, 128) * p.tiles_n, 1, splits);
        hipLaunchKernelGGL((igemm_kernel<MODE, 128, 64>), grid, block, 0, st, p);
        g_last_variant = MODE * 10 + 1;
    } else {
        p.tiles_n = cdiv(p.Ng, 64);
        dim3 grid(cdiv(p.Mg, 64) * p.tiles_n, 1, splits);
        hipLaunchKernelGGL((igemm_kernel<MODE, 64, 64>), grid, block, 0, st, p);
        g_last_variant = MODE * 10 + 2;
    }
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

int run_wgrad(IgemmParams& p, float* dw, int accumulate, float* ws, size_t ws_bytes, hipStream_t st) {
    const long out_elems = (long)p.Mg * p.Ng;
    DS6G_CHECK_ARG(out_elems % 4 == 0);
    // enough independent blocks to fill 256 CUs a few times over, bounded by the workspace
    const long tiles = (long)cdiv(p.Mg, 64) * cdiv(p.Ng, 64);
    long splits = (1024 + tiles - 1) / tiles;
    const long max_by_k = (p.Kg + 4 * BK - 1) / (4 * BK);
    if (splits > max_by_k) splits = max_by_k;
    const long max_by_ws = (long)(ws_bytes / (out_elems * sizeof(float)));
    if (splits > max_by_ws) splits = max_by_ws;
    if (splits < 1) splits = 1;
    int kps = cdiv(cdiv(p.Kg, splits), BK) * BK;
    splits = cdiv(p.Kg, kps);
    p.k_per_split = kps;
    p.split_stride = (size_t)out_elems;
    if (splits == 1 && !accumulate) {
        p.out = dw;
        return launch_igemm<MODE_WGRAD>(p, 1, st);
    }
    DS6G_CHECK_ARG(ws != nullptr && (size_t)splits * out_elems * sizeof(float) <= ws_bytes);
    p.out = ws;
    int rc = launch_igemm<MODE_WGRAD>(p, (int)splits, st);
    if (rc) return rc;
    const int thr = 256;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(out_elems / 4, thr)), dim3(thr), 0, st, ws, dw, out_elems,
                       (int)splits, (size_t)out_elems, accumulate);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

void fill_conv(IgemmParams& p, int N, int H, int W, int C, int K, int R, int S, int stride, int pad) {
    p = IgemmParams{};
    p.N = N; p.H = H; p.W = W; p.C = C; p.K = K; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
    p.Ho = (H + 2 * pad - R) / stride + 1;
    p.Wo = (W + 2 * pad - S) / stride + 1;
    p.drop_scale = 1.f;
}

}  // namespace

extern "C" {

int ds6g_last_igemm_variant(void) { return g_last_variant; }

int ds6g_conv2d_fwd(const float* x, const float* w, float* y, int N, int H, int W, int C, int K, int R,
                    int S, int stride, int pad, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(x && w && y && C % 4 == 0 && N > 0);
    IgemmParams p;
    fill_conv(p, N, H, W, C, K, R, S, stride, pad);
    p.a_src = x; p.b_src = w; p.out = y;
    p.Mg = N * p.Ho * p.Wo; p.Ng = K; p.Kg = R * S * C;
    return launch_igemm<MODE_FWD>(p, 1, (hipStream_t)stream);
}

int ds6g_conv2d_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int K, int R,
                      int S, int stride, int pad, int accumulate, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(dy && w && dx && C % 4 == 0 && K % 4 == 0);
    IgemmParams p;
    fill_conv(p, N, H, W, C, K, R, S, stride, pad);
    p.a_src = dy; p.b_src = w; p.out = dx; p.accumulate = accumulate;
    p.Mg = N * H * W; p.Ng = C; p.Kg = R * S * K;
    return launch_igemm<MODE_DGRAD>(p, 1, (hipStream_t)stream);
}

int ds6g_conv2d_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K, int R,
                      int S, int stride, int pad, int accumulate, float* ws, size_t ws_bytes, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(x && dy && dw && C % 4 == 0 && K % 4 == 0);
    IgemmParams p;
    fill_conv(p, N, H, W, C, K, R, S, stride, pad);
    p.a_src = dy; p.b_src = x;
    p.Mg = K; p.Ng = R * S * C; p.Kg = N * p.Ho * p.Wo;
    return run_wgrad(p, dw, accumulate, ws, ws_bytes, (hipStream_t)stream);
}

// y[M][N] = residual + dropout( act( x[M][K] @ w[N][K]^T + bias ) )
int ds6g_linear_fwd(const float* x, const float* w, const float* bias, float* y, int M, int N, int K, int relu,
                    const float* residual, float drop_p, uint64_t seed, uint64_t seed_off, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(x && w && y && K % 4 == 0 && M > 0 && drop_p >= 0.f && drop_p < 1.f);
    IgemmParams p;
    fill_conv(p, M, 1, 1, K, N, 1, 1, 1, 0);
    p.a_src = x; p.b_src = w; p.out = y; p.bias = bias; p.relu = relu; p.residual = residual;
    p.drop_thr = ds6g_drop_threshold(drop_p);
    p.drop_scale = 1.f / (1.f - drop_p);
    p.seed = seed; p.seed_off = seed_off;
    p.Mg = M; p.Ng = N; p.Kg = K;
    return launch_igemm<MODE_FWD>(p, 1, (hipStream_t)stream);
}

// dx[M][K] (+)= (dy[M][N] @ w[N][K]) * (mask_src > 0)
int ds6g_linear_dgrad(const float* dy, const float* w, float* dx, int M, int N, int K, const float* mask_src,
                      int accumulate, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(dy && w && dx && K % 4 == 0 && N % 4 == 0);
    IgemmParams p;
    fill_conv(p, M, 1, 1, K, N, 1, 1, 1, 0);
    p.a_src = dy; p.b_src = w; p.out = dx; p.mask_src = mask_src; p.accumulate = accumulate;
    p.Mg = M; p.Ng = K; p.Kg = N;
    return launch_igemm<MODE_DGRAD>(p, 1, (hipStream_t)stream);
}

// dw[N][K] (+)= dy[M][N]^T @ x[M][K]
int ds6g_linear_wgrad(const float* x, const float* dy, float* dw, int M, int N, int K, int accumulate, float* ws,
                      size_t ws_bytes, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(x && dy && dw && K % 4 == 0 && N % 4 == 0);
    IgemmParams p;
    fill_conv(p, M, 1, 1, K, N, 1, 1, 1, 0);
    p.a_src = dy; p.b_src = x;
    p.Mg = N; p.Ng = K; p.Kg = M;
    return run_wgrad(p, dw, accumulate, ws, ws_bytes, (hipStream_t)stream);
}

}  // extern "C"
