// Train-mode BatchNorm2d (+ReLU, +residual) and LayerNorm, forward and backward, on fp32
// channels-last rows ([M][C], C contiguous).  HBM-bound kernels: 16-B accesses per lane,
// per-thread fp32 partials over <=64 rows, fp64 above that (no E[x^2]-E[x]^2 cancellation
// problem and no float atomics: results are bitwise reproducible).
// Reference call sites: torchvision BasicBlock BN via model2_seq.py:496,501,506,510-512,
// 528-530,546-548,565-567 (train mode: batch mean / biased var, eps 1e-5, momentum .1 with
// unbiased running_var); LayerNorm model2_seq.py:118-119,131-132,199,274 (eps 1e-5).
#include "common.h"

namespace {

constexpr int BN_ROWS_PER_THREAD = 16;
// BatchNorm finalize kernels (152 launches per training step, each on the critical path of its trunk stream and
// latency-bound: one dependent chain of partial reads per thread): 2 columns x 128 partial-groups per block, so C / 2
// blocks each walk nblk / 128 partials (8 blocks x nblk / 32 before: 8.6 us per launch at C = 64)
constexpr int BNF_COLS = 2;
constexpr int BNF_GROUPS = 128;

// sum over the BNF_GROUPS partial-groups of a 256-thread block (thread = g * BNF_COLS + cl): lanes of one wave that share
// cl are reduced by shuffles, the four wave results meet in LDS; valid in the threads with g == 0
__device__ __forceinline__ void bnf_reduce2(double& a, double& b, double (*red)[4][BNF_COLS]) {
#pragma unroll
    for (int o = BNF_COLS; o < 64; o <<= 1) {
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < BNF_COLS) {
        red[0][wave][lane] = a;
        red[1][wave][lane] = b;
    }
    __syncthreads();
    if (threadIdx.x < BNF_COLS) {
        a = red[0][0][threadIdx.x] + red[0][1][threadIdx.x] + red[0][2][threadIdx.x] + red[0][3][threadIdx.x];
        b = red[1][0][threadIdx.x] + red[1][1][threadIdx.x] + red[1][2][threadIdx.x] + red[1][3][threadIdx.x];
    }
}

// partial[blk][0][c] = sum_rows a(r,c), partial[blk][1][c] = sum_rows b(r,c) in fp64.
// MODE 0: a = x, b = x*x.   MODE 1: a = dyeff, b = dyeff * xhat  (BN backward)
template <int MODE, typename TX, typename TA>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const TX* __restrict__ x, const TA* __restrict__ dy,
                                                        const TA* __restrict__ y_mask,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, long M, int C,
                                                        int rows_per_block, double* __restrict__ partial,
                                                        const float* __restrict__ mg, const float* __restrict__ mb,
                                                        const PoolGrad pg) {
    extern __shared__ __attribute__((aligned(16))) double red[];  // [rowlanes][2][C]
    const int cg = C >> 2;
    const int colg = threadIdx.x % cg;
    const int rowlane = threadIdx.x / cg;
    const int rowlanes = 256 / cg;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(M, r0 + rows_per_block);
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = {1.f, 1.f, 1.f, 1.f};
    f32x4 gam = {1.f, 1.f, 1.f, 1.f}, bet = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 1) {
        mu = *reinterpret_cast<const f32x4*>(mean + colg * 4);
        is = *reinterpret_cast<const f32x4*>(invstd + colg * 4);
        if (mg) {  // ReLU mask recomputed from x (no residual on this BN): sign of gamma * xhat + beta
            gam = *reinterpret_cast<const f32x4*>(mg + colg * 4);
            bet = *reinterpret_cast<const f32x4*>(mb + colg * 4);
        }
    }
    for (long r = r0 + rowlane; r < r1; r += rowlanes) {
        const size_t o = (size_t)r * C + colg * 4;
        const f32x4 xv = ld4(x + o);
        if (MODE == 0) {
            s1 += xv;
            s2 += xv * xv;
        } else {
            f32x4 g = pg.dp ? pooled_grad(pg, r, colg * 4, C) : ld4(dy + o);
            if (y_mask || mg) {
                const f32x4 yv = mg ? bn_affine(xv, mu, is, gam, bet) : ld4(y_mask + o);
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = yv[j] > 0.f ? g[j] : 0.f;
            }
            s1 += g;
            s2 += g * ((xv - mu) * is);
        }
    }
    double* mine = red + ((size_t)rowlane * 2) * C + colg * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mine[j] = (double)s1[j];
        mine[C + j] = (double)s2[j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        double acc = 0.0;
        for (int rl = 0; rl < rowlanes; ++rl) acc += red[(size_t)rl * 2 * C + i];
        partial[(size_t)blockIdx.x * 2 * C + i] = acc;
    }
}

// training statistics: mean / invstd from partials, running-stat update
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const double* __restrict__ partial, int nblk, long M,
                                                                int C, float eps, float momentum,
                                                                float* __restrict__ mean, float* __restrict__ invstd,
                                                                float* __restrict__ running_mean,
                                                                float* __restrict__ running_var) {
    __shared__ double red[2][4][BNF_COLS];
    const int cl = threadIdx.x & (BNF_COLS - 1), g = threadIdx.x / BNF_COLS;
    const int c = blockIdx.x * BNF_COLS + cl;
    double s = 0.0, q = 0.0;
    if (c < C)
        for (int b = g; b < nblk; b += BNF_GROUPS) {
            s += partial[(size_t)b * 2 * C + c];
            q += partial[(size_t)b * 2 * C + C + c];
        }
    bnf_reduce2(s, q, red);
    if (g != 0 || c >= C) return;
    const double mu = s / (double)M;
    double var = q / (double)M - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mu);
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
    }
}

__global__ void bn_eval_prepare_kernel(const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                       int C, float eps, float* __restrict__ mean, float* __restrict__ invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean[c] = running_mean[c];
    invstd[c] = 1.0f / sqrtf(running_var[c] + eps);
}

// y = relu?( (x - mean) * invstd * gamma + beta + residual? )
template <typename TA>
__global__ __launch_bounds__(256) void bn_apply_kernel(const TA* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       const TA* __restrict__ residual, TA* __restrict__ y,
                                                       long total4, int C, int relu) {
    const int cg = C >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % cg) * 4;
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c4);
        const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c4);
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c4);
        f32x4 v = ld4(x + i * 4);
        v = bn_affine(v, mu, is, g, b);
        if (residual) v += ld4(residual + i * 4);
        if (relu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        st4(y + i * 4, v);
    }
}

// dgamma/dbeta and the per-channel coefficients of dx = a * (dyeff - b - xhat * cc)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ partial, int nblk, long M,
                                                              int C, const float* __restrict__ gamma,
                                                              const float* __restrict__ invstd,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ coef, int accumulate) {
    __shared__ double red[2][4][BNF_COLS];
    const int cl = threadIdx.x & (BNF_COLS - 1), g = threadIdx.x / BNF_COLS;
    const int c = blockIdx.x * BNF_COLS + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
        for (int b = g; b < nblk; b += BNF_GROUPS) {
            s1 += partial[(size_t)b * 2 * C + c];
            s2 += partial[(size_t)b * 2 * C + C + c];
        }
    bnf_reduce2(s1, s2, red);
    if (g != 0 || c >= C) return;
    if (accumulate) {
        dgamma[c] += (float)s2;
        dbeta[c] += (float)s1;
    } else {
        dgamma[c] = (float)s2;
        dbeta[c] = (float)s1;
    }
    coef[c] = gamma[c] * invstd[c];
    coef[C + c] = (float)(s1 / (double)M);
    coef[2 * C + c] = (float)(s2 / (double)M);
}

template <typename TX, typename TA>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const TA* __restrict__ dy,
                                                           const TA* __restrict__ y_mask,
                                                           const TX* __restrict__ x,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ coef, TX* __restrict__ dx,
                                                           TA* __restrict__ dres, long total4, int C,
                                                           const float* __restrict__ mg, const float* __restrict__ mb,
                                                           const PoolGrad pg) {
    const int cg = C >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % cg) * 4;
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c4);
        const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c4);
        const f32x4 a = *reinterpret_cast<const f32x4*>(coef + c4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(coef + C + c4);
        const f32x4 cc = *reinterpret_cast<const f32x4*>(coef + 2 * C + c4);
        f32x4 g = pg.dp ? pooled_grad(pg, i / cg, c4, C) : ld4(dy + i * 4);
        const f32x4 xv = ld4(x + i * 4);
        if (y_mask || mg) {
            const f32x4 yv = mg ? bn_affine(xv, mu, is, *reinterpret_cast<const f32x4*>(mg + c4),
                                            *reinterpret_cast<const f32x4*>(mb + c4))
                                : ld4(y_mask + i * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = yv[j] > 0.f ? g[j] : 0.f;
        }
        const f32x4 xh = (xv - mu) * is;
        if (dres) st4(dres + i * 4, g);
        st4(dx + i * 4, a * (g - b - xh * cc));
    }
}

int bn_geometry(long M, int C, int* rows_per_block, int* nblk) {
    const int cg = C / 4;
    if (C % 4 != 0 || cg > 256 || 256 % cg != 0) return DS6G_ERR_ARG;
    *rows_per_block = BN_ROWS_PER_THREAD * (256 / cg);
    *nblk = cdiv(M, *rows_per_block);
    return DS6G_OK;
}

// ------------------------------------ LayerNorm ----------------------------------------------
// one wave per row; a lane owns columns 4*lane + 256*j + {0..3} (16-B accesses) when C >= 256, else the
// scalar columns lane + 64*j.  C in {64,128,256,512}.
typedef __bf16 bf16x4_ __attribute__((ext_vector_type(4)));

template <int C>
struct LnRow {
    static constexpr bool VEC = C >= 256;
    static constexpr int NV = VEC ? C / 256 : C / 64;  // float4 (VEC) or float (scalar) items per lane
    static constexpr int NE = VEC ? 4 * NV : NV;       // elements per lane
    __device__ static void load(float* v, const float* row, int lane) {
        if (VEC) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(row + 4 * lane + 256 * j);
                v[4 * j] = t[0]; v[4 * j + 1] = t[1]; v[4 * j + 2] = t[2]; v[4 * j + 3] = t[3];
            }
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j] = row[lane + 64 * j];
        }
    }
    __device__ static void store(const float* v, float* row, int lane) {
        if (VEC) {
#pragma unroll
            for (int j = 0; j < NV; ++j)
                *reinterpret_cast<f32x4*>(row + 4 * lane + 256 * j) = f32x4{v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]};
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) row[lane + 64 * j] = v[j];
        }
    }
    __device__ static int col(int e, int lane) { return VEC ? 4 * lane + 256 * (e >> 2) + (e & 3) : lane + 64 * e; }
    // bf16 rows (the bf16-storage path: GEMM-facing tensors are bf16, the arithmetic here stays fp32), same column map
    __device__ static void load(float* v, const __bf16* row, int lane) {
        if (VEC) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const bf16x4_ t = *reinterpret_cast<const bf16x4_*>(row + 4 * lane + 256 * j);
                v[4 * j] = (float)t[0]; v[4 * j + 1] = (float)t[1]; v[4 * j + 2] = (float)t[2]; v[4 * j + 3] = (float)t[3];
            }
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j] = (float)row[lane + 64 * j];
        }
    }
    __device__ static void store(const float* v, __bf16* row, int lane) {
        if (VEC) {
#pragma unroll
            for (int j = 0; j < NV; ++j)
                *reinterpret_cast<bf16x4_*>(row + 4 * lane + 256 * j) =
                    bf16x4_{(__bf16)v[4 * j], (__bf16)v[4 * j + 1], (__bf16)v[4 * j + 2], (__bf16)v[4 * j + 3]};
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) row[lane + 64 * j] = (__bf16)v[j];
        }
    }
};

template <int C, typename TY>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, TY* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int M,
                                                     float eps) {
    using R = LnRow<C>;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float v[R::NE], g[R::NE], b[R::NE];
    R::load(v, x + (size_t)row * C, lane);
    R::load(g, gamma, lane);
    R::load(b, beta, lane);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < R::NE; ++e) s += v[e];
    const float mu = wave_reduce_sum(s) * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < R::NE; ++e) {
        const float d = v[e] - mu;
        q += d * d;
    }
    const float rs = 1.0f / sqrtf(wave_reduce_sum(q) * (1.0f / C) + eps);
#pragma unroll
    for (int e = 0; e < R::NE; ++e) v[e] = (v[e] - mu) * rs * g[e] + b[e];
    R::store(v, y + (size_t)row * C, lane);
    if (lane == 0) {
        mean[row] = mu;
        rstd[row] = rs;
    }
}

constexpr int LN_BWD_ROWS = 16;  // rows per block (4 per wave): >= 700 blocks at M = 11544

template <int C, typename TDY, typename TDD>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TDY* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ add,
                                                     float* __restrict__ dx, float* __restrict__ partial, int M,
                                                     TDD* __restrict__ dx_drop, uint32_t thr, float dscale,
                                                     uint64_t seed, uint64_t seed_off_in,
                                                     const uint64_t* __restrict__ salt) {
    const uint64_t seed_off = seed_off_in + ((salt && dx_drop) ? *salt : (uint64_t)0);
    using R = LnRow<C>;
    __shared__ float red[4][2][C];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float g[R::NE], dg[R::NE], db[R::NE];
    R::load(g, gamma, lane);
#pragma unroll
    for (int e = 0; e < R::NE; ++e) { dg[e] = 0.f; db[e] = 0.f; }
    const int rbase = blockIdx.x * LN_BWD_ROWS + wave * (LN_BWD_ROWS / 4);
    for (int rr = 0; rr < LN_BWD_ROWS / 4; ++rr) {
        const int row = rbase + rr;
        if (row >= M) break;
        const size_t o = (size_t)row * C;
        const float mu = mean[row], rs = rstd[row];
        float dyv[R::NE], xh[R::NE];
        R::load(dyv, dy + o, lane);
        R::load(xh, x + o, lane);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < R::NE; ++e) {
            xh[e] = (xh[e] - mu) * rs;
            const float t = dyv[e] * g[e];
            s1 += t;
            s2 += t * xh[e];
            dg[e] += dyv[e] * xh[e];
            db[e] += dyv[e];
        }
        s1 = wave_reduce_sum(s1) * (1.0f / C);
        s2 = wave_reduce_sum(s2) * (1.0f / C);
        float av[R::NE];
        if (add) R::load(av, add + o, lane);
#pragma unroll
        for (int e = 0; e < R::NE; ++e) {
            float v = rs * (dyv[e] * g[e] - s1 - xh[e] * s2);
            if (add) v += av[e];
            dyv[e] = v;
        }
        R::store(dyv, dx + o, lane);
        if (dx_drop) {  // the consumer of dx on the dropout branch (resid_drop of the block below) wants dropout(dx)
#pragma unroll
            for (int e = 0; e < R::NE; ++e)
                dyv[e] = ds6g_keep(seed, seed_off + o + R::col(e, lane), thr) ? dyv[e] * dscale : 0.f;
            R::store(dyv, dx_drop + o, lane);
        }
    }
#pragma unroll
    for (int e = 0; e < R::NE; ++e) {
        red[wave][0][R::col(e, lane)] = dg[e];
        red[wave][1][R::col(e, lane)] = db[e];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int which = i / C, c = i % C;
        partial[(size_t)blockIdx.x * 2 * C + i] =
            red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
    }
}

// out[c] (+)= sum_b partial[b][c]   (generic column finalize over row-block partials)
__global__ __launch_bounds__(256) void col_finalize_kernel(const float* __restrict__ partial, int nblk, int ncol,
                                                           float* __restrict__ out0, float* __restrict__ out1,
                                                           int split_at, int accumulate) {
    // 4 columns x 64 partial-groups per block (like the BatchNorm finalize kernels: latency-bound launches, 68 per step)
    constexpr int CF_COLS = 4, CF_GROUPS = 64;
    __shared__ double red[4][CF_COLS];
    const int cl = threadIdx.x & (CF_COLS - 1), g = threadIdx.x / CF_COLS;
    const int c = blockIdx.x * CF_COLS + cl;
    double s = 0.0;
    if (c < ncol)
        for (int b = g; b < nblk; b += CF_GROUPS) s += (double)partial[(size_t)b * ncol + c];
#pragma unroll
    for (int o = CF_COLS; o < 64; o <<= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) < CF_COLS) red[threadIdx.x >> 6][threadIdx.x & 63] = s;
    __syncthreads();
    if (g != 0 || c >= ncol) return;
    s = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
    float* dst = (c < split_at) ? (out0 + c) : (out1 + (c - split_at));
    *dst = accumulate ? (*dst + (float)s) : (float)s;
}

// partial[blk][c] = sum over the block's rows of x[r][c]   (bias gradients, pos_emb gradient)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, long M, int C,
                                                             int rows_per_block, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float redf[];  // [rowlanes][C]
    const int cg = C >> 2;
    const int cols_per_pass = min(cg, 256);
    const int rowlanes = 256 / cols_per_pass;
    const int rowlane = threadIdx.x / cols_per_pass;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(M, r0 + rows_per_block);
    for (int cbase = 0; cbase < cg; cbase += cols_per_pass) {
        const int colg = cbase + threadIdx.x % cols_per_pass;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (colg < cg)
            for (long r = r0 + rowlane; r < r1; r += rowlanes) s += *reinterpret_cast<const f32x4*>(x + (size_t)r * C + colg * 4);
        if (colg < cg) *reinterpret_cast<f32x4*>(redf + (size_t)rowlane * C + colg * 4) = s;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float acc = 0.f;
        for (int rl = 0; rl < rowlanes; ++rl) acc += redf[(size_t)rl * C + c];
        partial[(size_t)blockIdx.x * C + c] = acc;
    }
}

// eval-mode BatchNorm folded into the preceding convolution: w_out[o][j] = w[o][j] * g[o] / sqrt(var[o] + eps),
// bias_out[o] = b[o] - mean[o] * g[o] / sqrt(var[o] + eps).  cin < cout_pad zero-pads each tap (stem: 3/1/2 -> 4 ch).
__global__ __launch_bounds__(256) void bn_fold_kernel(const float* __restrict__ w, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ mean,
                                                      const float* __restrict__ var, float eps, float* __restrict__ w_out,
                                                      float* __restrict__ bias_out, int K, int taps, int cin, int cpad) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per_o = (long)taps * cpad;
    if (i >= (long)K * per_o) return;
    const int o = (int)(i / per_o);
    const int rem = (int)(i - (long)o * per_o);
    const int tap = rem / cpad, c = rem - tap * cpad;
    const float scale = gamma[o] / sqrtf(var[o] + eps);
    w_out[i] = c < cin ? w[((long)o * taps + tap) * cin + c] * scale : 0.f;
    if (rem == 0) bias_out[o] = beta[o] - mean[o] * scale;
}

}  // namespace

extern "C" size_t ds6g_layernorm_bwd_workspace_bytes(int M, int C);
extern "C" size_t ds6g_bn_workspace_bytes(long M, int C);

// TX: storage type of the BN input x (and of dx); TA: of dy / y_mask / dres (the stem mixes fp32 x with a bf16 pool gradient)
template <typename TX>
static int bn_stats_run(const TX* x, long M, int C, float eps, float momentum, float* mean, float* invstd,
                        float* running_mean, float* running_var, void* ws, size_t ws_bytes, void* stream) {
    int rpb, nblk;
    DS6G_CHECK_ARG(x && mean && invstd && ws && M > 0);
    DS6G_CHECK_ARG(bn_geometry(M, C, &rpb, &nblk) == DS6G_OK);
    DS6G_CHECK_ARG(ws_bytes >= ds6g_bn_workspace_bytes(M, C));
    double* partial = (double*)ws;
    const size_t lds = (size_t)(256 / (C / 4)) * 2 * C * sizeof(double);
    hipLaunchKernelGGL((bn_reduce_kernel<0, TX, TX>), dim3(nblk), dim3(256), lds, (hipStream_t)stream, x, (const TX*)nullptr,
                       (const TX*)nullptr, (const float*)nullptr, (const float*)nullptr, M, C, rpb, partial,
                       (const float*)nullptr, (const float*)nullptr, PoolGrad{});
    DS6G_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(cdiv(C, BNF_COLS)), dim3(256), 0, (hipStream_t)stream, partial, nblk,
                       M, C, eps, momentum, mean, invstd, running_mean, running_var);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

template <typename TA>
static int bn_apply_run(const TA* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                        const TA* residual, TA* y, long M, int C, int relu, void* stream) {
    DS6G_CHECK_ARG(x && mean && invstd && gamma && beta && y && C % 4 == 0);
    const long total4 = M * C / 4;
    const int grid = (int)min((long)8192, (total4 + 255) / 256);
    hipLaunchKernelGGL((bn_apply_kernel<TA>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, mean, invstd, gamma, beta,
                       residual, y, total4, C, relu);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

// dyeff = dy * (y_mask > 0) (y_mask nullable); dx = gamma*invstd*(dyeff - mean(dyeff) - xhat*mean(dyeff*xhat));
// dgamma/dbeta (+)=; dres (nullable) receives dyeff (gradient of the residual branch).
// relu_beta (nullable, with y_mask NULL): the BN was followed by ReLU with NO residual added in between - the mask is
// re-derived as (gamma * xhat + relu_beta > 0) from x, which the kernels read anyway, instead of reading the activation
template <typename TX, typename TA>
static int bn_bwd_run(const TA* dy, const PoolGrad pg, const TA* y_mask, const TX* x, const float* mean,
                      const float* invstd, const float* gamma, const float* relu_beta, TX* dx, float* dgamma,
                      float* dbeta, TA* dres, long M, int C, int accumulate_param_grads, void* ws, size_t ws_bytes,
                      void* stream) {
    const float* mg = relu_beta ? gamma : nullptr;
    int rpb, nblk;
    DS6G_CHECK_ARG(bn_geometry(M, C, &rpb, &nblk) == DS6G_OK);
    DS6G_CHECK_ARG(ws_bytes >= ds6g_bn_workspace_bytes(M, C));
    double* partial = (double*)ws;
    float* coef = (float*)((char*)ws + (size_t)nblk * 2 * C * sizeof(double));
    const size_t lds = (size_t)(256 / (C / 4)) * 2 * C * sizeof(double);
    hipLaunchKernelGGL((bn_reduce_kernel<1, TX, TA>), dim3(nblk), dim3(256), lds, (hipStream_t)stream, x, dy, y_mask, mean,
                       invstd, M, C, rpb, partial, mg, relu_beta, pg);
    DS6G_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, BNF_COLS)), dim3(256), 0, (hipStream_t)stream, partial, nblk, M,
                       C, gamma, invstd, dgamma, dbeta, coef, accumulate_param_grads);
    DS6G_LAUNCH_CHECK();
    const long total4 = M * C / 4;
    const int grid = (int)min((long)8192, (total4 + 255) / 256);
    hipLaunchKernelGGL((bn_bwd_apply_kernel<TX, TA>), dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, y_mask, x, mean,
                       invstd, coef, dx, dres, total4, C, mg, relu_beta, pg);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

template <typename TY>
static int ln_fwd_launch(const float* x, const float* gamma, const float* beta, TY* y, float* mean, float* rstd, int M, int C,
                         float eps, hipStream_t st) {
    dim3 grid(cdiv(M, 4)), block(256);
    switch (C / 64) {
        case 1: hipLaunchKernelGGL((ln_fwd_kernel<64, TY>), grid, block, 0, st, x, gamma, beta, y, mean, rstd, M, eps); break;
        case 2: hipLaunchKernelGGL((ln_fwd_kernel<128, TY>), grid, block, 0, st, x, gamma, beta, y, mean, rstd, M, eps); break;
        case 4: hipLaunchKernelGGL((ln_fwd_kernel<256, TY>), grid, block, 0, st, x, gamma, beta, y, mean, rstd, M, eps); break;
        case 8: hipLaunchKernelGGL((ln_fwd_kernel<512, TY>), grid, block, 0, st, x, gamma, beta, y, mean, rstd, M, eps); break;
        default: DS6G_CHECK_ARG(!"LayerNorm width must be 64/128/256/512");
    }
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

template <typename TDY, typename TDD>
static int ln_bwd_launch(const TDY* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                         const float* add, float* dx, float* dgamma, float* dbeta, int M, int C, int accumulate_param_grads,
                         TDD* dx_drop, float drop_p, uint64_t seed, uint64_t seed_off, void* ws, size_t ws_bytes,
                         hipStream_t st) {
    DS6G_CHECK_ARG(dy && x && mean && rstd && gamma && dx && dgamma && dbeta && ws);
    DS6G_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f);
    const uint32_t thr = ds6g_drop_threshold(drop_p);
    const float dscale = 1.f / (1.f - drop_p);
    DS6G_CHECK_ARG(C % 64 == 0 && C <= 512);
    DS6G_CHECK_ARG(ws_bytes >= ds6g_layernorm_bwd_workspace_bytes(M, C));
    const int nblk = cdiv(M, LN_BWD_ROWS);
    float* partial = (float*)ws;
    dim3 grid(nblk), block(256);
    switch (C / 64) {
        case 1: hipLaunchKernelGGL((ln_bwd_kernel<64, TDY, TDD>), grid, block, 0, st, dy, x, mean, rstd, gamma, add, dx, partial, M, dx_drop, thr, dscale, seed, seed_off, g_ds6g_salt); break;
        case 2: hipLaunchKernelGGL((ln_bwd_kernel<128, TDY, TDD>), grid, block, 0, st, dy, x, mean, rstd, gamma, add, dx, partial, M, dx_drop, thr, dscale, seed, seed_off, g_ds6g_salt); break;
        case 4: hipLaunchKernelGGL((ln_bwd_kernel<256, TDY, TDD>), grid, block, 0, st, dy, x, mean, rstd, gamma, add, dx, partial, M, dx_drop, thr, dscale, seed, seed_off, g_ds6g_salt); break;
        case 8: hipLaunchKernelGGL((ln_bwd_kernel<512, TDY, TDD>), grid, block, 0, st, dy, x, mean, rstd, gamma, add, dx, partial, M, dx_drop, thr, dscale, seed, seed_off, g_ds6g_salt); break;
        default: DS6G_CHECK_ARG(!"LayerNorm width must be 64/128/256/512");
    }
    DS6G_LAUNCH_CHECK();
    hipLaunchKernelGGL(col_finalize_kernel, dim3(cdiv(2 * C, 4)), dim3(256), 0, st, partial, nblk, 2 * C, dgamma,
                       dbeta, C, accumulate_param_grads);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

// bgemm.hip: the conv epilogue wrote the partials (ds6g_bf16_conv2d_fwd_bnstats)
int ds6g_internal_bn_stats_finalize(const double* partial, int nblk, long M, int C, float eps, float momentum, float* mean,
                                    float* invstd, float* running_mean, float* running_var, hipStream_t st) {
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(cdiv(C, BNF_COLS)), dim3(256), 0, st, partial, nblk, M, C, eps,
                       momentum, mean, invstd, running_mean, running_var);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

extern "C" {

size_t ds6g_bn_workspace_bytes(long M, int C) {
    int rpb, nblk;
    if (bn_geometry(M, C, &rpb, &nblk)) return 0;
    return (size_t)nblk * 2 * C * sizeof(double) + 3 * (size_t)C * sizeof(float);
}

// training-mode statistics of x[M][C]; updates running stats in place when given
int ds6g_bn_stats(const float* x, long M, int C, float eps, float momentum, float* mean, float* invstd,
                  float* running_mean, float* running_var, void* ws, size_t ws_bytes, void* stream) {
    DS6G_ENTER();
    return bn_stats_run<float>(x, M, C, eps, momentum, mean, invstd, running_mean, running_var, ws, ws_bytes, stream);
}
// bf16-storage path: the conv output is bf16; statistics, running stats and all arithmetic stay fp32
int ds6g_bf16_bn_stats(const void* x, long M, int C, float eps, float momentum, float* mean, float* invstd,
                       float* running_mean, float* running_var, void* ws, size_t ws_bytes, void* stream) {
    DS6G_ENTER();
    return bn_stats_run<__bf16>((const __bf16*)x, M, C, eps, momentum, mean, invstd, running_mean, running_var, ws, ws_bytes,
                                stream);
}

int ds6g_bn_eval_prepare(const float* running_mean, const float* running_var, int C, float eps, float* mean,
                         float* invstd, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(running_mean && running_var && mean && invstd);
    hipLaunchKernelGGL(bn_eval_prepare_kernel, dim3(cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, running_mean,
                       running_var, C, eps, mean, invstd);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

int ds6g_bn_apply(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                  const float* residual, float* y, long M, int C, int relu, void* stream) {
    DS6G_ENTER();
    return bn_apply_run<float>(x, mean, invstd, gamma, beta, residual, y, M, C, relu, stream);
}
int ds6g_bf16_bn_apply(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                       const void* residual, void* y, long M, int C, int relu, void* stream) {
    DS6G_ENTER();
    return bn_apply_run<__bf16>((const __bf16*)x, mean, invstd, gamma, beta, (const __bf16*)residual, (__bf16*)y, M, C, relu,
                                stream);
}

int ds6g_bn_bwd(const float* dy, const float* y_mask, const float* x, const float* mean, const float* invstd,
                const float* gamma, const float* relu_beta, float* dx, float* dgamma, float* dbeta, float* dres, long M,
                int C, int accumulate_param_grads, void* ws, size_t ws_bytes, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(!(y_mask && relu_beta));
    DS6G_CHECK_ARG(dy && x && mean && invstd && gamma && dx && dgamma && dbeta && ws);
    return bn_bwd_run<float, float>(dy, PoolGrad{}, y_mask, x, mean, invstd, gamma, relu_beta, dx, dgamma, dbeta, dres, M, C,
                                    accumulate_param_grads, ws, ws_bytes, stream);
}
// bf16-storage path: dy / y_mask / x / dx / dres bf16 (dgamma / dbeta and the reductions fp32 / fp64)
int ds6g_bf16_bn_bwd(const void* dy, const void* y_mask, const void* x, const float* mean, const float* invstd,
                     const float* gamma, const float* relu_beta, void* dx, float* dgamma, float* dbeta, void* dres, long M,
                     int C, int accumulate_param_grads, void* ws, size_t ws_bytes, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(!(y_mask && relu_beta));
    DS6G_CHECK_ARG(dy && x && mean && invstd && gamma && dx && dgamma && dbeta && ws);
    return bn_bwd_run<__bf16, __bf16>((const __bf16*)dy, PoolGrad{}, (const __bf16*)y_mask, (const __bf16*)x, mean, invstd,
                                      gamma, relu_beta, (__bf16*)dx, dgamma, dbeta, (__bf16*)dres, M, C,
                                      accumulate_param_grads, ws, ws_bytes, stream);
}

// BN -> ReLU -> 3x3/2 max-pool (the ResNet stem, model2_seq.py:495-500 via torchvision) backward in one pass over x: the
// pool gradient is gathered from (dpool, idx) where the BN kernels would read dy, so the dense [N][H][W][C] gradient of
// the pool input is never written or read
static int bn_bwd_maxpool_impl(const void* dpool, int dp16, const uint8_t* idx, const float* x, const float* mean,
                               const float* invstd, const float* gamma, const float* relu_beta, float* dx, float* dgamma,
                               float* dbeta, int N, int H, int W, int C, int accumulate_param_grads, void* ws,
                               size_t ws_bytes, void* stream) {
    DS6G_CHECK_ARG(dpool && idx && x && mean && invstd && gamma && relu_beta && dx && dgamma && dbeta && ws);
    DS6G_CHECK_ARG(N > 0 && H > 0 && W > 0 && C % 4 == 0);
    const PoolGrad pg{dpool, idx, H, W, (H + 2 - 3) / 2 + 1, (W + 2 - 3) / 2 + 1, dp16};
    return bn_bwd_run<float, float>(nullptr, pg, nullptr, x, mean, invstd, gamma, relu_beta, dx, dgamma, dbeta, nullptr,
                                    (long)N * H * W, C, accumulate_param_grads, ws, ws_bytes, stream);
}
int ds6g_bn_bwd_maxpool(const float* dpool, const uint8_t* idx, const float* x, const float* mean, const float* invstd,
                        const float* gamma, const float* relu_beta, float* dx, float* dgamma, float* dbeta, int N, int H,
                        int W, int C, int accumulate_param_grads, void* ws, size_t ws_bytes, void* stream) {
    DS6G_ENTER();
    return bn_bwd_maxpool_impl(dpool, 0, idx, x, mean, invstd, gamma, relu_beta, dx, dgamma, dbeta, N, H, W, C,
                               accumulate_param_grads, ws, ws_bytes, stream);
}
// bf16-storage path: the stem's conv output x and its gradient dx stay fp32 (the 4-channel stem keeps the fp32-storage
// kernels), the gradient of the pooled tensor arrives as bf16
int ds6g_bn_bwd_maxpool_bf16in(const void* dpool, const uint8_t* idx, const float* x, const float* mean,
                               const float* invstd, const float* gamma, const float* relu_beta, float* dx, float* dgamma,
                               float* dbeta, int N, int H, int W, int C, int accumulate_param_grads, void* ws,
                               size_t ws_bytes, void* stream) {
    DS6G_ENTER();
    return bn_bwd_maxpool_impl(dpool, 1, idx, x, mean, invstd, gamma, relu_beta, dx, dgamma, dbeta, N, H, W, C,
                               accumulate_param_grads, ws, ws_bytes, stream);
}

// bf16 stem (csrc/stem.hip): conv output x and its gradient dx are bf16 too
int ds6g_bf16_stem_bn_bwd_maxpool(const void* dpool, const uint8_t* idx, const void* x, const float* mean,
                                  const float* invstd, const float* gamma, const float* relu_beta, void* dx, float* dgamma,
                                  float* dbeta, int N, int H, int W, int C, int accumulate_param_grads, void* ws,
                                  size_t ws_bytes, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(dpool && idx && x && mean && invstd && gamma && relu_beta && dx && dgamma && dbeta && ws);
    DS6G_CHECK_ARG(N > 0 && H > 0 && W > 0 && C % 4 == 0);
    const PoolGrad pg{dpool, idx, H, W, (H + 2 - 3) / 2 + 1, (W + 2 - 3) / 2 + 1, 1};
    return bn_bwd_run<__bf16, __bf16>(nullptr, pg, nullptr, (const __bf16*)x, mean, invstd, gamma, relu_beta, (__bf16*)dx,
                                      dgamma, dbeta, nullptr, (long)N * H * W, C, accumulate_param_grads, ws, ws_bytes, stream);
}

int ds6g_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                       int M, int C, float eps, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(x && gamma && beta && y && mean && rstd && M > 0);
    DS6G_CHECK_ARG(C % 64 == 0 && C <= 512);
    return ln_fwd_launch<float>(x, gamma, beta, y, mean, rstd, M, C, eps, (hipStream_t)stream);
}

// bf16-storage path: the normalised rows feed a GEMM and are written as bf16 (statistics and arithmetic fp32)
int ds6g_layernorm_fwd_bf16out(const float* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                               int M, int C, float eps, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(x && gamma && beta && y && mean && rstd && M > 0);
    DS6G_CHECK_ARG(C % 64 == 0 && C <= 512);
    return ln_fwd_launch<__bf16>(x, gamma, beta, (__bf16*)y, mean, rstd, M, C, eps, (hipStream_t)stream);
}

size_t ds6g_layernorm_bwd_workspace_bytes(int M, int C) { return (size_t)cdiv(M, LN_BWD_ROWS) * 2 * C * sizeof(float); }

// dx = add? + LN'(dy); dgamma/dbeta (+)=
int ds6g_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                       const float* add, float* dx, float* dgamma, float* dbeta, int M, int C,
                       int accumulate_param_grads, float* dx_drop, float drop_p, uint64_t seed, uint64_t seed_off,
                       void* ws, size_t ws_bytes, void* stream) {
    DS6G_ENTER();
    return ln_bwd_launch<float, float>(dy, x, mean, rstd, gamma, add, dx, dgamma, dbeta, M, C, accumulate_param_grads, dx_drop,
                                       drop_p, seed, seed_off, ws, ws_bytes, (hipStream_t)stream);
}

// bf16-storage path: dy is a GEMM output stored as bf16 (dy16) or fp32; dx (the residual-stream gradient) stays fp32;
// dx_drop (nullable) = dropout(dx) is the next GEMM's operand and is written as bf16 (with drop_p = 0: a bf16 copy of dx)
int ds6g_layernorm_bwd_bf16(const void* dy, int dy16, const float* x, const float* mean, const float* rstd,
                            const float* gamma, const float* add, float* dx, float* dgamma, float* dbeta, int M, int C,
                            int accumulate_param_grads, void* dx_drop, float drop_p, uint64_t seed, uint64_t seed_off,
                            void* ws, size_t ws_bytes, void* stream) {
    DS6G_ENTER();
    if (dy16)
        return ln_bwd_launch<__bf16, __bf16>((const __bf16*)dy, x, mean, rstd, gamma, add, dx, dgamma, dbeta, M, C,
                                             accumulate_param_grads, (__bf16*)dx_drop, drop_p, seed, seed_off, ws, ws_bytes,
                                             (hipStream_t)stream);
    return ln_bwd_launch<float, __bf16>((const float*)dy, x, mean, rstd, gamma, add, dx, dgamma, dbeta, M, C,
                                        accumulate_param_grads, (__bf16*)dx_drop, drop_p, seed, seed_off, ws, ws_bytes,
                                        (hipStream_t)stream);
}

constexpr int COLSUM_ROWS = 32;
size_t ds6g_colsum_workspace_bytes(long M, int C) { return (size_t)cdiv(M, COLSUM_ROWS) * C * sizeof(float); }

// out[c] (+)= sum_r x[r][c]
int ds6g_colsum(const float* x, long M, int C, float* out, int accumulate, void* ws, size_t ws_bytes, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(x && out && ws && C % 4 == 0 && M > 0);
    DS6G_CHECK_ARG(ws_bytes >= ds6g_colsum_workspace_bytes(M, C));
    const int rpb = COLSUM_ROWS;
    const int nblk = cdiv(M, rpb);
    const int cols_per_pass = (C / 4) < 256 ? (C / 4) : 256;
    DS6G_CHECK_ARG(256 % cols_per_pass == 0);
    const size_t lds = (size_t)(256 / cols_per_pass) * C * sizeof(float);
    float* partial = (float*)ws;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(256), lds, (hipStream_t)stream, x, M, C, rpb, partial);
    DS6G_LAUNCH_CHECK();
    hipLaunchKernelGGL(col_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), 0, (hipStream_t)stream, partial, nblk, C, out,
                       out, C, accumulate);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

int ds6g_bn_fold(const float* w, const float* gamma, const float* beta, const float* running_mean,
                 const float* running_var, float eps, float* w_out, float* bias_out, int K, int taps, int cin, int cpad,
                 void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(w && gamma && beta && running_mean && running_var && w_out && bias_out && K > 0 && taps > 0 &&
                   cin > 0 && cpad >= cin);
    const long n = (long)K * taps * cpad;
    hipLaunchKernelGGL(bn_fold_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, gamma, beta,
                       running_mean, running_var, eps, w_out, bias_out, K, taps, cin, cpad);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

}  // extern "C"
