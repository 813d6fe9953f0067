// Training-step kernels around the fusion model (fp32):
//   sigmoid focal loss on soft targets, fused forward + gradient   train2_seq.py:291-301
//   AdamW (decoupled wd) + EMA shadow update over a flat arena      train2_seq.py:131-134,315-320,539
//   tiny Linear layers (GPS embedding chain, join MLP; M <= 64)     model2_seq.py:422-425,863-869
// The optimizer is HBM-bound: one pass reads p,g,m,v(,shadow) and writes p,m,v(,shadow) with
// 16-B accesses; algorithmic traffic 28 B/param (+12 B/param with EMA).
#include "common.h"

int g_ds6g_bf16 = 0;
thread_local const uint64_t* g_ds6g_salt = nullptr;

namespace {

// loss = mean_i ce_i (1-p_t)^gamma (alpha t + (1-alpha)(1-t));  dlogits_i = dloss/dx_i * upstream
__global__ __launch_bounds__(256) void focal_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                    float* __restrict__ loss, float* __restrict__ dx, int n,
                                                    float alpha, float gamma, float upstream) {
    __shared__ float red[4];
    float acc = 0.f;
    const float inv_n = 1.0f / (float)n;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float xi = x[i], ti = t[i];
        const float p = 1.0f / (1.0f + expf(-xi));
        // BCE-with-logits, numerically stable: max(x,0) - x t + log(1 + exp(-|x|))
        const float ce = fmaxf(xi, 0.f) - xi * ti + log1pf(expf(-fabsf(xi)));
        const float pt = p * ti + (1.f - p) * (1.f - ti);
        const float om = 1.f - pt;
        const float at = alpha >= 0.f ? alpha * ti + (1.f - alpha) * (1.f - ti) : 1.f;
        const float mod = powf(om, gamma);
        acc += at * ce * mod;
        if (dx) {
            // d ce/dx = p - t ; d pt/dx = (2t-1) p (1-p) ; d mod/dx = -gamma om^(gamma-1) dpt/dx
            const float dpt = (2.f * ti - 1.f) * p * (1.f - p);
            const float dmod = om > 0.f ? -gamma * powf(om, gamma - 1.f) * dpt : 0.f;
            dx[i] = at * ((p - ti) * mod + ce * dmod) * inv_n * upstream;
        }
    }
    acc = wave_reduce_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *loss = (red[0] + red[1] + red[2] + red[3]) * inv_n;
}

struct AdamArgs {
    float lr, beta1, beta2, eps, wd, bc1, bc2_sqrt, ema_decay, grad_scale;
};
// device-resident per-step scalars of the optimizer (ds6g_adamw_state_advance / ds6g_adamw_step_dev): a captured hipGraph of
// the training step freezes its launch arguments, so what changes from step to step lives in device memory instead
struct AdamDev {
    float lr;        // written by the host when the schedule changes it
    float bc1;       // 1 - beta1^step
    float bc2_sqrt;  // sqrt(1 - beta2^step)
    int step;        // 1-based count of steps taken
};
__global__ void adamw_advance_kernel(AdamDev* st, float beta1, float beta2) {
    const int step = st->step + 1;
    st->step = step;
    st->bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    st->bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
}

// sum of squares of n4*4 floats: per-block double partials, the LAST block to finish (device-scope counter) adds them in
// index order (deterministic) and writes out[0] = ||g||_2, out[1] = min(1, max_norm / (||g||_2 + 1e-6)) - the
// clip coefficient of torch.nn.utils.clip_grad_norm_ (train2_seq_30to5.py:120), applied by adamw_kernel through
// grad_scale_dev so the gradient arena is not rewritten.  ws: [0] = counter (zero on entry, re-zeroed on exit), doubles
// from byte 8 on.
__global__ __launch_bounds__(256) void grad_norm_kernel(const float* __restrict__ g, long n4, float max_norm,
                                                        float pre_scale, float* __restrict__ out,
                                                        unsigned* __restrict__ counter, double* __restrict__ partial) {
    __shared__ double red[4];
    __shared__ bool last;
    double acc = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(g + i * 4);
        acc += (double)(v[0] * v[0] + v[1] * v[1]) + (double)(v[2] * v[2] + v[3] * v[3]);
    }
    acc = wave_reduce_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
        __threadfence();
        last = atomicAdd(counter, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    double tot = 0.0;
    for (unsigned i = threadIdx.x; i < gridDim.x; i += 256) tot += __builtin_nontemporal_load(partial + i);
    tot = wave_reduce_sum_d(tot);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = tot;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt((red[0] + red[1]) + (red[2] + red[3])) * pre_scale;
        out[0] = norm;
        const float coef = max_norm / (norm + 1e-6f);
        out[1] = coef < 1.f ? coef : 1.f;
        *counter = 0u;
    }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    float* __restrict__ shadow, long n4, AdamArgs a,
                                                    const float* __restrict__ grad_scale_dev,
                                                    const AdamDev* __restrict__ dev) {
    if (grad_scale_dev) a.grad_scale *= *grad_scale_dev;
    if (dev) { a.lr = dev->lr; a.bc1 = dev->bc1; a.bc2_sqrt = dev->bc2_sqrt; }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 pv = *reinterpret_cast<f32x4*>(p + i * 4);
        const f32x4 gv = a.grad_scale * *reinterpret_cast<const f32x4*>(g + i * 4);
        f32x4 mv = *reinterpret_cast<f32x4*>(m + i * 4);
        f32x4 vv = *reinterpret_cast<f32x4*>(v + i * 4);
        pv *= (1.f - a.lr * a.wd);
        mv = a.beta1 * mv + (1.f - a.beta1) * gv;
        vv = a.beta2 * vv + (1.f - a.beta2) * gv * gv;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float denom = sqrtf(vv[j]) / a.bc2_sqrt + a.eps;
            pv[j] -= (a.lr / a.bc1) * (mv[j] / denom);
        }
        *reinterpret_cast<f32x4*>(p + i * 4) = pv;
        *reinterpret_cast<f32x4*>(m + i * 4) = mv;
        *reinterpret_cast<f32x4*>(v + i * 4) = vv;
        if (shadow) {
            f32x4 sv = *reinterpret_cast<f32x4*>(shadow + i * 4);
            sv = (1.f - a.ema_decay) * pv + a.ema_decay * sv;
            *reinterpret_cast<f32x4*>(shadow + i * 4) = sv;
        }
    }
}

// dst[i] = bf16(src[i]) (RNE): the bf16 shadow of the parameter arena, refreshed once per training forward of the
// bf16-storage path (fp32 master weights stay in the arena)
// resident for `ticks` of the 100 MHz real-time counter (ds6g_debug_occupy_cus); touches its dynamic LDS so the allocation is real
__global__ __launch_bounds__(256) void occupy_kernel(unsigned long long ticks) {
    extern __shared__ unsigned char occ_lds[];
    if (threadIdx.x == 0) occ_lds[0] = 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long n4) {
    typedef __bf16 bf16x4_ __attribute__((ext_vector_type(4)));
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + i * 4);
        *reinterpret_cast<bf16x4_*>(dst + i * 4) = bf16x4_{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    }
}

// y[m][n] = act( sum_k x[row(m)][k] w[n][k] + b[n] ), row(m) = (m / rpg) * gstride + (m % rpg) * K
__global__ void small_linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                        const float* __restrict__ b, float* __restrict__ y, int M, int N, int K,
                                        int rpg, long gstride, int relu) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * N) return;
    const int n = i % N, m = i / N;
    const float* xr = x + (long)(m / rpg) * gstride + (long)(m % rpg) * K;
    const float* wr = w + (long)n * K;
    float s = b ? b[n] : 0.f;
    for (int k = 0; k < K; ++k) s = fmaf(xr[k], wr[k], s);
    y[i] = relu ? fmaxf(s, 0.f) : s;
}

// dyeff = dy * (y_mask > 0); dx[row(m)][k] (+)= sum_n dyeff[m][n] w[n][k]
__global__ void small_linear_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ y_mask,
                                          const float* __restrict__ w, float* __restrict__ dx, int M, int N, int K,
                                          int rpg, long gstride, int accumulate) {  // rpg/gstride address dx
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * K) return;
    const int k = i % K, m = i / K;
    // few threads (M*K) and a long reduction: four independent chains, eight loads each in flight
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const float* dyr = dy + (long)m * N;
    const float* mkr = y_mask ? y_mask + (long)m * N : nullptr;
    int n = 0;
#pragma unroll 2
    for (; n + 4 <= N; n += 4) {
        float g0 = dyr[n], g1 = dyr[n + 1], g2 = dyr[n + 2], g3 = dyr[n + 3];
        if (mkr) {
            if (!(mkr[n] > 0.f)) g0 = 0.f;
            if (!(mkr[n + 1] > 0.f)) g1 = 0.f;
            if (!(mkr[n + 2] > 0.f)) g2 = 0.f;
            if (!(mkr[n + 3] > 0.f)) g3 = 0.f;
        }
        s0 = fmaf(g0, w[(long)n * K + k], s0);
        s1 = fmaf(g1, w[(long)(n + 1) * K + k], s1);
        s2 = fmaf(g2, w[(long)(n + 2) * K + k], s2);
        s3 = fmaf(g3, w[(long)(n + 3) * K + k], s3);
    }
    for (; n < N; ++n) {
        float g = dyr[n];
        if (mkr && !(mkr[n] > 0.f)) g = 0.f;
        s0 = fmaf(g, w[(long)n * K + k], s0);
    }
    const float s = (s0 + s1) + (s2 + s3);
    float* d = dx + (long)(m / rpg) * gstride + (long)(m % rpg) * K + k;
    *d = accumulate ? *d + s : s;
}

// dw[n][k] (+)= sum_m dyeff[m][n] x[row(m)][k];  db[n] (+)= sum_m dyeff[m][n]
__global__ void small_linear_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ y_mask,
                                          const float* __restrict__ x, float* __restrict__ dw,
                                          float* __restrict__ db, int M, int N, int K, int rpg, long gstride,
                                          int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * (K + 1)) return;
    const int k = i % (K + 1), n = i / (K + 1);
    float s = 0.f;
    for (int m = 0; m < M; ++m) {
        float g = dy[(long)m * N + n];
        if (y_mask && !(y_mask[(long)m * N + n] > 0.f)) g = 0.f;
        const float xv = (k < K) ? x[(long)(m / rpg) * gstride + (long)(m % rpg) * K + k] : 1.f;
        s = fmaf(g, xv, s);
    }
    float* d = (k < K) ? (dw + (long)n * K + k) : (db + n);
    *d = accumulate ? *d + s : s;
}

}  // namespace

extern "C" {

// loss (1 float) and, when dlogits != NULL, dlogits = upstream * dloss/dlogits.  n = B*64 elements.
int ds6g_focal_loss(const float* logits, const float* target, float* loss, float* dlogits, int n, float alpha,
                    float gamma, float upstream, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(logits && target && loss && n > 0);
    hipLaunchKernelGGL(focal_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, target, loss, dlogits, n,
                       alpha, gamma, upstream);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

// one AdamW step (torch semantics: decoupled weight decay, bias correction, eps outside the sqrt
// scaling) over n contiguous floats; step is 1-based.  grad_scale multiplies g first (1/world for DP).
// shadow != NULL also does the EMA update shadow = (1-d) p_new + d shadow.
int ds6g_adamw_step(float* p, const float* g, float* m, float* v, float* shadow, long n, int step, float lr,
                    float beta1, float beta2, float eps, float wd, float ema_decay, float grad_scale,
                    const float* grad_scale_dev, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(p && g && m && v && n % 4 == 0 && step >= 1);
    AdamArgs a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = wd;
    a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    a.ema_decay = ema_decay; a.grad_scale = grad_scale;
    const long n4 = n / 4;
    const int grid = (int)(n4 + 255) / 256 < 4096 ? (int)((n4 + 255) / 256) : 4096;
    hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, shadow, n4, a,
                       grad_scale_dev, (const AdamDev*)nullptr);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

// state: 16 bytes of device memory {float lr, float bc1, float bc2_sqrt, int step} (zero-initialised; the host writes lr).
// advance: step += 1 and the bias corrections of that step, computed on the device.
int ds6g_adamw_state_advance(void* state, float beta1, float beta2, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(state);
    hipLaunchKernelGGL(adamw_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (AdamDev*)state, beta1, beta2);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}
// ds6g_adamw_step with lr / bias corrections read from `state` at run time (hipGraph-replayable: no per-step launch argument)
int ds6g_adamw_step_dev(float* p, const float* g, float* m, float* v, float* shadow, long n, const void* state, float beta1,
                        float beta2, float eps, float wd, float ema_decay, float grad_scale, const float* grad_scale_dev,
                        void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(p && g && m && v && state && n % 4 == 0);
    AdamArgs a;
    a.lr = 0.f; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = wd; a.bc1 = 1.f; a.bc2_sqrt = 1.f;
    a.ema_decay = ema_decay; a.grad_scale = grad_scale;
    const long n4 = n / 4;
    const int grid = (int)(n4 + 255) / 256 < 4096 ? (int)((n4 + 255) / 256) : 4096;
    hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, shadow, n4, a, grad_scale_dev,
                       (const AdamDev*)state);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

// out[0] = pre_scale * ||g||_2 over n floats, out[1] = min(1, max_norm / (out[0] + 1e-6)): the global-norm clip of
// torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm) on the flat gradient arena (pre_scale = 1/world when the
// arena holds a rank SUM).  Pass out + 1 as ds6g_adamw_step's grad_scale_dev.  ws: >= 8 + 8 * 1024 bytes, first 4 bytes
// zero on first use (the kernel leaves them zero).
int ds6g_grad_norm_clip(const float* g, long n, float max_norm, float pre_scale, float* out, void* ws, size_t ws_bytes,
                        void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(g && out && ws && n > 0 && n % 4 == 0 && max_norm > 0.f);
    const long n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 < 1024 ? (n4 + 255) / 256 : 1024);
    if (ws_bytes < 8 + 8 * (size_t)grid) return DS6G_ERR_WORKSPACE;
    hipLaunchKernelGGL(grad_norm_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, n4, max_norm, pre_scale, out,
                       (unsigned*)ws, (double*)((char*)ws + 8));
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

// x rows are addressed in groups: row m lives at x + (m / rows_per_group) * group_stride + (m % rows_per_group) * K
// (dense input: rows_per_group = M, group_stride = 0)
int ds6g_small_linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int N, int K,
                          int rows_per_group, long group_stride, int relu, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(x && w && y && rows_per_group > 0);
    hipLaunchKernelGGL(small_linear_fwd_kernel, dim3(cdiv((long)M * N, 256)), dim3(256), 0, (hipStream_t)stream, x, w,
                       b, y, M, N, K, rows_per_group, group_stride, relu);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

// dx (nullable) has its own row addressing (dx_rows_per_group, dx_group_stride)
int ds6g_small_linear_bwd(const float* dy, const float* y_mask, const float* x, const float* w, float* dx, float* dw,
                          float* db, int M, int N, int K, int rows_per_group, long group_stride,
                          int dx_rows_per_group, long dx_group_stride, int accumulate_dx, int accumulate_params,
                          void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(dy && x && w && dw && db && rows_per_group > 0);
    if (dx) {
        DS6G_CHECK_ARG(dx_rows_per_group > 0);
        hipLaunchKernelGGL(small_linear_dgrad_kernel, dim3(cdiv((long)M * K, 256)), dim3(256), 0, (hipStream_t)stream,
                           dy, y_mask, w, dx, M, N, K, dx_rows_per_group, dx_group_stride, accumulate_dx);
        DS6G_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(small_linear_wgrad_kernel, dim3(cdiv((long)N * (K + 1), 256)), dim3(256), 0,
                       (hipStream_t)stream, dy, y_mask, x, dw, db, M, N, K, rows_per_group, group_stride,
                       accumulate_params);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

// dst (bf16) = src (fp32), n % 4 == 0, both 8-byte aligned
int ds6g_cast_f32_bf16(const float* src, void* dst, long n, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(src && dst && n > 0 && n % 4 == 0);
    const long n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (__bf16*)dst, n4);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

// Multi-GPU rehearsal on one GPU (tests/test_dp_gpu.py, tools/coresidency.py): `workgroups` workgroups of 256 threads, each
// holding `lds_bytes` of LDS, stay resident for `microseconds` (100 MHz real-time counter; every wave leaves when the time is
// up, so the grid always drains) - a stand-in for the channel workgroups a collective-communication kernel keeps on a few CUs
// during the backward pass.  With lds_bytes > 32 KiB a persistent winograd_pc_kernel workgroup (128 KiB of LDS) cannot share
// the CU.
int ds6g_debug_occupy_cus(int workgroups, int lds_bytes, int microseconds, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(workgroups > 0 && workgroups <= 256 && lds_bytes >= 0 && lds_bytes <= 64 * 1024 && microseconds > 0 &&
                   microseconds <= 2000000);
    hipLaunchKernelGGL(occupy_kernel, dim3(workgroups), dim3(256), (size_t)lds_bytes, (hipStream_t)stream,
                       (unsigned long long)microseconds * 100ull);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

// dev_ptr: one uint64 in device memory (or NULL to switch the mechanism off) added to the dropout counter offset of every
// launch this thread makes from now on - read by the kernels when they run, not when they are launched
int ds6g_set_dropout_salt(const uint64_t* dev_ptr) {
    g_ds6g_salt = dev_ptr;
    return DS6G_OK;
}

int ds6g_version(void) { return 2; }

int ds6g_set_compute_mode(int mode) {
    if (mode < 0 || mode > 3) return DS6G_ERR_ARG;
    g_ds6g_bf16 = mode;
    return 0;
}
int ds6g_get_compute_mode(void) { return g_ds6g_bf16; }

}  // extern "C"
