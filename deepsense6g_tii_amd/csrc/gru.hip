// Autoregressive GRU beam-sequence head of the 30->5 variant (/root/reference/model2_seq_30to5.py:831-862):
//     x = 0;  h = join(fused)                      (B, 64)
//     for t in range(pred_len):  h = GRUCell(x, h);  x = x + Linear(h);  out[:, t] = x
// nn.GRUCell(64, 64) (gate rows ordered r, z, n in weight_ih / weight_hh) and nn.Linear(64, 64).
// The whole recurrence of one sample is ONE workgroup (B workgroups, 192 threads = one per gate row): 12 x 5 tiny
// mat-vecs are latency-bound, not worth a GEMM launch each.  The backward kernel walks the recurrence in reverse and
// writes each sample's parameter-gradient contribution to its own slab; the caller sums the slabs over the batch
// with ds6g_batch_sum (deterministic, no float atomics).
#include "common.h"

namespace {

constexpr int GH = 64;          // hidden = input = output width (model2_seq_30to5.py:842-843)
constexpr int G3 = 3 * GH;
// per-step record: r, z, n, (W_hn h + b_hn), h_t, x_{t-1}
constexpr int GRU_SAVED = 6 * GH;
// slab layout (floats): dW_ih [3H][H], dW_hh [3H][H], db_ih [3H], db_hh [3H], dW_out [H][H], db_out [H]
constexpr int OFF_WIH = 0, OFF_WHH = G3 * GH, OFF_BIH = 2 * G3 * GH, OFF_BHH = OFF_BIH + G3, OFF_WOUT = OFF_BHH + G3,
              OFF_BOUT = OFF_WOUT + GH * GH, GRU_NPARAM = OFF_BOUT + GH;

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

__global__ __launch_bounds__(G3) void gru_head_fwd_kernel(const float* __restrict__ h0, const float* __restrict__ w_ih,
                                                          const float* __restrict__ w_hh, const float* __restrict__ b_ih,
                                                          const float* __restrict__ b_hh, const float* __restrict__ w_out,
                                                          const float* __restrict__ b_out, float* __restrict__ pred,
                                                          float* __restrict__ saved, int T) {
    __shared__ float x[GH], h[GH], gi[G3], gh[G3];
    const int b = blockIdx.x, j = threadIdx.x;
    if (j < GH) { x[j] = 0.f; h[j] = h0[b * GH + j]; }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        float ai = b_ih[j], ah = b_hh[j];
        const float* wi = w_ih + j * GH;
        const float* wh = w_hh + j * GH;
#pragma unroll 8
        for (int c = 0; c < GH; ++c) { ai += wi[c] * x[c]; ah += wh[c] * h[c]; }
        gi[j] = ai; gh[j] = ah;
        __syncthreads();
        float* sv = saved ? saved + ((size_t)b * T + t) * GRU_SAVED : nullptr;
        float hn = 0.f;
        if (j < GH) {
            const float r = sigmoidf_(gi[j] + gh[j]);
            const float z = sigmoidf_(gi[GH + j] + gh[GH + j]);
            const float n = tanhf(gi[2 * GH + j] + r * gh[2 * GH + j]);
            hn = (1.f - z) * n + z * h[j];
            if (sv) { sv[j] = r; sv[GH + j] = z; sv[2 * GH + j] = n; sv[3 * GH + j] = gh[2 * GH + j]; sv[4 * GH + j] = hn; sv[5 * GH + j] = x[j]; }
        }
        __syncthreads();
        if (j < GH) h[j] = hn;
        __syncthreads();
        if (j < GH) {
            float d = b_out[j];
            const float* wo = w_out + j * GH;
#pragma unroll 8
            for (int c = 0; c < GH; ++c) d += wo[c] * h[c];
            const float xn = x[j] + d;  // x is only read by the dot products at the top of the step (behind barriers)
            pred[((size_t)b * T + t) * GH + j] = xn;
            x[j] = xn;
        }
        __syncthreads();
    }
}

// dpred: [B][T][H] gradient of the outputs.  Writes dh0 [B][H] and the sample's parameter-gradient slab.
__global__ __launch_bounds__(G3) void gru_head_bwd_kernel(const float* __restrict__ dpred, const float* __restrict__ h0,
                                                          const float* __restrict__ saved, const float* __restrict__ w_ih,
                                                          const float* __restrict__ w_hh, const float* __restrict__ w_out,
                                                          float* __restrict__ dh0, float* __restrict__ slabs, int T) {
    __shared__ float gx[GH], ghc[GH], gh[GH], gi_[G3], ghg[G3], hprev[GH], xin[GH], ht[GH], tmpx[GH], tmph[GH];
    const int b = blockIdx.x, j = threadIdx.x;
    float* slab = slabs + (size_t)b * GRU_NPARAM;
    // thread j owns gate row j of dW_ih / dW_hh (and row j of dW_out when j < H); the rows accumulate in the sample's
    // (L2-resident) slab across the T steps
    for (int i = j; i < GRU_NPARAM; i += G3) slab[i] = 0.f;
    if (j < GH) { gx[j] = 0.f; ghc[j] = 0.f; }
    __syncthreads();
    for (int t = T - 1; t >= 0; --t) {
        const float* sv = saved + ((size_t)b * T + t) * GRU_SAVED;
        if (j < GH) {
            gx[j] += dpred[((size_t)b * T + t) * GH + j];       // out_t = x_t
            ht[j] = sv[4 * GH + j];
            xin[j] = sv[5 * GH + j];
            hprev[j] = t > 0 ? (sv - GRU_SAVED)[4 * GH + j] : h0[b * GH + j];
        }
        __syncthreads();
        // x_t = x_{t-1} + W_out h_t + b_out  ->  g_d = gx ;  dW_out += g_d (x) h_t ; gh = W_out^T g_d + carry
        if (j < GH) {
            float acc = ghc[j];
            for (int o = 0; o < GH; ++o) acc += w_out[o * GH + j] * gx[o];
            gh[j] = acc;
            slab[OFF_BOUT + j] += gx[j];
            float* dwo = slab + OFF_WOUT + j * GH;
            const float g = gx[j];
            for (int c = 0; c < GH; ++c) dwo[c] += g * ht[c];
        }
        __syncthreads();
        // GRU cell backward
        if (j < GH) {
            const float r = sv[j], z = sv[GH + j], n = sv[2 * GH + j], hnl = sv[3 * GH + j];
            const float g = gh[j];
            const float gn_pre = g * (1.f - z) * (1.f - n * n);
            const float gz_pre = g * (hprev[j] - n) * z * (1.f - z);
            const float gr_pre = gn_pre * hnl * r * (1.f - r);
            gi_[j] = gr_pre; gi_[GH + j] = gz_pre; gi_[2 * GH + j] = gn_pre;
            ghg[j] = gr_pre; ghg[GH + j] = gz_pre; ghg[2 * GH + j] = gn_pre * r;
            ghc[j] = g * z;  // direct path h' <- h
        }
        __syncthreads();
        {   // parameter gradients of gate row j
            const float a = gi_[j], c2 = ghg[j];
            slab[OFF_BIH + j] += a;
            slab[OFF_BHH + j] += c2;
            float* dwi = slab + OFF_WIH + j * GH;
            float* dwh = slab + OFF_WHH + j * GH;
            for (int c = 0; c < GH; ++c) { dwi[c] += a * xin[c]; dwh[c] += c2 * hprev[c]; }
        }
        if (j < GH) {  // input / previous-hidden gradients (transposed mat-vecs, coalesced over j)
            float ax = 0.f, ah = 0.f;
            for (int o = 0; o < G3; ++o) { ax += w_ih[o * GH + j] * gi_[o]; ah += w_hh[o * GH + j] * ghg[o]; }
            tmpx[j] = ax;
            tmph[j] = ah;
        }
        __syncthreads();
        if (j < GH) {
            ghc[j] += tmph[j];
            gx[j] += tmpx[j];   // x_{t-1} feeds x_t directly (residual: gx stays) and the cell input
        }
        __syncthreads();
    }
    if (j < GH) dh0[b * GH + j] = ghc[j];
}

}  // namespace

extern "C" {

size_t ds6g_gru_head_saved_floats(int B, int T) { return (size_t)B * T * GRU_SAVED; }
size_t ds6g_gru_head_slab_floats(void) { return (size_t)GRU_NPARAM; }

int ds6g_gru_head_fwd(const float* h0, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                      const float* w_out, const float* b_out, float* pred, float* saved, int B, int T, int H,
                      void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(h0 && w_ih && w_hh && b_ih && b_hh && w_out && b_out && pred && B > 0 && T > 0 && H == GH);
    hipLaunchKernelGGL(gru_head_fwd_kernel, dim3(B), dim3(G3), 0, (hipStream_t)stream, h0, w_ih, w_hh, b_ih, b_hh, w_out,
                       b_out, pred, saved, T);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

int ds6g_gru_head_bwd(const float* dpred, const float* h0, const float* saved, const float* w_ih, const float* w_hh,
                      const float* w_out, float* dh0, float* slabs, int B, int T, int H, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(dpred && h0 && saved && w_ih && w_hh && w_out && dh0 && slabs && B > 0 && T > 0 && H == GH);
    hipLaunchKernelGGL(gru_head_bwd_kernel, dim3(B), dim3(G3), 0, (hipStream_t)stream, dpred, h0, saved, w_ih, w_hh, w_out,
                       dh0, slabs, T);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

}  // extern "C"
