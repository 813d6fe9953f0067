// Fused multi-head self-attention (no mask, softmax over all T keys, dropout on the
// probabilities) forward and backward, fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32.
// The T x T score matrix is never written to HBM (the reference materialises it three times:
// model2_seq.py:102-105).  q/k/v/o are token-major [B*T][ld] with head h at columns h*HD..h*HD+HD-1,
// exactly the layout the QKV linears produce, so no head transpose is needed (:97-99,:106).
//
// Register-direct operand trick: scores are computed TRANSPOSED, S^T = K Q^T, so an accumulator
// register r of lane (q = lane&31, half = lane>>5) holds key (r&3)+8(r>>2)+4*half of query q.
// That is exactly the B-operand shape (B[k = half][j = q]) of the next product
// O^T += V^T P^T when MFMA step r pairs the two keys {kA(r), kA(r)+4}: P never moves between
// lanes or through LDS, and the per-query softmax state (m, l) lives on the lane of its query.
// K/V tiles (32 keys) are staged in LDS row-major with an odd row stride (HD+1): both the
// "key on the lane" read (stride HD+1, odd -> conflict-free) and the "dim on the lane" read
// (consecutive floats) are conflict-free ds_read_b32 from the same image.
#include "common.h"

namespace {

struct AttnParams {
    const float* q;
    const float* k;
    const float* v;
    float* o;            // fwd: output; bwd: forward output (read)
    float* lse;          // [B][nh][T]
    const float* d_o;    // bwd
    float* delta;        // [B][nh][T]  rowsum(dO * O)
    float* dq;
    float* dk;
    float* dv;
    int T, nh;
    int ld;              // row stride (floats) of q/k/v/o/do/dq/dk/dv
    float scale;
    uint32_t thr;
    float dscale;
    uint64_t seed;
    uint64_t seed_off;
};

__device__ __forceinline__ int krow16(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

template <int HD, int NTHR>
struct TileRegs {
    static constexpr int CNT = (8 * HD + NTHR - 1) / NTHR;  // float4 per thread for a 32 x HD tile
    f32x4 v[CNT];
};

// global [32 rows][HD] (rows row0.., zero beyond T) -> registers
template <int HD, int NTHR>
__device__ __forceinline__ void tile_load(TileRegs<HD, NTHR>& t, const float* base, int ld, int row0, int T, int tid) {
#pragma unroll
    for (int i = 0; i < TileRegs<HD, NTHR>::CNT; ++i) {
        const int idx = tid + i * NTHR;
        const int row = idx / (HD / 4);
        const int c4 = (idx % (HD / 4)) * 4;
        const bool ok = idx < 8 * HD && (row0 + row) < T;
        t.v[i] = ok ? *reinterpret_cast<const f32x4*>(base + (long)(row0 + row) * ld + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

template <int HD, int NTHR>
__device__ __forceinline__ void tile_store(const TileRegs<HD, NTHR>& t, float* lds, int tid) {
#pragma unroll
    for (int i = 0; i < TileRegs<HD, NTHR>::CNT; ++i) {
        const int idx = tid + i * NTHR;
        if (idx < 8 * HD) {
            const int row = idx / (HD / 4);
            const int c4 = (idx % (HD / 4)) * 4;
            float* d = lds + row * (HD + 1) + c4;
            d[0] = t.v[i][0]; d[1] = t.v[i][1]; d[2] = t.v[i][2]; d[3] = t.v[i][3];
        }
    }
}

// write an accumulator set acc[blk][16] = X^T[d][row] to global X[row][d] (row on the lane)
template <int HD>
__device__ __forceinline__ void store_rows(const f32x16* acc, float* base, int ld, int row, int T, int half, float mul) {
    constexpr int NB = (HD + 31) / 32;
    if (row >= T) return;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = blk * 32 + 8 * g + 4 * half;
            if (d < HD) {
                f32x4 v = {acc[blk][4 * g] * mul, acc[blk][4 * g + 1] * mul, acc[blk][4 * g + 2] * mul,
                           acc[blk][4 * g + 3] * mul};
                *reinterpret_cast<f32x4*>(base + (long)row * ld + d) = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
template <int HD, int NW>
__global__ __launch_bounds__(NW * 64) void attn_fwd_kernel(const AttnParams p) {
    constexpr int NTHR = NW * 64;
    constexpr int LDS_LD = HD + 1;
    constexpr int NB = (HD + 31) / 32;
    __shared__ float Ks[32 * LDS_LD];
    __shared__ float Vs[32 * LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int T = p.T;
    const int q_row = blockIdx.x * (32 * NW) + wave * 32 + l31;
    const long head_off = (long)b * T * p.ld + h * HD;
    const float* qb = p.q + head_off;
    const float* kb = p.k + head_off;
    const float* vb = p.v + head_off;

    float qreg[HD / 2];
    {
        const int qr = q_row < T ? q_row : T - 1;
#pragma unroll
        for (int ss = 0; ss < HD / 2; ++ss) qreg[ss] = qb[(long)qr * p.ld + 2 * ss + half] * p.scale;
    }
    f32x16 oacc[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[blk][r] = 0.f;
    float m = -INFINITY, l = 0.f;

    const int ntiles = (T + 31) / 32;
    TileRegs<HD, NTHR> kr, vr;
    tile_load<HD, NTHR>(kr, kb, p.ld, 0, T, tid);
    tile_load<HD, NTHR>(vr, vb, p.ld, 0, T, tid);
    const long drop_row = ((long)(b * p.nh + h) * T + q_row) * T;
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();
        tile_store<HD, NTHR>(kr, Ks, tid);
        tile_store<HD, NTHR>(vr, Vs, tid);
        __syncthreads();
        if (kt + 1 < ntiles) {
            tile_load<HD, NTHR>(kr, kb, p.ld, (kt + 1) * 32, T, tid);
            tile_load<HD, NTHR>(vr, vb, p.ld, (kt + 1) * 32, T, tid);
        }
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ss = 0; ss < HD / 2; ++ss)
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[l31 * LDS_LD + 2 * ss + half], qreg[ss], s, 0, 0, 0);
        const int key0 = kt * 32;
        float mloc = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (key0 + krow16(r, half) >= T) s[r] = -INFINITY;
            mloc = fmaxf(mloc, s[r]);
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float mnew = fmaxf(m, mloc);
        const float alpha = __expf(m - mnew);
        float lsum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __expf(s[r] - mnew);
            lsum += s[r];
        }
        l = l * alpha + lsum;
        m = mnew;
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[blk][r] *= alpha;
        if (p.thr) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key0 + krow16(r, half);
                s[r] = ds6g_keep(p.seed, p.seed_off + (uint64_t)(drop_row + key), p.thr) ? s[r] * p.dscale : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* vrow = Vs + krow16(r, half) * LDS_LD;
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                const int d = blk * 32 + l31;
                oacc[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[d < HD ? d : HD - 1], s[r], oacc[blk], 0, 0, 0);
            }
        }
    }
    const float ltot = l + __shfl_xor(l, 32, 64);
    store_rows<HD>(oacc, p.o + head_off, p.ld, q_row, T, half, 1.0f / ltot);
    if (half == 0 && q_row < T) p.lse[(long)(b * p.nh + h) * T + q_row] = m + __logf(ltot);
}

// ------------------------------------------------------------------------------------------------
// dQ (and delta) : one wave = 32 queries, loop over key tiles
template <int HD, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dq_kernel(const AttnParams p) {
    constexpr int NTHR = NW * 64;
    constexpr int LDS_LD = HD + 1;
    constexpr int NB = (HD + 31) / 32;
    __shared__ float Ks[32 * LDS_LD];
    __shared__ float Vs[32 * LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int T = p.T;
    const int q_row = blockIdx.x * (32 * NW) + wave * 32 + l31;
    const bool q_ok = q_row < T;
    const long head_off = (long)b * T * p.ld + h * HD;
    const float* kb = p.k + head_off;
    const float* vb = p.v + head_off;

    float qreg[HD / 2], doreg[HD / 2];
    float delta = 0.f;
    {
        const long ro = head_off + (long)(q_ok ? q_row : T - 1) * p.ld;
#pragma unroll
        for (int ss = 0; ss < HD / 2; ++ss) {
            const int d = 2 * ss + half;
            qreg[ss] = p.q[ro + d] * p.scale;
            doreg[ss] = q_ok ? p.d_o[ro + d] : 0.f;
            delta += doreg[ss] * p.o[ro + d];
        }
        delta += __shfl_xor(delta, 32, 64);
    }
    const long stat_idx = (long)(b * p.nh + h) * T + q_row;
    const float lse = q_ok ? p.lse[stat_idx] : INFINITY;
    if (half == 0 && q_ok) p.delta[stat_idx] = delta;

    f32x16 dq[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[blk][r] = 0.f;

    const int ntiles = (T + 31) / 32;
    TileRegs<HD, NTHR> kr, vr;
    tile_load<HD, NTHR>(kr, kb, p.ld, 0, T, tid);
    tile_load<HD, NTHR>(vr, vb, p.ld, 0, T, tid);
    const long drop_row = ((long)(b * p.nh + h) * T + q_row) * T;
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();
        tile_store<HD, NTHR>(kr, Ks, tid);
        tile_store<HD, NTHR>(vr, Vs, tid);
        __syncthreads();
        if (kt + 1 < ntiles) {
            tile_load<HD, NTHR>(kr, kb, p.ld, (kt + 1) * 32, T, tid);
            tile_load<HD, NTHR>(vr, vb, p.ld, (kt + 1) * 32, T, tid);
        }
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ss = 0; ss < HD / 2; ++ss) {
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[l31 * LDS_LD + 2 * ss + half], qreg[ss], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[l31 * LDS_LD + 2 * ss + half], doreg[ss], dp, 0, 0, 0);
        }
        const int key0 = kt * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + krow16(r, half);
            const float pr = (key < T) ? __expf(s[r] - lse) : 0.f;
            float g = dp[r];
            if (p.thr) g = ds6g_keep(p.seed, p.seed_off + (uint64_t)(drop_row + key), p.thr) ? g * p.dscale : 0.f;
            s[r] = pr * (g - delta) * p.scale;  // dS (scaled)
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* krow = Ks + krow16(r, half) * LDS_LD;
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                const int d = blk * 32 + l31;
                dq[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(krow[d < HD ? d : HD - 1], s[r], dq[blk], 0, 0, 0);
            }
        }
    }
    store_rows<HD>(dq, p.dq + head_off, p.ld, q_row, T, half, 1.0f);
}

// dK, dV : one wave = 32 keys (key on the lane), loop over query tiles
template <int HD, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dkv_kernel(const AttnParams p) {
    constexpr int NTHR = NW * 64;
    constexpr int LDS_LD = HD + 1;
    constexpr int NB = (HD + 31) / 32;
    __shared__ float Qs[32 * LDS_LD];
    __shared__ float Os[32 * LDS_LD];  // dO tile
    __shared__ float lse_s[32];
    __shared__ float delta_s[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int T = p.T;
    const int key = blockIdx.x * (32 * NW) + wave * 32 + l31;
    const bool key_ok = key < T;
    const long head_off = (long)b * T * p.ld + h * HD;
    const float* qb = p.q + head_off;
    const float* dob = p.d_o + head_off;

    float kreg[HD / 2], vreg[HD / 2];
    {
        const long ro = head_off + (long)(key_ok ? key : T - 1) * p.ld;
#pragma unroll
        for (int ss = 0; ss < HD / 2; ++ss) {
            kreg[ss] = p.k[ro + 2 * ss + half];
            vreg[ss] = p.v[ro + 2 * ss + half];
        }
    }
    f32x16 dk[NB], dv[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[blk][r] = 0.f; dv[blk][r] = 0.f; }

    const int ntiles = (T + 31) / 32;
    const long stat_base = (long)(b * p.nh + h) * T;
    TileRegs<HD, NTHR> qr, dr;
    tile_load<HD, NTHR>(qr, qb, p.ld, 0, T, tid);
    tile_load<HD, NTHR>(dr, dob, p.ld, 0, T, tid);
    float lse_r = 0.f, delta_r = 0.f;
    if (tid < 32) {
        lse_r = tid < T ? p.lse[stat_base + tid] : INFINITY;
        delta_r = tid < T ? p.delta[stat_base + tid] : 0.f;
    }
    for (int qt = 0; qt < ntiles; ++qt) {
        __syncthreads();
        tile_store<HD, NTHR>(qr, Qs, tid);
        tile_store<HD, NTHR>(dr, Os, tid);
        if (tid < 32) { lse_s[tid] = lse_r; delta_s[tid] = delta_r; }
        __syncthreads();
        if (qt + 1 < ntiles) {
            tile_load<HD, NTHR>(qr, qb, p.ld, (qt + 1) * 32, T, tid);
            tile_load<HD, NTHR>(dr, dob, p.ld, (qt + 1) * 32, T, tid);
            if (tid < 32) {
                const int qn = (qt + 1) * 32 + tid;
                lse_r = qn < T ? p.lse[stat_base + qn] : INFINITY;
                delta_r = qn < T ? p.delta[stat_base + qn] : 0.f;
            }
        }
        // S[q][key], dP[q][key]: query rows in registers, key on the lane
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ss = 0; ss < HD / 2; ++ss) {
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[l31 * LDS_LD + 2 * ss + half], kreg[ss], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Os[l31 * LDS_LD + 2 * ss + half], vreg[ss], dp, 0, 0, 0);
        }
        const int q0 = qt * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ql = krow16(r, half);
            const int qg = q0 + ql;
            float pr = __expf(s[r] * p.scale - lse_s[ql]);  // lse = +inf for q >= T -> 0
            float g = dp[r];
            if (p.thr) {
                const bool keep = ds6g_keep(p.seed, p.seed_off + (uint64_t)((stat_base + qg) * (long)T + key), p.thr);
                g = keep ? g * p.dscale : 0.f;
                dp[r] = pr * (g - delta_s[ql]) * p.scale;
                pr = keep ? pr * p.dscale : 0.f;
            } else {
                dp[r] = pr * (g - delta_s[ql]) * p.scale;
            }
            s[r] = pr;  // dropped probabilities
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* orow = Os + krow16(r, half) * LDS_LD;
            const float* qrow = Qs + krow16(r, half) * LDS_LD;
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                const int d = blk * 32 + l31;
                const int dc = d < HD ? d : HD - 1;
                dv[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(orow[dc], s[r], dv[blk], 0, 0, 0);
                dk[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(qrow[dc], dp[r], dk[blk], 0, 0, 0);
            }
        }
    }
    store_rows<HD>(dk, p.dk + head_off, p.ld, key, T, half, 1.0f);
    store_rows<HD>(dv, p.dv + head_off, p.ld, key, T, half, 1.0f);
}

template <int KIND, int HD>
int launch_attn(const AttnParams& p, int B, hipStream_t st) {
    // 64-query (2-wave) blocks when 128-query blocks would leave the 256 CUs under-filled
    const long blocks4 = (long)cdiv(p.T, 128) * p.nh * B;
    // the 2-wave dK/dV kernel at HD=128 would spill (K,V fragments + two accumulator sets)
    const bool four = blocks4 >= 512 || (KIND == 2 && HD == 128);
    if (four) {
        dim3 grid(cdiv(p.T, 128), p.nh, B), block(256);
        if (KIND == 0) hipLaunchKernelGGL((attn_fwd_kernel<HD, 4>), grid, block, 0, st, p);
        if (KIND == 1) hipLaunchKernelGGL((attn_bwd_dq_kernel<HD, 4>), grid, block, 0, st, p);
        if (KIND == 2) hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD, 4>), grid, block, 0, st, p);
    } else {
        dim3 grid(cdiv(p.T, 64), p.nh, B), block(128);
        if (KIND == 0) hipLaunchKernelGGL((attn_fwd_kernel<HD, 2>), grid, block, 0, st, p);
        if (KIND == 1) hipLaunchKernelGGL((attn_bwd_dq_kernel<HD, 2>), grid, block, 0, st, p);
        if (KIND == 2) hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD, 2>), grid, block, 0, st, p);
    }
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

template <int KIND>
int dispatch_hd(const AttnParams& p, int B, int hd, hipStream_t st) {
    switch (hd) {
        case 16: return launch_attn<KIND, 16>(p, B, st);
        case 32: return launch_attn<KIND, 32>(p, B, st);
        case 64: return launch_attn<KIND, 64>(p, B, st);
        case 128: return launch_attn<KIND, 128>(p, B, st);
        default:
            fprintf(stderr, "[ds6g] attention head dim %d not supported (16/32/64/128)\n", hd);
            return DS6G_ERR_ARG;
    }
}

}  // namespace

extern "C" {

// o = dropout(softmax(q k^T / sqrt(hd))) v ; lse[b][h][t] = logsumexp of the scaled scores
int ds6g_attention_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int B, int T, int nh,
                       int hd, int ld, float drop_p, uint64_t seed, uint64_t seed_off, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(q && k && v && o && lse && B > 0 && T > 0 && ld % 4 == 0 && ld >= nh * hd);
    DS6G_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f);
    AttnParams p{};
    p.q = q; p.k = k; p.v = v; p.o = o; p.lse = lse; p.T = T; p.nh = nh; p.ld = ld;
    p.scale = 1.0f / sqrtf((float)hd);
    p.thr = ds6g_drop_threshold(drop_p); p.dscale = 1.f / (1.f - drop_p); p.seed = seed; p.seed_off = seed_off;
    return dispatch_hd<0>(p, B, hd, (hipStream_t)stream);
}

// gradients of the above; delta is a [B][nh][T] scratch (rowsum(dO*O)), written then read
int ds6g_attention_bwd(const float* q, const float* k, const float* v, const float* o, const float* d_o,
                       const float* lse, float* delta, float* dq, float* dk, float* dv, int B, int T, int nh, int hd,
                       int ld, float drop_p, uint64_t seed, uint64_t seed_off, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(q && k && v && o && d_o && lse && delta && dq && dk && dv && ld % 4 == 0 && ld >= nh * hd);
    AttnParams p{};
    p.q = q; p.k = k; p.v = v; p.o = const_cast<float*>(o); p.lse = const_cast<float*>(lse); p.d_o = d_o;
    p.delta = delta; p.dq = dq; p.dk = dk; p.dv = dv; p.T = T; p.nh = nh; p.ld = ld;
    p.scale = 1.0f / sqrtf((float)hd);
    p.thr = ds6g_drop_threshold(drop_p); p.dscale = 1.f / (1.f - drop_p); p.seed = seed; p.seed_off = seed_off;
    int rc = dispatch_hd<1>(p, B, hd, (hipStream_t)stream);
    if (rc) return rc;
    return dispatch_hd<2>(p, B, hd, (hipStream_t)stream);
}

}  // extern "C"
