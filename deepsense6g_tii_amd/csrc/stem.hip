// The three 7x7 / stride 2 / pad 3 stem convolutions (torchvision ResNet conv1, called from model2_seq.py:495,500,505) on
// bf16-STORED tensors - the last piece of the bf16 configuration (BASELINE configs[1] / [4]) that still ran on fp32 storage.
// The input has 4 channels (image 3 -> 4, LiDAR 1 -> 4, radar 2 -> 4, zero padded), so an im2col row is 7 x 7 x 4 = 196
// deep - too narrow for bgemm.hip's wave-uniform walk (64 channels per tap).  Instead:
//   * a workgroup owns 8 x 16 output pixels; their 21 x 38-pixel input patch (6.4 KB of bf16) goes to LDS ONCE by LDS-DMA
//     (16-byte pieces = pixel PAIRS, out-of-image pairs fall off the buffer descriptor: zero padding for free), instead of
//     being re-fetched per tap;
//   * the filter's horizontal taps are padded on the LEFT to 8 (tap s' = s + 1, s' = 0 is a zero tap): a tap PAIR x 4 channels
//     is then exactly one 16-byte piece at an even pixel column - the MFMA A fragment of v_mfma_f32_32x32x16_bf16 (8 k values
//     per lane) is one ds_read_b128 at (patch row 2 oy + r, pair ox + j'), conflict-free (16 consecutive pieces per group);
//     K = 7 rows x 8 taps x 4 channels = 224 = 14 MFMA k-steps (196 useful);
//   * forward: the 28 weight fragments of a wave (2 x 32 output channels x 14 k-steps) live in registers for the whole
//     persistent workgroup; the epilogue rounds to bf16, stores 16-byte row pieces through a wave-private LDS patch and
//     accumulates the train-mode BatchNorm statistics of the STORED values (as bgemm.hip's fused statistics);
//   * weight gradient: GEMM over pixels, dW[o][r][(s', c)] = sum_p dy[p][o] * patch[p][r][(s', c)].  Both operands are
//     row-contiguous in LDS ([pixel][64 channels] and the raw patch, where the 32 values (s', c) of pixel p and row r are 64
//     contiguous bytes) and are read TRANSPOSED with ds_read_b64_tr_b16, exactly as bgemm.hip's wgrad; per-workgroup fp32
//     slabs, one deterministic reduction kernel that also drops the padding (tap 0, channels >= cin).
// HBM-bound by construction: 31 MB of input + 126 MB of output per trunk at N = 60 (fp32 storage: 63 + 252).
#include "common.h"

// norm.hip: mean / invstd (+ running statistics) from nblk row-block partials [nblk][2][C] (fp64)
int ds6g_internal_bn_stats_finalize(const double* partial, int nblk, long M, int C, float eps, float momentum, float* mean,
                                    float* invstd, float* running_mean, float* running_var, hipStream_t st);

namespace {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int TH = 8, TW = 16;              // output pixels per workgroup tile
constexpr int PR = 2 * TH + 5;              // 21 patch rows   (input rows 2 oy0 - 3 .. 2 oy0 + 17)
constexpr int PP = TW + 3;                  // 19 pixel pairs  (input columns 2 ox0 - 4 .. 2 ox0 + 33)
constexpr int PIECES = PR * PP;             // 399 16-byte pieces
constexpr int PATCH_BYTES = 8192;           // 2 DMA rounds of 256 lanes x 16 B
constexpr int KO = 64;                      // output channels
constexpr int WROW = 7 * 8 * 4;             // padded filter row: [r 7][s' 8][c 4] = 224 bf16 per output channel

__device__ __forceinline__ unsigned lds_off(const void* p) { return (unsigned)(unsigned long)(lds_void*)p; }

// fp32 master filter [64][7][7][cin] (OHWI, the arena layout) -> bf16 [64][7][8][4]: tap s' = s + 1, tap 0 and channels >= cin zero
__global__ __launch_bounds__(256) void stem_pack_weights_kernel(const float* __restrict__ w, __bf16* __restrict__ w16, int cin) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= KO * WROW) return;
    const int c = i & 3, sp = (i >> 2) & 7, r = (i >> 5) % 7, o = i / WROW;
    float v = 0.f;
    if (sp >= 1 && c < cin) v = w[((o * 7 + r) * 7 + (sp - 1)) * cin + c];
    w16[i] = (__bf16)v;
}

struct StemParams {
    const __bf16* x;      // [N][H][W][4]
    const __bf16* w16;    // [64][7][8][4]
    __bf16* y;            // [N][Ho][Wo][64]      (forward)
    const __bf16* dy;     // [N][Ho][Wo][64]      (weight gradient)
    float* slabs;         // [grid][64][7][32]    (weight gradient)
    double* bn_partial;   // [grid][2][64]        (forward)
    int N, H, W, Ho, Wo, tiles_x, tiles_y, ntiles;
    unsigned x_bytes, dy_bytes;
};

// lane -> its two patch pieces (piece = k * 256 + tid): patch row and pixel pair, 0xffff.. when beyond the patch
struct PatchLane {
    int row[2], cp[2];
    __device__ __forceinline__ explicit PatchLane(int tid) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int p = k * 256 + tid;
            row[k] = p < PIECES ? p / PP : -1000000;
            cp[k] = p % PP;
        }
    }
    // DMA of the patch of tile (n, ty, tx) into `dst`; out-of-image pairs / rows point past the descriptor (zeros)
    __device__ __forceinline__ void issue(const StemParams& p, const i32x4 srd, unsigned dst, int wave, int n, int ty, int tx) const {
        const int iy0 = 2 * ty * TH - 3, ip0 = tx * TW - 2;
        const int Wp = p.W >> 1;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int iy = iy0 + row[k], ip = ip0 + cp[k];
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ip < (unsigned)Wp;
            const unsigned off = ok ? (unsigned)(((n * p.H + iy) * Wp + ip) * 16) : OOB_OFF;
            dma16(srd, dst + (unsigned)(k * 256 + wave * 64) * 16u, off);
        }
    }
};

__device__ __forceinline__ void tile_coords(const StemParams& p, int t, int& n, int& ty, int& tx) {
    tx = t % p.tiles_x;
    const int q = t / p.tiles_x;
    ty = q % p.tiles_y;
    n = q / p.tiles_y;
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void stem_fwd_kernel(const StemParams p) {
    __shared__ __attribute__((aligned(1024))) unsigned char patch[2][PATCH_BYTES];
    __shared__ __attribute__((aligned(16))) __bf16 outp[4][32 * KO];     // wave-private output patches [32 px][64 ch]
    __shared__ float red[4][2][KO];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, khalf = lane >> 5;
    const i32x4 x_srd = make_srd(p.x, p.x_bytes);
    const PatchLane pl(tid);

    // the wave's 28 weight fragments: output channel nt * 32 + l31, k-step (r, jj): taps s' = 4 jj + 2 khalf, +1 (x 4 channels)
    bf16x8 bw[7][2][2];
#pragma unroll
    for (int r = 0; r < 7; ++r)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                bw[r][jj][nt] = *reinterpret_cast<const bf16x8*>(p.w16 + (nt * 32 + l31) * WROW + (r * 8 + 4 * jj + 2 * khalf) * 4);

    // this lane's pixel of the wave's 32-pixel M tile (output rows 2 wave, 2 wave + 1 of the tile): piece (2 oy, ox) of the patch
    const int oy_rel = 2 * wave + (l31 >> 4), ox = l31 & 15;
    const unsigned a_base = (unsigned)((2 * oy_rel * PP + ox + khalf) * 16);   // + (r * PP + 2 jj) * 16 per k-step

    float cs[2] = {0.f, 0.f}, cq[2] = {0.f, 0.f};
    int buf = 0;
    int t = blockIdx.x;
    if (t < p.ntiles) {
        int n, ty, tx;
        tile_coords(p, t, n, ty, tx);
        pl.issue(p, x_srd, lds_off(patch[0]), wave, n, ty, tx);
    }
    for (; t < p.ntiles; t += gridDim.x) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();     // patch[buf] is complete; every wave is done reading patch[buf ^ 1]
        int n, ty, tx;
        tile_coords(p, t, n, ty, tx);
        if (t + (int)gridDim.x < p.ntiles) {
            int n2, ty2, tx2;
            tile_coords(p, t + gridDim.x, n2, ty2, tx2);
            pl.issue(p, x_srd, lds_off(patch[buf ^ 1]), wave, n2, ty2, tx2);
        }
        const unsigned char* P = patch[buf];
        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 7; ++r)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(P + a_base + (unsigned)((r * PP + 2 * jj) * 16));
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bw[r][jj][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bw[r][jj][1], acc[1], 0, 0, 0);
            }
        // D[pixel][channel]: channel = nt * 32 + l31, pixel = (reg & 3) + 8 (reg >> 2) + 4 khalf of the wave's 32
        __bf16* op = outp[wave];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int px = (r & 3) + 8 * (r >> 2) + 4 * khalf;
                const __bf16 vb = (__bf16)acc[nt][r];
                op[px * KO + nt * 32 + l31] = vb;
                const float vr = (float)vb;
                cs[nt] += vr;
                cq[nt] += vr * vr;
            }
        // rows of the output: pixel px of the wave = (row oy0 + 2 wave + (px >> 4), column ox0 + (px & 15)), 128 B each
        const int oy0 = ty * TH + 2 * wave, ox0 = tx * TW;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * 64 + lane, px = idx >> 3, pc = idx & 7;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(op + px * KO + pc * 8);
            const size_t o = (((size_t)n * p.Ho + oy0 + (px >> 4)) * p.Wo + ox0 + (px & 15)) * KO + pc * 8;
            *reinterpret_cast<bf16x8*>(p.y + o) = v;
        }
        buf ^= 1;
    }
    // BatchNorm partial sums of this workgroup (all its tiles): lane halves, then the four waves
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        cs[nt] += __shfl_xor(cs[nt], 32, 64);
        cq[nt] += __shfl_xor(cq[nt], 32, 64);
        if (khalf == 0) {
            red[wave][0][nt * 32 + l31] = cs[nt];
            red[wave][1][nt * 32 + l31] = cq[nt];
        }
    }
    __syncthreads();
    if (tid < 2 * KO) {
        const int which = tid >> 6, c = tid & 63;
        p.bn_partial[(size_t)blockIdx.x * 2 * KO + which * KO + c] =
            (double)red[0][which][c] + (double)red[1][which][c] + (double)red[2][which][c] + (double)red[3][which][c];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradient: per workgroup tile, A = dy^T (64 channels x 128 pixels), B = patch im2col (128 pixels x 7 x 32)
__global__ __launch_bounds__(256, 2) void stem_wgrad_kernel(const StemParams p) {
    __shared__ __attribute__((aligned(1024))) unsigned char patch[2][PATCH_BYTES];
    __shared__ __attribute__((aligned(1024))) unsigned char dyt[2][TH * TW * KO * 2];   // [128 px][64 ch] bf16, 128-B rows
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, khalf = lane >> 5;
    const i32x4 x_srd = make_srd(p.x, p.x_bytes), dy_srd = make_srd(p.dy, p.dy_bytes);
    const PatchLane pl(tid);
    // dy tile: 16 DMA pieces of 8 rows x 128 B; the 64-B column segment is XOR-swizzled by (row >> 1) & 1 on the SOURCE side
    // (bgemm.hip's k-major image for 128-byte rows), so that the transposed reads below are conflict-free
    int d_row[4], d_col[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pi = i * 4 + wave;
        d_row[i] = pi * 8 + (lane >> 3);
        const int slot = lane & 7, seg = slot >> 2, within = slot & 3;
        d_col[i] = ((seg ^ ((d_row[i] >> 1) & 1)) * 4 + within) * 8;
    }
    auto issue_dy = [&](unsigned dst, int n, int ty, int tx) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int oy = ty * TH + (d_row[i] >> 4), ox = tx * TW + (d_row[i] & 15);
            const unsigned off = (unsigned)((((n * p.Ho + oy) * p.Wo + ox) * KO + d_col[i]) * 2);
            dma16(dy_srd, dst + (unsigned)(i * 4 + wave) * 1024u, off);
        }
    };
    // transposed-read lane geometry (bgemm.hip): lane 4 q + pp of a 16-lane group addresses k row 8 khalf + 4 t + q of the
    // k-step, columns 16 (group & 1) + 4 pp .. + 3; it receives column (lane & 31) of those four k rows
    const int tq = (lane >> 2) & 3, tpp = lane & 3, tgrp = (lane >> 4) & 1;
    const int mt = wave & 1, rset = wave >> 1;              // output-channel tile; filter rows 0-3 / 4-6
    const int r0 = rset * 4, nr = rset ? 3 : 4;
    auto a_addr = [&](int s, int t) {                        // dy image: row = pixel 16 s + 8 khalf + 4 t + q, col = channel
        const int row = 16 * s + 8 * khalf + 4 * t + tq;
        const int col = mt * 32 + 16 * tgrp + 4 * tpp;
        const int pseg = (col >> 5) ^ ((row >> 1) & 1);
        return (unsigned)(row * 128 + pseg * 64 + (col & 31) * 2);
    };
    auto b_addr = [&](int s, int t, int r) {                 // patch: pixel (row s of the tile, column 8 khalf + 4 t + q), filter row r
        const int oxp = 8 * khalf + 4 * t + tq;
        return (unsigned)(((2 * s + r) * PP + oxp) * 16 + (16 * tgrp + 4 * tpp) * 2);
    };
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    int buf = 0;
    int t = blockIdx.x;
    if (t < p.ntiles) {
        int n, ty, tx;
        tile_coords(p, t, n, ty, tx);
        pl.issue(p, x_srd, lds_off(patch[0]), wave, n, ty, tx);
        issue_dy(lds_off(dyt[0]), n, ty, tx);
    }
    for (; t < p.ntiles; t += gridDim.x) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + (int)gridDim.x < p.ntiles) {
            int n2, ty2, tx2;
            tile_coords(p, t + gridDim.x, n2, ty2, tx2);
            pl.issue(p, x_srd, lds_off(patch[buf ^ 1]), wave, n2, ty2, tx2);
            issue_dy(lds_off(dyt[buf ^ 1]), n2, ty2, tx2);
        }
        const unsigned char* P = patch[buf];
        const unsigned char* D = dyt[buf];
#pragma unroll
        for (int s = 0; s < TH; ++s) {     // k-step s = output row s of the tile (16 pixels)
            const bf16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(lds_void*)(D + a_addr(s, 0)));
            const bf16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(lds_void*)(D + a_addr(s, 1)));
            const bf16x8 a = __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < nr) {
                    const bf16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(lds_void*)(P + b_addr(s, 0, r0 + j)));
                    const bf16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(lds_void*)(P + b_addr(s, 1, r0 + j)));
                    const bf16x8 b = __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
                }
            }
        }
        buf ^= 1;
    }
    // slab[o][r][n]: D[row = channel of the tile][col = (s', c)], col = l31, row = (reg & 3) + 8 (reg >> 2) + 4 khalf
    float* slab = p.slabs + (size_t)blockIdx.x * KO * 7 * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j < nr) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                slab[(o * 7 + r0 + j) * 32 + l31] = acc[j][r];
            }
        }
    }
}

// dw[o][r][s][c] (+)= sum over the workgroup slabs of slab[o][r][(s + 1) * 4 + c]   (tap 0 and channels >= cin are padding)
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ slabs, int nslab, float* __restrict__ dw,
                                                                int cin, int accumulate) {
    __shared__ float red[8][32];
    const int col = threadIdx.x & 31, part = threadIdx.x >> 5;     // 32 slab columns x 8 slab groups per block
    const int orow = blockIdx.x;                                   // (o, r) pair: 64 * 7 blocks
    float s = 0.f;
    for (int k = part; k < nslab; k += 8) s += slabs[((size_t)k * KO * 7 + orow) * 32 + col];
    red[part][col] = s;
    __syncthreads();
    if (part != 0) return;
#pragma unroll
    for (int k = 1; k < 8; ++k) s += red[k][col];
    const int sp = col >> 2, c = col & 3;
    if (sp == 0 || c >= cin) return;
    float* d = dw + ((size_t)orow * 7 + (sp - 1)) * cin + c;
    *d = accumulate ? *d + s : s;
}

int stem_geometry(StemParams& p, int N, int H, int W) {
    if (N <= 0 || H % (2 * TH) || W % (2 * TW)) return DS6G_ERR_ARG;
    p.N = N; p.H = H; p.W = W; p.Ho = H / 2; p.Wo = W / 2;
    p.tiles_x = p.Wo / TW; p.tiles_y = p.Ho / TH;
    p.ntiles = N * p.tiles_x * p.tiles_y;
    const size_t xb = (size_t)N * H * W * 4 * 2, yb = (size_t)N * p.Ho * p.Wo * KO * 2;
    if (xb >= OOB_OFF || yb >= OOB_OFF) return DS6G_ERR_ARG;
    p.x_bytes = (unsigned)xb; p.dy_bytes = (unsigned)yb;
    return DS6G_OK;
}

constexpr int STEM_GRID_FWD = 512, STEM_GRID_WGRAD = 256;

}  // namespace

extern "C" {

size_t ds6g_bf16_stem_workspace_bytes(void) {
    const size_t fwd = (size_t)KO * WROW * 2 + 1024 + (size_t)STEM_GRID_FWD * 2 * KO * sizeof(double);
    const size_t wg = (size_t)STEM_GRID_WGRAD * KO * 7 * 32 * sizeof(float);
    return fwd > wg ? fwd : wg;
}

// y = conv7x7/2(x, w) on bf16 storage + the train-mode BatchNorm statistics of y.  x [N][H][W][4] bf16 (channels >= cin
// zero), w: the fp32 master filter [64][7][7][cin] (OHWI), y [N][H/2][W/2][64] bf16; H % 16 == 0, W % 32 == 0.
// mean == NULL: convolution only (eval mode: the caller applies running statistics).
int ds6g_bf16_stem_fwd(const void* x, const float* w, int cin, void* y, int N, int H, int W, float eps, float momentum,
                       float* mean, float* invstd, float* running_mean, float* running_var, void* ws, size_t ws_bytes,
                       void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(x && w && y && ws && cin >= 1 && cin <= 4 && ws_bytes >= ds6g_bf16_stem_workspace_bytes());
    StemParams p{};
    DS6G_CHECK_ARG(stem_geometry(p, N, H, W) == DS6G_OK);
    __bf16* w16 = (__bf16*)ws;
    p.x = (const __bf16*)x; p.w16 = w16; p.y = (__bf16*)y;
    p.bn_partial = (double*)((char*)ws + (((size_t)KO * WROW * 2 + 1023) & ~(size_t)1023));
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(stem_pack_weights_kernel, dim3(cdiv(KO * WROW, 256)), dim3(256), 0, st, w, w16, cin);
    DS6G_LAUNCH_CHECK();
    const int grid = p.ntiles < STEM_GRID_FWD ? p.ntiles : STEM_GRID_FWD;
    hipLaunchKernelGGL(stem_fwd_kernel, dim3(grid), dim3(256), 0, st, p);
    DS6G_LAUNCH_CHECK();
    if (!mean) return DS6G_OK;
    DS6G_CHECK_ARG(invstd);
    return ds6g_internal_bn_stats_finalize(p.bn_partial, grid, (long)N * p.Ho * p.Wo, KO, eps, momentum, mean, invstd,
                                           running_mean, running_var, st);
}

// dw[64][7][7][cin] (+)= weight gradient of the same convolution from x [N][H][W][4] bf16 and dy [N][H/2][W/2][64] bf16
int ds6g_bf16_stem_wgrad(const void* x, const void* dy, float* dw, int cin, int N, int H, int W, int accumulate, void* ws,
                         size_t ws_bytes, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(x && dy && dw && ws && cin >= 1 && cin <= 4 && ws_bytes >= ds6g_bf16_stem_workspace_bytes());
    StemParams p{};
    DS6G_CHECK_ARG(stem_geometry(p, N, H, W) == DS6G_OK);
    p.x = (const __bf16*)x; p.dy = (const __bf16*)dy; p.slabs = (float*)ws;
    hipStream_t st = (hipStream_t)stream;
    const int grid = p.ntiles < STEM_GRID_WGRAD ? p.ntiles : STEM_GRID_WGRAD;
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3(grid), dim3(256), 0, st, p);
    DS6G_LAUNCH_CHECK();
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(KO * 7), dim3(256), 0, st, (const float*)ws, grid, dw, cin, accumulate);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

}  // extern "C"
