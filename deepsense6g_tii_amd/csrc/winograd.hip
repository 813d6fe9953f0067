// Winograd F(2x2, 3x3) convolution for the 3x3 / stride 1 / pad 1 layers of the ResNet trunks
// (torchvision BasicBlock conv1 / conv2 at call sites /root/reference/model2_seq.py:510-512,528-530,546-548,565-567):
// 16 small GEMMs over 4x4-transformed input tiles instead of 9 taps - 2.25x fewer matrix-core FLOPs, which is the
// lever that is left once the direct implicit GEMM sits at the chip's power-limited fp32 MFMA rate.
//
//   V_p[tile][c] = (B^T d B)_p        input transform  (p = 0..15, d = the 4x4 input patch of a 2x2 output tile)
//   M_p[tile][k] = sum_c V_p[tile][c] * U_p[k][c]      16 GEMMs on v_mfma_f32_32x32x2_f32 (exact fp32 products)
//   y            = A^T M A            output transform
// with U_p = (G g G^T)_p precomputed per call by winograd_weights_kernel (weights change every optimizer step).
//
// One workgroup = 32 tiles (BTH x BTW tile rows/columns) x 32 output channels; wave w owns transform positions
// 4w..4w+3.  Per 16-channel chunk: every thread loads its (tile, channel-pair) 4x4 patch straight from global memory
// (buffer loads, out-of-image pixels read 0), transforms it in registers and writes the 16 V values into the LDS image
// the MFMA A-fragments are read from (same [row][16] + XOR-swizzle image as igemm.hip); the B operand (U) goes
// global -> registers directly in fragment layout (each wave needs only its own 4 positions: nothing to share).
// The next chunk's patch and U fragments are prefetched into registers while the MFMAs of the current chunk run.
#include "common.h"
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <cstdio>

// bench instrumentation shared with igemm.hip (ds6g_profile_begin / _end)
void* ds6g_prof_open(int variant, double flops, hipStream_t st);
void ds6g_prof_close(void* rec, hipStream_t st);

int g_wino_kb64 = 0;  // timing experiments: 64 output channels per workgroup, one wave per SIMD (measured slower)

// timing ablations of the forward kernel (env DS6G_WINO_DBG) exist only in a -DDS6G_WINO_ABLATE build: in the shipped build the
// tests fold away, so the chunk loop has no control flow between the MFMA chain and the input transform
#ifdef DS6G_WINO_ABLATE
#define WINO_DBG(bit) (p.dbg & (bit))
#else
#define WINO_DBG(bit) false
#endif

namespace {

constexpr int WG_TILES = 32;   // tiles per workgroup (MFMA rows)
constexpr int WG_KB = 32;      // output channels per MFMA column block (a workgroup owns TNK of them)
constexpr int WG_CH = 16;      // channels per chunk

struct WinoParams {
    const float* x;      // [N][H][W][C]
    const float* u;      // [16][K/32][C/16] fragments of 64 lanes x 8 floats (see winograd_weights_kernel)
    float* y;            // [N][H][W][K]
    int N, H, W, C, K;
    int TH, TW;          // tiles per image column / row (H/2, W/2)
    int BTH, BTW;        // tile rows / columns per workgroup (BTH * BTW = 32)
    int rows_total;      // N * TH  (tile rows over the whole batch)
    int col_blocks;      // TW / BTW
    unsigned x_bytes, u_bytes;
    int dbg;             // timing ablations (DS6G_WINO_DBG): 1 no patch loads, 2 no U loads, 4 no transform/store, 8 no MFMA
    const float* bias;      // inference epilogue (BN folded into the filter): per-channel bias,
    const float* residual;  //   identity branch [N][H][W][K],
    int relu;               //   0 none, 1 before the residual add, 2 after it
    int accumulate;      // y += result (data gradient summed onto the gradient of the residual branch)
    int items;           // winograd_pc_kernel: work items = tile blocks x K / 64
    int xcd_gk;          //   XCD-aware item order: the 8 XCDs form gk channel-block groups x 8 / gk tile-block groups (0: off)
    int btw_shift;       //   log2(BTW)
    unsigned th_magic;   //   ceil(2^32 / TH), TH > 1: row / TH = umulhi(row, th_magic) for row * TH < 2^32
    unsigned long long* tdbg;  // -DDS6G_WINO_ABLATE, dbg 64: per-step barrier arrive / leave clocks of workgroup 0
};

// U[p][k][c] = (G g G^T)[p], G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]].  transpose_flip: build the dgrad filter
// instead - input/output channels swapped and taps rotated by 180 degrees: g'[c][r][s][k] = g[k][2-r][2-s][c].
__global__ __launch_bounds__(256) void winograd_weights_kernel(const float* __restrict__ w, float* __restrict__ u, int K,
                                                               int C, int transpose_flip_in) {
    // blockIdx.y = 1 (only launched when both are wanted): the dgrad filter into the second half of u
    const int transpose_flip = gridDim.y > 1 ? (int)blockIdx.y : transpose_flip_in;
    if (gridDim.y > 1 && blockIdx.y == 1) u += (size_t)16 * K * C;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int Ko = transpose_flip ? C : K, Co = transpose_flip ? K : C;  // output-side (rows of U) / reduction side
    if (i >= (long)Ko * Co) return;
    const int ko = (int)(i / Co), co = (int)(i - (long)ko * Co);
    float g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s)
            g[r][s] = transpose_flip ? w[(((long)co * 3 + (2 - r)) * 3 + (2 - s)) * C + ko]
                                     : w[(((long)ko * 3 + r) * 3 + s) * C + co];
    float t[4][3];  // G g
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        t[0][s] = g[0][s];
        t[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]);
        t[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]);
        t[3][s] = g[2][s];
    }
    // stored in MFMA B-fragment order: for position p, 32-row block kb, 16-column chunk cc the 512 values sit as
    // [lane = row + 32 * (col / 8)][col % 8], so a wave reads its fragment of a chunk as two fully coalesced 1 KiB loads
    const size_t frag = ((size_t)(ko >> 5) * (Co >> 4) + (co >> 4)) * 512 + (size_t)(((ko & 31) + 32 * ((co >> 3) & 1)) * 8 + (co & 7));
    const size_t pstride = (size_t)Ko * Co;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float o0 = t[r][0];
        const float o1 = 0.5f * (t[r][0] + t[r][1] + t[r][2]);
        const float o2 = 0.5f * (t[r][0] - t[r][1] + t[r][2]);
        const float o3 = t[r][2];
        const size_t base = (size_t)(r * 4) * pstride + frag;
        u[base] = o0;
        u[base + pstride] = o1;
        u[base + 2 * pstride] = o2;
        u[base + 3 * pstride] = o3;
    }
}


// TNK = 32-channel output blocks per workgroup.  TNK = 2 (one wave per SIMD, 64 MFMAs per wave and chunk) gives the
// register prefetch of the next chunk twice the time to land and halves the redundant input transforms.
template <int TNK>
__global__ __launch_bounds__(256, (TNK == 1 ? 2 : 1)) void winograd_fwd_kernel(const WinoParams p) {
    // two V images of [16 positions][32 tiles][16 ch] (2 KiB per position) = 64 KiB; reused as M [16][32 tiles][32 k]
    __shared__ __attribute__((aligned(16))) float lds[16 * WG_TILES * WG_KB];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, khalf = lane >> 5;
    // workgroup -> (tile-row block, tile-column block, output-channel block); k block fastest so the workgroups that
    // share an input patch are neighbours
    const int kblocks = p.K / (WG_KB * TNK);
    int wg = blockIdx.x;
    const int kb = wg % kblocks;
    wg /= kblocks;
    const int cb = wg % p.col_blocks;
    const int rb = wg / p.col_blocks;
    const int k0 = kb * WG_KB * TNK;

    const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const auto u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, p.u_bytes, 0x00020000);

    // ---- input-transform work item of this thread: tile (tid / 8), channel pair (tid % 8) ----
    const int tt = tid >> 3, cp = tid & 7;
    const int trow = rb * p.BTH + tt / p.BTW;          // global tile row (n * TH + th)
    const int tcol = cb * p.BTW + tt % p.BTW;
    const bool tile_ok = trow < p.rows_total;
    const int n = tile_ok ? trow / p.TH : 0;
    const int th = tile_ok ? trow - n * p.TH : 0;
    const int ih0 = 2 * th - 1, iw0 = 2 * tcol - 1;
    unsigned xoff[16];  // byte offsets of the 4x4 patch pixels at channel pair cp (OOB_OFF outside the image)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ih = ih0 + i, iw = iw0 + j;
            const bool ok = tile_ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            xoff[i * 4 + j] = ok ? (unsigned)((((n * p.H + ih) * p.W + iw) * p.C + cp * 2) * 4) : OOB_OFF;
        }
    // LDS destination of this thread's V values: position p -> p * 2048 B + tile * 64 B + swizzled channel-pair slot
    const unsigned vdst = (unsigned)(tt * 64 + ((((cp >> 1) ^ ((tt >> 2) & 3)) << 4) | ((cp & 1) << 3)));
    // ---- MFMA fragments ----
    const unsigned a_src = (unsigned)(l31 * 64);                         // + position * 2048 + swizzled chunk
    const int swz = (l31 >> 2) & 3;
    // U fragment of this lane: 8 floats at [pos][k0 / 32 (+j)][chunk][lane] (fragment-ordered by winograd_weights_kernel)
    const unsigned ubase = (unsigned)((((k0 >> 5) * (p.C >> 4)) * 512 + lane * 8) * 4);
    const unsigned upos = (unsigned)((size_t)p.K * p.C * 4);             // bytes between positions

    f32x16 acc[4][TNK];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < TNK; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][j][r] = 0.f;

    f32x2 raw[16];
    f32x4 ub[4][TNK][2];
    auto prefetch = [&](int c0) {
        if (!WINO_DBG(1))
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b64(x_rsrc, xoff[e], (unsigned)(c0 * 4), 0);
            raw[e][0] = __uint_as_float(v[0]);
            raw[e][1] = __uint_as_float(v[1]);
        }
        if (!WINO_DBG(2))
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < TNK; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const auto v = __builtin_amdgcn_raw_buffer_load_b128(
                        u_rsrc, ubase + (unsigned)(wave * 4 + q) * upos + (unsigned)(j * (p.C >> 4) * 2048 + h * 16),
                        (unsigned)((c0 >> 4) * 2048), 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ub[q][j][h][e] = __uint_as_float(v[e]);
                }
    };

    const int nchunks = p.C / WG_CH;
    // input transform of the prefetched patch, V = B^T d B with B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1], written to
    // V buffer `vb` (two buffers: the transform of chunk c+1 is written while other waves still multiply chunk c)
    auto transform_store = [&](int vb) {
        if (WINO_DBG(4)) return;
        f32x2 t[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[0][j] = vsub(raw[0 * 4 + j], raw[2 * 4 + j]);
            t[1][j] = raw[1 * 4 + j] + raw[2 * 4 + j];
            t[2][j] = vsub(raw[2 * 4 + j], raw[1 * 4 + j]);
            t[3][j] = vsub(raw[1 * 4 + j], raw[3 * 4 + j]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2 v0 = vsub(t[i][0], t[i][2]);
            const f32x2 v1 = t[i][1] + t[i][2];
            const f32x2 v2 = vsub(t[i][2], t[i][1]);
            const f32x2 v3 = vsub(t[i][1], t[i][3]);
            float* base = lds + vb * (16 * 512) + ((i * 4) * 2048 + vdst) / 4;
            *reinterpret_cast<f32x2*>(base) = v0;
            *reinterpret_cast<f32x2*>(base + 512) = v1;
            *reinterpret_cast<f32x2*>(base + 1024) = v2;
            *reinterpret_cast<f32x2*>(base + 1536) = v3;
        }
    };
    prefetch(0);
    transform_store(0);
    f32x4 bcur[4][TNK][2];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < TNK; ++j) { bcur[q][j][0] = ub[q][j][0]; bcur[q][j][1] = ub[q][j][1]; }
    __syncthreads();
    for (int ck = 0; ck < nchunks; ++ck) {
        const bool more = ck + 1 < nchunks;
        if (more) prefetch((ck + 1) * WG_CH);  // patch + U fragments of the next chunk fly under the MFMAs below
        // ---- 4 positions x 8 MFMAs: acc[q] += V_pos[tiles][16] * U_pos[k][16]^T ----
        const float* vbuf = lds + (ck & 1) * (16 * 512);
        if (!WINO_DBG(8))
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float* vrow = vbuf + ((wave * 4 + q) * 2048 + a_src) / 4;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(vrow + (((khalf * 2 + 0) ^ swz) << 2));
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(vrow + (((khalf * 2 + 1) ^ swz) << 2));
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < TNK; ++j)
                    acc[q][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], bcur[q][j][0][e], acc[q][j], 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < TNK; ++j)
                    acc[q][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], bcur[q][j][1][e], acc[q][j], 0, 0, 0);
        }
        if (more) {
            transform_store((ck + 1) & 1);  // the other buffer: last read one chunk ago, behind the barrier below
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < TNK; ++j) { bcur[q][j][0] = ub[q][j][0]; bcur[q][j][1] = ub[q][j][1]; }
        }
        __syncthreads();
    }
    // ---- output transform: M[pos][tile][k] through LDS (one 32-channel block at a time), y = A^T M A,
    //      A^T = [1 1 1 0; 0 1 -1 -1] ----
    const int kk = tid & 31;
#pragma unroll
    for (int j = 0; j < TNK; ++j) {
        if (j > 0) __syncthreads();  // the previous block's M has been consumed
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tile = (r & 3) + 8 * (r >> 2) + 4 * khalf;
                lds[((wave * 4 + q) * WG_TILES + tile) * WG_KB + l31] = acc[q][j][r];
            }
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int tile = pass * 8 + (tid >> 5);
            const int orow = rb * p.BTH + tile / p.BTW, ocol = cb * p.BTW + tile % p.BTW;
            float m[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) m[e] = lds[(e * WG_TILES + tile) * WG_KB + kk];
            float sr[2][4];  // A^T M
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                sr[0][c] = m[0 * 4 + c] + m[1 * 4 + c] + m[2 * 4 + c];
                sr[1][c] = m[1 * 4 + c] - m[2 * 4 + c] - m[3 * 4 + c];
            }
            if (orow < p.rows_total) {
                const int on = orow / p.TH, oth = orow - on * p.TH;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float y0 = sr[i][0] + sr[i][1] + sr[i][2];
                    const float y1 = sr[i][1] - sr[i][2] - sr[i][3];
                    const size_t oi = ((size_t)(on * p.H + 2 * oth + i) * p.W + 2 * ocol) * p.K + k0 + j * WG_KB + kk;
                    float* o = p.y + oi;
                    if (p.accumulate) {
                        o[0] += y0;
                        o[p.K] += y1;
                    } else if (p.bias || p.residual || p.relu) {
                        float v0 = y0, v1 = y1;
                        if (p.bias) { const float bb = p.bias[k0 + j * WG_KB + kk]; v0 += bb; v1 += bb; }
                        if (p.relu == 1) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                        if (p.residual) { v0 += p.residual[oi]; v1 += p.residual[oi + p.K]; }
                        if (p.relu == 2) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                        o[0] = v0;
                        o[p.K] = v1;
                    } else {
                        o[0] = y0;
                        o[p.K] = y1;
                    }
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Producer / consumer form of the forward kernel (the default when K % 64 == 0): the same arithmetic, but the two
// halves of the chunk loop no longer share a wave.  One PERSISTENT workgroup of 8 waves per CU walks a list of work
// items (32 tiles x 64 output channels each); waves 0-3 (one per SIMD) only multiply - wave w owns the position row
// 4w..4w+3 x two 32-channel blocks, 64 MFMAs per 16-channel chunk step, its U fragments refilled in place one step ahead
// right after the MFMAs that read them - and waves 4-7 only move data: patch loads two steps ahead, B^T d B in two
// half-steps, the V image of the NEXT step (double-buffered), and the output transform + stores of the PREVIOUS item
// while the multipliers are already in the next one.  One barrier per chunk step; the step sequence runs across item
// boundaries, so prologue and output transform are paid once per workgroup, not once per item.  The multipliers fold the
// column half of A^T M A in registers (their four positions are one row of the 4x4 grid), which halves the LDS hand-over
// image.  What bounds it (per-step clocks of workgroup 0, DS6G_WINO_DBG=64 in a -DDS6G_WINO_ABLATE build): a steady step
// is 5000-5200 cycles for 4096 cycles of MFMA - beside the MFMA stream every other vector instruction of the SIMD
// (24 of the multiplier, 48 of the mover per step) costs time: about 13 cycles in the multiplier's own stream, a whole
// MFMA slot in its partner's (tools/mfma_valu_coexec.hip) - plus 2-4 thousand cycles at each item boundary (S hand-over,
// output transform, patch offsets of the next item).  The same split was built for the weight-gradient kernel and
// measured slower (160 / 131 / 132 us against 125 / 103 / 104): its movers need 120-220 vector instructions per 64 MFMAs.
#define PC_DBG(bit) ((DBG & (bit)) != 0)  // compile-time ablations (a -DDS6G_WINO_ABLATE build instantiates them)
constexpr int PC_KB = 64;                                     // output channels per item
constexpr int PC_V_FLOATS = 2 * 16 * WG_TILES * WG_CH;        // two V images            (64 KiB)
constexpr int PC_S_FLOATS = 4 * 2 * WG_TILES * PC_KB;         // S = M A per position row (64 KiB)
constexpr int PC_LDS_BYTES = (PC_V_FLOATS + PC_S_FLOATS) * 4;

template <int DBG>
__global__ __launch_bounds__(512, 1) void winograd_pc_kernel(const WinoParams p) {
    extern __shared__ __attribute__((aligned(16))) float pc_lds[];
    float* const Vl = pc_lds;                // [2][16 positions][32 tiles][16 ch], swizzled as in winograd_fwd_kernel
    float* const Sl = pc_lds + PC_V_FLOATS;  // [4 rows][2][32 tiles][64 k]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, khalf = lane >> 5;
    const bool consumer = wave < 4;
    const int kblocks = p.K / PC_KB;
    const int G = (int)gridDim.x;
    const int nchunks = p.C / WG_CH;
    const int my_items = (p.items - (int)blockIdx.x + G - 1) / G;
    const int T = my_items * nchunks;  // chunk steps of this workgroup
    // Item order.  A workgroup walks item ids it0, it0 + its, it0 + 2 its, ...; id q decodes to channel block
    // kb0 + q % kdiv and tile block tb0 + q / kdiv.  Plain order: ids = blockIdx.x + i G over all items (kb fastest).
    // XCD-aware order (p.xcd_gk > 0; needs G % 8 == 0 and items % G == 0): the hardware deals workgroups round-robin over
    // the 8 XCDs (workgroup w runs on XCD w & 7, each with a private 4 MB L2), so the XCDs are arranged as gk channel-block
    // groups x gt = 8 / gk tile-block groups; an XCD owns kblocks / gk channel blocks x a CONTIGUOUS run of tile blocks and its
    // G / 8 workgroups walk that sub-list together, 30 consecutive ids at a time: the items that share an input tile block
    // (all its channel blocks) and the neighbouring tile blocks (overlapping 4x4 / stride-2 patches) meet in one L2 instead
    // of being fetched by up to 8 of them, and an XCD touches only its own slice of U.  The host picks gk per layer shape
    // (minimum of |U| gt + |x| gk).
    int it0 = (int)blockIdx.x, its = G, kdiv = kblocks, kb0 = 0, tb0 = 0;
    if (p.xcd_gk > 0) {
        const int xcd = (int)blockIdx.x & 7, gk = p.xcd_gk, gt = 8 / gk;
        kdiv = kblocks / gk;
        kb0 = (xcd % gk) * kdiv;
        tb0 = (xcd / gk) * ((p.items / kblocks) / gt);
        it0 = (int)blockIdx.x >> 3;
        its = G >> 3;
    }
    it0 = __builtin_amdgcn_readfirstlane(it0); its = __builtin_amdgcn_readfirstlane(its);
    kdiv = __builtin_amdgcn_readfirstlane(kdiv); kb0 = __builtin_amdgcn_readfirstlane(kb0); tb0 = __builtin_amdgcn_readfirstlane(tb0);
    auto item_kb = [&](int item) __attribute__((always_inline)) { return kb0 + item % kdiv; };
    auto item_tb = [&](int item) __attribute__((always_inline)) { return tb0 + item / kdiv; };

    const auto u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, p.u_bytes, 0x00020000);
    const unsigned upos = (unsigned)((size_t)p.K * p.C * 4);  // bytes between positions of U

    // ---------------- multiplier state ----------------
    f32x16 acc[4][2];
    f32x4 ub[4][2][2];
    const unsigned a_src = (unsigned)(l31 * 64);
    const int swz = (l31 >> 2) & 3;
    // byte offset (scalar) of the U fragment block of (item, chunk) for position row `wave`, q = 0, j = 0
    auto u_soff = [&](int item, int ck) -> unsigned {
        const int kb = item_kb(item);
        return (unsigned)(wave * 4) * upos + (unsigned)(((kb * (PC_KB / 32)) * (p.C >> 4) + ck) * 2048);
    };
    auto load_u = [&](int q, int j, unsigned soff) {
        if (PC_DBG(2)) return;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, (unsigned)(lane * 32 + h * 16),
                                                                 soff + (unsigned)q * upos + (unsigned)(j * (p.C >> 4) * 2048), 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) ub[q][j][h][e] = __uint_as_float(v[e]);
        }
    };

    // ---------------- mover state (waves 4-7) ----------------
    // Beside a wave that issues fp32 32x32x2 MFMAs back to back its SIMD partner gets about ONE vector instruction (VALU,
    // LDS or VMEM alike) per MFMA (tools/mfma_valu_coexec.hip): the movers' budget is 64 instruction slots per chunk step, so
    // everything here is 16 bytes wide.  Transform item = (tile, 4 channels): 128 items per chunk = two waves; waves 4, 5
    // produce the even chunk steps and waves 6, 7 the odd ones, each over TWO steps (stage A: column pass of B^T d B and
    // the loads two chunks on; stage B: row pass and the V image), so that every SIMD's mover has half an item per step.
    const int mw = wave & 3;
    const int pair = mw >> 1;                        // parity of the chunk steps this wave produces
    const int tt = (mw & 1) * 16 + (lane >> 2);      // tile of the transform item
    const int cq = lane & 3;                         // channel quad within the 16-channel chunk
    const unsigned vdst = (unsigned)(tt * 64 + ((cq ^ ((tt >> 2) & 3)) << 4));  // bytes within a position's [32][16] image
    // Patch addressing: the buffer descriptor starts one image row and one pixel BEFORE x, so the byte offset of patch pixel
    // (0, 0) of any tile is non-negative; pixel (i, j) adds the wave-uniform (i W + j) C 4 in the instruction's scalar offset
    // (not range-checked), and the per-lane part is one of 9 registers: base + [row i outside the image] + [column j outside],
    // each "outside" being 2^30 - beyond the descriptor, so the hardware returns 0 for the padding.  Only rows 0 / 3 and
    // columns 0 / 3 of a patch can fall outside.  (16 separately selected offsets cost twice the instructions per item.)
    const auto xs_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x) - (size_t)(p.W + 1) * p.C, 0,
                                                            p.x_bytes + (unsigned)((p.W + 1) * p.C * 4), 0x00020000);
    unsigned xo[3][3];  // [row class: 0, 1-2, 3][column class: 0, 1-2, 3]
    f32x4 raw[16], tc[16];
    auto setup_item = [&](int item) __attribute__((always_inline)) {
        constexpr unsigned BAD = 0x40000000u;
        const int wg = item_tb(item);
        const int cb = wg % p.col_blocks, rb = wg / p.col_blocks;
        const int trow = rb * p.BTH + (tt >> p.btw_shift), tcol = cb * p.BTW + (tt & (p.BTW - 1));
        const bool tile_ok = trow < p.rows_total;
        const int n = p.TH == 1 ? trow : (int)__umulhi((unsigned)trow, p.th_magic);  // trow / TH
        const int th = trow - n * p.TH;
        // pixel (2 th - 1, 2 tcol - 1) relative to the shifted descriptor = (2 th, 2 tcol) relative to x
        const unsigned base = (unsigned)((((n * p.H + 2 * th) * p.W + 2 * tcol) * p.C + cq * 4) * 4);
        const unsigned r[3] = {base + ((tile_ok && th > 0) ? 0u : BAD), base + (tile_ok ? 0u : BAD),
                               base + ((tile_ok && th < p.TH - 1) ? 0u : BAD)};
        const unsigned c0 = tcol > 0 ? 0u : BAD, c3 = tcol < p.TW - 1 ? 0u : BAD;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            xo[i][0] = r[i] + c0;
            xo[i][1] = r[i];
            xo[i][2] = r[i] + c3;
        }
    };
    auto load_patch = [&](int ck) __attribute__((always_inline)) {
        if (PC_DBG(1)) return;
        // (opaque strides: the 16 scalar offsets are formed here with a few scalar adds instead of living in 16 SGPRs through
        // the whole kernel, which spilled scalar registers into vector lanes)
        unsigned rowb = (unsigned)(p.W * p.C * 4), colb = (unsigned)(p.C * 4);
        asm volatile("" : "+s"(rowb), "+s"(colb));
        const unsigned s0 = (unsigned)(ck * WG_CH * 4);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = e >> 2, j = e & 3;
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(xs_rsrc, xo[i == 0 ? 0 : (i == 3 ? 2 : 1)][j == 0 ? 0 : (j == 3 ? 2 : 1)],
                                                                 s0 + (unsigned)i * rowb + (unsigned)j * colb, 0);
#pragma unroll
            for (int c = 0; c < 4; ++c) raw[e][c] = __uint_as_float(v[c]);
        }
    };
    // B^T d B with B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]: stage A = B^T d (rows), stage B = (.) B and the V image
    auto stage_a = [&]() __attribute__((always_inline)) {
        if (PC_DBG(4)) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tc[0 * 4 + j] = vsub(raw[0 * 4 + j], raw[2 * 4 + j]);
            tc[1 * 4 + j] = raw[1 * 4 + j] + raw[2 * 4 + j];
            tc[2 * 4 + j] = vsub(raw[2 * 4 + j], raw[1 * 4 + j]);
            tc[3 * 4 + j] = vsub(raw[1 * 4 + j], raw[3 * 4 + j]);
        }
    };
    auto stage_b = [&](int vb) __attribute__((always_inline)) {
        if (PC_DBG(4)) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float* base = Vl + vb * (16 * 512) + ((i * 4) * 2048 + vdst) / 4;
            *reinterpret_cast<f32x4*>(base) = vsub(tc[i * 4 + 0], tc[i * 4 + 2]);
            *reinterpret_cast<f32x4*>(base + 512) = tc[i * 4 + 1] + tc[i * 4 + 2];
            *reinterpret_cast<f32x4*>(base + 1024) = vsub(tc[i * 4 + 2], tc[i * 4 + 1]);
            *reinterpret_cast<f32x4*>(base + 1536) = vsub(tc[i * 4 + 1], tc[i * 4 + 3]);
        }
    };
    // y = A^T S of a finished item, half e (16 tiles) of it: thread = (tile, 4 output channels), 16-byte S reads and stores.
    // Two forms that share no control flow: the PLAIN one (forward in training) issues no load at all, so nothing in it waits on
    // the memory counter behind the patch prefetch; the FANCY one (accumulating data gradient; bias / ReLU / residual of the
    // inference path) reads what it adds (epi_issue) and then finishes (epi_finish).
    const bool fancy = p.accumulate || p.bias || p.residual || p.relu;
    f32x4 old[4];  // fancy: the values read by epi_issue for the half that epi_finish stores next
    auto epi_where = [&](int item, int eh, bool& ok, size_t& oi) __attribute__((always_inline)) {
        const int kb = item_kb(item);
        const int wg = item_tb(item);
        const int cb = wg % p.col_blocks, rb = wg / p.col_blocks;
        const int tile = eh * 16 + mw * 4 + (lane >> 4);
        const int orow = rb * p.BTH + (tile >> p.btw_shift), ocol = cb * p.BTW + (tile & (p.BTW - 1));
        ok = orow < p.rows_total;
        const int on = p.TH == 1 ? orow : (int)__umulhi((unsigned)orow, p.th_magic), oth = orow - on * p.TH;
        oi = ((size_t)(on * p.H + 2 * oth) * p.W + 2 * ocol) * p.K + (size_t)(kb * PC_KB + (lane & 15) * 4);
    };
    auto epi_y = [&](int eh, f32x4 (&y)[4]) __attribute__((always_inline)) {
        const int tile = eh * 16 + mw * 4 + (lane >> 4);
        f32x4 s[4][2];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                s[r][c] = *reinterpret_cast<const f32x4*>(Sl + ((r * 2 + c) * WG_TILES + tile) * PC_KB + (lane & 15) * 4);
        y[0] = s[0][0] + s[1][0] + s[2][0];
        y[1] = s[0][1] + s[1][1] + s[2][1];
        y[2] = vsub(vsub(s[1][0], s[2][0]), s[3][0]);
        y[3] = vsub(vsub(s[1][1], s[2][1]), s[3][1]);
    };
    auto epi_store = [&](bool ok, size_t oi, const f32x4 (&y)[4]) __attribute__((always_inline)) {
        if (!ok) return;
        const size_t rowstride = (size_t)p.W * p.K;
        float* o = p.y + oi;
        *reinterpret_cast<f32x4*>(o) = y[0];
        *reinterpret_cast<f32x4*>(o + p.K) = y[1];
        *reinterpret_cast<f32x4*>(o + rowstride) = y[2];
        *reinterpret_cast<f32x4*>(o + rowstride + p.K) = y[3];
    };
    auto epilogue_plain = [&](int item, int eh) __attribute__((always_inline)) {
        if (PC_DBG(16)) return;
        bool ok;
        size_t oi;
        epi_where(item, eh, ok, oi);
        f32x4 y[4];
        epi_y(eh, y);
        epi_store(ok, oi, y);
    };
    auto epi_issue = [&](int item, int eh) __attribute__((always_inline)) {
        if (PC_DBG(16)) return;
        bool ok;
        size_t oi;
        epi_where(item, eh, ok, oi);
        const float* side = p.accumulate ? p.y : p.residual;  // what is added to the result (either or none)
#pragma unroll
        for (int i = 0; i < 4; ++i) old[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (side && ok) {
            const size_t rowstride = (size_t)p.W * p.K;
            const float* sp = side + oi;
            old[0] = *reinterpret_cast<const f32x4*>(sp);
            old[1] = *reinterpret_cast<const f32x4*>(sp + p.K);
            old[2] = *reinterpret_cast<const f32x4*>(sp + rowstride);
            old[3] = *reinterpret_cast<const f32x4*>(sp + rowstride + p.K);
        }
    };
    auto epi_finish = [&](int item, int eh) __attribute__((always_inline)) {
        if (PC_DBG(16)) return;
        bool ok;
        size_t oi;
        epi_where(item, eh, ok, oi);
        f32x4 y[4];
        epi_y(eh, y);
        if (p.accumulate) {
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] += old[i];
        } else {
            f32x4 bb = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias) bb = *reinterpret_cast<const f32x4*>(p.bias + item_kb(item) * PC_KB + (lane & 15) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float v = y[i][c] + bb[c];
                    if (p.relu == 1) v = fmaxf(v, 0.f);
                    v += old[i][c];
                    if (p.relu == 2) v = fmaxf(v, 0.f);
                    y[i][c] = v;
                }
        }
        epi_store(ok, oi, y);
    };

    // The two roles run separate loops with the same number of barriers (T + 1), so neither holds the other's registers.
    if (consumer) {
        int it_cur = it0, ck_cur = 0;  // chunk step t
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][j][r] = 0.f;
                load_u(q, j, u_soff(it_cur, 0));
            }
        __syncthreads();
        for (int t = 0; t < T; ++t) {
            const bool last_of_item = ck_cur + 1 == nchunks;
            const int it_nxt = last_of_item ? it_cur + its : it_cur;
            const int ck_nxt = last_of_item ? 0 : ck_cur + 1;
            // U of the next chunk step (the last step re-reads its own: no branch in the chain)
            const bool more = t + 1 < T;
            const unsigned soff = u_soff(more ? it_nxt : it_cur, more ? ck_nxt : ck_cur);
            const float* vbuf = Vl + (t & 1) * (16 * 512);
            auto read_a = [&](int q, f32x4& a0, f32x4& a1) {
                const float* vrow = vbuf + ((wave * 4 + q) * 2048 + a_src) / 4;
                a0 = *reinterpret_cast<const f32x4*>(vrow + (((khalf * 2 + 0) ^ swz) << 2));
                a1 = *reinterpret_cast<const f32x4*>(vrow + (((khalf * 2 + 1) ^ swz) << 2));
            };
            f32x4 a0, a1, n0, n1;
            read_a(0, a0, a1);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < 3) read_a(q + 1, n0, n1);  // the next position's fragments land under this one's MFMAs
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (!PC_DBG(8)) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[q][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], ub[q][j][0][e], acc[q][j], 0, 0, 0);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[q][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], ub[q][j][1][e], acc[q][j], 0, 0, 0);
                    }
                    load_u(q, j, soff);  // refill in place for the next chunk step: a whole step of latency cover
                    __builtin_amdgcn_sched_barrier(0);  // ... which only holds if the refill stays here
                }
                a0 = n0;
                a1 = n1;
            }
            if (last_of_item) {
                // S = M A for this wave's position row: columns (1 1 1 0) and (0 1 -1 -1)
                // (the lane's base address is made opaque here so that the 64 store addresses are base + immediate offsets
                // formed on the spot instead of loop-invariant registers held through the MFMA loop)
                unsigned so = (unsigned)(wave * (2 * WG_TILES * PC_KB) + khalf * (4 * PC_KB) + l31);
                asm volatile("" : "+v"(so));  // (an opaque OFFSET: an opaque pointer would lose its LDS address space)
                float* sb = Sl + so;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {  // register pairs: packed fp32 adds
                        const int trow = (r & 3) + 8 * (r >> 2);  // + 4 * khalf = tile (r + 1: the next tile row)
                        const f32x2 m0 = {acc[0][j][r], acc[0][j][r + 1]}, m1 = {acc[1][j][r], acc[1][j][r + 1]};
                        const f32x2 m2 = {acc[2][j][r], acc[2][j][r + 1]}, m3 = {acc[3][j][r], acc[3][j][r + 1]};
                        const f32x2 s0 = m0 + m1 + m2, s1 = vsub(vsub(m1, m2), m3);
                        sb[trow * PC_KB + j * 32] = s0[0];
                        sb[(trow + 1) * PC_KB + j * 32] = s0[1];
                        sb[WG_TILES * PC_KB + trow * PC_KB + j * 32] = s1[0];
                        sb[WG_TILES * PC_KB + (trow + 1) * PC_KB + j * 32] = s1[1];
                    }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[q][j][r] = 0.f;
            }
            if (PC_DBG(64) && blockIdx.x == 0 && wave == 0 && lane == 0 && t < 64) p.tdbg[t * 8 + 0] = __builtin_readcyclecounter();
            __syncthreads();
            if (PC_DBG(64) && blockIdx.x == 0 && wave == 0 && lane == 0 && t < 64) p.tdbg[t * 8 + 1] = __builtin_readcyclecounter();
            it_cur = it_nxt;
            ck_cur = ck_nxt;
        }
    } else {
        if (PC_DBG(32)) __builtin_amdgcn_s_setprio(2);
        int it_cur = it0, ck_cur = 0;  // chunk step s (what the multipliers are on)
        int it_ld = it_cur, ck_ld = pair;          // the chunk step whose patch loads were issued last (this wave: s = pair mod 2)
        auto next_load = [&]() __attribute__((always_inline)) {
            ck_ld += 2;
            if (ck_ld >= nchunks) { ck_ld -= nchunks; it_ld += its; setup_item(it_ld); }
            load_patch(ck_ld);
        };
        setup_item(it_ld);
        load_patch(ck_ld);
        stage_a();
        if (pair + 2 < T) next_load();
        if (pair == 0) stage_b(0);
        __syncthreads();
        for (int s = 0; s < T; ++s) {
            if ((s & 1) == pair) {  // stage A of chunk step s + 2 (its patch is in raw), then the loads of step s + 4
                if (s + 2 < T) {
                    stage_a();
                    if (s + 4 < T) next_load();
                }
            } else if (s + 1 < T) {
                stage_b((s + 1) & 1);  // chunk step s + 1: the image the multipliers read after the barrier below
            }
            // output transform of the item the multipliers finished one step ago: its two halves in the first two steps of
            // this item (S is next written at the end of this item's last step)
            if (it_cur != it0) {
                if (!fancy) {
                    if (nchunks >= 3) {
                        if (ck_cur < 2) epilogue_plain(it_cur - its, ck_cur);
                    } else if (ck_cur == 0) {
                        epilogue_plain(it_cur - its, 0);
                        epilogue_plain(it_cur - its, 1);
                    }
                } else if (nchunks >= 3) {
                    // (issuing the reads of half e a step before finishing it was measured slower: 119 / 92 / 81 / 83 us against
                    // 113 / 88 / 79 / 86 for the accumulating form)
                    if (ck_cur < 2) {
                        epi_issue(it_cur - its, ck_cur);
                        epi_finish(it_cur - its, ck_cur);
                    }
                } else if (ck_cur == 0) {
                    for (int eh = 0; eh < 2; ++eh) {
                        epi_issue(it_cur - its, eh);
                        epi_finish(it_cur - its, eh);
                    }
                }
            }
            if (PC_DBG(64) && blockIdx.x == 0 && wave == 4 && lane == 0 && s < 64) p.tdbg[s * 8 + 2] = __builtin_readcyclecounter();
            __syncthreads();
            if (PC_DBG(64) && blockIdx.x == 0 && wave == 4 && lane == 0 && s < 64) p.tdbg[s * 8 + 3] = __builtin_readcyclecounter();
            if (++ck_cur == nchunks) { ck_cur = 0; it_cur += its; }
        }
        for (int eh = 0; eh < 2; ++eh) {
            if (!fancy) {
                epilogue_plain(it_cur - its, eh);
            } else {
                epi_issue(it_cur - its, eh);
                epi_finish(it_cur - its, eh);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the same convolutions in the Winograd domain: the forward is linear in U = G g G^T, so
//     dU_p[k][c] = sum_tiles E_p[tile][k] * V_p[tile][c],   E = A dY A^T (2x2 -> 4x4),   V = B^T d B,
//     dW = G^T dU G   (4x4 -> 3x3, winograd_wgrad_finish_kernel, which also sums the split slabs)
// - 16 GEMMs whose reduction runs over the tiles: 4 MFMA FLOPs per pixel, channel pair and filter instead of 9.
// One workgroup = one row i of the 4x4 positions (blockIdx.y; wave j owns position (i, j)), 64 output x 64 input
// channels, one split of the tile range.  Row i of B^T / A touches only two input rows / one or two dy rows, so the
// four row-workgroups of a tile share no transform arithmetic.  Per chunk of 16 tiles every thread transforms one
// (tile, 4-channel) item of x and one of dy in registers and writes them to the tile-major LDS images the MFMA
// fragments are read from (ds_read_b32, consecutive lanes = consecutive channels), double-buffered, the next chunk's
// pixels prefetched into registers under the MFMAs.
struct WinoWgradParams {
    const float* x;    // [N][H][W][C]
    const float* dy;   // [N][H][W][K]
    float* du;         // [splits][16][K][C]
    int N, H, W, C, K, TH, TW;
    int tiles_total, tiles_per_split;  // tiles_per_split is a multiple of 16
    unsigned x_bytes, dy_bytes;
    int xcd;             // XCD-contiguous workgroup order (DS6G_WW_XCD=0 switches it off)
};

constexpr int WW_T = 16;   // tiles per chunk (MFMA reduction depth)
constexpr int WW_KB = 64, WW_CB = 64;

__global__ __launch_bounds__(256, 2) void winograd_wgrad_kernel(const WinoWgradParams p) {
    __shared__ __attribute__((aligned(16))) float Es[2][4][WW_T][WW_KB];  // [buffer][j][tile][k]
    __shared__ __attribute__((aligned(16))) float Vs[2][4][WW_T][WW_CB];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, khalf = lane >> 5;
    const int cblocks = p.C / WW_CB;
    // XCD-aware order (round 3): the hardware deals workgroups round-robin over the 8 XCDs in dispatch order (x fastest, then
    // y, then z); remapped so that every XCD walks a CONTIGUOUS run of (split, row, channel-block pair) ids, channel-block pair
    // fastest: the K/64 x C/64 workgroups of one (split, row i) - which read the same two patch rows of the same x tiles and the
    // same dy tiles, each a different channel slice pair - meet in one L2 instead of eight
    int bx = (int)blockIdx.x, by = (int)blockIdx.y, bz = (int)blockIdx.z;
    if (p.xcd) {
        const int gx = (int)gridDim.x, nwg = gx * 4 * (int)gridDim.z, orig = bx + gx * (by + 4 * bz);
        const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
        const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
        bx = lin % gx;
        by = (lin / gx) & 3;
        bz = lin / (4 * gx);
    }
    const int k0 = (bx / cblocks) * WW_KB, c0 = (bx % cblocks) * WW_CB;
    const int ri = by;          // row i of the position grid
    const int split = bz;
    const int tbeg = split * p.tiles_per_split;
    const int tend = min(p.tiles_total, tbeg + p.tiles_per_split);
    const int nchunks = (tend - tbeg + WW_T - 1) / WW_T;

    // x through a descriptor that starts one image row and one pixel BEFORE x (patch pixel (0, 0) of any tile then has a
    // non-negative offset; see winograd_pc_kernel): per lane ONE base + "row outside" + "column outside" (2^30 each), per
    // patch pixel a wave-uniform scalar offset
    const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x) - (size_t)(p.W + 1) * p.C, 0,
                                                          p.x_bytes + (unsigned)((p.W + 1) * p.C * 4), 0x00020000);
    const auto y_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, p.dy_bytes, 0x00020000);
    // rows of the 4x4 patch row i of B^T combines: t = d[ra] + sb * d[rb];  rows of the 2x2 dy tile row i of A combines
    const int ra = (ri == 0) ? 0 : (ri == 2 ? 2 : 1);
    const int rb = (ri == 3) ? 3 : (ri == 2 ? 1 : 2);
    const float sb = (ri == 1) ? 1.f : -1.f;
    // e = ya * dy[0] + yb * dy[1]
    const float ya = (ri == 3) ? 0.f : 1.f;
    const float yb = (ri == 0) ? 0.f : (ri == 1 ? 1.f : -1.f);

    // transform work item: tile (tid / 16) of the chunk, channel quad (tid % 16)
    const int tt = tid >> 4, qd = tid & 15;
    int tile = tbeg + tt;  // running tile index of this thread
    int tn, tth, ttw;
    {
        const int per_img = p.TH * p.TW;
        tn = tile / per_img;
        const int rem = tile - tn * per_img;
        tth = rem / p.TW;
        ttw = rem - tth * p.TW;
    }
    f32x4 xr[8], yr[4];
    auto prefetch = [&]() {
        constexpr unsigned BAD = 0x40000000u;
        const bool ok = tile < tend;
        const int pix = (tn * p.H + 2 * tth) * p.W + 2 * ttw;  // pixel (2 tth, 2 ttw) of x and of dy
        const unsigned xbase = (unsigned)((pix * p.C + c0 + qd * 4) * 4);
        // patch rows 0 / 3 fall outside the image in the first / last tile row, columns 0 / 3 in the first / last tile column
        const unsigned rbad0 = (ok && tth > 0) ? 0u : BAD, rbad12 = ok ? 0u : BAD, rbad3 = (ok && tth < p.TH - 1) ? 0u : BAD;
        const unsigned cbad0 = ttw > 0 ? 0u : BAD, cbad3 = ttw < p.TW - 1 ? 0u : BAD;
        unsigned rowb = (unsigned)(p.W * p.C * 4), colb = (unsigned)(p.C * 4);
        asm volatile("" : "+s"(rowb), "+s"(colb));  // (opaque: the scalar pixel offsets are formed here, not kept in 8 SGPRs)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = h ? rb : ra;  // wave-uniform
            const unsigned vrow = xbase + (r == 0 ? rbad0 : (r == 3 ? rbad3 : rbad12));
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const unsigned voff = b == 0 ? vrow + cbad0 : (b == 3 ? vrow + cbad3 : vrow);
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, voff, (unsigned)r * rowb + (unsigned)b * colb, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) xr[h * 4 + b][e] = __uint_as_float(v[e]);
            }
        }
        const unsigned ybase = ok ? (unsigned)((pix * p.K + k0 + qd * 4) * 4) : OOB_OFF;
        unsigned yrow = (unsigned)(p.W * p.K * 4), ycol = (unsigned)(p.K * 4);
        asm volatile("" : "+s"(yrow), "+s"(ycol));
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(y_rsrc, ybase, (unsigned)a * yrow + (unsigned)b * ycol, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) yr[a * 2 + b][e] = __uint_as_float(v[e]);
            }
        // advance to the same slot of the next chunk
        tile += WW_T;
        ttw += WW_T;
        while (ttw >= p.TW) {
            ttw -= p.TW;
            if (++tth == p.TH) { tth = 0; ++tn; }
        }
    };
    auto transform_store = [&](int buf) {
        f32x4 t[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) t[b] = xr[b] + sb * xr[4 + b];
        f32x4* vd = reinterpret_cast<f32x4*>(&Vs[buf][0][tt][qd * 4]);
        constexpr int PS = WW_T * WW_CB / 4;  // f32x4 stride between positions j
        vd[0] = vsub(t[0], t[2]);
        vd[PS] = t[1] + t[2];
        vd[2 * PS] = vsub(t[2], t[1]);
        vd[3 * PS] = vsub(t[1], t[3]);
        const f32x4 e0 = ya * yr[0] + yb * yr[2], e1 = ya * yr[1] + yb * yr[3];
        f32x4* ed = reinterpret_cast<f32x4*>(&Es[buf][0][tt][qd * 4]);
        constexpr int QS = WW_T * WW_KB / 4;
        ed[0] = e0;
        ed[QS] = e0 + e1;
        ed[2 * QS] = vsub(e0, e1);
        ed[3 * QS] = e1 * -1.f;
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nchunks > 0) {
        prefetch();
        transform_store(0);
    }
    __syncthreads();
    for (int ck = 0; ck < nchunks; ++ck) {
        const bool more = ck + 1 < nchunks;
        prefetch();  // unconditional (past the split's end every lane is out of range and reads zeros): a conditional load
                     // makes the compiler stage the 48 loaded registers through copies
        const float* Eb = &Es[ck & 1][wave][0][0];
        const float* Vb = &Vs[ck & 1][wave][0][0];
        float af[2][8], bf[2][8];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                af[i][kk] = Eb[(khalf * 8 + kk) * WW_KB + i * 32 + l31];
                bf[i][kk] = Vb[(khalf * 8 + kk) * WW_CB + i * 32 + l31];
            }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][kk], bf[j][kk], acc[i][j], 0, 0, 0);
        if (more) transform_store((ck + 1) & 1);
        __syncthreads();
    }
    // dU slab: [split][pos = 4 * ri + wave][k][c]
    float* out = p.du + ((size_t)split * 16 + (size_t)(ri * 4 + wave)) * p.K * p.C;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = k0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                out[(size_t)k * p.C + c0 + j * 32 + l31] = acc[i][j][r];
            }
}

// dW[k][r][s][c] (+)= (G^T (sum_splits dU) G)[r][s],  G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1]
__global__ __launch_bounds__(256) void winograd_wgrad_finish_kernel(const float* __restrict__ du, float* __restrict__ dw,
                                                                    int K, int C, int splits, int accumulate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)K * C) return;
    const int k = (int)(i / C), c = (int)(i - (long)k * C);
    float m[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) m[e] = 0.f;
    const size_t pstride = (size_t)K * C;
    for (int s = 0; s < splits; ++s) {
        const float* src = du + (size_t)s * 16 * pstride + i;
#pragma unroll
        for (int e = 0; e < 16; ++e) m[e] += src[(size_t)e * pstride];
    }
    float t[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[0][j] = m[0 * 4 + j] + 0.5f * (m[1 * 4 + j] + m[2 * 4 + j]);
        t[1][j] = 0.5f * (m[1 * 4 + j] - m[2 * 4 + j]);
        t[2][j] = 0.5f * (m[1 * 4 + j] + m[2 * 4 + j]) + m[3 * 4 + j];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float g0 = t[r][0] + 0.5f * (t[r][1] + t[r][2]);
        const float g1 = 0.5f * (t[r][1] - t[r][2]);
        const float g2 = 0.5f * (t[r][1] + t[r][2]) + t[r][3];
        float* o = dw + (((size_t)k * 3 + r) * 3) * C + c;
        if (accumulate) {
            o[0] += g0;
            o[C] += g1;
            o[2 * (size_t)C] += g2;
        } else {
            o[0] = g0;
            o[C] = g1;
            o[2 * (size_t)C] = g2;
        }
    }
}

}  // namespace

extern "C" {

size_t ds6g_winograd_weight_floats(int K, int C) { return (size_t)16 * K * C; }

// transpose_flip: 0 forward filter, 1 dgrad filter, 2 both (u then holds 2 * 16*K*C floats: forward, then dgrad)
int ds6g_winograd_weights(const float* w, float* u, int K, int C, int transpose_flip, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(w && u && K > 0 && C > 0 && transpose_flip >= 0 && transpose_flip <= 2);
    const long n = (long)K * C;
    hipLaunchKernelGGL(winograd_weights_kernel, dim3((unsigned)((n + 255) / 256), transpose_flip == 2 ? 2 : 1), dim3(256), 0,
                       (hipStream_t)stream, w, u, K, C, transpose_flip);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

// 1 if the shape is supported by ds6g_conv3x3_winograd_fwd
int ds6g_winograd_supported(int N, int H, int W, int C, int K) {
    if (H % 2 || W % 2 || C % WG_CH || K % WG_KB || N <= 0) return 0;
    const int TW = W / 2;
    if (TW >= 8 ? TW % 8 != 0 : 32 % TW != 0) return 0;
    const size_t xb = (size_t)N * H * W * C * 4, ub = (size_t)16 * K * C * 4;
    return xb < OOB_OFF && ub < OOB_OFF;
}

// y[N][H][W][K] = conv3x3 (stride 1, pad 1) of x[N][H][W][C] with the filter whose Winograd transform is u
// (ds6g_winograd_weights; transpose_flip = 1 there turns this call into the data gradient of the same conv);
// accumulate: y += result
static int wino_fwd_launch(const float* x, const float* u, float* y, int N, int H, int W, int C, int K, int accumulate,
                           const float* bias, const float* residual, int relu, void* stream);

int ds6g_conv3x3_winograd_fwd(const float* x, const float* u, float* y, int N, int H, int W, int C, int K, int accumulate,
                              void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(x && u && y && ds6g_winograd_supported(N, H, W, C, K));
    return wino_fwd_launch(x, u, y, N, H, W, C, K, accumulate, nullptr, nullptr, 0, stream);
}

// inference form (eval-mode BN folded into the filter by ds6g_bn_fold before ds6g_winograd_weights):
// y = act(conv(x) + bias [+ residual]);  relu: 0 none, 1 before the residual add, 2 after it
int ds6g_conv3x3_winograd_bias_act_fwd(const float* x, const float* u, const float* bias, const float* residual, float* y,
                                       int N, int H, int W, int C, int K, int relu, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(x && u && y && ds6g_winograd_supported(N, H, W, C, K) && relu >= 0 && relu <= 2);
    return wino_fwd_launch(x, u, y, N, H, W, C, K, 0, bias, residual, relu, stream);
}

static bool pc_set_lds() {
    bool ok = true;
#define PC_ATTR(D) ok = ok && hipFuncSetAttribute((const void*)winograd_pc_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS_BYTES) == hipSuccess;
    PC_ATTR(0)
#ifdef DS6G_WINO_ABLATE
    PC_ATTR(1) PC_ATTR(2) PC_ATTR(3) PC_ATTR(4) PC_ATTR(7) PC_ATTR(8) PC_ATTR(16) PC_ATTR(24) PC_ATTR(9) PC_ATTR(10) PC_ATTR(18) PC_ATTR(17) PC_ATTR(19) PC_ATTR(32) PC_ATTR(64) PC_ATTR(80) PC_ATTR(96)
#endif
#undef PC_ATTR
    return ok;
}

// CU count of the calling thread's current device, with the 128 KiB dynamic-LDS attribute of winograd_pc_kernel set on
// that device - once per device and process, safe under concurrent first calls (a process may drive several devices, one
// thread each, as torch.nn.DataParallel callers do).  -> 0 on failure.
static int pc_device_cus() {
    static std::mutex mu;
    static int cus[64] = {};   // 0 = not initialised, -1 = failed
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    std::lock_guard<std::mutex> lock(mu);
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && pc_set_lds()) ? prop.multiProcessorCount : -1;
    }
    return cus[dev] > 0 ? cus[dev] : 0;
}

static int wino_fwd_launch(const float* x, const float* u, float* y, int N, int H, int W, int C, int K, int accumulate,
                           const float* bias, const float* residual, int relu, void* stream) {
    WinoParams p{};
    p.x = x; p.u = u; p.y = y; p.N = N; p.H = H; p.W = W; p.C = C; p.K = K;
    p.TH = H / 2; p.TW = W / 2;
    p.BTW = p.TW >= 8 ? 8 : p.TW;
    p.BTH = WG_TILES / p.BTW;
    p.rows_total = N * p.TH;
    p.col_blocks = p.TW / p.BTW;
    p.x_bytes = (unsigned)((size_t)N * H * W * C * 4);
    p.u_bytes = (unsigned)((size_t)16 * K * C * 4);
    p.accumulate = accumulate;
    p.bias = bias; p.residual = residual; p.relu = relu;
    { const char* e = getenv("DS6G_WINO_DBG"); p.dbg = e ? atoi(e) : 0; }
    const int row_blocks = (p.rows_total + p.BTH - 1) / p.BTH;
    static const int pc_on = [] { const char* e = getenv("DS6G_WINO_PC"); return e ? atoi(e) : 1; }();
    const bool use_pc = pc_on && K % PC_KB == 0 && C >= 2 * WG_CH && (size_t)N * H * W * C * 4 + (size_t)(W + 1) * C * 4 < 0x40000000u;
    // CUs of the CURRENT device = persistent workgroups of winograd_pc_kernel; the LDS attribute is per device too
    int n_cu = 0;
    if (use_pc && (n_cu = pc_device_cus()) <= 0) return DS6G_ERR_LAUNCH;
    // profiler variant 20000 (20002: the producer / consumer kernel): forward / data gradient; flops = those of the direct
    // 3x3 convolution it replaces
    void* rec = ds6g_prof_open(use_pc ? 20002 : 20000, 2.0 * N * H * W * (double)K * 9.0 * C, (hipStream_t)stream);
    if (use_pc && K % PC_KB == 0 && C >= 2 * WG_CH) {
        p.items = row_blocks * p.col_blocks * (K / PC_KB);
        p.btw_shift = __builtin_ctz((unsigned)p.BTW);
        p.th_magic = (unsigned)(((1ull << 32) + (unsigned)p.TH - 1) / (unsigned)p.TH);
        // one persistent workgroup per CU.  DS6G_PC_CU_RESERVE=n leaves n CUs out: a workgroup needs a whole CU (all its
        // VGPRs, 128 KiB of LDS), so when other resident kernels (RCCL channels under data parallelism) hold a few CUs, a
        // full-size grid would run its last workgroups in a second round
        static const int reserve = [] { const char* e = getenv("DS6G_PC_CU_RESERVE"); return e ? atoi(e) : 0; }();
        const int cus = n_cu - reserve > 1 ? n_cu - reserve : 1;
        // the smallest grid that needs no more rounds than all CUs would: at the model's batch (60 frames) every layer has a
        // multiple of 240 items, so 240 workgroups are as fast as 256 (measured: 175.5 vs 176.1 samples/s) and 16 CUs stay
        // free for whatever else is resident
        const int rounds = (p.items + cus - 1) / cus;
        const int grid = (p.items + rounds - 1) / rounds;
        // XCD-aware item order (see the kernel): gk channel-block groups x gt = 8 / gk tile-block groups of XCDs; per launch the
        // L2s then fetch about |U| gt + |x| gk bytes (every XCD of a tile-block group reads that group's x once per channel
        // group it is in; every U slice is read by the gt XCDs that share it) - pick the minimum.  DS6G_PC_XCD=0 switches
        // it off, 1 / 2 / 4 / 8 force gk.
        {
            static const int xcd_env = [] { const char* e = getenv("DS6G_PC_XCD"); return e ? atoi(e) : -1; }();
            const int kblocks = K / PC_KB, ntb = p.items / kblocks;
            p.xcd_gk = 0;
            if (xcd_env != 0 && grid % 8 == 0 && p.items % grid == 0) {
                double best = 0;
                for (int gk = 1; gk <= 8; gk *= 2) {
                    const int gt = 8 / gk;
                    if (kblocks % gk || ntb % gt || (xcd_env > 0 && gk != xcd_env)) continue;
                    const double cost = (double)p.u_bytes * gt + (double)p.x_bytes * gk;
                    if (!p.xcd_gk || cost < best) { best = cost; p.xcd_gk = gk; }
                }
            }
        }
#ifdef DS6G_WINO_ABLATE
        static unsigned long long* tdbg = nullptr;
        if ((p.dbg & 64) && !tdbg) { (void)hipMalloc(&tdbg, 64 * 8 * 8); }
        p.tdbg = tdbg;
        switch (p.dbg) {
#define PC_CASE(D) case D: hipLaunchKernelGGL(winograd_pc_kernel<D>, dim3((unsigned)grid), dim3(512), PC_LDS_BYTES, (hipStream_t)stream, p); break;
            PC_CASE(1) PC_CASE(2) PC_CASE(3) PC_CASE(4) PC_CASE(7) PC_CASE(8) PC_CASE(16) PC_CASE(24) PC_CASE(9) PC_CASE(10) PC_CASE(18) PC_CASE(17) PC_CASE(19) PC_CASE(32) PC_CASE(64) PC_CASE(80) PC_CASE(96)
#undef PC_CASE
            default: hipLaunchKernelGGL(winograd_pc_kernel<0>, dim3((unsigned)grid), dim3(512), PC_LDS_BYTES, (hipStream_t)stream, p);
        }
        if (p.dbg & 64) {
            static int shown = 0;
            if (shown++ % 64 == 3) {  // a warm launch of each shape when the caller loops a few dozen times
                unsigned long long h[64 * 8];
                (void)hipDeviceSynchronize();
                (void)hipMemcpy(h, tdbg, sizeof(h), hipMemcpyDeviceToHost);
                const int T = ((p.items + grid - 1) / grid) * (C / WG_CH);
                fprintf(stderr, "pc steps (C=%d K=%d H=%d), cycles\n", C, K, H);
                for (int t = 1; t < T && t < 64; ++t) {
                    const long long m0 = (long long)h[(t - 1) * 8 + 3];  // mover left the previous barrier
                    fprintf(stderr, "  %2d: step %6lld | mult work %6lld wait %6lld | mover work %6lld wait %6lld\n", t,
                            (long long)(h[t * 8 + 1] - h[(t - 1) * 8 + 1]), (long long)(h[t * 8 + 0] - h[(t - 1) * 8 + 1]),
                            (long long)(h[t * 8 + 1] - h[t * 8 + 0]), (long long)h[t * 8 + 2] - m0, (long long)(h[t * 8 + 3] - h[t * 8 + 2]));
                }
            }
        }
#else
        hipLaunchKernelGGL(winograd_pc_kernel<0>, dim3((unsigned)grid), dim3(512), PC_LDS_BYTES, (hipStream_t)stream, p);
#endif
    } else if (K % (2 * WG_KB) == 0 && g_wino_kb64) {
        const long blocks = (long)row_blocks * p.col_blocks * (K / (2 * WG_KB));
        hipLaunchKernelGGL(winograd_fwd_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    } else {
        const long blocks = (long)row_blocks * p.col_blocks * (K / WG_KB);
        hipLaunchKernelGGL(winograd_fwd_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    }
    ds6g_prof_close(rec, (hipStream_t)stream);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

// 1 if ds6g_conv3x3_winograd_wgrad supports the shape
int ds6g_winograd_wgrad_supported(int N, int H, int W, int C, int K) {
    if (H % 2 || W % 2 || C % WW_CB || K % WW_KB || N <= 0) return 0;
    const size_t xb = (size_t)N * H * W * C * 4 + (size_t)(W + 1) * C * 4, yb = (size_t)N * H * W * K * 4;
    return xb < 0x40000000u && yb < OOB_OFF;
}

// dw[K][3][3][C] (+)= weight gradient of the 3x3 / stride 1 / pad 1 conv from x [N][H][W][C] and dy [N][H][W][K];
// ws holds the per-split dU slabs (any size >= one slab of 16*K*C floats; more allows more splits)
int ds6g_conv3x3_winograd_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K,
                                int accumulate, float* ws, size_t ws_bytes, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(x && dy && dw && ws && ds6g_winograd_wgrad_supported(N, H, W, C, K));
    const size_t slab = (size_t)16 * K * C * sizeof(float);
    DS6G_CHECK_ARG(ws_bytes >= slab);
    WinoWgradParams p{};
    p.x = x; p.dy = dy; p.du = ws; p.N = N; p.H = H; p.W = W; p.C = C; p.K = K;
    p.TH = H / 2; p.TW = W / 2;
    p.tiles_total = N * p.TH * p.TW;
    p.x_bytes = (unsigned)((size_t)N * H * W * C * 4);
    p.dy_bytes = (unsigned)((size_t)N * H * W * K * 4);
    const int base = (K / WW_KB) * (C / WW_CB) * 4;
    const int chunks = (p.tiles_total + WW_T - 1) / WW_T;
    // ~1024 workgroups, at least 12 chunks per split, bounded by the scratch
    long splits = (1024 + base - 1) / base;
    if (splits > chunks / 12) splits = chunks / 12;
    const long by_ws = (long)(ws_bytes / slab);
    if (splits > by_ws) splits = by_ws;
    if (splits < 1) splits = 1;
    const int cps = (int)((chunks + splits - 1) / splits);
    p.tiles_per_split = cps * WW_T;
    splits = (chunks + cps - 1) / cps;
    { static const int xcd_env = [] { const char* e = getenv("DS6G_WW_XCD"); return e ? atoi(e) : 1; }(); p.xcd = xcd_env; }
    void* rec = ds6g_prof_open(20001, 2.0 * N * H * W * (double)K * 9.0 * C, (hipStream_t)stream);
    hipLaunchKernelGGL(winograd_wgrad_kernel, dim3((unsigned)((K / WW_KB) * (C / WW_CB)), 4, (unsigned)splits), dim3(256), 0,
                       (hipStream_t)stream, p);
    ds6g_prof_close(rec, (hipStream_t)stream);
    DS6G_LAUNCH_CHECK();
    const long n = (long)K * C;
    hipLaunchKernelGGL(winograd_wgrad_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)ws, dw, K, C, (int)splits, accumulate);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

}  // extern "C"
