// Implicit-GEMM convolution / linear kernels on bf16-STORED operands (the real bf16 path of BASELINE configs[1] / [4]:
// activations and a bf16 shadow of the weights live in HBM as bf16, accumulation is fp32, master weights / statistics /
// loss / optimizer stay fp32).  Same three products as igemm.hip,
//   FWD   : y[m][o]      = sum_{r,s,c} x[n, oh*st-p+r, ow*st-p+s, c] * w[o][r][s][c]
//   DGRAD : dx[m][c]     = sum_{r,s,o} dy[n, (h+p-r)/st, (w+p-s)/st, o] * w[o][r][s][c]
//   WGRAD : dw[o][r,s,c] = sum_{pixels} dy[pixel][o] * x[n, oh*st-p+r, ow*st-p+s, c]         (split-K, fp32 slabs)
// on NHWC bf16 activations and OHWI bf16 weights; a Linear layer is the 1x1 case over ONE 1 x M "image".
// Reference call sites: model2_seq.py:510-512,528-530,546-548,565-567 (BasicBlock convs), :83-90,:97-99,:109,:121-126
// (GPT linears); the reference itself has no reduced-precision mode (train2_seq.py:111-116 casts to fp32).
//
// What differs from igemm.hip (fp32 tiles, operands rounded on the way into the MFMA):
//   * tiles travel HBM -> LDS as bf16 by LDS-DMA (half the bytes), k-tile = 64 elements, and the fragments are fed to
//     v_mfma_f32_32x32x16_bf16 exactly as they lie in LDS - no conversion, no VALU in the MFMA loop;
//   * K-contiguous sources (im2col rows of x / dy, weight rows of the forward) use a [row][64] image of 128-byte rows whose
//     16-B chunk index is XOR-swizzled by (row >> 1) & 7 on the SOURCE address: one conflict-free ds_read_b128 per 32-row
//     fragment and 16-deep k-step;
//   * row-contiguous sources (both wgrad operands, the dgrad weights) are staged as they lie in memory, [k][cols], and
//     read TRANSPOSED by ds_read_b64_tr_b16 (gfx950's LDS transpose read: two per fragment and k-step); the 64-B column
//     segment index is XOR-swizzled by the k row so that the four rows of one transposed read hit different banks;
//   * a bf16 output tile is rounded once from the fp32 accumulators and leaves through LDS, so that rows are stored as
//     16-byte pieces instead of 2-byte elements.
// Only the wave-uniform k walk of igemm.hip exists here (channel counts multiples of 64; wgrad: see bgemm_wgrad_walk):
// every layer of the model but the 4-channel stems qualifies, and the stems keep the fp32-storage kernel.
#include "common.h"
#include <cstdlib>

void* ds6g_prof_open(int variant, double flops, hipStream_t st);
void ds6g_prof_close(void* rec, hipStream_t st);
// igemm.hip: sums split-K slabs (weight gradient + optional bias-gradient tail) into the fp32 gradient
int ds6g_internal_splitk_reduce(const float* ws, float* dw, long n4, float* dbias, long m4, int splits, size_t stride,
                                int accumulate, int accumulate_b, hipStream_t st);
// norm.hip: mean / invstd (+ running statistics) from nblk row-block partials [nblk][2][C] (fp64)
int ds6g_internal_bn_stats_finalize(const double* partial, int nblk, long M, int C, float eps, float momentum, float* mean,
                                    float* invstd, float* running_mean, float* running_var, hipStream_t st);

GCLK_STORAGE(g_bgemm_clk, g_bgemm_wg, ds6g_bgemm_clocks_read)

namespace {

enum { B_FWD = 0, B_DGRAD = 1, B_WGRAD = 2 };
constexpr int BK = 64;  // k-tile depth in elements (128 B of a K-contiguous row; 64 rows of a k-major image)

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct BgemmParams {
    const __bf16* a_src;
    const __bf16* b_src;
    void* out;            // float* (fp32 output / split-K slabs) or __bf16*
    int N, H, W, C;       // input-side tensor (x / dx), NHWC
    int Ho, Wo, K;        // output-side tensor (y / dy)
    int R, S, stride, pad;
    int Mg, Ng, Kg;
    const float* bias;
    const float* residual;   // fp32 [Mg][Ng] (the GPT residual stream stays fp32)
    const void* mask_src;    // ReLU mask source [Mg][Ng], fp32 or bf16 (mask16)
    int mask16;
    int relu;
    int accumulate;
    uint32_t drop_thr;
    float drop_scale;
    uint64_t seed, seed_off;
    const uint64_t* salt;    // device-resident addend of seed_off (nullable), see ds6g_set_dropout_salt
    int k_per_split;      // multiple of 64
    size_t split_stride;
    int tiles_n;
    int wg_rows;          // WGRAD: 0 = a k-tile of 64 pixels stays inside one output row; else output rows per k-tile
    int want_colsum;
    int xcd_splits;       // WGRAD: every XCD walks a contiguous run of (split, tile) pairs, tile fastest (see the kernel)
    unsigned a_bytes, b_bytes;
    // DGRAD of a strided conv runs per input-pixel parity class (blockIdx.y when nclass > 1, parameters derived in the
    // kernel): pixels h = h0 + hstep*hh (hh < Hs), taps r = r0 + rstep*ri (ri < nr) - only the taps that hit a real
    // output pixel, no structural zeros (as igemm.hip)
    int nclass;
    int h0, hstep, Hs, w0, wstep, Ws, r0, rstep, nr, s0, sstep, ns;
    // FWD, bf16 output, no epilogue: per-tile-row column sums / sums of squares of the STORED (rounded) tile for the
    // train-mode BatchNorm that follows the conv - partial[tile_m][0][col], partial[tile_m][1][col] (fp64, the layout
    // bn_stats_finalize_kernel reads), so the separate statistics pass over the conv output is not needed
    double* bn_partial;
};

__device__ __forceinline__ unsigned lds_off(const void* p) { return (unsigned)(unsigned long)(lds_void*)p; }

// MODE / BM x BN tile / OUT16: bf16 output through LDS (else fp32, direct) / EPI: fused epilogue (bias, ReLU, mask,
// dropout, residual, accumulate) instead of a plain store.
// What bounds the loop (round 4, -DDS6G_GEMM_CLOCKS + tools/gemm_clocks.py): a k-tile step of the 128 x 128 tile takes 2 800 -
// 3 000 cycles for 2 x 512 cycles of MFMA per SIMD (two workgroups per CU): DMA issue 800 - 1 000 (8 pieces per wave), fragment
// reads + MFMAs 1 100 - 1 200, DMA wait 550, barrier 330.  A 256 x 128 tile with 8 waves and THREE stages (two k-tiles in flight,
// counted vmcnt, raw s_barrier) was built and measured: 975 vs 964 TFLOP/s at 4096^3, 46.0 vs 43.6 us on the stage-4 fc1
// forward - the wait stays ~800 cycles per step with twice the prefetch distance, i.e. the L1 / texture-address load path
// (32 KB per step and workgroup = 24 B / cycle / CU at 964 TFLOP/s) bounds it, not latency; only a tile with fewer bytes per
// FLOP (256 x 256) would move it, and M = 11 544 rows / N <= 2 048 give that tile 92 - 368 workgroups on 256 CUs.  Not kept.
// Also built, parity-tested and measured: a producer / consumer form of the 128 x 128 tile (512 threads: waves 4 - 7 only issue
// the DMA two tiles ahead, waves 0 - 3 only multiply, three stages, one workgroup per CU) - 812 vs 1 004 TFLOP/s at 4096^3,
// slower on every model shape but the 240-workgroup 16x16x256 layers (28.3 -> 23.7 us): the step is then paced by the producer
// wave's own issue chain (8 pieces = 800 - 1 000 cycles, i.e. ~27 cycles per KiB and CU through the texture-address unit),
// which is the same load-path limit seen from the other side (profiles/r04_bgemm_classic_vs_producer_consumer.txt).  Not kept.
template <int MODE, int BM, int BN, int OUT16, int EPI>
__global__ __launch_bounds__(256) void bgemm_kernel(const BgemmParams pin) {
    BgemmParams p = pin;
    GCLK_DECL(g_bgemm_wg);
    if (EPI && p.drop_thr && p.salt) p.seed_off += *p.salt;
    if (MODE == B_DGRAD && pin.nclass > 1) {
        const int ph = blockIdx.y >> 1, pw = blockIdx.y & 1;
        p.h0 = ph;
        p.w0 = pw;
        p.r0 = (ph + p.pad) & 1;
        p.nr = p.r0 < p.R ? (p.R - p.r0 + 1) / 2 : 0;
        p.s0 = (pw + p.pad) & 1;
        p.ns = p.s0 < p.S ? (p.S - p.s0 + 1) / 2 : 0;
        p.Kg = p.nr * p.ns * p.K;
        if (p.ns == 0) p.ns = 1;
        if (p.Kg == 0 && p.accumulate) return;  // no tap hits this class: its pixels keep their value
    }
    constexpr int TM = BM / 64, TN = BN / 64;          // 32x32 MFMA tiles per wave (wave grid 2 x 2)
    constexpr int WR = BM / 2;                         // rows of the output tile per wave
    constexpr bool A_T = (MODE == B_WGRAD);            // operand staged [k][cols], read transposed
    constexpr bool B_T = (MODE != B_FWD);
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    constexpr int A_LD = A_BYTES / 4096, B_LD = B_BYTES / 4096;  // 1-KiB DMA pieces per wave and k-tile
    constexpr int A_RB = BM * 2, B_RB = BN * 2;                  // row bytes of a k-major image
    constexpr int STAGE = A_BYTES + B_BYTES;
    // ONE shared object (a second one beside LDS-DMA staging makes hipcc drain vmcnt before every fragment read)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, khalf = lane >> 5;
    int wg, split = blockIdx.z;
    if (MODE == B_WGRAD && p.xcd_splits) {
        // the hardware deals workgroups to the 8 XCDs (private L2s) in dispatch order, x fastest, then z.  The tiles of ONE
        // split read the same dy rows and the same / overlapping x rows: give every XCD a contiguous run of (split, tile)
        // pairs, tile fastest, so that a split's operands are fetched by one L2 (two at a run boundary) instead of eight
        const int nwg = gridDim.x * gridDim.z, orig = blockIdx.x + gridDim.x * blockIdx.z;
        const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
        split = lin / (int)gridDim.x;
        wg = lin - split * (int)gridDim.x;
    } else {   // XCD-aware tile order (bijective for any grid), as igemm.hip
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int tile_m = wg / p.tiles_n, tile_n = wg - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbegin = (MODE == B_WGRAD) ? split * p.k_per_split : 0;
    const int kend = (MODE == B_WGRAD) ? min(p.Kg, kbegin + p.k_per_split) : p.Kg;
    const int nk = (kend - kbegin + BK - 1) / BK;

    const i32x4 a_rsrc = make_srd(p.a_src, p.a_bytes);
    i32x4 b_rsrc = make_srd(p.b_src, p.b_bytes);

    // ---- lane -> (row, element offset) of its 16-B slot in DMA piece `pi` of an image ----------------------------
    // K image [rows][64]: piece = 8 rows x 128 B; slot (row, phys chunk) holds logical chunk (phys ^ (row >> 1) & 7)
    auto kimg_row = [&](int pi) { return pi * 8 + (lane >> 3); };
    auto kimg_kel = [&](int pi) { return (((lane & 7) ^ ((kimg_row(pi) >> 1) & 7)) << 3); };
    // T image [64 k rows][cols], RB bytes per row: piece = 1024 / RB rows; 64-B segment index swizzled by the row
    auto timg_row = [&](int pi, int RB) { return pi * (1024 / RB) + lane / (RB / 16); };
    auto timg_cel = [&](int pi, int RB) {
        const int row = timg_row(pi, RB), slot = lane % (RB / 16);
        const int seg = slot >> 2, within = slot & 3;
        const int lseg = (RB == 256) ? (seg ^ (row & 3)) : (RB == 128 ? (seg ^ ((row >> 1) & 1)) : seg);
        return (lseg * 4 + within) * 8;
    };

    // ---- wave-uniform walk state (SGPRs) + per-lane constant byte offsets, as igemm.hip's FAST walk ---------------
    int u_c0 = 0, u_r = 0, u_s = 0;
    unsigned u_kb = 0;
    int u_kpos = kbegin, u_n = 0, u_oh = 0, u_ow = 0;
    unsigned a_vo[A_LD], b_vo[B_LD];
    [[maybe_unused]] int a_y[A_LD], a_x[A_LD], a_c[A_LD];
    [[maybe_unused]] unsigned a_base[A_LD];
    [[maybe_unused]] bool a_ok[A_LD], b_ok[B_LD];
    [[maybe_unused]] int a_kr[A_LD], b_kr[B_LD], b_ihl[B_LD], b_iwl[B_LD];

    auto retap = [&]() {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            if (MODE == B_FWD) {
                const int ih = a_y[i] + u_r, iw = a_x[i] + u_s;
                const bool ok = a_ok[i] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                a_vo[i] = ok ? (a_base[i] + (unsigned)((ih * p.W + iw) * p.C + a_c[i])) * 2u : OOB_OFF;
            } else if (MODE == B_DGRAD) {
                const int th = a_y[i] - (p.r0 + u_r * p.rstep), tw = a_x[i] - (p.s0 + u_s * p.sstep);
                int oh = th, ow = tw;
                bool okk = th >= 0 && tw >= 0;
                if (p.stride == 2) {
                    oh = th >> 1;
                    ow = tw >> 1;
                    okk = okk && !((th | tw) & 1);
                }
                const bool ok = a_ok[i] && okk && oh < p.Ho && ow < p.Wo;
                a_vo[i] = ok ? (a_base[i] + (unsigned)((oh * p.Wo + ow) * p.K + a_c[i])) * 2u : OOB_OFF;
            }
        }
    };

    if (MODE == B_FWD || MODE == B_DGRAD) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int pi = i * 4 + wave;
            const int m = m0 + kimg_row(pi);
            a_ok[i] = m < p.Mg;
            const int mm = a_ok[i] ? m : 0;
            a_c[i] = kimg_kel(pi);
            if (MODE == B_FWD) {
                int ow, t, oh, n;
                fast_divmod(mm, p.Wo, t, ow);
                fast_divmod(t, p.Ho, n, oh);
                a_y[i] = oh * p.stride - p.pad;
                a_x[i] = ow * p.stride - p.pad;
                a_base[i] = (unsigned)n * (unsigned)(p.H * p.W * p.C);
            } else {
                int ww, t, hh, n;
                fast_divmod(mm, p.Ws, t, ww);
                fast_divmod(t, p.Hs, n, hh);
                a_y[i] = p.h0 + hh * p.hstep + p.pad;
                a_x[i] = p.w0 + ww * p.wstep + p.pad;
                a_base[i] = (unsigned)n * (unsigned)(p.Ho * p.Wo * p.K);
            }
        }
        retap();
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const int pi = i * 4 + wave;
            if (MODE == B_FWD) {
                const int n = n0 + kimg_row(pi);
                b_vo[i] = n < p.Ng ? ((unsigned)n * (unsigned)p.Kg + (unsigned)kimg_kel(pi)) * 2u : OOB_OFF;
            } else {  // weights as [k = out channel][col = in channel] of the current tap
                const int col = n0 + timg_cel(pi, B_RB);
                b_vo[i] = col < p.Ng ? ((unsigned)(timg_row(pi, B_RB) * (p.R * p.S) * p.C) + (unsigned)col) * 2u : OOB_OFF;
            }
        }
    } else {
        {   // pixel kbegin -> (image, output row, output column)
            const int t = kbegin / p.Wo;
            u_ow = p.wg_rows == 0 ? kbegin % p.Wo : 0;
            u_oh = t % p.Ho;
            u_n = t / p.Ho;
        }
        const int shift = (p.pad * p.W + p.pad) * p.C;  // negative tap offsets folded into the descriptor base
        b_rsrc = make_srd(p.b_src - shift, p.b_bytes + (unsigned)shift * 2u);
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int pi = i * 4 + wave;
            a_kr[i] = timg_row(pi, A_RB);
            const int col = m0 + timg_cel(pi, A_RB);
            a_ok[i] = col < p.Mg;
            a_vo[i] = ((unsigned)(a_kr[i] * p.K) + (unsigned)col) * 2u;
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const int pi = i * 4 + wave;
            b_kr[i] = timg_row(pi, B_RB);
            const int col = n0 + timg_cel(pi, B_RB);
            b_ok[i] = col < p.Ng;
            const int cc = b_ok[i] ? col : 0;
            const int tap = cc / p.C, c = cc - tap * p.C;
            const int br = tap / p.S, bs = tap - br * p.S;
            const int dl_oh = p.wg_rows == 0 ? 0 : b_kr[i] / p.Wo;
            const int dl_ow = p.wg_rows == 0 ? b_kr[i] : b_kr[i] % p.Wo;
            b_ihl[i] = dl_oh * p.stride - p.pad + br;
            b_iwl[i] = dl_ow * p.stride - p.pad + bs;
            b_vo[i] = (unsigned)(c + (b_ihl[i] * p.W + b_iwl[i]) * p.C + shift) * 2u;
        }
    }

    // DMA of the next k-tile into stage `st`; the walk state advances by one k-tile
    auto issue_tiles = [&](int st) {
        const unsigned la = lds_off(lds + st * STAGE), lb = la + A_BYTES;
        if (MODE == B_FWD || MODE == B_DGRAD) {
            const unsigned sa = (unsigned)u_c0 * 2u;
            const unsigned sb = (MODE == B_FWD) ? u_kb
                                                : (unsigned)((u_c0 * (p.R * p.S) + (p.r0 + u_r * p.rstep) * p.S + p.s0 + u_s * p.sstep) * p.C) * 2u;
#pragma unroll
            for (int i = 0; i < A_LD; ++i) dma16(a_rsrc, la + (unsigned)(i * 4 + wave) * 1024u, a_vo[i], sa);
#pragma unroll
            for (int i = 0; i < B_LD; ++i) dma16(b_rsrc, lb + (unsigned)(i * 4 + wave) * 1024u, b_vo[i], sb);
            u_kb += BK * 2;
            u_c0 += BK;
            if (u_c0 == ((MODE == B_FWD) ? p.C : p.K)) {
                u_c0 = 0;
                if (++u_s == ((MODE == B_FWD) ? p.S : p.ns)) { u_s = 0; ++u_r; }
                retap();
            }
        } else {
            const int rows_left = kend - u_kpos;
            const unsigned sa = (unsigned)(u_kpos * p.K) * 2u;
            const int ihu = u_oh * p.stride, iwu = u_ow * p.stride;
            const unsigned sb = (unsigned)(((u_n * p.H + ihu) * p.W + iwu) * p.C) * 2u;
            if (rows_left >= BK && p.R == 1 && p.pad == 0) {
                // whole k-tile of a filter without halo (the linears, the 1x1 convs): every test below is the lane constant
                // of the set-up - no per-piece VALU between the DMA instructions
#pragma unroll
                for (int i = 0; i < A_LD; ++i) dma16(a_rsrc, la + (unsigned)(i * 4 + wave) * 1024u, a_ok[i] ? a_vo[i] : OOB_OFF, sa);
#pragma unroll
                for (int i = 0; i < B_LD; ++i) dma16(b_rsrc, lb + (unsigned)(i * 4 + wave) * 1024u, b_ok[i] ? b_vo[i] : OOB_OFF, sb);
            } else {
#pragma unroll
            for (int i = 0; i < A_LD; ++i) {
                const bool ok = a_ok[i] && a_kr[i] < rows_left;
                dma16(a_rsrc, la + (unsigned)(i * 4 + wave) * 1024u, ok ? a_vo[i] : OOB_OFF, sa);
            }
#pragma unroll
            for (int i = 0; i < B_LD; ++i) {
                const bool ok = b_ok[i] && b_kr[i] < rows_left && (unsigned)(ihu + b_ihl[i]) < (unsigned)p.H &&
                                (unsigned)(iwu + b_iwl[i]) < (unsigned)p.W;
                dma16(b_rsrc, lb + (unsigned)(i * 4 + wave) * 1024u, ok ? b_vo[i] : OOB_OFF, sb);
            }
            }
            u_kpos += BK;
            if (p.wg_rows == 0) {
                u_ow += BK;
                if (u_ow >= p.Wo) {
                    u_ow = 0;
                    if (++u_oh == p.Ho) { u_oh = 0; ++u_n; }
                }
            } else {
                u_oh += p.wg_rows;
                if (u_oh >= p.Ho) { u_oh = 0; ++u_n; }
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

    // ---- fragment addresses (bytes inside an operand image) --------------------------------------------------------
    // K image: lane (row, half) reads logical chunk 2s + half of its row for k-step s: one ds_read_b128
    const unsigned ka_row = (unsigned)((wm * WR + l31) * 128), kb_row = (unsigned)((wn * (BN / 2) + l31) * 128);
    const unsigned kswz = (unsigned)((l31 >> 1) & 7);
    // T image: lane 4q + pp of a 16-lane group addresses row 16s + 8 half + 4t + q, columns c0 + 16 (group & 1) + 4 pp .. + 3
    const int tq = (lane >> 2) & 3, tpp = lane & 3, tgrp = (lane >> 4) & 1;
    auto t_addr = [&](int RB, int cbase, int s, int t) {
        const int row = 16 * s + 8 * khalf + 4 * t + tq;
        const int col = cbase + 16 * tgrp + 4 * tpp;
        const int seg = col >> 5;
        const int pseg = (RB == 256) ? (seg ^ (row & 3)) : (RB == 128 ? (seg ^ ((row >> 1) & 1)) : seg);
        return (unsigned)(row * RB + pseg * 64 + (col & 31) * 2);
    };

    float csum = 0.f;  // WGRAD bias gradient: column sums of the dy tile (threads tid < BM of the tile_n == 0 blocks)
    const bool do_csum = (MODE == B_WGRAD) && p.want_colsum && tile_n == 0 && tid < BM;

    auto compute = [&](int st) {
        const unsigned char* Ac = lds + st * STAGE;
        const unsigned char* Bc = Ac + A_BYTES;
        if (MODE == B_WGRAD && do_csum) {
            const int seg = tid >> 5;
#pragma unroll 8
            for (int k = 0; k < BK; ++k) {
                const int pseg = (A_RB == 256) ? (seg ^ (k & 3)) : (seg ^ ((k >> 1) & 1));
                csum += (float)*reinterpret_cast<const __bf16*>(Ac + k * A_RB + pseg * 64 + (tid & 31) * 2);
            }
        }
        bf16x8 af[4][TM], bfr[4][TN];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (A_T) {
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4*)(lds_void*)(Ac + t_addr(A_RB, wm * WR + i * 32, s, 0)));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4*)(lds_void*)(Ac + t_addr(A_RB, wm * WR + i * 32, s, 1)));
                    af[s][i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                } else {
                    af[s][i] = *reinterpret_cast<const bf16x8*>(Ac + ka_row + i * 32 * 128 + ((((unsigned)(2 * s + khalf)) ^ kswz) << 4));
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (B_T) {
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4*)(lds_void*)(Bc + t_addr(B_RB, wn * (BN / 2) + j * 32, s, 0)));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4*)(lds_void*)(Bc + t_addr(B_RB, wn * (BN / 2) + j * 32, s, 1)));
                    bfr[s][j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                } else {
                    bfr[s][j] = *reinterpret_cast<const bf16x8*>(Bc + kb_row + j * 32 * 128 + ((((unsigned)(2 * s + khalf)) ^ kswz) << 4));
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][i], bfr[s][j], acc[i][j], 0, 0, 0);
    };

    // two stages: the DMA of tile t+1 flies under the MFMAs of tile t; one wait + barrier per k-tile
    // the epilogue's bias values, fetched BEFORE the k loop (their memory latency hides under it; loaded at their use they
    // were most of an 11 000-cycle epilogue in a 40 000-cycle workgroup life, -DDS6G_GEMM_CLOCKS)
    [[maybe_unused]] float bias_pre[TN];
    if (EPI) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (BN / 2) + j * 32 + l31;
            bias_pre[j] = (p.bias && col < p.Ng) ? p.bias[col] : 0.f;
        }
    }
    GCLK(g_bgemm_clk, 0);   // set-up (walk state, fragment addresses, accumulator zeroing)
    if (nk > 0) issue_tiles(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    GCLK(g_bgemm_clk, 1);   // first tile: DMA issue + its full latency + barrier
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) issue_tiles((kt + 1) & 1);
        GCLK(g_bgemm_clk, 2);   // DMA issue of the next k-tile
        compute(kt & 1);
        __builtin_amdgcn_sched_barrier(0);
        GCLK(g_bgemm_clk, 3);   // fragment reads + MFMA chain
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GCLK(g_bgemm_clk, 4);   // wait for the next tile's DMA
        __syncthreads();
        GCLK(g_bgemm_clk, 5);   // barrier
        GCLK_COUNT(g_bgemm_clk, 15);
    }

    // ---- epilogue: D[row][col], col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) -------------------------
    // strided dgrad: rows of the parity-class sub-grid scatter to full-resolution pixels
    auto out_row = [&](int row) -> size_t {
        if (!(MODE == B_DGRAD && p.hstep != 1)) return (size_t)row;
        const int ww = row % p.Ws, t = row / p.Ws, hh = t % p.Hs;
        return ((size_t)(t / p.Hs) * p.H + p.h0 + hh * p.hstep) * p.W + p.w0 + ww * p.wstep;
    };
    if (!OUT16 && MODE == B_DGRAD && p.hstep != 1) {
        float* outp = reinterpret_cast<float*>(p.out);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * (BN / 2) + j * 32 + l31;
                if (col >= p.Ng) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * WR + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    if (row >= p.Mg) continue;
                    const size_t o = out_row(row) * p.Ng + col;
                    outp[o] = p.accumulate ? outp[o] + acc[i][j][r] : acc[i][j][r];
                }
            }
        GCLK(g_bgemm_clk, 6);
        GCLK_END(g_bgemm_clk, g_bgemm_wg);
        return;
    }
    if (!OUT16) {
        float* outp = reinterpret_cast<float*>(p.out) + ((MODE == B_WGRAD) ? (size_t)split * p.split_stride : (size_t)0);
        if (MODE == B_WGRAD && do_csum && m0 + tid < p.Mg) outp[(size_t)p.Mg * p.Ng + m0 + tid] = csum;
        const unsigned nbytes = (unsigned)p.Mg * (unsigned)p.Ng * 4u;
        const auto o_rsrc = __builtin_amdgcn_make_buffer_rsrc(outp, 0, nbytes, 0x00020000);
        const auto r_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.residual ? p.residual : outp), 0, nbytes, 0x00020000);
        const unsigned rowbytes = (unsigned)p.Ng * 4u;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * (BN / 2) + j * 32 + l31;
                const int row0 = m0 + wm * WR + i * 32 + 4 * khalf;
                const bool cok = col < p.Ng;
                const float bias = EPI ? bias_pre[j] : 0.f;
                const unsigned base = cok ? (unsigned)row0 * rowbytes + (unsigned)col * 4u : OOB_OFF;
                const unsigned e0 = (unsigned)row0 * (unsigned)p.Ng + (unsigned)col;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned dr = (unsigned)((r & 3) + 8 * (r >> 2));
                    const unsigned off = base + dr * rowbytes;
                    float v = acc[i][j][r];
                    if (EPI) {
                        v += bias;
                        if (p.relu == 1) v = fmaxf(v, 0.f);
                        if (p.drop_thr)
                            v = ds6g_keep(p.seed, p.seed_off + (uint64_t)(e0 + dr * (unsigned)p.Ng), p.drop_thr) ? v * p.drop_scale : 0.f;
                        if (p.residual) v += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_rsrc, off, 0, 0));
                    }
                    if (MODE != B_WGRAD && p.accumulate) v += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(o_rsrc, off, 0, 0));
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), o_rsrc, off, 0, 0);
                }
            }
        }
        GCLK(g_bgemm_clk, 6);   // epilogue (stores issued)
        GCLK_END(g_bgemm_clk, g_bgemm_wg);
        return;
    }
    // bf16 output: epilogue arithmetic in fp32 on the accumulators, one rounding, then through a wave-private LDS patch
    // [BM/2 rows][BN/2 cols] so that global stores are 16-byte row pieces (the loop's last barrier has been passed by
    // every wave: the staging buffers are free)
    {
        __bf16* outp = reinterpret_cast<__bf16*>(p.out);
        constexpr int PR = WR, PC = BN / 2;
        __bf16* patch = reinterpret_cast<__bf16*>(lds) + wave * PR * PC;
        const bool stats = !EPI && MODE == B_FWD && p.bn_partial != nullptr;
        float cs[TN], cq[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) cs[j] = cq[j] = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * PC + j * 32 + l31;
                const int row0 = m0 + wm * PR + i * 32 + 4 * khalf;
                const float bias = EPI ? bias_pre[j] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    float v = acc[i][j][r];
                    if (EPI) {
                        v += bias;
                        if (p.relu == 1) v = fmaxf(v, 0.f);
                        if (p.drop_thr)
                            v = ds6g_keep(p.seed, p.seed_off + (uint64_t)((size_t)(row0 + dr) * p.Ng + col), p.drop_thr) ? v * p.drop_scale : 0.f;
                    }
                    const __bf16 vb = (__bf16)v;
                    patch[(i * 32 + 4 * khalf + dr) * PC + j * 32 + l31] = vb;
                    if (!EPI && MODE == B_FWD) {   // rows >= Mg are products of zero-filled im2col rows: they add 0
                        const float vr = (float)vb;
                        cs[j] += vr;
                        cq[j] += vr * vr;
                    }
                }
            }
        }
        if (stats) {
            // column sums of the tile: lane halves (rows 4 khalf + ...) by a cross-lane add, the two row-waves through LDS
            // (above the four output patches: 4 * PR * PC * 2 <= half of the staging buffers)
            float* red = reinterpret_cast<float*>(lds + 4 * PR * PC * 2);   // [2 (wm)][2 (sum, sumsq)][BN]
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                cs[j] += __shfl_xor(cs[j], 32, 64);
                cq[j] += __shfl_xor(cq[j], 32, 64);
                if (khalf == 0) {
                    red[(wm * 2 + 0) * BN + wn * PC + j * 32 + l31] = cs[j];
                    red[(wm * 2 + 1) * BN + wn * PC + j * 32 + l31] = cq[j];
                }
            }
            __syncthreads();
            if (tid < BN && n0 + tid < p.Ng) {
                double* dst = p.bn_partial + (size_t)tile_m * 2 * p.Ng + n0 + tid;
                dst[0] = (double)red[tid] + (double)red[2 * BN + tid];
                dst[p.Ng] = (double)red[BN + tid] + (double)red[3 * BN + tid];
            }
        }
        // wave-private patch: the wave's own LDS writes are ordered before its reads by the compiler's lgkmcnt waits
        constexpr int PPR = PC / 8;  // 16-B pieces per patch row
#pragma unroll
        for (int t = 0; t < PR * PPR / 64; ++t) {
            const int idx = t * 64 + lane;
            const int rl = idx / PPR, pc = idx - rl * PPR;
            const int grow = m0 + wm * PR + rl, gcol = n0 + wn * PC + pc * 8;
            if (grow < p.Mg && gcol < p.Ng) {
                bf16x8 v = *reinterpret_cast<const bf16x8*>(patch + rl * PC + pc * 8);
                if (EPI && p.mask_src) {
                    // ReLU mask of the forward activation ([Mg][Ng], the output's own shape): applied here, where a lane holds
                    // 8 consecutive columns of one row - one 16-byte (bf16) or two 16-byte (fp32) loads instead of 8 scalar
                    // ones per lane in accumulator order (the masked dgrad ran 54.8 us against 27.7 us for the plain one)
                    const size_t e = (size_t)grow * p.Ng + gcol;
                    if (p.mask16) {
                        const bf16x8 mv = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(p.mask_src) + e);
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = (float)mv[q] > 0.f ? v[q] : (__bf16)0.f;
                    } else {
                        const float* mp = reinterpret_cast<const float*>(p.mask_src) + e;
                        const f32x4 m0v = *reinterpret_cast<const f32x4*>(mp), m1v = *reinterpret_cast<const f32x4*>(mp + 4);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            v[q] = m0v[q] > 0.f ? v[q] : (__bf16)0.f;
                            v[4 + q] = m1v[q] > 0.f ? v[4 + q] : (__bf16)0.f;
                        }
                    }
                }
                bf16x8* dst = reinterpret_cast<bf16x8*>(outp + out_row(grow) * p.Ng + gcol);
                if (p.accumulate) {
                    const bf16x8 old = *dst;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (__bf16)((float)v[e] + (float)old[e]);
                }
                *dst = v;
            }
        }
    }
    GCLK(g_bgemm_clk, 6);
    GCLK_END(g_bgemm_clk, g_bgemm_wg);
}

// ------------------------------------------------------------------------------------------------
// 128 x 128 tiles once they give this many workgroups, else 64 x 64 (tools/bench_bgemm.py with DS6G_BG_TILE forced: the large
// tile wins from ~240 workgroups up - 32x32 x 128 ch and 16x16 x 256 ch convs at N = 60 633-677 vs 536-562 TFLOP/s, stage-4 GPT
// linears +15-20 % - and loses on the 64-wide and the 8x8 x 512 layers, 120 workgroups)
int g_bg_min_blocks = 200;
// wgrad split-K: workgroups aimed at (measured sweep 128 .. 4096, tools/bench_bgemm.py with DS6G_BG_WTARGET: few, long splits
// win - convs peak at ~384, the GPT linears at ~512; 1024 costs 15-25 %, 2048+ up to 2x through the slab reduction)
int g_bg_wgrad_target = 0;
int g_bg_force_tile = -1;   // tuning experiments (env DS6G_BG_TILE / DS6G_BG_MINBLOCKS, read once)
// wgrad (split, tile) -> XCD mapping (env DS6G_BG_WXCD): 0 plain (tiles of each split dealt XCD-contiguously), 1 (default)
// contiguous (split, tile) runs where they measured faster (tools/bench_bwgrad.py, N = 60: convs with <= 24 output tiles -
// 64x64x64 75 -> 57 us / 499 -> 72 MB read, 32x32x128 45 -> 35 us / 247 -> 37 MB, the stride-2 layers 31 -> 25 us; the
// 36 .. 144-tile layers lose 1 - 2 us although their traffic falls 2 - 4x - and the large linears whose workgroup count is a
// multiple of 8, i.e. whole splits per XCD: fc1 / fc2 / proj of stage 4 44.8 -> 40.2 us, 142 - 213 -> 59 MB), 2 every conv,
// 3 everything
int g_bg_wxcd = 1;
void bg_env() {
    static bool done = false;
    if (done) return;
    done = true;
    if (const char* e = getenv("DS6G_BG_TILE")) g_bg_force_tile = atoi(e);
    if (const char* e = getenv("DS6G_BG_MINBLOCKS")) g_bg_min_blocks = atoi(e);
    if (const char* e = getenv("DS6G_BG_WTARGET")) g_bg_wgrad_target = atoi(e);
    if (const char* e = getenv("DS6G_BG_WXCD")) g_bg_wxcd = atoi(e);
}

void fill_conv(BgemmParams& p, int N, int H, int W, int C, int K, int R, int S, int stride, int pad) {
    p = BgemmParams{};
    p.N = N; p.H = H; p.W = W; p.C = C; p.K = K; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
    p.Ho = (H + 2 * pad - R) / stride + 1;
    p.Wo = (W + 2 * pad - S) / stride + 1;
    p.drop_scale = 1.f;
    p.h0 = 0; p.hstep = 1; p.Hs = H; p.w0 = 0; p.wstep = 1; p.Ws = W;
    p.r0 = 0; p.rstep = 1; p.nr = R; p.s0 = 0; p.sstep = 1; p.ns = S;
}

// wgrad: the 64 pixels of a k-tile share one (image, output row), or cover whole output rows of one image
bool bgemm_wgrad_walk(BgemmParams& p) {
    if (p.Wo % BK == 0 || p.N * p.Ho == 1) { p.wg_rows = 0; return true; }
    if (BK % p.Wo == 0 && p.Ho % (BK / p.Wo) == 0) { p.wg_rows = BK / p.Wo; return true; }
    return false;
}

template <int MODE, int OUT16, int EPI>
int launch_tiles(BgemmParams& p, int splits, int tile, hipStream_t st) {
    dim3 block(256);
    const int ny = p.nclass > 1 ? p.nclass : 1;
    if (tile == 0) {
        p.tiles_n = cdiv(p.Ng, 128);
        dim3 grid(cdiv(p.Mg, 128) * p.tiles_n, ny, splits);
        hipLaunchKernelGGL((bgemm_kernel<MODE, 128, 128, OUT16, EPI>), grid, block, 0, st, p);
    } else {
        p.tiles_n = cdiv(p.Ng, 64);
        dim3 grid(cdiv(p.Mg, 64) * p.tiles_n, ny, splits);
        hipLaunchKernelGGL((bgemm_kernel<MODE, 64, 64, OUT16, EPI>), grid, block, 0, st, p);
    }
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

template <int MODE>
int launch_bgemm(BgemmParams& p, int out16, int splits, int tile, hipStream_t st) {
    if ((size_t)p.Mg * p.Ng * 4 >= OOB_OFF) return DS6G_ERR_ARG;
    const bool epi = p.bias || p.relu || p.mask_src || p.drop_thr || p.residual;
    void* rec = ds6g_prof_open(30000 + 100 * out16 + 10 * MODE + tile, 2.0 * p.Mg * p.Ng * p.Kg, st);
    int rc;
    if (MODE == B_WGRAD) rc = launch_tiles<MODE, 0, 0>(p, splits, tile, st);
    else if (out16) rc = epi ? launch_tiles<MODE, 1, 1>(p, splits, tile, st) : launch_tiles<MODE, 1, 0>(p, splits, tile, st);
    else rc = epi ? launch_tiles<MODE, 0, 1>(p, splits, tile, st) : launch_tiles<MODE, 0, 0>(p, splits, tile, st);
    ds6g_prof_close(rec, st);
    return rc;
}

int pick_tile(int Mg, int Ng, long splits) {
    bg_env();
    if (g_bg_force_tile >= 0) return g_bg_force_tile;
    return ((long)cdiv(Mg, 128) * cdiv(Ng, 128) * splits >= g_bg_min_blocks && Mg > 64 && Ng > 64) ? 0 : 1;
}

int run_wgrad(BgemmParams& p, float* dw, int accumulate, float* dbias, float* ws, size_t ws_bytes, hipStream_t st) {
    const long out_elems = (long)p.Mg * p.Ng;
    DS6G_CHECK_ARG(out_elems % 4 == 0 && p.Mg % 4 == 0);
    const long slab_elems = out_elems + (dbias ? p.Mg : 0);
    const int tile = (p.Mg >= 128 && p.Ng >= 128) ? 0 : 1;
    const long tiles = (long)cdiv(p.Mg, tile == 0 ? 128 : 64) * cdiv(p.Ng, tile == 0 ? 128 : 64);
    bg_env();
    const long target = g_bg_wgrad_target > 0 ? g_bg_wgrad_target : (p.N * p.Ho == 1 ? 512 : 384);
    long splits = (target + tiles - 1) / tiles;
    const long max_by_k = (p.Kg + 4 * BK - 1) / (4 * BK);   // >= 4 k-tiles per split
    if (splits > max_by_k) splits = max_by_k;
    const long max_by_ws = (long)(ws_bytes / (slab_elems * sizeof(float)));
    if (splits > max_by_ws) splits = max_by_ws;
    if (splits < 1) splits = 1;
    const int kps = cdiv(cdiv(p.Kg, splits), BK) * BK;
    splits = cdiv(p.Kg, kps);
    p.k_per_split = kps;
    p.split_stride = (size_t)slab_elems;
    p.want_colsum = dbias != nullptr;
    {
        const bool linear = p.N * p.Ho == 1;
        const bool pays = linear ? ((tiles * splits) % 8 == 0 && (long)p.Mg * p.Ng >= 512L * 512) : tiles <= 24;
        p.xcd_splits = (splits >= 2 && ((g_bg_wxcd == 1 && pays) || (g_bg_wxcd == 2 && (!linear || pays)) || g_bg_wxcd >= 3)) ? 1 : 0;
    }
    if (splits == 1 && !accumulate && !dbias) {
        p.out = dw;
        return launch_bgemm<B_WGRAD>(p, 0, 1, tile, st);
    }
    DS6G_CHECK_ARG(ws != nullptr && (size_t)splits * slab_elems * sizeof(float) <= ws_bytes);
    p.out = ws;
    const int rc = launch_bgemm<B_WGRAD>(p, 0, (int)splits, tile, st);
    if (rc) return rc;
    return ds6g_internal_splitk_reduce(ws, dw, out_elems / 4, dbias, dbias ? p.Mg / 4 : 0, (int)splits, (size_t)slab_elems,
                                       accumulate, accumulate, st);
}

bool sizes_ok(const BgemmParams& p) {
    const size_t x = (size_t)p.N * p.H * p.W * p.C * 2, y = (size_t)p.N * p.Ho * p.Wo * p.K * 2, w = (size_t)p.K * p.R * p.S * p.C * 2;
    return x < OOB_OFF && y < OOB_OFF && w < OOB_OFF;
}

}  // namespace

extern "C" {

// y = conv(x, w): x [N][H][W][C] bf16, w [K][R][S][C] bf16 (the bf16 weight shadow), y [N][Ho][Wo][K] bf16 (out16) or fp32.
// C % 64 == 0, K % 8 == 0.
int ds6g_bf16_conv2d_fwd(const void* x, const void* w, void* y, int out16, int N, int H, int W, int C, int K, int R, int S,
                         int stride, int pad, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(x && w && y && C % BK == 0 && K % 8 == 0 && N > 0);
    BgemmParams p;
    fill_conv(p, N, H, W, C, K, R, S, stride, pad);
    DS6G_CHECK_ARG(sizes_ok(p));
    p.a_src = (const __bf16*)x; p.b_src = (const __bf16*)w; p.out = y;
    p.a_bytes = (unsigned)((size_t)N * H * W * C * 2); p.b_bytes = (unsigned)((size_t)K * R * S * C * 2);
    p.Mg = N * p.Ho * p.Wo; p.Ng = K; p.Kg = R * S * C;
    return launch_bgemm<B_FWD>(p, out16, 1, pick_tile(p.Mg, p.Ng, 1), (hipStream_t)stream);
}

// y = conv(x, w) as above with a bf16 output, plus the train-mode BatchNorm statistics of y (batch mean / invstd, running
// statistics updated in place when given) from per-tile column partials written by the conv's epilogue: the statistics are
// those of the STORED bf16 tensor (what ds6g_bf16_bn_stats would compute), without a pass over it.
// ws: >= ds6g_bf16_conv_bnstats_workspace_bytes(N * Ho * Wo, K).
int ds6g_bf16_conv2d_fwd_bnstats(const void* x, const void* w, void* y, int N, int H, int W, int C, int K, int R, int S,
                                 int stride, int pad, float eps, float momentum, float* mean, float* invstd,
                                 float* running_mean, float* running_var, void* ws, size_t ws_bytes, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(x && w && y && mean && invstd && ws && C % BK == 0 && K % 8 == 0 && N > 0);
    BgemmParams p;
    fill_conv(p, N, H, W, C, K, R, S, stride, pad);
    DS6G_CHECK_ARG(sizes_ok(p));
    p.a_src = (const __bf16*)x; p.b_src = (const __bf16*)w; p.out = y;
    p.a_bytes = (unsigned)((size_t)N * H * W * C * 2); p.b_bytes = (unsigned)((size_t)K * R * S * C * 2);
    p.Mg = N * p.Ho * p.Wo; p.Ng = K; p.Kg = R * S * C;
    const int tile = pick_tile(p.Mg, p.Ng, 1);
    const int nblk = cdiv(p.Mg, tile == 0 ? 128 : 64);
    DS6G_CHECK_ARG(ws_bytes >= (size_t)nblk * 2 * K * sizeof(double));
    p.bn_partial = (double*)ws;
    const int rc = launch_bgemm<B_FWD>(p, 1, 1, tile, (hipStream_t)stream);
    if (rc) return rc;
    return ds6g_internal_bn_stats_finalize(p.bn_partial, nblk, (long)p.Mg, K, eps, momentum, mean, invstd, running_mean,
                                           running_var, (hipStream_t)stream);
}

size_t ds6g_bf16_conv_bnstats_workspace_bytes(long M, int K) { return (size_t)cdiv(M, 64) * 2 * K * sizeof(double); }

// dx (+)= conv^T(dy, w): dy bf16, w bf16, dx bf16 / fp32.  K % 64 == 0, C % 8 == 0; stride 1, or 2 with even H, W (the four
// input-pixel parity classes of a stride-2 layer run as one launch).
int ds6g_bf16_conv2d_dgrad(const void* dy, const void* w, void* dx, int out16, int N, int H, int W, int C, int K, int R,
                           int S, int stride, int pad, int accumulate, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(dy && w && dx && K % BK == 0 && C % 8 == 0 && (stride == 1 || (stride == 2 && H % 2 == 0 && W % 2 == 0)));
    BgemmParams p;
    fill_conv(p, N, H, W, C, K, R, S, stride, pad);
    DS6G_CHECK_ARG(sizes_ok(p));
    p.a_src = (const __bf16*)dy; p.b_src = (const __bf16*)w; p.out = dx; p.accumulate = accumulate;
    p.a_bytes = (unsigned)((size_t)N * p.Ho * p.Wo * K * 2); p.b_bytes = (unsigned)((size_t)K * R * S * C * 2);
    if (stride == 2) {
        p.nclass = 4;
        p.hstep = 2; p.Hs = H / 2; p.wstep = 2; p.Ws = W / 2; p.rstep = 2; p.sstep = 2;
        p.nr = (R + 1) / 2; p.ns = (S + 1) / 2;
        p.Mg = N * p.Hs * p.Ws; p.Ng = C; p.Kg = p.nr * p.ns * K;   // the largest class (tile choice); classes derive their own
        return launch_bgemm<B_DGRAD>(p, out16, 1, pick_tile(p.Mg, p.Ng, 4), (hipStream_t)stream);
    }
    p.Mg = N * H * W; p.Ng = C; p.Kg = R * S * K;
    return launch_bgemm<B_DGRAD>(p, out16, 1, pick_tile(p.Mg, p.Ng, 1), (hipStream_t)stream);
}

// dw (+)= dy^T im2col(x): x, dy bf16 -> dw fp32 [K][R][S][C] (the gradient arena).  ws: split-K slabs.
int ds6g_bf16_conv2d_wgrad(const void* x, const void* dy, float* dw, int N, int H, int W, int C, int K, int R, int S,
                           int stride, int pad, int accumulate, float* ws, size_t ws_bytes, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(x && dy && dw && C % 8 == 0 && K % 8 == 0);
    BgemmParams p;
    fill_conv(p, N, H, W, C, K, R, S, stride, pad);
    DS6G_CHECK_ARG(sizes_ok(p) && bgemm_wgrad_walk(p));
    p.a_src = (const __bf16*)dy; p.b_src = (const __bf16*)x;
    p.a_bytes = (unsigned)((size_t)N * p.Ho * p.Wo * K * 2); p.b_bytes = (unsigned)((size_t)N * H * W * C * 2);
    p.Mg = K; p.Ng = R * S * C; p.Kg = N * p.Ho * p.Wo;
    return run_wgrad(p, dw, accumulate, nullptr, ws, ws_bytes, (hipStream_t)stream);
}

// y[M][N] = residual + dropout(act(x[M][K] @ w[N][K]^T + bias)): x, w bf16; bias, residual fp32; y bf16 (out16) or fp32.
// K % 64 == 0, N % 8 == 0.
int ds6g_bf16_linear_fwd(const void* x, const void* w, const float* bias, void* y, int out16, int M, int N, int K, int relu,
                         const float* residual, float drop_p, uint64_t seed, uint64_t seed_off, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(x && w && y && K % BK == 0 && N % 8 == 0 && M > 0 && drop_p >= 0.f && drop_p < 1.f);
    DS6G_CHECK_ARG(!(out16 && residual));   // the residual stream is fp32: a fused residual add writes fp32
    BgemmParams p;
    fill_conv(p, 1, 1, M, K, N, 1, 1, 1, 0);
    DS6G_CHECK_ARG(sizes_ok(p));
    p.a_src = (const __bf16*)x; p.b_src = (const __bf16*)w; p.out = y; p.bias = bias; p.relu = relu; p.residual = residual;
    p.drop_thr = ds6g_drop_threshold(drop_p);
    p.drop_scale = 1.f / (1.f - drop_p);
    p.seed = seed; p.seed_off = seed_off; p.salt = g_ds6g_salt;
    p.a_bytes = (unsigned)((size_t)M * K * 2); p.b_bytes = (unsigned)((size_t)N * K * 2);
    p.Mg = M; p.Ng = N; p.Kg = K;
    return launch_bgemm<B_FWD>(p, out16, 1, pick_tile(M, N, 1), (hipStream_t)stream);
}

// dx[M][K] (+)= (dy[M][N] @ w[N][K]) * (mask_src > 0): dy, w bf16; mask_src [M][K] bf16 (mask16) or fp32, bf16 output only;
// dx bf16 (out16) or fp32.  N % 64 == 0, K % 8 == 0.
int ds6g_bf16_linear_dgrad(const void* dy, const void* w, void* dx, int out16, int M, int N, int K, const void* mask_src,
                           int mask16, int accumulate, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(dy && w && dx && N % BK == 0 && K % 8 == 0);
    DS6G_CHECK_ARG(!(mask_src && !out16));
    BgemmParams p;
    fill_conv(p, 1, 1, M, K, N, 1, 1, 1, 0);
    DS6G_CHECK_ARG(sizes_ok(p));
    p.a_src = (const __bf16*)dy; p.b_src = (const __bf16*)w; p.out = dx; p.mask_src = mask_src; p.mask16 = mask16;
    p.accumulate = accumulate;
    p.a_bytes = (unsigned)((size_t)M * N * 2); p.b_bytes = (unsigned)((size_t)N * K * 2);
    p.Mg = M; p.Ng = K; p.Kg = N;
    return launch_bgemm<B_DGRAD>(p, out16, 1, pick_tile(M, K, 1), (hipStream_t)stream);
}

// dw[N][K] (+)= dy[M][N]^T @ x[M][K] (fp32, the gradient arena); dbias[N] (+)= column sums of dy (nullable): x, dy bf16.
int ds6g_bf16_linear_wgrad(const void* x, const void* dy, float* dw, float* dbias, int M, int N, int K, int accumulate,
                           float* ws, size_t ws_bytes, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(x && dy && dw && K % 8 == 0 && N % 8 == 0);
    BgemmParams p;
    fill_conv(p, 1, 1, M, K, N, 1, 1, 1, 0);
    DS6G_CHECK_ARG(sizes_ok(p) && bgemm_wgrad_walk(p));
    p.a_src = (const __bf16*)dy; p.b_src = (const __bf16*)x;
    p.a_bytes = (unsigned)((size_t)M * N * 2); p.b_bytes = (unsigned)((size_t)M * K * 2);
    p.Mg = N; p.Ng = K; p.Kg = M;
    return run_wgrad(p, dw, accumulate, dbias, ws, ws_bytes, (hipStream_t)stream);
}

}  // extern "C"
