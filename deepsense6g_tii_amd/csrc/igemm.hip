// Implicit-GEMM convolution / linear kernels on the exact-fp32 matrix pipe of gfx950
// (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate).  One kernel template covers
//   FWD   : y[m][o]      = sum_{r,s,c} x[n, oh*st-p+r, ow*st-p+s, c] * w[o][r][s][c]
//   DGRAD : dx[m][c]     = sum_{r,s,o} dy[n, (h+p-r)/st, (w+p-s)/st, o] * w[o][r][s][c]
//   WGRAD : dw[o][r,s,c] = sum_{pixels} dy[pixel][o] * x[n, oh*st-p+r, ow*st-p+s, c]   (split-K)
// on NHWC activations and OHWI weights; a Linear layer is the 1x1 case over an M x 1 x 1 "image".
// Covers the reference's Conv2d / Linear call sites: model2_seq.py:495-512,528-530,546-548,
// 565-567 (ResNet trunks), :83-90,:97-99,:109,:121-126 (GPT linears), :422-425,:863-869.
//
// Tiling: 256 threads = 4 waves (2x2); block tile BM x BN, BK = 16.  Operands are staged in LDS
// row-major.  The MFMA k-order inside a tile is permuted - step kk of lane half h consumes tile
// column 8h+kk for BOTH operands - so a lane's 8 k-values are two 16-B chunks: two ds_read_b128 per
// 32-row fragment, all issued before the 8-step MFMA chain (one exposed LDS latency per k-tile
// instead of one per k-step).
// Operands go global -> LDS directly (`buffer_load_dwordx4 ... lds`, 1 KiB per wave-instruction, no VGPR
// staging, no ds_write), one k-tile ahead, two LDS buffers, one barrier per k-tile; out-of-range lanes
// (conv halo, tile / k tails) point past the buffer descriptor and the DMA writes zeros for them.
// K-contiguous sources use a [row][16] image whose 16-B chunk index is XOR-swizzled with (row>>2)&3 on
// the source address (the DMA destination is lane-linear) - fragment reads stay conflict-free
// ds_read_b128.  Row-contiguous sources (both wgrad operands, dgrad weights) use a k-major [16][rows]
// image: whole 256/512-B rows per wave-instruction and 8 conflict-free ds_read_b32 per fragment.
#include "common.h"
#include <vector>
#include <cstdlib>

extern int g_ds6g_attn_fused128;  // attention.hip: hd = 128 backward with the fused dK + dV kernel (experiment)
extern int g_ds6g_attn_percu;  // attention.hip: split-heuristic override (timing experiments)
extern int g_ds6g_attn_handover;
extern int g_wino_kb64;         // winograd.hip: 64-channel workgroups (timing experiments)

// LDS stages of the k loop (DMA runs STAGES-1 k-tiles ahead).  Measured on gfx950 (tools/bench_igemm.py): fwd / dgrad
// gain 2-11 % from a third stage (more bytes in flight per CU outweigh 8 -> 6 resident workgroups), wgrad (128x64
// tiles, 12 KB per stage) loses 1-3 %.
#ifndef IGEMM_STAGES_FD
#define IGEMM_STAGES_FD 3
#endif
#ifndef IGEMM_STAGES_W
#define IGEMM_STAGES_W 2
#endif

GCLK_STORAGE(g_igemm_clk, g_igemm_wg, ds6g_igemm_clocks_read)

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
    const uint64_t* salt;  // device-resident addend of seed_off (nullable), see ds6g_set_dropout_salt
    int k_per_split;     // multiple of 16
    size_t split_stride; // elements between split-K slabs
    int tiles_n;
    int nclass;          // DGRAD: > 1 = parity classes of a strided conv in blockIdx.y (class params derived in-kernel)
    double prof_flops;   // host-side only: flops of the launch for the profiler (0 = 2*Mg*Ng*Kg)
    int wg_rows;         // FAST WGRAD: 0 = a k-tile of 16 pixels stays inside one output row; else rows per k-tile (16 / Wo)
    int is_linear;       // host-side only: the problem is a Linear layer (split-K heuristics)
    int want_colsum;     // WGRAD: also emit column sums of the A operand (bias gradient) behind each slab
    int xcd_splits;      // WGRAD: map (tile, split) so that the column tiles of ONE split run on one XCD (they read the same
                         // dy rows and overlapping x rows: few output tiles x many splits, i.e. the 64-channel 3x3 layers)
    // DGRAD of a strided conv is run per input-pixel parity class: pixels h = h0 + hstep*hh (hh < Hs), taps
    // r = r0 + rstep*ri (ri < nr) - only the taps that hit a real output pixel, no structural zeros
    int h0, hstep, Hs, w0, wstep, Ws, r0, rstep, nr, s0, sstep, ns;
    unsigned a_bytes, b_bytes;  // sizes of the a_src / b_src tensors (buffer descriptors: OOB lanes read zeros)
    int dbg;  // ablation flags (timing experiments only): 1 skip in-loop global loads, 2 skip LDS stores, 4 skip barrier
};


// EPI 0: store (or accumulate) only - branch-free bounds handling through a buffer descriptor;
// EPI 1: full epilogue (bias, ReLU, ReLU-mask, dropout, residual, accumulate, strided-dgrad row remap).
// BF 1: operands are rounded to bf16 (RNE) on the way from the LDS fragment to the matrix core and one
// v_mfma_f32_32x32x16_bf16 replaces the eight fp32 MFMAs of a k-tile (fp32 accumulate, fp32 storage everywhere):
// the "bf16 forward/backward" throughput configuration of BASELINE.json; BF 0 is the exact-fp32 parity path.
// BF 2: split-bf16 ("bf16x3"): every fp32 operand is split into hi = bf16(a), lo = bf16(a - hi) and the product is
// hi*hi + hi*lo + lo*hi on the bf16 matrix cores (fp32 accumulate): relative product error <= ~2^-16 - between fp32
// and TF32 - for 3/16 of the exact path's matrix-core cycles.
// BF 3: three-way split (hi, mid, lo = 24 significand bits), the six products above 2^-24: fp32-grade results
// for 6/16 of the exact path's matrix-core cycles.
// FAST 1: the k walk is wave-uniform - a k-tile never straddles a filter tap (FWD: C % BK == 0, DGRAD: K % BK == 0) or
// an image row group (WGRAD: see wgrad_fast_ok) - so the per-lane byte offsets are constants of the current tap / of
// the lane, and the per-k-tile advance lives in SGPRs and rides in the buffer instruction's scalar offset: the loop
// spends ~2 (FWD/DGRAD) to ~8 (WGRAD) VALU instructions per DMA piece instead of ~25.  FAST 0 is the general walk
// (any channel count, any image size), kept for the stem (C = 4) and odd shapes.
template <int MODE, int BM, int BN, int EPI, int BK, int BF, int FAST>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmParams pin) {
    IgemmParams p = pin;
    if (EPI && p.drop_thr && p.salt) p.seed_off += *p.salt;
    GCLK_DECL(g_igemm_wg);
    if (MODE == MODE_DGRAD && pin.nclass > 1) {
        // strided dgrad: blockIdx.y = input-pixel parity class (ph, pw); class (ph, pw) only sees the taps
        // r = (ph + pad) mod 2 (+2, ...), s likewise - all four classes of a layer run as ONE launch
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
    constexpr int CH = BK / 4;              // 16-B chunks per K-contiguous row
    constexpr int RPB = 16 / CH;            // rows per 256-B LDS bank row
    constexpr int KH = BK / 2;              // k values per lane half
    constexpr int TM = BM / 64;  // 32x32 MFMA tiles per wave along M (wave grid 2x2)
    constexpr int TN = BN / 64;
    constexpr int A_LD = BM * CH / 256;  // 16-B DMA pieces per thread per k-tile
    constexpr int B_LD = BN * CH / 256;
    constexpr bool A_KMAJOR = (MODE == MODE_WGRAD);  // A source contiguous along rows (m): image [k][BM]
    constexpr bool B_KMAJOR = (MODE != MODE_FWD);    // B source contiguous along rows (n): image [k][BN]
    // K-contiguous sources: image [row][BK] (no padding - the DMA destination is lane-linear) with the 16-B chunk
    // index XOR-swizzled by (row / RPB) & (CH-1) on the SOURCE side; readers apply the same XOR.
    // four separate objects (not [2][..] arrays): the compiler must be able to prove that the fragment reads of
    // one buffer do not alias the DMA in flight into the other, or it drains vmcnt before every read
    __shared__ __attribute__((aligned(16))) float As0[BM * BK];
    __shared__ __attribute__((aligned(16))) float As1[BM * BK];
    __shared__ __attribute__((aligned(16))) float Bs0[BN * BK];
    __shared__ __attribute__((aligned(16))) float Bs1[BN * BK];
    constexpr int NS = (MODE == MODE_WGRAD) ? IGEMM_STAGES_W : (BK == 32 ? 2 : IGEMM_STAGES_FD);
    static_assert(NS >= 2 && NS <= 4, "2..4 LDS stages");
    __shared__ __attribute__((aligned(16))) float As2[NS >= 3 ? BM * BK : 4];
    __shared__ __attribute__((aligned(16))) float Bs2[NS >= 3 ? BN * BK : 4];
    __shared__ __attribute__((aligned(16))) float As3[NS >= 4 ? BM * BK : 4];
    __shared__ __attribute__((aligned(16))) float Bs3[NS >= 4 ? BN * BK : 4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order: hardware deals consecutive workgroups round-robin over the 8 XCDs (each
    // with a private L2); remap so that every XCD walks a contiguous run of tiles (tile_n fastest),
    // i.e. tiles sharing A rows / neighbouring image rows hit the same L2.  Bijective for any grid.
    int wg, split = blockIdx.z;
    if (MODE == MODE_WGRAD && p.xcd_splits) {
        // the hardware deals workgroups to XCDs in dispatch order (x fastest, then z): give every XCD a contiguous run of
        // (split, tile) pairs, tile fastest, so the tiles of a split - same dy rows, overlapping x rows - share one L2
        const int nwg = gridDim.x * gridDim.z, orig = blockIdx.x + gridDim.x * blockIdx.z;
        const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
        split = lin / (int)gridDim.x;
        wg = lin - split * (int)gridDim.x;
    } else {
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int tile_m = wg / p.tiles_n;
    const int tile_n = wg - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbegin = (MODE == MODE_WGRAD) ? split * p.k_per_split : 0;
    const int kend = (MODE == MODE_WGRAD) ? min(p.Kg, kbegin + p.k_per_split) : p.Kg;
    const int nk = (kend - kbegin + BK - 1) / BK;

    const i32x4 a_rsrc = make_srd(p.a_src, p.a_bytes);
    const i32x4 b_rsrc = make_srd(p.b_src, p.b_bytes);

    // ---- per-thread DMA state.  Everything that needs a division happens ONCE here; the k-loop only advances
    // (channel, tap) / pixel counters by BK with compare-and-wrap.  Offsets are in floats (tensors < 2 GB).
    unsigned a_base[A_LD];       // FWD/DGRAD: image base of this row; WGRAD: column base
    int a_y[A_LD], a_x[A_LD];    // FWD: ih0, iw0 ; DGRAD: h+pad, w+pad
    int a_c[A_LD], a_r[A_LD], a_s[A_LD], a_k[A_LD];
    bool a_ok[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int idx = tid + i * 256;
        if (MODE == MODE_FWD || MODE == MODE_DGRAD) {
            const int row = idx / CH;
            const int m = m0 + row;
            a_ok[i] = m < p.Mg;
            const int mm = a_ok[i] ? m : 0;
            const int cin = (MODE == MODE_FWD) ? p.C : p.K;
            const int kg = ((idx % CH) ^ ((row / RPB) & (CH - 1))) * 4;  // swizzled source chunk
            a_k[i] = kg;
            const int tap = kg / cin;
            a_c[i] = kg - tap * cin;
            const int nsx = (MODE == MODE_DGRAD) ? p.ns : p.S;
            a_r[i] = tap / nsx;
            a_s[i] = tap - a_r[i] * nsx;
            if (MODE == MODE_FWD) {
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
        } else {
            const int mc = idx % (BM / 4);
            a_ok[i] = (m0 + mc * 4) < p.Mg;
            a_k[i] = kbegin + idx / (BM / 4);  // pixel
            a_base[i] = (unsigned)(m0 + mc * 4);
            a_y[i] = a_x[i] = a_c[i] = a_r[i] = a_s[i] = 0;
        }
    }
    unsigned b_base[B_LD];
    int b_r[B_LD], b_s[B_LD];      // WGRAD: fixed tap of this thread's column chunk
    int b_k[B_LD];                 // running global k (FWD/DGRAD) or pixel (WGRAD)
    int b_o[B_LD], b_t[B_LD];      // DGRAD: running (out channel, tap); WGRAD: running (ow, oh)
    int b_n[B_LD];                 // WGRAD: running image index
    bool b_ok[B_LD];
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
        const int idx = tid + i * 256;
        b_r[i] = b_s[i] = b_o[i] = b_t[i] = b_n[i] = 0;
        if (MODE == MODE_FWD) {
            const int row = idx / CH;
            b_ok[i] = (n0 + row) < p.Ng;
            b_k[i] = ((idx % CH) ^ ((row / RPB) & (CH - 1))) * 4;
            b_base[i] = (unsigned)(n0 + row) * (unsigned)p.Kg;
        } else if (MODE == MODE_DGRAD) {
            b_ok[i] = (n0 + (idx % (BN / 4)) * 4) < p.Ng;
            const int kg = idx / (BN / 4);
            b_k[i] = kg;
            const int tap = kg / p.K;
            b_o[i] = kg - tap * p.K;
            b_t[i] = tap / p.ns;          // running tap row index ri
            b_n[i] = tap - b_t[i] * p.ns; // running tap column index si
            b_base[i] = (unsigned)(n0 + (idx % (BN / 4)) * 4);
        } else {
            const int ncol = n0 + (idx % (BN / 4)) * 4;
            b_ok[i] = ncol < p.Ng;
            const int nn = b_ok[i] ? ncol : 0;
            const int tap = nn / p.C;
            b_r[i] = tap / p.S;
            b_s[i] = tap - b_r[i] * p.S;
            b_base[i] = (unsigned)(nn - tap * p.C);
            const int pix = kbegin + idx / (BN / 4);
            b_k[i] = pix;
            const int pp = pix < p.Kg ? pix : 0;
            b_o[i] = pp % p.Wo;           // ow
            const int t = pp / p.Wo;
            b_t[i] = t % p.Ho;            // oh
            b_n[i] = t / p.Ho;
        }
    }

    // ---- FAST walk state: uniform counters (SGPRs) + per-lane constant byte offsets ------------------------------
    [[maybe_unused]] int u_c0 = 0, u_r = 0, u_s = 0;          // FWD/DGRAD: channel offset inside the tap, tap
    [[maybe_unused]] unsigned u_kb = 0;                       // FWD: byte offset of the k-tile in a weight row
    [[maybe_unused]] int u_kpos = kbegin, u_n = 0, u_oh = 0, u_ow = 0;  // WGRAD: first pixel of the k-tile
    [[maybe_unused]] unsigned a_vo[A_LD], b_vo[B_LD];
    [[maybe_unused]] int a_kr[A_LD], b_kr[B_LD], b_ihl[B_LD], b_iwl[B_LD];
    [[maybe_unused]] i32x4 b_rsrc_f = b_rsrc;
    // FWD/DGRAD: byte offsets of this lane's A pieces for tap (u_r, u_s); OOB_OFF where the tap misses the image
    auto retap = [&]() {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            if (MODE == MODE_FWD) {
                const int ih = a_y[i] + u_r, iw = a_x[i] + u_s;
                const bool ok = a_ok[i] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                a_vo[i] = ok ? (a_base[i] + (unsigned)((ih * p.W + iw) * p.C + a_c[i])) * 4u : OOB_OFF;
            } else if (MODE == MODE_DGRAD) {
                const int th = a_y[i] - (p.r0 + u_r * p.rstep), tw = a_x[i] - (p.s0 + u_s * p.sstep);
                int oh = th, ow = tw;
                bool okk = th >= 0 && tw >= 0;
                if (p.stride == 2) {
                    oh = th >> 1;
                    ow = tw >> 1;
                    okk = okk && !((th | tw) & 1);
                } else if (p.stride != 1) {
                    oh = th / p.stride;
                    ow = tw / p.stride;
                    okk = okk && (oh * p.stride == th) && (ow * p.stride == tw);
                }
                const bool ok = a_ok[i] && okk && oh < p.Ho && ow < p.Wo;
                a_vo[i] = ok ? (a_base[i] + (unsigned)((oh * p.Wo + ow) * p.K + a_c[i])) * 4u : OOB_OFF;
            }
        }
    };
    if (FAST) {
        if (MODE == MODE_FWD) {
            retap();
#pragma unroll
            for (int i = 0; i < B_LD; ++i) b_vo[i] = b_ok[i] ? (b_base[i] + (unsigned)b_k[i]) * 4u : OOB_OFF;
        } else if (MODE == MODE_DGRAD) {
            retap();
#pragma unroll
            for (int i = 0; i < B_LD; ++i)  // k-row (out channel within the tile) of this lane: b_o
                b_vo[i] = b_ok[i] ? (b_base[i] + (unsigned)(b_o[i] * (p.R * p.S) * p.C)) * 4u : OOB_OFF;
        } else {
            // pixel kbegin -> (image, output row, output column); kbegin is a multiple of 32
            if (p.wg_rows == 0) {
                u_ow = kbegin % p.Wo;
                const int t = kbegin / p.Wo;
                u_oh = t % p.Ho;
                u_n = t / p.Ho;
            } else {
                const int t = kbegin / p.Wo;
                u_oh = t % p.Ho;
                u_n = t / p.Ho;
            }
            // negative tap offsets (halo) are folded into the descriptor base so that lane constants stay >= 0
            const int shift = (p.pad * p.W + p.pad) * p.C;
            b_rsrc_f = make_srd(p.b_src - shift, p.b_bytes + (unsigned)shift * 4u);
#pragma unroll
            for (int i = 0; i < A_LD; ++i) {
                a_kr[i] = (tid + i * 256) / (BM / 4);
                a_vo[i] = (a_base[i] + (unsigned)(a_kr[i] * p.K)) * 4u;
            }
#pragma unroll
            for (int i = 0; i < B_LD; ++i) {
                b_kr[i] = (tid + i * 256) / (BN / 4);
                const int dl_oh = p.wg_rows == 0 ? 0 : b_kr[i] / p.Wo;
                const int dl_ow = p.wg_rows == 0 ? b_kr[i] : b_kr[i] % p.Wo;
                b_ihl[i] = dl_oh * p.stride - p.pad + b_r[i];
                b_iwl[i] = dl_ow * p.stride - p.pad + b_s[i];
                b_vo[i] = (unsigned)((int)b_base[i] + (b_ihl[i] * p.W + b_iwl[i]) * p.C + shift) * 4u;
            }
        }
    }

    // issues the DMA of the NEXT k-tile into LDS buffer `buf` (state is advanced by BK afterwards)
    auto issue_tiles = [&](float* Ad, float* Bd) {
        if (FAST) {
            if (MODE == MODE_FWD || MODE == MODE_DGRAD) {
                const unsigned sa = (unsigned)u_c0 * 4u;
                unsigned sb;
                if (MODE == MODE_FWD) {
                    sb = u_kb;
                } else {
                    const int tapw = (p.r0 + u_r * p.rstep) * p.S + p.s0 + u_s * p.sstep;
                    sb = (unsigned)((u_c0 * (p.R * p.S) + tapw) * p.C) * 4u;
                }
#pragma unroll
                for (int i = 0; i < A_LD; ++i)
                    dma16(a_rsrc, lds_addr(Ad) + (unsigned)(i * 256 + wave * 64) * 16u, (p.dbg & 32) ? OOB_OFF : a_vo[i], sa);
#pragma unroll
                for (int i = 0; i < B_LD; ++i)
                    dma16(b_rsrc, lds_addr(Bd) + (unsigned)(i * 256 + wave * 64) * 16u, (p.dbg & 32) ? OOB_OFF : b_vo[i], sb);
                u_kb += BK * 4;
                u_c0 += BK;
                if (u_c0 == ((MODE == MODE_FWD) ? p.C : p.K)) {
                    u_c0 = 0;
                    if (++u_s == ((MODE == MODE_FWD) ? p.S : p.ns)) { u_s = 0; ++u_r; }
                    retap();
                }
            } else {
                const int rows_left = kend - u_kpos;
                const unsigned sa = (unsigned)(u_kpos * p.K) * 4u;
                const int ihu = u_oh * p.stride, iwu = u_ow * p.stride;
                const unsigned sb = (unsigned)(((u_n * p.H + ihu) * p.W + iwu) * p.C) * 4u;
#pragma unroll
                for (int i = 0; i < A_LD; ++i) {
                    const bool ok = a_ok[i] && a_kr[i] < rows_left && !(p.dbg & 32);
                    dma16(a_rsrc, lds_addr(Ad) + (unsigned)(i * 256 + wave * 64) * 16u, ok ? a_vo[i] : OOB_OFF, sa);
                }
#pragma unroll
                for (int i = 0; i < B_LD; ++i) {
                    const bool ok = b_ok[i] && b_kr[i] < rows_left && (unsigned)(ihu + b_ihl[i]) < (unsigned)p.H &&
                                    (unsigned)(iwu + b_iwl[i]) < (unsigned)p.W && !(p.dbg & 32);
                    dma16(b_rsrc_f, lds_addr(Bd) + (unsigned)(i * 256 + wave * 64) * 16u, ok ? b_vo[i] : OOB_OFF, sb);
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
            return;
        }
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            unsigned off;
            bool ok = a_ok[i] && a_k[i] < kend;
            if (MODE == MODE_FWD) {
                const int ih = a_y[i] + a_r[i], iw = a_x[i] + a_s[i];
                ok = ok && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
                off = a_base[i] + (unsigned)((ih * p.W + iw) * p.C + a_c[i]);
            } else if (MODE == MODE_DGRAD) {
                const int th = a_y[i] - (p.r0 + a_r[i] * p.rstep), tw = a_x[i] - (p.s0 + a_s[i] * p.sstep);
                int oh = th, ow = tw;
                bool okk = th >= 0 && tw >= 0;
                if (p.stride == 2) {
                    oh = th >> 1;
                    ow = tw >> 1;
                    okk = okk && !((th | tw) & 1);
                } else if (p.stride != 1) {
                    oh = th / p.stride;
                    ow = tw / p.stride;
                    okk = okk && (oh * p.stride == th) && (ow * p.stride == tw);
                }
                ok = ok && okk && oh < p.Ho && ow < p.Wo;
                off = a_base[i] + (unsigned)((oh * p.Wo + ow) * p.K + a_c[i]);
            } else {
                off = a_base[i] + (unsigned)a_k[i] * (unsigned)p.K;
            }
            if (p.dbg & 32) ok = false;
            dma16(a_rsrc, lds_addr(Ad) + (unsigned)(i * 256 + wave * 64) * 16u, ok ? off * 4u : OOB_OFF, 0u);
            a_k[i] += BK;
            if (MODE != MODE_WGRAD) {
                const int cin = (MODE == MODE_FWD) ? p.C : p.K;
                a_c[i] += BK;
                const int nsx = (MODE == MODE_DGRAD) ? p.ns : p.S;
                while (a_c[i] >= cin) {
                    a_c[i] -= cin;
                    if (++a_s[i] == nsx) { a_s[i] = 0; ++a_r[i]; }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            unsigned off;
            bool ok = b_ok[i] && b_k[i] < kend;
            if (MODE == MODE_FWD) {
                off = b_base[i] + (unsigned)b_k[i];
            } else if (MODE == MODE_DGRAD) {
                const int tap = (p.r0 + b_t[i] * p.rstep) * p.S + p.s0 + b_n[i] * p.sstep;
                off = b_base[i] + (unsigned)((b_o[i] * (p.R * p.S) + tap) * p.C);
                b_o[i] += BK;
                while (b_o[i] >= p.K) {
                    b_o[i] -= p.K;
                    if (++b_n[i] == p.ns) { b_n[i] = 0; ++b_t[i]; }
                }
            } else {
                const int ih = b_t[i] * p.stride - p.pad + b_r[i];
                const int iw = b_o[i] * p.stride - p.pad + b_s[i];
                ok = ok && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
                off = b_base[i] + (unsigned)(((b_n[i] * p.H + ih) * p.W + iw) * p.C);
                b_o[i] += BK;
                while (b_o[i] >= p.Wo) {
                    b_o[i] -= p.Wo;
                    if (++b_t[i] == p.Ho) { b_t[i] = 0; ++b_n[i]; }
                }
            }
            if (p.dbg & 32) ok = false;
            dma16(b_rsrc, lds_addr(Bd) + (unsigned)(i * 256 + wave * 64) * 16u, ok ? off * 4u : OOB_OFF, 0u);
            b_k[i] += BK;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_row = wm * (BM / 2) + (lane & 31);
    const int b_row = wn * (BN / 2) + (lane & 31);
    const int khalf = lane >> 5;
    const int swz = ((lane & 31) / RPB) & (CH - 1);  // chunk XOR of every fragment row of this lane

    // one k-tile: start the DMA of the next tile into (An, Bn), multiply the tile resident in (Ac, Bc)
    float csum = 0.f;  // WGRAD bias gradient: column sum of the dy tile, owned by thread tid < BM of tile_n == 0 blocks
    const bool do_csum = (MODE == MODE_WGRAD) && p.want_colsum && tile_n == 0 && tid < BM;
    // NS >= 3 stages, DMA NS-1 k-tiles ahead: wait for THIS tile only (younger ones may still be in flight), barrier
    // (everyone has this tile and is done with the previous one), refill the previous tile's stage, multiply.
    // rem = k-tiles after this one.  NS == 2: issue the next tile, multiply, wait + barrier at the end.
    auto k_step = [&](const float* Ac, const float* Bc, float* An, float* Bn, int rem) {
        if (NS >= 3) {
            if (NS >= 4 && rem >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (A_LD + B_LD)) : "memory");
            else if (rem >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_LD + B_LD) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            GCLK(g_igemm_clk, 4);   // wait for this tile's DMA
            if (!(p.dbg & 4)) __syncthreads();
            GCLK(g_igemm_clk, 5);   // barrier
        }
        if (rem >= NS - 1 && !(p.dbg & 1)) issue_tiles(An, Bn);
        GCLK(g_igemm_clk, 2);       // DMA issue
        if (MODE == MODE_WGRAD && do_csum) {
#pragma unroll
            for (int k = 0; k < BK; ++k) csum += Ac[k * BM + tid];
        }
        // fragments: lane (row, khalf) holds tile columns KH*khalf .. KH*khalf+KH-1 of its row
        float af[TM][KH], bf[TN][KH];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (A_KMAJOR) {
#pragma unroll
                for (int kk = 0; kk < KH; ++kk) af[i][kk] = Ac[(khalf * KH + kk) * BM + a_row + i * 32];
            } else {
                const float* rp = &Ac[(a_row + i * 32) * BK];
#pragma unroll
                for (int c = 0; c < CH / 2; ++c) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(rp + (((khalf * (CH / 2) + c) ^ swz) << 2));
#pragma unroll
                    for (int e = 0; e < 4; ++e) af[i][4 * c + e] = v[e];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (B_KMAJOR) {
#pragma unroll
                for (int kk = 0; kk < KH; ++kk) bf[j][kk] = Bc[(khalf * KH + kk) * BN + b_row + j * 32];
            } else {
                const float* rp = &Bc[(b_row + j * 32) * BK];
#pragma unroll
                for (int c = 0; c < CH / 2; ++c) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(rp + (((khalf * (CH / 2) + c) ^ swz) << 2));
#pragma unroll
                    for (int e = 0; e < 4; ++e) bf[j][4 * c + e] = v[e];
                }
            }
        }
        // keep every fragment read ahead of the MFMA chain (the scheduler otherwise sinks each read next to
        // its use and exposes the LDS latency once per k-step)
        __builtin_amdgcn_sched_barrier(0);
        if (BF) {
            // one 32x32x16 bf16 MFMA per 16 tile columns: in MFMA q lane half h supplies its fragment values 8q..8q+7
            // (tile columns KH*h + 8q ..), the same column set for both operands
            static_assert(!BF || BK % 16 == 0, "bf16 k-tiles are multiples of 16");
#pragma unroll
            for (int q = 0; q < BK / 16; ++q) {
                if (BF == 3) {
                    bf16x8 ah[TM], am[TM], al3[TM], bh[TN], bm[TN], bl3[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) split3_bf16x8(&af[i][8 * q], ah[i], am[i], al3[i]);
#pragma unroll
                    for (int j = 0; j < TN; ++j) split3_bf16x8(&bf[j][8 * q], bh[j], bm[j], bl3[j]);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = mfma_x6(ah[i], am[i], al3[i], bh[j], bm[j], bl3[j], acc[i][j]);
                    continue;
                }
                bf16x8 a8[TM], b8[TN], al[TM], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if (BF == 2) split_bf16x8(&af[i][8 * q], a8[i], al[i]);
                    else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) a8[i][e] = (__bf16)af[i][8 * q + e];
                    }
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (BF == 2) split_bf16x8(&bf[j][8 * q], b8[j], bl[j]);
                    else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) b8[j][e] = (__bf16)bf[j][8 * q + e];
                    }
                }
                if (BF == 2) {  // the two cross terms first (small), the leading term last
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], b8[j], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[i], bl[j], acc[i][j], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[i], b8[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][kk], bf[j][kk], acc[i][j], 0, 0, 0);
            }
        }
        // the DMA of the next tile must have landed before any wave reads it, and every wave must be done reading
        // the current buffers before the DMA after next overwrites them.  sched_barrier pins the wait BEHIND the
        // MFMA chain (an asm wait does not order register-only instructions) so the DMA flies under the MFMAs.
        __builtin_amdgcn_sched_barrier(0);
        GCLK(g_igemm_clk, 3);       // fragment reads + MFMA chain
        GCLK_COUNT(g_igemm_clk, 15);
        if (NS == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            GCLK(g_igemm_clk, 4);
            if (!(p.dbg & 4)) __syncthreads();
            GCLK(g_igemm_clk, 5);
        }
    };

    float* const Ast[4] = {As0, As1, As2, As3};
    float* const Bst[4] = {Bs0, Bs1, Bs2, Bs3};
    // the epilogue's bias values, fetched BEFORE the k loop: loaded where they are used, the (two) dependent global loads
    // cost a full memory latency under load at the very end of every workgroup (measured with -DDS6G_GEMM_CLOCKS: epilogue
    // of the 128 x 64 forward 19 700 cycles of a 274 000-cycle workgroup life)
    [[maybe_unused]] float bias_pre[TN];
    if (EPI) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
            bias_pre[j] = (p.bias && col < p.Ng) ? p.bias[col] : 0.f;
        }
    }
    GCLK(g_igemm_clk, 0);           // set-up
#pragma unroll
    for (int u = 0; u < NS - 1; ++u)
        if (u < nk && !(p.dbg & 16)) issue_tiles(Ast[u], Bst[u]);
    if (NS == 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    GCLK(g_igemm_clk, 1);           // first tile(s) issued (NS == 2: and landed)
    for (int kt0 = 0; kt0 < nk; kt0 += NS) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int kt = kt0 + u;
            if (kt < nk) k_step(Ast[u], Bst[u], Ast[(u + NS - 1) % NS], Bst[(u + NS - 1) % NS], nk - 1 - kt);
        }
    }

    // ---- epilogue: D[row][col], col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ---------
    float* outp = p.out + ((MODE == MODE_WGRAD) ? (size_t)split * p.split_stride : (size_t)0);
    if (MODE == MODE_WGRAD && do_csum && m0 + tid < p.Mg) outp[(size_t)p.Mg * p.Ng + m0 + tid] = csum;
    if (EPI == 0) {
        // rows >= Mg land beyond the descriptor (Mg*Ng*4 bytes) and are dropped by the hardware range check;
        // columns >= Ng are sent there explicitly.  No divergent control flow, one add per store.
        const auto o_rsrc = __builtin_amdgcn_make_buffer_rsrc(outp, 0, (unsigned)p.Mg * (unsigned)p.Ng * 4u, 0x00020000);
        const unsigned rowbytes = (unsigned)p.Ng * 4u;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
                const int row0 = m0 + wm * (BM / 2) + i * 32 + 4 * khalf;
                const unsigned base = (col < p.Ng && !((p.dbg & 8) && (i | j)))
                                          ? (unsigned)row0 * rowbytes + (unsigned)col * 4u
                                          : OOB_OFF;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned off = base + (unsigned)((r & 3) + 8 * (r >> 2)) * rowbytes;
                    float v = acc[i][j][r];
                    // the b32 builtins move raw bits (unsigned): bit-cast, do not convert
                    if (MODE != MODE_WGRAD && p.accumulate)
                        v += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(o_rsrc, off, 0, 0));
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), o_rsrc, off, 0, 0);
                }
            }
        }
        GCLK(g_igemm_clk, 6);   // epilogue (stores issued)
        GCLK_END(g_igemm_clk, g_igemm_wg);
        return;
    }
    if (!(MODE == MODE_DGRAD && p.hstep != 1)) {
        // full epilogue with the same branch-free addressing: mask, residual and output all have the output's
        // [Mg][Ng] shape, so ONE 32-bit byte offset serves every per-element tensor and out-of-tile rows / columns fall
        // off the buffer descriptors (loads return 0, stores are dropped)
        const unsigned nbytes = (unsigned)p.Mg * (unsigned)p.Ng * 4u;
        const auto o_rsrc = __builtin_amdgcn_make_buffer_rsrc(outp, 0, nbytes, 0x00020000);
        const auto m_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.mask_src ? p.mask_src : outp), 0, nbytes, 0x00020000);
        const auto r_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.residual ? p.residual : outp), 0, nbytes, 0x00020000);
        const unsigned rowbytes = (unsigned)p.Ng * 4u;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
                const int row0 = m0 + wm * (BM / 2) + i * 32 + 4 * khalf;
                const bool cok = col < p.Ng;
                const float bias = bias_pre[j];
                const unsigned base = cok ? (unsigned)row0 * rowbytes + (unsigned)col * 4u : OOB_OFF;
                const unsigned e0 = (unsigned)row0 * (unsigned)p.Ng + (unsigned)col;  // element index (dropout counter)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned dr = (unsigned)((r & 3) + 8 * (r >> 2));
                    const unsigned off = base + dr * rowbytes;
                    float v = acc[i][j][r] + bias;
                    if (p.relu == 1) v = fmaxf(v, 0.f);
                    if (p.mask_src) v = (__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(m_rsrc, off, 0, 0)) > 0.f) ? v : 0.f;
                    if (p.drop_thr)
                        v = ds6g_keep(p.seed, p.seed_off + (uint64_t)(e0 + dr * (unsigned)p.Ng), p.drop_thr) ? v * p.drop_scale : 0.f;
                    if (p.residual) v += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_rsrc, off, 0, 0));
                    if (p.accumulate) v += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(o_rsrc, off, 0, 0));
                    if (p.relu == 2) v = fmaxf(v, 0.f);  // activation after the residual add (BasicBlock tail)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), o_rsrc, off, 0, 0);
                }
            }
        }
        GCLK(g_igemm_clk, 6);   // epilogue (stores issued)
        GCLK_END(g_igemm_clk, g_igemm_wg);
        return;
    }
    // strided dgrad: rows of the parity-class sub-grid scatter to full-resolution pixels
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
            if (col >= p.Ng) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (row >= p.Mg) continue;
                const int ww = row % p.Ws;
                const int t = row / p.Ws;
                const int hh = t % p.Hs;
                const size_t orow = ((size_t)(t / p.Hs) * p.H + p.h0 + hh * p.hstep) * p.W + p.w0 + ww * p.wstep;
                const size_t o = orow * p.Ng + col;
                float v = acc[i][j][r];
                if (p.accumulate) v += outp[o];
                outp[o] = v;
            }
        }
    }
    GCLK(g_igemm_clk, 6);
    GCLK_END(g_igemm_clk, g_igemm_wg);
}

// out[i] = (accumulate ? out[i] : 0) + sum_s part[s*stride + i].  256 threads = 16 float4 columns x 16 split-lanes
// (small outputs with many splits are latency-bound: spread the splits over lanes, keep many loads in flight).
// The slab holds n4 float4 of weight gradient followed (optionally) by m4 float4 of bias gradient -> out_b.
constexpr int RED_COLS = 16, RED_LANES = 16;
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                            long n4, float* __restrict__ out_b, long m4, int splits,
                                                            size_t stride, int accumulate, int accumulate_b) {
    __shared__ f32x4 red[RED_LANES][RED_COLS];
    const int col = threadIdx.x & (RED_COLS - 1), sl = threadIdx.x / RED_COLS;
    const long i = (long)blockIdx.x * RED_COLS + col;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < n4 + m4)
        for (int k = sl; k < splits; k += RED_LANES) s += *reinterpret_cast<const f32x4*>(part + (size_t)k * stride + i * 4);
    red[sl][col] = s;
    __syncthreads();
    if (sl != 0 || i >= n4 + m4) return;
#pragma unroll
    for (int l = 1; l < RED_LANES; ++l) s += red[l][col];
    const bool is_b = i >= n4;
    f32x4* o = is_b ? reinterpret_cast<f32x4*>(out_b + (i - n4) * 4) : reinterpret_cast<f32x4*>(out + i * 4);
    if (is_b ? accumulate_b : accumulate) s += *o;
    *o = s;
}

// last launched variant, for the bench's live per-kernel timing (see ds6g_last_igemm_variant in the header)
int g_last_variant = -1;

// bench instrumentation: HIP events around every implicit-GEMM kernel launch, on the launch stream (ds6g_profile_*)
struct ProfRec { int variant; double flops; hipEvent_t e0, e1; };
std::vector<ProfRec>* g_prof = nullptr;
size_t g_prof_cap = 0;
int g_dbg = 0;

// tile choice: 0 = 128x128, 1 = 128x64, 2 = 64x64 - the largest tile that still yields >= 384 workgroups
int g_min_blocks = 2560;  // measured: many small (64x64, 8 waves/SIMD) workgroups beat larger tiles up to here
int g_wgrad_tile = 1;
int g_bf16_bk32 = 1;
int g_f32_bk32 = 2;  // 0 never, 1 wherever the walk allows it (experiment), 2 small-spatial convs only
int pick_tile(int Mg, int Ng, long splits) {
    const long t128 = (long)cdiv(Mg, 128) * cdiv(Ng, 128) * splits;
    const long t12864 = (long)cdiv(Mg, 128) * cdiv(Ng, 64) * splits;
    if (Ng > 64 && Mg > 64 && t128 >= g_min_blocks) return 0;
    if (Mg > 64 && t12864 >= g_min_blocks) return 1;
    return 2;
}

template <int MODE, int EPI, int BKV, int BF, int FAST>
void launch_tile(IgemmParams& p, int splits, int tile, hipStream_t st) {
    dim3 block(256);
    const int ny = p.nclass > 1 ? p.nclass : 1;
    if (tile == 0) {
        p.tiles_n = cdiv(p.Ng, 128);
        dim3 grid(cdiv(p.Mg, 128) * p.tiles_n, ny, splits);
        hipLaunchKernelGGL((igemm_kernel<MODE, 128, 128, EPI, BKV, BF, FAST>), grid, block, 0, st, p);
    } else if (tile == 1) {
        p.tiles_n = cdiv(p.Ng, 64);
        dim3 grid(cdiv(p.Mg, 128) * p.tiles_n, ny, splits);
        hipLaunchKernelGGL((igemm_kernel<MODE, 128, 64, EPI, BKV, BF, FAST>), grid, block, 0, st, p);
    } else {
        p.tiles_n = cdiv(p.Ng, 64);
        dim3 grid(cdiv(p.Mg, 64) * p.tiles_n, ny, splits);
        hipLaunchKernelGGL((igemm_kernel<MODE, 64, 64, EPI, BKV, BF, FAST>), grid, block, 0, st, p);
    }
}

// FAST walk eligibility (see igemm_kernel); also fills wg_rows for WGRAD
template <int MODE>
bool fast_walk_ok(IgemmParams& p, int bk = 16) {
    if (MODE == MODE_FWD) return p.C % bk == 0;
    if (MODE == MODE_DGRAD) return p.K % bk == 0;
    // WGRAD: the bk pixels of a k-tile share one (image, output row) - or cover whole rows of one image
    if (p.Wo % bk == 0 || p.N * p.Ho == 1) { p.wg_rows = 0; return true; }
    if (bk % p.Wo == 0 && p.Ho % (bk / p.Wo) == 0) { p.wg_rows = bk / p.Wo; return true; }
    return false;
}

template <int MODE>
int launch_igemm(IgemmParams& p, int splits, int tile, hipStream_t st) {
    p.dbg = g_dbg;
    const bool full = p.bias || p.relu || p.mask_src || p.drop_thr || p.residual || (MODE == MODE_DGRAD && p.hstep != 1);
    if ((size_t)p.Mg * p.Ng * sizeof(float) >= OOB_OFF) return DS6G_ERR_ARG;
    // BK = 32 was measured (tools/bench_igemm.py): within +-5 % on fwd/dgrad, 10-30 % slower on wgrad -> BK = 16
    const bool fast = fast_walk_ok<MODE>(p) && !(g_dbg & 0x80);
    const bool epi = MODE != MODE_WGRAD && full;
    // 32-column k-tiles: bf16 mode - one MFMA per 16 columns makes the k-tile bookkeeping (DMA issue, barrier) the
    // bottleneck, so fwd / dgrad use two MFMAs per barrier wherever the uniform walk allows it (wgrad was measured with
    // 32-pixel k-tiles too: no gain, its k-major fragments are LDS-read bound); fp32 - pays on the small-spatial conv
    // layers (16x16, 8x8: few workgroups per CU, +4-15 %), not on the large ones, the GPT linears or any wgrad
    // (measured, tools/bench_igemm.py)
    const bool can32 = MODE != MODE_WGRAD && fast && fast_walk_ok<MODE>(p, 32);
    const bool bk32 = can32 && (g_ds6g_bf16 ? g_bf16_bk32 != 0
                                            : (g_f32_bk32 == 1 || (g_f32_bk32 == 2 && !p.is_linear && p.R * p.S > 1 &&
                                                                   (long)cdiv(p.Mg, 64) * cdiv(p.Ng, 64) <= 1024)));
    const int variant = 10000 * (int)bk32 + 1000 * (int)epi + 100 * (int)fast + 10 * MODE + tile;
    ProfRec* rec = nullptr;
    if (g_prof && g_prof->size() < g_prof_cap) {
        g_prof->push_back(ProfRec{});
        rec = &g_prof->back();
        rec->variant = variant;
        rec->flops = p.prof_flops > 0 ? p.prof_flops : 2.0 * p.Mg * p.Ng * p.Kg;
        (void)hipEventCreate(&rec->e0);
        (void)hipEventCreate(&rec->e1);
        (void)hipEventRecord(rec->e0, st);
    }
    const int bfm = g_ds6g_bf16;
#define DS6G_TILE(EPI_, BK_, FAST_)                                               \
    do {                                                                          \
        if (bfm == 3) launch_tile<MODE, EPI_, BK_, 3, FAST_>(p, splits, tile, st); \
        else if (bfm == 2) launch_tile<MODE, EPI_, BK_, 2, FAST_>(p, splits, tile, st); \
        else if (bfm == 1) launch_tile<MODE, EPI_, BK_, 1, FAST_>(p, splits, tile, st); \
        else launch_tile<MODE, EPI_, BK_, 0, FAST_>(p, splits, tile, st);         \
    } while (0)
    if (bk32) {
        if constexpr (MODE != MODE_WGRAD) {
            if (epi) DS6G_TILE(1, 32, 1); else DS6G_TILE(0, 32, 1);
        }
    } else if (fast) {
        if (epi) DS6G_TILE(1, 16, 1); else DS6G_TILE(0, 16, 1);
    } else {
        if (epi) DS6G_TILE(1, 16, 0); else DS6G_TILE(0, 16, 0);
    }
#undef DS6G_TILE
    if (rec) (void)hipEventRecord(rec->e1, st);
    g_last_variant = variant;
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

template <int MODE>
int launch_igemm(IgemmParams& p, hipStream_t st) {
    return launch_igemm<MODE>(p, 1, pick_tile(p.Mg, p.Ng, 1), st);
}

int run_wgrad(IgemmParams& p, float* dw, int accumulate, float* dbias, int accumulate_b, float* ws, size_t ws_bytes,
              hipStream_t st) {
    const long out_elems = (long)p.Mg * p.Ng;
    DS6G_CHECK_ARG(out_elems % 4 == 0 && p.Mg % 4 == 0);
    const long slab_elems = out_elems + (dbias ? p.Mg : 0);
    // the output (a weight tensor) has few tiles and the reduction (pixels) is long: split K until the grid
    // holds enough workgroups for latency hiding, bounded by the workspace and by >= 4 k-tiles per split
    const int tile = (p.Mg >= 128 && g_wgrad_tile == 1) ? 1 : 2;
    const long tiles = (long)cdiv(p.Mg, tile == 1 ? 128 : 64) * cdiv(p.Ng, 64);
    // measured on gfx950 (tools/bench_igemm.py): convs are fastest with ~2048 workgroups in flight, the GPT
    // linears (large outputs, costlier slab reduction) with ~1024
    static long t_lin = -1, t_conv = -1;   // env overrides for tuning sweeps
    if (t_lin < 0) { const char* e = getenv("DS6G_WGRAD_TARGET_LIN"); t_lin = e ? atol(e) : 1024; }
    if (t_conv < 0) { const char* e = getenv("DS6G_WGRAD_TARGET_CONV"); t_conv = e ? atol(e) : 2048; }
    const long target_blocks = p.is_linear ? t_lin : t_conv;
    long splits = (target_blocks + tiles - 1) / tiles;
    const long max_by_k = (p.Kg + 64 - 1) / 64;
    if (splits > max_by_k) splits = max_by_k;
    const long max_by_ws = (long)(ws_bytes / (slab_elems * sizeof(float)));
    if (splits > max_by_ws) splits = max_by_ws;
    if (splits < 1) splits = 1;
    int kps = cdiv(cdiv(p.Kg, splits), 32) * 32;
    splits = cdiv(p.Kg, kps);
    p.k_per_split = kps;
    p.split_stride = (size_t)slab_elems;
    p.want_colsum = dbias != nullptr;
    {
        static int xcd_mode = -1;   // env DS6G_WGRAD_XCD: 0 off, 1 (default) few-tile convs, 2 every conv wgrad (experiment)
        if (xcd_mode < 0) { const char* e = getenv("DS6G_WGRAD_XCD"); xcd_mode = e ? atoi(e) : 1; }
        p.xcd_splits = (!p.is_linear && splits >= 8 && ((xcd_mode == 1 && tiles <= 12) || xcd_mode == 2)) ? 1 : 0;
    }
    if (splits == 1 && !accumulate && !dbias) {
        p.out = dw;
        return launch_igemm<MODE_WGRAD>(p, 1, tile, st);
    }
    DS6G_CHECK_ARG(ws != nullptr && (size_t)splits * slab_elems * sizeof(float) <= ws_bytes);
    p.out = ws;
    int rc = launch_igemm<MODE_WGRAD>(p, (int)splits, tile, st);
    if (rc) return rc;
    const long n4 = out_elems / 4, m4 = dbias ? p.Mg / 4 : 0;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(n4 + m4, RED_COLS)), dim3(256), 0, st, ws, dw, n4, dbias, m4,
                       (int)splits, (size_t)slab_elems, accumulate, accumulate_b);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

void fill_conv(IgemmParams& p, int N, int H, int W, int C, int K, int R, int S, int stride, int pad) {
    p = IgemmParams{};
    p.N = N; p.H = H; p.W = W; p.C = C; p.K = K; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
    p.Ho = (H + 2 * pad - R) / stride + 1;
    p.Wo = (W + 2 * pad - S) / stride + 1;
    p.drop_scale = 1.f;
    p.h0 = 0; p.hstep = 1; p.Hs = H; p.w0 = 0; p.wstep = 1; p.Ws = W;
    p.r0 = 0; p.rstep = 1; p.nr = R; p.s0 = 0; p.sstep = 1; p.ns = S;
}

// A Linear over M rows is the 1x1 convolution of ONE 1 x M image (not M 1x1 images): the per-k-tile pixel walk of the
// kernels then never wraps (a wrap costs a VALU loop iteration per crossed image row).
void fill_linear(IgemmParams& p, int M, int N, int K) {
    fill_conv(p, 1, 1, M, K, N, 1, 1, 1, 0);
    p.is_linear = 1;
}

// byte sizes of the three tensors of a conv (for the buffer descriptors); all must stay below the OOB sentinel
struct ConvBytes { size_t x, y, w; bool ok; };
ConvBytes conv_bytes(const IgemmParams& p) {
    ConvBytes b;
    b.x = (size_t)p.N * p.H * p.W * p.C * sizeof(float);
    b.y = (size_t)p.N * p.Ho * p.Wo * p.K * sizeof(float);
    b.w = (size_t)p.K * p.R * p.S * p.C * sizeof(float);
    b.ok = b.x < OOB_OFF && b.y < OOB_OFF && b.w < OOB_OFF;
    return b;
}

}  // namespace

extern "C" {

int ds6g_last_igemm_variant(void) { return g_last_variant; }

int ds6g_profile_begin(int max_records) {
    if (max_records <= 0) return DS6G_ERR_ARG;
    if (!g_prof) g_prof = new std::vector<ProfRec>();
    for (auto& r : *g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_prof->clear();
    g_prof->reserve((size_t)max_records);
    g_prof_cap = (size_t)max_records;
    return DS6G_OK;
}

int ds6g_profile_end(int* variants, double* flops, float* ms, int cap) {
    if (!g_prof) return 0;
    int n = 0;
    for (auto& r : *g_prof) {
        (void)hipEventSynchronize(r.e1);
        float t = 0.f;
        (void)hipEventElapsedTime(&t, r.e0, r.e1);
        if (n < cap && variants && flops && ms) { variants[n] = r.variant; flops[n] = r.flops; ms[n] = t; ++n; }
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    g_prof->clear();
    g_prof_cap = 0;
    return n;
}

}  // extern "C"

// shared with bgemm.hip: the split-K slab reduction (weight gradient + optional bias-gradient tail)
int ds6g_internal_splitk_reduce(const float* ws, float* dw, long n4, float* dbias, long m4, int splits, size_t stride,
                                int accumulate, int accumulate_b, hipStream_t st) {
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(n4 + m4, RED_COLS)), dim3(256), 0, st, ws, dw, n4, dbias, m4, splits,
                       stride, accumulate, accumulate_b);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

// shared with winograd.hip: open / close one profiler record around a kernel launch (no-ops unless profiling is on)
void* ds6g_prof_open(int variant, double flops, hipStream_t st) {
    if (!g_prof || g_prof->size() >= g_prof_cap) return nullptr;
    g_prof->push_back(ProfRec{});
    ProfRec* rec = &g_prof->back();
    rec->variant = variant;
    rec->flops = flops;
    (void)hipEventCreate(&rec->e0);
    (void)hipEventCreate(&rec->e1);
    (void)hipEventRecord(rec->e0, st);
    return rec;
}
void ds6g_prof_close(void* rec, hipStream_t st) {
    if (rec) (void)hipEventRecord(static_cast<ProfRec*>(rec)->e1, st);
}

extern "C" {
int ds6g_set_debug_flags(int flags) {
    g_dbg = flags & 0xbf;  // 0x80: force the general (FAST 0) walk
    g_ds6g_attn_handover = (flags & 0x01000000) ? 0 : 1;
    g_ds6g_attn_fused128 = (flags & 0x04000000) ? 0 : 1;
    g_ds6g_attn_percu = (flags >> 20) & 0xf;  // attention: resident-workgroups-per-CU assumption of the split heuristic
    g_wino_kb64 = (flags & 0x02000000) ? 1 : 0;
    g_bf16_bk32 = (flags & 0x10000000) ? 0 : 1;
    g_f32_bk32 = (flags & 0x20000000) ? 1 : ((flags & 0x40000000) ? 0 : 2);   // fp32 32-column k-tiles: everywhere / never
    g_wgrad_tile = (flags & 0x40) ? 2 : 1;
    if ((flags >> 8) & 0xfff) g_min_blocks = (flags >> 8) & 0xfff;
    return 0;
}

int ds6g_conv2d_fwd(const float* x, const float* w, float* y, int N, int H, int W, int C, int K, int R,
                    int S, int stride, int pad, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(x && w && y && C % 4 == 0 && N > 0);
    IgemmParams p;
    fill_conv(p, N, H, W, C, K, R, S, stride, pad);
    p.a_src = x; p.b_src = w; p.out = y;
    const ConvBytes cb = conv_bytes(p);
    DS6G_CHECK_ARG(cb.ok);
    p.a_bytes = (unsigned)cb.x; p.b_bytes = (unsigned)cb.w;
    p.Mg = N * p.Ho * p.Wo; p.Ng = K; p.Kg = R * S * C;
    return launch_igemm<MODE_FWD>(p, (hipStream_t)stream);
}

int ds6g_conv2d_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int K, int R,
                      int S, int stride, int pad, int accumulate, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(dy && w && dx && C % 4 == 0 && K % 4 == 0);
    IgemmParams p;
    fill_conv(p, N, H, W, C, K, R, S, stride, pad);
    p.a_src = dy; p.b_src = w; p.out = dx; p.accumulate = accumulate;
    const ConvBytes cb = conv_bytes(p);
    DS6G_CHECK_ARG(cb.ok);
    p.a_bytes = (unsigned)cb.y; p.b_bytes = (unsigned)cb.w;
    if (stride == 2 && H % 2 == 0 && W % 2 == 0) {
        // four input-pixel parity classes in blockIdx.y of ONE launch (class parameters are derived in the kernel);
        // host-side values describe the largest class (tile choice, eligibility of the uniform walk)
        IgemmParams q = p;
        q.nclass = 4;
        q.hstep = 2; q.Hs = H / 2; q.wstep = 2; q.Ws = W / 2; q.rstep = 2; q.sstep = 2;
        q.h0 = 0; q.w0 = 0; q.r0 = 0; q.s0 = 0; q.nr = (R + 1) / 2; q.ns = (S + 1) / 2;
        q.Mg = N * q.Hs * q.Ws; q.Ng = C; q.Kg = q.nr * q.ns * K;
        q.prof_flops = 2.0 * q.Mg * q.Ng * (double)(R * S) * K;   // all taps are visited exactly once over the classes
        return launch_igemm<MODE_DGRAD>(q, (hipStream_t)stream);
    }
    p.Mg = N * H * W; p.Ng = C; p.Kg = R * S * K;
    return launch_igemm<MODE_DGRAD>(p, (hipStream_t)stream);
}

int ds6g_conv2d_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K, int R,
                      int S, int stride, int pad, int accumulate, float* ws, size_t ws_bytes, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(x && dy && dw && C % 4 == 0 && K % 4 == 0);
    IgemmParams p;
    fill_conv(p, N, H, W, C, K, R, S, stride, pad);
    p.a_src = dy; p.b_src = x;
    const ConvBytes cb = conv_bytes(p);
    DS6G_CHECK_ARG(cb.ok);
    p.a_bytes = (unsigned)cb.y; p.b_bytes = (unsigned)cb.x;
    p.Mg = K; p.Ng = R * S * C; p.Kg = N * p.Ho * p.Wo;
    return run_wgrad(p, dw, accumulate, nullptr, 0, ws, ws_bytes, (hipStream_t)stream);
}

// inference form of Conv2d + eval-mode BatchNorm2d (+ residual) (+ ReLU) with the BN folded into w / bias
// (ds6g_bn_fold): y = act(conv(x, w) + bias [+ residual]);  relu: 0 none, 1 before the residual add, 2 after it
int ds6g_conv2d_bias_act_fwd(const float* x, const float* w, const float* bias, const float* residual, float* y, int N,
                             int H, int W, int C, int K, int R, int S, int stride, int pad, int relu, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(x && w && y && C % 4 == 0 && N > 0 && relu >= 0 && relu <= 2);
    IgemmParams p;
    fill_conv(p, N, H, W, C, K, R, S, stride, pad);
    p.a_src = x; p.b_src = w; p.out = y; p.bias = bias; p.residual = residual; p.relu = relu;
    const ConvBytes cb = conv_bytes(p);
    DS6G_CHECK_ARG(cb.ok);
    p.a_bytes = (unsigned)cb.x; p.b_bytes = (unsigned)cb.w;
    p.Mg = N * p.Ho * p.Wo; p.Ng = K; p.Kg = R * S * C;
    return launch_igemm<MODE_FWD>(p, (hipStream_t)stream);
}

// y[M][N] = residual + dropout( act( x[M][K] @ w[N][K]^T + bias ) )
int ds6g_linear_fwd(const float* x, const float* w, const float* bias, float* y, int M, int N, int K, int relu,
                    const float* residual, float drop_p, uint64_t seed, uint64_t seed_off, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(x && w && y && K % 4 == 0 && M > 0 && drop_p >= 0.f && drop_p < 1.f);
    IgemmParams p;
    fill_linear(p, M, N, K);
    p.a_src = x; p.b_src = w; p.out = y; p.bias = bias; p.relu = relu; p.residual = residual;
    p.drop_thr = ds6g_drop_threshold(drop_p);
    p.drop_scale = 1.f / (1.f - drop_p);
    p.seed = seed; p.seed_off = seed_off; p.salt = g_ds6g_salt;
    const ConvBytes cb = conv_bytes(p);
    DS6G_CHECK_ARG(cb.ok);
    p.a_bytes = (unsigned)cb.x; p.b_bytes = (unsigned)cb.w;
    p.Mg = M; p.Ng = N; p.Kg = K;
    return launch_igemm<MODE_FWD>(p, (hipStream_t)stream);
}

// dx[M][K] (+)= (dy[M][N] @ w[N][K]) * (mask_src > 0)
int ds6g_linear_dgrad(const float* dy, const float* w, float* dx, int M, int N, int K, const float* mask_src,
                      int accumulate, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(dy && w && dx && K % 4 == 0 && N % 4 == 0);
    IgemmParams p;
    fill_linear(p, M, N, K);
    p.a_src = dy; p.b_src = w; p.out = dx; p.mask_src = mask_src; p.accumulate = accumulate;
    const ConvBytes cb = conv_bytes(p);
    DS6G_CHECK_ARG(cb.ok);
    p.a_bytes = (unsigned)cb.y; p.b_bytes = (unsigned)cb.w;
    p.Mg = M; p.Ng = K; p.Kg = N;
    return launch_igemm<MODE_DGRAD>(p, (hipStream_t)stream);
}

// dw[N][K] (+)= dy[M][N]^T @ x[M][K];  dbias[N] (+)= column sums of dy (nullable; fused into the same kernel)
int ds6g_linear_wgrad(const float* x, const float* dy, float* dw, float* dbias, int M, int N, int K, int accumulate,
                      float* ws, size_t ws_bytes, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(x && dy && dw && K % 4 == 0 && N % 4 == 0);
    IgemmParams p;
    fill_linear(p, M, N, K);
    p.a_src = dy; p.b_src = x;
    const ConvBytes cb = conv_bytes(p);
    DS6G_CHECK_ARG(cb.ok);
    p.a_bytes = (unsigned)cb.y; p.b_bytes = (unsigned)cb.x;
    p.Mg = N; p.Ng = K; p.Kg = M;
    return run_wgrad(p, dw, accumulate, dbias, accumulate, ws, ws_bytes, (hipStream_t)stream);
}

}  // extern "C"
