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
//
// Tiles (32 rows x HD) of K/V (or Q/dO in the dK/dV kernel) go global -> LDS by buffer_load...lds DMA, one tile
// ahead, two buffers; the image is [row][HD] with the 16-B chunk index XOR-swizzled by the row, which makes both
// access shapes conflict-free: "row on the lane" (ds_read_b128: 4 reduction steps per read, the reduction order
// over the head dim is permuted so a lane half owns a contiguous half of it) and "dim on the lane" (ds_read_b32).
// Workgroup = 4 waves = 128 queries (K/V tile reuse 4x).  The key loop (query loop for dK/dV) can be SPLIT
// across workgroups so the grid fills the 256 CUs evenly at any batch size; partial results are merged by a
// small second kernel (flash-decoding style for the forward: rescale by exp(m_s - m)).  Deterministic: no atomics.
#include "common.h"

int g_ds6g_attn_percu = 0;
int g_ds6g_attn_handover = 1;  // 0: backward recomputes S / dP in every kernel (ds6g_set_debug_flags 0x01000000)
int g_ds6g_attn_fused128 = 1;  // 0 (ds6g_set_debug_flags 0x04000000): hd = 128 backward as dK kernel + dropped-P tiles + dV kernel

// -DDS6G_ATTN_CLOCKS (tools/attn_clocks.py builds its own library): wave 0 of workgroup 0 of attn_bwd_dkv_kernel sums the
// clocks it spends in each phase of a tile step into g_attn_clk[phase] (s_memtime, 100 MHz-independent shader clock)
#ifdef DS6G_ATTN_CLOCKS
__device__ unsigned long long g_attn_clk[16];
extern "C" int ds6g_attn_clocks_read(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_attn_clk), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_attn_clk), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#define ATTN_CLK_DECL() unsigned long long clk_t_ = 0; const bool clk_on_ = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0;
#define ATTN_CLK_START() do { __builtin_amdgcn_sched_barrier(0); clk_t_ = __builtin_readcyclecounter(); } while (0)
#define ATTN_CLK(ph) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_readcyclecounter(); \
        if (clk_on_) g_attn_clk[ph] += n_ - clk_t_; clk_t_ = n_; } while (0)
#else
#define ATTN_CLK_DECL()
#define ATTN_CLK_START()
#define ATTN_CLK(ph)
#endif

namespace {

struct AttnParams {
    const float* q;
    const float* k;
    const float* v;
    float* o;            // fwd: output (or partial slabs); bwd: forward output (read)
    float* lse;          // [B][nh][T]
    float* ml;           // fwd split: [S][B][nh][T][2] running max / partial sum
    const float* d_o;    // bwd
    float* delta;        // [B][nh][T]  rowsum(dO * O)
    float* dq;
    float* dk;
    float* dv;
    int T, nh, B;
    int ld;              // row stride (floats) of o / do and of the split slabs
    int ldq;             // row stride of q / k / v (3C when they are column blocks of one fused projection output)
    int ldd;             // row stride of the final dq / dk / dv
    int splits, tiles_per_split;
    size_t slab;         // floats between split slabs of o / dq / dk / dv
    unsigned bytes;      // size of each [B*T][ld] tensor
    unsigned bytes_q;    // bytes addressable from the q / k / v pointers
    float scale;
    uint32_t thr;
    float dscale;
    uint64_t seed;
    uint64_t seed_off;
    const uint64_t* salt;   // device-resident addend of seed_off (nullable), see ds6g_set_dropout_salt
    // backward hand-over (see attn_bwd_dkv_kernel): 32 x 32 tiles of dS / dropped P in the dK/dV kernel's accumulator
    // order, [b*nh + h][query tile < nkg][key group < nkg][reg / 4][lane 64][reg % 4]
    float* hs;
    float* hp;
    int nkg;
    // bf16-storage path: the final o (forward) / dq, dk, dv (backward) are GEMM operands and are written as bf16 (the
    // split slabs and every kernel-internal tensor stay fp32); o16: the backward reads its forward output as bf16
    int out16;
    int o16;
    int in16;   // q / k / v / d_o are bf16 in HBM (kernels instantiated with BF == 4)
};

__device__ __forceinline__ int krow16(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// XOR applied to the 16-B chunk index of row r (NC = HD/4 chunks per row)
template <int HD>
__device__ __forceinline__ int swz(int r) {
    constexpr int NC = HD / 4;
    if (NC >= 16) return r & 15;
    if (NC == 8) return (r >> 1) & 7;
    return (r >> 2) & 3;
}

// Tiles are addressed by their LDS BYTE address, laundered once per loop step through an empty asm: the swizzled
// per-lane addresses are loop-invariant, and hipcc would otherwise hoist ~250 of them out of the tile loop into
// VGPRs (spilling the dK/dV kernel).  Recomputing them next to each read is free under the 64-cycle MFMAs.
typedef __attribute__((address_space(3))) const float lds_cf;
typedef __attribute__((address_space(3))) const f32x4 lds_cf4;
__device__ __forceinline__ unsigned opaque_tile(const float* tile) {
    unsigned a = lds_addr(tile);
    asm volatile("" : "+v"(a));
    return a;
}
// "row on the lane": 4 consecutive reduction elements (chunk c) of row r
template <int HD>
__device__ __forceinline__ f32x4 row_read(unsigned tile, int r, int c) {
    return *reinterpret_cast<lds_cf4*>(tile + (unsigned)(r * HD * 4 + ((c ^ swz<HD>(r)) << 4)));
}
// "dim on the lane": element d = 32*blk + (lane&31) of row krow16(r, half).  The swizzled address splits into a
// lane part that takes only 8 distinct values (dtab, built once per kernel) and a compile-time part that folds
// into the ds_read offset field:   row = cr + 4*half with bit 2 of cr clear, so the row XOR is (const) ^ (half term).
// Exact-fp32 path at HD >= 64 ("vector dims"): a lane owns NB = HD / 32 CONSECUTIVE head dims (NB * l31 ..) instead of one dim
// per 32-dim block, so its A operands of all NB blocks of one k-step are ONE ds_read_b128 (HD 128) / ds_read_b64 (HD 64) -
// 16 LDS reads per product instead of 64 / 32 (measured in attn_bwd_dkv_kernel<128>: the dim-on-the-lane product took 5 750
// clocks per tile for 4 096 of MFMA, the row-on-the-lane products with their 16 ds_read_b128 4 700).  Accumulator register r
// of block b then holds dim NB * krow16(r, half) + b (store_rows).  The reads stay conflict-free: the 32 lanes of a half
// cover one whole swizzled row.
template <int HD, int BF>
__device__ __forceinline__ constexpr bool vdims() { return BF == 0 && HD >= 64; }
// Exact-fp32 path at HD = 16: the products whose OUTPUT dimension is the head dim (P V, dS^T Q, P^T dO, dS K) have M = 16.
// On the 32x32x2 MFMA half of every result row is padding; v_mfma_f32_16x16x1_4b_f32 (four independent 16x16 outer
// products per instruction, 32 cycles) does the same work without it: lane l belongs to block l / 16, supplies
// A[dim l % 16] (row krow16(r, half) of the tile) and B[column l % 16] = its own p[r], so the four blocks are
// (columns 0-15 | 16-31) x (rows of lane-half 0 | 1).  Result register 4 blk + i of lane l = element (dim 4 (l / 16) + i,
// column l % 16) of block blk: a column's total is block (c / 16) + block (c / 16 + 2), summed when the rows are stored.
template <int HD, int BF>
__device__ __forceinline__ constexpr bool quad16() { return BF == 0 && HD == 16; }
template <int HD>
__device__ __forceinline__ void make_dtab_v(unsigned* dtab, int l31, int half) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        // row XOR (swz<HD>, HD >= 64) = row & 15 = (cr & 15) | 4 half with cr & 15 in {0,1,2,3,8,9,10,11} <-> k = (cr & 3) | bit3 << 2
        const int xv = (k & 3) | ((k >> 2) << 3);
        if (HD >= 128) dtab[k] = (unsigned)(4 * half * HD * 4 + (((l31 ^ (4 * half) ^ xv) & 31) << 4));
        else dtab[k] = (unsigned)(4 * half * HD * 4 + ((((l31 >> 1) ^ (4 * half) ^ xv) & 15) << 4) + (l31 & 1) * 8);
    }
}
typedef float f32x2_ __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const f32x2_ lds_cf2;
// the lane's NB consecutive dims of row krow16(r, half): v[b] is the A operand of block b
template <int HD>
__device__ __forceinline__ void dimv_read(float* v, unsigned tile, const unsigned* dtab, int r) {
    const int cr = krow16(r, 0);
    if (HD >= 128) {
        const f32x4 t = *reinterpret_cast<lds_cf4*>(tile + dtab[(cr & 3) | (((cr >> 3) & 1) << 2)] + (unsigned)(cr * HD * 4));
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    } else {
        const f32x2_ t = *reinterpret_cast<lds_cf2*>(tile + dtab[(cr & 3) | (((cr >> 3) & 1) << 2)] + (unsigned)(cr * HD * 4));
        v[0] = t[0]; v[1] = t[1];
    }
}

template <int HD>
__device__ __forceinline__ void make_dtab(unsigned* dtab, int l31, int half) {
    constexpr int NC = HD / 4;
    const int dc = HD < 32 ? (l31 & (HD - 1)) : l31;  // HD = 16: lanes 16..31 read dim l31 - 16 (the A operand of their 16x16 block)
    const int hx = NC >= 16 ? 4 * half : (NC == 8 ? 2 * half : half);
#pragma unroll
    for (int k = 0; k < 8; ++k)
        dtab[k] = (unsigned)(4 * half * HD * 4 + ((((dc >> 2) ^ hx ^ k) & (NC - 1)) << 4) + (dc & 3) * 4);
}
template <int HD>
__device__ __forceinline__ float dim_read(unsigned tile, const unsigned* dtab, int r, int blk) {
    constexpr int NC = HD / 4;
    const int cr = krow16(r, 0);
    const int k = NC >= 16 ? (cr & 7) : (NC == 8 ? ((cr >> 1) & 7) : ((cr >> 2) & 3));
    const int cst = cr * HD * 4 + (NC >= 16 ? 128 * (blk ^ ((cr >> 3) & 1)) : 0);
    return *reinterpret_cast<lds_cf*>(tile + dtab[k] + (unsigned)cst);
}

// ---- bf16-STORED tiles (template value BF == 4: q / k / v / dO live in HBM as bf16; implies the bf16 matrix cores) -------
// LDS image [32 rows][HD] bf16, row = HD*2 bytes, filled by LDS-DMA; the 16-B chunk index (8 dims) is XOR-swizzled by the
// row so that BOTH read patterns are conflict-free: "row on the lane" (one ds_read_b128 = the MFMA operand of a 16-deep
// k-step as it lies in memory) and "dim on the lane" (ds_read_b64_tr_b16: 4 rows x 16 dims per 16-lane group, delivered
// transposed - two of them per k-step and 32-dim block).  HD = 128 uses the XOR the CDNA guide gives for 256-byte rows.
template <int HD>
__device__ __forceinline__ int swz16(int r) {
    if (HD >= 128) return ((r & 3) << 2) | ((r >> 2) & 3);
    if (HD == 64) return (((r >> 1) & 1) << 2) | ((r >> 2) & 3);
    if (HD == 32) return (r >> 2) & 3;
    return (r >> 3) & 1;
}
typedef __attribute__((address_space(3))) const bf16x8 lds_cb8;
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
template <int HD>
__device__ __forceinline__ bf16x8 row_read16(unsigned tile, int r, int c) {
    return *reinterpret_cast<lds_cb8*>(tile + (unsigned)(r * HD * 2 + ((c ^ swz16<HD>(r)) << 4)));
}
// operand fragment of k-step s2 for "dim on the lane": element e <-> row krow16(8 s2 + e, half), dim 32 blk + (lane & 31)
template <int HD>
__device__ __forceinline__ bf16x8 dim_read16(unsigned tile, int s2, int blk, int lane) {
    const int half = lane >> 5, g = (lane >> 4) & 1, q = (lane >> 2) & 3, pp = lane & 3;
    int dim0 = blk * 32 + 16 * g + 4 * pp;
    if (HD < 32 && dim0 >= HD) dim0 -= 16;  // HD = 16: lanes 16..31 re-read valid dims (their outputs are never stored)
    bf16x4v part[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int row = 16 * s2 + 8 * t + 4 * half + q;
        const unsigned a = tile + (unsigned)(row * HD * 2 + (((dim0 >> 3) ^ swz16<HD>(row)) << 4) + (dim0 & 7) * 2);
        part[t] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4v*)(unsigned long)a);
    }
    return __builtin_shufflevector(part[0], part[1], 0, 1, 2, 3, 4, 5, 6, 7);
}
// the lane's half row (dims half*HD/2 ..) as packed bf16 in the first HD/4 floats of reg (raw bits; no scaling possible)
template <int HD>
__device__ __forceinline__ void load_frag16(float* reg, const float* base, long row_off, int half) {
    const __bf16* b16 = reinterpret_cast<const __bf16*>(base);
#pragma unroll
    for (int j = 0; j < HD / 16; ++j) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(b16 + row_off + half * (HD / 2) + 8 * j);
        reg[4 * j] = v[0]; reg[4 * j + 1] = v[1]; reg[4 * j + 2] = v[2]; reg[4 * j + 3] = v[3];
    }
}
template <int HD, int BF>
__device__ __forceinline__ void load_frag_any(float* reg, const float* base, long row_off, int half, float mul);

// per-lane operand fragment of row `row` for the transposed products: element ss <-> dim half*HD/2 + ss
template <int HD>
__device__ __forceinline__ void load_frag(float* reg, const float* base, long row_off, int half, float mul) {
#pragma unroll
    for (int j = 0; j < HD / 8; ++j) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(base + row_off + half * (HD / 2) + 4 * j);
        reg[4 * j] = v[0] * mul; reg[4 * j + 1] = v[1] * mul; reg[4 * j + 2] = v[2] * mul; reg[4 * j + 3] = v[3] * mul;
    }
}

// LDS-DMA of 32 x HD tiles (rows row0 .. row0 + 31 of ONE batch element's [T][ld] block, head h) into the swizzled LDS
// image: fp32 tiles [row][HD] with the 16-B chunk index XOR swz<HD>(row), bf16 tiles (BF == 4) with swz16<HD>(row); one
// wave-instruction moves 1 KiB, wave w issues the pieces w, w + 4, ...  Everything that does not change from tile to tile
// is computed ONCE per kernel: the lane's byte offset of each of its pieces inside a tile at row 0 (division, modulo,
// swizzle, head column), so a tile costs one add and one DMA instruction per piece (measured before: 1 200 clocks of address
// arithmetic per tile pair in attn_bwd_dkv_kernel<128>).  The buffer descriptor covers exactly this batch element's T
// rows, so the tail rows of the last tile (and whole tiles beyond T) fall off its end and the hardware writes zeros for
// them - no per-lane row test.
template <int HD, int BF>
struct TileLoader {
    static constexpr int EL = BF == 4 ? 2 : 4;        // bytes per element
    static constexpr int CH = 16 / EL;                // elements per 16-B chunk
    static constexpr int NC = HD / CH;                // chunks per row
    static constexpr int NWI = NC / 2;                // wave-instructions (64 chunks) per 32-row tile
    static constexpr int NP = (NWI + 3) / 4;          // pieces per wave
    i32x4 srd;
    unsigned voff[NP];
    unsigned row_bytes;
    __device__ __forceinline__ TileLoader(const float* tensor, long batch_off, int T, int ld, int col0, int wave, int lane) {
        const char* base = reinterpret_cast<const char*>(tensor) + batch_off * EL;
        srd = make_srd(base, (unsigned)((long)T * ld * EL));
        row_bytes = (unsigned)(ld * EL);
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int g = (wave + 4 * j) * 64 + lane;
            const int row = g / NC, pos = g % NC;
            const int c = pos ^ (BF == 4 ? swz16<HD>(row) : swz<HD>(row));
            voff[j] = (unsigned)((row * ld + col0 + c * CH) * EL);
        }
    }
    __device__ __forceinline__ void issue(const float* lds_tile, int row0, int wave) const {
        const unsigned rb = (unsigned)row0 * row_bytes;
#pragma unroll
        for (int j = 0; j < NP; ++j)
            if (NWI >= 4 || wave + 4 * j < NWI) dma16(srd, lds_addr(lds_tile) + (unsigned)(wave + 4 * j) * 1024u, voff[j] + rb);
    }
};

// bf16 storage: the fragment stays packed bf16 and is NOT scaled (callers scale the product instead; mul 0 = zero it)
template <int HD, int BF>
__device__ __forceinline__ void load_frag_any(float* reg, const float* base, long row_off, int half, float mul) {
    if (BF == 4) {
        load_frag16<HD>(reg, base, row_off, half);
        if (mul == 0.f) {
#pragma unroll
            for (int j = 0; j < HD / 4; ++j) reg[j] = 0.f;
        }
    } else {
        load_frag<HD>(reg, base, row_off, half, mul);
    }
}

// acc += A . B^T over the head dim: A rows come from the LDS tile (row = lane&31), B from registers
template <int HD, int BF>
__device__ __forceinline__ void mma_rows(f32x16& acc, unsigned tile, const float* breg, int l31, int half) {
    constexpr int NCH = HD / 8;  // 16-B chunks per lane half
    if (BF == 4) {  // bf16-stored: LDS chunk and register chunk ARE the operands (lane half h: dims h*HD/2 + 8c .. + 7)
#pragma unroll
        for (int c = 0; c < HD / 16; ++c) {
            const bf16x8 a8 = row_read16<HD>(tile, l31, half * (HD / 16) + c);
            const bf16x8 b8 = __builtin_bit_cast(bf16x8, f32x4{breg[4 * c], breg[4 * c + 1], breg[4 * c + 2], breg[4 * c + 3]});
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc, 0, 0, 0);
        }
        return;
    }
    if (BF) {  // bf16 matrix cores: 16 head dims per MFMA (lane half h supplies dims h*HD/2 + 8c .. +7 of chunk c)
#pragma unroll
        for (int c = 0; c < HD / 16; ++c) {
            const f32x4 v0 = row_read<HD>(tile, l31, half * NCH + 2 * c), v1 = row_read<HD>(tile, l31, half * NCH + 2 * c + 1);
            bf16x8 a8, b8;
            if (BF == 3) {
                const float av[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                bf16x8 am, al, bm, bl;
                split3_bf16x8(av, a8, am, al);
                split3_bf16x8(breg + 8 * c, b8, bm, bl);
                acc = mfma_x6(a8, am, al, b8, bm, bl, acc);
                continue;
            }
            if (BF == 2) {  // split bf16: hi*hi + hi*lo + lo*hi (common.h)
                const float av[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                bf16x8 al, bl;
                split_bf16x8(av, a8, al);
                split_bf16x8(breg + 8 * c, b8, bl);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b8, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc, 0, 0, 0);
                continue;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a8[e] = (__bf16)v0[e];
                a8[4 + e] = (__bf16)v1[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) b8[e] = (__bf16)breg[8 * c + e];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc, 0, 0, 0);
        }
        return;
    }
    f32x4 cur = row_read<HD>(tile, l31, half * NCH);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const f32x4 nxt = row_read<HD>(tile, l31, half * NCH + (j + 1 < NCH ? j + 1 : j));
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[e], breg[4 * j + e], acc, 0, 0, 0);
        cur = nxt;
    }
}

// acc[blk] (X^T[d][col-on-lane]) += sum_r tile[krow16(r,half)][d] * p[r]
template <int HD, int BF>
__device__ __forceinline__ void mma_dims(f32x16* acc, unsigned tile, const f32x16& p, const unsigned* dtab) {
    constexpr int NB = (HD + 31) / 32;
    if (BF == 4) {  // bf16-stored tile: the A fragment comes transposed out of LDS (ds_read_b64_tr_b16), no conversion
        const int lane_ = (int)(threadIdx.x & 63);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 b8;
#pragma unroll
            for (int e = 0; e < 8; ++e) b8[e] = (__bf16)p[8 * s2 + e];
#pragma unroll
            for (int blk = 0; blk < NB; ++blk)
                acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dim_read16<HD>(tile, s2, blk, lane_), b8, acc[blk], 0, 0, 0);
        }
        return;
    }
    if (BF) {  // accumulator registers 8s..8s+7 are exactly the bf16 operand fragment of k-step s (key order krow16)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 b8, bl, bm;
            if (BF >= 2) {
                float pv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) pv[e] = p[8 * s2 + e];
                if (BF == 3) split3_bf16x8(pv, b8, bm, bl); else split_bf16x8(pv, b8, bl);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) b8[e] = (__bf16)p[8 * s2 + e];
            }
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                bf16x8 a8;
                if (BF == 3) {
                    float av[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) av[e] = dim_read<HD>(tile, dtab, 8 * s2 + e, blk);
                    bf16x8 am, al;
                    split3_bf16x8(av, a8, am, al);
                    acc[blk] = mfma_x6(a8, am, al, b8, bm, bl, acc[blk]);
                    continue;
                }
                if (BF == 2) {
                    float av[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) av[e] = dim_read<HD>(tile, dtab, 8 * s2 + e, blk);
                    bf16x8 al;
                    split_bf16x8(av, a8, al);
                    acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b8, acc[blk], 0, 0, 0);
                    acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, bl, acc[blk], 0, 0, 0);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) a8[e] = (__bf16)dim_read<HD>(tile, dtab, 8 * s2 + e, blk);
                }
                acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[blk], 0, 0, 0);
            }
        }
        return;
    }
    if constexpr (quad16<HD, BF>()) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x1f32(dim_read<HD>(tile, dtab, r, 0), p[r], acc[0], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int r = 0; r < 14; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        return;
    }
    if constexpr (vdims<HD, BF>()) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float a[NB];
            dimv_read<HD>(a, tile, dtab, r);
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) acc[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[blk], p[r], acc[blk], 0, 0, 0);
        }
        // one wide LDS read per k-step, issued one step ahead of the NB MFMAs that consume it
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int r = 0; r < 14; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, NB, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NB, 0);
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
            acc[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(dim_read<HD>(tile, dtab, r, blk), p[r], acc[blk], 0, 0, 0);
    }
    // pin the schedule: LDS reads run exactly one r-step ahead of the MFMAs that consume them (left alone the
    // scheduler issues all 16*NB reads first and keeps 16*NB values + addresses live)
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * NB, 0);
#pragma unroll
    for (int r = 0; r < 14; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, NB, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NB, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 2 * NB, 0);
}

// write an accumulator set acc[blk][16] = X^T[d][row] to global X[row][d] (row on the lane)
template <int HD, int BF>
__device__ __forceinline__ void store_rows16(const f32x16* acc, __bf16* base, int ld, int row, int T, int half, float mul) {
    constexpr int NB = (HD + 31) / 32;
    typedef __bf16 bf16x4_ __attribute__((ext_vector_type(4)));
    if constexpr (quad16<HD, BF>()) {  // (see quad16) this lane holds dims 4 (lane / 16) .. + 3 of rows l16 and 16 + l16
        const int lane = (int)(threadIdx.x & 63), l16 = lane & 15, d = 4 * (lane >> 4);
        const int r_lo = row - (lane & 31) + l16;
        const float m_lo = __shfl(mul, l16, 64), m_hi = __shfl(mul, 16 + l16, 64);
        if (r_lo < T)
            *reinterpret_cast<bf16x4_*>(base + (long)r_lo * ld + d) =
                bf16x4_{(__bf16)((acc[0][0] + acc[0][8]) * m_lo), (__bf16)((acc[0][1] + acc[0][9]) * m_lo),
                        (__bf16)((acc[0][2] + acc[0][10]) * m_lo), (__bf16)((acc[0][3] + acc[0][11]) * m_lo)};
        if (r_lo + 16 < T)
            *reinterpret_cast<bf16x4_*>(base + (long)(r_lo + 16) * ld + d) =
                bf16x4_{(__bf16)((acc[0][4] + acc[0][12]) * m_hi), (__bf16)((acc[0][5] + acc[0][13]) * m_hi),
                        (__bf16)((acc[0][6] + acc[0][14]) * m_hi), (__bf16)((acc[0][7] + acc[0][15]) * m_hi)};
        return;
    }
    if (row >= T) return;
    if constexpr (vdims<HD, BF>()) {   // register r of block b = dim NB * krow16(r, half) + b
#pragma unroll
        for (int r = 0; r < 16; r += (NB == 4 ? 1 : 2)) {
            const int d = NB * krow16(r, half);
            const float v0 = acc[0][r] * mul, v1 = acc[1][r] * mul;
            const float v2 = (NB == 4 ? acc[2][r] : acc[0][r + 1]) * mul, v3 = (NB == 4 ? acc[3][r] : acc[1][r + 1]) * mul;
            *reinterpret_cast<bf16x4_*>(base + (long)row * ld + d) = bf16x4_{(__bf16)v0, (__bf16)v1, (__bf16)v2, (__bf16)v3};
        }
        return;
    }
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = blk * 32 + 8 * g + 4 * half;
            if (d < HD)
                *reinterpret_cast<bf16x4_*>(base + (long)row * ld + d) =
                    bf16x4_{(__bf16)(acc[blk][4 * g] * mul), (__bf16)(acc[blk][4 * g + 1] * mul),
                            (__bf16)(acc[blk][4 * g + 2] * mul), (__bf16)(acc[blk][4 * g + 3] * mul)};
        }
    }
}
template <int HD, int BF>
__device__ __forceinline__ void store_rows(const f32x16* acc, float* base, int ld, int row, int T, int half, float mul) {
    constexpr int NB = (HD + 31) / 32;
    if constexpr (quad16<HD, BF>()) {  // (see quad16) this lane holds dims 4 (lane / 16) .. + 3 of rows l16 and 16 + l16
        const int lane = (int)(threadIdx.x & 63), l16 = lane & 15, d = 4 * (lane >> 4);
        const int r_lo = row - (lane & 31) + l16;
        const float m_lo = __shfl(mul, l16, 64), m_hi = __shfl(mul, 16 + l16, 64);
        if (r_lo < T)
            *reinterpret_cast<f32x4*>(base + (long)r_lo * ld + d) =
                f32x4{(acc[0][0] + acc[0][8]) * m_lo, (acc[0][1] + acc[0][9]) * m_lo, (acc[0][2] + acc[0][10]) * m_lo,
                      (acc[0][3] + acc[0][11]) * m_lo};
        if (r_lo + 16 < T)
            *reinterpret_cast<f32x4*>(base + (long)(r_lo + 16) * ld + d) =
                f32x4{(acc[0][4] + acc[0][12]) * m_hi, (acc[0][5] + acc[0][13]) * m_hi, (acc[0][6] + acc[0][14]) * m_hi,
                      (acc[0][7] + acc[0][15]) * m_hi};
        return;
    }
    if (row >= T) return;
    if constexpr (vdims<HD, BF>()) {   // register r of block b = dim NB * krow16(r, half) + b
#pragma unroll
        for (int r = 0; r < 16; r += (NB == 4 ? 1 : 2)) {
            const int d = NB * krow16(r, half);
            const f32x4 v = {acc[0][r] * mul, acc[1][r] * mul, (NB == 4 ? acc[2][r] : acc[0][r + 1]) * mul,
                             (NB == 4 ? acc[3][r] : acc[1][r + 1]) * mul};
            *reinterpret_cast<f32x4*>(base + (long)row * ld + d) = v;
        }
        return;
    }
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

#define ATTN_COMMON()                                                                                  \
    __shared__ __attribute__((aligned(16))) float Xa0[32 * HD];                                        \
    __shared__ __attribute__((aligned(16))) float Xa1[32 * HD];                                        \
    __shared__ __attribute__((aligned(16))) float Xb0[32 * HD];                                        \
    __shared__ __attribute__((aligned(16))) float Xb1[32 * HD];                                        \
    ATTN_GEOM()
#define ATTN_GEOM()                                                                                    \
    constexpr int NB = (HD + 31) / 32;                                                                 \
    const int tid = threadIdx.x, lane = tid & 63;                                                      \
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                                         \
    const int l31 = lane & 31, half = lane >> 5;                                                       \
    /* XCD-aware order: the hardware deals consecutive workgroups round-robin over the 8 XCDs (private L2s); \
       remap so that the workgroups sharing one (batch, head)'s K/V (or Q/dO) tiles run on ONE XCD and hit its  \
       L2 instead of each re-fetching them (measured: 5.7x the algorithmic bytes without this). */            \
    int bx_, by_, bz_;                                                                                 \
    {                                                                                                  \
        const int nwg = gridDim.x * gridDim.y * gridDim.z;                                             \
        const int orig = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);               \
        const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;                                         \
        const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);    \
        bx_ = wg % gridDim.x;                                                                          \
        by_ = (wg / gridDim.x) % gridDim.y;                                                            \
        bz_ = wg / (gridDim.x * gridDim.y);                                                            \
    }                                                                                                  \
    const int h = by_ % p.nh, split = by_ / p.nh, b = bz_;                                             \
    const int T = p.T;                                                                                 \
    [[maybe_unused]] const uint64_t seed_off_ = p.seed_off + (p.salt ? *p.salt : (uint64_t)0);         \
    const long head_off = (long)b * T * p.ld + h * HD;                                                 \
    [[maybe_unused]] const long head_offq = (long)b * T * p.ldq + h * HD;                                            \
    [[maybe_unused]] const long head_offd = (long)b * T * p.ldd + h * HD;                              \
    const int ntiles = (T + 31) / 32;                                                                  \
    const int t_begin = split * p.tiles_per_split;                                                     \
    const int t_end = min(ntiles, t_begin + p.tiles_per_split);                                        \
    unsigned dtab[8];                                                                                  \
    if (vdims<HD, BF>()) make_dtab_v<HD>(dtab, l31, half); else make_dtab<HD>(dtab, l31, half);        \
    (void)NB;

// ------------------------------------------------------------------------------------------------
template <int HD, int BF>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const AttnParams p) {
    ATTN_COMMON();
    const int q_row = bx_ * 128 + wave * 32 + l31;
    const TileLoader<HD, BF> k_ld(p.k, (long)b * T * p.ldq, T, p.ldq, h * HD, wave, lane);
    const TileLoader<HD, BF> v_ld(p.v, (long)b * T * p.ldq, T, p.ldq, h * HD, wave, lane);

    float qreg[HD / 2];
    // scores are kept in LOG2 units (q pre-scaled by scale * log2 e): the softmax exponentials are bare v_exp_f32
    const float scale2 = p.scale * 1.4426950408889634f;
    load_frag_any<HD, BF>(qreg, p.q, head_offq + (long)(q_row < T ? q_row : T - 1) * p.ldq, half, scale2);
    f32x16 oacc[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[blk][r] = 0.f;
    float m = -INFINITY, l = 0.f;
    const long drop_row = ((long)(b * p.nh + h) * T + q_row) * ((T + 3) >> 2);  // key quads before this row (common.h: Ds6gKeep4Base)

    auto step = [&](const float* Kcp, const float* Vcp, const float* Kn, const float* Vn, int kt, bool more) {
        if (more) {
            k_ld.issue(Kn, (kt + 1) * 32, wave);
            v_ld.issue(Vn, (kt + 1) * 32, wave);
        }
        const unsigned Kc = opaque_tile(Kcp), Vc = opaque_tile(Vcp);
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        mma_rows<HD, BF>(s, Kc, qreg, l31, half);
        if (BF == 4) {  // the bf16-stored q fragment is unscaled: scale the scores (exact in fp32)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] *= scale2;
        }
        const int key0 = kt * 32;
#ifndef ATTN_ABLATE_SOFTMAX
        float mloc = -INFINITY;
        if (key0 + 32 > T) {  // only the last key tile has a tail (wave-uniform)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (key0 + krow16(r, half) >= T) s[r] = -INFINITY;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[r]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float mnew = fmaxf(m, mloc);
        const float alpha = __builtin_amdgcn_exp2f(m - mnew);
        float lsum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __builtin_amdgcn_exp2f(s[r] - mnew);
            lsum += s[r];
        }
        l = l * alpha + lsum;
        m = mnew;
        if constexpr (quad16<HD, BF>()) {  // this lane's accumulators belong to queries l16 and 16 + l16 (see quad16)
            const float a_lo = __shfl(alpha, lane & 15, 64), a_hi = __shfl(alpha, 16 + (lane & 15), 64);
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[0][r] *= ((r >> 2) & 1) ? a_hi : a_lo;
        } else {
#pragma unroll
            for (int blk = 0; blk < NB; ++blk)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[blk][r] *= alpha;
        }
#else
        l += s[0];  // timing experiment only: no softmax arithmetic between the two products
#endif
        if (p.thr) {  // attn_drop: this lane's 16 keys are four key quads (registers 4 g .. 4 g + 3), one hash each (common.h)
            const Ds6gKeep4Base kb(p.seed, seed_off_ + (uint64_t)(drop_row + (key0 >> 2) + half));
            const uint32_t thi = p.thr & 0xffff0000u;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint32_t w0, w1;
                kb.words((uint32_t)(2 * g), w0, w1);
                s[4 * g] *= ds6g_keep4<0>(w0, w1, thi) ? p.dscale : 0.f;
                s[4 * g + 1] *= ds6g_keep4<1>(w0, w1, thi) ? p.dscale : 0.f;
                s[4 * g + 2] *= ds6g_keep4<2>(w0, w1, thi) ? p.dscale : 0.f;
                s[4 * g + 3] *= ds6g_keep4<3>(w0, w1, thi) ? p.dscale : 0.f;
            }
        }
        mma_dims<HD, BF>(oacc, Vc, s, dtab);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    if (t_begin < t_end) {
        k_ld.issue(Xa0, t_begin * 32, wave);
        v_ld.issue(Xb0, t_begin * 32, wave);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = t_begin; kt < t_end; kt += 2) {
        step(Xa0, Xb0, Xa1, Xb1, kt, kt + 1 < t_end);
        if (kt + 1 < t_end) step(Xa1, Xb1, Xa0, Xb0, kt + 1, kt + 2 < t_end);
    }
    const float ltot = l + __shfl_xor(l, 32, 64);
    const long stat = (long)(b * p.nh + h) * T + q_row;
    if (p.splits == 1) {
        if (p.out16) store_rows16<HD, BF>(oacc, reinterpret_cast<__bf16*>(p.o) + head_off, p.ld, q_row, T, half, 1.0f / ltot);
        else store_rows<HD, BF>(oacc, p.o + head_off, p.ld, q_row, T, half, 1.0f / ltot);
        if (half == 0 && q_row < T) p.lse[stat] = (m + __log2f(ltot)) * 0.6931471805599453f;  // natural-log units in HBM
    } else {
        store_rows<HD, BF>(oacc, p.o + (size_t)split * p.slab + head_off, p.ld, q_row, T, half, 1.0f);
        if (half == 0 && q_row < T) {
            float* mlp = p.ml + ((size_t)split * p.B * p.nh * T + stat) * 2;
            mlp[0] = m;
            mlp[1] = ltot;
        }
    }
}

// o = sum_s o_s exp(m_s - m) / sum_s l_s exp(m_s - m); lse = m + log(...)
__global__ __launch_bounds__(256) void attn_fwd_merge_kernel(const float* __restrict__ part, const float* __restrict__ ml,
                                                             float* __restrict__ o, float* __restrict__ lse, int B, int T,
                                                             int nh, int hd, int ld, int splits, size_t slab, int out16) {
    const int c4n = nh * hd / 4;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * T * c4n) return;
    const int c4 = (int)(i % c4n);
    const long row = i / c4n;  // b*T + t
    const int h = (c4 * 4) / hd;
    const int b = (int)(row / T), t = (int)(row % T);
    const size_t stat = ((size_t)(b * nh + h) * T + t);
    const size_t sstride = (size_t)B * nh * T;
    float m = -INFINITY;
    for (int s = 0; s < splits; ++s) m = fmaxf(m, ml[(s * sstride + stat) * 2]);
    float l = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < splits; ++s) {
        const float ms = ml[(s * sstride + stat) * 2];
        const float w = ms == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(ms - m);  // running maxima are in log2 units
        l += ml[(s * sstride + stat) * 2 + 1] * w;
        acc += w * *reinterpret_cast<const f32x4*>(part + (size_t)s * slab + row * ld + c4 * 4);
    }
    acc = acc * (1.0f / l);
    if (out16) {
        typedef __bf16 bf16x4_ __attribute__((ext_vector_type(4)));
        *reinterpret_cast<bf16x4_*>(reinterpret_cast<__bf16*>(o) + row * ld + c4 * 4) =
            bf16x4_{(__bf16)acc[0], (__bf16)acc[1], (__bf16)acc[2], (__bf16)acc[3]};
    } else {
        *reinterpret_cast<f32x4*>(o + row * ld + c4 * 4) = acc;
    }
    if ((c4 * 4) % hd == 0) lse[stat] = (m + __log2f(l)) * 0.6931471805599453f;
}

// ------------------------------------------------------------------------------------------------
// dQ (and delta): one wave = 32 queries, loop over (a split of) the key tiles
template <int HD, int BF>
__global__ __launch_bounds__(256, (HD >= 128 ? 1 : 2)) void attn_bwd_dq_kernel(const AttnParams p) {
    ATTN_COMMON();
    const int q_row = bx_ * 128 + wave * 32 + l31;
    const bool q_ok = q_row < T;
    const TileLoader<HD, BF> k_ld(p.k, (long)b * T * p.ldq, T, p.ldq, h * HD, wave, lane);
    const TileLoader<HD, BF> v_ld(p.v, (long)b * T * p.ldq, T, p.ldq, h * HD, wave, lane);

    float qreg[HD / 2], doreg[HD / 2];
    float delta = 0.f;
    {
        const long ro = head_off + (long)(q_ok ? q_row : T - 1) * p.ld;
        load_frag<HD>(qreg, p.q, head_offq + (long)(q_ok ? q_row : T - 1) * p.ldq, half, p.scale);
        load_frag<HD>(doreg, p.d_o, ro, half, q_ok ? 1.f : 0.f);
#pragma unroll
        for (int j = 0; j < HD / 8; ++j) {
            const f32x4 ov = *reinterpret_cast<const f32x4*>(p.o + ro + half * (HD / 2) + 4 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e) delta += doreg[4 * j + e] * ov[e];
        }
        delta += __shfl_xor(delta, 32, 64);
    }
    const long stat_idx = (long)(b * p.nh + h) * T + q_row;
    const float lse = q_ok ? p.lse[stat_idx] : INFINITY;
    if (half == 0 && q_ok && split == 0) p.delta[stat_idx] = delta;

    f32x16 dq[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[blk][r] = 0.f;
    const long drop_row = ((long)(b * p.nh + h) * T + q_row) * ((T + 3) >> 2);  // key quads before this row (common.h: Ds6gKeep4Base)

    auto step = [&](const float* Kcp, const float* Vcp, const float* Kn, const float* Vn, int kt, bool more) {
        if (more) {
            k_ld.issue(Kn, (kt + 1) * 32, wave);
            v_ld.issue(Vn, (kt + 1) * 32, wave);
        }
        const unsigned Kc = opaque_tile(Kcp), Vc = opaque_tile(Vcp);
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        mma_rows<HD, BF>(s, Kc, qreg, l31, half);
        mma_rows<HD, BF>(dp, Vc, doreg, l31, half);
        const int key0 = kt * 32;
        if (p.thr) {
            const Ds6gKeep4Base kb(p.seed, seed_off_ + (uint64_t)(drop_row + (key0 >> 2) + half));
            const uint32_t thi = p.thr & 0xffff0000u;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint32_t w0, w1;
                kb.words((uint32_t)(2 * g), w0, w1);
                dp[4 * g] *= ds6g_keep4<0>(w0, w1, thi) ? p.dscale : 0.f;
                dp[4 * g + 1] *= ds6g_keep4<1>(w0, w1, thi) ? p.dscale : 0.f;
                dp[4 * g + 2] *= ds6g_keep4<2>(w0, w1, thi) ? p.dscale : 0.f;
                dp[4 * g + 3] *= ds6g_keep4<3>(w0, w1, thi) ? p.dscale : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + krow16(r, half);
            const float pr = (key < T) ? __expf(s[r] - lse) : 0.f;
            s[r] = pr * (dp[r] - delta) * p.scale;  // dS (scaled)
        }
        mma_dims<HD, BF>(dq, Kc, s, dtab);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    if (t_begin < t_end) {
        k_ld.issue(Xa0, t_begin * 32, wave);
        v_ld.issue(Xb0, t_begin * 32, wave);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = t_begin; kt < t_end; kt += 2) {
        step(Xa0, Xb0, Xa1, Xb1, kt, kt + 1 < t_end);
        if (kt + 1 < t_end) step(Xa1, Xb1, Xa0, Xb0, kt + 1, kt + 2 < t_end);
    }
    if (p.splits == 1 && p.out16) store_rows16<HD, BF>(dq, reinterpret_cast<__bf16*>(p.dq) + head_offd, p.ldd, q_row, T, half, 1.0f);
    else if (p.splits == 1) store_rows<HD, BF>(dq, p.dq + head_offd, p.ldd, q_row, T, half, 1.0f);
    else store_rows<HD, BF>(dq, p.dq + (size_t)split * p.slab + head_off, p.ld, q_row, T, half, 1.0f);
}

// dK, dV: one wave = 32 keys (key on the lane), loop over (a split of) the query tiles.
// PART 0 = both (4 products per tile); 1 = dV only (S, dV); 2 = dK only (S, dP, dK).  At HD = 128 the fused form
// needs > 512 registers (K, V fragments + two accumulator sets), so it runs as PART 1 + PART 2.
// HAND 1: the kernel also writes its dS tiles (and, PART 2, the dropped probabilities) to HBM in accumulator order -
// 16 fully coalesced 256-B stores per wave and tile.  attn_bwd_dq2_kernel then forms dQ = dS K from them (one product
// instead of the three of attn_bwd_dq_kernel, after a 32 x 32 transpose through LDS: the two kernels hold the score tile
// in opposite orientations), and at HD = 128 attn_bwd_dv2_kernel forms dV = P^T dO without recomputing S: the whole
// backward is the minimal 5 products instead of 7 (8 at HD = 128).
template <int HD, int PART, int BF, int HAND = 0>
__global__ __launch_bounds__(256, (HD <= 16 ? 3 : (HD <= 64 ? 2 : (PART == 1 ? 2 : 1)))) void attn_bwd_dkv_kernel(const AttnParams p) {
    constexpr bool DO_DV = PART != 2, DO_DK = PART != 1;
    ATTN_COMMON();
    __shared__ float lse_s[2][32];
    __shared__ float delta_s[2][32];
    const int key = bx_ * 128 + wave * 32 + l31;
    const bool key_ok = key < T;
    const TileLoader<HD, BF> q_ld(p.q, (long)b * T * p.ldq, T, p.ldq, h * HD, wave, lane);
    const TileLoader<HD, BF> do_ld(p.d_o, (long)b * T * p.ld, T, p.ld, h * HD, wave, lane);

    float kreg[HD / 2], vreg[HD / 2];
    {
        const long ro = head_offq + (long)(key_ok ? key : T - 1) * p.ldq;
        load_frag_any<HD, BF>(kreg, p.k, ro, half, 1.f);
        if (DO_DK) load_frag_any<HD, BF>(vreg, p.v, ro, half, 1.f);
    }
    f32x16 dk[NB], dv[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[blk][r] = 0.f; dv[blk][r] = 0.f; }
    const long stat_base = (long)(b * p.nh + h) * T;

    auto stats = [&](int qt, int buf) {  // per-query lse / delta of tile qt -> LDS (first 32 threads)
        if (tid < 32) {
            const int qn = qt * 32 + tid;
            lse_s[buf][tid] = qn < T ? p.lse[stat_base + qn] * 1.4426950408889634f : INFINITY;  // log2 units: P = exp2(S scale2 - lse2)
            delta_s[buf][tid] = qn < T ? p.delta[stat_base + qn] : 0.f;
        }
    };
    ATTN_CLK_DECL();
    auto step = [&](const float* Qcp, const float* Ocp, const float* Qn, const float* On, int qt, int buf, bool more) {
        ATTN_CLK_START();
        if (more) {
            q_ld.issue(Qn, (qt + 1) * 32, wave);
            do_ld.issue(On, (qt + 1) * 32, wave);
            stats(qt + 1, buf ^ 1);
        }
        ATTN_CLK(0);   // DMA issue
        const unsigned Qc = opaque_tile(Qcp), Oc = opaque_tile(Ocp);
        // S[q][key], dP[q][key]: query rows in registers, key on the lane
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        mma_rows<HD, BF>(s, Qc, kreg, l31, half);
        __builtin_amdgcn_sched_barrier(0);
        ATTN_CLK(1);   // S product
        if (DO_DK) mma_rows<HD, BF>(dp, Oc, vreg, l31, half);
        __builtin_amdgcn_sched_barrier(0);
        ATTN_CLK(2);   // dP product
        const int q0 = qt * 32;
        const float scale2 = p.scale * 1.4426950408889634f;
        // P = exp(S - lse), dS = P (dP - delta) scale; with attn_drop both carry the keep mask m (0 or 1 / (1 - p)) of their
        // element: dS = P (m dP - delta) scale, dropped P = m P.  The counters of this lane's 16 queries are one 64-bit base
        // (query q0 + 4 half, this key) plus krow16 * T: no 64-bit arithmetic and no branch per element.
        if (p.thr) {
            // attn_drop (common.h: four decisions per hash, quads run along the KEY axis): the four lanes that hold one key quad
            // need the same 16 hashes (one per query row of the tile) and pick the 16-bit half of their own key from each.
            // Lane j of the quad computes the four rows 4 g + j, the words go round by DPP quad broadcasts.
            const int Tq4 = (T + 3) >> 2, jq = lane & 3;
            const Ds6gKeep4Base kb(p.seed, seed_off_ + (uint64_t)((stat_base + q0 + 4 * half + jq) * (long)Tq4 + (key >> 2)));
            const uint32_t thi = p.thr & 0xffff0000u;
            uint32_t w0[4], w1[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) kb.words((uint32_t)(8 * g * Tq4), w0[g], w1[g]);
            const bool hi_word = (lane & 2) != 0;
            const uint32_t shl = (lane & 1) ? 0u : 16u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ql = krow16(r, half);
                const float pr = __builtin_amdgcn_exp2f(fmaf(s[r], scale2, -lse_s[buf][ql]));  // lse = +inf for q >= T -> 0
                constexpr int QP[4] = {0x00, 0x55, 0xaa, 0xff};  // quad_perm broadcasts of lane 0 / 1 / 2 / 3 of the quad
                uint32_t a, c;
                switch (r & 3) {
                    case 0: a = __builtin_amdgcn_mov_dpp((int)w0[r >> 2], QP[0], 0xf, 0xf, true); c = __builtin_amdgcn_mov_dpp((int)w1[r >> 2], QP[0], 0xf, 0xf, true); break;
                    case 1: a = __builtin_amdgcn_mov_dpp((int)w0[r >> 2], QP[1], 0xf, 0xf, true); c = __builtin_amdgcn_mov_dpp((int)w1[r >> 2], QP[1], 0xf, 0xf, true); break;
                    case 2: a = __builtin_amdgcn_mov_dpp((int)w0[r >> 2], QP[2], 0xf, 0xf, true); c = __builtin_amdgcn_mov_dpp((int)w1[r >> 2], QP[2], 0xf, 0xf, true); break;
                    default: a = __builtin_amdgcn_mov_dpp((int)w0[r >> 2], QP[3], 0xf, 0xf, true); c = __builtin_amdgcn_mov_dpp((int)w1[r >> 2], QP[3], 0xf, 0xf, true); break;
                }
                const float m = (((hi_word ? c : a) << shl) >= thi) ? p.dscale : 0.f;
                dp[r] = pr * (dp[r] * m - delta_s[buf][ql]) * p.scale;
                s[r] = pr * m;  // dropped probabilities
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ql = krow16(r, half);
                const float pr = __builtin_amdgcn_exp2f(fmaf(s[r], scale2, -lse_s[buf][ql]));
                dp[r] = pr * (dp[r] - delta_s[buf][ql]) * p.scale;
                s[r] = pr;
            }
        }
        ATTN_CLK(3);   // elementwise (exp, dropout hash, dS)
        if (HAND) {  // tile image: [reg / 4][lane][reg % 4] - four 1-KiB stores per wave and tensor
            const size_t tile = ((((size_t)(b * p.nh + h) * p.nkg + qt) * p.nkg) + (bx_ * 4 + wave)) * 1024 + lane * 4;
            if (BF == 4) {
                // bf16-stored path: attn_bwd_dq2_kernel rounds dS to bf16 on its way into the MFMA anyway, so the tile is handed
                // over AS bf16 - [reg / 8][lane][reg % 8], two 1-KiB stores, half the HBM bytes in both kernels (dQ from dS
                // was HBM-bound on these tiles: 201 MB per layer at bs = 12), bit-identical dQ
                bf16x8* ds16 = reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(p.hs) + (tile - lane * 4) + lane * 8);
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    bf16x8 v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = key_ok ? (__bf16)dp[8 * g + e] : (__bf16)0.f;
                    ds16[g * 64] = v;
                }
            } else {
            f32x4* ds_out = reinterpret_cast<f32x4*>(p.hs + tile);
#pragma unroll
            for (int g = 0; g < 4; ++g)
                ds_out[g * 64] = key_ok ? f32x4{dp[4 * g], dp[4 * g + 1], dp[4 * g + 2], dp[4 * g + 3]} : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (PART == 2) {
                f32x4* p_out = reinterpret_cast<f32x4*>(p.hp + tile);
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    p_out[g * 64] = key_ok ? f32x4{s[4 * g], s[4 * g + 1], s[4 * g + 2], s[4 * g + 3]} : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // phase fences keep the LDS-read prefetch of one product from
        ATTN_CLK(4);   // hand-over stores issued
        if (DO_DV) mma_dims<HD, BF>(dv, Oc, s, dtab);  // overlapping the live registers of the next
        __builtin_amdgcn_sched_barrier(0);
        ATTN_CLK(5);   // dV product
        if (DO_DK) mma_dims<HD, BF>(dk, Qc, dp, dtab);
        __builtin_amdgcn_sched_barrier(0);
        ATTN_CLK(6);   // dK product
        // the next tile's DMA is older than this step's hand-over stores and vmcnt retires in order: wait for the DMA
        // only, the 4 (8) stores drain under the next step
        if (HAND) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((BF == 4 ? 2 : 4) + (PART == 2 ? 4 : 0)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ATTN_CLK(7);   // wait for the next tile's DMA
        __syncthreads();
        ATTN_CLK(8);   // barrier
#ifdef DS6G_ATTN_CLOCKS
        if (clk_on_) g_attn_clk[15] += 1;   // steps counted
#endif
    };

    if (t_begin < t_end) {
        q_ld.issue(Xa0, t_begin * 32, wave);
        do_ld.issue(Xb0, t_begin * 32, wave);
        stats(t_begin, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int qt = t_begin; qt < t_end; qt += 2) {
        step(Xa0, Xb0, Xa1, Xb1, qt, 0, qt + 1 < t_end);
        if (qt + 1 < t_end) step(Xa1, Xb1, Xa0, Xb0, qt + 1, 1, qt + 2 < t_end);
    }
    if (p.splits == 1 && p.out16) {
        if (DO_DK) store_rows16<HD, BF>(dk, reinterpret_cast<__bf16*>(p.dk) + head_offd, p.ldd, key, T, half, 1.0f);
        if (DO_DV) store_rows16<HD, BF>(dv, reinterpret_cast<__bf16*>(p.dv) + head_offd, p.ldd, key, T, half, 1.0f);
    } else if (p.splits == 1) {
        if (DO_DK) store_rows<HD, BF>(dk, p.dk + head_offd, p.ldd, key, T, half, 1.0f);
        if (DO_DV) store_rows<HD, BF>(dv, p.dv + head_offd, p.ldd, key, T, half, 1.0f);
    } else {
        if (DO_DK) store_rows<HD, BF>(dk, p.dk + (size_t)split * p.slab + head_off, p.ld, key, T, half, 1.0f);
        if (DO_DV) store_rows<HD, BF>(dv, p.dv + (size_t)split * p.slab + head_off, p.ld, key, T, half, 1.0f);
    }
}

// delta[b][h][t] = sum_d dO[t][h*hd + d] * O[t][h*hd + d]  (hand-over path: the dK/dV kernel runs first and needs it)
__global__ __launch_bounds__(256) void attn_delta_kernel(const float* __restrict__ o, const float* __restrict__ d_o,
                                                         float* __restrict__ delta, int B, int T, int nh, int hd, int ld,
                                                         int o16) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * T * nh) return;
    const int h = (int)(i % nh);
    const long row = i / nh;
    const float* a = o + row * ld + h * hd;
    const float* g = d_o + row * ld + h * hd;
    float acc = 0.f;
    if (o16 == 3) {       // o and dO both bf16
        typedef __bf16 bf16x4_ __attribute__((ext_vector_type(4)));
        const __bf16* a16 = reinterpret_cast<const __bf16*>(o) + row * ld + h * hd;
        const __bf16* g16 = reinterpret_cast<const __bf16*>(d_o) + row * ld + h * hd;
        for (int d = 0; d < hd; d += 4) {
            const bf16x4_ x = *reinterpret_cast<const bf16x4_*>(a16 + d), y = *reinterpret_cast<const bf16x4_*>(g16 + d);
            acc += (float)x[0] * (float)y[0] + (float)x[1] * (float)y[1] + (float)x[2] * (float)y[2] + (float)x[3] * (float)y[3];
        }
    } else if (o16) {
        typedef __bf16 bf16x4_ __attribute__((ext_vector_type(4)));
        const __bf16* a16 = reinterpret_cast<const __bf16*>(o) + row * ld + h * hd;
        for (int d = 0; d < hd; d += 4) {
            const bf16x4_ x = *reinterpret_cast<const bf16x4_*>(a16 + d);
            const f32x4 y = *reinterpret_cast<const f32x4*>(g + d);
            acc += (float)x[0] * y[0] + (float)x[1] * y[1] + (float)x[2] * y[2] + (float)x[3] * y[3];
        }
    } else
    for (int d = 0; d < hd; d += 4) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(a + d), y = *reinterpret_cast<const f32x4*>(g + d);
        acc += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
    }
    const int b = (int)(row / T), t = (int)(row % T);
    delta[((long)b * nh + h) * T + t] = acc;
}

// dQ from the dS tiles of attn_bwd_dkv_kernel<.., HAND = 1>: one wave = 32 queries (= one query tile), loop over (a split
// of) the key tiles; the stored tile holds dS[query krow16(r, half)][key lane&31], the B operand of dQ^T += K^T dS^T
// wants dS[query lane&31][key krow16(r, half)]: a 32 x 32 transpose through a wave-private LDS patch (row stride 33).
template <int HD, int BF>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq2_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) float Xa0[32 * HD];
    __shared__ __attribute__((aligned(16))) float Xa1[32 * HD];
    __shared__ float Tr[4][32 * 33];
    ATTN_GEOM();
    const int q_row = bx_ * 128 + wave * 32 + l31;
    const int qt = bx_ * 4 + wave;
    const TileLoader<HD, BF> k_ld(p.k, (long)b * T * p.ldq, T, p.ldq, h * HD, wave, lane);
    f32x16 dq[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[blk][r] = 0.f;
    const float* tiles = p.hs + (((size_t)(b * p.nh + h) * p.nkg + qt) * p.nkg) * 1024 + lane * 4;
    const bool have = qt < ntiles;  // wave-uniform: query tiles beyond T were never written
    float f[16], fn[16];
    auto fetch = [&](float* dst, int kt) {
        if (BF == 4) {   // bf16 tiles [reg / 8][lane][reg % 8] (see attn_bwd_dkv_kernel)
            const __bf16* t16 = reinterpret_cast<const __bf16*>(p.hs) + ((((size_t)(b * p.nh + h) * p.nkg + qt) * p.nkg) + kt) * 1024 + lane * 8;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                bf16x8 v;
                if (have) v = *reinterpret_cast<const bf16x8*>(t16 + g * 512);
#pragma unroll
                for (int e = 0; e < 8; ++e) dst[8 * g + e] = have ? (float)v[e] : 0.f;
            }
            return;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = have ? *reinterpret_cast<const f32x4*>(tiles + (size_t)kt * 1024 + g * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
            dst[4 * g] = v[0]; dst[4 * g + 1] = v[1]; dst[4 * g + 2] = v[2]; dst[4 * g + 3] = v[3];
        }
    };
    float* tw = Tr[wave];
    auto step = [&](const float* Kcp, const float* Kn, int kt, bool more) {
        if (more) {
            k_ld.issue(Kn, (kt + 1) * 32, wave);
            fetch(fn, kt + 1);
        }
        const unsigned Kc = opaque_tile(Kcp);
#pragma unroll
        for (int r = 0; r < 16; ++r) tw[krow16(r, half) * 33 + l31] = f[r];
        __builtin_amdgcn_wave_barrier();
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = tw[l31 * 33 + krow16(r, half)];
        __builtin_amdgcn_wave_barrier();
        mma_dims<HD, BF>(dq, Kc, s, dtab);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (more) {
#pragma unroll
            for (int r = 0; r < 16; ++r) f[r] = fn[r];
        }
    };
    if (t_begin < t_end) {
        k_ld.issue(Xa0, t_begin * 32, wave);
        fetch(f, t_begin);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = t_begin; kt < t_end; kt += 2) {
        step(Xa0, Xa1, kt, kt + 1 < t_end);
        if (kt + 1 < t_end) step(Xa1, Xa0, kt + 1, kt + 2 < t_end);
    }
    if (p.splits == 1 && p.out16) store_rows16<HD, BF>(dq, reinterpret_cast<__bf16*>(p.dq) + head_offd, p.ldd, q_row, T, half, 1.0f);
    else if (p.splits == 1) store_rows<HD, BF>(dq, p.dq + head_offd, p.ldd, q_row, T, half, 1.0f);
    else store_rows<HD, BF>(dq, p.dq + (size_t)split * p.slab + head_off, p.ld, q_row, T, half, 1.0f);
}

// dV from the dropped-probability tiles of attn_bwd_dkv_kernel<HD, 2, .., HAND = 1> (HD = 128, where the fused dK/dV form
// does not fit the register file): same orientation as the producer, the tile is the B operand as stored.
template <int HD, int BF>
__global__ __launch_bounds__(256, 2) void attn_bwd_dv2_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) float Xb0[32 * HD];
    __shared__ __attribute__((aligned(16))) float Xb1[32 * HD];
    ATTN_GEOM();
    const int key = bx_ * 128 + wave * 32 + l31;
    const TileLoader<HD, BF> do_ld(p.d_o, (long)b * T * p.ld, T, p.ld, h * HD, wave, lane);
    f32x16 dv[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) dv[blk][r] = 0.f;
    const float* tiles = p.hp + ((size_t)(b * p.nh + h) * p.nkg * p.nkg + (bx_ * 4 + wave)) * 1024 + lane * 4;
    const size_t qstride = (size_t)p.nkg * 1024;
    f32x16 s, sn;
    auto fetch = [&](f32x16& dst, int qt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(tiles + (size_t)qt * qstride + g * 256);
            dst[4 * g] = v[0]; dst[4 * g + 1] = v[1]; dst[4 * g + 2] = v[2]; dst[4 * g + 3] = v[3];
        }
    };
    auto step = [&](const float* Ocp, const float* On, int qt, bool more) {
        if (more) {
            do_ld.issue(On, (qt + 1) * 32, wave);
            fetch(sn, qt + 1);
        }
        const unsigned Oc = opaque_tile(Ocp);
        mma_dims<HD, BF>(dv, Oc, s, dtab);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (more) s = sn;
    };
    if (t_begin < t_end) {
        do_ld.issue(Xb0, t_begin * 32, wave);
        fetch(s, t_begin);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int qt = t_begin; qt < t_end; qt += 2) {
        step(Xb0, Xb1, qt, qt + 1 < t_end);
        if (qt + 1 < t_end) step(Xb1, Xb0, qt + 1, qt + 2 < t_end);
    }
    if (p.splits == 1 && p.out16) store_rows16<HD, BF>(dv, reinterpret_cast<__bf16*>(p.dv) + head_offd, p.ldd, key, T, half, 1.0f);
    else if (p.splits == 1) store_rows<HD, BF>(dv, p.dv + head_offd, p.ldd, key, T, half, 1.0f);
    else store_rows<HD, BF>(dv, p.dv + (size_t)split * p.slab + head_off, p.ld, key, T, half, 1.0f);
}

// out[row][0..cols) = sum_s part[s][row][0..cols)   (dq / dk / dv split slabs; rows of stride ld_in -> ld_out).
// blockIdx.y picks the (part, out) pair, so dK and dV are reduced by one launch.
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ part0, float* __restrict__ out0,
                                                       const float* __restrict__ part1, float* __restrict__ out1,
                                                       long n4, int cols4, int ld_in, int ld_out, int splits, size_t slab,
                                                       int out16) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float* part = blockIdx.y ? part1 : part0;
    float* out = blockIdx.y ? out1 : out0;
    const long row = i / cols4;
    const int c = (int)(i - row * cols4) * 4;
    const float* src = part + row * ld_in + c;
    f32x4 s = *reinterpret_cast<const f32x4*>(src);
    for (int k = 1; k < splits; ++k) s += *reinterpret_cast<const f32x4*>(src + (size_t)k * slab);
    if (out16) {
        typedef __bf16 bf16x4_ __attribute__((ext_vector_type(4)));
        *reinterpret_cast<bf16x4_*>(reinterpret_cast<__bf16*>(out) + row * ld_out + c) =
            bf16x4_{(__bf16)s[0], (__bf16)s[1], (__bf16)s[2], (__bf16)s[3]};
    } else {
        *reinterpret_cast<f32x4*>(out + row * ld_out + c) = s;
    }
}

// number of loop splits: fills the chip evenly.  cap = workgroups resident at once (256 CUs x per-CU residency)
// pen: what one more split costs in loop trips (its share of the merge / slab-sum kernel and of the prologue).  0.35 for the
// fp32-storage kernels; 4 on the bf16-stored path, where a tile step is 3 - 4x shorter while the merge kernels cost the same
// (measured: bf16 step 451.9 -> 460.3 samples/s at 4, 459.5 at 8; fp32 184.5 at 0.35, 183.4 at 4).  DS6G_ATTN_SPLIT_PENALTY overrides.
int pick_splits(long base_blocks, int ntiles, int per_cu, int max_splits, double pen_default = 0.35) {
    if (g_ds6g_attn_percu > 0) per_cu = g_ds6g_attn_percu;  // timing experiments (ds6g_set_debug_flags bits 20-23)
    const long cap = 256L * per_cu;
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= max_splits && s <= ntiles; ++s) {
        const int tps = cdiv(ntiles, s);
        const int eff = cdiv(ntiles, tps);  // splits actually used
        const long rounds = cdiv(base_blocks * eff, cap);
        static const double pen_env = [] { const char* e = getenv("DS6G_ATTN_SPLIT_PENALTY"); return e ? atof(e) : -1.0; }();
        const double pen = pen_env >= 0.0 ? pen_env : pen_default;
        const double cost = (double)rounds * tps + pen * eff;  // loop trips per CU slot + a merge/prologue penalty
        if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
    }
    return best;
}

template <int KIND, int BF>
int launch_hd_bf(const AttnParams& p, int hd, dim3 grid, hipStream_t st) {
#define ATTN_CASE(HDV)                                                                                        \
    case HDV:                                                                                                 \
        if (KIND == 0) hipLaunchKernelGGL((attn_fwd_kernel<HDV, BF>), grid, dim3(256), 0, st, p);             \
        if (KIND == 1) hipLaunchKernelGGL((attn_bwd_dq_kernel<HDV, BF>), grid, dim3(256), 0, st, p);          \
        if (KIND == 2) {                                                                                      \
            if (HDV >= 128) {                                                                                 \
                hipLaunchKernelGGL((attn_bwd_dkv_kernel<HDV, 1, BF>), grid, dim3(256), 0, st, p);             \
                hipLaunchKernelGGL((attn_bwd_dkv_kernel<HDV, 2, BF>), grid, dim3(256), 0, st, p);             \
            } else {                                                                                          \
                hipLaunchKernelGGL((attn_bwd_dkv_kernel<(HDV >= 128 ? 64 : HDV), 0, BF>), grid, dim3(256), 0, st, p); \
            }                                                                                                 \
        }                                                                                                     \
        if (KIND == 3) {                                                                                      \
            if (HDV >= 128) hipLaunchKernelGGL((attn_bwd_dkv_kernel<HDV, 2, BF, 1>), grid, dim3(256), 0, st, p); \
            else hipLaunchKernelGGL((attn_bwd_dkv_kernel<(HDV >= 128 ? 64 : HDV), 0, BF, 1>), grid, dim3(256), 0, st, p); \
        }                                                                                                     \
        if (KIND == 4) hipLaunchKernelGGL((attn_bwd_dv2_kernel<(HDV >= 128 ? HDV : 128), BF>), grid, dim3(256), 0, st, p); \
        if (KIND == 5) hipLaunchKernelGGL((attn_bwd_dq2_kernel<HDV, BF>), grid, dim3(256), 0, st, p);         \
        if (KIND == 6) hipLaunchKernelGGL((attn_bwd_dkv_kernel<(BF == 0 ? HDV : 16), 0, (BF == 0 ? 0 : BF), 1>), grid, dim3(256), 0, st, p); \
        break;
    switch (hd) {
        ATTN_CASE(16)
        ATTN_CASE(32)
        ATTN_CASE(64)
        ATTN_CASE(128)
        default:
            fprintf(stderr, "[ds6g] attention head dim %d not supported (16/32/64/128)\n", hd);
            return DS6G_ERR_ARG;
    }
#undef ATTN_CASE
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

// bf16-stored q / k / v / dO (BF == 4): forward, dK/dV with hand-over, dV from P (HD = 128), dQ from dS
template <int KIND>
int launch_hd_st(const AttnParams& p, int hd, dim3 grid, hipStream_t st) {
#define ATTN_CASE16(HDV)                                                                                      \
    case HDV:                                                                                                 \
        if constexpr (KIND == 0) hipLaunchKernelGGL((attn_fwd_kernel<HDV, 4>), grid, dim3(256), 0, st, p);    \
        if constexpr (KIND == 3) {                                                                            \
            if constexpr (HDV >= 128) hipLaunchKernelGGL((attn_bwd_dkv_kernel<HDV, 2, 4, 1>), grid, dim3(256), 0, st, p); \
            else hipLaunchKernelGGL((attn_bwd_dkv_kernel<HDV, 0, 4, 1>), grid, dim3(256), 0, st, p);          \
        }                                                                                                     \
        if constexpr (KIND == 4 && HDV >= 128) hipLaunchKernelGGL((attn_bwd_dv2_kernel<HDV, 4>), grid, dim3(256), 0, st, p); \
        if constexpr (KIND == 5) hipLaunchKernelGGL((attn_bwd_dq2_kernel<HDV, 4>), grid, dim3(256), 0, st, p); \
        if constexpr (KIND == 6 && HDV >= 128) hipLaunchKernelGGL((attn_bwd_dkv_kernel<HDV, 0, 4, 1>), grid, dim3(256), 0, st, p); \
        break;
    switch (hd) {
        ATTN_CASE16(16)
        ATTN_CASE16(32)
        ATTN_CASE16(64)
        ATTN_CASE16(128)
        default:
            fprintf(stderr, "[ds6g] attention head dim %d not supported (16/32/64/128)\n", hd);
            return DS6G_ERR_ARG;
    }
#undef ATTN_CASE16
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

template <int KIND>
int launch_hd(const AttnParams& p, int hd, dim3 grid, hipStream_t st) {
    if (p.in16) {
        if constexpr (KIND == 0 || KIND == 3 || KIND == 4 || KIND == 5 || KIND == 6) return launch_hd_st<KIND>(p, hd, grid, st);
        else return DS6G_ERR_ARG;
    }
    if (g_ds6g_bf16 == 3) return launch_hd_bf<KIND, 3>(p, hd, grid, st);
    if (g_ds6g_bf16 == 2) return launch_hd_bf<KIND, 2>(p, hd, grid, st);
    return g_ds6g_bf16 ? launch_hd_bf<KIND, 1>(p, hd, grid, st) : launch_hd_bf<KIND, 0>(p, hd, grid, st);
}

}  // namespace

extern "C" {

// scratch that lets the split paths use up to 8 splits (forward: partial outputs + (m,l); backward: dk + dv slabs)
static size_t handover_bytes(int B, int T, int nh, int hd) {
    const size_t nkg = (size_t)cdiv(T, 128) * 4;
    return (size_t)B * nh * nkg * nkg * 1024 * sizeof(float) * (hd >= 128 ? 2 : 1);
}
// ... plus, for the backward, the dS (and at hd = 128 the dropped-P) tiles handed from the dK/dV kernel to the dQ / dV
// kernels; with less than this the backward falls back to recomputing the scores in every kernel
size_t ds6g_attention_workspace_bytes(int B, int T, int nh, int hd, int ld) {
    const size_t slab = (size_t)B * T * ld * sizeof(float);
    return 2 * 8 * slab + (size_t)8 * B * nh * T * 2 * sizeof(float) + handover_bytes(B, T, nh, hd);
}

// o = dropout(softmax(q k^T / sqrt(hd))) v ; lse[b][h][t] = logsumexp of the scaled scores
static int attention_fwd_impl(const float* q, const float* k, const float* v, int in16, float* o, int out16, float* lse, int B, int T,
                              int nh, int hd, int ld_qkv, int ld, float drop_p, uint64_t seed, uint64_t seed_off, void* ws,
                              size_t ws_bytes, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(q && k && v && o && lse && B > 0 && T > 0 && ld % 4 == 0 && ld >= nh * hd);
    DS6G_CHECK_ARG(ld_qkv % 4 == 0 && ld_qkv >= nh * hd && (size_t)B * T * ld_qkv * sizeof(float) < OOB_OFF);
    DS6G_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f);
    const size_t slab = (size_t)B * T * ld;
    DS6G_CHECK_ARG(slab * sizeof(float) < OOB_OFF);
    hipStream_t st = (hipStream_t)stream;
    AttnParams p{};
    p.q = q; p.k = k; p.v = v; p.lse = lse; p.T = T; p.nh = nh; p.B = B; p.ld = ld; p.ldq = ld_qkv; p.ldd = ld;
    const size_t esz = in16 ? 2 : 4;
    p.in16 = in16;
    p.bytes = (unsigned)(slab * esz); p.slab = slab;
    p.bytes_q = (unsigned)((((size_t)B * T - 1) * ld_qkv + nh * hd) * esz);
    p.scale = 1.0f / sqrtf((float)hd);
    ds6g_attn_drop_params(drop_p, &p.thr, &p.dscale); p.seed = seed; p.seed_off = seed_off; p.salt = g_ds6g_salt;
    const int ntiles = cdiv(T, 32);
    const int qblocks = cdiv(T, 128);
    const int per_cu = 3;  // measured (tools/bench_attn.py, DBG bits 20-23): 3 is best or tied for every head dim
    const size_t max_by_ws = ws ? ws_bytes / (slab * sizeof(float) + (size_t)B * nh * T * 2 * sizeof(float)) : 1;
    int splits = pick_splits((long)qblocks * nh * B, ntiles, per_cu, (int)(max_by_ws < 8 ? max_by_ws : 8), in16 ? 4.0 : 0.35);
    if (splits < 1) splits = 1;
    p.tiles_per_split = cdiv(ntiles, splits);
    splits = cdiv(ntiles, p.tiles_per_split);
    p.splits = splits;
    float* part = (float*)ws;
    p.out16 = out16;
    p.o = splits == 1 ? o : part;
    p.ml = splits == 1 ? nullptr : part + (size_t)splits * slab;
    int rc = launch_hd<0>(p, hd, dim3(qblocks, nh * splits, B), st);
    if (rc || splits == 1) return rc;
    const long n = (long)B * T * (nh * hd / 4);
    hipLaunchKernelGGL(attn_fwd_merge_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, part, p.ml, o, lse, B, T, nh, hd, ld,
                       splits, slab, out16);
    DS6G_LAUNCH_CHECK();
    return DS6G_OK;
}

int ds6g_attention_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int B, int T, int nh,
                       int hd, int ld_qkv, int ld, float drop_p, uint64_t seed, uint64_t seed_off, void* ws,
                       size_t ws_bytes, void* stream) {
    return attention_fwd_impl(q, k, v, 0, o, 0, lse, B, T, nh, hd, ld_qkv, ld, drop_p, seed, seed_off, ws, ws_bytes, stream);
}
// bf16-storage path: q / k / v fp32 (column blocks of the fused projection output), o written as bf16 [B*T][ld]
int ds6g_attention_fwd_bf16out(const float* q, const float* k, const float* v, void* o, float* lse, int B, int T, int nh,
                               int hd, int ld_qkv, int ld, float drop_p, uint64_t seed, uint64_t seed_off, void* ws,
                               size_t ws_bytes, void* stream) {
    return attention_fwd_impl(q, k, v, 0, (float*)o, 1, lse, B, T, nh, hd, ld_qkv, ld, drop_p, seed, seed_off, ws, ws_bytes,
                              stream);
}
// bf16-storage path, all operands bf16 in HBM: q / k / v [B*T][ld_qkv] and o [B*T][ld] bf16 (tiles reach the bf16 MFMA
// unconverted: bf16 LDS images, transposed LDS reads for the P.V product); lse and the split scratch fp32
int ds6g_attention_fwd_bf16(const void* q, const void* k, const void* v, void* o, float* lse, int B, int T, int nh, int hd,
                            int ld_qkv, int ld, float drop_p, uint64_t seed, uint64_t seed_off, void* ws, size_t ws_bytes,
                            void* stream) {
    DS6G_CHECK_ARG(ld_qkv % 8 == 0 && ld % 8 == 0);
    return attention_fwd_impl((const float*)q, (const float*)k, (const float*)v, 1, (float*)o, 1, lse, B, T, nh, hd, ld_qkv, ld,
                              drop_p, seed, seed_off, ws, ws_bytes, stream);
}

// gradients of the above; delta is a [B][nh][T] scratch (rowsum(dO*O)), written then read
static int attention_bwd_impl(const float* q, const float* k, const float* v, int in16, const float* o, int o16, const float* d_o,
                              const float* lse, float* delta, float* dq, float* dk, float* dv, int out16, int B, int T,
                              int nh, int hd, int ld_qkv, int ld, int ld_dqkv, float drop_p, uint64_t seed,
                              uint64_t seed_off, void* ws, size_t ws_bytes, void* stream) {
    DS6G_ENTER();
    DS6G_CHECK_ARG(q && k && v && o && d_o && lse && delta && dq && dk && dv && ld % 4 == 0 && ld >= nh * hd);
    DS6G_CHECK_ARG(ld_qkv % 4 == 0 && ld_qkv >= nh * hd && (size_t)B * T * ld_qkv * sizeof(float) < OOB_OFF);
    DS6G_CHECK_ARG(ld_dqkv % 4 == 0 && ld_dqkv >= nh * hd);
    const size_t slab = (size_t)B * T * ld;
    DS6G_CHECK_ARG(slab * sizeof(float) < OOB_OFF);
    hipStream_t st = (hipStream_t)stream;
    AttnParams p{};
    p.q = q; p.k = k; p.v = v; p.o = const_cast<float*>(o); p.lse = const_cast<float*>(lse); p.d_o = d_o;
    p.delta = delta; p.T = T; p.nh = nh; p.B = B; p.ld = ld; p.ldq = ld_qkv; p.ldd = ld_dqkv;
    p.out16 = out16; p.o16 = o16; p.in16 = in16;
    const size_t esz = in16 ? 2 : 4;
    p.bytes = (unsigned)(slab * esz); p.slab = slab;
    p.bytes_q = (unsigned)((((size_t)B * T - 1) * ld_qkv + nh * hd) * esz);
    p.scale = 1.0f / sqrtf((float)hd);
    ds6g_attn_drop_params(drop_p, &p.thr, &p.dscale); p.seed = seed; p.seed_off = seed_off; p.salt = g_ds6g_salt;
    const int ntiles = cdiv(T, 32);
    const int blocks128 = cdiv(T, 128);
    const int cols4 = nh * hd / 4;
    const long n4 = (long)B * T * cols4;
    const size_t hand = handover_bytes(B, T, nh, hd);
    if (g_ds6g_attn_handover && ws && ws_bytes >= hand + 2 * slab * sizeof(float)) {
        // ---- hand-over path: delta -> dK(/dV) kernel that also writes its dS (and P) tiles -> [dV from P] -> dQ from dS
        p.hs = (float*)ws;
        p.hp = hd >= 128 ? p.hs + hand / sizeof(float) / 2 : nullptr;
        p.nkg = blocks128 * 4;
        float* wsf = (float*)((char*)ws + hand);
        const size_t wsb = ws_bytes - hand;
        hipLaunchKernelGGL(attn_delta_kernel, dim3(cdiv((long)B * T * nh, 256)), dim3(256), 0, st, o, d_o, delta, B, T, nh,
                           hd, ld, o16 | (in16 << 1));
        DS6G_LAUNCH_CHECK();
        auto plan = [&](int per_cu, size_t slabs_per_split) {
            const size_t cap = wsb / (slabs_per_split * slab * sizeof(float));
            int splits = pick_splits((long)blocks128 * nh * B, ntiles, per_cu, (int)(cap < 8 ? cap : 8), in16 ? 4.0 : 0.35);
            if (splits < 1) splits = 1;
            p.tiles_per_split = cdiv(ntiles, splits);
            p.splits = cdiv(ntiles, p.tiles_per_split);
        };
        // hd = 128, exact fp32: dK and dV in ONE kernel (one wave per SIMD, 512 registers, 37 of them spilled) instead of the
        // dK kernel + dropped-P tiles through HBM (201 MB per layer) + attn_bwd_dv2_kernel: measured 690 -> 669 us per
        // backward at B = 12 (ds6g_set_debug_flags 0x04000000 restores the two-kernel form; the operand-rounding bf16 modes keep
        // it).  bf16-STORED tiles (round 4): K / V fragments are packed bf16, the fused kernel needs 484 registers and no spill -
        // bf16 step 431 -> 439 samples/s (DS6G_ATTN_FUSED128_BF16=0: the two-kernel form)
        static const int fused16 = [] { const char* e = getenv("DS6G_ATTN_FUSED128_BF16"); return e ? atoi(e) : 1; }();
        const bool fused128 = hd >= 128 && g_ds6g_attn_fused128 && (in16 ? fused16 != 0 : !g_ds6g_bf16);
        const bool two_pass = hd >= 128 && !fused128;
        plan(hd >= 128 ? 1 : (hd >= 64 ? 2 : 3), two_pass ? 1 : 2);
        p.dk = p.splits == 1 ? dk : wsf;
        p.dv = p.splits == 1 ? dv : wsf + (size_t)p.splits * slab;
        int rc = fused128 ? launch_hd<6>(p, hd, dim3(blocks128, nh * p.splits, B), st)
                          : launch_hd<3>(p, hd, dim3(blocks128, nh * p.splits, B), st);
        if (rc) return rc;
        if (p.splits > 1) {
            hipLaunchKernelGGL(slab_sum_kernel, dim3(cdiv(n4, 256), two_pass ? 1 : 2), dim3(256), 0, st, (const float*)wsf, dk,
                               (const float*)(wsf + (size_t)p.splits * slab), dv, n4, cols4, ld, ld_dqkv, p.splits, slab, out16);
            DS6G_LAUNCH_CHECK();
        }
        if (two_pass) {
            plan(2, 1);
            p.dv = p.splits == 1 ? dv : wsf;
            rc = launch_hd<4>(p, hd, dim3(blocks128, nh * p.splits, B), st);
            if (rc) return rc;
            if (p.splits > 1) {
                hipLaunchKernelGGL(slab_sum_kernel, dim3(cdiv(n4, 256), 1), dim3(256), 0, st, (const float*)wsf, dv,
                                   (const float*)nullptr, (float*)nullptr, n4, cols4, ld, ld_dqkv, p.splits, slab, out16);
                DS6G_LAUNCH_CHECK();
            }
        }
        plan(hd >= 64 ? 2 : 3, 1);
        p.dq = p.splits == 1 ? dq : wsf;
        rc = launch_hd<5>(p, hd, dim3(blocks128, nh * p.splits, B), st);
        if (rc) return rc;
        if (p.splits > 1) {
            hipLaunchKernelGGL(slab_sum_kernel, dim3(cdiv(n4, 256), 1), dim3(256), 0, st, (const float*)wsf, dq,
                               (const float*)nullptr, (float*)nullptr, n4, cols4, ld, ld_dqkv, p.splits, slab, out16);
            DS6G_LAUNCH_CHECK();
        }
        return DS6G_OK;
    }
    // the recomputing form reads o inside its dQ kernel (fp32 only): the bf16-storage path needs the hand-over workspace
    DS6G_CHECK_ARG(!o16 && !in16);
    // ---- dQ (split over keys)
    {
        const size_t max_by_ws = ws ? ws_bytes / (slab * sizeof(float)) : 1;
        // workgroups resident per CU (register budget of each instantiation): hd 128 -> 1, 64 -> 2, <= 32 -> 3
        int splits = pick_splits((long)blocks128 * nh * B, ntiles, hd >= 128 ? 1 : (hd >= 64 ? 2 : 3), (int)(max_by_ws < 8 ? max_by_ws : 8));
        p.tiles_per_split = cdiv(ntiles, splits);
        splits = cdiv(ntiles, p.tiles_per_split);
        p.splits = splits;
        p.dq = splits == 1 ? dq : (float*)ws;
        int rc = launch_hd<1>(p, hd, dim3(blocks128, nh * splits, B), st);
        if (rc) return rc;
        if (splits > 1) {
            hipLaunchKernelGGL(slab_sum_kernel, dim3(cdiv(n4, 256), 1), dim3(256), 0, st, (const float*)ws, dq,
                               (const float*)nullptr, (float*)nullptr, n4, cols4, ld, ld_dqkv, splits, slab, out16);
            DS6G_LAUNCH_CHECK();
        }
    }
    // ---- dK, dV (split over queries)
    {
        const size_t max_by_ws = ws ? ws_bytes / (2 * slab * sizeof(float)) : 1;
        int splits = pick_splits((long)blocks128 * nh * B, ntiles, hd >= 128 ? 1 : (hd >= 64 ? 2 : 3), (int)(max_by_ws < 8 ? max_by_ws : 8));
        p.tiles_per_split = cdiv(ntiles, splits);
        splits = cdiv(ntiles, p.tiles_per_split);
        p.splits = splits;
        float* wsf = (float*)ws;
        p.dk = splits == 1 ? dk : wsf;
        p.dv = splits == 1 ? dv : wsf + (size_t)splits * slab;
        int rc = launch_hd<2>(p, hd, dim3(blocks128, nh * splits, B), st);
        if (rc) return rc;
        if (splits > 1) {
            hipLaunchKernelGGL(slab_sum_kernel, dim3(cdiv(n4, 256), 2), dim3(256), 0, st, (const float*)wsf, dk,
                               (const float*)(wsf + (size_t)splits * slab), dv, n4, cols4, ld, ld_dqkv, splits, slab, out16);
            DS6G_LAUNCH_CHECK();
        }
    }
    return DS6G_OK;
}

int ds6g_attention_bwd(const float* q, const float* k, const float* v, const float* o, const float* d_o,
                       const float* lse, float* delta, float* dq, float* dk, float* dv, int B, int T, int nh, int hd,
                       int ld_qkv, int ld, int ld_dqkv, float drop_p, uint64_t seed, uint64_t seed_off, void* ws,
                       size_t ws_bytes, void* stream) {
    return attention_bwd_impl(q, k, v, 0, o, 0, d_o, lse, delta, dq, dk, dv, 0, B, T, nh, hd, ld_qkv, ld, ld_dqkv, drop_p, seed,
                              seed_off, ws, ws_bytes, stream);
}
// bf16-storage path: o (the forward output) is bf16, dq / dk / dv are written as bf16 [B*T][ld_dqkv]; q / k / v / d_o fp32.
// Needs the full ds6g_attention_workspace_bytes (hand-over form).
int ds6g_attention_bwd_bf16(const float* q, const float* k, const float* v, const void* o, const float* d_o,
                            const float* lse, float* delta, void* dq, void* dk, void* dv, int B, int T, int nh, int hd,
                            int ld_qkv, int ld, int ld_dqkv, float drop_p, uint64_t seed, uint64_t seed_off, void* ws,
                            size_t ws_bytes, void* stream) {
    return attention_bwd_impl(q, k, v, 0, (const float*)o, 1, d_o, lse, delta, (float*)dq, (float*)dk, (float*)dv, 1, B, T, nh, hd,
                              ld_qkv, ld, ld_dqkv, drop_p, seed, seed_off, ws, ws_bytes, stream);
}
// all operands bf16: q / k / v / o / d_o in, dq / dk / dv out (the hand-over tiles in ws, lse and delta stay fp32)
int ds6g_attention_bwd_bf16io(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                              float* delta, void* dq, void* dk, void* dv, int B, int T, int nh, int hd, int ld_qkv, int ld,
                              int ld_dqkv, float drop_p, uint64_t seed, uint64_t seed_off, void* ws, size_t ws_bytes,
                              void* stream) {
    DS6G_CHECK_ARG(ld_qkv % 8 == 0 && ld % 8 == 0 && ld_dqkv % 4 == 0);
    return attention_bwd_impl((const float*)q, (const float*)k, (const float*)v, 1, (const float*)o, 1, (const float*)d_o, lse, delta,
                              (float*)dq, (float*)dk, (float*)dv, 1, B, T, nh, hd, ld_qkv, ld, ld_dqkv, drop_p, seed, seed_off, ws,
                              ws_bytes, stream);
}

}  // extern "C"
