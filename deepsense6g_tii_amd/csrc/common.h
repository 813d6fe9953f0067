// Shared device/host helpers for the gfx950 kernel library (libds6g.so).
// All activations are fp32 NHWC in HBM; tokens are (B, T, C) row-major.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define DS6G_OK 0
#define DS6G_ERR_ARG 1
#define DS6G_ERR_LAUNCH 2
#define DS6G_ERR_WORKSPACE 3

#define DS6G_CHECK_ARG(cond)                                                                   \
    do {                                                                                       \
        if (!(cond)) {                                                                         \
            fprintf(stderr, "[ds6g] bad argument: %s (%s:%d)\n", #cond, __FILE__, __LINE__);   \
            return DS6G_ERR_ARG;                                                               \
        }                                                                                      \
    } while (0)

// hipGetLastError() is sticky per thread: a benign failure inside the caller's runtime (e.g. PyTorch probing a
// host pointer) would otherwise be reported by our next launch check.  Every entry point clears it first.
#define DS6G_ENTER() (void)hipGetLastError()

#define DS6G_LAUNCH_CHECK()                                                                    \
    do {                                                                                       \
        hipError_t e_ = hipGetLastError();                                                     \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "[ds6g] launch failed: %s (%s:%d)\n", hipGetErrorString(e_),       \
                    __FILE__, __LINE__);                                                       \
            return DS6G_ERR_LAUNCH;                                                            \
        }                                                                                      \
    } while (0)

// -DDS6G_GEMM_CLOCKS (tools/gemm_clocks.py builds its own library, never the product one): per-phase shader clocks of the
// implicit-GEMM kernels.  Lane 0 of wave 0 of ONE workgroup (the middle one of the grid) adds the s_memtime cycles it spends
// in each phase to CLK[phase]; every workgroup also records the 100 MHz real-time stamps of its entry and exit in
// WG[linear id][0 / 1], which gives the dispatch ramp and the tail of a launch.
#ifdef DS6G_GEMM_CLOCKS
#define GCLK_STORAGE(CLK, WG, READ)                                                                                      \
    __device__ unsigned long long CLK[16];                                                                               \
    __device__ unsigned long long WG[16384][2];                                                                          \
    extern "C" int READ(unsigned long long* clk16, unsigned long long* wg, int nwg, int reset) {                         \
        if (hipMemcpyFromSymbol(clk16, HIP_SYMBOL(CLK), sizeof(unsigned long long) * 16) != hipSuccess) return -1;       \
        if (wg && hipMemcpyFromSymbol(wg, HIP_SYMBOL(WG), sizeof(unsigned long long) * 2 * nwg) != hipSuccess) return -1; \
        if (reset) {                                                                                                     \
            unsigned long long z[16] = {};                                                                               \
            if (hipMemcpyToSymbol(HIP_SYMBOL(CLK), z, sizeof(z)) != hipSuccess) return -1;                               \
            void* wga_ = nullptr;                                                                                        \
            if (hipGetSymbolAddress(&wga_, HIP_SYMBOL(WG)) != hipSuccess || hipMemset(wga_, 0, sizeof(WG)) != hipSuccess) return -1; \
        }                                                                                                                \
        return 0;                                                                                                        \
    }
#define GCLK_DECL(WG)                                                                                                    \
    const int gclk_id_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                                 \
    const bool gclk_on_ = gclk_id_ == (int)(gridDim.x * gridDim.y * gridDim.z) / 2 && threadIdx.x == 0;                  \
    if (threadIdx.x == 0 && gclk_id_ < 16384) WG[gclk_id_][0] = __builtin_amdgcn_s_memrealtime();                        \
    unsigned long long gclk_t_ = __builtin_readcyclecounter();                                                           \
    const unsigned long long gclk_t0_ = gclk_t_;
#define GCLK(CLK, ph)                                                                                                    \
    do {                                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        const unsigned long long n_ = __builtin_readcyclecounter();                                                      \
        if (gclk_on_) CLK[ph] += n_ - gclk_t_;                                                                           \
        gclk_t_ = n_;                                                                                                    \
    } while (0)
#define GCLK_COUNT(CLK, ph) do { if (gclk_on_) CLK[ph] += 1; } while (0)
#define GCLK_END(CLK, WG)                                                                                                \
    do {                                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        if (gclk_on_) CLK[14] += __builtin_readcyclecounter() - gclk_t0_;                                                \
        if (threadIdx.x == 0 && gclk_id_ < 16384) WG[gclk_id_][1] = __builtin_amdgcn_s_memrealtime();                    \
    } while (0)
#else
#define GCLK_STORAGE(CLK, WG, READ)
#define GCLK_DECL(WG)
#define GCLK(CLK, ph)
#define GCLK_COUNT(CLK, ph)
#define GCLK_END(CLK, WG)
#endif

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// a - b on a float vector as packed FMAs: hipcc lowers a vector subtraction to one v_sub_f32 per element, while
// fma(b, -1, a) - the same single rounding - becomes v_pk_fma_f32 (two elements per instruction).  Beside an MFMA stream every
// vector instruction costs matrix-pipe time, so the transform code uses this for its differences.
template <typename V>
__device__ __forceinline__ V vsub(V a, V b) {
    float neg1 = -1.f;
    asm("" : "+s"(neg1));  // opaque, or instcombine folds fma(b, -1, a) back into a subtraction
    return __builtin_elementwise_fma(b, (V)neg1, a);
}
#ifdef __HIPCC__
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#endif

// process-wide matrix-core mode (ds6g_set_compute_mode): 0 = exact fp32 MFMA, 1 = bf16 operands / fp32 accumulate,
// 2 = split bf16 (hi*hi + hi*lo + lo*hi, fp32 accumulate), 3 = three-way split bf16 (six products, fp32-grade)
extern int g_ds6g_bf16;

#ifdef __HIPCC__
// BN output before the activation, ONE spelling shared by the forward kernels (bn_apply, the fused BN + ReLU + max-pool of
// the stem) and by the backward kernels that re-derive the ReLU mask from it instead of reading the activation tensor
// (identical rounding -> identical mask)
__device__ __forceinline__ f32x4 bn_affine(const f32x4 x, const f32x4 mu, const f32x4 is, const f32x4 g, const f32x4 b) {
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = fmaf((x[j] - mu[j]) * is[j], g[j], b[j]);
    return r;
}
// four consecutive activation elements as fp32, whatever the storage type (fp32, or bf16 on the bf16-storage path:
// the arithmetic of the normalisation / pooling / resampling kernels is fp32 either way)
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4(const __bf16* p) {
    const bf16x4_t t = *reinterpret_cast<const bf16x4_t*>(p);
    return f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
}
__device__ __forceinline__ void st4(float* p, const f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void st4(__bf16* p, const f32x4 v) {
    *reinterpret_cast<bf16x4_t*>(p) = bf16x4_t{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
}
// gradient of a 3x3 / stride 2 / pad 1 max-pool gathered on the fly: row = (n*H + h)*W + w of the pool INPUT; the up to
// four windows covering that pixel contribute where their stored argmax (r*3 + s, one byte per channel) points at it
// (dp: fp32, or bf16 when dp16)
struct PoolGrad { const void* dp; const uint8_t* idx; int H, W, Ho, Wo; int dp16; };
__device__ __forceinline__ f32x4 pooled_grad(const PoolGrad& pg, long row, int c4, int C) {
    const int w = (int)(row % pg.W);
    const long t = row / pg.W;
    const int h = (int)(t % pg.H);
    const long n = t / pg.H;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int oh = h >> 1; oh <= (h + 1) >> 1; ++oh) {
        if (oh >= pg.Ho) continue;
        const int r = h - (oh * 2 - 1);
        for (int ow = w >> 1; ow <= (w + 1) >> 1; ++ow) {
            if (ow >= pg.Wo) continue;
            const int s = w - (ow * 2 - 1);
            const long o = ((n * pg.Ho + oh) * pg.Wo + ow) * C + c4;
            const uint32_t pk = *reinterpret_cast<const uint32_t*>(pg.idx + o);
            const f32x4 g = pg.dp16 ? ld4(reinterpret_cast<const __bf16*>(pg.dp) + o) : ld4(reinterpret_cast<const float*>(pg.dp) + o);
            const uint32_t me = (uint32_t)(r * 3 + s);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (((pk >> (8 * j)) & 0xff) == me) acc[j] += g[j];
        }
    }
    return acc;
}
// a ~= hi + lo with hi = bf16(a) (RNE), lo = bf16(a - hi): |a - hi - lo| <= 2^-16 |a|
__device__ __forceinline__ void split_bf16x8(const float* f, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h = (__bf16)f[e];
        hi[e] = h;
        lo[e] = (__bf16)(f[e] - (float)h);
    }
}
// a == hi + mid + lo (three bf16 pieces, 24 significand bits)
__device__ __forceinline__ void split3_bf16x8(const float* f, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
    // truncating split: hi / mid are the top two 8-bit slices of the significand by masking, lo is what is left (<= 8 bits,
    // exact): a == hi + mid + lo exactly, and fewer VALU operations than a rounding split (measured +2 % step throughput;
    // per-kernel error 1.0-2.3e-6 vs 1.0-2.5e-6 for the fp32 MFMA kernels, tools/probe_modes.py)
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    u32x4_ hp, mp;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const unsigned u0 = __float_as_uint(f[e]), u1 = __float_as_uint(f[e + 1]);
        const float r0 = f[e] - __uint_as_float(u0 & 0xffff0000u), r1 = f[e + 1] - __uint_as_float(u1 & 0xffff0000u);
        const unsigned m0 = __float_as_uint(r0), m1 = __float_as_uint(r1);
        hp[e >> 1] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
        mp[e >> 1] = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
        lo[e] = (__bf16)(r0 - __uint_as_float(m0 & 0xffff0000u));
        lo[e + 1] = (__bf16)(r1 - __uint_as_float(m1 & 0xffff0000u));
    }
    hi = __builtin_bit_cast(bf16x8, hp);
    mid = __builtin_bit_cast(bf16x8, mp);
}
// acc += a * b with both operands split three ways: the six products above 2^-24 relative, smallest first
__device__ __forceinline__ f32x16 mfma_x6(const bf16x8& ah, const bf16x8& am, const bf16x8& al, const bf16x8& bh,
                                          const bf16x8& bm, const bf16x8& bl, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}
#endif

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Device-resident dropout salt (ds6g_set_dropout_salt): when set, every kernel that draws a dropout mask adds *salt to its
// counter offset at RUN time, so a captured hipGraph of the training step draws fresh masks on every replay although its
// launch arguments are frozen.  Thread-local: a caller thread sets it around its own launches (no process-wide state).
extern thread_local const uint64_t* g_ds6g_salt;

// Counter-based dropout RNG: keep(idx) is a pure function of (seed, idx), so backward kernels
// regenerate the mask instead of storing it.  One round of a 32-bit integer finalizer (lowbias32) over the low
// counter word, keyed by the seed and the high counter word (the attention kernels evaluate it T*T times per head
// in each of their three passes, so its instruction count is visible in the step time).
__host__ __device__ __forceinline__ uint32_t ds6g_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t ds6g_rand_u32(uint64_t seed, uint64_t idx) {
    const uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
    const uint32_t key = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x85ebca6bU);  // uniform per launch
    return ds6g_hash32(lo ^ key ^ (hi * 0x9E3779B9U));
}
// threshold = floor(p * 2^32); element is DROPPED when rand < threshold.
__host__ __device__ __forceinline__ bool ds6g_keep(uint64_t seed, uint64_t idx, uint32_t threshold) {
    return ds6g_rand_u32(seed, idx) >= threshold;
}
// The same keep(idx) for many counters idx = base + off with small 32-bit offsets (one attention tile: 16 elements per lane
// and step): the 64-bit add, the multiply of the high counter word and the seed key are paid once per (lane, tile), an
// element costs the finalizer plus a 32-bit add and a carry select.  Bit-identical to ds6g_keep(seed, base + off, thr)
// for off < 2^32.
struct Ds6gKeepBase {
    uint32_t lo, hic, key;
    __device__ __forceinline__ Ds6gKeepBase(uint64_t seed, uint64_t base)
        : lo((uint32_t)base), hic((uint32_t)(base >> 32) * 0x9E3779B9U),
          key((uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x85ebca6bU)) {}
    __device__ __forceinline__ bool keep(uint32_t off, uint32_t threshold) const {
        const uint32_t l = lo + off;
        const uint32_t h = hic + (l < off ? 0x9E3779B9U : 0U);   // carry into the high word: (hi + 1) * C = hi * C + C
        return ds6g_hash32(l ^ key ^ h) >= threshold;
    }
};
// Dropout on the attention probabilities ([B * nh][T][T] per site; evaluated T * T times per head in the forward and again
// in the backward): FOUR keep decisions per hash.  Element (row, key) - row = (b * nh + h) * T + query - belongs to the
// key quad  row * ceil(T / 4) + (key >> 2)  (counter = the site's offset + that index); the quad's hash yields two words
//   x = finalizer rounds 1-2 of ds6g_hash32 on the keyed counter;  w0 = fin(x * 0x846ca68b), w1 = fin(x * 0xC2B2AE35),
//   fin(v) = v ^ (v >> 16)      (w0 IS ds6g_hash32 of the keyed counter)
// whose four 16-bit halves are the decisions of keys 4 k .. 4 k + 3: key & 3 = 0 -> low half of w0, 1 -> high half of w0,
// 2 -> low half of w1, 3 -> high half of w1; the element is kept when its half >= floor(p * 2^16).  The halves of one
// word are independent by construction (fin is a bijection of the two halves), the two words are two multiplicative
// hashes of one well-mixed value; measured pairwise correlation of the four decisions over 11 M quads: within 2 sigma of 0
// (tools/attn_mask_stats.py).  Cost per element: a quarter of the hash plus a shift and a compare, against the whole hash.
struct Ds6gKeep4Base {
    uint32_t lo, hic, key;
    __device__ __forceinline__ Ds6gKeep4Base(uint64_t seed, uint64_t base)
        : lo((uint32_t)base), hic((uint32_t)(base >> 32) * 0x9E3779B9U),
          key((uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x85ebca6bU)) {}
    // the two words of the quad at counter base + off
    __device__ __forceinline__ void words(uint32_t off, uint32_t& w0, uint32_t& w1) const {
        const uint32_t l = lo + off;
        uint32_t x = l ^ key ^ (hic + (l < off ? 0x9E3779B9U : 0U));
        x ^= x >> 16; x *= 0x7feb352dU;
        x ^= x >> 15;
        w0 = x * 0x846ca68bU; w0 ^= w0 >> 16;
        w1 = x * 0xC2B2AE35U; w1 ^= w1 >> 16;
    }
};
// decision of key & 3 == F from the quad's words against thi = floor(p * 2^16) << 16 (= threshold32 & 0xffff0000): a low
// half is shifted to the top; the bits below it never change the outcome of >= against a multiple of 2^16
template <int F>
__device__ __forceinline__ bool ds6g_keep4(uint32_t w0, uint32_t w1, uint32_t thi) {
    const uint32_t w = (F & 2) ? w1 : w0;
    return ((F & 1) ? w : (w << 16)) >= thi;
}
static inline uint32_t ds6g_drop_threshold(float p) {
    if (p <= 0.f) return 0u;
    double t = (double)p * 4294967296.0;
    if (t > 4294967295.0) t = 4294967295.0;
    return (uint32_t)t;
}

// attn_drop sites draw four 16-bit decisions per hash (ds6g_keep4): the drop probability they realise is floor(p * 2^16) / 2^16,
// not p.  The inverted-dropout scale follows the REALISED probability so that E[mask * scale] = 1 (at p = 0.1 the exact-p scale
// is off by 1e-5); a probability below 2^-16 drops nothing and is treated as no dropout (scale 1, threshold 0).
static inline void ds6g_attn_drop_params(float p, uint32_t* thr, float* dscale) {
    uint32_t t = ds6g_drop_threshold(p);
    const uint32_t t16 = t >> 16;
    if (t16 == 0) { *thr = 0u; *dscale = 1.f; return; }
    *thr = t;
    *dscale = 1.f / (1.f - (float)t16 * (1.f / 65536.f));
}

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_reduce_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

#ifdef __HIPCC__
// ---- global -> LDS DMA (buffer_load_dwordx4 ... lds) -------------------------------------------------
// One LDS-DMA piece: 64 lanes x 16 B land at lds_base + lane*16 (lane-linear); a lane whose byte offset is
// >= the descriptor's size (OOB_OFF) gets ZEROS written (measured on gfx950: tools/t_glds.hip) - that is how
// conv halos, tile tails and k tails are zero-filled without touching a VGPR.
#define OOB_OFF 0x80000000u
typedef __attribute__((address_space(3))) void lds_void;
typedef int i32x4 __attribute__((ext_vector_type(4)));

// raw buffer descriptor: base, stride 0, num_records = bytes, DATA_FORMAT 32 (gfx9 raw buffer)
__device__ __forceinline__ i32x4 make_srd(const void* ptr, unsigned bytes) {
    const unsigned long a = (unsigned long)ptr;
    return i32x4{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}

// The DMA is issued from inline asm on purpose: hipcc treats a builtin LDS-DMA as aliasing every LDS read and
// drains vmcnt before the fragment reads of the OTHER buffer, which serialises DMA and MFMA.  From asm the
// compiler does not see the load; its completion is waited for by the explicit `s_waitcnt vmcnt(0)` ahead of
// the barrier that precedes the reads.  M0 (LDS destination base) is saved/restored inside the statement.
// soffset: wave-uniform byte offset added to the address (an SGPR; NOT part of the range check, so an OOB_OFF lane
// stays out of range whatever soffset is).
__device__ __forceinline__ void dma16(const i32x4 srd, unsigned lds_byte_addr, unsigned voffset, unsigned soffset = 0u) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voffset), "s"(lds_byte_addr), "s"(srd), "s"(soffset)
        : "memory");
}

// m / d and m % d for 0 <= m < 2^22 without the ~40-instruction integer division sequence: float quotient estimate
// (v_rcp_f32 is good to 1 ulp, so the estimate is off by at most one) and one correction step.  The kernels' set-up code
// turns tile rows into (image, row, column) with these (16 divisions per lane in the 128 x 128 bgemm tile: ~1 us of every
// workgroup's life before its first load is issued, -DDS6G_GEMM_CLOCKS).  Falls back to the exact division above 2^22.
__device__ __forceinline__ void fast_divmod(int m, int d, int& q, int& r) {
    if ((unsigned)m >= (1u << 22)) {
        q = m / d;
        r = m - q * d;
        return;
    }
    q = (int)((float)m * __builtin_amdgcn_rcpf((float)d));
    r = m - q * d;
    if (r < 0) { --q; r += d; }
    else if (r >= d) { ++q; r -= d; }
}

__device__ __forceinline__ unsigned lds_addr(const float* p) {
    return (unsigned)(unsigned long)(lds_void*)p;
}
#endif
