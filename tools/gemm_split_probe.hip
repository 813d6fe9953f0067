// Probe (not part of the product path): C[M][N] = A[M][K] . B[N][K]^T in fp32 storage with the operands split into bf16
// pieces ONCE per workgroup while they are staged global -> registers -> LDS (bf16 planes), so the MFMA loop is free of
// VALU work: PLANES 3 = three-way split, six products ("f32x6", fp32-grade); PLANES 2 = hi/lo, three products ("f32x3").
// Answers the question DESIGN.md section 9 (1) asks: how far above the power-limited fp32 MFMA rate does this form get?
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/gemm_split_probe.hip -o tools/gemm_split_probe.bin && tools/gemm_split_probe.bin
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 128, BN = 128, BK = 32;

// 4 consecutive-k floats -> PLANES x (4 bf16 = 8 bytes)
template <int PLANES>
__device__ __forceinline__ void split4(const f32x4 v, u32x2* out) {
    f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
#pragma unroll
    for (int p = 0; p < PLANES; ++p) {
        const bf16x2 ha = __builtin_convertvector(a, bf16x2), hb = __builtin_convertvector(b, bf16x2);
        out[p] = u32x2{__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb)};
        if (p + 1 < PLANES) {
            a = a - __builtin_convertvector(ha, f32x2);
            b = b - __builtin_convertvector(hb, f32x2);
        }
    }
}

// ABL (timing ablations, results wrong): 1 no global loads / staging, 2 no MFMAs, 3 no fragment reads
template <int PLANES, int ABL = 0>
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                            float* __restrict__ C, int M, int N, int K) {
    // plane image: [row 128][32 bf16] = 64 B per row, 16-B chunk index XOR-swizzled by (row >> 1) & 3
    __shared__ __attribute__((aligned(16))) unsigned char As[PLANES][BM * 64];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[PLANES][BN * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, khalf = lane >> 5;
    const int tiles_n = N / BN;
    const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging: 1024 float4 chunks per operand tile, 4 per thread: chunk c -> row c / 8, k chunk c % 8
    f32x4 ra[4], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + i * 256, row = c >> 3, kc = c & 7;
            ra[i] = *reinterpret_cast<const f32x4*>(A + (size_t)(m0 + row) * K + k0 + kc * 4);
            rb[i] = *reinterpret_cast<const f32x4*>(B + (size_t)(n0 + row) * K + k0 + kc * 4);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + i * 256, row = c >> 3, kc = c & 7;
            const int off = row * 64 + ((((kc >> 1) ^ ((row >> 1) & 3)) << 4) | ((kc & 1) << 3));
            u32x2 pa[PLANES], pb[PLANES];
            split4<PLANES>(ra[i], pa);
            split4<PLANES>(rb[i], pb);
#pragma unroll
            for (int p = 0; p < PLANES; ++p) {
                *reinterpret_cast<u32x2*>(&As[p][off]) = pa[p];
                *reinterpret_cast<u32x2*>(&Bs[p][off]) = pb[p];
            }
        }
    };

    const int nk = K / BK;
    gload(0);
    for (int kt = 0; kt < nk; ++kt) {
        if (ABL != 1) stage();
        __syncthreads();
        if (ABL != 1 && kt + 1 < nk) gload((kt + 1) * BK);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            bf16x8 a[2][PLANES], b[2][PLANES];
            if (ABL == 3) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int p = 0; p < PLANES; ++p) { a[i][p] = __builtin_bit_cast(bf16x8, ra[i]); b[i][p] = __builtin_bit_cast(bf16x8, rb[i]); }
            }
#pragma unroll
            for (int i = 0; i < 2 && ABL != 3; ++i) {
                const int row = wm * 64 + i * 32 + l31;
                const int off = row * 64 + (((2 * q + khalf) ^ ((row >> 1) & 3)) << 4);
#pragma unroll
                for (int p = 0; p < PLANES; ++p) a[i][p] = *reinterpret_cast<const bf16x8*>(&As[p][off]);
            }
#pragma unroll
            for (int j = 0; j < 2 && ABL != 3; ++j) {
                const int row = wn * 64 + j * 32 + l31;
                const int off = row * 64 + (((2 * q + khalf) ^ ((row >> 1) & 3)) << 4);
#pragma unroll
                for (int p = 0; p < PLANES; ++p) b[j][p] = *reinterpret_cast<const bf16x8*>(&Bs[p][off]);
            }
            // products above 2^-24 (PLANES 3) / 2^-16 (PLANES 2), smallest first; index = plane of a, plane of b
#pragma unroll
            for (int s = (PLANES == 3 ? 2 : PLANES - 1); s >= 0; --s) {
#pragma unroll
                for (int pa = 0; pa < PLANES; ++pa) {
                    const int pb = s - pa;
                    if (pb < 0 || pb >= PLANES) continue;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            if (ABL != 2) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][pa], b[j][pb], acc[i][j], 0, 0, 0);
                            else acc[i][j][0] += (float)a[i][pa][0] * (float)b[j][pb][0];
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                const int col = n0 + wn * 64 + j * 32 + l31;
                C[(size_t)row * N + col] = acc[i][j][r];
            }
}

// Second form: two LDS buffers, ONE barrier per k-tile; the split + LDS writes of tile t+1 sit in the same basic block
// as the MFMA chain of tile t, so the scheduler can issue the VALU work in the shadow of the matrix pipe.
template <int PLANES>
__global__ __launch_bounds__(256, PLANES == 3 ? 1 : 2) void gemm_split_db_kernel(const float* __restrict__ A,
                                                                                 const float* __restrict__ B,
                                                                                 float* __restrict__ C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) unsigned char As[2][PLANES][BM * 64];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2][PLANES][BN * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, khalf = lane >> 5;
    const int tiles_n = N / BN;
    const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ra[4], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + i * 256, row = c >> 3, kc = c & 7;
            ra[i] = *reinterpret_cast<const f32x4*>(A + (size_t)(m0 + row) * K + k0 + kc * 4);
            rb[i] = *reinterpret_cast<const f32x4*>(B + (size_t)(n0 + row) * K + k0 + kc * 4);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + i * 256, row = c >> 3, kc = c & 7;
            const int off = row * 64 + ((((kc >> 1) ^ ((row >> 1) & 3)) << 4) | ((kc & 1) << 3));
            u32x2 pa[PLANES], pb[PLANES];
            split4<PLANES>(ra[i], pa);
            split4<PLANES>(rb[i], pb);
#pragma unroll
            for (int p = 0; p < PLANES; ++p) {
                *reinterpret_cast<u32x2*>(&As[buf][p][off]) = pa[p];
                *reinterpret_cast<u32x2*>(&Bs[buf][p][off]) = pb[p];
            }
        }
    };
    auto compute = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            bf16x8 a[2][PLANES], b[2][PLANES];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wm * 64 + i * 32 + l31;
                const int off = row * 64 + (((2 * q + khalf) ^ ((row >> 1) & 3)) << 4);
#pragma unroll
                for (int p = 0; p < PLANES; ++p) a[i][p] = *reinterpret_cast<const bf16x8*>(&As[buf][p][off]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = wn * 64 + j * 32 + l31;
                const int off = row * 64 + (((2 * q + khalf) ^ ((row >> 1) & 3)) << 4);
#pragma unroll
                for (int p = 0; p < PLANES; ++p) b[j][p] = *reinterpret_cast<const bf16x8*>(&Bs[buf][p][off]);
            }
#pragma unroll
            for (int s = (PLANES == 3 ? 2 : PLANES - 1); s >= 0; --s) {
#pragma unroll
                for (int pa = 0; pa < PLANES; ++pa) {
                    const int pb = s - pa;
                    if (pb < 0 || pb >= PLANES) continue;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][pa], b[j][pb], acc[i][j], 0, 0, 0);
                }
            }
        }
    };
    const int nk = K / BK;
    gload(0);
    stage(0);
    if (nk > 1) gload(BK);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage(buf ^ 1);          // regs hold tile kt+1 (loaded during the previous iteration)
        if (kt + 2 < nk) gload((kt + 2) * BK);    // lands during this and the next compute phase
        compute(buf);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                const int col = n0 + wn * 64 + j * 32 + l31;
                C[(size_t)row * N + col] = acc[i][j][r];
            }
}

// Third form: 256 x 128 tile, 8 waves (4 x 2, 64 x 64 each): 42.7 flop per L2 byte instead of 32, and the staging work
// (split + LDS writes) per MFMA is 3/4 of the 128 x 128 form.  Single LDS buffer, two barriers per k-tile.
template <int PLANES>
__global__ __launch_bounds__(512, 1) void gemm_split_256_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                                float* __restrict__ C, int M, int N, int K) {
    constexpr int TM_ = 256, TN_ = 128;
    __shared__ __attribute__((aligned(16))) unsigned char As[PLANES][TM_ * 64];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[PLANES][TN_ * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, khalf = lane >> 5;
    const int tiles_n = N / TN_;
    const int m0 = (blockIdx.x / tiles_n) * TM_, n0 = (blockIdx.x % tiles_n) * TN_;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ra[4], rb[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + i * 512, row = c >> 3, kc = c & 7;
            ra[i] = *reinterpret_cast<const f32x4*>(A + (size_t)(m0 + row) * K + k0 + kc * 4);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + i * 512, row = c >> 3, kc = c & 7;
            rb[i] = *reinterpret_cast<const f32x4*>(B + (size_t)(n0 + row) * K + k0 + kc * 4);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + i * 512, row = c >> 3, kc = c & 7;
            const int off = row * 64 + ((((kc >> 1) ^ ((row >> 1) & 3)) << 4) | ((kc & 1) << 3));
            u32x2 pa[PLANES];
            split4<PLANES>(ra[i], pa);
#pragma unroll
            for (int p = 0; p < PLANES; ++p) *reinterpret_cast<u32x2*>(&As[p][off]) = pa[p];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + i * 512, row = c >> 3, kc = c & 7;
            const int off = row * 64 + ((((kc >> 1) ^ ((row >> 1) & 3)) << 4) | ((kc & 1) << 3));
            u32x2 pb[PLANES];
            split4<PLANES>(rb[i], pb);
#pragma unroll
            for (int p = 0; p < PLANES; ++p) *reinterpret_cast<u32x2*>(&Bs[p][off]) = pb[p];
        }
    };
    const int nk = K / BK;
    gload(0);
    for (int kt = 0; kt < nk; ++kt) {
        stage();
        __syncthreads();
        if (kt + 1 < nk) gload((kt + 1) * BK);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            bf16x8 a[2][PLANES], b[2][PLANES];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wm * 64 + i * 32 + l31;
                const int off = row * 64 + (((2 * q + khalf) ^ ((row >> 1) & 3)) << 4);
#pragma unroll
                for (int p = 0; p < PLANES; ++p) a[i][p] = *reinterpret_cast<const bf16x8*>(&As[p][off]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = wn * 64 + j * 32 + l31;
                const int off = row * 64 + (((2 * q + khalf) ^ ((row >> 1) & 3)) << 4);
#pragma unroll
                for (int p = 0; p < PLANES; ++p) b[j][p] = *reinterpret_cast<const bf16x8*>(&Bs[p][off]);
            }
#pragma unroll
            for (int s = (PLANES == 3 ? 2 : PLANES - 1); s >= 0; --s) {
#pragma unroll
                for (int pa = 0; pa < PLANES; ++pa) {
                    const int pb = s - pa;
                    if (pb < 0 || pb >= PLANES) continue;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][pa], b[j][pb], acc[i][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                const int col = n0 + wn * 64 + j * 32 + l31;
                C[(size_t)row * N + col] = acc[i][j][r];
            }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int PLANES, int ABL = 0>
void run(int M, int N, int K, bool check) {
    std::vector<float> ha((size_t)M * K), hb((size_t)N * K);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : ha) v = rnd();
    for (auto& v : hb) v = rnd() * 0.05f;
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, ha.size() * 4)); CK(hipMalloc(&dB, hb.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemcpy(dA, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    dim3 grid(ABL == 8 ? (M / 256) * (N / 128) : (M / BM) * (N / BN)), block(ABL == 8 ? 512 : 256);
    if (ABL == 9) hipLaunchKernelGGL((gemm_split_db_kernel<PLANES>), grid, block, 0, 0, dA, dB, dC, M, N, K);
    else if (ABL == 8) hipLaunchKernelGGL((gemm_split_256_kernel<PLANES>), grid, block, 0, 0, dA, dB, dC, M, N, K);
    else hipLaunchKernelGGL((gemm_split_kernel<PLANES, (ABL >= 8) ? 0 : ABL>), grid, block, 0, 0, dA, dB, dC, M, N, K);
    CK(hipDeviceSynchronize());
    if (check) {
        std::vector<float> hc((size_t)M * N);
        CK(hipMemcpy(hc.data(), dC, hc.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, big = 0;
        for (int m = 0; m < M; m += 7)
            for (int n = 0; n < N; n += 5) {
                double ref = 0;
                for (int k = 0; k < K; ++k) ref += (double)ha[(size_t)m * K + k] * hb[(size_t)n * K + k];
                worst = fmax(worst, fabs(ref - hc[(size_t)m * N + n]));
                big = fmax(big, fabs(ref));
            }
        printf("planes %d  M %d N %d K %d  max err / max |c| = %.2e\n", PLANES, M, N, K, worst / big);
    } else {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int reps = 20;
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) {
            if (ABL == 9) hipLaunchKernelGGL((gemm_split_db_kernel<PLANES>), grid, block, 0, 0, dA, dB, dC, M, N, K);
            else if (ABL == 8) hipLaunchKernelGGL((gemm_split_256_kernel<PLANES>), grid, block, 0, 0, dA, dB, dC, M, N, K);
            else hipLaunchKernelGGL((gemm_split_kernel<PLANES, (ABL >= 8) ? 0 : ABL>), grid, block, 0, 0, dA, dB, dC, M, N, K);
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        printf("planes %d abl %d  M %5d N %4d K %4d  %8.1f us  %6.1f TFLOP/s (algorithmic)\n", PLANES, ABL, M, N, K, us,
               2.0 * M * N * K / us / 1e6);
    }
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
}

int main() {
    run<3>(256, 256, 128, true);
    run<2>(256, 256, 128, true);
    run<3>(384, 128, 2048, true);
    run<3, 9>(384, 128, 2048, true);
    run<3, 8>(512, 256, 256, true);
    run<2, 9>(256, 256, 128, true);
    const int shapes[][3] = {{11520, 2048, 512}, {11520, 512, 2048}, {11520, 1536, 512}, {11520, 512, 512}, {12288, 4096, 4096}};
    for (auto& sh : shapes) {
        run<3>(sh[0], sh[1], sh[2], false);
        run<2>(sh[0], sh[1], sh[2], false);
        run<1>(sh[0], sh[1], sh[2], false);
        run<3, 8>(sh[0], sh[1], sh[2], false);
        run<2, 8>(sh[0], sh[1], sh[2], false);
        run<1, 8>(sh[0], sh[1], sh[2], false);
    }
    if (getenv("ABLATE")) {
        run<3, 1>(12288, 4096, 4096, false); run<3, 2>(12288, 4096, 4096, false); run<3, 3>(12288, 4096, 4096, false);
        run<2, 1>(12288, 4096, 4096, false); run<2, 2>(12288, 4096, 4096, false); run<2, 3>(12288, 4096, 4096, false);
        run<1, 1>(12288, 4096, 4096, false); run<1, 2>(12288, 4096, 4096, false); run<1, 3>(12288, 4096, 4096, false);
    }
    return 0;
}
