// Sustained fp32 MFMA throughput under the chip's power management (GPU box): v_mfma_f32_32x32x2_f32 vs
// v_mfma_f32_16x16x4_f32 on RANDOM operands, ~100 ms per launch, one / two waves per SIMD, with the in-kernel clock
// (s_memtime / s_memrealtime).  Question: does the smaller shape hold a higher clock / deliver more FLOP/s at the power cap
// (as the guide reports for the bf16 shapes)?   hipcc --offload-arch=gfx950 -O3 tools/mfma_f32_sustained.hip -o /tmp/mfma_sus
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float rnd(unsigned& s) {
    s = s * 1664525u + 1013904223u;
    return (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
}

__global__ __launch_bounds__(256) void k32(float* out, unsigned long long* clk, int iters) {
    unsigned s = blockIdx.x * 256 + threadIdx.x + 1;
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float af[2][8], bf[2][8];
    for (int i = 0; i < 2; ++i) for (int k = 0; k < 8; ++k) { af[i][k] = rnd(s); bf[i][k] = rnd(s); }
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][kk], bf[j][kk], acc[i * 2 + j], 0, 0, 0);
        af[it & 1][it & 7] = rnd(s);   // keep the operands changing (one VALU op per 32 MFMAs)
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float t = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) t += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = t;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

__global__ __launch_bounds__(256) void k16(float* out, unsigned long long* clk, int iters) {
    unsigned s = blockIdx.x * 256 + threadIdx.x + 1;
    f32x4 acc[16];
    for (int a = 0; a < 16; ++a) for (int r = 0; r < 4; ++r) acc[a][r] = 0.f;
    f32x4 af[4], bf[4];
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 4; ++k) { af[i][k] = rnd(s); bf[i][k] = rnd(s); }
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][kk], bf[j][kk], acc[i * 4 + j], 0, 0, 0);
        af[it & 3][it & 3] = rnd(s);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float t = 0.f;
    for (int a = 0; a < 16; ++a) for (int r = 0; r < 4; ++r) t += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = t;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

template <typename F>
void run(const char* name, F launch, int blocks_per_cu, double flops_per_wave_iter) {
    float* out; unsigned long long* clk;
    const int grid = 256 * blocks_per_cu;
    hipMalloc(&out, grid * 256 * sizeof(float));
    hipMalloc(&clk, 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 60000 / blocks_per_cu;     // ~100 ms
    launch(grid, out, clk, iters);               // warm: brings the chip to its sustained state
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(grid, out, clk, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flops = flops_per_wave_iter * 4.0 * grid * iters;
    printf("%-10s waves/SIMD %d : %7.1f TFLOP/s  (%.1f ms, in-kernel clock %.2f GHz)\n", name, blocks_per_cu,
           flops / (ms * 1e-3) / 1e12, ms, (double)h[0] / (double)h[1] * 0.1);
    hipFree(out); hipFree(clk);
}

int main() {
    for (int rep = 0; rep < 2; ++rep)
        for (int b = 1; b <= 2; ++b) {
            run("32x32x2", [](int g, float* o, unsigned long long* c, int it) { hipLaunchKernelGGL(k32, dim3(g), dim3(256), 0, 0, o, c, it); }, b, 2.0 * 64 * 64 * 16);
            run("16x16x4", [](int g, float* o, unsigned long long* c, int it) { hipLaunchKernelGGL(k16, dim3(g), dim3(256), 0, 0, o, c, it); }, b, 2.0 * 64 * 64 * 16);
        }
    return 0;
}
