// Ceiling probe (GPU box): fp32 MFMA issue rate with and without LDS fragment reads, 32x32x2 vs 16x16x4,
// at 1..3 workgroups (of 4 waves) per CU.   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int LDS_READS>
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 128 * 20 + 128];
    for (int i = threadIdx.x; i < 2 * 128 * 20; i += 256) lds[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float af[2][8], bf[2][8];
    for (int i = 0; i < 2; ++i) for (int k = 0; k < 8; ++k) { af[i][k] = 1.0f + lane * 0.001f + k; bf[i][k] = 0.5f + k; }
    const int arow = (wave >> 1) * 64 + (lane & 31), brow = (wave & 1) * 64 + (lane & 31), kh = lane >> 5;
    for (int it = 0; it < iters; ++it) {
        if (LDS_READS) {
            const float* A = lds + (it & 1) * 64;  // alternate offset: reads cannot be hoisted
            for (int i = 0; i < 2; ++i) {
                const f32x4* ap = (const f32x4*)&A[(arow + i * 32) * 20 + kh * 8];
                const f32x4* bp = (const f32x4*)&A[128 * 20 + ((brow + i * 32) & 127) * 20 + kh * 8];
                f32x4 a0 = ap[0], a1 = ap[1], b0 = bp[0], b1 = bp[1];
                for (int k = 0; k < 4; ++k) { af[i][k] = a0[k]; af[i][4 + k] = a1[k]; bf[i][k] = b0[k]; bf[i][4 + k] = b1[k]; }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][kk], bf[j][kk], acc[i * 2 + j], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int LDS_READS>
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 128 * 20 + 128];
    for (int i = threadIdx.x; i < 2 * 128 * 20; i += 256) lds[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc[16];
    for (int a = 0; a < 16; ++a) for (int r = 0; r < 4; ++r) acc[a][r] = 0.f;
    f32x4 af[4], bf[4];
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 4; ++k) { af[i][k] = 1.0f + lane * 0.001f + k; bf[i][k] = 0.5f + k; }
    const int arow = (wave >> 1) * 64 + (lane & 15), brow = (wave & 1) * 64 + (lane & 15), kq = lane >> 4;
    for (int it = 0; it < iters; ++it) {
        if (LDS_READS) {
            for (int i = 0; i < 4; ++i) {
                af[i] = *(const f32x4*)&lds[(it & 1) * 64 + (arow + i * 16) * 20 + kq * 4];
                bf[i] = *(const f32x4*)&lds[(it & 1) * 64 + 128 * 20 + ((brow + i * 16) & 127) * 20 + kq * 4];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][kk], bf[j][kk], acc[i * 4 + j], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < 16; ++a) for (int r = 0; r < 4; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
void run(const char* name, F launch, int blocks_per_cu) {
    float* out;
    const int grid = 256 * blocks_per_cu, iters = 4000;
    hipMalloc(&out, grid * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(grid, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(grid, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per wave per iter: 64x64x16 MACs
    const double flops = 2.0 * 64 * 64 * 16 * 4.0 * grid * iters;
    printf("%-28s blocks/CU %d : %7.1f TFLOP/s (%.2f ms)\n", name, blocks_per_cu, flops / (ms * 1e-3) / 1e12, ms);
    hipFree(out);
}

int main() {
    for (int b = 1; b <= 3; ++b) {
        run("32x32x2 regs only", [](int g, float* o, int it) { hipLaunchKernelGGL(k32<0>, dim3(g), dim3(256), 0, 0, o, it); }, b);
        run("32x32x2 + ds_read_b128", [](int g, float* o, int it) { hipLaunchKernelGGL(k32<1>, dim3(g), dim3(256), 0, 0, o, it); }, b);
        run("16x16x4 regs only", [](int g, float* o, int it) { hipLaunchKernelGGL(k16<0>, dim3(g), dim3(256), 0, 0, o, it); }, b);
        run("16x16x4 + ds_read_b128", [](int g, float* o, int it) { hipLaunchKernelGGL(k16<1>, dim3(g), dim3(256), 0, 0, o, it); }, b);
    }
    return 0;
}
