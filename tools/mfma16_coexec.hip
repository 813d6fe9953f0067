// Microbenchmark (round 3): does the fp32 MFMA's pressure on the SIMD's VGPR write port depend on its SHAPE?
// v_mfma_f32_32x32x2_f32 writes 16 result rows in its 64 cycles, v_mfma_f32_16x16x4_f32 4 rows in its 32 cycles - half the
// rows per FLOP at the same FLOP rate (64 FLOP / clk / SIMD).  Measured here, per shape:
//   (a) the MFMA wave alone (cycles per 262 144 FLOP per lane-row... i.e. per 64 32x32x2 or 128 16x16x4 MFMAs),
//   (b) a partner wave of the same SIMD issuing 64 v_add_f32 per iteration beside it,
//   (c) the SAME wave interleaving n independent v_add_f32 (or ds_read_b32) after every MFMA.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma16_coexec.hip -o tools/mfma16_coexec.bin && ./tools/mfma16_coexec.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// SHAPE 0: 8 accumulators x 8 MFMAs 32x32x2 per iteration; SHAPE 1: 16 accumulators x 8 MFMAs 16x16x4 (same FLOPs)
// MODE 0: partner idle; 1: partner 64 v_add_f32 per iteration; 2..4: same wave, 1 / 2 / 4 v_add_f32 after every MFMA;
// 5: same wave, one ds_read_b32 after every MFMA
template <int SHAPE, int MODE>
__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* t, int iters) {
    __shared__ float lds[4096];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (wave < 4) {
        float a = lane * 0.001f, b = 1.f + lane * 0.002f;
        float x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3;
        float s = 0.f;
        unsigned long long t0, t1;
        constexpr int NV = MODE == 2 ? 1 : (MODE == 3 ? 2 : (MODE == 4 ? 4 : 0));
        auto filler = [&](int i) {
            if (NV >= 1) asm volatile("v_add_f32 %0, %0, 1.0" : "+v"(x0));
            if (NV >= 2) asm volatile("v_add_f32 %0, %0, 1.0" : "+v"(x1));
            if (NV >= 4) { asm volatile("v_add_f32 %0, %0, 1.0" : "+v"(x2)); asm volatile("v_add_f32 %0, %0, 1.0" : "+v"(x3)); }
            if (MODE == 5) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(x0) : "v"((unsigned)(lane * 4)), "n"(0) : "memory");
        };
        if (SHAPE == 0) {
            f32x16 acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            t0 = __builtin_readcyclecounter();
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                        filler(i);
                    }
                if (MODE == 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            t1 = __builtin_readcyclecounter();
#pragma unroll
            for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
        } else {
            f32x4 acc[16];
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
            t0 = __builtin_readcyclecounter();
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                        filler(i);
                    }
                if (MODE == 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            t1 = __builtin_readcyclecounter();
#pragma unroll
            for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
        }
        out[blockIdx.x * 512 + threadIdx.x] = s + x0 + x1 + x2 + x3;
        if (wave == 0 && lane == 0) t[blockIdx.x * 2] = t1 - t0;
    } else {
        float x = lane;
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
            if (MODE == 1) {
#pragma unroll
                for (int r = 0; r < 64; ++r) asm volatile("v_add_f32 %0, %0, 1.0" : "+v"(x));
            }
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        out[blockIdx.x * 512 + threadIdx.x] = x + lds[lane];
        if (wave == 4 && lane == 0) t[blockIdx.x * 2 + 1] = t1 - t0;
    }
}

template <int SHAPE, int MODE>
void run(const char* name, float* out, unsigned long long* t) {
    const int iters = 200;
    hipLaunchKernelGGL((k<SHAPE, MODE>), dim3(256), dim3(512), 0, 0, out, t, iters);
    hipLaunchKernelGGL((k<SHAPE, MODE>), dim3(256), dim3(512), 0, 0, out, t, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[2];
    (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-62s mfma wave %7.1f cycles / iteration (4096 = MFMA only)   partner %7.1f\n", name, (double)h[0] / iters, (double)h[1] / iters);
}

int main() {
    float* out; unsigned long long* t;
    (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMalloc(&t, 256 * 2 * 8);
    run<0, 0>("32x32x2 x64, partner idle", out, t);
    run<1, 0>("16x16x4 x128, partner idle", out, t);
    run<0, 1>("32x32x2 x64, partner wave 64 v_add_f32", out, t);
    run<1, 1>("16x16x4 x128, partner wave 64 v_add_f32", out, t);
    run<0, 2>("32x32x2 x64 + 1 v_add_f32 after every MFMA (64 / iteration)", out, t);
    run<1, 2>("16x16x4 x128 + 1 v_add_f32 after every MFMA (128 / iteration)", out, t);
    run<0, 3>("32x32x2 x64 + 2 v_add_f32 after every MFMA (128)", out, t);
    run<1, 3>("16x16x4 x128 + 2 v_add_f32 after every MFMA (256)", out, t);
    run<0, 4>("32x32x2 x64 + 4 v_add_f32 after every MFMA (256)", out, t);
    run<1, 4>("16x16x4 x128 + 4 v_add_f32 after every MFMA (512)", out, t);
    run<0, 5>("32x32x2 x64 + 1 ds_read_b32 after every MFMA (64)", out, t);
    run<1, 5>("16x16x4 x128 + 1 ds_read_b32 after every MFMA (128)", out, t);
    return 0;
}
