// Microbenchmark: what a partner wave's VALU / LDS / VMEM stream gets while the older wave of its SIMD issues fp32 32x32x2 MFMAs
// back to back - with the accumulators in VGPRs (what hipcc picks when the kernel fits 256 VGPRs) and in AGPRs.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_coexec.hip -o /tmp/coexec && /tmp/coexec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int AGPR, int PARTNER, int PRIO>
__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* t, int iters) {
    __shared__ float lds[4096];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    __syncthreads();
    if (wave < 4) {
        f32x16 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            if (AGPR) asm volatile("" : "+a"(acc[i]));
        }
        float a = lane * 0.001f, b = 1.f + lane * 0.002f;
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
        out[blockIdx.x * 512 + threadIdx.x] = s;
        if (wave == 0 && lane == 0) t[blockIdx.x * 2] = t1 - t0;
    } else {
        if (PRIO) __builtin_amdgcn_s_setprio(2);
        f32x2 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = f32x2{lane * 0.5f + i, 1.f * i};
        float x = lane;
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
            if (PARTNER == 1) {  // 64 packed fp32 adds
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = v[i] + v[(i + 1) & 7];
            } else if (PARTNER == 2) {  // 64 plain fp32 adds
#pragma unroll
                for (int r = 0; r < 64; ++r) x = x + 1.5f;
                asm volatile("" : "+v"(x));
            } else if (PARTNER == 3) {  // 16 ds_write_b64
#pragma unroll
                for (int i = 0; i < 16; ++i) *reinterpret_cast<f32x2*>(&lds[(i * 64 + lane) * 2 & 4095]) = v[i & 7];
            }
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        float s = x;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
        out[blockIdx.x * 512 + threadIdx.x] = s + lds[lane];
        if (wave == 4 && lane == 0) t[blockIdx.x * 2 + 1] = t1 - t0;
    }
}

template <int AGPR, int PARTNER, int PRIO>
void run(const char* name, float* out, unsigned long long* t) {
    const int iters = 200;
    hipLaunchKernelGGL((k<AGPR, PARTNER, PRIO>), dim3(256), dim3(512), 0, 0, out, t, iters);
    hipLaunchKernelGGL((k<AGPR, PARTNER, PRIO>), dim3(256), dim3(512), 0, 0, out, t, iters);
    hipDeviceSynchronize();
    unsigned long long h[2];
    hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s mfma wave: %7.1f cycles / 64 MFMAs   partner: %7.1f cycles / iteration\n", name, (double)h[0] / iters, (double)h[1] / iters);
}

int main() {
    float* out; unsigned long long* t;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&t, 256 * 2 * 8);
    run<0, 0, 0>("acc VGPR, partner idle", out, t);
    run<1, 0, 0>("acc AGPR, partner idle", out, t);
    run<0, 1, 0>("acc VGPR, partner 64 v_pk_add_f32", out, t);
    run<1, 1, 0>("acc AGPR, partner 64 v_pk_add_f32", out, t);
    run<0, 1, 1>("acc VGPR, partner 64 v_pk_add_f32, prio 2", out, t);
    run<1, 1, 1>("acc AGPR, partner 64 v_pk_add_f32, prio 2", out, t);
    run<0, 2, 0>("acc VGPR, partner 64 v_add_f32", out, t);
    run<1, 2, 0>("acc AGPR, partner 64 v_add_f32", out, t);
    run<0, 2, 1>("acc VGPR, partner 64 v_add_f32, prio 2", out, t);
    run<1, 2, 1>("acc AGPR, partner 64 v_add_f32, prio 2", out, t);
    run<0, 3, 0>("acc VGPR, partner 16 ds_write_b64", out, t);
    run<1, 3, 0>("acc AGPR, partner 16 ds_write_b64", out, t);
    run<0, 3, 1>("acc VGPR, partner 16 ds_write_b64, prio 2", out, t);
    run<1, 3, 1>("acc AGPR, partner 16 ds_write_b64, prio 2", out, t);
    return 0;
}
