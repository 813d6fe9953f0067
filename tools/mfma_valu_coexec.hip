// Microbenchmark: what a partner wave's VALU / LDS / VMEM stream gets while the older wave of its SIMD issues fp32 32x32x2 MFMAs
// back to back - with the accumulators in VGPRs (what hipcc picks when the kernel fits 256 VGPRs) and in AGPRs.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_coexec.hip -o tools/coexec.bin && ./tools/coexec.bin
// Measured on MI355X (cycles per 64 MFMAs / per partner iteration): the MFMA wave keeps 4097-4119 cycles whatever the partner
// does, accumulators in VGPRs or AGPRs alike; 64 v_pk_add_f32 or v_add_f32 of the partner take 4390-4450 cycles (ONE per MFMA,
// s_setprio or not); 16 ds_write_b64 + wait 386 cycles; 16 ds_read_b64 + wait 4100 and 16 global loads + wait 5120 - what
// returns data to VGPRs waits for the end of the unrolled MFMA sequence.  Reading: a 32x32x2 fp32 MFMA writes 16 result rows
// in its 64 cycles, i.e. the SIMD's VGPR write port is the resource it saturates; every other instruction that writes VGPRs
// (VALU result, LDS / memory return) either takes one of its own wave's gaps (about 4 cycles per 256-byte register row in
// the MFMA wave's own stream) or waits behind the partner's stream.  DESIGN.md 3 prices the kernels with this.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int AGPR, int PARTNER, int PRIO>
__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* t, int iters) {
    __shared__ float lds[4096];  // the asm LDS accesses below address it from 0
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
            } else if (PARTNER == 3) {  // 16 ds_write_b64 + wait
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"((unsigned)(lane * 8)), "v"(v[i & 7]), "n"(i * 512) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if (PARTNER == 4) {  // 16 ds_read_b64 + wait
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[i]) : "v"((unsigned)(lane * 8)), "n"(i * 512) : "memory");
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[i]) : "v"((unsigned)(lane * 8)), "n"(4096 + i * 512) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if (PARTNER == 5) {  // 16 global loads (L2-resident) + wait
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=v"(v[i]) : "v"(out + lane * 2), "n"(i * 512) : "memory");
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=v"(v[i]) : "v"(out + 4096 + lane * 2), "n"(i * 512) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
    (void)hipDeviceSynchronize();
    unsigned long long h[2];
    (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s mfma wave: %7.1f cycles / 64 MFMAs   partner: %7.1f cycles / iteration\n", name, (double)h[0] / iters, (double)h[1] / iters);
}

int main() {
    float* out; unsigned long long* t;
    (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMalloc(&t, 256 * 2 * 8);
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
    run<0, 3, 0>("acc VGPR, partner 16 ds_write_b64 + wait", out, t);
    run<0, 3, 1>("acc VGPR, partner 16 ds_write_b64 + wait, prio 2", out, t);
    run<0, 4, 0>("acc VGPR, partner 16 ds_read_b64 + wait", out, t);
    run<0, 4, 1>("acc VGPR, partner 16 ds_read_b64 + wait, prio 2", out, t);
    run<0, 5, 0>("acc VGPR, partner 16 global_load_dwordx2 + wait", out, t);
    run<0, 5, 1>("acc VGPR, partner 16 global_load_dwordx2 + wait, prio 2", out, t);
    return 0;
}
