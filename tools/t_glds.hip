// Probe: does an out-of-range `buffer_load_dwordx4 ... lds` lane write zeros to LDS (or leave it untouched)?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* src, float* dst, int n) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 4];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = -7.0f;  // sentinel
    __syncthreads();
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, n * 4, 0x00020000);
    int voff = threadIdx.x * 16;
    if (threadIdx.x & 1) voff = 0x7ffffff0;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) dst[i] = lds[i];
}
int main() {
    float *src, *dst, h[256], hs[256];
    for (int i = 0; i < 256; ++i) hs[i] = 100.f + i;
    hipMalloc(&src, 1024); hipMalloc(&dst, 1024);
    hipMemcpy(src, hs, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, dst, 256);
    hipMemcpy(h, dst, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 6; ++l) printf("lane %d: %g %g %g %g\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    int ok = 1;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
        float e = (l & 1) ? 0.f : 100.f + l * 4 + j;
        if (h[l * 4 + j] != e) ok = 0;
    }
    printf("zero-fill semantics %s\n", ok ? "CONFIRMED" : "NOT as expected");
    return 0;
}
