// Is v_cvt_rpi_i32_f32 exactly floor(x + 0.5) (no intermediate fp32 rounding of x + 0.5) on gfx950?  Checked for every float in [0, 1024)
// against floor(x) + (fract(x) >= 0.5), the form k_wvm_prefilter used before (HistEq64Filter.cpp:118: (uchar)floor(LUTeq + 0.5) in double).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* bad, unsigned int* firstBad) {
    const unsigned int hi = 0x44800000u;   // 1024.0f
    for (unsigned int b = blockIdx.x * blockDim.x + threadIdx.x; b < hi; b += gridDim.x * blockDim.x) {
        const float x = __uint_as_float(b);
        int r;
        asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
        const int ref = (int)(unsigned int)x + (__builtin_amdgcn_fractf(x) >= 0.5f ? 1 : 0);
        const int refd = (int)floor((double)x + 0.5);
        if (r != ref || r != refd) { atomicAdd(bad, 1ull); atomicMin(firstBad, b); }
    }
}
int main() {
    unsigned long long* bad; unsigned int* fb;
    hipMalloc(&bad, 8); hipMalloc(&fb, 4);
    hipMemset(bad, 0, 8); hipMemset(fb, 0xff, 4);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, bad, fb);
    unsigned long long h; unsigned int f;
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, fb, 4, hipMemcpyDeviceToHost);
    printf("v_cvt_rpi_i32_f32 vs floor(x + 0.5) over all floats in [0, 1024): %llu mismatches (first bits 0x%08x)\n", h, f);
    return h != 0;
}
