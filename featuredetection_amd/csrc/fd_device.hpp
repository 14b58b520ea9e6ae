// featuredetection_amd/csrc/fd_device.hpp -- small wave-level device helpers shared by the patch-filter kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace fd_dev {

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ unsigned char sat_u8_d(double v) {   // saturate_cast<uchar>(double): cvRound, clamp
    const int i = (int)rint(v);
    return (unsigned char)(i < 0 ? 0 : (i > 255 ? 255 : i));
}


// cv::equalizeHist of the n-pixel image `px` (LDS) by one wave; hist/lut: 256 ints each (LDS)
__device__ __forceinline__ void equalize_hist_wave(const unsigned char* px, unsigned char* out, int n, int* hist, int* lut, int lane) {
    for (int i = lane; i < 256; i += 64) hist[i] = 0;
    wave_sync();
    for (int i = lane; i < n; i += 64) atomicAdd(&hist[px[i]], 1);
    wave_sync();
    // lane l owns bins 4l..4l+3
    int c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) c[k] = hist[4 * lane + k];
    const int mine = c[0] + c[1] + c[2] + c[3];
    const unsigned long long nz = __ballot(mine != 0);
    const int l0 = __builtin_ctzll(nz);                        // lane holding the first non-empty bin
    int i0 = 0;
    if (lane == l0) i0 = c[0] ? 0 : (c[1] ? 1 : (c[2] ? 2 : 3));
    i0 = 4 * l0 + __builtin_amdgcn_readlane(i0, l0);
    const int h0 = hist[i0];
    if (h0 == n) {
        for (int i = lane; i < n; i += 64) out[i] = (unsigned char)i0;
        wave_sync();
        return;
    }
    const float scale = (256 - 1.f) / (float)(n - h0);
    // inclusive prefix over lanes of the per-lane totals (integer: order-free)
    int incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    int run = incl - mine;   // sum of all bins before 4*lane
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int bin = 4 * lane + k;
        run += c[k];
        // lut[i0] = 0; lut[i] = saturate(sum_{j in (i0, i]} hist[j] * scale)
        lut[bin] = bin <= i0 ? 0 : (int)sat_u8_d((double)((float)(run - h0) * scale));
    }
    wave_sync();
    for (int i = lane; i < n; i += 64) out[i] = (unsigned char)lut[px[i]];
    wave_sync();
}


// HistEq64Filter::applyTo (HistEq64Filter.cpp:32-125) of the n-pixel image px (LDS) by one wave; hist: 64 ints (LDS).
// The fp32 cdf is summed in bin order (63-step DPP chain: lane l holds cdf[l] from step l on).
__device__ __forceinline__ void histeq64_wave(const unsigned char* px, unsigned char* out, int n, int* hist, int lane) {
    hist[lane] = 0;
    wave_sync();
    for (int i = lane; i < n; i += 64) atomicAdd(&hist[px[i] >> 2], 1);
    wave_sync();
    const float stretch = 255.0f / (float)n;
    const float pdf = (float)hist[lane] * stretch;
    float x = pdf;
#pragma unroll
    for (int t = 1; t < 64; ++t) {
        const float sh = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x138 /* wave_shr:1 */, 0xf, 0xf, true));
        x = sh + pdf;
    }
    wave_sync();
    hist[lane] = (int)(unsigned int)(unsigned char)floor((double)x + 0.5);   // becomes the LUT
    wave_sync();
    for (int i = lane; i < n; i += 64) out[i] = (unsigned char)hist[px[i] >> 2];
    wave_sync();
}

}  // namespace fd_dev
