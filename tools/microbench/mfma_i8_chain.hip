// What does the contraction of one k_wvb_chain2 tile cost a wavefront?  26 x v_mfma_i32_32x32x32_i8 on two alternating accumulators
// (a) operands in registers, (b) B operands from LDS in batches of eight ds_read_b128, (c) plus sixteen 1 KB fragment loads from an
// L2-resident table interleaved, (d) like (c) with one accumulator (dependent chain).  s_memtime ticks of one wavefront, 1..8
// wavefronts per CU.  hipcc --offload-arch=gfx950 -O3 mfma_i8_chain.hip -o bin/mfma_i8_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const v4i* __restrict__ A, unsigned long long* out, int* sink, int reps) {
    __shared__ v4i lds[64 * 28];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 28; i += blockDim.x) lds[i] = v4i{i, i + 1, i + 2, i + 3};
    __syncthreads();
    v16i acc0 = {}, acc1 = {};
    v4i a[16];
    const v4i* Ap = A + lane;
#pragma unroll
    for (int q = 0; q < 16; ++q) a[q] = Ap[q * 64];
    v4i b[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) b[q] = lds[q * 64 + lane];
    unsigned long long total = 0;
    for (int r = 0; r < reps; ++r) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 13; ++q) {
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[q], b[q], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[q], b[(q + 1) & 15], acc1, 0, 0, 0);
            }
        } else if (MODE == 3) {
#pragma unroll
            for (int q = 0; q < 13; ++q) {
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[q], b[q], acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[q], b[(q + 1) & 15], acc0, 0, 0, 0);
            }
        } else {
            const v4i* An = Ap + (size_t)(r + 1) * 16 * 64;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                v4i bb[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) bb[i] = lds[((g * 8 + i) % 28) * 64 + lane];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int q = g * 4 + i;
                    const v4i aa = a[q];
                    if (MODE == 2) a[q] = An[q * 64];
                    if (q < 13) {
                        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(aa, bb[2 * i], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(aa, bb[2 * i + 1], acc1, 0, 0, 0);
                    }
                }
            }
        }
        asm volatile("s_nop 0" : "+v"(acc0), "+v"(acc1));
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        total += t1 - t0;
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    if (s == 0x12345678) *sink = s;
    if (threadIdx.x == 0) out[blockIdx.x] = total;
}

int main() {
    const size_t tableBytes = 64u << 20;
    v4i* A; unsigned long long* out; int* sink;
    hipMalloc(&A, tableBytes); hipMemset(A, 1, tableBytes);
    hipMalloc(&out, 8 * 4096); hipMalloc(&sink, 4);
    const int reps = 200;
    for (int mode = 0; mode < 4; ++mode)
        for (int wg : {1, 256, 512, 1024}) {
            for (int threads : {64, 256}) {
                auto launch = [&](int m) {
                    if (m == 0) hipLaunchKernelGGL(k<0>, dim3(wg), dim3(threads), 0, 0, A, out, sink, reps);
                    if (m == 1) hipLaunchKernelGGL(k<1>, dim3(wg), dim3(threads), 0, 0, A, out, sink, reps);
                    if (m == 2) hipLaunchKernelGGL(k<2>, dim3(wg), dim3(threads), 0, 0, A, out, sink, reps);
                    if (m == 3) hipLaunchKernelGGL(k<3>, dim3(wg), dim3(threads), 0, 0, A, out, sink, reps);
                };
                launch(mode); launch(mode);
                hipDeviceSynchronize();
                std::vector<unsigned long long> h(wg);
                hipMemcpy(h.data(), out, 8 * wg, hipMemcpyDeviceToHost);
                double s = 0; for (auto v : h) s += (double)v;
                printf("mode %d (%s) %4d workgroups x %3d threads: %.0f ticks per 26-MFMA tile\n", mode,
                       mode == 0 ? "registers" : mode == 1 ? "B from LDS" : mode == 2 ? "B from LDS + 16 fragment loads" : "registers, one accumulator",
                       wg, threads, s / wg / reps);
            }
        }
    return 0;
}
