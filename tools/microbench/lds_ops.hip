// Dev microbenchmark: LDS instruction cost on gfx950 as seen by one CU with W wavefronts (the pre-filter's histogram questions).
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench/lds_ops.hip -o tools/microbench/bin/lds_ops ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) unsigned int lds_u32;
typedef __attribute__((address_space(3))) unsigned short lds_u16;
typedef __attribute__((address_space(3))) unsigned char lds_u8;

template <int MODE, bool LIGHT>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, const unsigned int* seed, int iters) {
    extern __shared__ unsigned int lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // per wave 8 KB: [64 bins][32 dwords]
    const unsigned int base = (unsigned int)(uintptr_t)(lds_u32*)lds + wave * 8192u;
    unsigned int r = seed[threadIdx.x] | 1u;
    for (int i = lane; i < 2048; i += 64) lds[wave * 2048 + i] = 0;
    __syncthreads();
    unsigned int acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            // cheap address variation (one VALU op): the LDS pipe, not the VALU, should be the limit
            r += 0x9E3779B9u;
            const unsigned int bin = (LIGHT ? (r >> 26) : ((r * 1664525u + 1013904223u) >> 26));
            if (MODE == 0) {        // ds_add_u32, lanes l / l+32 share a dword (the pre-filter's layout)
                const unsigned int a = base + (bin << 7) + (lane & 31) * 4;
                __hip_atomic_fetch_add((lds_u32*)(uintptr_t)a, 1u << (16 * (lane >> 5)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            } else if (MODE == 1) { // ds_add_u32, every lane its own dword (lanes l, l+32: same bank, different rows)
                const unsigned int a = base + ((bin & 31) << 8) + lane * 4;
                __hip_atomic_fetch_add((lds_u32*)(uintptr_t)a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            } else if (MODE == 2) { // ds_write_b32 to the same addresses
                const unsigned int a = base + (bin << 7) + (lane & 31) * 4;
                *(lds_u32*)(uintptr_t)a = r;
            } else if (MODE == 3) { // ds_read_u8
                const unsigned int a = base + (bin << 7) + (lane & 31) * 4 + (lane >> 5) * 2;
                acc += *(lds_u8*)(uintptr_t)a;
            } else if (MODE == 4) { // ds_read_b32 conflict-free
                const unsigned int a = base + (bin << 7) + (lane & 31) * 4;
                acc += *(lds_u32*)(uintptr_t)a;
            } else if (MODE == 5) { // no LDS: the address arithmetic alone
                acc += (bin << 7) + r;
            } else if (MODE == 7) { // ds_read_b128 (16 B per lane, conflict-free)
                const unsigned int a = base + ((bin & 7) << 10) + lane * 16;
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 v = *(__attribute__((address_space(3))) u32x4*)(uintptr_t)a;
                acc += v.x ^ v.y ^ v.z ^ v.w;
            } else if (MODE == 8) { // ds_read_u16
                const unsigned int a = base + (bin << 7) + (lane & 31) * 4 + (lane >> 5) * 2;
                acc += *(lds_u16*)(uintptr_t)a;
            } else if (MODE == 9) { // ds_write_b8
                const unsigned int a = base + (bin << 7) + (lane & 31) * 4 + (lane >> 5) * 2;
                *(lds_u8*)(uintptr_t)a = (unsigned char)r;
            } else if (MODE == 6) { // ds_add_u32 all lanes distinct banks over 64 dwords rows (256 B rows, one wave per row set)
                const unsigned int a = base + ((bin & 31) << 8) + (lane ^ (bin & 32)) * 4;
                __hip_atomic_fetch_add((lds_u32*)(uintptr_t)a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE, bool LIGHT = true>
static void run(const char* name, int waves, unsigned long long* dout, const unsigned int* dseed) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<MODE, LIGHT>), dim3(256), dim3(waves * 64), waves * 8192, 0, dout, dseed, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 16);
    hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; int n = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) { s += (double)h[b * 16 + w]; ++n; }
    const double perOp = s / n / (iters * 16.0);
    printf("%-34s waves/CU %2d: %.1f ticks per op per wave -> %.2f ticks of the CU per op\n", name, waves, perOp, perOp / waves);
}

int main() {
    unsigned long long* dout; unsigned int* dseed;
    hipMalloc(&dout, 256 * 16 * 8); hipMalloc(&dseed, 1024 * 4);
    std::vector<unsigned int> seed(1024);
    for (int i = 0; i < 1024; ++i) seed[i] = 2654435761u * (i + 1);
    hipMemcpy(dseed, seed.data(), 4096, hipMemcpyHostToDevice);
    for (int waves : {1, 8, 16}) {
        run<5, false>("no LDS, LCG address math", waves, dout, dseed);
        run<0, false>("ds_add_u32 shared dword + LCG", waves, dout, dseed);
        run<5>("no LDS (light address math)", waves, dout, dseed);
        run<7>("ds_read_b128", waves, dout, dseed);
        run<8>("ds_read_u16", waves, dout, dseed);
        run<9>("ds_write_b8", waves, dout, dseed);
        run<0>("ds_add_u32 shared dword l/l+32", waves, dout, dseed);
        run<1>("ds_add_u32 own dword, 2 rows", waves, dout, dseed);
        run<6>("ds_add_u32 own dword, 64 banks?", waves, dout, dseed);
        run<2>("ds_write_b32", waves, dout, dseed);
        run<3>("ds_read_u8", waves, dout, dseed);
        run<4>("ds_read_b32", waves, dout, dseed);
    }
    return 0;
}
