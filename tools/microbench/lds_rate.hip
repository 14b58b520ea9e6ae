// Dev microbenchmark: cycles of the CU's LDS pipe per wave64 LDS instruction on gfx950 (16 wavefronts per CU, one VALU add per LDS op, so
// the VALU needs 1 cycle of the CU per op and anything above that is the LDS).
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench/lds_rate.hip -o tools/microbench/bin/lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X, o) X(o) X(o + 1) X(o + 2) X(o + 3) X(o + 4) X(o + 5) X(o + 6) X(o + 7)
#define REP32(X) REP8(X, 0) REP8(X, 8) REP8(X, 16) REP8(X, 24)

template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, int iters, int stride) {
    extern __shared__ unsigned int lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16 * 2048; i += blockDim.x) lds[i] = i;
    __syncthreads();
    // per wave 8 KB; lane address pattern: stride bytes between lanes
    const unsigned int base = (unsigned int)(uintptr_t)(__attribute__((address_space(3))) unsigned int*)lds + wave * 8192u + (unsigned int)(lane * stride) % 4096u;
    unsigned int acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        unsigned int v[32];
        unsigned int v2[32];
#define RD_U8(i) asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(v[i]) : "v"(base), "n"((i) * 4));
#define RD_U16(i) asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(v[i]) : "v"(base), "n"((i) * 4));
#define RD_B32(i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[i]) : "v"(base), "n"((i) * 4));
#define RD2_B32(i) asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(*(unsigned long long*)&v2[(i) & 30]) : "v"(base), "n"((i) & 127), "n"(((i) + 1) & 127));
#define WR_B8(i) asm volatile("ds_write_b8 %0, %1 offset:%2" :: "v"(base), "v"(acc), "n"((i) * 4));
#define WR_B32(i) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(base), "v"(acc), "n"((i) * 4));
#define ADD_U32(i) asm volatile("ds_add_u32 %0, %1 offset:%2" :: "v"(base), "v"(acc), "n"((i) * 4));
        if (MODE == 0) { REP32(RD_U8) }
        if (MODE == 1) { REP32(RD_U16) }
        if (MODE == 2) { REP32(RD_B32) }
        if (MODE == 3) { REP32(RD2_B32) }
        if (MODE == 4) { REP32(WR_B8) }
        if (MODE == 5) { REP32(WR_B32) }
        if (MODE == 6) { REP32(ADD_U32) }
        asm volatile("s_waitcnt lgkmcnt(0)");
        if (MODE <= 2) {
#pragma unroll
            for (int i = 0; i < 32; ++i) acc += v[i];
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 32; ++i) acc += v2[i];
        } else {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc) : "v"(base));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE>
static void run(const char* name, int stride, unsigned long long* dout) {
    const int iters = 500, waves = 16;
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(waves * 64), waves * 8192, 0, dout, iters, stride);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 16);
    hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) s += (double)h[b * 16 + w];
    const double perOp = s / (256.0 * waves) / (iters * 32.0);
    printf("%-16s lane stride %3d B: %6.1f ticks per op per wave = %5.2f ticks of the CU per LDS instruction (16 wavefronts)\n", name, stride, perOp, perOp / waves);
}

int main() {
    unsigned long long* dout;
    hipMalloc(&dout, 256 * 16 * 8);
    for (int stride : {4, 2, 1}) {
        run<0>("ds_read_u8", stride, dout);
        run<1>("ds_read_u16", stride == 1 ? 2 : stride, dout);
        run<2>("ds_read_b32", 4, dout);
        run<3>("ds_read2_b32", 4, dout);
        run<4>("ds_write_b8", stride, dout);
        run<5>("ds_write_b32", 4, dout);
        run<6>("ds_add_u32", 4, dout);
    }
    return 0;
}
