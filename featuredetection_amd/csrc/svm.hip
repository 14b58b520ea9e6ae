// featuredetection_amd/csrc/svm.hip -- kernel SVM scoring (SvmClassifier.cpp:55-60 with
// RbfKernel.hpp:32-108, HistogramIntersectionKernel.hpp:28-93, LinearKernel.hpp:27-29,
// PolynomialKernel.hpp:35-37,73-81).
//
// Two paths:
//  * k_svm_generic: one workgroup per feature vector, one wavefront per support vector, all four
//    kernels, u8 (exact integer SSD / min-sum / dot via v_dot4) or f32 features.  Used for the
//    second cascade stage (a few hundred survivors per image) and for fd_svm_distance_batch.
//  * k_svm_rbf_mfma: the dense stage of config 2 -- N patches x 1024 support vectors x 324 dims
//    as a patch x support-vector contraction on the f32 MFMA pipe (v_mfma_f32_32x32x2_f32), with
//    the RBF epilogue exp(-gamma (|x|^2 + |s|^2 - 2 x.s)) * coeff fused and the sum over support
//    vectors kept in registers; nothing but the 8-byte result per patch goes back to HBM.
//    Both operands are stored "fragment-major" in HBM (see FragLayout below) so that a wave's
//    operand fetch is one fully coalesced 1 KiB access: A (64 patches) is staged once per
//    workgroup in LDS, B (support vectors, L2-resident) streams straight into registers.
// MFMA-bound: 2*d*n_sv flop per patch; peak 157.3 TFLOP/s dense f32 MFMA.
#include "fd_internal.hpp"
#include <algorithm>
#include <cstring>
#include <memory>

// ---- fragment-major operand layout --------------------------------------------------------------
// A matrix [rows][K] (K padded to a multiple of 8 with zeros, rows padded to a multiple of 32) is
// stored as tiles of 32 rows; within a tile, for each group q of 8 consecutive k there are 64
// float4 slots: slot (h*32 + r) holds row r, k = 8q + 4h .. 8q + 4h + 3.  Lane l of a wave reads
// slot l: that is exactly its operand for four consecutive v_mfma_f32_32x32x2_f32 issues (lanes
// 0-31 supply k = 8q+t, lanes 32-63 supply k = 8q+4+t, t = 0..3).
static inline size_t frag_index(int64_t row, int k, int KP) {
    const int64_t tile = row >> 5;
    const int r = (int)(row & 31), q = k >> 3, h = (k >> 2) & 1, t = k & 3;
    return (size_t)tile * 32 * KP + (size_t)q * 256 + (size_t)(h * 32 + r) * 4 + t;
}

struct SvmDev {
    int32_t kernel, nsv, dim, dtype;
    int32_t dpad;            // generic path: row stride in elements (u8: multiple of 16, f32: == dim)
    double p0, p1;
    int32_t degree;
    float bias;
    const void* sv;          // [nsv][dpad]
    const float* coeff;      // [nsv]
    const uint32_t* ss_u32;  // u8 rbf: sum of squares per SV
    int32_t nsvp;            // u8: nsv padded to a multiple of 64
    const uint32_t* svT;     // u8: [dpad/16][nsvp][4]: words 4q..4q+3 of support vector s at svT[(q*nsvp + s)*4 ..] (one 16-byte load per lane)
    const float* coeffP;     // u8: [nsvp], zero padded
    const uint32_t* ssP;     // u8: [nsvp]
    // u8 RBF on the i8 MFMA pipe (k_svm_u8_rbf_mfma): support vectors as s - 128 in the A-operand order of v_mfma_i32_32x32x32_i8,
    // [SV tile of 32][k-step of 32 dims][64 lanes] 16 bytes (lane l: SV l & 31, dims ks * 32 + (l >> 5) * 16 .. + 15, zero padded)
    const void* svA;
    const int32_t* ssShift;  // [nsv32] sum of (s - 128)^2 per support vector
    const double* coeffD;    // [nsv32], zero padded
    int32_t nsv32, KS;       // support vectors padded to 32, k-steps
    // MFMA path (f32 RBF)
    int32_t KP;              // padded feature length (multiple of 8)
    int32_t nsv_pad;         // multiple of 256
    const float* svFrag;     // fragment-major [nsv_pad][KP]
    const float* ss_f32;     // [nsv_pad]
    const float* coeffPad;   // [nsv_pad], zero padded
};

struct fd_svm {
    fd_ctx* ctx;
    SvmDev dev;
    float threshold;
    double logisticA, logisticB;
    DevBuf sv, coeff, ss, svFrag, ssF, coeffPad, svT, coeffP, ssP, svA, ssShift, coeffD;
    DevBuf feat, dist, idx;  // scratch for batch calls
};

namespace {

__device__ __forceinline__ double wave_sum_d(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

__device__ __forceinline__ double powi(double base, int exponent) {  // PolynomialKernel.hpp:73-81
    double tmp = base, ret = 1.0;
    for (int t = exponent; t > 0; t /= 2) {
        if (t % 2 == 1) ret *= tmp;
        tmp = tmp * tmp;
    }
    return ret;
}

// One workgroup (256 threads) per feature vector; wave w handles support vectors w, w+4, ...
// features: [n][dpad] (u8 or f32), optionally gathered through idx (slot indices).
__global__ __launch_bounds__(256) void k_svm_generic(SvmDev m, const void* __restrict__ features, const uint32_t* __restrict__ idx,
                                                     int64_t feat_stride_bytes, double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t item = blockIdx.x;
    const int64_t slot = idx ? (int64_t)idx[item] : item;
    const unsigned char* x = (const unsigned char*)features + slot * feat_stride_bytes;
    const int nbytes = m.dtype == FD_DTYPE_U8 ? m.dpad : m.dim * 4;
    for (int i = threadIdx.x; i < nbytes; i += 256) smem[i] = i < (m.dtype == FD_DTYPE_U8 ? m.dim : nbytes) ? x[i] : 0;
    __syncthreads();
    double acc = 0.0;
    if (m.dtype == FD_DTYPE_U8) {
        const uint32_t* xs = (const uint32_t*)smem;
        const int nd = m.dpad >> 2;
        int xxp = 0;
        for (int i = lane; i < nd; i += 64) xxp = __builtin_amdgcn_udot4(xs[i], xs[i], xxp, false);
        const int xx = __builtin_amdgcn_readfirstlane(wave_sum_i(xxp));
        for (int s = wave; s < m.nsv; s += 4) {
            const uint32_t* sv = (const uint32_t*)((const unsigned char*)m.sv + (size_t)s * m.dpad);
            int part = 0;
            if (m.kernel == FD_KERNEL_HIK) {
                for (int i = lane; i < nd; i += 64) {
                    uint32_t a = xs[i], b = sv[i];
#pragma unroll
                    for (int q = 0; q < 4; ++q) part += min((a >> (8 * q)) & 255u, (b >> (8 * q)) & 255u);
                }
            } else {
                for (int i = lane; i < nd; i += 64) part = __builtin_amdgcn_udot4(xs[i], sv[i], part, false);
            }
            const int tot = wave_sum_i(part);
            if (lane == 0) {
                double kv;
                if (m.kernel == FD_KERNEL_RBF) {
                    const int ssd = xx + (int)m.ss_u32[s] - 2 * tot;  // exact integer SSD (RbfKernel.hpp:78-88)
                    kv = exp(-m.p0 * (double)ssd);
                } else if (m.kernel == FD_KERNEL_HIK) {
                    kv = (double)tot;
                } else if (m.kernel == FD_KERNEL_LINEAR) {
                    kv = (double)tot;
                } else {
                    kv = powi(m.p0 * (double)tot + m.p1, m.degree);
                }
                acc += (double)m.coeff[s] * kv;
            }
        }
    } else {
        const float* xs = (const float*)smem;
        for (int s = wave; s < m.nsv; s += 4) {
            const float* sv = (const float*)m.sv + (size_t)s * m.dim;
            double kv;
            if (m.kernel == FD_KERNEL_RBF) {
                float part = 0.f;
                for (int i = lane; i < m.dim; i += 64) { float df = xs[i] - sv[i]; part += df * df; }
                kv = exp(-m.p0 * (double)wave_sum_f(part));
            } else if (m.kernel == FD_KERNEL_HIK) {
                float part = 0.f;
                for (int i = lane; i < m.dim; i += 64) part += fminf(xs[i], sv[i]);
                kv = (double)wave_sum_f(part);
            } else {
                double part = 0.0;
                for (int i = lane; i < m.dim; i += 64) part += (double)xs[i] * (double)sv[i];
                part = wave_sum_d(part);
                kv = m.kernel == FD_KERNEL_LINEAR ? part : powi(m.p0 * part + m.p1, m.degree);
            }
            if (lane == 0) acc += (double)m.coeff[s] * kv;
        }
    }
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[item] = -(double)m.bias + ((red[0] + red[1]) + (red[2] + red[3]));
}


// u8 features (the HistEq64 patches of the second cascade stage): lane == support vector.  The
// support vectors are stored word-transposed so that a wave's fetch of word i of 64 consecutive
// vectors is one coalesced 256-byte access; every lane finishes its own kernel value (exp in fp64),
// so the transcendental runs 64-wide instead of on one lane per support vector.  Integer SSD /
// min-sum / dot stay exact (RbfKernel.hpp:78-88, HistogramIntersectionKernel.hpp:60-72).
// SU_PB feature vectors per workgroup: each fetched support-vector word is used SU_PB times.  The kernel is bound by the L2 traffic of
// the support vectors (all of them once per workgroup), so large batches take 8 per workgroup (the multi-frame cascade scores
// thousands of patches per launch), small ones 2 to keep enough workgroups in flight; the arithmetic per vector is the same.
template <bool HIK, int SU_PB>
__global__ __launch_bounds__(256) void k_svm_u8_lanes(SvmDev m, const void* __restrict__ features, const uint32_t* __restrict__ idx,
                                                      int64_t feat_stride_bytes, int64_t n, double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double red[SU_PB][4];
    __shared__ int xxs[SU_PB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nd = m.dpad >> 2;
    const int64_t item0 = (int64_t)blockIdx.x * SU_PB;
#pragma unroll
    for (int p = 0; p < SU_PB; ++p) {
        const int64_t item = item0 + p;
        const bool valid = item < n;
        const int64_t slot = valid ? (idx ? (int64_t)idx[item] : item) : 0;
        const unsigned char* x = (const unsigned char*)features + slot * feat_stride_bytes;
        for (int i = threadIdx.x; i < m.dpad; i += 256) smem[p * m.dpad + i] = (valid && i < m.dim) ? x[i] : 0;
    }
    __syncthreads();
    const uint32_t* xs = (const uint32_t*)smem;
    for (int p = wave; p < SU_PB; p += 4) {
        int xxp = 0;
        for (int i = lane; i < nd; i += 64) xxp = __builtin_amdgcn_udot4(xs[p * nd + i], xs[p * nd + i], xxp, false);
        xxp = wave_sum_i(xxp);
        if (lane == 0) xxs[p] = xxp;
    }
    __syncthreads();
    double acc[SU_PB];
#pragma unroll
    for (int p = 0; p < SU_PB; ++p) acc[p] = 0.0;
    const uint4* xs4 = reinterpret_cast<const uint4*>(smem);
    const int nq = nd >> 2;
    for (int s0 = wave * 64; s0 < m.nsvp; s0 += 256) {
        const int s = s0 + lane;
        const uint4* colp = reinterpret_cast<const uint4*>(m.svT) + s;
        int dot[SU_PB];
#pragma unroll
        for (int p = 0; p < SU_PB; ++p) dot[p] = 0;
#pragma unroll 4
        for (int i = 0; i < nq; ++i) {
            const uint4 b = colp[(size_t)i * m.nsvp];
#pragma unroll
            for (int p = 0; p < SU_PB; ++p) {
                const uint4 a = xs4[p * nq + i];
                if (HIK) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        dot[p] += (int)min((a.x >> (8 * q)) & 255u, (b.x >> (8 * q)) & 255u) + (int)min((a.y >> (8 * q)) & 255u, (b.y >> (8 * q)) & 255u) +
                                  (int)min((a.z >> (8 * q)) & 255u, (b.z >> (8 * q)) & 255u) + (int)min((a.w >> (8 * q)) & 255u, (b.w >> (8 * q)) & 255u);
                } else {
                    dot[p] = __builtin_amdgcn_udot4(a.x, b.x, dot[p], false);
                    dot[p] = __builtin_amdgcn_udot4(a.y, b.y, dot[p], false);
                    dot[p] = __builtin_amdgcn_udot4(a.z, b.z, dot[p], false);
                    dot[p] = __builtin_amdgcn_udot4(a.w, b.w, dot[p], false);
                }
            }
        }
        const double cf = (double)m.coeffP[s];
        const int ssv = (int)m.ssP[s];
#pragma unroll
        for (int p = 0; p < SU_PB; ++p) {
            double kv;
            if (m.kernel == FD_KERNEL_RBF) {
                const int ssd = xxs[p] + ssv - 2 * dot[p];
                kv = exp(-m.p0 * (double)ssd);
            } else if (m.kernel == FD_KERNEL_POLY) {
                kv = powi(m.p0 * (double)dot[p] + m.p1, m.degree);
            } else {
                kv = (double)dot[p];
            }
            acc[p] += cf * kv;
        }
    }
#pragma unroll
    for (int p = 0; p < SU_PB; ++p) {
        const double v = wave_sum_d(acc[p]);
        if (lane == 0) red[p][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < SU_PB && item0 + threadIdx.x < n) {
        const int p = threadIdx.x;
        out[item0 + p] = -(double)m.bias + ((red[p][0] + red[p][1]) + (red[p][2] + red[p][3]));
    }
}

// ---- u8 RBF on the i8 MFMA pipe (second cascade stage: HistEq64 patches against the RBF support vectors) -------------------
// RbfKernel.hpp:78-88 on CV_8U data is an exact integer sum of squared differences; (x - s) = (x - 128) - (s - 128), so
// ssd = |x'|^2 + |s'|^2 - 2 x'.s' with the cross term from v_mfma_i32_32x32x32_i8 on the shifted (signed) bytes: exact.  A workgroup
// (8 wavefronts) takes 32 feature vectors: their bytes are staged in LDS once (B operand), wavefront w streams the support-vector
// tiles w, w + 8, ... from L2 (A operand, four fragments in flight), and every lane finishes the 16 kernel values of its patch with
// the fp64 exp of the reference and adds them into its fp64 sum.  The lane-per-support-vector kernel (k_svm_u8_lanes) re-read all
// support vectors per 2 patches (426 KB of L2 traffic per pair) and spent 68 us per 64-frame call; this one reads them once per 32.
typedef int svm_v4i __attribute__((ext_vector_type(4)));
typedef int svm_v16i __attribute__((ext_vector_type(16)));
// SU_W wavefronts per workgroup (8, or 16 when the launch has fewer workgroups than the chip has CUs).  The result does not depend on
// it: every tile of 32 support vectors leaves one partial sum per patch, and the partials are added in a fixed pairwise order.
template <int SU_W>
__global__ __launch_bounds__(64 * SU_W) void k_svm_u8_rbf_mfma(SvmDev m, const void* __restrict__ features, const uint32_t* __restrict__ idx,
                                                               int64_t feat_stride_bytes, int64_t n, double* __restrict__ out,
                                                               const unsigned int* __restrict__ dcount) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (dcount) {   // the number of vectors is on the device (five-stage tail: survivors of the overlap elimination kernel in front);
                    // the launch was sized for n, workgroups past the count have nothing to do
        const int64_t dn = (int64_t)*dcount;
        n = dn < n ? dn : n;
        if ((int64_t)blockIdx.x * 32 >= n) return;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int KS = m.KS, DS = KS * 32 + 16;   // row stride: an odd number of 16-byte slots (conflict-free ds_read_b128 across the rows)
    unsigned char* xs = smem;                                                   // [32][DS] x - 128
    int* xxs = reinterpret_cast<int*>(smem + 32 * DS);                          // [32] |x'|^2
    int64_t* slots = reinterpret_cast<int64_t*>(smem + 32 * DS + 128);          // [32] where the patches are (-1: past the end)
    double* ts = reinterpret_cast<double*>(smem + 32 * DS + 384);               // [tiles][32]: a tile's partial sum for every patch
    const int64_t item0 = (int64_t)blockIdx.x * 32;
    if (threadIdx.x < 32) {   // the slot list may live in host memory (zero-copy): ONE read per patch, not one per staged dword
        const int64_t item = item0 + threadIdx.x;
        slots[threadIdx.x] = item < n ? (idx ? (int64_t)idx[item] : item) : -1;
    }
    __syncthreads();
    {   // stage the 32 feature vectors (gathered through idx), zero padded; dwords where the layout allows
        const bool words = (m.dim & 3) == 0 && (feat_stride_bytes & 3) == 0 && ((uintptr_t)features & 3) == 0;
        const int wpr = DS >> 2;
        if (words) {
            // all of a thread's dwords (seven for a 20 x 20 patch) in flight together, from clamped addresses: one load per trip through the
            // loop was seven dependent memory round trips, a third of the kernel's 25 us (the launch is a chain of latencies, not of work)
            constexpr int GU = 8;
            for (int i0 = threadIdx.x; i0 < 32 * wpr; i0 += 64 * SU_W * GU) {
                uint32_t v[GU];
                bool ok[GU];
#pragma unroll
                for (int k = 0; k < GU; ++k) {
                    const int i = min(i0 + k * 64 * SU_W, 32 * wpr - 1);
                    const int row = i / wpr, col = (i - row * wpr) * 4;
                    const int64_t slot = slots[row];
                    ok[k] = slot >= 0 && col < m.dim;
                    v[k] = *reinterpret_cast<const uint32_t*>((const unsigned char*)features + (ok[k] ? slot * feat_stride_bytes + col : 0));
                }
#pragma unroll
                for (int k = 0; k < GU; ++k) {
                    const int i = i0 + k * 64 * SU_W;
                    if (i < 32 * wpr) {
                        const int row = i / wpr, col = (i - row * wpr) * 4;
                        *reinterpret_cast<uint32_t*>(xs + row * DS + col) = ok[k] ? v[k] ^ 0x80808080u : 0u;
                    }
                }
            }
        } else
        for (int i = threadIdx.x; i < 32 * wpr; i += 64 * SU_W) {
            const int row = i / wpr, col = (i - row * wpr) * 4;
            const int64_t slot = slots[row];
            uint32_t v = 0;
            if (slot >= 0 && col < m.dim) {
                const unsigned char* x = (const unsigned char*)features + slot * feat_stride_bytes + col;
                if (words) {
                    v = *reinterpret_cast<const uint32_t*>(x) ^ 0x80808080u;
                } else {
                    for (int b = 0; b < 4; ++b)
                        if (col + b < m.dim) v |= (uint32_t)(x[b] ^ 0x80u) << (8 * b);
                }
            }
            *reinterpret_cast<uint32_t*>(xs + row * DS + col) = v;
        }
    }
    __syncthreads();
    {   // |x'|^2 of the 32 vectors: P consecutive lanes share a vector (exact integers: any order)
        constexpr int P = 2 * SU_W;
        const int row = threadIdx.x / P, part = threadIdx.x % P;
        const uint32_t* r = reinterpret_cast<const uint32_t*>(xs + row * DS);
        int acc = 0;
        for (int i = part; i < KS * 8; i += P) acc = __builtin_amdgcn_sdot4((int)r[i], (int)r[i], acc, false);
#pragma unroll
        for (int o = P / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (part == 0) xxs[row] = acc;
    }
    __syncthreads();
    const int xxv = xxs[lane & 31];
    const unsigned char* xb = xs + (lane & 31) * DS + (lane >> 5) * 16;
    const svm_v4i* A = reinterpret_cast<const svm_v4i*>(m.svA);
    const int ntiles = m.nsv32 >> 5;
    for (int t = wave; t < ntiles; t += SU_W) {
        const svm_v4i* Ap = A + (size_t)t * KS * 64 + lane;
        svm_v16i acc = {};
        if constexpr (SU_W == 16) {
            // Small launches (a tile or two per wavefront) wait for latency: the tile's operands sixteen k-steps ahead -- a 20 x 20 patch
            // has 13 steps: one L2 round trip per tile instead of one per four steps
            constexpr int AD = 16;
            svm_v4i an[AD];
#pragma unroll
            for (int q = 0; q < AD; ++q) an[q] = Ap[min(q, KS - 1) * 64];
            for (int ks = 0; ks < KS; ks += AD) {
#pragma unroll
                for (int q = 0; q < AD; ++q) {
                    const svm_v4i a = an[q];
                    if (ks + AD < KS) an[q] = Ap[min(ks + AD + q, KS - 1) * 64];
                    if (ks + q < KS) {
                        const svm_v4i b = *reinterpret_cast<const svm_v4i*>(xb + (ks + q) * 32);
                        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
                    }
                }
            }
        } else {   // large launches keep their registers for occupancy: four steps ahead
            svm_v4i an[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) an[q] = Ap[min(q, KS - 1) * 64];
            for (int ks = 0; ks < KS; ks += 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const svm_v4i a = an[q];
                    an[q] = Ap[min(ks + 4 + q, KS - 1) * 64];
                    if (ks + q < KS) {
                        const svm_v4i b = *reinterpret_cast<const svm_v4i*>(xb + (ks + q) * 32);
                        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
                    }
                }
            }
        }
        // acc[r] = x' . s' for support vector t * 32 + row(r) and this lane's patch
        double sum = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int sv = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int ssd = xxv + m.ssShift[sv] - 2 * acc[r];
            sum += m.coeffD[sv] * exp(-m.p0 * (double)ssd);   // padded support vectors have coefficient 0
        }
        sum += __shfl_xor(sum, 32, 64);   // the two halves of the tile's rows
        if (lane < 32) ts[t * 32 + lane] = sum;
    }
    __syncthreads();
    if (threadIdx.x < 32) {   // the tiles' partials of patch threadIdx.x, pairwise: (t0 + t1) + (t2 + t3) ...
        double* c = ts + threadIdx.x;
        for (int step = 1; step < ntiles; step <<= 1)
            for (int t = 0; t + step < ntiles; t += 2 * step) c[t * 32] += c[(t + step) * 32];
        if (item0 + threadIdx.x < n) out[item0 + threadIdx.x] = -(double)m.bias + c[0];
    }
}

// ---- dense RBF stage on the f32 MFMA pipe --------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int RB_THREADS = 512;   // 8 waves: 2 per SIMD (one can run its exp epilogue while the other issues MFMAs)
constexpr int RB_BM = 64;         // patches per workgroup (2 row tiles), A resident in LDS
constexpr int RB_DEPTH = 4;       // B prefetch ring (q-groups in flight per wave)

// xFrag: fragment-major features [npad][KP]; xx: |x|^2 per patch; out: hyperplane distance per patch
__global__ __launch_bounds__(RB_THREADS, 2) void k_svm_rbf_mfma(const float* __restrict__ xFrag, const float* __restrict__ xx,
                                                                 SvmDev m, float negGamma, int64_t npatches,
                                                                 double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int KP = m.KP, Q = KP >> 3;
    float* ldsA = lds;                       // 2 tiles * Q * 256 floats
    float* ldsXX = lds + (size_t)2 * Q * 256;  // 64 floats
    double* ldsRed = (double*)(ldsXX + 64);  // 8 waves * 64 rows
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * RB_BM;

    // stage A: the two 32-row tiles of this workgroup are contiguous in HBM (2*Q KiB)
    {
        const f32x4* src = (const f32x4*)(xFrag + (size_t)(row0 >> 5) * 32 * KP);
        f32x4* dst = (f32x4*)ldsA;
        const int n4 = 2 * Q * 64;
        for (int i = threadIdx.x; i < n4; i += RB_THREADS) dst[i] = src[i];
        if (threadIdx.x < 64) ldsXX[threadIdx.x] = xx[row0 + threadIdx.x];
    }
    __syncthreads();

    double rs0[16], rs1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { rs0[r] = 0.0; rs1[r] = 0.0; }

    const int ntilesB = m.nsv_pad >> 5;
    for (int nt = wave; nt < ntilesB; nt += 8) {
        const f32x4* bsrc = (const f32x4*)(m.svFrag + (size_t)nt * 32 * KP) + lane;
        f32x16 acc0 = {0}, acc1 = {0};
        const f32x4* a0p = (const f32x4*)ldsA + lane;
        const f32x4* a1p = a0p + (size_t)Q * 64;
        // B prefetch ring: RB_DEPTH q-groups in flight per wave.  All loads are unconditional (index
        // clamped to Q-1) so the loop body is branch-free and the compiler keeps counted vmcnt waits.
        f32x4 ring[RB_DEPTH];
#pragma unroll
        for (int i = 0; i < RB_DEPTH; ++i) ring[i] = bsrc[(size_t)min(i, Q - 1) * 64];
        f32x4 a0 = a0p[0], a1 = a1p[0];
        int q = 0;
        // Issue order is pinned with sched_barrier(0): [A LDS reads for step q+1] [8 MFMAs of step q]
        // [B load for step q+DEPTH].  Left to itself the scheduler sinks the ring loads to the end of
        // the body and waits vmcnt(0) at the top (measured: 87 TFLOP/s instead of the MFMA rate).
#define RB_STEP(I, LOADB)                                                                          \
        {                                                                                          \
            const int qn = min(q + I + 1, Q - 1);                                                  \
            const f32x4 a0n = a0p[(size_t)qn * 64];                                                \
            const f32x4 a1n = a1p[(size_t)qn * 64];                                                \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[0], ring[I][0], acc0, 0, 0, 0);          \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[0], ring[I][0], acc1, 0, 0, 0);          \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[1], ring[I][1], acc0, 0, 0, 0);          \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[1], ring[I][1], acc1, 0, 0, 0);          \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[2], ring[I][2], acc0, 0, 0, 0);          \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[2], ring[I][2], acc1, 0, 0, 0);          \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[3], ring[I][3], acc0, 0, 0, 0);          \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[3], ring[I][3], acc1, 0, 0, 0);          \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            /* refill this ring slot only after its last reader has issued: the load can then reuse \
               the same VGPRs (no loop-carried copy, no vmcnt(0) drain at the back-edge) */          \
            if (LOADB) ring[I] = bsrc[(size_t)min(q + I + RB_DEPTH, Q - 1) * 64];                   \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            a0 = a0n;                                                                              \
            a1 = a1n;                                                                              \
        }
        for (; q + RB_DEPTH <= Q; q += RB_DEPTH) {
            RB_STEP(0, true)
            RB_STEP(1, true)
            RB_STEP(2, true)
            RB_STEP(3, true)
        }
        // tail: Q % RB_DEPTH steps; after the main loop step q + i uses ring[i]
        if (q + 0 < Q) RB_STEP(0, false)
        if (q + 1 < Q) RB_STEP(1, false)
        if (q + 2 < Q) RB_STEP(2, false)
#undef RB_STEP
        // epilogue: C/D layout col = lane & 31 (support vector), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (patch)
        const int j = nt * 32 + (lane & 31);
        const float ssj = m.ss_f32[j];
        const double cj = (double)m.coeffPad[j];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const float d0 = (ldsXX[row] + ssj) - 2.f * acc0[r];
            const float d1 = (ldsXX[32 + row] + ssj) - 2.f * acc1[r];
            rs0[r] += cj * (double)__expf(negGamma * fmaxf(d0, 0.f));
            rs1[r] += cj * (double)__expf(negGamma * fmaxf(d1, 0.f));
        }
    }
    // reduce over the 32 support-vector columns held by lanes with equal (lane >> 5)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        double a = rs0[r], b = rs1[r];
        for (int o = 16; o > 0; o >>= 1) {
            a += __shfl_xor(a, o, 64);
            b += __shfl_xor(b, o, 64);
        }
        if ((lane & 31) == 0) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            ldsRed[wave * 64 + row] = a;
            ldsRed[wave * 64 + 32 + row] = b;
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < 8; ++w) s += ldsRed[w * 64 + threadIdx.x];
        const int64_t row = row0 + threadIdx.x;
        if (row < npatches) out[row] = -(double)m.bias + s;
    }
}


// ---- support-vector-stationary variant (feature length <= 8*SVQ) -----------------------------------
// Each wavefront keeps ONE 32-SV fragment tile entirely in registers (Q float4 = up to 168 VGPRs) for
// the whole kernel and streams 32-patch tiles through a double-buffered LDS image filled by
// global_load_lds (LDS-DMA, no VGPR round trip).  C[i = SV][j = patch]: a lane's 16 accumulators are
// 16 support vectors of ONE patch, so the sum over support vectors is in-register; only one
// lane<->lane+32 exchange remains.  Every wave writes its own fp64 partial sum per patch (256 B per
// tile, coalesced; k_sum_partials adds the 8 * ngroups partials): an 8-wave LDS reduction behind the
// per-tile barrier cost 9 % of the kernel (1.55 -> 1.41 ms).  Workgroup = 8 waves = 256 support
// vectors; blockIdx.y selects the 256-SV group.  Running the two waves of a SIMD half a tile out of
// phase (deferred epilogue) was measured to change nothing: the matrix pipe is already 95 % busy at
// the ~2.1 GHz the part sustains under this load (PMC: SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE).

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

template <int Q>   // exact number of 8-wide k groups (KP / 8): keeps the K loop free of branches
__global__ __launch_bounds__(512, 2) void k_svm_rbf_mfma_svs(const float* __restrict__ xFrag, const float* __restrict__ xx, SvmDev m,
                                                             float negGamma, int64_t ntiles, double* __restrict__ partial,
                                                             int64_t npadRows) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int KP = Q * 8;
    constexpr int tileFloats = Q * 256;
    float* const buf0 = lds;
    float* const buf1 = lds + tileFloats;
    float* red = lds + 2 * tileFloats;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nt = blockIdx.y * 8 + wave;        // this wave's support-vector tile
    // stationary operand: Q float4 per lane
    f32x4 sv[Q];
    {
        const f32x4* src = (const f32x4*)(m.svFrag + (size_t)nt * 32 * KP) + lane;
#pragma unroll
        for (int q = 0; q < Q; ++q) sv[q] = src[(size_t)q * 64];
    }
    // epilogue constants of this wave's 32 support vectors live in LDS (keeps the VGPR budget for the
    // stationary operand): svc[wave][0..31] = |s|^2, svc[wave][32..63] = coefficient
    float* svc = red + wave * 64;
    svc[lane] = lane < 32 ? m.ss_f32[nt * 32 + lane] : m.coeffPad[nt * 32 + lane - 32];
    auto issue_tile = [&](int64_t tile, float* dst) {
        // LDS-DMA, 1 KiB per wave instruction; wave w moves q-groups w, w+8, ...  Issued through inline asm so
        // that the compiler does not see a pending LDS write (it would wait vmcnt(0) before every ds_read of
        // the tile being computed); completion is awaited explicitly (vmcnt(0) + barrier) before the buffer
        // is read.  Recipe: cdna_hip_programming.md section 5.7 (M0 = wave-uniform LDS destination).
        const char* g = (const char*)(xFrag + (size_t)tile * 32 * KP) + (size_t)lane * 16;
        for (int q = wave; q < Q; q += 8) {
            const unsigned ldsDst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void_t*)(dst + q * 256));
            const char* gsrc = g + (size_t)q * 1024;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gsrc), "s"(ldsDst) : "memory");
        }
    };
    // epilogue of one tile: RBF values of this wave's 32 support vectors for the 32 patches, summed over the support
    // vectors in fp64 and written as this wave's partial sum (k_sum_partials adds the 8 * ngroups partials)
    double* const myPartial = partial + ((size_t)blockIdx.y * 8 + wave) * npadRows;
    auto epilogue = [&](const f32x16& a, float xxv, int64_t t) {
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);   // support vector of accumulator r
            const float d2 = (xxv + svc[row]) - 2.f * a[r];
            s += (double)svc[32 + row] * (double)__expf(negGamma * fmaxf(d2, 0.f));
        }
        s += __shfl_xor(s, 32, 64);
        if (lane < 32) myPartial[t * 32 + lane] = s;
    };
    int cur = 0;
    int64_t tile = blockIdx.x;
    float xxj = tile < ntiles ? xx[tile * 32 + (lane & 31)] : 0.f;
    if (tile < ntiles) issue_tile(tile, buf0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (; tile < ntiles; tile += gridDim.x) {
        const int64_t next = tile + gridDim.x;
        // |x|^2 of the NEXT tile is fetched one iteration ahead, before the LDS-DMA is queued, so that no
        // compiler-inserted vmcnt wait for it can land behind the DMA inside this iteration's compute
        const float xxn = next < ntiles ? xx[next * 32 + (lane & 31)] : 0.f;
        if (next < ntiles) issue_tile(next, cur ? buf0 : buf1);
        const f32x4* ap = (const f32x4*)(cur ? buf1 : buf0) + lane;
        f32x16 acc = {0};
        f32x4 p = ap[0];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const f32x4 pn = ap[(size_t)(q + 1 < Q ? q + 1 : q) * 64];
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[q][0], p[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[q][1], p[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[q][2], p[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[q][3], p[3], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            p = pn;
        }
        epilogue(acc, xxj, tile);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next tile has landed in LDS (and the partial sums are out)
        __syncthreads();
        xxj = xxn;
        cur ^= 1;
    }
}

// out[i] = -bias + sum over SV groups
__global__ void k_sum_partials(const double* __restrict__ partial, int ngroups, int64_t npadRows, int64_t n, float bias,
                               double* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double s = 0.0;
        int g = 0;
        for (; g + 8 <= ngroups; g += 8) {   // eight loads in flight, added in group order
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = partial[(size_t)(g + k) * npadRows + i];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; g < ngroups; ++g) s += partial[(size_t)g * npadRows + i];
        out[i] = -(double)bias + s;
    }
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------------
bool fd_svm_has_mfma_path(const fd_svm* m) { return m->dev.svFrag != nullptr; }
int fd_svm_KP(const fd_svm* m) { return m->dev.KP; }
float fd_svm_threshold(const fd_svm* m) { return m->threshold; }
int fd_svm_dim(const fd_svm* m) { return m->dev.dim; }
bool fd_svm_is_u8(const fd_svm* m) { return m->dev.dtype == FD_DTYPE_U8; }
double fd_svm_probability(const fd_svm* m, double d) {  // ProbabilisticSvmClassifier.cpp:54-58 (host, libm)
    double fABp = m->logisticA + m->logisticB * d;
    return fABp >= 0 ? std::exp(-fABp) / (1.0 + std::exp(-fABp)) : 1.0 / (1.0 + std::exp(fABp));
}

bool fd_svm_fused_view(const fd_svm* m, FdSvmFusedView* v) {
    if (!m->dev.svFrag || m->dev.kernel != FD_KERNEL_RBF) return false;
    v->svFrag = m->dev.svFrag; v->ss = m->dev.ss_f32; v->coeff = m->dev.coeffPad;
    v->nsv_pad = m->dev.nsv_pad; v->KP = m->dev.KP; v->bias = m->dev.bias; v->gamma = m->dev.p0;
    return true;
}

size_t fd_svm_rbf_lds_bytes(int KP) { return (size_t)2 * (KP / 8) * 256 * 4 + 64 * 4 + 8 * 64 * 8; }

// dense MFMA scoring of npatches feature vectors already on the device in fragment-major layout
// (rows padded to a multiple of 64, pad rows zero).  partial: scratch for the SV-stationary kernel.
void fd_svm_rbf_mfma_launch(fd_ctx* ctx, const fd_svm* m, const float* xFrag, const float* xx, int64_t npatches, double* out) {
    if (!m->dev.svFrag) FD_THROW(FD_ERR_INVALID_ARGUMENT, "SVM has no MFMA path (needs an RBF kernel on f32 vectors)");
    const int KP = m->dev.KP;
    static int variant = -1;
    if (variant < 0) { const char* e = getenv("FD_SVM_KERNEL"); variant = e ? atoi(e) : 2; }
    if (variant == 2 && (KP == 328 || KP == 336)) {   // instantiated sizes (HOG-324 -> 328); others use the A-resident kernel
        fd_svm* mm = const_cast<fd_svm*>(m);
        const int64_t ntiles = (npatches + 31) / 32;
        const int64_t npadRows = ((npatches + 63) / 64) * 64;
        const int ngroups = m->dev.nsv_pad / 256;
        mm->dist.reserve(sizeof(double) * (size_t)ngroups * 8 * npadRows);
        const size_t ldsBytes = (size_t)2 * (KP / 8) * 256 * 4 + 8 * 64 * 4;
        static uint64_t lds41 = 0, lds42 = 0;
        fd_allow_lds(ctx, (const void*)k_svm_rbf_mfma_svs<41>, 160 * 1024, lds41);
        fd_allow_lds(ctx, (const void*)k_svm_rbf_mfma_svs<42>, 160 * 1024, lds42);
        const int gx = (int)std::min<int64_t>(ntiles, std::max(1, ctx->num_cus / ngroups));
        if (KP == 328)
            hipLaunchKernelGGL(k_svm_rbf_mfma_svs<41>, dim3(gx, ngroups), dim3(512), ldsBytes, ctx->stream, xFrag, xx, m->dev,
                               (float)(-m->dev.p0), ntiles, mm->dist.as<double>(), npadRows);
        else
            hipLaunchKernelGGL(k_svm_rbf_mfma_svs<42>, dim3(gx, ngroups), dim3(512), ldsBytes, ctx->stream, xFrag, xx, m->dev,
                               (float)(-m->dev.p0), ntiles, mm->dist.as<double>(), npadRows);
        hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)std::min<int64_t>((npatches + 255) / 256, 2048)), dim3(256), 0, ctx->stream,
                           mm->dist.as<double>(), ngroups * 8, npadRows, npatches, m->dev.bias, out);
        HIP_CHECK(hipGetLastError());
        return;
    }
    const size_t ldsBytes = fd_svm_rbf_lds_bytes(KP);
    if (ldsBytes > 160 * 1024) FD_THROW(FD_ERR_INVALID_ARGUMENT, "feature length %d too large for the MFMA SVM kernel", m->dev.dim);
    static uint64_t lds_allowed = 0;
    fd_allow_lds(ctx, (const void*)k_svm_rbf_mfma, 160 * 1024, lds_allowed);
    const int64_t blocks = (npatches + RB_BM - 1) / RB_BM;
    hipLaunchKernelGGL(k_svm_rbf_mfma, dim3((unsigned)blocks), dim3(RB_THREADS), ldsBytes, ctx->stream, xFrag, xx, m->dev,
                       (float)(-m->dev.p0), npatches, out);
    HIP_CHECK(hipGetLastError());
}

bool fd_svm_u8_mfma_available(const fd_svm* m);
// generic scoring of n device-resident feature vectors (optionally gathered by slot index)
void fd_svm_generic_launch_on(hipStream_t st, const fd_svm* m, const void* dfeat, const uint32_t* didx, int64_t stride_bytes,
                              int64_t n, double* dout);
void fd_svm_generic_launch(fd_ctx* ctx, const fd_svm* m, const void* dfeat, const uint32_t* didx, int64_t stride_bytes,
                           int64_t n, double* dout) {
    fd_svm_generic_launch_on(ctx->stream, m, dfeat, didx, stride_bytes, n, dout);
}
// explicit stream: callable from the worker threads of the batch entry points (touches no context state)
void fd_svm_generic_launch_on(hipStream_t st, const fd_svm* m, const void* dfeat, const uint32_t* didx, int64_t stride_bytes,
                              int64_t n, double* dout) {
    if (n <= 0) return;
    // u8 RBF: k_svm_u8_rbf_mfma wherever fd_svm_u8_mfma_available says so -- ONE predicate for this launcher and for the device tail
    // of the five-stage entry points, so a call never mixes two kernels with different summation orders.  (Its LDS: 32 patches + one
    // partial sum per (tile of 32 support vectors, patch); beyond the 64 KB a launch gets without opting in -- very long vectors with
    // thousands of support vectors -- the lane-per-support-vector kernel takes over.  A 16-wavefront variant made a lone small launch
    // faster, 25 -> 21 us, and cost config 3 a third of its throughput: whole CUs taken from the other calls' kernels; removed.)
    if (fd_svm_u8_mfma_available(m)) {
        const size_t lb = (size_t)32 * (m->dev.KS * 32 + 16) + 384 + sizeof(double) * (size_t)(m->dev.nsv32 >> 5) * 32;
        const unsigned grid = (unsigned)((n + 31) / 32);
        hipLaunchKernelGGL(k_svm_u8_rbf_mfma<8>, dim3(grid), dim3(64 * 8), lb, st, m->dev, dfeat, didx, stride_bytes, n, dout, (const unsigned int*)nullptr);
        HIP_CHECK(hipGetLastError());
        return;
    }
    if (m->dev.dtype == FD_DTYPE_U8) {
        const int pb = n >= 16384 ? 8 : (n >= 8192 ? 4 : 2);   // measured: 4 per workgroup at n ~ 2000-4000 is slower than 2 (fewer workgroups)
        const size_t lb = (size_t)pb * m->dev.dpad;
        if (lb > 64 * 1024) FD_THROW(FD_ERR_INVALID_ARGUMENT, "feature vector too long (%d)", m->dev.dim);
        const unsigned grid = (unsigned)((n + pb - 1) / pb);
#define FD_SVM_U8(H, P) hipLaunchKernelGGL((k_svm_u8_lanes<H, P>), dim3(grid), dim3(256), lb, st, m->dev, dfeat, didx, stride_bytes, n, dout)
        if (m->dev.kernel == FD_KERNEL_HIK) {
            if (pb == 8) FD_SVM_U8(true, 8); else if (pb == 4) FD_SVM_U8(true, 4); else FD_SVM_U8(true, 2);
        } else {
            if (pb == 8) FD_SVM_U8(false, 8); else if (pb == 4) FD_SVM_U8(false, 4); else FD_SVM_U8(false, 2);
        }
#undef FD_SVM_U8
        return;
    }
    const size_t ldsBytes = m->dev.dtype == FD_DTYPE_U8 ? (size_t)m->dev.dpad : (size_t)m->dev.dim * 4;
    if (ldsBytes > 64 * 1024) FD_THROW(FD_ERR_INVALID_ARGUMENT, "feature vector too long (%d)", m->dev.dim);
    hipLaunchKernelGGL(k_svm_generic, dim3((unsigned)n), dim3(256), ldsBytes, st, m->dev, dfeat, didx, stride_bytes, dout);
    HIP_CHECK(hipGetLastError());
}

// The u8 RBF stage with the vector count on the device (the five-stage tail: overlap elimination and this launch are queued behind the
// cascade without a host round trip).  The launch covers nmax vectors; min(*dcount, nmax) are scored.  false: this model has no
// k_svm_u8_rbf_mfma tables (the caller keeps the host-driven path).
bool fd_svm_u8_mfma_available(const fd_svm* m) {
    const size_t lb = (size_t)32 * (m->dev.KS * 32 + 16) + 384 + sizeof(double) * (size_t)(m->dev.nsv32 >> 5) * 32;
    return m->dev.dtype == FD_DTYPE_U8 && m->dev.kernel == FD_KERNEL_RBF && m->dev.svA && lb <= 64 * 1024;
}
void fd_svm_u8_mfma_launch_counted(hipStream_t st, const fd_svm* m, const void* dfeat, const uint32_t* didx, int64_t stride_bytes, int64_t nmax,
                                   const unsigned int* dcount, double* dout) {
    if (nmax <= 0) return;
    const size_t lb = (size_t)32 * (m->dev.KS * 32 + 16) + 384 + sizeof(double) * (size_t)(m->dev.nsv32 >> 5) * 32;
    // a frame's worth of vectors (a handful of workgroups: the launch is a chain of latencies) takes sixteen wavefronts per workgroup,
    // half the support-vector tiles per wavefront and a tile's operands requested at once; the sums are the same (fixed pairwise order)
    if (nmax <= 32 * 64)
        hipLaunchKernelGGL(k_svm_u8_rbf_mfma<16>, dim3((unsigned)((nmax + 31) / 32)), dim3(64 * 16), lb, st, m->dev, dfeat, didx, stride_bytes, nmax, dout, dcount);
    else
        hipLaunchKernelGGL(k_svm_u8_rbf_mfma<8>, dim3((unsigned)((nmax + 31) / 32)), dim3(64 * 8), lb, st, m->dev, dfeat, didx, stride_bytes, nmax, dout, dcount);
    HIP_CHECK(hipGetLastError());
}

extern "C" {

int fd_svm_create(fd_ctx* ctx, const fd_svm_model* md, fd_svm** out) {
    return fd_guard(ctx, [&] {
        if (!ctx || !md || !out) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_svm_create: NULL argument");
        if (md->kernel < 0 || md->kernel > 3) FD_THROW(FD_ERR_RUNTIME, "SvmClassifier: Invalid kernel type: %d", md->kernel);
        if (md->dtype != FD_DTYPE_U8 && md->dtype != FD_DTYPE_F32)
            FD_THROW(FD_ERR_INVALID_ARGUMENT, "SvmClassifier: support vectors must be u8 or f32");
        if (md->num_sv < 1 || md->dim < 1 || !md->support_vectors || !md->coefficients)
            FD_THROW(FD_ERR_INVALID_ARGUMENT, "SvmClassifier: empty model");
        HIP_CHECK(hipSetDevice(ctx->device));
        std::unique_ptr<fd_svm> m(new fd_svm());
        m->ctx = ctx;
        SvmDev& d = m->dev;
        std::memset(&d, 0, sizeof(d));
        d.kernel = md->kernel; d.nsv = md->num_sv; d.dim = md->dim; d.dtype = md->dtype;
        d.p0 = md->p0; d.p1 = md->p1; d.degree = (int)md->p2; d.bias = md->bias;
        m->threshold = md->threshold;
        m->logisticA = md->logistic_a;
        m->logisticB = md->logistic_b;
        auto up = [&](DevBuf& b, const void* src, size_t bytes) {
            b.reserve(std::max<size_t>(bytes, 16));
            HIP_CHECK(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
        };
        up(m->coeff, md->coefficients, sizeof(float) * d.nsv);
        d.coeff = m->coeff.as<float>();
        if (d.dtype == FD_DTYPE_U8) {
            d.dpad = (d.dim + 15) & ~15;   // whole 16-byte groups: the lane == support-vector kernel fetches four words per load
            std::vector<uint8_t> sv((size_t)d.nsv * d.dpad, 0);
            std::vector<uint32_t> ss(d.nsv, 0);
            const uint8_t* src = (const uint8_t*)md->support_vectors;
            for (int s = 0; s < d.nsv; ++s)
                for (int k = 0; k < d.dim; ++k) {
                    uint8_t v = src[(size_t)s * d.dim + k];
                    sv[(size_t)s * d.dpad + k] = v;
                    ss[s] += (uint32_t)v * v;
                }
            up(m->sv, sv.data(), sv.size());
            up(m->ss, ss.data(), sizeof(uint32_t) * ss.size());
            d.sv = m->sv.p;
            d.ss_u32 = m->ss.as<uint32_t>();
            {   // word-transposed copy for the lane == support-vector kernel
                d.nsvp = (d.nsv + 63) & ~63;
                const int nd = d.dpad >> 2;
                std::vector<uint32_t> svT((size_t)nd * d.nsvp, 0u), ssP(d.nsvp, 0u);
                std::vector<float> cP(d.nsvp, 0.f);
                for (int sI = 0; sI < d.nsv; ++sI) {
                    for (int i = 0; i < nd; ++i) {
                        uint32_t wv;
                        std::memcpy(&wv, &sv[(size_t)sI * d.dpad + 4 * i], 4);
                        svT[((size_t)(i >> 2) * d.nsvp + sI) * 4 + (i & 3)] = wv;
                    }
                    ssP[sI] = ss[sI];
                    cP[sI] = md->coefficients[sI];
                }
                up(m->svT, svT.data(), sizeof(uint32_t) * svT.size());
                up(m->ssP, ssP.data(), sizeof(uint32_t) * ssP.size());
                up(m->coeffP, cP.data(), sizeof(float) * cP.size());
                d.svT = m->svT.as<uint32_t>(); d.ssP = m->ssP.as<uint32_t>(); d.coeffP = m->coeffP.as<float>();
            }
            if (d.kernel == FD_KERNEL_RBF && d.dim <= 1536) {   // operand tables of k_svm_u8_rbf_mfma (LDS: 32 rows of KS * 32 + 16 bytes)
                d.KS = (d.dim + 31) / 32;
                d.nsv32 = (d.nsv + 31) & ~31;
                std::vector<int8_t> A((size_t)(d.nsv32 / 32) * d.KS * 64 * 16, 0);
                std::vector<int32_t> ssS(d.nsv32, 0);
                std::vector<double> cD(d.nsv32, 0.0);
                for (int sI = 0; sI < d.nsv; ++sI) {
                    const int tile = sI >> 5, row = sI & 31;
                    for (int k = 0; k < d.dim; ++k) {
                        const int v = (int)src[(size_t)sI * d.dim + k] - 128;
                        ssS[sI] += v * v;
                        const int ks = k >> 5, h = (k >> 4) & 1, t = k & 15;
                        A[((((size_t)tile * d.KS + ks) * 64) + (size_t)(h * 32 + row)) * 16 + t] = (int8_t)v;
                    }
                    cD[sI] = (double)md->coefficients[sI];
                }
                up(m->svA, A.data(), A.size());
                up(m->ssShift, ssS.data(), sizeof(int32_t) * ssS.size());
                up(m->coeffD, cD.data(), sizeof(double) * cD.size());
                d.svA = m->svA.p; d.ssShift = m->ssShift.as<int32_t>(); d.coeffD = m->coeffD.as<double>();
            }
        } else {
            d.dpad = d.dim;
            up(m->sv, md->support_vectors, sizeof(float) * (size_t)d.nsv * d.dim);
            d.sv = m->sv.p;
            if (d.kernel == FD_KERNEL_RBF) {
                d.KP = (d.dim + 7) & ~7;
                d.nsv_pad = (d.nsv + 255) & ~255;
                std::vector<float> frag((size_t)d.nsv_pad * d.KP, 0.f), ss(d.nsv_pad, 0.f), cp(d.nsv_pad, 0.f);
                const float* src = (const float*)md->support_vectors;
                for (int s = 0; s < d.nsv; ++s) {
                    double acc = 0;
                    for (int k = 0; k < d.dim; ++k) {
                        float v = src[(size_t)s * d.dim + k];
                        frag[frag_index(s, k, d.KP)] = v;
                        acc += (double)v * v;
                    }
                    ss[s] = (float)acc;
                    cp[s] = md->coefficients[s];
                }
                up(m->svFrag, frag.data(), sizeof(float) * frag.size());
                up(m->ssF, ss.data(), sizeof(float) * ss.size());
                up(m->coeffPad, cp.data(), sizeof(float) * cp.size());
                d.svFrag = m->svFrag.as<float>();
                d.ss_f32 = m->ssF.as<float>();
                d.coeffPad = m->coeffPad.as<float>();
            }
        }
        *out = m.release();
    });
}

void fd_svm_destroy(fd_svm* m) { delete m; }

int fd_svm_distance_batch(fd_ctx* ctx, const fd_svm* m_, const void* features, int64_t n, double* out_distance) {
    return fd_guard(ctx, [&] {
        if (!ctx || !m_ || (n > 0 && (!features || !out_distance))) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_svm_distance_batch: NULL argument");
        if (n <= 0) return;
        fd_svm* m = const_cast<fd_svm*>(m_);
        HIP_CHECK(hipSetDevice(ctx->device));
        const size_t es = m->dev.dtype == FD_DTYPE_U8 ? 1 : 4;
        const size_t stride = (size_t)m->dev.dim * es;
        m->feat.reserve(stride * (size_t)n + 16);
        m->dist.reserve(sizeof(double) * (size_t)n);
        HIP_CHECK(hipMemcpyAsync(m->feat.p, features, stride * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        fd_svm_generic_launch(ctx, m, m->feat.p, nullptr, (int64_t)stride, n, m->dist.as<double>());
        HIP_CHECK(hipMemcpyAsync(out_distance, m->dist.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

// Test hook (include/fd_hip_bench.h): the u8 RBF stage through BOTH instantiations of k_svm_u8_rbf_mfma for the same vectors -- <8>
// (every launch of the survivors' SVM stage) and <16> (a single frame's launch over all its WVM positives; the launcher picks it
// for up to 2048 vectors).  The five-stage entry points rely on the two giving identical bits (one partial per tile of 32 support
// vectors, added in a fixed pairwise order).
int fd_debug_svm_u8_both(fd_ctx* ctx, const fd_svm* m_, const uint8_t* features, int64_t n, double* out8, double* out16) {
    return fd_guard(ctx, [&] {
        if (!ctx || !m_ || n < 1 || !features || !out8 || !out16) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_debug_svm_u8_both: bad argument");
        fd_svm* m = const_cast<fd_svm*>(m_);
        if (!fd_svm_u8_mfma_available(m)) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_debug_svm_u8_both: the model has no u8 RBF MFMA path");
        HIP_CHECK(hipSetDevice(ctx->device));
        const size_t stride = (size_t)m->dev.dim;
        m->feat.reserve(stride * (size_t)n + 16);
        m->dist.reserve(sizeof(double) * (size_t)n * 2);
        HIP_CHECK(hipMemcpyAsync(m->feat.p, features, stride * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        const size_t lb = (size_t)32 * (m->dev.KS * 32 + 16) + 384 + sizeof(double) * (size_t)(m->dev.nsv32 >> 5) * 32;
        const unsigned grid = (unsigned)((n + 31) / 32);
        hipLaunchKernelGGL(k_svm_u8_rbf_mfma<8>, dim3(grid), dim3(64 * 8), lb, ctx->stream, m->dev, m->feat.p, (const uint32_t*)nullptr, (int64_t)stride, n,
                           m->dist.as<double>(), (const unsigned int*)nullptr);
        hipLaunchKernelGGL(k_svm_u8_rbf_mfma<16>, dim3(grid), dim3(64 * 16), lb, ctx->stream, m->dev, m->feat.p, (const uint32_t*)nullptr, (int64_t)stride, n,
                           m->dist.as<double>() + n, (const unsigned int*)nullptr);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(out8, m->dist.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipMemcpyAsync(out16, m->dist.as<double>() + n, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

}  // extern "C"
