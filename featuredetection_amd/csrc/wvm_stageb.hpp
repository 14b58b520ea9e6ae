// featuredetection_amd/csrc/wvm_stageb.hpp -- stage B of the WVM cascade as dense contractions (included by wvm.hip only, inside
// its anonymous namespace, after wvm_locate / wvm_prepare).
//
// The windows that reach stage B (the dense pre-filter's queue, or the survivors of the exact stage A) are evaluated exactly from
// level 0, like before, but no longer one window per wavefront with rect lookups and LDS atomics.  WvmClassifier.cpp:191-346 per
// level k (class n = k mod numPer):
//     S_v   = sum of the rects of grey value v >= 1 on the integral image           -- exact integers (fd_wvm_create checks < 2^24)
//     sum_xp = sum_v S_v val[v] (fp64, in order) + S_0 val[0] + u[n];  u[n] = (float)sum_xp
//     K_k   = (float)exp(-basis * (sxx - 2 sum_xp + pp[k]))
//     res_k = -bias + sum_{p <= k} w[k][p] K_p   (fp32, in order);  leave at the first k with res_k < thr[k]
// is regrouped into three data-parallel steps per phase (a phase = a range of generations; survivors are packed densely between
// phases so that late-rejecting models keep their early exit):
//   k_wvb_chain  S_v = x . M_{k,v} for ALL rows (k, v) of a class as an int8 contraction on v_mfma_i32_32x32x32_i8 (M = how many
//                rects of (k, v) cover a pixel; x - 128 as int8, 128 * sum(M) added back): exact.  Then lane == window: the fp64
//                chain and exp of the class's levels in the reference's order, one wavefront per (64 windows, class); the classes
//                are independent of each other.  Writes K[level][window].
//   k_wvb_sums   lane == window, one wavefront per (64 windows, 8 consecutive rows k): res_k in the reference's term order, the
//                weights are scalar operands, every K_p load (256 B, coalesced) feeds 8 multiply-adds.  Writes R[level][window].
//   k_wvb_exit   lane == window: first level of the phase with res < thr (or the last used level) -> outputs / positives;
//                survivors are appended to the next phase's dense list together with their state (patch, u, K history).
//   k_wvb_prepare (once, in front): HistEq64 + sums of the queued windows (wave == window, wvm_prepare) -> state set 0.
// Everything is bit-identical to the rectangle-sum formulation: integer sums are exact, the fp64 / fp32 chains keep their order.
#pragma once

#define WVB_CONST(T, p) ((const __attribute__((address_space(4))) T*)(uintptr_t)(p))

// windows alive at the start of a phase (wave-uniform), never more than the state holds
__device__ __forceinline__ unsigned int wvb_count(const unsigned int* countPtr, const WvbState& s) {
    const unsigned int c = (unsigned int)__builtin_amdgcn_readfirstlane((int)*countPtr);
    return (int64_t)c > s.cap ? (unsigned int)s.cap : c;
}

template <int PW_, int PH_, bool RAW>
__global__ __launch_bounds__(256) void k_wvb_prepare(const uint8_t* __restrict__ arena, WinTable wt, WvmDev m, WvbDev mv, WvbState s, const int64_t* q,
                                                     const unsigned int* qcount) {
    __shared__ WaveLds<PW_, PH_> lds[4];
    __shared__ int64_t sFirst[WVM_MAX_LAYERS];
    constexpr int RHMAX = Geo<PW_, PH_>::RHMAX;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WaveLds<PW_, PH_>& L = lds[wave];
    const Geo<PW_, PH_> g(m, lane);
    if (!RAW) {
        if (threadIdx.x < WVM_MAX_LAYERS) sFirst[threadIdx.x] = (int)threadIdx.x < wt.n ? wt.l[threadIdx.x].first : INT64_MAX;
        __syncthreads();
    }
    const unsigned int n = wvb_count(qcount, s);
    for (unsigned int pos = blockIdx.x * 4 + wave; pos < n; pos += gridDim.x * 4) {
        const int64_t wid = q[pos];
        int srcStride;
        const uint8_t* src = wvm_locate<RAW>(arena, wt, sFirst, wid, lane, g.pw, g.d, srcStride);
        unsigned int px[RHMAX];
        float sxx;
        int sx_total;
        wvm_prepare<PW_, PH_, RAW>(g, src, srcStride, m.stretch, lane, L.hist, L.ii, px, sxx, sx_total);
        int8_t* xr = s.X[0] + (size_t)pos * mv.dstride;
        if (g.colok) {
#pragma unroll
            for (int j = 0; j < RHMAX; ++j)
                if (g.rowok(j)) xr[(g.r0 + j) * g.pw + g.col] = (int8_t)(px[j] ^ 0x80u);   // x - 128
        }
        for (int i = g.d + lane; i < mv.dstride; i += 64) xr[i] = 0;
        if (lane == 0) {
            s.wid[0][pos] = wid;
            s.aux[0][pos] = make_int2(sx_total, __float_as_int(sxx));
        }
        wave_sync();
    }
}

// LDS: [64 windows][dstride] equalised pixels of the tile, then per wavefront [32 rows][64 windows] rect sums of the current tile
__global__ __launch_bounds__(256) void k_wvb_chain(WvbDev mv, WvbState s, int phase, const unsigned int* countPtr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wvb_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned int n = wvb_count(countPtr, s);
    const int ntiles = (int)((n + 63u) >> 6);
    const int NP = mv.numPer, NU = mv.numUsed, KS = mv.KS, DS = mv.dstride;
    const int NQ = (NP + 3) >> 2;
    const int set = phase & 1;
    const int g0 = mv.phaseGen[phase], g1 = mv.phaseGen[phase + 1];
    const auto* lvl = WVB_CONST(int32_t, mv.lvl);
    const auto* c128 = WVB_CONST(int32_t, mv.c128);
    const auto* ppC = WVB_CONST(double, mv.pp);
    const auto* valC = WVB_CONST(double, mv.val);
    int* Sw = reinterpret_cast<int*>(wvb_lds + 64 * DS) + wave * (32 * 64);
    const int cpr = DS >> 4;   // 16-byte slots per row
    for (int unit = blockIdx.x; unit < ntiles * NQ; unit += gridDim.x) {
        const int t = unit / NQ, cq = unit - t * NQ;   // neighbouring workgroups share the window tile (L2)
        __syncthreads();   // the previous unit's MFMA operand reads are done
        {
            const uint4* xg = reinterpret_cast<const uint4*>(s.X[set] + (size_t)t * 64 * DS);
            const unsigned int rows = min(64u, n - (unsigned int)t * 64u);
            for (int c = threadIdx.x; c < 64 * cpr; c += 256) {
                const unsigned int row = (unsigned int)c / (unsigned int)cpr;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (row < rows) v = xg[c];
                reinterpret_cast<uint4*>(wvb_lds)[c] = v;
            }
        }
        __syncthreads();
        const int cls = cq * 4 + wave;
        if (cls < NP) {
            const unsigned int pos = (unsigned int)t * 64u + (unsigned int)lane;
            const bool valid = pos < n;
            float u = 0.f;
            int sx_total = 0;
            float sxx = 0.f;
            if (valid) {
                const int2 ax = s.aux[set][pos];
                sx_total = ax.x;
                sxx = __int_as_float(ax.y);
                if (phase > 0) u = s.U[set][(size_t)cls * s.kstride + pos];
            }
            int curTile = -1;
            for (int g = g0; g < g1; ++g) {
                const int k = g * NP + cls;
                if (k >= NU) break;
                const int tile = lvl[4 * k], row0 = lvl[4 * k + 1], cnt = lvl[4 * k + 2], vo = lvl[4 * k + 3];
                if (tile != curTile) {
                    // ---- rect sums of the tile's rows for the 64 windows: C[row][window] = sum_pixel M[row][pixel] * x[window][pixel]
                    curTile = tile;
                    wvb_v16i acc0 = {}, acc1 = {};
                    const wvb_v4i* Ap = mv.A + (size_t)tile * KS * 64 + lane;
                    const unsigned char* xb = wvb_lds + (lane & 31) * DS + (lane >> 5) * 16;
                    // the tile's operand streams from L2 through a ring of four fragments (a k-step is two MFMAs: far shorter than a load)
                    wvb_v4i an[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) an[i] = Ap[min(i, KS - 1) * 64];
                    for (int ks = 0; ks < KS; ks += 4) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const wvb_v4i a = an[i];
                            an[i] = Ap[min(ks + 4 + i, KS - 1) * 64];
                            if (ks + i < KS) {
                                const wvb_v4i b0 = *reinterpret_cast<const wvb_v4i*>(xb + (ks + i) * 32);
                                const wvb_v4i b1 = *reinterpret_cast<const wvb_v4i*>(xb + 32 * DS + (ks + i) * 32);
                                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b0, acc0, 0, 0, 0);
                                acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b1, acc1, 0, 0, 0);
                            }
                        }
                    }
                    wave_sync();   // the chain reads of the previous tile are done
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        Sw[row * 64 + (lane & 31)] = acc0[r];
                        Sw[row * 64 + 32 + (lane & 31)] = acc1[r];
                    }
                    wave_sync();
                }
                // ---- the reference's chain for this lane's window (WvmClassifier.cpp:277-333)
                double sum_xp = 0.0;
                int sumv0 = sx_total;
                for (int v = 1; v < cnt; ++v) {
                    const int sv = Sw[(row0 + v - 1) * 64 + lane] + c128[tile * 32 + row0 + v - 1];
                    sumv0 -= sv;
                    const double prod = (double)sv * valC[vo + v];
                    sum_xp = sum_xp + prod;
                }
                const double t0 = (double)sumv0 * valC[vo];
                sum_xp = sum_xp + t0;
                sum_xp = sum_xp + (double)u;
                u = (float)sum_xp;
                double norm = (double)sxx;
                norm = norm - 2 * sum_xp;
                norm = norm + ppC[k];
                const float Kk = (float)exp((double)mv.negBasis * norm);
                if (valid) s.K[set][(size_t)k * s.kstride + pos] = Kk;
            }
            if (valid && phase + 1 < mv.nphase) s.U[set][(size_t)cls * s.kstride + pos] = u;
        }
    }
}

// res_k = -bias + sum_{p <= k} w[k][p] K_p for the rows of the phase; one wavefront per (64 windows, 8 rows)
__global__ __launch_bounds__(256) void k_wvb_sums(WvbDev mv, WvbState s, int phase, const unsigned int* countPtr) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned int n = wvb_count(countPtr, s);
    const int ntiles = (int)((n + 63u) >> 6);
    const int NU = mv.numUsed, Fr = mv.Fr;
    const int k0 = min(mv.phaseGen[phase] * mv.numPer, NU), k1 = min(mv.phaseGen[phase + 1] * mv.numPer, NU);
    const int nrb = (k1 - k0 + 7) >> 3;
    const int set = phase & 1;
    const auto* wR = WVB_CONST(float, mv.wR);
    const size_t ks = (size_t)s.kstride;
    for (int unit = blockIdx.x * 4 + wave; unit < ntiles * nrb; unit += gridDim.x * 4) {
        const int t = unit / nrb, rb = nrb - 1 - (unit - t * nrb);   // the longest rows first
        const int kb = k0 + rb * 8;
        const unsigned int pos = (unsigned int)t * 64u + (unsigned int)lane;
        const bool valid = pos < n;
        const float* Kp = s.K[set] + (valid ? pos : 0u);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = mv.negBias;
        int p = 0;
        // terms every row of the block takes (p < kb), four loads in flight
        for (; p + 4 <= kb; p += 4) {
            const float ka = Kp[(size_t)p * ks], kb_ = Kp[(size_t)(p + 1) * ks], kc = Kp[(size_t)(p + 2) * ks], kd = Kp[(size_t)(p + 3) * ks];
            const auto* w0 = wR + (size_t)p * Fr + kb;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float tt = w0[j] * ka; acc[j] = acc[j] + tt; }
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float tt = w0[Fr + j] * kb_; acc[j] = acc[j] + tt; }
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float tt = w0[2 * Fr + j] * kc; acc[j] = acc[j] + tt; }
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float tt = w0[3 * Fr + j] * kd; acc[j] = acc[j] + tt; }
        }
        for (; p < kb; ++p) {
            const float ka = Kp[(size_t)p * ks];
            const auto* w0 = wR + (size_t)p * Fr + kb;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float tt = w0[j] * ka; acc[j] = acc[j] + tt; }
        }
        // the diagonal block: row kb + j ends with term p = kb + j
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            if (kb + jj < k1) {
                const float ka = Kp[(size_t)(kb + jj) * ks];
                const auto* w0 = wR + (size_t)(kb + jj) * Fr + kb;
#pragma unroll
                for (int j = jj; j < 8; ++j) { const float tt = w0[j] * ka; acc[j] = acc[j] + tt; }
            }
        }
        if (valid) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (kb + j < k1) s.R[(size_t)(kb + j - k0) * ks + pos] = acc[j];
        }
    }
}

// end of the last phase: see CascadeOut::host_count; also hands the queue length to the host (overflow check) and clears the
// phase counters for the next run
__device__ __forceinline__ void wvb_finalize(const CascadeOut& o, const WvbState& s) {
    if (!o.host_count) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned int done = atomicAdd(o.done_blocks, 1u);
        if (done == gridDim.x - 1) {
            __threadfence();
            const unsigned int cnt = __hip_atomic_load(o.pos_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned int dq = __hip_atomic_load(o.deep_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(o.host_count + 1, dq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(o.host_count, cnt, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(o.pos_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(o.deep_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(o.deep_count + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(o.done_blocks, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int i = 0; i <= WVB_MAXPHASE; ++i) __hip_atomic_store(s.cnt + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// lane == window: the cascade's exit rule on the phase's rows (WvmClassifier.cpp:139-141), outputs, positives, and the dense
// list + state of the windows that go on to the next phase
__global__ __launch_bounds__(256) void k_wvb_exit(WvbDev mv, WvbState s, CascadeOut o, int phase, const unsigned int* countPtr, unsigned int* nextCount) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned int n = wvb_count(countPtr, s);
    const int ntiles = (int)((n + 63u) >> 6);
    const int NU = mv.numUsed, NP = mv.numPer, DS = mv.dstride, d = mv.d;
    const int k0 = min(mv.phaseGen[phase] * NP, NU), k1 = min(mv.phaseGen[phase + 1] * NP, NU);
    const int set = phase & 1;
    const size_t ks = (size_t)s.kstride;
    const auto* thrC = WVB_CONST(float, mv.thr);
    for (int t = blockIdx.x * 4 + wave; t < ntiles; t += gridDim.x * 4) {
        const unsigned int pos = (unsigned int)t * 64u + (unsigned int)lane;
        const bool valid = pos < n;
        int exitk = -1;
        float fout = 0.f;
        {
            const float* Rp = s.R + (valid ? pos : 0u);
            for (int k = k0; k < k1; ++k) {
                const float r = Rp[(size_t)(k - k0) * ks];
                if (valid && exitk < 0 && (!(r >= thrC[k]) || k + 1 == NU)) { exitk = k; fout = r; }
                if (__ballot(valid && exitk < 0) == 0ull) break;
            }
        }
        const bool exited = valid && exitk >= 0;
        const int64_t wid = valid ? s.wid[set][pos] : 0;
        if (exited) {
            if (o.all_level) o.all_level[wid] = exitk;
            if (o.all_fout) o.all_fout[wid] = fout;
        }
        // ---- positives (WvmClassifier.cpp:143-148: all filters passed)
        const bool positive = exited && (exitk + 1 == mv.numFilters) && (fout >= thrC[exitk]);
        const unsigned long long pmask = __ballot(positive);
        if (pmask) {
            unsigned int base = 0;
            if (lane == 0) base = atomicAdd(o.pos_count, (unsigned int)__popcll(pmask));
            base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
            const unsigned int slot = base + (unsigned int)__popcll(pmask & ((1ull << lane) - 1ull));
            if (positive && slot < o.pos_cap) o.pos[slot] = PosRec{(uint32_t)wid, (uint32_t)(wid >> 32), exitk, fout};
            unsigned long long rest = pmask;
            unsigned int sl = base;
            while (rest) {   // the equalised patch of each positive, copied by the whole wavefront
                const int b = __builtin_ctzll(rest);
                rest &= rest - 1;
                if (sl < o.pos_cap) {
                    const int8_t* src = s.X[set] + (size_t)((unsigned int)t * 64u + (unsigned int)b) * DS;
                    uint8_t* dst = o.pos_patches + (size_t)sl * d;
                    if ((d & 3) == 0) {
                        for (int i = lane; i < (d >> 2); i += 64)
                            reinterpret_cast<unsigned int*>(dst)[i] = reinterpret_cast<const unsigned int*>(src)[i] ^ 0x80808080u;
                    } else {
                        for (int i = lane; i < d; i += 64) dst[i] = (uint8_t)((unsigned int)(uint8_t)src[i] ^ 0x80u);
                    }
                }
                ++sl;
            }
        }
        // ---- windows that go on: dense list of the next phase, state copied to the other set
        const bool surv = valid && exitk < 0;
        const unsigned long long smask = __ballot(surv);
        if (smask) {
            unsigned int base = 0;
            if (lane == 0) base = atomicAdd(nextCount, (unsigned int)__popcll(smask));
            base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
            const unsigned int np = base + (unsigned int)__popcll(smask & ((1ull << lane) - 1ull));
            if (surv) {
                s.wid[set ^ 1][np] = wid;
                s.aux[set ^ 1][np] = s.aux[set][pos];
                for (int c = 0; c < NP; ++c) s.U[set ^ 1][(size_t)c * ks + np] = s.U[set][(size_t)c * ks + pos];
                for (int p = 0; p < k1; ++p) s.K[set ^ 1][(size_t)p * ks + np] = s.K[set][(size_t)p * ks + pos];
            }
            unsigned long long rest = smask;
            unsigned int dp = base;
            const int cpr = DS >> 4;
            while (rest) {
                const int b = __builtin_ctzll(rest);
                rest &= rest - 1;
                const uint4* src = reinterpret_cast<const uint4*>(s.X[set] + (size_t)((unsigned int)t * 64u + (unsigned int)b) * DS);
                uint4* dst = reinterpret_cast<uint4*>(s.X[set ^ 1] + (size_t)dp * DS);
                for (int i = lane; i < cpr; i += 64) dst[i] = src[i];
                ++dp;
            }
        }
    }
    if (phase + 1 == mv.nphase) wvb_finalize(o, s);
}

// queues stage B on `st` behind whatever filled the queue (o.deep_q / o.deep_count)
template <int PW_, int PH_, bool RAW>
static void launch_stageb(fd_ctx* ctx, hipStream_t st, int64_t total, fd_wvm* m, const uint8_t* arena, const WinTable& wt, const CascadeOut& o) {
    const WvbDev& mv = m->wvb;
    const WvbState& s = m->sb;
    const int64_t ub = std::min<int64_t>(total, s.cap);   // upper bound of the windows in any phase
    if (ub <= 0) return;
    const int cus = ctx->num_cus;
    const int64_t tiles = (ub + 63) / 64;
    const int ldsBytes = 64 * mv.dstride + 4 * 32 * 64 * (int)sizeof(int);
    static uint64_t ldsDone = 0;
    fd_allow_lds(ctx, (const void*)k_wvb_chain, 160 * 1024, ldsDone);
    const int perCuC = std::max(1, std::min(8, (160 * 1024) / ldsBytes));
    const int gridP = (int)std::min<int64_t>((ub + 3) / 4, (int64_t)cus * 8);
    hipLaunchKernelGGL((k_wvb_prepare<PW_, PH_, RAW>), dim3(gridP), dim3(256), 0, st, arena, wt, m->dev, mv, s, o.deep_q, o.deep_count);
    const int NQ = (mv.numPer + 3) / 4;
    for (int ph = 0; ph < mv.nphase; ++ph) {
        const unsigned int* countPtr = ph == 0 ? o.deep_count : s.cnt + ph;
        const int k0 = std::min(mv.phaseGen[ph] * mv.numPer, mv.numUsed), k1 = std::min(mv.phaseGen[ph + 1] * mv.numPer, mv.numUsed);
        const int nrb = (k1 - k0 + 7) / 8;
        const int gridC = (int)std::min<int64_t>(tiles * NQ, (int64_t)cus * perCuC);
        hipLaunchKernelGGL(k_wvb_chain, dim3(gridC), dim3(256), ldsBytes, st, mv, s, ph, countPtr);
        const int gridH = (int)std::min<int64_t>((tiles * nrb + 3) / 4, (int64_t)cus * 8);
        hipLaunchKernelGGL(k_wvb_sums, dim3(gridH), dim3(256), 0, st, mv, s, ph, countPtr);
        const int gridE = (int)std::min<int64_t>((tiles + 3) / 4, (int64_t)cus * 4);
        hipLaunchKernelGGL(k_wvb_exit, dim3(gridE), dim3(256), 0, st, mv, s, o, ph, countPtr, s.cnt + ph + 1);
    }
}
