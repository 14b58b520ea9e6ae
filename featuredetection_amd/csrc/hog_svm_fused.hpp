// featuredetection_amd/csrc/hog_svm_fused.hpp -- BASELINE config 2 as ONE kernel: DirectPyramidFeatureExtractor::extract
// (DirectPyramidFeatureExtractor.cpp:75-123) + HogFilter (HogFilter.cpp:58-122 on HistogramFilter.cpp:134-163) + the RBF
// SvmClassifier::computeHyperplaneDistance (SvmClassifier.cpp:55-60, RbfKernel.hpp) for every sliding window, without the
// feature vectors ever leaving the compute unit.  Included by hog.hip (inside its anonymous namespace).
//
// The two-kernel path (k_hog_tile -> 365 MB of fragment-major features per 640x480 frame -> k_svm_rbf_mfma_svs, which reads them
// four times, -> k_sum_partials) keeps the SUPPORT VECTORS stationary in registers, so a tile of windows has to visit four
// compute units.  Here the roles are swapped: a wavefront computes the HOG vectors of 32 windows straight into the 164 VGPRs of
// its MFMA B operand (lane (h, r) holds elements 8q + 4h .. + 3 of window r for q = 0 .. 40: the fragment layout of svm.hip), and
// the support-vector tiles (1.3 MB, L2 resident) stream through a double-buffered LDS image filled by global_load_lds -- the
// inner loop, its barrier per tile and its epilogue are those of k_svm_rbf_mfma_svs with A and B exchanged.  The sum over support
// vectors stays in the lane (fp64, tile after tile), so there are no partial sums to add in a second kernel.
//
// HOG of a tile, lane == (window r, half h): the lane loads ITS window's pixel rows 10h .. 10h + 9 of the (bin, weight) layer
// (unaligned dwordx4 / x2 loads, 100 dwords), owns the eight cells of those rows and accumulates their histograms in LDS
// (hist[cell][bin][r]: the 32 windows of a half-wave fall into 32 different banks whatever their bins).  One step = one pixel
// of each of the eight cells: eight independent read-add-write chains in flight, every accumulator still sees its addends in the
// reference's row-major order (HistogramFilter.cpp:150-163), so the fp32 histograms, energies, block norms and block vectors are
// bit-identical to k_hog_tile's and the CPU path's.  Cell energies stay in registers (the partner half arrives through one
// ds_bpermute each), both lanes of a window compute its nine block norms, and each lane multiplies out the 164 elements it owns.
// The histograms alias the support-vector buffers (they are idle while a round's windows are prepared); the eight wavefronts
// prepare their tiles in two groups of four so that the kernel needs 92 KB of LDS, not 160: the other streams' pyramid kernels can
// still be resident on the same compute units.
//
// Work: a round = 8 tiles (one per wavefront) x all support-vector tiles; workgroup g takes tile groups g, g + G, ...  The
// groups left over after the last full round are split over the support vectors (S parts, S a power of two) so that the tail
// costs 1 / S of a round instead of a whole one; their partial sums are added by k_hsf_finish.
#pragma once

struct HsfSvm {
    const float* svFrag;   // fragment-major support vectors [nsvt][Q * 256]
    const float* ss;       // |s|^2 per support vector (padded ones 0)
    const float* coeff;    // coefficients (padded ones 0)
    int32_t nsvt;          // tiles of 32 support vectors
    float bias, negGamma;
};
struct HsfPlan {
    int32_t G, R, rem, S;  // workgroups, full rounds, tile groups of the remainder round, support-vector parts of a remainder group
    int64_t T, N, npadRows;
};
// Positive windows (SvmClassifier.cpp:44-46: distance >= threshold) go straight to host-mapped pinned memory: records 1 .. of `pos`,
// their number into record 0 by the last workgroup of k_hsf_finish, which also clears the device header {count, retired workgroups}
// for the next run -- no selection kernel, no memset and no copy on the stream (the host reads the buffer after the stream's event).
struct HsfOut {
    double* dist;          // every window's hyperplane distance (device)
    double* part;          // partial sums of the remainder round
    HogPos* pos;           // pinned, host-mapped
    unsigned int* header;  // device: [0] positives so far, [1] retired workgroups of k_hsf_finish
    unsigned int cap;
    float threshold;
};

// appends this wavefront's positive windows: one atomic per wavefront
__device__ __forceinline__ void hsf_append(const HsfOut& o, bool positive, int64_t w, double d) {
    const unsigned long long mask = __ballot(positive);
    if (!mask) return;
    const int lane = threadIdx.x & 63;
    unsigned int base = 0;
    if (lane == __ffsll((long long)mask) - 1) base = atomicAdd(o.header, (unsigned int)__popcll(mask));
    base = __shfl(base, __ffsll((long long)mask) - 1, 64);
    if (positive) {
        const unsigned int slot = base + (unsigned int)__popcll(mask & ((1ull << lane) - 1ull));
        if (slot < o.cap) o.pos[1 + slot] = HogPos{(uint32_t)w, (uint32_t)(w >> 32), d};
    }
}

typedef float hsf_f32x16 __attribute__((ext_vector_type(16)));
typedef float hsf_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void hsf_lds_void_t;

constexpr int HSF_Q = 41;          // 8-wide k groups of the 324-element vector (KP = 328)
constexpr int HSF_F = 324;
constexpr int HSF_HS = 144;        // floats per window in the histogram scratch (16 cells x 9 bins)
constexpr int HSF_TILE_FLOATS = HSF_Q * 256;

// element k of the feature vector -> offset of its histogram value inside a window's 144 floats, and its block
// (HogFilter.cpp:85-97: blocks row-major, the block's 2 x 2 cells row-major, 9 bins each)
__host__ __device__ constexpr int hsf_off(int k) {
    return (((k / 36) / 3 + ((k % 36) / 9) / 2) * 4 + ((k / 36) % 3 + ((k % 36) / 9) % 2)) * 9 + (k % 9);
}
__host__ __device__ constexpr int hsf_block(int k) { return k / 36; }

__device__ __forceinline__ void hsf_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// HOG vectors of the 32 windows of `tile` -> B (this lane's 41 float4 fragment slots) and |x|^2 of window lane & 31
__device__ __forceinline__ void hsf_hog_tile(const uint8_t* __restrict__ arena, const HogWinTable& wt, int64_t tile, float* hist,
                                             hsf_f32x4 (&B)[HSF_Q], float& xxOut) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    int64_t wid = tile * 32 + r;
    wid = wid < wt.total ? wid : wt.total - 1;   // rows behind the last window repeat it (their results are never written)
    int li = 0;
    for (int l = 1; l < wt.n; ++l) li = wt.l[l].first <= wid ? l : li;
    const HogWinLayer wl = wt.l[li];
    const int local = (int)(wid - wl.first);
    const int iy = local / wl.nx, ix = local - iy * wl.nx;
    const uint8_t* src = arena + wl.off + 2 * ((size_t)(wl.by + iy * wt.sy + 10 * h) * wl.lw + (wl.bx + ix * wt.sx));
    const size_t rowBytes = (size_t)wl.lw * 2;
    // the lane's ten pixel rows: 20 (bin, weight) pairs = 10 dwords each
    uint32_t px[10][10];
#pragma unroll
    for (int y = 0; y < 10; ++y) {
        const uint8_t* p = src + y * rowBytes;
        uint4 a, b;
        uint2 c;
        __builtin_memcpy(&a, p, 16);
        __builtin_memcpy(&b, p + 16, 16);
        __builtin_memcpy(&c, p + 32, 8);
        px[y][0] = a.x; px[y][1] = a.y; px[y][2] = a.z; px[y][3] = a.w;
        px[y][4] = b.x; px[y][5] = b.y; px[y][6] = b.z; px[y][7] = b.w;
        px[y][8] = c.x; px[y][9] = c.y;
    }
    // hist[cell][bin][window]: a wavefront's accesses differ in the window (and, in the accumulation, in the data-dependent bin, which
    // moves an address by whole rows of 32 floats), so the 32 lanes of a half always fall into 32 different banks.  (Window-major rows
    // of 145 floats were conflict-free only for equal bins: ~3-way conflicts made the LDS pipe the bound of the whole phase.)
    float* win = hist + r;                   // element `off` of this window's 16 x 9 histogram values: win[off * 32]
    float* mine = win + h * (72 * 32);       // the eight cells this lane owns (cell rows 2h, 2h + 1)
#pragma unroll
    for (int i = 0; i < 72; ++i) mine[i * 32] = 0.f;
    const float factor = 1.f / 255.f;
    // cell histograms (HistogramFilter.cpp:150-163): step (py, px) adds pixel (py, px) of each of the lane's eight cells
#pragma unroll
    for (int py = 0; py < 5; ++py) {
#pragma unroll
        for (int pxx = 0; pxx < 5; ++pxx) {
            float* a[8];
            float w[8], v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int y = 5 * (c >> 2) + py, x = 5 * (c & 3) + pxx;
                const uint32_t d = px[y][x >> 1];
                const uint32_t bin = (x & 1) ? ((d >> 16) & 255u) : (d & 255u);
                const uint32_t wgt = (x & 1) ? (d >> 24) : ((d >> 8) & 255u);
                a[c] = mine + (c * 9 + bin) * 32;
                w[c] = factor * (float)wgt;
            }
            // the eight cells are eight different accumulators whatever the bins: read all, add, write all (the compiler cannot
            // know that and would serialise eight read-add-write chains).  ds_add_f32 instead: measured 210 cycles per wave64
            // instruction -- the LDS unit adds floats one lane at a time (kernel 1.45 -> 2.04 ms)
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = *a[c];
#pragma unroll
            for (int c = 0; c < 8; ++c) *a[c] = v[c] + w[c];
        }
    }
    hsf_wave_sync();
    // cell energies (HogFilter.cpp:102-122, no signed / unsigned combination): own eight, the partner's eight through the crossbar
    float own[8], oth[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float en = 0.f;
#pragma unroll
        for (int b = 0; b < 9; ++b) { const float hv = mine[(c * 9 + b) * 32]; en = en + hv * hv; }
        own[c] = en;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) oth[c] = __shfl_xor(own[c], 32, 64);
    float E[16];
#pragma unroll
    for (int c = 0; c < 8; ++c) { E[c] = h ? oth[c] : own[c]; E[8 + c] = h ? own[c] : oth[c]; }
    // block normalisers (HogFilter.cpp:78-84)
    const float eps = 1e-4f;
    float nrm[9];
#pragma unroll
    for (int bl = 0; bl < 9; ++bl) {
        const int br = bl / 3, bc = bl % 3;
        float en = 0.f;
#pragma unroll
        for (int cr = br; cr < br + 2; ++cr)
#pragma unroll
            for (int cc = bc; cc < bc + 2; ++cc) en = en + E[cr * 4 + cc];
        nrm[bl] = 1.f / sqrtf(en + eps);
    }
    // block vectors (HogFilter.cpp:85-97): this lane's elements 8q + 4h + t.  The two halves of the wavefront run one after the
    // other under their exec masks, so that every LDS offset and every normaliser is a compile-time choice (a per-lane select of
    // two constants per element cost four VALU instructions where this costs half a multiply)
    if (h == 0) {
#pragma unroll
        for (int q = 0; q < HSF_Q; ++q) {
            hsf_f32x4 o;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = 8 * q + t;
                o[t] = k < HSF_F ? nrm[hsf_block(k < HSF_F ? k : 0)] * win[(k < HSF_F ? hsf_off(k) : 0) * 32] : 0.f;
            }
            B[q] = o;
        }
    } else {
#pragma unroll
        for (int q = 0; q < HSF_Q; ++q) {
            hsf_f32x4 o;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = 8 * q + 4 + t;
                o[t] = k < HSF_F ? nrm[hsf_block(k < HSF_F ? k : 0)] * win[(k < HSF_F ? hsf_off(k) : 0) * 32] : 0.f;
            }
            B[q] = o;
        }
    }
    // |x|^2 = sum over the blocks of norm^2 x (the block's four cell energies), in fp64: nine terms instead of 324 squares (the vector
    // itself is what the MFMAs read; |x|^2 only enters d^2 = |x|^2 + |s|^2 - 2 x.s, and this sum is the more accurate of the two)
    double sqd = 0.0;
#pragma unroll
    for (int bl = 0; bl < 9; ++bl) {
        const int br = bl / 3, bc = bl % 3;
        const double e = ((double)E[br * 4 + bc] + (double)E[br * 4 + bc + 1]) + ((double)E[(br + 1) * 4 + bc] + (double)E[(br + 1) * 4 + bc + 1]);
        sqd += ((double)nrm[bl] * (double)nrm[bl]) * e;
    }
    xxOut = (float)sqd;
    hsf_wave_sync();
}

// SUBS: groups the eight wavefronts prepare their tiles in (2: four at a time, 92 KB of LDS; 1: all at once, 157 KB; 0: no HOG at
// all -- a timing experiment, results meaningless)
template <int SUBS>
__global__ __launch_bounds__(512, 2) void k_hog_svm_fused(const uint8_t* __restrict__ arena, HogWinTable wt, HsfSvm m, HsfPlan plan, HsfOut out) {
    extern __shared__ __attribute__((aligned(16))) float hsf_lds[];
    float* const buf0 = hsf_lds;
    float* const buf1 = hsf_lds + HSF_TILE_FLOATS;
    constexpr int HISTW = SUBS == 1 ? 8 : 4;            // wavefronts whose histogram scratch exists at the same time
    constexpr int SCR = (HISTW * 32 * HSF_HS > 2 * HSF_TILE_FLOATS) ? HISTW * 32 * HSF_HS : 2 * HSF_TILE_FLOATS;
    float* const svc = hsf_lds + SCR;   // [nsvt][64]: |s|^2 of the tile's 32 support vectors, then their coefficients
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < m.nsvt * 64; i += 512) {
        const int t = i >> 6, j = i & 63;
        svc[i] = j < 32 ? m.ss[t * 32 + j] : m.coeff[t * 32 + j - 32];
    }
    float* const hist = hsf_lds + (wave & (HISTW - 1)) * (32 * HSF_HS);   // aliases buf0 / buf1 (4 x 18,432 B <= 83,968 B)
    auto issue_tile = [&](int sv, float* dst) {
        // LDS-DMA, 1 KiB per wave instruction; wave w moves q-groups w, w + 8, ...  (svm.hip: issued through inline asm so that the
        // compiler does not wait vmcnt(0) before every ds_read of the tile being computed; completion is awaited explicitly)
        const char* g = (const char*)(m.svFrag + (size_t)sv * HSF_TILE_FLOATS) + (size_t)lane * 16;
        for (int q = wave; q < HSF_Q; q += 8) {
            const unsigned ldsDst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(hsf_lds_void_t*)(dst + q * 256));
            const char* gsrc = g + (size_t)q * 1024;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gsrc), "s"(ldsDst) : "memory");
        }
    };
    hsf_f32x4 B[HSF_Q];
#pragma unroll
    for (int q = 0; q < HSF_Q; ++q) B[q] = hsf_f32x4{0.f, 0.f, 0.f, 0.f};
    float xxj = 0.f;
    for (int round = 0; round <= plan.R; ++round) {
        int64_t group;
        int s0 = 0, s1 = m.nsvt, prt = -1;
        if (round < plan.R) {
            group = (int64_t)round * plan.G + blockIdx.x;
        } else {
            if ((int)blockIdx.x >= plan.rem * plan.S) break;
            group = (int64_t)plan.R * plan.G + blockIdx.x / plan.S;
            prt = blockIdx.x % plan.S;
            s0 = prt * m.nsvt / plan.S;
            s1 = (prt + 1) * m.nsvt / plan.S;
        }
        const int64_t tile = group * 8 + wave;
        const bool have = tile < plan.T;
        __syncthreads();   // svc staged; the previous round's last support-vector tile has been read by every wavefront
        if (SUBS == 2) {
            for (int sub = 0; sub < 2; ++sub) {
                if ((wave >> 2) == sub && have) hsf_hog_tile(arena, wt, tile, hist, B, xxj);
                __syncthreads();
            }
        } else if (SUBS == 1) {
            if (have) hsf_hog_tile(arena, wt, tile, hist, B, xxj);
            __syncthreads();
        } else {
#pragma unroll
            for (int q = 0; q < HSF_Q; ++q) B[q] = hsf_f32x4{(float)tile, 1.f, 2.f, (float)q};
        }
        int cur = 0;
        issue_tile(s0, buf0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        double s = 0.0;
        for (int sv = s0; sv < s1; ++sv) {
            if (sv + 1 < s1) issue_tile(sv + 1, cur ? buf0 : buf1);
            if (have) {
                const hsf_f32x4* ap = (const hsf_f32x4*)(cur ? buf1 : buf0) + lane;
                hsf_f32x16 acc = {0};
                hsf_f32x4 p = ap[0];
#pragma unroll
                for (int q = 0; q < HSF_Q; ++q) {
                    const hsf_f32x4 pn = ap[(size_t)(q + 1 < HSF_Q ? q + 1 : q) * 64];
                    __builtin_amdgcn_sched_barrier(0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p[0], B[q][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p[1], B[q][1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p[2], B[q][2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p[3], B[q][3], acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    p = pn;
                }
                // RBF values of the tile's 32 support vectors for this lane's window, added to its fp64 sum (C[i = SV][j = window]:
                // accumulator r is support vector (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
                const float* c = svc + sv * 64;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const float d2 = (xxj + c[row]) - 2.f * acc[r];
                    s += (double)c[32 + row] * (double)__expf(m.negGamma * fmaxf(d2, 0.f));
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next tile has landed in LDS
            __syncthreads();
            cur ^= 1;
        }
        s += __shfl_xor(s, 32, 64);   // the two halves of the support-vector rows
        const int64_t w = tile * 32 + lane;
        if (prt < 0) {
            const double d = SUBS == 0 ? -1e30 + 1e-30 * s : -(double)m.bias + s;
            const bool mine = have && lane < 32 && w < plan.N;
            if (mine) out.dist[w] = d;
            hsf_append(out, mine && d >= (double)out.threshold, w, d);
        } else if (have && lane < 32) {
            out.part[(size_t)prt * plan.npadRows + w] = SUBS == 0 ? -1e30 + 1e-30 * s : s;
        }
    }
}

// windows of the remainder round: dist = -bias + the S parts' sums, in part order; their positives; the last workgroup to retire
// hands the positive count to the host and clears the device header
__global__ __launch_bounds__(256) void k_hsf_finish(HsfOut o, int S, int64_t npadRows, int64_t first, int64_t n, float bias) {
    for (int64_t i0 = first + (int64_t)blockIdx.x * blockDim.x; i0 < n; i0 += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = i0 + threadIdx.x;
        double d = 0.0;
        if (i < n) {
            double s = 0.0;
            for (int p = 0; p < S; ++p) s += o.part[(size_t)p * npadRows + i];
            d = -(double)bias + s;
            o.dist[i] = d;
        }
        hsf_append(o, i < n && d >= (double)o.threshold, i, d);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // no fences: the counters are agent-scope atomics, the records went to pinned memory, and the host reads them only after the
        // stream's completion event (wvm.hip: wvm_finalize)
        const unsigned int done = atomicAdd(o.header + 1, 1u);
        if (done == gridDim.x - 1) {
            const unsigned int cnt = __hip_atomic_load(o.header, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&o.pos[0].wid_lo, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(o.header, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(o.header + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

static inline size_t hsf_lds_bytes(int subs, int nsvt) {
    const size_t scr = std::max<size_t>((size_t)(subs == 1 ? 8 : 4) * 32 * HSF_HS, (size_t)2 * HSF_TILE_FLOATS);
    return sizeof(float) * (scr + (size_t)nsvt * 64);
}

static inline HsfPlan hsf_plan(int64_t N, int G, int nsvt) {
    HsfPlan pl;
    pl.N = N;
    pl.T = (N + 31) / 32;
    pl.npadRows = ((N + 255) / 256) * 256;
    const int64_t U = (pl.T + 7) / 8;
    pl.G = G;
    pl.R = (int)(U / G);
    pl.rem = (int)(U - (int64_t)pl.R * G);
    int S = 1;
    if (pl.rem > 0)
        while (2 * S <= nsvt && 2 * S * pl.rem <= G) S *= 2;
    pl.S = S;
    return pl;
}
