// featuredetection_amd/csrc/fs_tail.hpp -- stage 2 of FiveStageSlidingWindowDetector::detect on the device (included by wvm.hip only).
//
// FiveStageSlidingWindowDetector.cpp:200-240 hands the WVM positives of a frame to OverlapElimination::eliminate
// (OverlapElimination.cpp:44-105: std::sort by probability, descending, then a greedy sweep -- an element survives iff no earlier
// survivor overlaps it) and the survivors to the SVM.  Rounds 1-3 did that on the host between two kernel chains: positives over PCIe,
// ~0.7 ms of host work per 64-frame call (ordering ~10 K positives, their geometry and logistic, the sweep), SVM launch, second wait --
// the host stages, not the GPU, bounded the headline once the content of the frames varied, and a single frame paid two round trips.
// k_fs_oe runs the elimination where the positives are: one workgroup per frame gathers the frame's records, computes the geometry of
// ImagePyramid / PyramidFeatureExtractor (cvRound(x / scale) + w / 2, IEEE double division and round-half-even like the host),
// sorts, sweeps, and appends the survivors' patch slots to the list the SVM kernel behind it reads (its count stays on the device).
//
// Exactness.  The reference's order is std::sort on the probability 1 / (1 + exp(A + B fout)) computed with libm in double.  The
// kernel sorts by the fp32 filter output instead (descending for B < 0), which is the same permutation iff the probabilities of the
// frame are pairwise different and ordered like the outputs.  It proves that per frame or gives up: equal outputs (the reference's
// order of ties is whatever its std::sort does with them), B = 0 or not finite, and neighbours in sorted order whose probabilities
// could collapse in double (the logistic saturates: relative gap of 1 + e below 1e-14, a margin of ~100 ulp over the error of any
// libm) set FST_AMBIGUOUS, and the host then runs its own elimination for the whole call (five_stage_frames_host's first form).
// Everything else the sweep needs is integer or correctly rounded fp32 arithmetic, the same on both sides.
#pragma once

constexpr int FST_RANKMAX = 512;        // up to here the order comes from a rank sort (one barrier), above from a bitonic network
constexpr int FST_NMAX = 1024;           // positives of one frame the device sweep takes (more: FST_OVERFLOW, the host path runs)
constexpr unsigned int FST_AMBIGUOUS = 1u, FST_OVERFLOW = 2u, FST_WIDE_ID = 4u;

struct FstLayer {        // a non-empty window layer of the call (fd_window_to_detection)
    double scale;
    int32_t first, nx, bx, by, ow, oh;
};
struct FstTable {
    int32_t n, sx, sy, nimg;
    uint32_t perImage, magic;      // windows per frame; mulhi(id, magic) = id / perImage or one less (ids of a call fit 32 bits)
    float oeDist, oeRatio;         // OverlapElimination(dist, ratio) with the constructor's clamping of ratio applied
    double logA, logB;             // ProbabilisticWvmClassifier's logistic
    FstLayer l[WVM_MAX_LAYERS];
};
struct FstKeep { uint32_t slot, wid; float fout; int32_t level; };   // a survivor: patch slot, window id inside its frame, WVM output
struct FstFrame { uint32_t npos, nkeep, base, flags; };              // per frame: positives, survivors, where they start in the lists
struct FstHdr { uint32_t keepTotal, done, svmCount, pad; };          // device counters of a handle
struct FstIO {
    const PosRec* pos;              // the cascade's positives (device memory), in no particular order
    const unsigned int* posCount;   // their number (written by stage B's last workgroup)
    unsigned int posCap;
    FstHdr* hdr;
    unsigned int* frameCount;       // [nimg] positives of every frame (k_wvb_exit counted them; this kernel clears its frame's counter)
    const uint32_t* frameList;      // [nimg][FST_NMAX] their slots in pos
    uint32_t* slots;                // [posCap] device: patch slots of the survivors, frame after frame in sweep order -> the SVM kernel
    FstKeep* keep;                  // [posCap] pinned host memory: the same survivors for the host's result records
    FstFrame* frames;               // [nimg] pinned host memory
    uint32_t* hostHdr;              // pinned host memory: {survivors of the call, OR of the frames' flags}
};

namespace {

// -DFD_FST_PROF (dev builds): s_memtime ticks of workgroup 0 at the phase boundaries of k_fs_oe
#ifdef FD_FST_PROF
__device__ unsigned long long fd_fst_prof[8];
#define FST_T(i) if (threadIdx.x == 0 && blockIdx.x == 0) fd_fst_prof[i] = __builtin_amdgcn_s_memtime()
#else
#define FST_T(i)
#endif

__device__ __forceinline__ unsigned int fst_sortable(float x) {   // ascending with x (no NaNs reach this: outputs of finite sums)
    const unsigned int b = __float_as_uint(x);
    return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float fst_unsortable(unsigned int s) {
    return __uint_as_float(s ^ ((s >> 31) ? 0x80000000u : 0xffffffffu));
}

__global__ __launch_bounds__(256) void k_fs_oe(FstTable T, FstIO io) {
    __shared__ unsigned long long key[FST_NMAX];   // sort key << 32 | entry
    __shared__ uint32_t eSlot[FST_NMAX], eWid[FST_NMAX], eFoutBits[FST_NMAX];   // eFoutBits is dead after the keys are built: sCx's storage
    __shared__ FstLayer sL[WVM_MAX_LAYERS];   // the layer table (a per-thread index into the kernel arguments would spill them)
    __shared__ int sCy[FST_NMAX];   // geometry in sorted order (sCx aliases eFoutBits)
    __shared__ unsigned short sW[FST_NMAX], aIdx[FST_NMAX];   // patch width in sorted order; sorted positions of the accepted elements
    __shared__ int aCx[FST_NMAX], aCy[FST_NMAX];              // the accepted elements' geometry, packed in acceptance order
    __shared__ unsigned short aW[FST_NMAX];
    __shared__ unsigned int deadPart[4][64];                   // sweep: candidate c is overlapped by an accepted element (per wavefront)
    __shared__ unsigned short rowPart[4][64];                  // sweep: which of the candidates 16 q .. 16 q + 15 candidate c overlaps
    __shared__ unsigned int cnt, flags, nkeepS, baseS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned int frame = blockIdx.x;
    int* sCx = reinterpret_cast<int*>(eFoutBits);
    if (tid == 0) { cnt = 0; flags = 0; nkeepS = 0; baseS = 0; }
    for (int l = 0; l < T.n; ++l)   // uniform index: scalar loads
        if (tid == 0) sL[l] = T.l[l];
    __syncthreads();
    FST_T(0);
    // ---- 1. this frame's positives: k_wvb_exit filed their slots under the frame (with every workgroup scanning all records of a 64-frame
    //         call the scan alone was 13 us: 64 CUs pulling the same lines).  Ids beyond 32 bits cannot occur here (the launcher checks).
    const unsigned int npos = min(*io.posCount, io.posCap);
    const unsigned int nmine = io.frameCount[frame];
    for (unsigned int e = tid; e < min(nmine, (unsigned int)FST_NMAX); e += 256) {
        const unsigned int i = io.frameList[(size_t)frame * FST_NMAX + e];
        const PosRec r = io.pos[i < npos ? i : 0];
        if (r.wid_hi || i >= npos) { atomicOr(&flags, FST_WIDE_ID); continue; }
        eSlot[e] = i;
        eWid[e] = T.nimg > 1 ? r.wid_lo - frame * T.perImage : r.wid_lo;
        eFoutBits[e] = __float_as_uint(r.fout);
    }
    if (tid == 0) { cnt = nmine; io.frameCount[frame] = 0u; }   // the counter is this workgroup's: clean for the next run
    __syncthreads();
    const unsigned int n = cnt;
    unsigned int fl = flags;
    if (n > (unsigned int)FST_NMAX) fl |= FST_OVERFLOW;
    if (!(T.logB < 0.0 || T.logB > 0.0) || !(fabs(T.logB) < 1e300) || !(fabs(T.logA) < 1e300)) fl |= FST_AMBIGUOUS;
    unsigned int nkeep = 0;
    if (fl == 0 && n > 0) {
        FST_T(1);
        // ---- 2. sort by probability, descending: descending output for B < 0 (p falls with A + B x), ascending for B > 0
        unsigned int npad = 64;
        while (npad < n) npad <<= 1;
        const bool desc = T.logB < 0.0;
        for (unsigned int i = tid; i < npad; i += 256) {
            unsigned long long k = ~0ull;
            if (i < n) {
                const unsigned int s = fst_sortable(__uint_as_float(eFoutBits[i]));
                k = ((unsigned long long)(desc ? ~s : s) << 32) | i;
            }
            key[i] = k;
        }
        __syncthreads();
        if (n <= (unsigned int)FST_RANKMAX) {
            // small frames (the usual case): every thread ranks its key against all others (broadcast reads, one barrier) -- the keys
            // are pairwise different (entry number in the low word), so the ranks are a permutation
            unsigned long long mine[FST_RANKMAX / 256];
            unsigned int rank[FST_RANKMAX / 256];
#pragma unroll
            for (int q = 0; q < FST_RANKMAX / 256; ++q) { mine[q] = tid + 256 * q < (int)n ? key[tid + 256 * q] : ~0ull; rank[q] = 0; }
            for (unsigned int j = 0; j < n; ++j) {
                const unsigned long long kj = key[j];
#pragma unroll
                for (int q = 0; q < FST_RANKMAX / 256; ++q) rank[q] += kj < mine[q] ? 1u : 0u;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < FST_RANKMAX / 256; ++q)
                if (tid + 256 * q < (int)n) key[rank[q]] = mine[q];
            __syncthreads();
        } else {
            for (unsigned int k = 2; k <= npad; k <<= 1)
                for (unsigned int j = k >> 1; j > 0; j >>= 1) {
                    for (unsigned int i = tid; i < npad; i += 256) {
                        const unsigned int x = i ^ j;
                        if (x > i) {
                            const unsigned long long a = key[i], b = key[x];
                            if ((a > b) == ((i & k) == 0)) { key[i] = b; key[x] = a; }
                        }
                    }
                    __syncthreads();
                }
        }
        FST_T(2);
        // ---- 3. geometry in sorted order (fd_window_to_detection), and the proof that the order is the reference's
        unsigned int amb = 0;
        for (unsigned int i = tid; i < n; i += 256) {
            const unsigned int e = (unsigned int)key[i];
            const unsigned int wid = eWid[e];
            int li = 0;
            for (int l = 1; l < T.n; ++l) li += wid >= (unsigned int)sL[l].first ? 1 : 0;
            const FstLayer L = sL[li];
            const unsigned int local = wid - (unsigned int)L.first;
            const unsigned int iy = local / (unsigned int)L.nx, ix = local - iy * (unsigned int)L.nx;
            const int lx = L.bx + (int)ix * T.sx, ly = L.by + (int)iy * T.sy;
            const int cxv = (int)rint((double)lx / L.scale) + L.ow / 2;   // written behind the barrier: sCx is the keys' source array
            sCy[i] = (int)rint((double)ly / L.scale) + L.oh / 2;
            sW[i] = (unsigned short)L.ow;
            if (L.ow > 65535) amb = 1;
            if (i + 1 < n) {
                const unsigned int e2 = (unsigned int)key[i + 1];
                const float x1 = fst_unsortable(desc ? ~(unsigned int)(key[i] >> 32) : (unsigned int)(key[i] >> 32));
                const float x2 = fst_unsortable(desc ? ~(unsigned int)(key[i + 1] >> 32) : (unsigned int)(key[i + 1] >> 32));
                (void)e2;
                if ((unsigned int)(key[i] >> 32) == (unsigned int)(key[i + 1] >> 32)) amb = 1;   // equal outputs: a tie of the reference's sort
                const double t1 = T.logA + T.logB * (double)x1, t2 = T.logA + T.logB * (double)x2;
                const double ex = exp(t1 < t2 ? t1 : t2);   // the smaller e of the pair: where 1 + e resolves least
                const double gap = ex / (1.0 + ex) * fabs(T.logB) * fabs((double)x1 - (double)x2);
                // gap = relative distance of the two (1 + e) values, to first order.  The host evaluates p = 1.0f / (1.0f + exp(t)) in double:
                // exp is within 1 ulp (glibc states < 1 ulp for its double exp; ocml the same), the sum and the quotient add 0.5 ulp each,
                // so a computed p is within ~3 ulp (6.7e-16 relative) of the exact one, and two p whose exact values differ by more than
                // ~1.4e-15 relative keep their order whatever the libm.  1e-14 (45 ulp) leaves a factor of seven for a libm with a 5-ulp
                // exp and for the first-order estimate; anything closer goes to the host path, which sorts the doubles it computed itself.
                // tests/test_host_logic.py::test_probability_order_margin_of_the_device_overlap_elimination sweeps adjacent fp32 outputs
                // up to saturation against THIS host's libm.
                if (!(gap > 1e-14)) amb = 1;
            }
            sCx[i] = cxv;   // entry i of eFoutBits: only this thread's key was built from it, and the keys are complete
        }
        if (amb) atomicOr(&flags, FST_AMBIGUOUS);
        __syncthreads();
        fl |= flags;
        FST_T(3);
        // ---- 4. the greedy sweep (OverlapElimination.cpp:60-100: an element survives iff no earlier survivor overlaps it), 64
        //         candidates at a time.  All pair tests are data parallel: A) the chunk against the survivors so far, thread = (candidate,
        //         every fourth survivor); M) the chunk's own 64 x 64 overlap matrix, thread = (candidate, 16 others) -> one 64-bit row
        //         per candidate in the first wavefront's registers.  What is sequential -- the first candidate still alive survives and
        //         kills the later ones it overlaps -- is then a loop over mask words: find-first-set, two v_readlane, an and-not.
        if (fl == 0) {
            const float dist = T.oeDist, ratio = T.oeRatio;
            const bool scaled = dist <= 1.0f;
            auto overlaps = [&](int acx, int acy, int aw, int pcx, int pcy, int pw) {
                const int wmax = max(aw, pw), wmin = min(aw, pw);
                const float d = scaled ? dist * (float)wmax : dist;
                // (float)min / (float)max > ratio; for ratio == 0 that is min > 0 (a quotient of positive integers is positive)
                const bool r = ratio == 0.0f ? wmin > 0 : ((float)wmin / (float)wmax) > ratio;
                return ((float)abs(acx - pcx) < d) && ((float)abs(acy - pcy) < d) && r;
            };
            // the usual parameters (an absolute distance, no size ratio: dist > 1, ratio == 0 -- FaceFrontal.cfg's 5 / 0) make the test two
            // integer window checks: |dx| < d for an integer |dx| is |dx| < ceil(d), and min(w) / max(w) > 0 always holds
            const bool simple = !scaled && ratio == 0.0f && dist > 1.0f && dist < 1e6f;
            const int dI = simple ? (int)ceilf(dist) : 0;
            auto near = [&](int acx, int acy, int pcx, int pcy) {   // |dx| < dI && |dy| < dI
                return (unsigned int)(acx - pcx + dI - 1) < (unsigned int)(2 * dI - 1) && (unsigned int)(acy - pcy + dI - 1) < (unsigned int)(2 * dI - 1);
            };
            unsigned int na = 0;   // survivors so far (the same in every thread)
            for (unsigned int i0 = 0; i0 < n; i0 += 64) {
                const unsigned int i = i0 + lane;
                const bool valid = i < n;
                const int pcx = valid ? sCx[i] : 0, pcy = valid ? sCy[i] : 0, pw = valid ? (int)sW[i] : 1;
                // A: against the survivors wave, wave + 4, ... (four at a time: their reads are in flight together)
                bool dead = false;
                if (simple) {
                    for (unsigned int a0 = wave; a0 < na; a0 += 16) {
                        int ax[4], ay[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const unsigned int a = min(a0 + 4u * u, na - 1); ax[u] = aCx[a]; ay[u] = aCy[a]; }
#pragma unroll
                        for (int u = 0; u < 4; ++u) dead |= (a0 + 4u * u < na) && near(ax[u], ay[u], pcx, pcy);
                    }
                } else {
                    for (unsigned int a = wave; a < na; a += 4)
                        if (overlaps(aCx[a], aCy[a], (int)aW[a], pcx, pcy, pw)) dead = true;
                }
                deadPart[wave][lane] = dead ? 1u : 0u;
                // M: against the candidates 16 wave .. 16 wave + 15 of the chunk (bit b: candidate 16 wave + b)
                unsigned int bits = 0;
                if (simple) {
#pragma unroll
                    for (int b0 = 0; b0 < 16; b0 += 8) {
                        int ox[8], oy[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) { const unsigned int o = min(i0 + 16u * wave + b0 + u, n - 1); ox[u] = sCx[o]; oy[u] = sCy[o]; }
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (i0 + 16u * wave + b0 + u < n && near(ox[u], oy[u], pcx, pcy)) bits |= 1u << (b0 + u);
                    }
                } else {
#pragma unroll 4
                    for (int b = 0; b < 16; ++b) {
                        const unsigned int o = i0 + 16u * wave + b;
                        if (o < n && overlaps(sCx[o], sCy[o], (int)sW[o], pcx, pcy, pw)) bits |= 1u << b;
                    }
                }
                rowPart[wave][lane] = (unsigned short)bits;
                __syncthreads();
                unsigned long long acc = 0;   // the chunk's survivors
                if (wave == 0) {
                    const bool alive = valid && !(deadPart[0][lane] | deadPart[1][lane] | deadPart[2][lane] | deadPart[3][lane]);
                    const unsigned int rlo = (unsigned int)rowPart[0][lane] | ((unsigned int)rowPart[1][lane] << 16);
                    const unsigned int rhi = (unsigned int)rowPart[2][lane] | ((unsigned int)rowPart[3][lane] << 16);
                    unsigned long long m = __ballot(alive);
                    while (m) {
                        const int c = __builtin_ctzll(m);   // the first candidate still alive: it survives
                        const unsigned long long row = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)rhi, c) << 32) |
                                                       (unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)rlo, c);
                        acc |= 1ull << c;
                        m &= ~(row | (1ull << c));   // (row has bit c set: an element overlaps itself)
                    }
                    // the survivors' geometry, packed in order behind the earlier ones
                    if ((acc >> lane) & 1ull) {
                        const unsigned int r = na + (unsigned int)__popcll(acc & ((1ull << lane) - 1ull));
                        aIdx[r] = (unsigned short)i; aCx[r] = pcx; aCy[r] = pcy; aW[r] = (unsigned short)pw;
                    }
                    if (lane == 0) nkeepS = na + (unsigned int)__popcll(acc);
                }
                __syncthreads();
                na = nkeepS;
            }
        }
        __syncthreads();
        nkeep = fl == 0 ? nkeepS : 0u;
    }
    FST_T(4);
    // ---- 5. survivors -> the SVM's slot list (device) and the host's records, in sweep order
    if (tid == 0) baseS = nkeep ? atomicAdd(&io.hdr->keepTotal, nkeep) : 0u;
    __syncthreads();
    const unsigned int base = baseS;
    for (unsigned int j = tid; j < nkeep; j += 256) {
        const unsigned int e = (unsigned int)key[aIdx[j]];
        const unsigned int slot = eSlot[e];
        if (base + j < io.posCap) {
            io.slots[base + j] = slot;
            io.keep[base + j] = FstKeep{slot, eWid[e], io.pos[slot].fout, io.pos[slot].level};
        }
    }
    if (tid == 0) io.frames[frame] = FstFrame{n, nkeep, base, fl};
    FST_T(5);
    // ---- 6. the last workgroup publishes the call's totals and leaves the counters clean for the next run
    __syncthreads();
    if (tid == 0) {
        if (fl) atomicOr(&io.hdr->pad, fl);
        const unsigned int done = atomicAdd(&io.hdr->done, 1u);
        if (done == gridDim.x - 1) {
            const unsigned int total = __hip_atomic_load(&io.hdr->keepTotal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned int anyFlags = __hip_atomic_load(&io.hdr->pad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&io.hdr->svmCount, anyFlags ? 0u : total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // read by the SVM launch behind this kernel
            io.hostHdr[0] = total;
            io.hostHdr[1] = anyFlags;
            __hip_atomic_store(&io.hdr->keepTotal, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&io.hdr->pad, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&io.hdr->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    FST_T(6);
}

// ---- the same stage for ONE frame with any number of positives (the 15-detector batch of config 3 / 5: ~13 K WVM positives per
//      detector on a busy 1080p frame, where k_fs_oe's LDS-resident arrays end at FST_NMAX = 1024) ------------------------------------
// One workgroup of 1024 threads per job, its arrays in device memory (L2 resident: 20 bytes per positive):
//   1. sort keys (descending probability = the fp32 output's order, entry number in the low word), LSD radix sort, four 8-bit passes:
//      every wavefront counts the digits of its contiguous segment into its own LDS histogram, a scan over (digit, wavefront) gives
//      the wavefronts their output cursors, and a wavefront scatters its segment 64 keys at a time with the rank among equal digits
//      from eight ballots -- stable, no atomics on the output.
//   2. geometry in sorted order + the proof obligations on the order (below).
//   3. the greedy sweep with a PAINTED MAP like hostalgo.cpp's (the usual parameters: an absolute distance > 1, no size ratio -- every
//      cfg of ffpDetectApp -- make "overlaps an accepted element" a property of the candidate's centre alone): one bit per pixel of
//      the frame (+ a margin) in device memory; an accepted element paints the (2 ceil(d) - 1)^2 square of centres it eliminates.
//      A step reads the bits of the next 1024 candidates (atomic loads: the L2 is where the atomicOr's of the previous step landed)
//      and keeps the first <= 256 that are still live; their overlap matrix resolves the step's internal dependencies in a few parallel
//      rounds (a candidate no undecided earlier one overlaps is accepted; the accepted eliminate what they overlap), the accepted ones
//      paint, s_waitcnt vmcnt(0) + barrier.  Two L2 round trips per ~1000 candidates; the quadratic candidates x survivors test of
//      k_fs_oe is gone.
// Order and ties.  Like k_fs_oe the kernel orders by the fp32 output and has to show that this is the reference's order (std::sort by
// the double probability).  Among 13 K positives EQUAL outputs are the rule (a few pairs per job), and giving up on every tie would
// send every job to the host.  Two elements with equal keys are adjacent in every order consistent with the probabilities, and
// swapping two adjacent elements of a greedy sweep that do not overlap each other changes neither the survivors nor any other element's
// position: so a tie is harmless for the SET of survivors unless its members overlap each other (then FST_AMBIGUOUS: the host path
// decides, with the reference's own std::sort).  What a tie can still change is the ORDER of tied survivors in the list, which the
// later stages can see (the block NMS picks the first element of a centre): the host checks that no two SVM positives of the job
// have equal outputs and otherwise takes the host path for that job (five_stage.hpp).  Near-ties (different outputs whose
// probabilities could collapse in double) stay FST_AMBIGUOUS as in k_fs_oe.
constexpr int FSB_T = 1024, FSB_W = 16, FSB_CH = 256, FSB_PAD = 64;
constexpr unsigned int FST_UNSUPPORTED = 8u;   // parameters / coordinates the painted map does not cover: the host path runs
struct FsbIO {
    unsigned long long* keyA;   // [cap] sort buffers
    unsigned long long* keyB;
    int2* geo;                  // [cap] centre of the element at sorted position i
    unsigned int* acc;          // [cap] sorted positions of the accepted elements
    unsigned int* map;          // painted map: mapStride words per row, mapH rows, pixel (x, y) at bit x + FSB_PAD of row y + FSB_PAD
    int mapStride, mapH;
    unsigned int cap;
};

__global__ __launch_bounds__(FSB_T) void k_fs_oe_big(FstTable T, FstIO io, FsbIO S) {
    __shared__ unsigned int hist[FSB_W][256];
    __shared__ unsigned int dtot[256];
    __shared__ FstLayer sL[WVM_MAX_LAYERS];
    __shared__ int cCx[FSB_CH], cCy[FSB_CH];
    __shared__ unsigned long long rowsS[FSB_CH][4];
    __shared__ unsigned long long accS[4];
    __shared__ unsigned int cIdx[FSB_CH], wcnt[FSB_W];
    __shared__ unsigned int flagsS, naS, cutS;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) { flagsS = 0; naS = 0; }
    for (int l = 0; l < T.n; ++l)
        if (tid == 0) sL[l] = T.l[l];
    const unsigned int nposAll = *io.posCount;
    const unsigned int n = min(min(nposAll, io.posCap), S.cap);
    unsigned int fl0 = 0;
    if (nposAll > S.cap || nposAll > io.posCap) fl0 |= FST_OVERFLOW;
    const float dist = T.oeDist;
    const bool simple = dist > 1.0f && T.oeRatio == 0.0f && dist <= 16.0f;
    if (!simple) fl0 |= FST_UNSUPPORTED;
    if (!(T.logB < 0.0 || T.logB > 0.0) || !(fabs(T.logB) < 1e300) || !(fabs(T.logA) < 1e300)) fl0 |= FST_AMBIGUOUS;
    const int dI = simple ? (int)ceilf(dist) : 1;
    const bool desc = T.logB < 0.0;
    // ---- 0. the map starts empty; keys
    {
        uint4* mz = reinterpret_cast<uint4*>(S.map);
        const int nq = (S.mapStride * S.mapH) >> 2;   // the launcher rounds the map to whole uint4
        for (int i = tid; i < nq; i += FSB_T) mz[i] = make_uint4(0, 0, 0, 0);
    }
    unsigned int wide = 0;
    for (unsigned int i = tid; i < n; i += FSB_T) {
        const PosRec r = io.pos[i];
        if (r.wid_hi) wide = 1;
        const unsigned int sk = fst_sortable(r.fout);
        S.keyA[i] = ((unsigned long long)(desc ? ~sk : sk) << 32) | i;
    }
    if (wide) atomicOr(&flagsS, FST_WIDE_ID);
    __threadfence();   // the zeroed map is in the L2 before the first atomicOr / atomic load touches it
    __syncthreads();
    FST_T(0);
    unsigned int fl = fl0 | flagsS;
    unsigned int nkeep = 0;
    if (fl == 0 && n > 0) {
        // ---- 1. LSD radix sort of the high words, 8 bits per pass, keyA -> keyB -> keyA -> keyB -> keyA
        const unsigned int seg = (((n + FSB_W - 1) / FSB_W) + 63u) & ~63u;   // keys per wavefront, whole 64-key groups
        const unsigned int s0 = min((unsigned int)wave * seg, n), s1 = min(s0 + seg, n);
        for (int pass = 0; pass < 4; ++pass) {
            const unsigned long long* src = (pass & 1) ? S.keyB : S.keyA;
            unsigned long long* dst = (pass & 1) ? S.keyA : S.keyB;
            const int sh = 32 + 8 * pass;
            for (int i = tid; i < FSB_W * 256; i += FSB_T) (&hist[0][0])[i] = 0u;
            __syncthreads();
            // (keys in blocks of 16 groups of 64: sixteen loads in flight per lane -- one L2 round trip per block, not per group)
            for (unsigned int b0 = s0; b0 < s1; b0 += 16 * 64) {
                unsigned long long kb[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) { const unsigned int i = b0 + 64u * j + lane; kb[j] = src[min(i, s1 - 1)]; }
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (b0 + 64u * j + lane < s1) atomicAdd(&hist[wave][(unsigned int)(kb[j] >> sh) & 255u], 1u);
            }
            __syncthreads();
            // cursor of (digit d, wavefront w) = keys with a smaller digit + keys of digit d in earlier wavefronts
            if (tid < 256) {
                unsigned int run = 0;
                for (int w = 0; w < FSB_W; ++w) { const unsigned int c = hist[w][tid]; hist[w][tid] = run; run += c; }
                dtot[tid] = run;
            }
            __syncthreads();
            if (wave == 0) {   // exclusive scan of the 256 digit totals: four per lane
                unsigned int v[4], sum = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = dtot[4 * lane + j]; sum += v[j]; }
                unsigned int inc = sum;
                for (int o = 1; o < 64; o <<= 1) { const unsigned int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
                unsigned int ex = inc - sum;
#pragma unroll
                for (int j = 0; j < 4; ++j) { dtot[4 * lane + j] = ex; ex += v[j]; }
            }
            __syncthreads();
            for (int i = tid; i < FSB_W * 256; i += FSB_T) (&hist[0][0])[i] += dtot[i & 255];
            __syncthreads();
            for (unsigned int b0 = s0; b0 < s1; b0 += 16 * 64) {   // this wavefront's segment, in order
                unsigned long long kb[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) { const unsigned int i = b0 + 64u * j + lane; kb[j] = src[min(i, s1 - 1)]; }
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const unsigned int i = b0 + 64u * j + lane;
                    if (b0 + 64u * j >= s1) break;   // wave-uniform
                    const bool have = i < s1;
                    const unsigned long long k = kb[j];
                    const unsigned int d = (unsigned int)(k >> sh) & 255u;
                    unsigned long long same = __ballot(have);
#pragma unroll
                    for (int b = 0; b < 8; ++b) {
                        const unsigned long long mb = __ballot((d >> b) & 1u);
                        same &= ((d >> b) & 1u) ? mb : ~mb;
                    }
                    const unsigned int before = (unsigned int)__popcll(same & ((1ull << lane) - 1ull));
                    const unsigned int base = have ? hist[wave][d] : 0u;
                    __builtin_amdgcn_wave_barrier();
                    if (have) {
                        dst[base + before] = k;
                        if (before + 1 == (unsigned int)__popcll(same)) hist[wave][d] = base + before + 1;   // the digit's last key of the group moves the cursor
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
            __syncthreads();
        }
        FST_T(1);
        // ---- 2. geometry in sorted order (fd_window_to_detection) and the order's proof obligations
        unsigned int amb = 0, unsup = 0;
        const int mapWbits = S.mapStride * 32;
        for (unsigned int i = tid; i < n; i += FSB_T) {
            const unsigned long long ki = S.keyA[i];
            const unsigned int e = (unsigned int)ki;
            const unsigned int wid = io.pos[e].wid_lo;
            int li = 0;
            for (int l = 1; l < T.n; ++l) li += wid >= (unsigned int)sL[l].first ? 1 : 0;
            const FstLayer L = sL[li];
            const unsigned int local = wid - (unsigned int)L.first;
            const unsigned int iy = local / (unsigned int)L.nx, ix = local - iy * (unsigned int)L.nx;
            const int lx = L.bx + (int)ix * T.sx, ly = L.by + (int)iy * T.sy;
            const int cx = (int)rint((double)lx / L.scale) + L.ow / 2, cy = (int)rint((double)ly / L.scale) + L.oh / 2;
            S.geo[i] = make_int2(cx, cy);
            if (cx + FSB_PAD - dI + 1 < 0 || cx + FSB_PAD + dI - 1 >= mapWbits || cy + FSB_PAD - dI + 1 < 0 || cy + FSB_PAD + dI - 1 >= S.mapH) unsup = 1;
            if (i + 1 < n) {
                const unsigned long long kn = S.keyA[i + 1];
                if ((unsigned int)(ki >> 32) != (unsigned int)(kn >> 32)) {   // different outputs: could the probabilities collapse? (k_fs_oe)
                    const float x1 = fst_unsortable(desc ? ~(unsigned int)(ki >> 32) : (unsigned int)(ki >> 32));
                    const float x2 = fst_unsortable(desc ? ~(unsigned int)(kn >> 32) : (unsigned int)(kn >> 32));
                    const double t1 = T.logA + T.logB * (double)x1, t2 = T.logA + T.logB * (double)x2;
                    const double ex = exp(t1 < t2 ? t1 : t2);
                    const double gap = ex / (1.0 + ex) * fabs(T.logB) * fabs((double)x1 - (double)x2);
                    if (!(gap > 1e-14)) amb = 1;
                }
            }
        }
        if (unsup) atomicOr(&flagsS, FST_UNSUPPORTED);
        __syncthreads();
        // equal outputs: harmless unless two members of the run overlap each other (see the header)
        for (unsigned int i = tid; i + 1 < n; i += FSB_T) {
            const unsigned int hi = (unsigned int)(S.keyA[i] >> 32);
            if ((unsigned int)(S.keyA[i + 1] >> 32) != hi) continue;
            const int2 g = S.geo[i];
            for (unsigned int j = i + 1; j < n && (unsigned int)(S.keyA[j] >> 32) == hi; ++j) {
                const int2 h = S.geo[j];
                if ((unsigned int)(g.x - h.x + dI - 1) < (unsigned int)(2 * dI - 1) && (unsigned int)(g.y - h.y + dI - 1) < (unsigned int)(2 * dI - 1)) amb = 1;
            }
        }
        if (amb) atomicOr(&flagsS, FST_AMBIGUOUS);
        __syncthreads();
        fl |= flagsS;
        FST_T(2);
        // ---- 3. the sweep.  A step looks at up to 1024 candidates (one per thread) and takes the first <= 256 of them that no accepted
        //         element has painted: most candidates are dead on arrival (~80 % on config 3's content), so a step advances ~1000
        //         positions for two L2 round trips, and the pair tests run among the live ones only.
        if (fl == 0) {
            unsigned int na = 0;
            unsigned int i0 = 0;
            while (i0 < n) {
                // A: one candidate per thread: is its centre painted?  The live ones are compacted, in order, into cCx / cCy / cIdx
                const unsigned int i = i0 + (unsigned int)tid;
                const bool valid = i < n;
                const int2 g = valid ? S.geo[i] : make_int2(0, 0);
                bool alive = false;
                if (valid) {
                    const unsigned int w = __hip_atomic_load(S.map + (size_t)(g.y + FSB_PAD) * S.mapStride + ((g.x + FSB_PAD) >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    alive = !((w >> ((g.x + FSB_PAD) & 31)) & 1u);
                }
                const unsigned long long am = __ballot(alive);
                if (lane == 0) wcnt[wave] = (unsigned int)__popcll(am);
                if (tid == 0) cutS = FSB_T;
                __syncthreads();
                unsigned int before = 0, totalAlive = 0;
#pragma unroll
                for (int w = 0; w < FSB_W; ++w) { const unsigned int c = wcnt[w]; before += w < wave ? c : 0u; totalAlive += c; }
                const unsigned int cp = before + (unsigned int)__popcll(am & ((1ull << lane) - 1ull));   // position among the live ones
                if (alive && cp < (unsigned int)FSB_CH) { cCx[cp] = g.x; cCy[cp] = g.y; cIdx[cp] = i; }
                if (alive && cp == (unsigned int)FSB_CH) cutS = (unsigned int)tid;   // the 257th live candidate: this step ends in front of it
                __syncthreads();
                const unsigned int nl = min(totalAlive, (unsigned int)FSB_CH);
                const unsigned int consumed = cutS;
                // M: thread = (live candidate c, 64 others): which of them c overlaps
                {
                    const unsigned int c = (unsigned int)tid & (FSB_CH - 1u), q = (unsigned int)tid >> 8;
                    unsigned long long bits = 0;
                    if (c < nl && 64u * q < nl) {
                        const int pcx = cCx[c], pcy = cCy[c];
                        const unsigned int oend = min(64u, nl - 64u * q);
                        for (unsigned int o = 0; o < oend; ++o) {
                            const int ox = cCx[64 * q + o], oy = cCy[64 * q + o];
                            const bool nr = (unsigned int)(ox - pcx + dI - 1) < (unsigned int)(2 * dI - 1) && (unsigned int)(oy - pcy + dI - 1) < (unsigned int)(2 * dI - 1);
                            bits |= nr ? (1ull << o) : 0ull;
                        }
                    }
                    rowsS[c][q] = bits;
                }
                __syncthreads();
                // R: the greedy rule on the live candidates, in parallel rounds: a candidate that no UNDECIDED earlier candidate overlaps is
                // accepted (everything earlier that overlaps it has been eliminated, and an accepted one would have eliminated it), the
                // accepted ones eliminate what they overlap; the first undecided candidate is always accepted, so the rounds end.  The result
                // is the sequential sweep's (induction over the order).  One wavefront, four candidates per lane.
                if (wave == 0) {
                    unsigned long long rr[4][4];
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                        for (int k = 0; k < 4; ++k) rr[gq][k] = rowsS[64 * gq + lane][k];
                    unsigned long long U[4], AC[4] = {0, 0, 0, 0};
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) U[gq] = __ballot(64u * gq + (unsigned int)lane < nl);
                    const unsigned long long lt = (1ull << lane) - 1ull;
                    while (U[0] | U[1] | U[2] | U[3]) {
                        unsigned long long A[4];
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            unsigned long long earlier = 0;
#pragma unroll
                            for (int k = 0; k < 4; ++k) earlier |= rr[gq][k] & U[k] & (k < gq ? ~0ull : (k == gq ? lt : 0ull));
                            A[gq] = __ballot(((U[gq] >> lane) & 1ull) && earlier == 0ull);
                        }
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            const unsigned long long hit = (rr[gq][0] & A[0]) | (rr[gq][1] & A[1]) | (rr[gq][2] & A[2]) | (rr[gq][3] & A[3]);
                            const unsigned long long killed = __ballot(hit != 0ull);   // includes the accepted ones (an element overlaps itself)
                            AC[gq] |= A[gq];
                            U[gq] &= ~killed;
                        }
                    }
                    if (lane < 4) accS[lane] = lane == 0 ? AC[0] : (lane == 1 ? AC[1] : (lane == 2 ? AC[2] : AC[3]));
                }
                __syncthreads();
                // P: the accepted candidates go to the list (in sorted order) and paint the centres they eliminate
                if (tid < FSB_CH) {
                    const unsigned long long a0 = accS[0], a1 = accS[1], a2 = accS[2], a3 = accS[3];
                    const unsigned long long mine = tid < 64 ? a0 : (tid < 128 ? a1 : (tid < 192 ? a2 : a3));
                    if ((mine >> lane) & 1ull) {
                        unsigned int r = na + (unsigned int)__popcll(mine & ((1ull << lane) - 1ull));
                        if (tid >= 64) r += (unsigned int)__popcll(a0);
                        if (tid >= 128) r += (unsigned int)__popcll(a1);
                        if (tid >= 192) r += (unsigned int)__popcll(a2);
                        S.acc[r] = cIdx[tid];
                        const int ax = cCx[tid] + FSB_PAD, ay = cCy[tid] + FSB_PAD;
                        const int lo = ax - dI + 1, hi = ax + dI - 1;   // inside the map (checked with the geometry); at most 31 bits: two words
                        const int w0 = lo >> 5, w1 = hi >> 5;
                        const unsigned int m0 = (~0u << (lo & 31)) & (w0 == w1 ? (~0u >> (31 - (hi & 31))) : ~0u);
                        const unsigned int m1 = ~0u >> (31 - (hi & 31));
                        for (int y = ay - dI + 1; y <= ay + dI - 1; ++y) {
                            atomicOr(S.map + (size_t)y * S.mapStride + w0, m0);
                            if (w1 != w0) atomicOr(S.map + (size_t)y * S.mapStride + w1, m1);
                        }
                    }
                    if (tid == 0) naS = na + (unsigned int)(__popcll(a0) + __popcll(a1) + __popcll(a2) + __popcll(a3));
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the paint has reached the L2 before the next step reads the map
                __syncthreads();
                na = naS;
                i0 += consumed;
            }
            nkeep = na;
        }
    }
    __syncthreads();
    FST_T(3);
    fl |= flagsS;
    if (fl) nkeep = 0;
    // ---- 4. survivors -> the SVM's slot list (device) and the host's records, in sweep order; the call's totals
    for (unsigned int j = tid; j < nkeep; j += FSB_T) {
        const unsigned int e = (unsigned int)S.keyA[S.acc[j]];
        const PosRec r = io.pos[e];
        io.slots[j] = e;
        io.keep[j] = FstKeep{e, r.wid_lo, r.fout, r.level};
    }
    if (tid == 0) {
        io.frames[0] = FstFrame{nposAll, nkeep, 0u, fl};
        io.frameCount[0] = 0u;   // k_wvb_exit counted the frame's positives for k_fs_oe: clean for the next run
        __hip_atomic_store(&io.hdr->svmCount, fl ? 0u : nkeep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // read by the SVM launch behind this kernel
        io.hostHdr[0] = nkeep;
        io.hostHdr[1] = fl;
#ifdef FD_FST_PROF
        fd_fst_prof[4] = __builtin_amdgcn_s_memtime(); fd_fst_prof[5] = n; fd_fst_prof[6] = nkeep;
#endif
    }
}

}  // namespace
