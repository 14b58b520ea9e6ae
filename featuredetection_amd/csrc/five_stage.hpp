// featuredetection_amd/csrc/five_stage.hpp -- the five-stage detector's glue (included by wvm.hip only, inside its extern "C" block,
// behind the WVM entry points it builds on).
//
// detection::FiveStageSlidingWindowDetector::detect (FiveStageSlidingWindowDetector.cpp:187-320, :331-380) around the WVM cascade of
// wvm.hip: the host tail (overlap elimination -> SVM launch -> block NMS: FiveStageTail), the device tail (fs_tail.hpp: k_fs_oe + the
// counted SVM launch queued behind the cascade), and the entry points fd_detect_five_stage, fd_detect_five_stage_image,
// fd_detect_five_stage_frames[_begin / _end] (all frames of a multi-frame pyramid in one cascade run) and
// fd_detect_five_stage_batch / fd_five_stage_batch_begin / _end (several detectors on shared pyramids, ffpDetectApp.cpp:557-600).
#pragma once
// Stages 4-5 of FiveStageSlidingWindowDetector::detect on the SVM positives of one image (FiveStageSlidingWindowDetector.cpp:
// 262-320; the roi variant :360-380 only sorts): block NMS on the probability map, one detection per maximum, sorted by probability.
static void five_stage_nms(const fd_pyramid* p, const int* roi, std::vector<fd_detection>& svmPos, fd_detection* out, int cap, int* count,
                           int32_t* stage_counts) {
    if (stage_counts) stage_counts[2] = (int)svmPos.size();
    auto byProb = [](const fd_detection& a, const fd_detection& b) { return a.probability > b.probability; };
    bool sortAtEnd = true;
    if (!roi) {
        std::vector<int> maxima;
        fd_host_block_nms_sparse(svmPos, p->img_w, p->img_h, 35, true, maxima);
        if (maxima.empty()) fd_host_block_nms_sparse(svmPos, p->img_w, p->img_h, 35, false, maxima);
        if (maxima.empty()) {
            sortAtEnd = false;  // "return svmPatchesPositive; // Should be empty." (:292-294), unsorted
        } else {
            std::sort(svmPos.begin(), svmPos.end(), byProb);
            std::vector<fd_detection> res;
            for (size_t i = 0; i + 1 < maxima.size(); i += 2) {
                const int x = maxima[i], y = maxima[i + 1];
                auto it = std::find_if(svmPos.begin(), svmPos.end(), [&](const fd_detection& a) { return a.cx == x && a.cy == y; });
                if (it != svmPos.end()) res.push_back(*it);
            }
            svmPos.swap(res);
        }
    }
    if (sortAtEnd) std::sort(svmPos.begin(), svmPos.end(), byProb);
    if (stage_counts) stage_counts[3] = (int)svmPos.size();
    *count = (int)svmPos.size();
    for (size_t i = 0; i < svmPos.size() && (int)i < cap && out; ++i) out[i] = svmPos[i];
    if (out && (int)svmPos.size() > cap) FD_THROW(FD_ERR_CAPACITY, "five-stage: %zu detections, capacity %d", svmPos.size(), cap);
}

// Stages 2-5 of FiveStageSlidingWindowDetector::detect (FiveStageSlidingWindowDetector.cpp:200-320 / :340-380) on a
// finished WVM run, in two halves so that a batch can keep the host busy while the GPU is: begin() does the host stages up to
// overlap elimination and queues the SVM on the survivors + its read-back on the stream `st` (event m->tailDone); end() waits
// for that event and finishes with the block NMS.
struct FiveStageTail {
    fd_ctx* ctx = nullptr;
    fd_pyramid* p = nullptr;
    fd_wvm* m = nullptr;
    const fd_svm* svm = nullptr;
    const int* roi = nullptr;
    fd_detection* out = nullptr;
    int cap = 0;
    int* count = nullptr;
    int32_t* stage_counts = nullptr;
    std::vector<fd_detection> wvmPos;
    std::vector<int> keep;
    size_t distOff = 0;
    bool pending = false, finished = false;
    std::chrono::steady_clock::time_point t0;

    void lap(const char* what) {
        static const bool trace = getenv("FD_TRACE") != nullptr;
        if (!trace) return;
        auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[fd five-stage] %-12s %8.1f us\n", what, std::chrono::duration<double, std::micro>(t1 - t0).count());
        t0 = t1;
    }

    void begin(fd_ctx* ctx_, fd_pyramid* p_, fd_wvm* m_, const fd_svm* svm_, const WvmRun& run, float oe_dist, float oe_ratio, int sx, int sy,
               const int* roi_, hipStream_t st, fd_detection* out_, int cap_, int* count_, int32_t* stage_counts_) {
        ctx = ctx_; p = p_; m = m_; svm = svm_; roi = roi_; out = out_; cap = cap_; count = count_; stage_counts = stage_counts_;
        t0 = std::chrono::steady_clock::now();
        // everything below launches on `st` explicitly and touches no shared context state: tails of different jobs may run on
        // different host threads (five_stage_batch_end)
        fd_wvm_positives_to_detections(p, m, run, sx, sy, wvmPos);
        if (stage_counts) stage_counts[0] = (int)wvmPos.size();
        lap("to_dets");
        // stage 2: overlap elimination
        fd_host_overlap_elimination(wvmPos.data(), (int)wvmPos.size(), oe_dist, oe_ratio, keep);
        if (stage_counts) stage_counts[1] = (int)keep.size();
        lap("oe");
        // stage 3: SVM on the survivors' HistEq64 patches (still resident in HBM, gathered by slot)
        if (!keep.empty()) {
            std::vector<uint32_t> slots(keep.size());
            for (size_t i = 0; i < keep.size(); ++i) slots[i] = run.slots[keep[i]];
            DevBuf& idx = m->all_level;  // reuse scratch (not used by this call)
            idx.reserve(sizeof(uint32_t) * slots.size());
            m->all_fout.reserve(sizeof(double) * slots.size());
            // pinned staging of this detector: [slots (u32) | distances (f64)].  The kernel reads the slot list from, and writes
            // the distances to, this host-mapped buffer directly (a few KB over the fabric instead of two blit kernels and their
            // launch gaps on the tail stream); FD_WVM_ZEROCOPY=0: explicit copies through device scratch.
            static const bool zcOff = [] { const char* e = getenv("FD_WVM_ZEROCOPY"); return e && atoi(e) == 0; }();
            distOff = (sizeof(uint32_t) * slots.size() + 15) & ~(size_t)15;
            m->h_tail.reserve(distOff + sizeof(double) * slots.size());
            char* pin = m->h_tail.as<char>();
            std::memcpy(pin, slots.data(), sizeof(uint32_t) * slots.size());
            if (zcOff) {
                HIP_CHECK(hipMemcpyAsync(idx.p, pin, sizeof(uint32_t) * slots.size(), hipMemcpyHostToDevice, st));
                fd_svm_generic_launch_on(st, svm, m->pos_patches.p, idx.as<uint32_t>(), (int64_t)m->dev.d, (int64_t)slots.size(), m->all_fout.as<double>());
                HIP_CHECK(hipMemcpyAsync(pin + distOff, m->all_fout.p, sizeof(double) * slots.size(), hipMemcpyDeviceToHost, st));
            } else {
                fd_svm_generic_launch_on(st, svm, m->pos_patches.p, (const uint32_t*)pin, (int64_t)m->dev.d, (int64_t)slots.size(), (double*)(pin + distOff));
            }
            if (!m->tailDone) HIP_CHECK(hipEventCreateWithFlags(&m->tailDone, hipEventDisableTiming));
            HIP_CHECK(hipEventRecord(m->tailDone, st));
            pending = true;
        }
        lap("svm launch");
    }

    // true when end() would not block
    bool ready() const { return !pending || hipEventQuery(m->tailDone) == hipSuccess; }

    void end() {
        if (finished) return;
        finished = true;
        t0 = std::chrono::steady_clock::now();
        std::vector<fd_detection> svmPos;
        if (pending) {
            HIP_CHECK(hipEventSynchronize(m->tailDone));
            pending = false;
            const double* dist = (const double*)(m->h_tail.as<char>() + distOff);
            for (size_t i = 0; i < keep.size(); ++i) {
                if (dist[i] >= (double)fd_svm_threshold(svm)) {  // strongClassifier->classify(): bool only
                    fd_detection d = wvmPos[keep[i]];
                    d.score = (float)dist[i];
                    d.positive = 1;
                    d.probability = 0.5;  // ClassifiedPatch(patch, bool) default probability (ClassifiedPatch.hpp:29-30)
                    svmPos.push_back(d);
                }
            }
        }
        lap("svm wait");
        five_stage_nms(p, roi, svmPos, out, cap, count, stage_counts);
        lap("nms");
    }
};

static void five_stage_tail(fd_ctx* ctx, fd_pyramid* p, fd_wvm* m, const fd_svm* svm, const WvmRun& run, float oe_dist, float oe_ratio,
                            int sx, int sy, const int* roi, hipStream_t st, fd_detection* out, int cap, int* count, int32_t* stage_counts) {
    FiveStageTail t;
    t.begin(ctx, p, m, svm, run, oe_dist, oe_ratio, sx, sy, roi, st, out, cap, count, stage_counts);
    t.end();
}

static void five_stage_check(const fd_wvm* m, const fd_svm* svm) {
    if (fd_svm_dim(svm) != m->dev.d || !fd_svm_is_u8(svm))
        FD_THROW(FD_ERR_INVALID_ARGUMENT, "second classifier must work on the %d-byte HistEq64 patch", m->dev.d);
}

// ---- stages 2-3 on the device (fs_tail.hpp) ------------------------------------------------------------------------------------
// Whether a five-stage run of (m, svm) can keep its tail on the device: the cascade ends in the dense stage B with its zero-copy
// header, the second classifier has the u8 RBF MFMA kernel, and the window ids of the call fit 32 bits.  FD_FS_TAIL=0: never.
static bool fst_possible(const fd_wvm* m, const fd_svm* svm, int nimg) {
    // FD_FS_TAIL (read per call: tests toggle it): 0 never, 1 always, unset: for multi-frame calls only.  A single frame has ~150
    // positives: the host sorts and sweeps them in ~7 us, less than k_fs_oe's launch + its 15 us (measured: 155 vs 167 us per blocking
    // single-frame call); a 64-frame call has ~10 K, 0.7 ms of host work that the device does in the shadow of the next call's kernels.
    const char* e = getenv("FD_FS_TAIL");
    const int mode = e ? atoi(e) : -1;
    if (mode == 0 || (mode != 1 && nimg < 2)) return false;
    return m->wvbOk && m->dev.numUsed > WVM_LCAP && fd_svm_u8_mfma_available(svm);
}
// The jobs of a batch (one frame each, fd_detect_five_stage_batch / fd_five_stage_batch_begin): overlap elimination on the device by
// k_fs_oe_big, whatever the number of positives (config 3 / 5: ~13 K per detector and frame, 0.9 ms of std::sort + painted map per job on a
// host thread) with FD_FS_TAIL=1.  Unset / 0: the host stages on the context's worker threads -- the default, because what the device tail
// buys depends on the HIP runtime's hardware queues (GPU_MAX_HW_QUEUES, read when the runtime starts): a job's tail is ~0.3 ms of ONE
// compute unit on the job's stream, and with the default four queues the streams behind such a kernel stall (config 3: 4.8 G patches/s
// against the host stages' 5.3 G on eight threads); with six queues the tails overlap and the device wins with a quarter of the host
// threads (5.6 G on two threads).  The results are the same bytes either way.
static bool fst_possible_batch(const fd_wvm* m, const fd_svm* svm) {
    const char* e = getenv("FD_FS_TAIL");
    if (!e || atoi(e) != 1) return false;
    return m->wvbOk && m->dev.numUsed > WVM_LCAP && fd_svm_u8_mfma_available(svm);
}
static bool spec_possible(const fd_wvm* m, const fd_svm* svm) {
    const char* e = getenv("FD_FS_SPEC");   // read per call: the tests compare both orders
    const bool off = e && atoi(e) == 0;
    return !off && m->wvbOk && m->dev.numUsed > WVM_LCAP && fd_svm_u8_mfma_available(svm);
}
static size_t fst_host_offsets(int nimg, int64_t cap, size_t& keepOff, size_t& distOff) {
    keepOff = (16 + sizeof(FstFrame) * (size_t)nimg + 15) & ~(size_t)15;
    distOff = (keepOff + sizeof(FstKeep) * (size_t)cap + 15) & ~(size_t)15;
    return distOff + sizeof(double) * (size_t)cap;
}
// queues k_fs_oe and the SVM stage behind the cascade of `run` (m->tailRun is set) and records m->tailDone
// big: k_fs_oe_big (one frame, any number of positives, painted map in device memory) instead of k_fs_oe (<= 1024 positives per frame)
static void fst_launch(fd_ctx* ctx, hipStream_t st, fd_pyramid* p, fd_wvm* m, const fd_svm* svm, const WvmRun& run, float oe_dist, float oe_ratio, int sx, int sy,
                       bool big = false) {
    const int nimg = p->nimg > 1 ? p->nimg : 1;
    if (big && nimg != 1) FD_THROW(FD_ERR_LOGIC, "k_fs_oe_big takes one frame");
    FstTable T;
    std::memset(&T, 0, sizeof(T));
    T.sx = sx; T.sy = sy; T.nimg = nimg;
    const int64_t perImage = run.total / nimg;
    T.perImage = (uint32_t)perImage;
    T.magic = perImage > 0 ? (uint32_t)(0xffffffffu / (uint32_t)perImage) : 0u;
    T.oeDist = oe_dist;
    T.oeRatio = ((oe_ratio > 0.0f) && (oe_ratio <= 1.0f)) ? oe_ratio : 0.0f;   // OverlapElimination's constructor
    T.logA = m->logisticA; T.logB = m->logisticB;
    for (const WindowLayer& w : run.wls) {
        if (w.nx == 0 || w.ny == 0) continue;
        FstLayer& l = T.l[T.n++];
        l.scale = p->all[p->kept[w.layer]].scale;
        l.first = (int32_t)w.first; l.nx = w.nx; l.bx = w.bx; l.by = w.by; l.ow = w.ow; l.oh = w.oh;
    }
    size_t keepOff, distOff;
    const size_t bytes = fst_host_offsets(nimg, m->pos_cap, keepOff, distOff);
    m->h_fst.reserve(bytes);
    m->fstSlots.reserve(sizeof(uint32_t) * (size_t)m->pos_cap);
    m->fstFrames = nimg;
    char* hb = m->h_fst.as<char>();
    FstIO io;
    io.pos = m->pos.as<PosRec>() + 1;
    io.posCount = m->fstHdr.as<unsigned int>() + 8;
    io.posCap = (unsigned int)m->pos_cap;
    io.hdr = m->fstHdr.as<FstHdr>();
    io.slots = m->fstSlots.as<uint32_t>();
    io.frameCount = m->fstFrameCount.as<unsigned int>();
    io.frameList = m->fstFrameList.as<uint32_t>();
    io.hostHdr = reinterpret_cast<uint32_t*>(hb);
    io.frames = reinterpret_cast<FstFrame*>(hb + 16);
    io.keep = reinterpret_cast<FstKeep*>(hb + keepOff);
    io.hostHdr[0] = 0xffffffffu;
    if (big) {
        FsbIO S;
        S.cap = (unsigned int)m->pos_cap;
        S.mapStride = (((p->img_w + 2 * FSB_PAD + 31) >> 5) + 3) & ~3;   // words per row, whole uint4 (the kernel zeroes the map 16 bytes at a time)
        S.mapH = p->img_h + 2 * FSB_PAD;
        m->fsbKeys.reserve(sizeof(unsigned long long) * 2 * (size_t)S.cap);
        m->fsbGeo.reserve(sizeof(int2) * (size_t)S.cap);
        m->fsbAcc.reserve(sizeof(unsigned int) * (size_t)S.cap);
        m->fsbMap.reserve(sizeof(unsigned int) * (size_t)S.mapStride * S.mapH);
        S.keyA = m->fsbKeys.as<unsigned long long>();
        S.keyB = S.keyA + S.cap;
        S.geo = m->fsbGeo.as<int2>();
        S.acc = m->fsbAcc.as<unsigned int>();
        S.map = m->fsbMap.as<unsigned int>();
        hipLaunchKernelGGL(k_fs_oe_big, dim3(1), dim3(FSB_T), 0, st, T, io, S);
    } else {
        hipLaunchKernelGGL(k_fs_oe, dim3((unsigned)nimg), dim3(256), 0, st, T, io);
    }
    HIP_CHECK(hipGetLastError());
    m->fstDirty = false;   // k_fs_oe is queued: it leaves the tail's counters clean for the next run
    // the SVM launch covers what the previous run kept, with a margin; a run that keeps more gets a second launch for the rest
    int64_t nmax = m->fstPrevKeep >= 0 ? m->fstPrevKeep * 2 + 256 : std::max<int64_t>(1024, run.total / 256);
    nmax = std::min<int64_t>(std::max<int64_t>(nmax, 256), m->pos_cap);
    m->fstLaunched = nmax;
    fd_svm_u8_mfma_launch_counted(st, svm, m->pos_patches.p, io.slots, (int64_t)m->dev.d, nmax, &io.hdr->svmCount, reinterpret_cast<double*>(hb + distOff));
    if (!m->tailDone) HIP_CHECK(hipEventCreateWithFlags(&m->tailDone, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(m->tailDone, st));
}
// Waits for a tail run and hands out its survivors.  false: the device gave up (ambiguous order, too many positives in a frame,
// stage-B queue overflow, ...): the caller runs the host stages (fd_wvm_finish + host elimination) on the same cascade results.
struct FstResult {
    const FstFrame* frames = nullptr;
    const FstKeep* keep = nullptr;
    const double* dist = nullptr;
};
static bool fst_collect(fd_ctx* ctx, fd_wvm* m, const fd_svm* svm, const WvmRun& run, FstResult& R) {
    HIP_CHECK(hipEventSynchronize(m->tailDone));
    const PosRec* hraw = m->h_pos.as<PosRec>();
    const unsigned int cnt = hraw[0].wid_lo;
    if (cnt == 0xffffffffu) FD_THROW(FD_ERR_HIP, "WVM stage B did not deliver its positive count");
    if (run.timed) wvm_read_timing(ctx);
    m->fstLastState = 0x100;
    if ((int64_t)hraw[0].wid_hi > m->deepCap || (int64_t)cnt > m->pos_cap) return false;   // overflow: fd_wvm_finish grows / reports
    char* hb = m->h_fst.as<char>();
    const uint32_t* hh = reinterpret_cast<const uint32_t*>(hb);
    if (hh[0] == 0xffffffffu) FD_THROW(FD_ERR_HIP, "five-stage tail did not deliver its survivor count");
    m->fstLastState = (int)hh[1];
    if (hh[1] != 0u) return false;
    wvm_finish_header(m, hraw);
    size_t keepOff, distOff;
    fst_host_offsets(m->fstFrames, m->pos_cap, keepOff, distOff);
    const int64_t total = (int64_t)hh[0];
    if (total > m->fstLaunched) {   // more survivors than the SVM launch covered: score the rest now
        const int64_t rest = total - m->fstLaunched;
        fd_svm_generic_launch_on(ctx->stream, svm, m->pos_patches.p, m->fstSlots.as<uint32_t>() + m->fstLaunched, (int64_t)m->dev.d, rest,
                                 reinterpret_cast<double*>(hb + distOff) + m->fstLaunched);
        HIP_CHECK(hipEventRecord(m->tailDone, ctx->stream));
        HIP_CHECK(hipEventSynchronize(m->tailDone));
    }
    m->fstPrevKeep = total;
    R.frames = reinterpret_cast<const FstFrame*>(hb + 16);
    R.keep = reinterpret_cast<const FstKeep*>(hb + keepOff);
    R.dist = reinterpret_cast<const double*>(hb + distOff);
    (void)run;
    return true;
}
// a survivor record -> the detection the reference's ClassifiedPatch stands for
static fd_detection fst_detection(const fd_pyramid* p, const fd_wvm* m, const WvmRun& run, int sx, int sy, const FstKeep& k) {
    fd_detection d;
    std::memset(&d, 0, sizeof(d));
    fd_window_to_detection(p, run.wls, sx, sy, (int64_t)k.wid, d);
    d.level = k.level;
    d.positive = 1;
    d.score = k.fout;
    d.probability = wvm_probability(m, (double)k.fout);
    return d;
}

// detection::FiveStageSlidingWindowDetector::detect, FiveStageSlidingWindowDetector.cpp:187-320 / :331-380
int fd_detect_five_stage(fd_ctx* ctx, fd_pyramid* p, const fd_wvm* wvm_, const fd_svm* svm, float oe_dist, float oe_ratio,
                         int sx, int sy, const int* roi, fd_detection* out, int cap, int* count, int32_t* stage_counts) {
    return fd_guard(ctx, [&] {
        if (!ctx || !p || !wvm_ || !svm || !count) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_detect_five_stage: NULL argument");
        fd_pyramid_require_single(p, "fd_detect_five_stage");
        fd_wvm* m = const_cast<fd_wvm*>(wvm_);
        five_stage_check(m, svm);
        // stage 1: WVM over all windows (SlidingWindowDetector::detect); stages 2-3 (overlap elimination, SVM) are queued behind it on
        // the device where the model allows (fs_tail.hpp): one wait instead of two round trips
        WvmRun run;
        m->tailWanted = fst_possible(m, svm, 1);
        // One frame without the device tail: the SVM scores of ALL its WVM positives (~150) are queued straight behind the cascade -- the
        // positive count stays on the device (CascadeOut::tail_count), the kernel covers twice the previous frame's count -- and the host
        // waits ONCE, then eliminates overlaps and looks the survivors' scores up.  A patch's score does not depend on what else is in
        // the launch, so the detections are the ones of the two-round-trip order (host elimination, then the survivors' SVM launch),
        // which stays the path of a frame with more positives than the launch covered, of a stage-B rerun, and of FD_FS_SPEC=0.
        m->specWanted = !m->tailWanted && spec_possible(m, svm);
        m->specLastState = -1;
        fd_wvm_launch(ctx, p, m, sx, sy, roi, false, run, ctx->kernel_timing);
        const bool specQueued = m->specRun;
        if (m->specRun) {
            int64_t nmax = m->specPrev >= 0 ? m->specPrev * 2 + 64 : 1024;
            nmax = std::min<int64_t>(std::max<int64_t>(nmax, 64), m->pos_cap);
            m->h_spec.reserve(sizeof(double) * (size_t)nmax);
            m->specLaunched = nmax;
            fd_svm_u8_mfma_launch_counted(ctx->stream, svm, m->pos_patches.p, nullptr, (int64_t)m->dev.d, nmax, m->specCnt.as<unsigned int>(), m->h_spec.as<double>());
            HIP_CHECK(hipEventRecord(m->done, ctx->stream));   // fd_wvm_finish's wait now covers the scores too
        }
        if (m->tailRun) {
            fst_launch(ctx, ctx->stream, p, m, svm, run, oe_dist, oe_ratio, sx, sy);
            FstResult R;
            if (fst_collect(ctx, m, svm, run, R)) {
                const FstFrame fr = R.frames[0];
                std::vector<fd_detection> svmPos;
                for (uint32_t j = 0; j < fr.nkeep; ++j) {
                    const double dv = R.dist[fr.base + j];
                    if (dv >= (double)fd_svm_threshold(svm)) {
                        fd_detection d = fst_detection(p, m, run, sx, sy, R.keep[fr.base + j]);
                        d.score = (float)dv;
                        d.probability = 0.5;   // ClassifiedPatch(patch, bool) default probability (ClassifiedPatch.hpp:29-30)
                        svmPos.push_back(d);
                    }
                }
                if (stage_counts) { stage_counts[0] = (int)fr.npos; stage_counts[1] = (int)fr.nkeep; }
                five_stage_nms(p, roi, svmPos, out, cap, count, stage_counts);
                return;
            }
            run.timed = false;   // read above
        }
        fd_wvm_finish(ctx, m, run);
        if (specQueued) m->specLastState = 1;
        if (m->specRun && (int64_t)run.pos.size() <= m->specLaunched) {
            m->specPrev = (int64_t)run.pos.size();
            m->specLastState = 0;
            std::vector<fd_detection> wvmPos, svmPos;
            std::vector<int> keep;
            fd_wvm_positives_to_detections(p, m, run, sx, sy, wvmPos);
            if (stage_counts) stage_counts[0] = (int)wvmPos.size();
            fd_host_overlap_elimination(wvmPos.data(), (int)wvmPos.size(), oe_dist, oe_ratio, keep);
            if (stage_counts) stage_counts[1] = (int)keep.size();
            const double* dist = m->h_spec.as<double>();
            for (size_t i = 0; i < keep.size(); ++i) {
                const double dv = dist[run.slots[keep[i]]];
                if (dv >= (double)fd_svm_threshold(svm)) {   // strongClassifier->classify(): bool only
                    fd_detection d = wvmPos[keep[i]];
                    d.score = (float)dv;
                    d.positive = 1;
                    d.probability = 0.5;   // ClassifiedPatch(patch, bool) default probability (ClassifiedPatch.hpp:29-30)
                    svmPos.push_back(d);
                }
            }
            five_stage_nms(p, roi, svmPos, out, cap, count, stage_counts);
            return;
        }
        if (specQueued) m->specPrev = (int64_t)run.pos.size();
        five_stage_tail(ctx, p, m, svm, run, oe_dist, oe_ratio, sx, sy, roi, ctx->stream, out, cap, count, stage_counts);
    });
}

// Detector::detect(const Mat& image) (Detector.hpp:59; FiveStageSlidingWindowDetector.cpp:187-190: update the extractor with the image,
// then detect): the pyramid update and the detection of ONE frame as one entry point -- what a caller of the reference's interface does
// per image, without a second trip through the binding between the two halves.
int fd_detect_five_stage_image(fd_ctx* ctx, fd_pyramid* p, const fd_wvm* wvm, const fd_svm* svm, const uint8_t* image, int width, int height,
                               int channels, int image_is_device, float oe_dist, float oe_ratio, int sx, int sy, const int* roi, fd_detection* out,
                               int cap, int* count, int32_t* stage_counts) {
    const int rc = fd_guard(ctx, [&] {
        if (!ctx || !p || !image) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_detect_five_stage_image: NULL argument");
        if (p->ctx != ctx) FD_THROW(FD_ERR_INVALID_ARGUMENT, "objects belong to different contexts");
        HIP_CHECK(hipSetDevice(ctx->device));
        fd_pyramid_update_on(p, image, width, height, channels, image_is_device, ctx->stream);
    });
    if (rc != FD_OK) return rc;
    return fd_detect_five_stage(ctx, p, wvm, svm, oe_dist, oe_ratio, sx, sy, roi, out, cap, count, stage_counts);
}

// FiveStageSlidingWindowDetector::detect on every frame of a multi-frame pyramid (fd_pyramid_set_frames /
// fd_pyramid_update_frames): ONE cascade run (pre-filter + stage B over the windows of all frames) and ONE SVM launch serve the
// whole call; the host stages (overlap elimination, NMS) run per frame.  A 640x480 frame is a chain of ~12 dependent launches
// and ~90 us of latency however many frames are in flight; here that chain is paid once per call.
struct fd_five_stage_frames {   // ticket of fd_detect_five_stage_frames_begin
    fd_pyramid* p = nullptr;
    fd_wvm* m = nullptr;
    const fd_svm* svm = nullptr;
    float oe_dist = 5.f, oe_ratio = 0.f;
    int sx = 1, sy = 1;
    bool has_roi = false;
    bool tail = false;                            // overlap elimination + SVM were queued on the device (fs_tail.hpp)
    int roi[4] = {0, 0, 0, 0};
    WvmRun run;
    std::vector<std::vector<fd_detection>> res;   // per frame, after NMS
    std::vector<int32_t> stages;                  // [frames][4]
    std::shared_ptr<FdAsyncTask> task;            // host stages in flight on fd_async_queue() (ticket entry points)
    ~fd_five_stage_frames() {                     // never freed under a running task
        if (task) { try { task->wait(); } catch (...) {} }
    }
};

static void five_stage_frames_begin(fd_ctx* ctx, fd_pyramid* p, const fd_wvm* wvm_, const fd_svm* svm, float oe_dist, float oe_ratio, int sx, int sy,
                                    const int* roi, fd_five_stage_frames& t) {
    if (!ctx || !p || !wvm_ || !svm) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_detect_five_stage_frames: NULL argument");
    t.p = p; t.m = const_cast<fd_wvm*>(wvm_); t.svm = svm; t.oe_dist = oe_dist; t.oe_ratio = oe_ratio; t.sx = sx; t.sy = sy;
    t.has_roi = roi != nullptr;
    if (roi) std::memcpy(t.roi, roi, sizeof(t.roi));
    five_stage_check(t.m, svm);
    t.m->tailWanted = fst_possible(t.m, svm, p->nimg);
    fd_wvm_launch(ctx, p, t.m, sx, sy, roi, false, t.run, ctx->kernel_timing);   // one cascade run over the windows of all frames
    t.tail = t.m->tailRun;
    if (t.tail) fst_launch(ctx, ctx->stream, p, t.m, svm, t.run, oe_dist, oe_ratio, sx, sy);   // overlap elimination + SVM behind it, no host in between
}

// Host stages of a multi-frame run (everything behind the cascade kernels): per-frame results into the ticket.  Runs on the calling
// thread (blocking entry point) or on a thread of fd_async_queue() (ticket entry points): touches the ticket, the handles the
// ticket holds and the context's stream only.
// FD_TRACE: where the host stages of the multi-frame calls spend their time (averages per call on stderr at process exit)
struct FramesHostTrace {
    const bool on = getenv("FD_TRACE") != nullptr;
    std::atomic<int64_t> ns[6], calls{0};
    FramesHostTrace() { for (auto& v : ns) v = 0; }
    ~FramesHostTrace() {
        const int64_t n = calls.load();
        if (!on || n == 0) return;
        static const char* name[6] = {"wait for the cascade + order positives", "positives -> detections", "overlap elimination", "SVM launch + wait", "NMS", "total"};
        for (int i = 0; i < 6; ++i) fprintf(stderr, "[fd frames host] %-40s %9.1f us per call (%lld calls)\n", name[i], ns[i].load() / 1e3 / n, (long long)n);
    }
};
static FramesHostTrace g_framesTrace;

static void five_stage_frames_host(fd_ctx* ctx, fd_five_stage_frames& t) {
    using clk = std::chrono::steady_clock;
    const bool tr = g_framesTrace.on;
    clk::time_point tp0 = tr ? clk::now() : clk::time_point(), tp = tp0;
    auto lap = [&](int i) {
        if (!tr) return;
        const clk::time_point n = clk::now();
        g_framesTrace.ns[i] += std::chrono::duration_cast<std::chrono::nanoseconds>(n - tp).count();
        tp = n;
    };
    fd_pyramid* p = t.p;
    fd_wvm* m = t.m;
    const fd_svm* svm = t.svm;
    const int* roi = t.has_roi ? t.roi : nullptr;
    WvmRun& run = t.run;
    const int NF = p->nimg;
    t.res.assign((size_t)NF, {});
    t.stages.assign((size_t)NF * 4, 0);
    if (t.tail) {
        FstResult R;
        if (fst_collect(ctx, m, svm, run, R)) {   // the device did stages 2-3: per frame the SVM's verdicts, then the block NMS
            lap(0);
            for (int f = 0; f < NF; ++f) {
                const FstFrame fr = R.frames[f];
                t.stages[4 * (size_t)f] = (int)fr.npos;
                t.stages[4 * (size_t)f + 1] = (int)fr.nkeep;
                std::vector<fd_detection>& svmPos = t.res[(size_t)f];
                for (uint32_t j = 0; j < fr.nkeep; ++j) {
                    const double dv = R.dist[fr.base + j];
                    if (dv >= (double)fd_svm_threshold(svm)) {
                        fd_detection d = fst_detection(p, m, run, t.sx, t.sy, R.keep[fr.base + j]);
                        d.score = (float)dv;
                        d.probability = 0.5;   // ClassifiedPatch(patch, bool) default probability (ClassifiedPatch.hpp:29-30)
                        svmPos.push_back(d);
                    }
                }
                int cnt = 0;
                five_stage_nms(p, roi, svmPos, nullptr, 0, &cnt, &t.stages[4 * (size_t)f]);
            }
            lap(4);
            if (tr) {
                g_framesTrace.ns[5] += std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now() - tp0).count();
                ++g_framesTrace.calls;
            }
            return;
        }
        run.timed = false;   // fst_collect has read the kernel timing
    }
    fd_wvm_finish(ctx, m, run);
    lap(0);
    const int64_t perImage = NF > 0 ? run.total / NF : 0;
    std::vector<fd_detection> dets;
    fd_wvm_positives_to_detections(p, m, run, t.sx, t.sy, dets);   // sorted by window id = by frame, extraction order inside
    lap(1);
    // frame boundaries, overlap elimination per frame, survivors of all frames -> one slot list
    std::vector<size_t> begin((size_t)NF + 1, dets.size());
    {
        size_t i = 0;
        for (int f = 0; f < NF; ++f) {
            begin[(size_t)f] = i;
            while (i < dets.size()) {
                const int64_t wid = (int64_t)(((uint64_t)run.pos[i].wid_hi << 32) | run.pos[i].wid_lo);
                if (perImage == 0 || wid / perImage != f) break;
                ++i;
            }
        }
        begin[(size_t)NF] = i;
    }
    std::vector<std::vector<int>> keep((size_t)NF);
    std::vector<uint32_t> slots;
    for (int f = 0; f < NF; ++f) {
        const size_t b = begin[(size_t)f], e = begin[(size_t)f + 1];
        t.stages[4 * (size_t)f] = (int)(e - b);
        if (e == b) continue;
        fd_host_overlap_elimination(dets.data() + b, (int)(e - b), t.oe_dist, t.oe_ratio, keep[(size_t)f]);
        t.stages[4 * (size_t)f + 1] = (int)keep[(size_t)f].size();
        for (int k : keep[(size_t)f]) slots.push_back(run.slots[b + (size_t)k]);
    }
    lap(2);
    const double* dist = nullptr;
    if (!slots.empty()) {   // the SVM stage of all frames: the kernel reads the slot list from / writes the distances to pinned memory
        hipStream_t st = ctx->stream;
        const size_t distOff = (sizeof(uint32_t) * slots.size() + 15) & ~(size_t)15;
        m->h_tail.reserve(distOff + sizeof(double) * slots.size());
        char* pin = m->h_tail.as<char>();
        std::memcpy(pin, slots.data(), sizeof(uint32_t) * slots.size());
        fd_svm_generic_launch_on(st, svm, m->pos_patches.p, (const uint32_t*)pin, (int64_t)m->dev.d, (int64_t)slots.size(), (double*)(pin + distOff));
        if (!m->tailDone) HIP_CHECK(hipEventCreateWithFlags(&m->tailDone, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(m->tailDone, st));
        HIP_CHECK(hipEventSynchronize(m->tailDone));   // only this call's SVM stage, not what the caller queued behind it
        dist = (const double*)(pin + distOff);
    }
    lap(3);
    size_t si = 0;
    for (int f = 0; f < NF; ++f) {
        std::vector<fd_detection>& svmPos = t.res[(size_t)f];
        const size_t b = begin[(size_t)f];
        for (int k : keep[(size_t)f]) {
            const double dv = dist[si++];
            if (dv >= (double)fd_svm_threshold(svm)) {
                fd_detection d = dets[b + (size_t)k];
                d.score = (float)dv;
                d.positive = 1;
                d.probability = 0.5;   // ClassifiedPatch(patch, bool) default probability (ClassifiedPatch.hpp:29-30)
                svmPos.push_back(d);
            }
        }
        int cnt = 0;
        five_stage_nms(p, roi, svmPos, nullptr, 0, &cnt, &t.stages[4 * (size_t)f]);   // svmPos becomes the frame's result
    }
    lap(4);
    if (tr) {
        g_framesTrace.ns[5] += std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now() - tp0).count();
        ++g_framesTrace.calls;
    }
}

static void five_stage_frames_end(fd_ctx* ctx, fd_five_stage_frames& t, fd_detection* out, int cap_per_frame, int32_t* counts, int32_t* stage_counts) {
    if (t.task) {
        std::shared_ptr<FdAsyncTask> task = std::move(t.task);
        t.task.reset();
        task->wait();   // rethrows what the host stages threw
    } else {
        five_stage_frames_host(ctx, t);
    }
    if (!counts || cap_per_frame < 0) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_detect_five_stage_frames: bad argument");
    const int NF = t.p->nimg;
    size_t most = 0;
    for (int f = 0; f < NF; ++f) {   // every frame's count first: a caller whose buffer is too small sizes the retry from them
        const std::vector<fd_detection>& r = t.res[(size_t)f];
        counts[f] = (int)r.size();
        most = std::max(most, r.size());
        if (stage_counts) std::memcpy(stage_counts + 4 * f, &t.stages[4 * (size_t)f], 4 * sizeof(int32_t));
        for (size_t i = 0; i < r.size() && (int)i < cap_per_frame && out; ++i) out[(size_t)f * cap_per_frame + i] = r[i];
    }
    if (out && (int)most > cap_per_frame) FD_THROW(FD_ERR_CAPACITY, "five-stage: %zu detections in one frame, capacity %d", most, cap_per_frame);
}

int fd_detect_five_stage_frames(fd_ctx* ctx, fd_pyramid* p, const fd_wvm* wvm_, const fd_svm* svm, float oe_dist, float oe_ratio, int sx,
                                int sy, const int* roi, fd_detection* out, int cap_per_frame, int32_t* counts, int32_t* stage_counts) {
    return fd_guard(ctx, [&] {
        fd_five_stage_frames t;
        five_stage_frames_begin(ctx, p, wvm_, svm, oe_dist, oe_ratio, sx, sy, roi, t);
        five_stage_frames_end(ctx, t, out, cap_per_frame, counts, stage_counts);
    });
}

int fd_detect_five_stage_frames_begin(fd_ctx* ctx, fd_pyramid* p, const fd_wvm* wvm_, const fd_svm* svm, float oe_dist, float oe_ratio, int sx,
                                      int sy, const int* roi, fd_five_stage_frames** ticket) {
    if (ticket) *ticket = nullptr;
    return fd_guard(ctx, [&] {
        if (!ticket) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_detect_five_stage_frames_begin: NULL ticket");
        std::unique_ptr<fd_five_stage_frames> t(new fd_five_stage_frames());
        five_stage_frames_begin(ctx, p, wvm_, svm, oe_dist, oe_ratio, sx, sy, roi, *t);
        // the host stages follow on a queue thread as soon as the cascade kernels retire (zero-copy read-back runs only: the
        // other read-back path and the kernel timer use per-context state); FD_FRAMES_ASYNC=0 keeps them inside _end
        static const bool asyncOn = [] { const char* e = getenv("FD_FRAMES_ASYNC"); return !e || atoi(e) != 0; }();
        if (asyncOn && t->m->zcRun && !t->run.timed) {
            fd_five_stage_frames* tp = t.get();
            tp->task = fd_async_queue().submit([ctx, tp] {
                HIP_CHECK(hipSetDevice(ctx->device));
                five_stage_frames_host(ctx, *tp);
            });
        }
        *ticket = t.release();
    });
}

int fd_detect_five_stage_frames_end(fd_ctx* ctx, fd_five_stage_frames* ticket, fd_detection* out, int cap_per_frame, int32_t* counts,
                                    int32_t* stage_counts) {
    std::unique_ptr<fd_five_stage_frames> t(ticket);   // released whatever happens
    return fd_guard(ctx, [&] {
        if (!ctx || !t) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_detect_five_stage_frames_end: NULL argument");
        five_stage_frames_end(ctx, *t, out, cap_per_frame, counts, stage_counts);
    });
}

struct fd_five_stage_batch {
    fd_five_stage_job* jobs = nullptr;
    int n = 0;
    std::vector<WvmRun> runs;
    std::vector<char> tail;   // job i: overlap elimination + SVM were queued on the device behind its cascade (k_fs_oe_big)
    // ticket entry points: the host stages of job i as a task of fd_batch_queue() (null: taken by fd_five_stage_batch_end itself)
    std::vector<std::shared_ptr<FdAsyncTask>> tasks;
    std::vector<FiveStageTail> tails;
    std::vector<int> counts;
    ~fd_five_stage_batch() {   // the tasks reference this object: never released while one is running
        for (std::shared_ptr<FdAsyncTask>& t : tasks)
            if (t) { try { t->wait(); } catch (...) {} }
    }
};

// The host stages of the jobs of a batch in flight, one task per job on the process-wide batch queue: a worker waits for the job's
// cascade, reads its positives back, runs the overlap elimination, queues the SVM stage on the shared high-priority stream, waits
// for it and finishes with the NMS -- everything a task touches belongs to its job.  With the stages inside _end (the blocking entry
// point) a frame's _end lasts as long as its heaviest detector's tail (config 3 on busy content: one detector with 70 K WVM positives,
// ~5.5 ms of serial host work, while the other fourteen are done after ~2.5 ms and the GPU idles half of the time); as tasks the
// tails of the frames in flight overlap each other and the host stages run at (sum of the work) / threads.  FD_BATCH_ASYNC=0 keeps them in _end.
static void five_stage_batch_submit(fd_ctx* ctx, fd_five_stage_batch& b) {
    if (const char* e = getenv("FD_BATCH_ASYNC")) if (atoi(e) == 0) return;
    const int n = b.n;
    if (n < 2) return;
    for (int i = 0; i < n; ++i)
        if (b.runs[(size_t)i].timed) return;
    hipStream_t tailStream = fd_tail_stream(ctx);   // created here, on the caller's thread
    (void)fd_aux_stream(ctx);
    b.tasks.assign((size_t)n, nullptr);
    b.tails.assign((size_t)n, FiveStageTail());
    b.counts.assign((size_t)n, 0);
    std::vector<int> order;
    for (int i = 0; i < n; ++i)
        if (!b.tail[(size_t)i]) order.push_back(i);
    std::stable_sort(order.begin(), order.end(), [&](int a, int c) { return b.jobs[a].wvm->prevPos > b.jobs[c].wvm->prevPos; });   // heaviest first
    fd_five_stage_batch* bp = &b;
    for (int i : order) {
        b.tasks[(size_t)i] = fd_batch_queue().submit([ctx, bp, i, tailStream] {
            HIP_CHECK(hipSetDevice(ctx->device));
            fd_five_stage_job& j = bp->jobs[i];
            fd_wvm* m = const_cast<fd_wvm*>(j.wvm);
            fd_wvm_finish(ctx, m, bp->runs[(size_t)i]);
            FiveStageTail& t = bp->tails[(size_t)i];
            t.begin(ctx, j.pyramid, m, j.svm, bp->runs[(size_t)i], j.oe_dist, j.oe_ratio, j.step_x, j.step_y, j.roi, tailStream, j.out, j.cap,
                    &bp->counts[(size_t)i], j.stage_counts);
            t.end();
            j.count = bp->counts[(size_t)i];
        });
    }
}

static void five_stage_batch_begin(fd_ctx* ctx, fd_five_stage_job* jobs, int n, fd_five_stage_batch& b) {
    if (!ctx || n < 0 || (n > 0 && !jobs)) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_detect_five_stage_batch: bad argument");
    const auto tBegin0 = std::chrono::steady_clock::now();
    b.jobs = jobs;
    b.n = n;
    b.runs.assign((size_t)n, WvmRun());
    b.tail.assign((size_t)n, 0);
    for (int i = 0; i < n; ++i) {
        fd_five_stage_job& j = jobs[i];
        j.count = 0;
        j.status = FD_OK;
        if (!j.pyramid || !j.wvm || !j.svm) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_detect_five_stage_batch: NULL handle in job %d", i);
        fd_pyramid_require_single(j.pyramid, "fd_detect_five_stage_batch");
        for (int k = 0; k < i; ++k)
            if (jobs[k].wvm == j.wvm) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_detect_five_stage_batch: jobs %d and %d share a WVM handle", k, i);
        five_stage_check(j.wvm, j.svm);
    }
    // jobs are spread over a few streams so that the small kernels of different jobs (pyramid levels, the deep
    // cascade stage, the SVM stage) overlap each other; a job's optional frame upload / pyramid update runs on its stream
    for (int i = 0; i < n; ++i) {
        (void)fd_pool_stream(ctx, i);
        if (!jobs[i].image) continue;
        for (int k = 0; k < i; ++k)
            if (jobs[k].image && jobs[k].pyramid == jobs[i].pyramid)
                FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_detect_five_stage_batch: jobs %d and %d update the same pyramid", k, i);
    }
    auto updateJob = [&](int i) {
        fd_five_stage_job& j = jobs[i];
        if (j.image) fd_pyramid_update_on(j.pyramid, j.image, j.image_w, j.image_h, j.image_channels, j.image_is_device, fd_pool_stream(ctx, i));
    };
    // Detectors that scan the same windows -- one pyramid, one patch size, one stepping, no ROI -- share their dense pre-filter
    // (k_wvm_prefilter_group: seven of ffpDetectApp's fifteen detectors are one such group, the ears and the profile faces two more).
    // A unit is a group (or a single job), the largest groups first.
    std::vector<std::vector<int>> units;
    for (int i = 0; i < n; ++i) {
        const fd_five_stage_job& j = jobs[i];
        bool placed = false;
        if (!j.roi && fd_wvm_groupable(j.wvm)) {
            for (std::vector<int>& u : units) {
                const fd_five_stage_job& h = jobs[u[0]];
                if ((int)u.size() < WVD_GMAX && !h.roi && fd_wvm_groupable(h.wvm) && h.pyramid == j.pyramid && h.step_x == j.step_x && h.step_y == j.step_y &&
                    h.wvm->dev.fw == j.wvm->dev.fw && h.wvm->dev.fh == j.wvm->dev.fh) {
                    u.push_back(i);
                    placed = true;
                    break;
                }
            }
        }
        if (!placed) units.push_back(std::vector<int>(1, i));
    }
    std::stable_sort(units.begin(), units.end(), [](const std::vector<int>& a, const std::vector<int>& c) { return a.size() > c.size(); });
    const int nunits = (int)units.size();
    auto tailOf = [&](int i) {
        fd_five_stage_job& j = jobs[i];
        fd_wvm* m = const_cast<fd_wvm*>(j.wvm);
        if (m->tailRun) {
            // (on the job's own stream.  k_fs_oe_big is ONE workgroup working for ~0.3 ms per job, 4.7 ms per 15-detector frame: dedicated
            // tail streams behind the cascades' events were measured and lost -- 1 / 2 / 4 of them: 3670 / 4426 / 4926 Mpatches/s against
            // 5565 on the jobs' streams with six hardware queues -- the tails of a frame have to overlap each other)
            hipStream_t ts = fd_pool_stream(ctx, i);
            fst_launch(ctx, ts, j.pyramid, m, j.svm, b.runs[i], j.oe_dist, j.oe_ratio, j.step_x, j.step_y, true);
            b.tail[i] = 1;
        }
    };
    auto cascadeJob = [&](int ui) {
        const std::vector<int>& u = units[(size_t)ui];
        if (u.size() > 1) {
            hipStream_t sts[WVD_GMAX];
            fd_wvm* ms[WVD_GMAX];
            WvmRun* rs[WVD_GMAX];
            for (size_t q = 0; q < u.size(); ++q) {
                fd_five_stage_job& j = jobs[u[q]];
                ms[q] = const_cast<fd_wvm*>(j.wvm);
                ms[q]->tailWanted = fst_possible_batch(ms[q], j.svm);
                sts[q] = fd_pool_stream(ctx, u[q]);
                rs[q] = &b.runs[u[q]];
            }
            fd_wvm_launch_group(ctx, sts, jobs[u[0]].pyramid, ms, (int)u.size(), jobs[u[0]].step_x, jobs[u[0]].step_y, rs);
            for (int i : u) tailOf(i);
            return;
        }
        const int i = u[0];
        fd_five_stage_job& j = jobs[i];
        fd_wvm* m = const_cast<fd_wvm*>(j.wvm);
        // stages 2-3 on the device where the model allows (fs_tail.hpp: k_fs_oe_big + the counted SVM launch behind the cascade, on the
        // job's stream); everything here touches the job's own handles only (the jobs of a batch may be queued by different threads)
        m->tailWanted = !j.roi && fst_possible_batch(m, j.svm);
        fd_wvm_launch_on(ctx, fd_pool_stream(ctx, i), j.pyramid, m, j.step_x, j.step_y, j.roi, false, b.runs[i], false);
        tailOf(i);
    };
    // A frame costs ~15 runtime calls (pyramid kernels, cascade kernels, copies, events): with many small jobs the single host
    // thread issuing them is the bottleneck, so batches of >= 6 jobs are issued by the worker pool -- all pyramid updates first
    // (jobs may share a pyramid one of them updates), then all cascades.  Each job has its own stream, handles and buffers.
    static const int nthreads = [] { const char* e = getenv("FD_BATCH_THREADS"); const int v = e ? atoi(e) : 8; return v < 1 ? 1 : (v > 16 ? 16 : v); }();
    if (nthreads > 1 && n >= 6) {
        if (!ctx->workers) ctx->workers.reset(new FdWorkerPool(nthreads - 1));
        std::mutex errMu;
        FdError firstErr{FD_OK, std::string()};
        auto phase = [&](const std::function<void(int)>& f, int count) {
            std::atomic<int> next{0};
            ctx->workers->run([&] {
                (void)hipSetDevice(ctx->device);
                for (int i; (i = next.fetch_add(1)) < count;) {
                    // nothing may leave a pool thread's body (std::terminate): vector / DevBuf growth can throw bad_alloc etc.
                    try { f(i); } catch (const FdError& e) {
                        std::lock_guard<std::mutex> lk(errMu);
                        if (firstErr.code == FD_OK) firstErr = e;
                    } catch (const std::exception& e) {
                        std::lock_guard<std::mutex> lk(errMu);
                        if (firstErr.code == FD_OK) firstErr = FdError{FD_ERR_RUNTIME, e.what()};
                    } catch (...) {
                        std::lock_guard<std::mutex> lk(errMu);
                        if (firstErr.code == FD_OK) firstErr = FdError{FD_ERR_RUNTIME, "unknown error on a batch worker thread"};
                    }
                }
            });
            if (firstErr.code != FD_OK) throw firstErr;
        };
        bool anyImage = false;
        for (int i = 0; i < n; ++i) anyImage |= jobs[i].image != nullptr;
        if (anyImage) phase(updateJob, n);
        phase(cascadeJob, nunits);
    } else {
        for (int i = 0; i < n; ++i) updateJob(i);
        for (int ui = 0; ui < nunits; ++ui) cascadeJob(ui);
    }
    static const bool trace = getenv("FD_TRACE") != nullptr;
    if (trace)
        fprintf(stderr, "[fd batch] begin: %d cascades queued in %.1f us\n", n,
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tBegin0).count());
}

static void five_stage_batch_end(fd_ctx* ctx, fd_five_stage_batch& b) {
    fd_five_stage_job* jobs = b.jobs;
    const int n = b.n;
    static const bool trace = getenv("FD_TRACE") != nullptr;
    const auto tEnd0 = std::chrono::steady_clock::now();
    double waitUs = 0;
    // The host takes the detectors one by one: as soon as a cascade is done, its positives are read back and
    // thinned out by the overlap elimination while the GPU works on the later cascades; the SVM stage of the survivors is
    // only queued (high-priority stream), and its NMS runs whenever the result has arrived, at the latest after the loop.
    int firstError = FD_OK;
    std::vector<FiveStageTail> tails((size_t)n);
    std::vector<int> counts((size_t)n, 0);
    auto fail = [&](int i, const FdError& e) {   // the remaining jobs are still collected; the first failure is reported
        jobs[i].status = e.code;
        if (firstError == FD_OK) { firstError = e.code; ctx->error = e.msg; }
    };
    auto finish = [&](int i) {
        try {
            tails[i].end();
            jobs[i].count = counts[i];
        } catch (const FdError& e) { fail(i, e); }
    };
    // Large batches (config 3: 15 detectors, thousands of WVM positives each) take ~1 ms of host work per detector -- more than
    // the GPU needs for its cascade since the dense pre-filter -- so the detectors are handed to a few host threads: each worker
    // claims a detector whose cascade has finished, reads its positives back, runs the overlap elimination, queues the SVM stage
    // on the shared high-priority stream, waits for it and finishes with the NMS.  Everything a worker touches belongs to its
    // job (WVM handle, pinned staging, events); the streams are created up front.  FD_BATCH_THREADS=1 keeps it on the caller.
    static const int nthreads = [] { const char* e = getenv("FD_BATCH_THREADS"); const int v = e ? atoi(e) : 8; return v < 1 ? 1 : (v > 16 ? 16 : v); }();
    // ---- jobs whose stages 2-3 ran on the device: the SVM's verdicts, then the block NMS (a few hundred elements per job: this thread).
    // A job whose device tail gave up (overlapping ties, an overflow, parameters the painted map does not cover) or whose SVM positives
    // hold two equal WVM outputs (fs_tail.hpp: the order of tied survivors is the reference's std::sort's) takes the host stages below.
    std::vector<char> doneByTail((size_t)n, 0);
    int ndone = 0;
    if (!b.tasks.empty()) {   // host stages already running (or done) on the batch queue
        for (int i = 0; i < n; ++i) {
            if (!b.tasks[(size_t)i]) continue;
            std::shared_ptr<FdAsyncTask> task = std::move(b.tasks[(size_t)i]);
            try { task->wait(); } catch (const FdError& e) { fail(i, e); } catch (const std::exception& e) { fail(i, FdError{FD_ERR_RUNTIME, e.what()}); }
            doneByTail[(size_t)i] = 1;
            tails[(size_t)i].finished = true;
            ++ndone;
        }
    }
    for (int i = 0; i < n; ++i) {
        if (!b.tail[i]) continue;
        fd_five_stage_job& j = jobs[i];
        fd_wvm* m = const_cast<fd_wvm*>(j.wvm);
        try {
            FstResult R;
            if (!fst_collect(ctx, m, j.svm, b.runs[i], R)) continue;
            const FstFrame fr = R.frames[0];
            std::vector<fd_detection> svmPos;
            std::vector<uint32_t> bits;
            for (uint32_t q = 0; q < fr.nkeep; ++q) {
                const double dv = R.dist[fr.base + q];
                if (dv >= (double)fd_svm_threshold(j.svm)) {
                    const FstKeep& k = R.keep[fr.base + q];
                    fd_detection d = fst_detection(j.pyramid, m, b.runs[i], j.step_x, j.step_y, k);
                    d.score = (float)dv;
                    d.probability = 0.5;   // ClassifiedPatch(patch, bool) default probability (ClassifiedPatch.hpp:29-30)
                    svmPos.push_back(d);
                    uint32_t fb;
                    std::memcpy(&fb, &k.fout, 4);
                    bits.push_back(fb);
                }
            }
            std::sort(bits.begin(), bits.end());
            if (std::adjacent_find(bits.begin(), bits.end()) != bits.end()) { m->fstLastState = 0x200; continue; }   // tied survivors among the SVM positives
            if (j.stage_counts) { j.stage_counts[0] = (int)fr.npos; j.stage_counts[1] = (int)fr.nkeep; }
            five_stage_nms(j.pyramid, j.roi, svmPos, j.out, j.cap, &counts[i], j.stage_counts);
            j.count = counts[i];
            doneByTail[i] = 1;
            ++ndone;
        } catch (const FdError& e) {
            doneByTail[i] = 1;
            ++ndone;
            fail(i, e);
        }
    }
    if (ndone == n) {
        if (firstError != FD_OK) throw FdError{firstError, ctx->error};
        return;
    }
    int64_t totalWindows = 0;
    for (int i = 0; i < n; ++i) totalWindows += b.runs[i].total;
    if (nthreads > 1 && (n - ndone >= 6 || totalWindows >= (int64_t)4 << 20) && n - ndone >= 2) {
        if (!ctx->workers) ctx->workers.reset(new FdWorkerPool(nthreads - 1));
        hipStream_t tailStream = fd_tail_stream(ctx);
        (void)fd_aux_stream(ctx);
        std::vector<std::atomic<char>> claimed((size_t)n);
        for (int i = 0; i < n; ++i) claimed[i].store(doneByTail[i]);
        std::mutex errMu;
        auto cascadeDone = [&](int i) {
            return b.runs[i].total == 0 || hipEventQuery(jobs[i].wvm->done) != hipErrorNotReady;
        };
        auto work = [&] {
            (void)hipSetDevice(ctx->device);
            for (;;) {
                int pick = -1;
                bool open = false;
                for (int i = 0; i < n && pick < 0; ++i) {
                    if (claimed[i].load(std::memory_order_relaxed)) continue;
                    open = true;
                    if (cascadeDone(i) && claimed[i].exchange(1) == 0) pick = i;
                }
                if (pick < 0) {
                    if (!open) return;
                    std::this_thread::yield();
                    continue;
                }
                fd_five_stage_job& j = jobs[pick];
                fd_wvm* m = const_cast<fd_wvm*>(j.wvm);
                try {
                    fd_wvm_finish(ctx, m, b.runs[pick]);
                    tails[pick].begin(ctx, j.pyramid, m, j.svm, b.runs[pick], j.oe_dist, j.oe_ratio, j.step_x, j.step_y, j.roi,
                                      tailStream, j.out, j.cap, &counts[pick], j.stage_counts);
                    tails[pick].end();
                    j.count = counts[pick];
                } catch (const FdError& e) {
                    std::lock_guard<std::mutex> lk(errMu);
                    tails[pick].finished = true;
                    fail(pick, e);
                } catch (const std::exception& e) {   // pool thread: nothing else may escape
                    std::lock_guard<std::mutex> lk(errMu);
                    tails[pick].finished = true;
                    fail(pick, FdError{FD_ERR_RUNTIME, e.what()});
                } catch (...) {
                    std::lock_guard<std::mutex> lk(errMu);
                    tails[pick].finished = true;
                    fail(pick, FdError{FD_ERR_RUNTIME, "unknown error on a batch worker thread"});
                }
            }
        };
        ctx->workers->run(work);
        if (trace)
            fprintf(stderr, "[fd batch] end: %.1f us in total on %d host threads\n",
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tEnd0).count(), nthreads);
        if (firstError != FD_OK) throw FdError{firstError, ctx->error};
        return;
    }
    // Detectors are taken in the order their cascades complete, not in job order: the streams of the pool do not advance evenly
    // (a trace showed the first job's event 34 ms into a 38 ms frame while jobs 1..14 had long finished), and blocking on job 0
    // would push every host stage behind the last cascade.  While nothing is ready the host polls (events of the cascades, then of
    // the queued SVM stages).
    std::vector<char> begun((size_t)n, 0);
    int nbegun0 = 0;
    for (int i = 0; i < n; ++i)
        if (doneByTail[i]) { begun[i] = 1; tails[i].finished = true; ++nbegun0; }
    auto cascadeReady = [&](int i) {
        const fd_wvm* m = jobs[i].wvm;
        return b.runs[i].total == 0 || hipEventQuery(m->done) != hipErrorNotReady;   // an error surfaces in fd_wvm_finish
    };
    for (int nbegun = nbegun0; nbegun < n;) {
        int pick = -1;
        for (int i = 0; i < n && pick < 0; ++i)
            if (!begun[i] && cascadeReady(i)) pick = i;
        if (pick < 0) {   // nothing to start: use the time for the NMS of a detector whose SVM stage has arrived, else yield
            bool did = false;
            for (int k = 0; k < n && !did; ++k)
                if (begun[k] && !tails[k].finished && tails[k].ready()) { finish(k); did = true; }
            if (!did) std::this_thread::yield();
            continue;
        }
        const int i = pick;
        begun[i] = 1;
        ++nbegun;
        fd_five_stage_job& j = jobs[i];
        fd_wvm* m = const_cast<fd_wvm*>(j.wvm);
        try {
            const auto tw0 = std::chrono::steady_clock::now();
            fd_wvm_finish(ctx, m, b.runs[i]);
            if (trace) {
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tw0).count();
                waitUs += us;
                fprintf(stderr, "[fd batch] job %2d cascade wait + read-back %8.1f us (%zu positives)\n", i, us, b.runs[i].pos.size());
            }
            tails[i].begin(ctx, j.pyramid, m, j.svm, b.runs[i], j.oe_dist, j.oe_ratio, j.step_x, j.step_y, j.roi,
                           fd_tail_stream(ctx), j.out, j.cap, &counts[i], j.stage_counts);
        } catch (const FdError& e) {
            tails[i].finished = true;
            fail(i, e);
        }
        for (int k = 0; k < n; ++k)
            if (begun[k] && !tails[k].finished && tails[k].ready()) finish(k);
    }
    for (int i = 0; i < n; ++i)
        if (!tails[i].finished) finish(i);
    if (trace)
        fprintf(stderr, "[fd batch] end: %.1f us in total, %.1f us of it waiting for cascades\n",
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tEnd0).count(), waitUs);
    if (firstError != FD_OK) throw FdError{firstError, ctx->error};
}

// Several five-stage detectors on (possibly shared) pyramids, as ffpDetectApp.cpp:557-600 loops over its detectors.  All
// cascades are queued first (begin); while the GPU works through them the host finishes the detectors one by one (end).
int fd_detect_five_stage_batch(fd_ctx* ctx, fd_five_stage_job* jobs, int n) {
    return fd_guard(ctx, [&] {
        fd_five_stage_batch b;
        five_stage_batch_begin(ctx, jobs, n, b);
        five_stage_batch_end(ctx, b);
    });
}

int fd_five_stage_batch_begin(fd_ctx* ctx, fd_five_stage_job* jobs, int n, fd_five_stage_batch** ticket) {
    if (ticket) *ticket = nullptr;
    return fd_guard(ctx, [&] {
        if (!ticket) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_five_stage_batch_begin: NULL ticket");
        std::unique_ptr<fd_five_stage_batch> b(new fd_five_stage_batch());
        five_stage_batch_begin(ctx, jobs, n, *b);
        five_stage_batch_submit(ctx, *b);
        *ticket = b.release();
    });
}

int fd_five_stage_batch_end(fd_ctx* ctx, fd_five_stage_batch* ticket) {
    std::unique_ptr<fd_five_stage_batch> b(ticket);   // released whatever happens
    return fd_guard(ctx, [&] {
        if (!ctx || !b) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_five_stage_batch_end: NULL argument");
        five_stage_batch_end(ctx, *b);
    });
}

