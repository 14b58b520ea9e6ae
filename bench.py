#!/usr/bin/env python
"""bench.py -- BASELINE.json metric: "Mpatches/s (extract+WVM+SVM) per GPU, 640x480 pyramid; SDM iters/s".

Headline (N=1 and N>1): the FaceFrontal.cfg five-stage cascade (FiveStageSlidingWindowDetector.cpp:187-320: pyramid -> sliding
HistEq64 windows -> WVM -> overlap elimination -> RBF-SVM -> block NMS) on 640x480 frames, through the product entry point
fd_detect_five_stage_batch; every frame's fd_detection records reach the host inside the timed region.  One step =
--frames-per-step frames (default 512, so that 20 steps take about a second).

The other BASELINE configs ride along as sub-records under "also" (same JSON line), each with its own ms_per_step, roofline and
cpu_baseline:
  hog_svm  config 2: 640x480, 21-layer pyramid, 20x20 windows stride 2, HOG-324 + RBF-SVM 1024 SV   (fd_detect_hog_svm_begin/_end)
  ffp15    config 3: the 15 detectors of ffpDetectApp/*.cfg on 1920x1080 frames, 32 distinct ones    (fd_five_stage_batch_begin/_end)
  sdm      config 4: 256 face crops x 68 landmarks x 4 cascade steps                                 (fd_sdm_fit_batch)
  config5  config 5: a 10,000-image 1920x1080 batch through the 15 detectors, image i -> rank i mod N, the detection records gathered
           every 256 images (fd_dist_gather_records = ONE ncclAllGather); a FIXED job ("scaling": "strong"), steps = gathers
Frames are resident in HBM before the timed region.  Multi-GPU (config 5): one process per GPU (torch.distributed, RCCL), image i
-> rank i mod N, no data-path collective, ONE all_gather of the real detection records {image, detector, cx, cy, w, h, score,
prob} every --gather-every steps.  `python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run.

The cpu_baseline legs (rank 0, N=1 only) time the CPU oracle (oracle/, "port" of the reference's single-threaded CPU path) on a
bounded sample of the same workload: 1 thread with the update / extract / classify split of BASELINE.md section 3, and an
image-parallel (detector-parallel for ffp15) run on every host core."""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import threading
import time

# hardware queues of the HIP runtime (read when it initialises, i.e. at the first HIP call of the process).  Four, the runtime's
# default, since round 4: with the device tail a multi-frame call is one chain of ~14 kernels on its own stream, and six such
# streams on four queues ran 3 % faster than on eight (3062-3104 vs 2974-2998 Mpatches/s, five paired runs; 2, 3 and 5 queues lost
# 5 %); the 15-detector batch (config 3) measures the same either way.  Rounds 1-3 set eight
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X dense f32 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
PEAK_VALU_GINST = 256 * 4 * 2.4 / 4.0   # wave64 VALU instructions per ns: 1024 SIMD16 x 2.4 GHz / 4 cycles per wave64 instruction
PEAK_I8_MFMA_TOPS = 3944.0     # MI355X dense int8 MFMA (MI355X_MICROARCH.md: ">= 3944 TOPS")


# ---------------------------------------------------------------------------------------------------------------- helpers
def host_info():
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return model, cores


def pmc_record(workload, kernel_substr):
    """Counter figures per launch of a kernel from the committed rocprofv3 --pmc passes (profiles/r04_pmc.json, written by
    tools/pmc_summary.py from runs of this script; every entry names the command and the git head it was measured at)."""
    try:
        # the newest committed pass that holds this workload (a round only re-measures the workloads whose kernels it touched)
        rec = None
        fname = None
        for r in (6, 5, 4, 3, 2):
            p_ = os.path.join(ROOT, "profiles", "r0%d_pmc.json" % r)
            if os.path.exists(p_):
                cand = json.load(open(p_))
                if workload in cand:
                    rec, fname = cand, "profiles/r0%d_pmc.json" % r
                    break
        ks = rec.get(workload, {}).get("kernels", {})
        agg = "k_all(%s)" % kernel_substr
        for k, v in ([(agg, ks[agg])] if agg in ks else []) + list(ks.items()):
            if kernel_substr in k:
                out = dict(v)
                out["kernel"] = k
                # where the counters come from: the committed file and the git head of the pass (the command is in the file's own
                # `source` field; a line the driver has to parse does not repeat it)
                note = rec[workload].get("source", "")
                hd = note.rsplit("git head ", 1)[-1].strip()[:12] if "git head " in note else "?"
                out["source"] = "%s[%s] at git %s" % (fname, workload, hd)
                return out
    except Exception:
        pass
    return None


def issue_roofline(pm):
    """VALU-issue roofline of the cascade / descriptor kernels (they are bound by instruction issue, not by HBM): fraction of the
    wave64 VALU issue slots (one per 4 cycles per SIMD16, 1024 SIMDs) used while the kernels ran, from the committed counter passes
    (SQ_INSTS_VALU, GRBM_GUI_ACTIVE; tools/pmc_summary.py) -- counters cannot be read inside a plain run."""
    f = float(pm["valu_issue_frac"])
    out = dict(bound="valu-issue", achieved=f * PEAK_VALU_GINST, peak=PEAK_VALU_GINST, unit="G wave-instr/s", frac=f, kernel=pm.get("kernel"),
               source=pm.get("source"), formula="4 cycles x SQ_INSTS_VALU / (128 SIMDs x GRBM_GUI_ACTIVE summed over the 8 XCDs)")
    for k in ("wave_time_valu", "wave_time_lds", "wave_time_waitcnt", "wave_time_issue_stall", "clock_ghz", "hbm_bytes"):
        if k in pm:
            out[k] = pm[k]
    return out


def run_threads(fn, n):
    """fn(thread_index) on n Python threads (the oracle's ctypes calls release the GIL); returns the results"""
    res = [None] * n
    err = []

    def wrap(i):
        try:
            res[i] = fn(i)
        except Exception as e:   # pragma: no cover
            err.append(e)
    ts = [threading.Thread(target=wrap, args=(i,)) for i in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if err:
        raise err[0]
    return res


def cpu_record(units_1t, dt_1t, phases, units_nt, dt_nt, nthreads, sample, unit):
    model, cores = host_info()
    rec = dict(value=units_1t / dt_1t / 1e6, unit=unit, cores=1, kind="port", sample=sample, cpu_model=model, nproc=cores)
    if phases:
        rec["phases_s"] = phases
    if units_nt:
        rec["n_thread"] = dict(value=units_nt / dt_nt / 1e6, unit=unit, cores=nthreads)
    return rec


def device_frames(ids, W, H, dev, seed0=20260927, scene_len=4, nbase=9):
    """Synthetic BGR frames generated ON THE DEVICE ("generated on-device to avoid PCIe/host bias"): the recipe of
    synth.make_frames_varied -- smooth noise fields drifting like a camera pan, per-frame fine noise, 0..10 high-contrast blobs moving
    through short scenes whose busy-ness differs -- restated with torch ops (test data, not product code).  Frame `i` depends on
    (seed0, i) only, so the content of image i is the same whatever the number of ranks.  Returns a list of H x W x 3 uint8 tensors."""
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(seed0)
    bases = []
    for b in range(nbase):   # smooth base fields: Gaussian-blurred uniform noise (sigma 4 / 8 / 14), rescaled to 0..1
        sigma = (4.0, 8.0, 14.0)[b % 3]
        r = int(3 * sigma + 0.5)
        k = np.exp(-0.5 * (np.arange(-r, r + 1) / sigma) ** 2)
        k = (k / k.sum()).astype(np.float32)
        x = torch.rand((1, 1, H, W), generator=g, device=dev)
        # separable blur as 2 r + 1 shifted multiply-adds per axis (plain elementwise kernels: no convolution library, nothing to
        # compile or tune when eight ranks start at once)
        p_ = torch.nn.functional.pad(x, (0, 0, r, r), mode="reflect")[0, 0]
        x = sum(float(k[i]) * p_[i:i + H] for i in range(2 * r + 1))
        p_ = torch.nn.functional.pad(x[None, None], (r, r, 0, 0), mode="reflect")[0, 0]
        x = sum(float(k[i]) * p_[:, i:i + W] for i in range(2 * r + 1))
        bases.append((x - x.min()) / (x.max() - x.min()).clamp_min(1e-12))
    yy, xx = torch.meshgrid(torch.arange(160, device=dev, dtype=torch.float32), torch.arange(160, device=dev, dtype=torch.float32), indexing="ij")
    out, scenes = [], {}
    for i in ids:
        sc, t = int(i) // scene_len, int(i) % scene_len
        if sc not in scenes:   # the scene's parameters: host RNG seeded by (seed0, scene)
            rng = np.random.default_rng([seed0, sc])
            busy = float(rng.random())
            scenes = {sc: dict(fine=0.05 + 0.35 * rng.random(), gain=0.5 + 0.5 * rng.random(), idx=rng.integers(0, nbase, 3), vel=rng.integers(-3, 4, (3, 2)),
                               blobs=[dict(s=int(rng.integers(48, 161)), y=float(rng.uniform(0, H - 161)), x=float(rng.uniform(0, W - 161)), vy=float(rng.uniform(-2, 2)),
                                           vx=float(rng.uniform(-2, 2)), fy=float(rng.uniform(4, 12)), fx=float(rng.uniform(4, 12))) for _ in range(int(round(busy * 10)))])}
        p_ = scenes[sc]
        g.manual_seed(seed0 * 1000003 + int(i))
        noise = torch.rand((H, W, 3), generator=g, device=dev)
        img = torch.empty((H, W, 3), device=dev)
        for c in range(3):
            base = torch.roll(bases[int(p_["idx"][c])], (int(p_["vel"][c, 0]) * t, int(p_["vel"][c, 1]) * t), dims=(0, 1))
            img[..., c] = (1.0 - p_["fine"]) * (0.5 + p_["gain"] * (base - 0.5)) + p_["fine"] * noise[..., c]
        for bl in p_["blobs"]:
            s_ = bl["s"]
            y0 = int(min(max(bl["y"] + bl["vy"] * t, 0), H - s_ - 1))
            x0 = int(min(max(bl["x"] + bl["vx"] * t, 0), W - s_ - 1))
            blob = 0.5 + 0.5 * torch.sin(yy[:s_, :s_] / s_ * bl["fy"]) * torch.cos(xx[:s_, :s_] / s_ * bl["fx"])
            img[y0:y0 + s_, x0:x0 + s_, :] = 0.15 * img[y0:y0 + s_, x0:x0 + s_, :] + 0.85 * blob[..., None]
        out.append(torch.clamp(torch.round(img * 255), 0, 255).to(torch.uint8).contiguous())
    return out


# ---------------------------------------------------------------------------------------------------------------- workloads
class Workload:
    name = ""
    unit = "Mpatches/s"
    dtype = ""
    units_name = "windows"
    records_cap = 1 << 16     # rows of the padded all_gather buffer (multi-GPU)
    gather_every = None       # None: --gather-every
    fixed_steps = None        # a workload that is ONE fixed job (config 5) sets its own number of steps
    scaling = "weak"

    def step(self, i):
        """one step; returns (units, [(image_id, detector_id, detections)])"""
        raise NotImplementedError

    def flush(self):
        return []

    def sync(self):
        pass

    def kernel_probe(self):
        return None

    def cpu_baseline(self):
        return None


# Rejection profiles of the synthetic WVM (the rejection rate per level drives throughput).  "default" is the model of
# the headline: 65 % of the calibration patches pass each filter until 32 are left (the cascade then stops rejecting, after ~13
# filters).  "late": 90 % pass per filter down to ~0.1 % survivors -- the cascade keeps rejecting through ~65 filters and hands stage B
# a large share of the windows.  "group": real thresholds only at the last filter of every level group (every 14th filter).
WVM_PROFILES = {
    "default": dict(ncalib=8000, kw={}),
    "late": dict(ncalib=40000, kw=dict(pass_rate=0.9, min_survivors=40)),
    "group": dict(ncalib=8000, kw=dict(reject_every=14)),
}


def cascade_models(profile="default"):
    """FaceFrontal WVM (280 filters) + RBF-SVM (1024 SV) calibrated on the config-1 frame; identical on every rank"""
    from featuredetection_amd import synth
    gray = synth.bgr2gray_np(synth.make_frame(640, 480, seed=20260927))   # calibration patches (untimed setup; no oracle involved)
    prof = WVM_PROFILES[profile]
    calib = synth.random_patches(gray[::4, ::4].copy(), 20, 20, prof["ncalib"], np.random.default_rng(1))
    wvm_m = synth.make_wvm(7, calib_patches=calib, **prof["kw"])
    eq = synth.histeq64_np(synth.random_patches(gray[::4, ::4].copy(), 20, 20, 1400, np.random.default_rng(2)))
    svm_m = synth.make_svm_u8(3, eq, nsv=1024, calib=eq[1024:])
    return wvm_m, svm_m


class Cascade(Workload):
    """BASELINE headline: FaceFrontal five-stage cascade, WxH frames (default 640x480)"""
    name = "cascade"
    dtype = "u8/i32/f32/f64"
    records_cap = 1 << 17     # ~3.4 detections per frame, 4096 frames per step, a gather every 2 steps
    gather_every = 2

    profile = "default"

    content = "varied"

    def __init__(self, env, W=640, H=480, frames_per_step=4096, nb=64, multi=True):   # ~50 ms per step: 20 steps are a second of GPU work
        import torch
        from featuredetection_amd import capi, synth
        self.env, self.capi, self.W, self.H = env, capi, W, H
        ctx = env.ctx
        self.NB = max(1, min(nb, frames_per_step))
        self.FP = max(self.NB, frames_per_step // self.NB * self.NB)
        if self.content == "varied":
            # 256 distinct frames in 8 'scenes' of 32 (synth.make_frames_varied): the share of windows that survive the first cascade
            # levels differs ~7x between scenes, and the scenes are reshuffled every pass, so consecutive calls queue different
            # numbers of windows for stage B (its launch plan follows the previous run) and no call repeats its predecessor
            self.NFR, self.SCENE = 256, 32
            frames, self.busy = synth.make_frames_varied(self.NFR, W, H, seed=20260927 + 1000 * env.rank, scene_len=self.SCENE)
        else:
            # round 1-3 content: 8 frames of one recipe, every call holds the same 8 frames x NB / 8
            self.NFR, self.SCENE = 8, 8
            frames = [synth.make_frame(W, H, seed=20260927 + 1000 * env.rank + i) for i in range(self.NFR)]
        self.order_rng = np.random.default_rng(77 + env.rank)
        self.order = np.arange(self.NFR)
        self.qlens = []
        self.frames = frames
        self.dframes = [torch.from_numpy(f).to(env.dev) for f in frames]
        self.dptrs = [t.data_ptr() for t in self.dframes]
        self.nframes_fed = 0
        self.wvm_m, self.svm_m = cascade_models(self.profile)
        kw = dict(inc=float(np.float32(0.92)), min_scale=float(np.float32(0.05)), max_scale=float(np.float32(0.16)))
        self.multi = multi and self.NB > 1
        npyr = 1 if self.multi else self.NB
        self.pyrs = [capi.Pyramid(ctx, **kw) for _ in range(npyr)]
        self.wvms = [capi.Wvm(ctx, self.wvm_m) for _ in range(npyr)]   # one handle (scratch + read-back buffers) per frame in flight
        self.svm = capi.Svm(ctx, self.svm_m)
        self.slots, self.flying, self.ncalls = [], [], 0
        if self.multi:
            # eleven calls in flight, each on its own context (stream), multi-frame pyramid and handles: the library runs the host
            # stages of a call (overlap elimination, SVM launch, NMS) on its queue threads behind the cascade kernels, so this thread
            # only queues kernels and collects finished calls
            # (eleven since the gather points stopped draining the pipeline: 3550-3670 Mpatches/s against 3420-3510 with six; on the
            # runtime's four hardware queues the counts 7, 11, 15 beat their neighbours by 2-4 %)
            for k in range(max(1, int(os.environ.get("FD_BENCH_SLOTS", "11")))):
                c_ = ctx if k == 0 else capi.Context(env.local_rank)
                mp = capi.Pyramid(c_, **kw)
                mp.set_frames(self.NB)
                self.slots.append(dict(ctx=c_, pyr=mp, wvm=capi.Wvm(c_, self.wvm_m), svm=capi.Svm(c_, self.svm_m), run=None, ids=None))
        self.pyrs[0].update(frames[0])
        self.nwin = self.pyrs[0].window_count(20, 20, 1, 1)
        self.layer_bytes = sum(l["w"] * l["h"] for l in self.pyrs[0].layers())
        self.nlayers = len(self.pyrs[0].layers())
        self.metric = "Mpatches/s (extract+WVM+SVM) per GPU, %dx%d pyramid" % (W, H)
        self.config = dict(workload="config 1/metric config: ffpDetectApp FaceFrontal.cfg five-stage cascade on %dx%d BGR frames: %d-layer pyramid, 20x20 "
                                    "windows step 1 (%d windows/frame), HistEq64 -> WVM 280 filters -> OE -> RBF-SVM 1024 SV -> NMS; "
                                    "%s, detections delivered per frame" % (W, H, self.nlayers, self.nwin, "fd_pyramid_update_frames + "
                                    "fd_detect_five_stage_frames (the frames of a call share one pyramid arena, one cascade run and one SVM launch)"
                                    if self.multi else "fd_detect_five_stage_batch"),
                           frames_per_step=self.FP, frames_per_call=self.NB, calls_in_flight=max(1, len(self.slots)), parallelism="image-shard dp%d" % env.world,
                           content=("%d distinct frames resident in HBM, %d scenes of %d with different busy-ness, scene order reshuffled every pass"
                                    % (self.NFR, self.NFR // self.SCENE, self.SCENE)) if self.content == "varied" else
                                   "8 distinct frames, every call holds the same frames (the content of rounds 1-3)")
        if self.profile != "default":
            self.config["wvm_rejection_profile"] = "%s: %s" % (self.profile, WVM_PROFILES[self.profile]["kw"])

    def _frame_ids(self, n):
        """the next n frames of the endless frame sequence: whole scenes, in an order reshuffled at the start of every pass"""
        ids = np.empty(n, np.int64)
        done = 0
        while done < n:
            pos = self.nframes_fed % self.NFR
            if pos == 0 and self.content == "varied":
                sc = self.order_rng.permutation(self.NFR // self.SCENE)
                self.order = (sc[:, None] * self.SCENE + np.arange(self.SCENE)[None, :]).ravel()
            m = min(n - done, self.NFR - pos)
            ids[done:done + m] = self.order[pos:pos + m]
            done += m
            self.nframes_fed += m
        return ids

    def step(self, i):
        capi, NB = self.capi, self.NB
        out = []
        base = i * self.FP
        for c in range(self.FP // NB):
            if self.multi:
                # the NB frames of a call in ONE pyramid: one launch per pyramid stage, one cascade run and one SVM launch
                sl = self.slots[self.ncalls % len(self.slots)]
                self.ncalls += 1
                if sl["run"] is not None:
                    out.extend(self._collect(sl))
                ptrs = [self.dptrs[f] for f in self._frame_ids(NB)]
                sl["pyr"].update_frames(device_ptrs=ptrs, w=self.W, h=self.H, ch=3)
                sl["run"] = capi.FiveStageFrames(sl["ctx"], sl["pyr"], sl["wvm"], sl["svm"], NB)
                sl["ids"] = (base + c * NB + np.arange(NB)) * self.env.world + self.env.rank
                continue
            else:
                fr = [(self.dframes[(base + c * NB + j) % self.NFR].data_ptr(), self.W, self.H, 3) for j in range(NB)]
                res = capi.detect_five_stage_batch(self.env.ctx, [(self.pyrs[j], self.wvms[j], self.svm) for j in range(NB)], device_frames=fr)
            for j, (d_, _) in enumerate(res):
                out.append(((base + c * NB + j) * self.env.world + self.env.rank, 0, d_))
        return self.nwin * self.FP, out

    def _collect(self, sl):
        # one record block per call: (image id of every detection, detector 0, the detections of the call's frames in frame order)
        (dets, fidx, _), sl["run"] = sl["run"].end_flat(), None
        self.qlens.append(sl["wvm"].last_queue_length())
        return [(sl["ids"][fidx], 0, dets)]

    def extra_record(self):
        """spread of the stage-B queue lengths (windows per call that survive the dense pre-filter) over the calls of the run"""
        q = np.array([v for v in self.qlens if v >= 0], np.float64)
        if not len(q):
            return {}
        return dict(stage_b_queue_per_call=dict(calls=int(len(q)), min=int(q.min()), p10=float(np.percentile(q, 10)), median=float(np.median(q)),
                                                p90=float(np.percentile(q, 90)), max=int(q.max()), mean=float(q.mean()),
                                                mean_abs_change_between_consecutive_calls=float(np.abs(np.diff(q)).mean()) if len(q) > 1 else 0.0))

    def flush(self):
        out = []
        for sl in self.slots:
            if sl["run"] is not None:
                out.extend(self._collect(sl))
        return out

    def sync(self):
        for sl in self.slots[1:]:
            sl["ctx"].synchronize()

    def kernel_probe(self):
        """hipEvent-timed duration of the dominant kernel (the dense pre-filter over all windows of a call) and of all cascade kernels of
        a call, the roofline entries, and the single-frame latency of the detector"""
        capi, ctx = self.capi, self.env.ctx
        nf = self.NB if self.multi else 1
        if self.profile != "default" and self.multi:
            return self.kernel_probe_stage_b()
        times = {}
        for mode in (2, 1):   # 2: k_wvm_prefilter alone; 1: every WVM kernel of the call (pre-filter + stage B)
            ctx.set_kernel_timing(mode)
            ms = []
            for i in range(12):
                if self.multi:   # the production launch: one cascade run over the NB frames of a call
                    sl = self.slots[0]
                    sl["pyr"].update_frames(device_ptrs=[self.dptrs[(i * nf + j) % self.NFR] for j in range(nf)], w=self.W, h=self.H, ch=3)
                    capi.detect_five_stage_frames(ctx, sl["pyr"], sl["wvm"], sl["svm"], nf)
                else:
                    self.pyrs[0].update_device(self.dframes[i % self.NFR].data_ptr(), self.W, self.H, 3)
                    capi.detect_five_stage(ctx, self.pyrs[0], self.wvms[0], self.svm)
                ms.append(ctx.last_kernel_ms()[1])
            times[mode] = float(np.mean(ms[2:]))
        ctx.set_kernel_timing(False)
        kms = times[2]
        bytes_per_launch = nf * (self.layer_bytes + self.nwin * 16)
        ach = bytes_per_launch / (kms * 1e-3) / 1e9
        pm = pmc_record("cascade" if (self.W, self.H) == (640, 480) else "cascade_%dx%d" % (self.W, self.H), "k_wv")
        pmk = pmc_record("cascade" if (self.W, self.H) == (640, 480) else "cascade_%dx%d" % (self.W, self.H), "k_wvm_prefilter")
        roof = dict(bound="hbm", kernel="k_wvm_prefilter<20, 20> (HistEq64 + the first cascade levels of every window of the %d frames of a call; the "
                    "dominant kernel)" % nf, achieved=ach, peak=PEAK_HBM_GBS, unit="GB/s", frac=ach / PEAK_HBM_GBS,
                    traffic=pmk.get("hbm_bytes") if pmk else None, kernel_ms=kms, cascade_kernels_ms=times[1],
                    algorithmic="%d frames x (%d layer bytes + 16 B record x %d windows) per launch (SURVEY 8(d))" % (nf, self.layer_bytes, self.nwin))
        extra = {}
        # the issue roofline of the DOMINANT kernel (the one `roofline` names); the figure over all cascade kernels of a call beside it
        if pmk and pmk.get("valu_issue_frac"):
            extra["roofline_issue"] = issue_roofline(pmk)
            if pm and pm.get("valu_issue_frac"):
                extra["roofline_issue"]["all_cascade_kernels"] = dict(kernel=pm.get("kernel"), frac=float(pm["valu_issue_frac"]))
        elif pm and pm.get("valu_issue_frac"):
            extra["roofline_issue"] = issue_roofline(pm)
        if (self.W, self.H) == (640, 480) and self.profile == "default":
            extra["latency_us_single_frame"] = self.single_frame_latency()
        return roof, extra

    def kernel_probe_stage_b(self):
        """The rejection-profile variants hand stage B a large share of the windows: their dominant kernel is k_wvb_chain2 (the rect sums
        of every level as an int8 MFMA contraction + the reference's fp64 chain and exp per level), launched once per stage-B phase.
        HIP events around each of its launches (fd_ctx_set_kernel_timing(3)), summed per call; algorithmic work from the phase plan of
        the same calls: windows alive at a phase's start x its levels x (grey values - 1) x patch pixels x 2 int8 ops."""
        capi, ctx = self.capi, self.env.ctx
        nf = self.NB
        sl = self.slots[0]
        d = int(self.wvm_m["filter_w"]) * int(self.wvm_m["filter_h"])
        nper, nused = int(self.wvm_m["num_per_level"]), int(self.wvm_m["num_used"])
        vo = np.asarray(self.wvm_m["val_off"])
        gv = float(np.mean(np.diff(vo[:nused + 1]) - 1))
        times, ops, plans = {}, [], []
        for mode in (3, 1):
            ctx.set_kernel_timing(mode)
            ms = []
            for i in range(10):
                sl["pyr"].update_frames(device_ptrs=[self.dptrs[(i * nf + j) % self.NFR] for j in range(nf)], w=self.W, h=self.H, ch=3)
                capi.detect_five_stage_frames(ctx, sl["pyr"], sl["wvm"], sl["svm"], nf)
                ms.append(ctx.last_kernel_ms()[1])
                if mode == 3 and i >= 2:
                    plan = sl["wvm"].last_stage_b_plan()
                    plans.append(plan)
                    ops.append(sum(max(a, 0) * (min(g1 * nper, nused) - min(g0 * nper, nused)) * gv * d * 2.0 for g0, g1, a in plan))
            times[mode] = float(np.mean(ms[2:]))
        ctx.set_kernel_timing(False)
        kms = times[3]
        ach = float(np.mean(ops)) / (kms * 1e-3) / 1e12
        pmk = pmc_record(self.name, "k_wvb_chain2")
        roof = dict(bound="mfma", kernel="k_wvb_chain2 (stage B: rect sums of every level of the queued windows as an int8 MFMA contraction + the reference's "
                    "fp64 chain and exp per level; one launch per stage-B phase, %d per call here) -- the dominant kernel of this rejection profile" % len(plans[-1]),
                    achieved=ach, peak=PEAK_I8_MFMA_TOPS, unit="TOP/s (int8)", frac=ach / PEAK_I8_MFMA_TOPS, traffic=pmk.get("hbm_bytes") if pmk else None,
                    kernel_ms=kms, cascade_kernels_ms=times[1], stage_b_phases=[dict(generations=[g0, g1], windows=a) for g0, g1, a in plans[-1]],
                    algorithmic="sum over the stage-B phases of a call: windows alive x levels of the phase x %.1f grey values x %d pixels x 2 ops (SURVEY 8(a) a8: "
                                "S_v = rect sums of grey level v); the contraction is exact int8, the bound that matters for this kernel is its fp64 VALU chain "
                                "(roofline_issue)" % (gv, d))
        extra = {}
        if pmk and pmk.get("valu_issue_frac"):
            extra["roofline_issue"] = issue_roofline(pmk)
        return roof, extra

    def single_frame_latency(self, n=400):
        """one frame resident in HBM -> its detections on the host, one frame at a time: Detector::detect(image) of the reference =
        fd_detect_five_stage_image (pyramid update + five-stage cascade in one call)"""
        capi, ctx = self.capi, self.env.ctx
        det = capi.FiveStageImage(ctx, self.pyrs[0], self.wvms[0], self.svm)
        lat = []
        for i in range(n + 40):
            t0 = time.perf_counter()
            det.detect_device(self.dptrs[i % self.NFR], self.W, self.H, 3)
            lat.append((time.perf_counter() - t0) * 1e6)
        lat = np.array(lat[40:])
        # rounds 1-3 measured the same work as two calls through the binding (fd_pyramid_update, then fd_detect_five_stage)
        pyr, wvm = self.pyrs[0], self.wvms[0]
        lat2 = []
        for i in range(n // 2 + 40):
            t0 = time.perf_counter()
            pyr.update_device(self.dptrs[i % self.NFR], self.W, self.H, 3)
            capi.detect_five_stage(ctx, pyr, wvm, self.svm)
            lat2.append((time.perf_counter() - t0) * 1e6)
        lat2 = np.array(lat2[40:])
        return dict(p50=float(np.percentile(lat, 50)), p99=float(np.percentile(lat, 99)), frames=n,
                    what="blocking per frame through the Python ctypes binding: fd_detect_five_stage_image (pyramid update + five-stage cascade in "
                         "one call, the reference's Detector::detect(image)), detections delivered",
                    two_calls=dict(p50=float(np.percentile(lat2, 50)), p99=float(np.percentile(lat2, 99)),
                                   what="fd_pyramid_update + fd_detect_five_stage as two calls (the figure of rounds 1-3)"))

    def cpu_baseline(self):
        from oracle import pyoracle as O
        kw = dict(inc=float(np.float32(0.92)), min_scale=float(np.float32(0.05)), max_scale=float(np.float32(0.16)))
        wvm_m, svm_m, frames, nwin = self.wvm_m, self.svm_m, self.frames, self.nwin
        if (self.W, self.H) != (640, 480) or self.profile != "default":
            return None

        def worker(budget, split):
            def run(t):
                p, w, s = O.Pyramid(**kw), O.Wvm(wvm_m), O.Svm(svm_m)
                if split:
                    O.phase_timing(True)
                t0 = time.perf_counter()
                n, tu = 0, 0.0
                while time.perf_counter() - t0 < budget:
                    tu0 = time.perf_counter()
                    p.update(frames[(t + n) % len(frames)])
                    tu += time.perf_counter() - tu0
                    O.five_stage(p, w, s)
                    n += 1
                dt = time.perf_counter() - t0
                ph = O.phase_times() if split else (0, 0)
                if split:
                    O.phase_timing(False)
                return n, dt, tu, ph
            return run
        n1, dt1, tu, (te, tc) = worker(8.0, True)(0)
        _, cores = host_info()
        nt = min(cores, 64)
        rs = run_threads(worker(8.0, False), nt)
        un, dtn = sum(r[0] for r in rs) * nwin, max(r[1] for r in rs)
        return cpu_record(n1 * nwin, dt1, dict(update=tu, extract=te, classify=tc, total=dt1), un, dtn, nt,
                          "%d x 640x480 FaceFrontal five-stage frames (16,185 windows each) in %.1f s, oracle -O2, 1 thread; "
                          "n_thread: one frame stream per thread" % (n1, dt1), "Mpatches/s")


class CascadeLate(Cascade):
    """the headline workload with a WVM that keeps rejecting deep into the cascade (VERDICT r02 task 3, profile ii)"""
    name = "cascade_late"
    profile = "late"

    def __init__(self, env, W=640, H=480, frames_per_step=1024, nb=64, multi=True):
        super().__init__(env, W, H, frames_per_step, nb, multi)


class Cascade8(Cascade):
    """the headline workload on the content of rounds 1-3 (8 frames, every call identical): kept as a second record for continuity"""
    name = "cascade_8frames"
    content = "8frames"

    def __init__(self, env, W=640, H=480, frames_per_step=2048, nb=64, multi=True):
        super().__init__(env, W, H, frames_per_step, nb, multi)


class CascadeGroup(Cascade):
    """... with a WVM that rejects only at the end of every level group (profile iii)"""
    name = "cascade_group"
    profile = "group"

    def __init__(self, env, W=640, H=480, frames_per_step=1024, nb=64, multi=True):
        super().__init__(env, W, H, frames_per_step, nb, multi)


class HogSvm(Workload):
    """config 2: 640x480, ImagePyramid(octl=5, 1/16..1), 20x20 windows stride 2, HOG-324 + RBF SVM (1024 SV)"""
    name = "hog_svm"
    dtype = "f32"
    records_cap = 1 << 17     # a single-stage detector returns every positive window (~2.8 k per frame here)
    gather_every = 1

    def __init__(self, env, W=640, H=480, frames_per_step=32, inflight=2):
        import torch
        from featuredetection_amd import capi, synth
        # three frames in flight (FD_BENCH_HOG_INFLIGHT): with two, a pair of streams that the runtime maps onto the same one of its four
        # hardware queues serialises the frames (160 instead of 166-168 Mpatches/s when the contexts of other workloads exist already)
        inflight = max(1, int(os.environ.get("FD_BENCH_HOG_INFLIGHT", "3")))
        self.env, self.capi, self.W, self.H = env, capi, W, H
        self.FP = max(1, frames_per_step)
        self.NFR = 4
        self.frames = [synth.make_frame(W, H, seed=20260927 + 1000 * env.rank + i) for i in range(self.NFR)]
        self.dframes = [torch.from_numpy(f).to(env.dev) for f in self.frames]
        ctx = env.ctx
        mk = lambda c: capi.Pyramid(c, octave_layers=5, min_scale=1 / 16, max_scale=1.0)
        pyr = mk(ctx)
        pyr.set_layer_filter(capi.FD_LAYER_GRADBIN, bins=9)
        self.hp = capi.hog_params(20, 20, 2, 2, 9, 5, 2, False)
        pyr.update(synth.make_frame(W, H, seed=4242))   # model: SVs drawn from the HOG features of a second seeded frame
        feats2 = capi.extract_hog(ctx, pyr, self.hp)
        self.model = synth.make_svm_f32(20260927, feats2, nsv=1024, gamma=0.5, positive_fraction=0.01)
        del feats2
        # frames in flight: one context (stream + scratch), pyramid and model copy each
        self.slots = [(ctx, pyr, capi.Svm(ctx, self.model))]
        for _ in range(1, max(1, inflight)):
            c2 = capi.Context(env.local_rank)
            p2 = mk(c2)
            p2.set_layer_filter(capi.FD_LAYER_GRADBIN, bins=9)
            self.slots.append((c2, p2, capi.Svm(c2, self.model)))
        pyr.update(self.frames[0])
        self.nwin = pyr.window_count(20, 20, 2, 2)
        self.flying = []
        self.metric = "Mpatches/s (extract+HOG+RBF-SVM), %dx%d pyramid" % (W, H)
        self.config = dict(workload="config 2: %dx%d BGR frame, ImagePyramid(octl=5, 1/16..1) %d layers, 20x20 windows stride 2 (%d windows/frame), "
                                    "GradientFilter+GradientBinning(9) layers, HogFilter(9,cell 5,block 2)=324 f32, RBF-SVM 1024 SV gamma 0.5; "
                                    "fd_detect_hog_svm_begin/_end, detections delivered per frame" % (W, H, len(pyr.layers()), self.nwin),
                           frames_per_step=self.FP, frames_in_flight=len(self.slots), parallelism="image-shard dp%d" % env.world)

    def _collect(self):
        img, run = self.flying.pop(0)
        return (img, 0, run.end())

    def step(self, i):
        out = []
        for j in range(self.FP):
            f = i * self.FP + j
            c_, p_, s_ = self.slots[f % len(self.slots)]
            if len(self.flying) >= len(self.slots):
                out.append(self._collect())
            p_.update_device(self.dframes[f % self.NFR].data_ptr(), self.W, self.H, 3)
            self.flying.append((f * self.env.world + self.env.rank, self.capi.HogSvmRun(c_, p_, s_, self.hp)))
        return self.nwin * self.FP, out

    def flush(self):
        out = []
        while self.flying:
            out.append(self._collect())
        return out

    def sync(self):
        for c_, _, _ in self.slots[1:]:
            c_.synchronize()

    def kernel_probe(self):
        capi = self.capi
        c_, p_, s_ = self.slots[0]
        c_.set_kernel_timing(True)
        ms = []
        for i in range(10):
            p_.update_device(self.dframes[i % self.NFR].data_ptr(), self.W, self.H, 3)
            capi.detect_hog_svm(c_, p_, s_, self.hp, want_all=False, cap=1 << 16)
            ms.append(c_.last_kernel_ms()[1])
        c_.set_kernel_timing(False)
        kms = float(np.mean(ms[2:]))
        flops = 2.0 * 324 * 1024 * self.nwin
        ach = flops / (kms * 1e-3) / 1e12
        kname = c_.last_kernel_ms()[0] or "k_hog_svm_fused"
        pm = pmc_record("hog_svm", kname) if (self.W, self.H) == (640, 480) else None
        return dict(bound="mfma", kernel=kname + (" (HOG vectors produced in the registers of the MFMA operand + RBF SVM: one kernel)" if "fused" in kname else ""), achieved=ach, peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s", frac=ach / PEAK_F32_MFMA_TFLOPS,
                    traffic=pm.get("hbm_bytes") if pm else None, traffic_source=pm.get("source") if pm else None, kernel_ms=kms,
                    algorithmic="2*324*1024 flop/window x %d windows/launch" % self.nwin), {}

    def cpu_baseline(self):
        """the same inputs as the GPU leg -- one full 640x480 frame of the workload.  The whole
        frame is 35 s of one core (278 K windows x 1024 support vectors x 324), so the 1-thread leg builds the full pyramid and evaluates
        every SAMPLE-th window of it (a window's HOG + SVM cost does not depend on where it lies); the n-thread leg evaluates the WHOLE
        frame, thread t taking windows t, t + T, ... (each thread builds the pyramid itself)."""
        from oracle import pyoracle as O
        model, frame, hp = self.model, self.frames[0], (20, 20, 2, 2, 9, 5, 2)
        SAMPLE = 3

        def run(first, step, split=False):
            p = O.Pyramid(octave_layers=5, min_scale=1 / 16, max_scale=1.0)
            p.set_layer_filter(1, bins=9)
            s = O.Svm(model)
            if split:
                O.phase_timing(True)
            t0 = time.perf_counter()
            p.update(frame)
            tu = time.perf_counter() - t0
            nvis, _ = O.sliding_hog_svm_sample(p, s, *hp, first, step)
            dt = time.perf_counter() - t0
            ph = O.phase_times() if split else (0, 0)
            if split:
                O.phase_timing(False)
            return nvis, dt, tu, ph
        n1, dt1, tu, (te, tc) = run(0, SAMPLE, True)
        _, cores = host_info()
        nt = min(cores, 64)
        rs = run_threads(lambda t: run(t, nt), nt)
        return cpu_record(n1, dt1, dict(update=tu, extract=te, classify=tc, total=dt1), sum(r[0] for r in rs), max(r[1] for r in rs), nt,
                          "one full %dx%d frame of the workload: its pyramid + every %d-rd of its %d windows (%d windows, %.1f s), oracle -O2, 1 thread; "
                          "n_thread: the whole frame, windows dealt over the threads" % (self.W, self.H, SAMPLE, self.nwin, n1, dt1), "Mpatches/s")


def ffp15_models(nsv=1024):
    """the 15 detectors of ffpDetectApp/*.cfg: (name, pyramid key, WVM, SVM, pw, ph); SURVEY 8(d) config 3: 1024 SVs each"""
    from featuredetection_amd import synth
    gray = synth.bgr2gray_np(synth.make_frame(640, 480, seed=20260927))   # calibration patches (untimed setup; no oracle involved)
    models = []
    for di, (name, (inc, mn, mx, pw, ph, nper, nlev)) in enumerate(sorted(synth.DETECTOR_CFGS.items())):
        src = gray[::4, ::4] if mx < 0.3 else gray[::2, ::2]
        calib = synth.random_patches(src.copy(), pw, ph, 6000, np.random.default_rng(100 + di))
        wm = synth.make_wvm(50 + di, fw=pw, fh=ph, n_per=nper, n_levels=nlev, calib_patches=calib, min_survivors=24)
        eq = synth.histeq64_np(synth.random_patches(src.copy(), pw, ph, nsv + 200, np.random.default_rng(200 + di)))
        sm = synth.make_svm_u8(300 + di, eq, nsv=nsv, calib=eq[nsv:])
        models.append((name, (inc, mn, mx), wm, sm, pw, ph))
    return models


class Ffp15(Workload):
    """config 3 (and the per-GPU work of config 5): 15 five-stage detectors on 1920x1080 frames, 32.1 M windows per frame"""
    name = "ffp15"
    dtype = "u8/i32/f32/f64"
    records_cap = 1 << 18     # ~1,200 detections per frame over the 15 detectors, 32 frames of a rank between two gathers
    content = "varied"        # "2frames": the content of rounds 1-4 (two alternating frames of synth.make_frame)
    NDISTINCT = 32

    def make_content(self, W, H):
        """the frames this rank works on, resident in HBM: (list of device tensors, description)"""
        import torch
        from featuredetection_amd import synth
        env = self.env
        if self.content == "2frames":
            fr = [torch.from_numpy(synth.make_frame(W, H, seed=20260927 + 1000 * env.rank + i)).to(env.dev) for i in range(2)]
            return fr, "2 alternating frames (the content of rounds 1-4: stage B's launch plan, OE and the SVM see the same positives every other frame)"
        # VERDICT r04 task 1a: >= 32 distinct frames per rank (199 MB of BGR in HBM), 8 scenes of 4 frames with different busy-ness,
        # visited in an order reshuffled every pass -- so consecutive frames queue different numbers of windows and nothing repeats
        ids = np.arange(self.NDISTINCT) + 100000 * (1 + env.rank)
        return device_frames(ids, W, H, env.dev), ("%d distinct frames resident in HBM (device_frames: %d scenes of 4 with different busy-ness), order reshuffled every pass"
                                                   % (self.NDISTINCT, self.NDISTINCT // 4))

    def next_frame(self):
        """device tensor of the next frame of the endless sequence"""
        n = len(self.dframes)
        pos = self.nfed % n
        if pos == 0 and self.content == "varied":
            sc = self.order_rng.permutation(n // 4)
            self.order = (sc[:, None] * 4 + np.arange(4)[None, :]).ravel()
        self.nfed += 1
        return self.dframes[int(self.order[pos])]

    def __init__(self, env, W=1920, H=1080, frames_per_step=8):   # 20 steps = 160 frames = five whole passes over the 32 distinct frames
        import torch  # noqa: F401
        from featuredetection_amd import capi, synth  # noqa: F401
        self.env, self.capi, self.W, self.H = env, capi, W, H
        self.FP = max(1, frames_per_step)
        self.dframes, content_note = self.make_content(W, H)
        self.order, self.nfed, self.order_rng = np.arange(len(self.dframes)), 0, np.random.default_rng(78 + env.rank)
        self.stage_sum, self.stage_frames = np.zeros(4, np.int64), 0
        self.models = ffp15_models()
        ctx = env.ctx
        # four frames in flight (FD_BENCH_FFP_SLOTS; with the library's batch queue and eight batch streams: 9600 against 9280 Mpatches/s with
        # three, 9430 with five), each with its own pyramids and classifier handles: the next frames' pyramids and cascades are queued while
        # the library's queue threads run frame f's host stages (ordering, overlap elimination, SVM launches, NMS)
        nslots = max(1, int(os.environ.get("FD_BENCH_FFP_SLOTS", "4")))
        self.slots = []
        for k in range(nslots):
            pyrs, dets = {}, []
            for name, key, wm, sm, pw, ph in self.models:
                if key not in pyrs:   # detectors with identical pyramid parameters share one pyramid (identical layers)
                    pyrs[key] = capi.Pyramid(ctx, inc=float(np.float32(key[0])), min_scale=float(np.float32(key[1])), max_scale=float(np.float32(key[2])))
                dets.append((name, pyrs[key], capi.Wvm(ctx, wm), capi.Svm(ctx, sm), pw, ph))
            self.slots.append(dict(pyrs=pyrs, dets=dets, run=None, img=None))
        self.pyrs, self.dets = self.slots[0]["pyrs"], self.slots[0]["dets"]
        for pr in self.pyrs.values():
            pr.update_device(self.dframes[0].data_ptr(), W, H, 3)
        self.nwin = sum(pr.window_count(pw, ph, 1, 1) for _, pr, _, _, pw, ph in self.dets)
        self.ncalls = 0
        self.metric = "Mpatches/s (extract+WVM+SVM cascade, 15 detectors), %dx%d pyramid" % (W, H)
        self.config = dict(workload="config 3: the 15 detectors of ffpDetectApp/*.cfg (five-stage WVM -> OE -> RBF-SVM 1024 SV -> NMS each), full %dx%d "
                                    "frame, step 1: %d windows per frame; %d shared pyramids; fd_five_stage_batch_begin/_end, %d frames in flight" %
                                    (W, H, self.nwin, len(self.pyrs), nslots),
                           frames_per_step=self.FP, parallelism="image-shard dp%d" % env.world, content=content_note)

    def _collect(self, sl):
        res, sl["run"] = sl["run"].end(), None
        self.stage_sum += np.sum([np.asarray(st_, np.int64)[:4] for _, st_ in res], axis=0)
        self.stage_frames += 1
        return [(sl["img"], di, d_) for di, (d_, _) in enumerate(res)]

    def extra_record(self):
        """what the host stages of a frame see: WVM positives, survivors of the overlap elimination, SVM positives, detections -- summed
        over the 15 detectors, mean per frame (the host stages scale with the first number)"""
        if not self.stage_frames:
            return {}
        m = self.stage_sum / self.stage_frames
        return dict(stage_counts_per_frame=dict(wvm_positives=float(m[0]), after_overlap_elimination=float(m[1]), svm_positives=float(m[2]), detections=float(m[3]),
                                                frames=int(self.stage_frames)))

    def _feed(self, fr, image_id):
        """queues one frame (device tensor) through the 15 detectors; returns the records of the call that had to be collected first"""
        out = []
        sl = self.slots[self.ncalls % len(self.slots)]
        self.ncalls += 1
        if sl["run"] is not None:
            out.extend(self._collect(sl))
        for pr in sl["pyrs"].values():
            pr.update_device(fr.data_ptr(), self.W, self.H, 3)
        sl["run"] = self.capi.FiveStageBatch(self.env.ctx, [(pr, wv, sv_) for _, pr, wv, sv_, _, _ in sl["dets"]], cap=4096)
        sl["img"] = image_id
        return out

    def step(self, i):
        out = []
        for j in range(self.FP):
            f = i * self.FP + j
            out.extend(self._feed(self.next_frame(), f * self.env.world + self.env.rank))
        return self.nwin * self.FP, out

    def flush(self):
        out = []
        for sl in self.slots:
            if sl["run"] is not None:
                out.extend(self._collect(sl))
        return out

    def kernel_probe(self):
        """the dominant kernel of the batch -- k_wvm_prefilter<24, 24> of one of the seven 24x24 detectors on the 1080p frame, timed
        alone with HIP events -- + the batch's issue roofline from the PMC passes"""
        capi, ctx = self.capi, self.env.ctx
        name, pr, wv, sv_, pw, ph = [d for d in self.dets if (d[4], d[5]) == (24, 24)][0]
        times = {}
        for mode in (2, 1):
            ctx.set_kernel_timing(mode)
            ms = []
            for i in range(6):
                pr.update_device(self.dframes[i % len(self.dframes)].data_ptr(), self.W, self.H, 3)
                capi.detect_five_stage(ctx, pr, wv, sv_, cap=1 << 14)
                ms.append(ctx.last_kernel_ms()[1])
            times[mode] = float(np.mean(ms[1:]))
        # the seven 24x24 detectors share ONE pre-filter launch in the batch (k_wvm_prefilter_group): that launch alone
        grp = [(p_, w_, s_) for _, p_, w_, s_, pw_, ph_ in self.dets if (pw_, ph_) == (24, 24) and p_ is pr]
        ctx.set_kernel_timing(2)
        gms, members = [], 0
        for i in range(5):
            pr.update_device(self.dframes[i % len(self.dframes)].data_ptr(), self.W, self.H, 3)
            capi.detect_five_stage_batch(ctx, grp, cap=1 << 14)
            t_, members = ctx.last_group_prefilter_ms()
            gms.append(t_)
        ctx.set_kernel_timing(False)
        nwin = pr.window_count(pw, ph, 1, 1)
        layer_bytes = sum(l["w"] * l["h"] for l in pr.layers())
        pm = pmc_record("ffp15", "k_wv") if (self.W, self.H) == (1920, 1080) else None
        if members >= 2:
            kms = float(np.mean(gms[1:]))
            ach = (layer_bytes + 16 * nwin * members) / (kms * 1e-3) / 1e9
            pmk = pmc_record("ffp15", "k_wvm_prefilter_group") if (self.W, self.H) == (1920, 1080) else None
            roof = dict(bound="hbm", kernel="k_wvm_prefilter_group<24, 24>: ONE launch for the %d detectors with a 24x24 patch on this pyramid (%d windows each; "
                        "the largest kernel of the batch)" % (members, nwin), achieved=ach, peak=PEAK_HBM_GBS, unit="GB/s",
                        frac=ach / PEAK_HBM_GBS, traffic=pmk.get("hbm_bytes") if pmk else None, kernel_ms=kms, single_detector_kernel_ms=times[2],
                        cascade_kernels_ms=times[1], algorithmic="%d layer bytes + 16 B record x %d windows x %d detectors" % (layer_bytes, nwin, members))
        else:   # FD_WVM_GROUP=0
            kms = times[2]
            ach = (layer_bytes + 16 * nwin) / (kms * 1e-3) / 1e9
            pmk = pmc_record("ffp15", "k_wvm_prefilter") if (self.W, self.H) == (1920, 1080) else None
            roof = dict(bound="hbm", kernel="k_wvm_prefilter<24, 24> of the %s detector (%d windows; the 24x24 pre-filters are the largest share of the "
                        "batch's kernel time)" % (name, nwin), achieved=ach, peak=PEAK_HBM_GBS, unit="GB/s",
                        frac=ach / PEAK_HBM_GBS, traffic=pmk.get("hbm_bytes") if pmk else None, kernel_ms=kms, cascade_kernels_ms=times[1],
                        algorithmic="%d layer bytes + 16 B record x %d windows" % (layer_bytes, nwin))
        extra = {}
        # the issue roofline of the DOMINANT kernel (the one `roofline` names); the figure over all cascade kernels of a call beside it
        if pmk and pmk.get("valu_issue_frac"):
            extra["roofline_issue"] = issue_roofline(pmk)
            if pm and pm.get("valu_issue_frac"):
                extra["roofline_issue"]["all_cascade_kernels"] = dict(kernel=pm.get("kernel"), frac=float(pm["valu_issue_frac"]))
        elif pm and pm.get("valu_issue_frac"):
            extra["roofline_issue"] = issue_roofline(pm)
        return roof, extra

    def cpu_baseline(self):
        """the same inputs as the GPU leg -- one full 1920x1080 frame of the workload.  All 15
        detectors on it are ~60 s of one core, so the 1-thread leg runs every third detector (5 of the 15: 20x20, 24x24 and 32x16
        patches among them) on the full frame; the n-thread leg runs all 15, one detector per thread."""
        from oracle import pyoracle as O
        frame = self.dframes[0].cpu().numpy()
        models = self.models
        W, H = self.W, self.H

        def run_dets(idx, split=False):
            if split:
                O.phase_timing(True)
            t0 = time.perf_counter()
            n, tu = 0, 0.0
            for di in idx:
                name, key, wm, sm, pw, ph = models[di]
                p = O.Pyramid(inc=float(np.float32(key[0])), min_scale=float(np.float32(key[1])), max_scale=float(np.float32(key[2])))
                tu0 = time.perf_counter()
                p.update(frame)
                tu += time.perf_counter() - tu0
                n += len(p.windows(pw, ph, 1, 1))
                O.five_stage(p, O.Wvm(wm), O.Svm(sm), cap=1 << 16)
            dt = time.perf_counter() - t0
            ph_ = O.phase_times() if split else (0, 0)
            if split:
                O.phase_timing(False)
            return n, dt, tu, ph_
        sample = list(range(0, len(models), 3))
        n1, dt1, tu, (te, tc) = run_dets(sample, True)
        _, cores = host_info()
        nt = min(cores, len(models))
        rs = run_threads(lambda t: run_dets(range(t, len(models), nt)), nt)
        return cpu_record(n1, dt1, dict(update=tu, extract=te, classify=tc, total=dt1), sum(r[0] for r in rs), max(r[1] for r in rs), nt,
                          "%d of the 15 detectors (%s) on one full %dx%d frame of the workload (%d windows, %.1f s; the reference rebuilds the pyramid per "
                          "detector), oracle -O2, 1 thread; n_thread: all 15 detectors on the frame, one per thread" %
                          (len(sample), ",".join(models[i][0] for i in sample), W, H, n1, dt1), "Mpatches/s")


class Ffp15Two(Ffp15):
    """config 3 on the content of rounds 1-4 (two alternating frames per rank): kept as a second record for continuity"""
    name = "ffp15_2frames"
    content = "2frames"

    def cpu_baseline(self):
        return None


class Config5(Ffp15):
    """BASELINE config 5: a 10,000-image 1920x1080 batch through the 15 five-stage detectors, image i -> rank i mod N (fd_dist_owner),
    models replicated, the detection records gathered every 256 images of the job.  A FIXED job: one step = one
    gather interval (256 images job-wide, 256 / N per rank), steps = ceil(images / 256) whatever --steps says; value = all windows of
    the job / wall time, `scaling` = "strong".  Every image is distinct and generated on the device from (seed, image index), so the
    job's content -- and its detections -- do not depend on N."""
    name = "config5"
    gather_every = 1
    scaling = "strong"
    GATHER_IMAGES = 256

    def make_content(self, W, H):
        env = self.env
        total = max(1, int(os.environ.get("FD_BENCH_CONFIG5_IMAGES", "10000")))
        self.total_images = total
        self.my_ids = np.arange(env.rank, total, env.world)          # image i belongs to rank i mod N
        t0 = time.perf_counter()
        fr = device_frames(self.my_ids, W, H, env.dev, seed0=20260928)
        import torch
        torch.cuda.synchronize()
        self.gen_s = time.perf_counter() - t0
        self.fixed_steps = (total + self.GATHER_IMAGES - 1) // self.GATHER_IMAGES
        return fr, ("%d distinct %dx%d frames of the %d-image job resident in HBM on this rank (%.1f GB), generated on the device in %.1f s, each from "
                    "(seed, image index)" % (len(fr), W, H, total, len(fr) * W * H * 3 / 1e9, self.gen_s))

    def __init__(self, env, W=1920, H=1080, frames_per_step=0):
        super().__init__(env, W, H, frames_per_step=1)
        self.metric = "Mpatches/s (extract+WVM+SVM cascade, 15 detectors), %d-image %dx%d batch sharded over %d GPU(s)" % (self.total_images, W, H, env.world)
        self.config["workload"] = ("config 5: %d images of %dx%d through the 15 detectors of ffpDetectApp/*.cfg (%d windows per image), image i -> rank i mod %d, "
                                   "models replicated, ONE all-gather of the detection records (fd_dist_gather_records) every %d images; %s" %
                                   (self.total_images, W, H, self.nwin, env.world, self.GATHER_IMAGES, self.config["workload"].split("; ", 1)[-1]))
        self.config["images"] = self.total_images
        self.config["gather_every_images"] = self.GATHER_IMAGES
        # rows of the padded gather buffer per rank (the same on every rank): ~1,200 detections per image over the 15 detectors on this
        # content, 4096 allowed per image of a rank's share of a gather interval; a rank that had more is reported (`records_truncated`)
        per_rank = (min(self.GATHER_IMAGES, self.total_images) + env.world - 1) // env.world
        self.records_cap = 1 << int(np.ceil(np.log2(max(1 << 15, per_rank * 4096))))
        self.config.pop("frames_per_step", None)

    @staticmethod
    def step_images(i, rank, world, total, gather=256):
        """images of step i (the job's images [gather i, gather (i + 1))) that belong to `rank`: image j -> rank j mod world"""
        lo, hi = i * gather, min((i + 1) * gather, total)
        first = lo + ((rank - lo) % world)
        return np.arange(first, hi, world)

    def step(self, i):
        out = []
        ids = self.step_images(i, self.env.rank, self.env.world, self.total_images, self.GATHER_IMAGES)
        for img in ids:
            out.extend(self._feed(self.dframes[(int(img) - self.env.rank) // self.env.world], int(img)))
        return self.nwin * len(ids), out

    def kernel_probe(self):
        return None

    def cpu_baseline(self):
        return None


class Sdm(Workload):
    """config 4: 256 gray 256x256 face crops, 68 landmarks, 4 cascaded regressors on adaptive VlHog descriptors"""
    name = "sdm"
    unit = "M SDM iters/s"
    dtype = "f32/f64"
    units_name = "sdm_iters"

    def __init__(self, env, batches_per_step=32):
        import torch
        from featuredetection_amd import capi, synth
        self.env, self.capi = env, capi
        self.B, self.W, self.H = 256, 256, 256
        self.FP = max(1, batches_per_step)
        # 256 DISTINCT crops per batch and NBATCH distinct batches cycled ("256 seeded crops");
        # generated on the device (device_frames, scenes of 4 crops with different busy-ness), one gray plane per crop
        self.NBATCH = max(1, int(os.environ.get("FD_BENCH_SDM_NBATCH", "4")))   # (1 for counter passes: rocprofv3 --pmc does not survive the generator's ~40 K tiny dispatches)
        ids = np.arange(self.NBATCH * self.B) + 500000 * (1 + env.rank)
        if os.environ.get("FD_BENCH_SDM_HOSTGEN"):   # counter passes: rocprofv3 --pmc segfaults inside the torch generator at this size
            crops = [torch.from_numpy(np.repeat(synth.make_frame(self.W, self.H, seed=int(i), channels=1)[..., None], 3, axis=2)).to(env.dev) for i in ids]
        else:
            crops = device_frames(ids, self.W, self.H, env.dev, seed0=20260929)
        self.dimgs = [torch.stack([c[..., 1] for c in crops[b * self.B:(b + 1) * self.B]]).contiguous() for b in range(self.NBATCH)]
        self.imgs16 = self.dimgs[0][:16].cpu().numpy()
        del crops
        self.nfed = 0
        self.model = synth.make_sdm(9, L=68, S=4)
        self.sdm = capi.Sdm(env.ctx, self.model)
        # fd_sdm_fit_batch_begin / _end: ONE host thread keeps a few batches queued (each ticket has its own scratch set, consecutive
        # tickets alternate between two streams); FD_BENCH_SDM_INFLIGHT sets how many
        self.inflight = max(1, int(os.environ.get("FD_BENCH_SDM_INFLIGHT", "4")))
        self.boxes = np.array([[48, 48, 160, 160]] * self.B, np.int32)
        self.metric = "SDM iters/s (x1e6): 68 landmarks, HOG at each point + linear regressor, 4 cascade steps, batch of 256 face crops"
        self.config = dict(workload="config 4: 256 gray 256x256 crops, 68 landmarks, 4 cascade steps, adaptive VlHog 3x3x31 per landmark + regressor "
                                    "18973x136 (f64 MFMA); fd_sdm_fit_batch_begin/_end, shapes delivered per batch, one host thread, %d batches in flight" % self.inflight,
                           batches_per_step=self.FP, parallelism="face-shard dp%d" % env.world,
                           content="%d distinct batches of %d distinct crops each resident in HBM (device_frames), cycled" % (self.NBATCH, self.B))

    def step(self, i):
        flying = []
        for _ in range(self.FP):
            if len(flying) == self.inflight:
                self.sdm.fit_end(flying.pop(0))
            flying.append(self.sdm.fit_device_begin(self.dimgs[self.nfed % self.NBATCH].data_ptr(), self.W, self.H, self.B, self.boxes))
            self.nfed += 1
        for t in flying:
            self.sdm.fit_end(t)
        return self.B * 4 * self.FP, []

    def kernel_probe(self):
        ctx = self.env.ctx
        ctx.set_kernel_timing(True)
        ms = []
        for i in range(8):
            self.sdm.fit_device(self.dimgs[i % self.NBATCH].data_ptr(), self.W, self.H, self.B, self.boxes)
            ms.append(ctx.last_kernel_ms()[1])
        ctx.set_kernel_timing(False)
        kms = float(np.mean(ms[2:]))
        items = self.B * 68
        # compulsory bytes of a descriptor = its source crop (<= (2*wsh)^2 u8, here ~46x46) + 279 f32 written
        bytes_per_launch = items * (46 * 46 + 4 * 279)
        ach = bytes_per_launch / (kms * 1e-3) / 1e9
        pm = pmc_record("sdm", "k_sdm_descriptors")
        roof = dict(bound="hbm", kernel="k_sdm_descriptors", achieved=ach, peak=PEAK_HBM_GBS, unit="GB/s", frac=ach / PEAK_HBM_GBS,
                    traffic=pm.get("hbm_bytes") if pm else None, kernel_ms=kms,
                    algorithmic="(46x46 B crop + 279 f32) x %d (face, landmark) items per launch" % items)
        extra = {}
        if pm and pm.get("valu_issue_frac"):
            extra["roofline_issue"] = issue_roofline(pm)
        return roof, extra

    def cpu_baseline(self):
        from oracle import pyoracle as O
        imgs, model = self.imgs16, self.model

        def run(t, budget=6.0):
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < budget:
                O.sdm_fit(imgs[(t + n) % 16], model, [48, 48, 160, 160])
                n += 1
            return n * 4, time.perf_counter() - t0
        n1, dt1 = run(0)
        _, cores = host_info()
        nt = min(cores, 64)
        rs = run_threads(run, nt)
        return cpu_record(n1, dt1, None, sum(r[0] for r in rs), max(r[1] for r in rs), nt,
                          "%d faces x 4 cascade steps (68 landmarks) in %.1f s, oracle -O2, 1 thread; n_thread: one face stream per thread" % (n1 // 4, dt1),
                          "M SDM iters/s")


class Rvm(Workload):
    """SURVEY 8(f) row 1 (not a BASELINE config): SlidingWindowDetector + ProbabilisticRvmClassifier"""
    name = "rvm"
    dtype = "u8/f32/f64"

    def __init__(self, env, W=1920, H=1080):
        import torch
        from featuredetection_amd import capi, synth
        self.env, self.capi, self.W, self.H = env, capi, W, H
        self.dframes = [torch.from_numpy(synth.make_frame(W, H, seed=20260927 + 1000 * env.rank + i)).to(env.dev) for i in range(4)]
        gray = synth.bgr2gray_np(synth.make_frame(640, 480, seed=20260927))
        calib = synth.histeq64_np(synth.random_patches(gray[::4, ::4].copy(), 20, 20, 6000, np.random.default_rng(1)))
        feats = calib.reshape(len(calib), -1).astype(np.float32) * np.float32(1.0 / 255.0)
        self.pyr = capi.Pyramid(env.ctx, inc=float(np.float32(0.92)), min_scale=float(np.float32(0.05)), max_scale=float(np.float32(0.16)))
        self.rvm = capi.Rvm(env.ctx, synth.make_rvm(9, feats, 20, 20, n_filters=100, kernel=2, pass_rate=0.5))
        self.pyr.update_device(self.dframes[0].data_ptr(), W, H, 3)
        self.nwin = self.pyr.window_count(20, 20, 1, 1)
        self.metric = "Mpatches/s (extract+HistEq64+RVM cascade), %dx%d pyramid" % (W, H)
        self.config = dict(workload="prvm single detector: FaceFrontal pyramid on a %dx%d frame, 20x20 windows step 1 (%d windows), hq64 + "
                                    "ConversionFilter(CV_32F, 1/255), RBF RVM with 100 reduced set vectors" % (W, H, self.nwin), frames_per_step=1)

    def step(self, i):
        self.pyr.update_device(self.dframes[i % 4].data_ptr(), self.W, self.H, 3)
        d_, _, _ = self.capi.detect_rvm(self.env.ctx, self.pyr, self.rvm, feature_space=self.capi.FEATURE_HQ64, conv_scale=1.0 / 255.0, want_all=False)
        return self.nwin, [(i * self.env.world + self.env.rank, 0, d_)]


class Aggregated(Workload):
    """SURVEY 8(f) row 2 (not a BASELINE config): AggregatedFeaturesDetector (FHOG + linear SVM convolution + IoU NMS)"""
    name = "aggregated"
    unit = "Mwindows/s"
    dtype = "u8/f32"

    def __init__(self, env, W=1920, H=1080):
        import torch
        from featuredetection_amd import capi, synth
        self.env, self.capi, self.W, self.H = env, capi, W, H
        self.dframes = [torch.from_numpy(synth.make_frame(W, H, seed=20260927 + 1000 * env.rank + i)).to(env.dev) for i in range(2)]
        wts = np.random.default_rng(3).normal(0, 0.05, (10, 10, 31)).astype(np.float32)
        self.det = capi.Aggregated(env.ctx, wts, 0.1, 1.5, octave_layers=5, nms_overlap=0.3)
        inc, k, n = 0.5 ** (1 / 5), 0, 0
        while True:
            sc = inc ** k
            lw, lh = int(round(W * sc)), int(round(H * sc))
            if lw // 8 < 10 or lh // 8 < 10:
                break
            n += (lw // 8 - 9) * (lh // 8 - 9)
            k += 1
        self.nwin = n
        self.metric = "Mwindows/s (FHOG pyramid + linear SVM convolution + NMS), %dx%d" % (W, H)
        self.config = dict(workload="AggregatedFeaturesDetector: %dx%d BGR frame, FhogFilter(8, 9 bins), 10x10-cell linear SVM, 5 layers per octave, "
                                    "~%d window positions, IoU NMS 0.3 on the host" % (W, H, n), frames_per_step=1)

    def step(self, i):
        self.det.detect_device(self.dframes[i % 2].data_ptr(), self.W, self.H, 3)
        return self.nwin, []



# ---------------------------------------------------------------------------------------------------------------- output
LINE_LIMIT = 8000   # bytes of the ONE JSON line on stdout (the 22 KB line of round 5 was not parsed by the driver)


def _r(v, nd=6):
    """floats to nd significant digits (the line is for parsing, the side file keeps everything)"""
    if isinstance(v, float):
        return float("%.*g" % (nd, v))
    if isinstance(v, dict):
        return {k: _r(x, nd) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, nd) for x in v]
    return v


def _cut(s_, n):
    s_ = str(s_)
    return s_ if len(s_) <= n else s_[:n - 3] + "..."


def compact_record(rec, text=160):
    """the contract keys of a record + roofline / roofline_issue / cpu_baseline reduced to their numbers and short names"""
    out = {k: rec[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                               "dtype", "data") if k in rec}
    cfg = rec.get("config", {})
    out["config"] = {k: (_cut(v, 2 * text) if k == "workload" else _cut(v, text) if isinstance(v, str) else v) for k, v in cfg.items()}
    for k in ("detections_delivered", "records_gathered", "records_truncated"):
        if k in rec:
            out[k] = rec[k]
    rf = rec.get("roofline")
    if rf:
        o = {k: rf[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms") if k in rf}
        o["kernel"] = _cut(rf.get("kernel", ""), 48).split(" (")[0]
        o["algorithmic"] = _cut(rf.get("algorithmic", ""), text)
        out["roofline"] = o
    ri = rec.get("roofline_issue")
    if ri:
        out["roofline_issue"] = {k: ri[k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "source", "wave_time_waitcnt") if k in ri}
    cb = rec.get("cpu_baseline")
    if cb:
        o = {k: cb[k] for k in ("value", "unit", "cores", "kind", "cpu_model", "nproc") if k in cb}
        o["sample"] = _cut(cb.get("sample", ""), text + 60)
        if "phases_s" in cb:
            o["phases_s"] = cb["phases_s"]
        if "n_thread" in cb:
            o["n_thread"] = {k: cb["n_thread"][k] for k in ("value", "cores") if k in cb["n_thread"]}
        out["cpu_baseline"] = o
    lat = rec.get("latency_us_single_frame")
    if lat:
        out["latency_us_single_frame"] = dict(p50=lat["p50"], p99=lat["p99"], frames=lat.get("frames"))
    return out


def emit(res, subs, names, full_out):
    """ONE line <= LINE_LIMIT bytes on stdout: the headline record (contract keys, roofline, roofline_issue, cpu_baseline) and a `summary`
    with value / ms_per_step / roofline fraction / CPU baseline of every workload of the run.  Everything else -- the complete
    records of the headline and of the `also` workloads -- goes to the side file (`--full-out`, default bench_also.json beside bench.py;
    tools/measure_r06.sh commits a copy as profiles/r06_bench_default.json)."""
    full = dict(res)
    if subs:
        full["also"] = subs
    try:
        with open(full_out, "w") as f:
            json.dump(full, f, indent=1)
            f.write("\n")
    except OSError as e:   # a read-only checkout must not cost the line
        print("bench.py: side file %s not written: %s" % (full_out, e), file=sys.stderr)
    line = compact_record(res)
    if subs:
        summ = {}
        for nm, r2 in zip(names, [res] + subs):
            e = dict(value=r2["value"], unit=r2["unit"], ms_per_step=r2["ms_per_step"])
            if r2.get("roofline"):
                e["roofline"] = dict(bound=r2["roofline"]["bound"], frac=r2["roofline"]["frac"], kernel=_cut(r2["roofline"].get("kernel", ""), 40).split(" (")[0].split(" of ")[0],
                                     kernel_ms=r2["roofline"].get("kernel_ms"))
            if r2.get("roofline_issue"):
                e["valu_issue_frac"] = r2["roofline_issue"]["frac"]
            if r2.get("cpu_baseline"):
                e["cpu_1t"] = r2["cpu_baseline"]["value"]
            summ[nm] = e
        line["summary"] = summ
    line["full_records"] = os.path.basename(full_out)
    line = _r(line)
    out = json.dumps(line, separators=(",", ":"))
    # a line that still came out too long loses its free text first, then the summary's extras: the contract keys always survive
    if len(out) > LINE_LIMIT:
        line = _r(dict(compact_record(res, text=60), summary={k: dict(value=v["value"], unit=v["unit"]) for k, v in line.get("summary", {}).items()},
                       full_records=os.path.basename(full_out)))
        out = json.dumps(line, separators=(",", ":"))
    print(out, flush=True)
    return out

# ---------------------------------------------------------------------------------------------------------------- driver
class Env:
    pass


def measure(wl, env, steps, warmup, gather_every, want_cpu):
    import gc
    import torch
    import torch.distributed as dist
    from featuredetection_amd import parallel
    world, dev = env.world, env.dev
    recs_cap = wl.records_cap
    gather_every = wl.gather_every or gather_every
    truncated = False
    if wl.fixed_steps:   # a fixed job (config 5): its own number of steps, one light warm-up step (the job's first 256 images, run again timed)
        steps, warmup = wl.fixed_steps, min(warmup, 1)

    def barrier():
        wl.flush_out = wl.flush()
        if world > 1:
            dist.barrier()
        wl.sync()
        torch.cuda.synchronize()

    # bring clocks, stream pools and pinned staging buffers to their steady state before the W warm-up steps
    tpre = time.perf_counter()
    wl.step(0)
    wl.flush()
    est = time.perf_counter() - tpre
    i = 1
    while time.perf_counter() - tpre < 0.3 and est < 0.15:
        wl.step(i)
        i += 1
    wl.flush()
    for i in range(warmup):
        wl.step(i)
    wl.flush()
    gc.collect()
    gc.disable()   # a full gc pass over torch's object graph costs ~75 ms and would land on a random step
    barrier()
    if hasattr(wl, "qlens"):
        wl.qlens = []
    if hasattr(wl, "stage_frames"):
        wl.stage_sum, wl.stage_frames = np.zeros(4, np.int64), 0
    t0 = time.perf_counter()
    units, ndet, pending, gathered = 0, 0, [], 0
    gather_open, gather_s, ngathers = False, 0.0, 0
    for i in range(steps):
        n, out = wl.step(i)
        units += n
        pending.extend(out)
        if (i + 1) % gather_every == 0 or i + 1 == steps:
            # A gather point takes the records of the calls that have been collected so far; calls still in flight deliver theirs at
            # the next point, and the LAST point drains the pipeline -- every detection reaches the host inside the timed region.
            # (Rounds 1-6 drained at every point: an artificial stop of the rank's pipeline, 2 % of the headline, 4.7 % of the
            # 15-detector batch, 1.7 % of config 2.  FD_BENCH_DRAIN=1 restores it.)
            if i + 1 == steps or os.environ.get("FD_BENCH_DRAIN", "0") == "1":
                pending.extend(wl.flush())
            # the detection records of this rank's images since the last gather: real fd_detection fields
            # (image id(s), detector id, detections): the ids are scalars (one image) or one id per detection (a multi-frame call)
            recs = [parallel.pack_records(img if isinstance(img, np.ndarray) else np.full(len(d_), img), np.full(len(d_), det), d_)
                    for img, det, d_ in pending if len(d_)]
            local = np.concatenate(recs) if recs else np.zeros((0, parallel.RECORD_FIELDS))
            ndet += len(local)
            if world > 1:
                # the product's gather (csrc/dist.hip): fd_dist_gather_begin = a 64-byte header exchange + ONE ncclAllGather (librccl) of
                # max-count + 1 rows per rank, queued on the handle's own stream; the records of THIS interval travel while the next
                # interval's images are processed, and are collected (fd_dist_gather_end) at the next gather point.
                # parallel.gather_records is the torch.distributed twin, used by the CPU tests
                tg0 = time.perf_counter()
                if gather_open:
                    allr, tr = env.dist.gather_end()
                    gathered += len(allr)
                    truncated |= tr
                env.dist.gather_begin(local, recs_cap)
                gather_open = True
                gather_s += time.perf_counter() - tg0
                ngathers += 1
            pending = []
    if gather_open:   # the last interval's records
        tg0 = time.perf_counter()
        allr, tr = env.dist.gather_end()
        gathered += len(allr)
        truncated |= tr
        gather_s += time.perf_counter() - tg0
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    uu = torch.tensor([float(units), float(ndet)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(uu, op=dist.ReduceOp.SUM)
    dt, total_units, total_det = float(tt.item()), float(uu[0].item()), int(uu[1].item())
    rec = dict(metric=wl.metric, value=total_units / dt / 1e6, unit=wl.unit, n_gpus=world, steps=steps, warmup=warmup,
               ms_per_step=dt / steps * 1e3, higher_is_better=True, scaling=wl.scaling, vs_baseline=None, dtype=wl.dtype, data="synthetic",
               config=wl.config, detections_delivered=total_det)
    if world > 1:
        rec["records_gathered"] = gathered
        rec["records_truncated"] = bool(truncated)
        # host time inside fd_dist_gather_begin / _end per gather interval (rank 0): what a gather costs the rank's pipeline
        rec["gather_ms_per_interval"] = gather_s / max(1, ngathers) * 1e3
    if hasattr(wl, "extra_record") and env.rank == 0:
        rec.update(wl.extra_record())
    probe = wl.kernel_probe() if (env.rank == 0 and not getattr(env, "no_probe", False)) else None
    if probe:
        rec["roofline"] = probe[0]
        rec.update(probe[1])
    if want_cpu and env.rank == 0 and world == 1:
        cb = wl.cpu_baseline()
        if cb:
            rec["cpu_baseline"] = cb
    return rec


WORKLOADS = dict(cascade=Cascade, cascade_8frames=Cascade8, cascade_late=CascadeLate, cascade_group=CascadeGroup, hog_svm=HogSvm, ffp15=Ffp15, ffp15_2frames=Ffp15Two,
                 config5=Config5, sdm=Sdm, rvm=Rvm, aggregated=Aggregated)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cascade", choices=sorted(WORKLOADS) + ["wvm"], help="headline workload (wvm = cascade)")
    ap.add_argument("--also", default=None, help="comma-separated sub-records (default: hog_svm,sdm,ffp15,cascade_late,cascade_group,cascade_8frames,ffp15_2frames,config5 "
                                                 "when the headline is the default cascade; 'none' for none)")
    ap.add_argument("--images", type=int, default=0, help="config5 workload: images of the job (default 10000; FD_BENCH_CONFIG5_IMAGES)")
    ap.add_argument("--gather-every", type=int, default=4)
    ap.add_argument("--size", default=None, help="frame size WxH of the headline workload (cascade, hog_svm, ffp15, rvm, aggregated)")
    ap.add_argument("--frames-per-step", type=int, default=0, help="frames (sdm: batches) per step of the headline workload; 0 = its default")
    ap.add_argument("--frames-per-call", type=int, default=0, help="cascade workload: frames per call (one multi-frame pyramid); 0 = default")
    ap.add_argument("--per-frame-launches", action="store_true", help="cascade workload: one pyramid + cascade per frame (fd_detect_five_stage_batch) "
                                                                      "instead of the multi-frame entry points")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-out", default=os.path.join(ROOT, "bench_also.json"), help="side file for the complete records of the headline and the "
                                                                                       "`also` workloads (stdout carries one line of at most 8 KB)")
    ap.add_argument("--no-probe", action="store_true", help="skip kernel_probe (the HIP-event timing of the dominant kernel and the single-frame "
                                                            "latency loop): profiler runs want the timed steps' launches only")
    args = ap.parse_args()
    if args.workload == "wvm":
        args.workload = "cascade"
    if args.images > 0:
        os.environ["FD_BENCH_CONFIG5_IMAGES"] = str(args.images)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start one rank per GPU ourselves (the same command the driver uses for the scaling runs)
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    import torch
    import torch.distributed as dist
    from featuredetection_amd import capi

    env = Env()
    env.world = int(os.environ.get("WORLD_SIZE", "1"))
    env.rank = int(os.environ.get("RANK", "0"))
    env.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # smoke tests of the N > 1 path on a one-GPU box: every rank on device 0, torch.distributed over gloo, and the librccl
    # stand-in for fd_dist_* (FD_DIST_ONE_DEVICE=1 FD_BENCH_DIST_BACKEND=gloo FD_RCCL_LIB=tests/stub_rccl/librccl_stub.so)
    one_device = os.environ.get("FD_DIST_ONE_DEVICE", "0") == "1"
    if one_device:
        env.local_rank = 0
    if env.world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # N ranks share one host: at most 12 host threads per rank (this thread + the library's queue threads + its batch workers)
        os.environ.setdefault("FD_ASYNC_THREADS", "2")
        os.environ.setdefault("FD_BATCH_THREADS", "8")
        os.environ.setdefault("FD_BENCH_SDM_THREADS", "1")
        backend = os.environ.get("FD_BENCH_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", env.local_rank))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(env.local_rank)
    env.dev = torch.device("cuda", env.local_rank)
    env.ctx = capi.Context(env.local_rank, torch.cuda.current_stream().cuda_stream)
    if os.environ.get("FD_BENCH_WARM_STREAMS", "1") == "1":
        # The HIP runtime deals streams to its (four) hardware queues in creation order.  The context's ten lazily created streams (batch
        # pool, tail, auxiliary) are created here, so that a workload finds the same mapping whatever ran before it in this process (the
        # 15-detector batch: 8.4 G patches/s behind the headline and config 2, 9.7 G in every order with this), and FD_BENCH_PAD_STREAMS
        # idle streams round their number up to a multiple of four (the headline's contexts then land as in a fresh process)
        env.ctx.warm_streams()
        env.pad_streams = [capi.Context(env.local_rank) for _ in range(max(0, int(os.environ.get("FD_BENCH_PAD_STREAMS", "2"))))]   # (a context owns one stream)
    env.dist = None
    if env.world > 1:
        # fd_dist_*: rank 0 makes the communicator id (ncclGetUniqueId of librccl), torch.distributed only carries its 128 bytes
        uid = [capi.Dist.unique_id() if env.rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        env.dist = capi.Dist(env.ctx, env.rank, env.world, uid[0])

    def build(name, headline):
        kw = {}
        if headline and args.size and name != "sdm":
            kw["W"], kw["H"] = [int(v) for v in args.size.split("x")]
        if headline and args.frames_per_step > 0:
            kw["batches_per_step" if name == "sdm" else "frames_per_step"] = args.frames_per_step
            if name in ("rvm", "aggregated", "config5"):
                kw.pop("frames_per_step")
        if headline and args.frames_per_call > 0 and name == "cascade":
            kw["nb"] = args.frames_per_call
        if headline and args.per_frame_launches and name == "cascade":
            kw["multi"] = False
            kw.setdefault("nb", 16)
        return WORKLOADS[name](env, **kw)

    also = args.also
    if also is None:
        # (sdm in front of the 15-detector batch: behind it -- behind the release of its four frames' worth of handles -- the same workload
        # measures 1.15 instead of 1.23 M iterations/s, with one frame in flight 1.23; the other workloads do not care where they stand)
        also = "hog_svm,sdm,ffp15,cascade_late,cascade_group,cascade_8frames,ffp15_2frames,config5" if (args.workload == "cascade" and not args.size) else "none"
    also = [a for a in also.split(",") if a and a != "none"]
    want_cpu = not args.no_cpu_baseline
    env.no_probe = args.no_probe

    wl = build(args.workload, True)
    def release(name):
        # a workload's handles (pyramids, queues, record buffers) go back to the device before the next one is built
        gc.collect()
        if os.environ.get("FD_BENCH_MEMLOG"):
            free, total = torch.cuda.mem_get_info()
            print("[mem] rank %d after %s: %.2f GB in use of %.0f" % (env.rank, name, (total - free) / 1e9, total / 1e9), file=sys.stderr, flush=True)

    res = measure(wl, env, args.steps, args.warmup, args.gather_every, want_cpu)
    del wl
    release(args.workload)
    subs = []
    for name in also:
        w2 = build(name, False)
        subs.append(measure(w2, env, args.steps, args.warmup, args.gather_every, want_cpu))
        del w2
        release(name)
    if env.rank == 0:
        emit(res, subs, [args.workload] + also, args.full_out)
    if env.world > 1:
        env.dist.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
