/* include/fd_hip_bench.h -- measurement hooks of libfd_hip.so.  NOT part of the drop-in boundary (include/fd_hip.h): nothing a
 * maintainer of the reference would bind.  bench.py and tools/ use them to time the dominant kernel of a product call with HIP
 * events on the stream the kernel is launched on.
 */
#ifndef FD_HIP_BENCH_H_
#define FD_HIP_BENCH_H_
#include "fd_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* enable != 0: the detect entry points (fd_detect_five_stage, fd_detect_wvm, fd_detect_hog_svm[_begin/_end], fd_sdm_fit_batch)
 * bracket their dominant kernel(s) with two hipEvents on the context's stream. */
int fd_ctx_set_kernel_timing(fd_ctx* ctx, int enable);   /* enable == 2: WVM cascades bracket the dense pre-filter kernel only;
                                                          * enable == 3: every k_wvb_chain2 launch of stage B (one per phase), summed */
/* duration (ms) between those events for the last timed call on this context and the name of the bracketed kernel(s) */
int fd_last_kernel_ms(fd_ctx* ctx, const char** kernel_name, float* ms);
/* With fd_ctx_set_kernel_timing(ctx, 2): duration (ms) of the last k_wvm_prefilter_group launch of a batch entry point on this context
 * (HIP events on the stream it was launched on) and the number of detectors it served; members = 0: no group launch was timed.  Meant
 * for a batch that holds ONE group (several groups of a batch are queued by different host threads: the last writer wins). */
int fd_last_group_prefilter_ms(fd_ctx* ctx, float* ms, int* members);

/* Windows the last finished cascade run of this handle handed to stage B (the pre-filter's queue; all frames of a multi-frame call
 * together); -1 before the first run.  bench.py reports the spread over the calls of a run. */
int64_t fd_wvm_last_queue_length(const fd_wvm* wvm);
/* The stage-B plan of the last finished run: out[0] = phases, then per phase {first generation, end generation, windows alive at its
 * start} (13 int64 at most: 1 + 3 * 4, the most phases a plan has); -1 where unknown.  bench.py derives the chain kernel's algorithmic work from it. */
int fd_wvm_last_stage_b_plan(const fd_wvm* wvm, int64_t* out);
/* How the last finished five-stage run of this handle did its overlap elimination: -1 on the host (no device tail was queued), 0 on
 * the device (csrc/fs_tail.hpp), > 0: the device kernel gave up and the host redid it -- 1 the order of the positives could not be
 * proven to be the reference's (tied or saturated probabilities), 2 more positives in a frame than the kernel holds, 4 window ids
 * beyond 32 bits, 0x100 stage-B queue / positive buffer overflow. */
int fd_wvm_last_tail_state(const fd_wvm* wvm);
/* How the last fd_detect_five_stage / fd_detect_five_stage_image call of this handle got its SVM scores: -1 the survivors of the host
 * overlap elimination were scored in a launch of their own (two host round trips: FD_FS_SPEC=0, the device tail, models without the
 * dense stage B or the u8 MFMA SVM), 0 the scores of all WVM positives were queued behind the cascade and used (one wait), 1 they were
 * queued but the frame had more positives than the launch covered (or stage B was rerun): the call fell back to the two round trips. */
int fd_wvm_last_spec_state(const fd_wvm* wvm);

/* Test hook, needs no GPU: the rect sums (WvmClassifier.cpp:277-306) of every used level of `md` for n equalised patches, computed
 * from the tables of the dense stage B with the operand addressing of its MFMA kernel.  out[i * ncols + c]: c runs over the levels
 * 0 .. numUsed - 1, grey values 1 .. cntval - 1 inside a level.  Returns ncols (out may be NULL), -1 when the model has no dense
 * stage B.  phase_gen (5 ints, may be NULL) receives the generation boundaries of the phases, -1 padded. */
int64_t fd_debug_wvb_rect_sums(const fd_wvm_model* md, const uint8_t* patches, int64_t n, int32_t* out, int32_t* phase_gen);

/* Test hook, needs no GPU: the work plan of the dense pre-filter (k_wvm_prefilter) for n_layers layers of nx[i] x ny[i] windows,
 * `frames` frames per launch, vertical window step sy, patch height ph, on a device with `slots` resident wavefronts.  A lane walks
 * down a column through K windows; lane task t of layer i is column t % nx[i], row group t / nx[i] (rows g K .. min(ny, (g + 1) K) - 1),
 * 64 tasks per tile.  Returns K (FD_WVD_K pins it); tile_first[i] = first tile of layer i inside a frame, tile_first[n_layers] = tiles
 * per frame. */
int fd_debug_wvd_plan(const int32_t* nx, const int32_t* ny, int n_layers, int frames, int sy, int ph, int slots, int32_t* tile_first);

/* Test hook: hyperplane distances of n u8 vectors through both instantiations of the u8 RBF MFMA kernel (8 and 16 wavefronts per
 * workgroup).  fd_detect_five_stage scores one frame's positives with either, depending on how many the previous frame had, and
 * relies on bit-identical sums; tests/test_gpu_cascade_hardening.py compares them.  (fd_detect_five_stage keeps per-call state in
 * the fd_wvm handle: one handle must not be used from two threads at a time.) */
int fd_debug_svm_u8_both(fd_ctx* ctx, const fd_svm* svm, const uint8_t* features, int64_t n, double* out8, double* out16);

#ifdef __cplusplus
}
#endif
#endif /* FD_HIP_BENCH_H_ */
