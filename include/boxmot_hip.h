/*
 * boxmot_hip.h -- C ABI of the MI355X (gfx950) BoT-SORT update path.
 *
 * Drop-in boundary: these entry points are what the reference's native FFI for
 * this path binds (boxmot/native/cpp/trackers/botsort/include/botsort/c_api.hpp:17-61,
 * mirrored in Python by boxmot/native/trackers/botsort.py:94-146 and called through
 * boxmot/native/trackers/_common.py:158-221).  Conventions are the reference's:
 *   - plain C, opaque handle, caller owns every buffer, inputs row-major
 *     contiguous fp32 / uint8, output buffer caller-allocated (rows x 9 fp32);
 *   - return 1 on success, 0 on failure; the message of the last failure on the
 *     calling thread is returned by boxmot_hip_last_error();
 *   - a handle is not thread-safe; distinct handles may be used concurrently from
 *     distinct threads (the reference's contract, reid_capi.h:61-70; tests/test_gpu_threads.py);
 *   - Devices: a handle belongs to the HIP device that is current on the creating
 *     thread at create (hipSetDevice before create selects it).  Every entry point
 *     that takes a handle makes that device current for the call and restores the
 *     caller's device before returning, so one process may own handles on several
 *     GPUs and call them from any thread; device pointers passed to a handle
 *     (_step_device, _apply_device, ingest rings) must live on the handle's device.
 *     boxmot_hip_botsort_device() reports it.
 * Differences from the precedent, all deliberate (DESIGN.md "Boundary"):
 *   - thresholds are double: the Python tracker (the parity oracle) compares in
 *     fp64 and its YAML defaults are not representable in fp32;
 *   - second_match_thresh / unconfirmed_match_thresh / unconfirmed_emb_scale /
 *     removed_stracks_buffer are configurable (the C++ reference hard-codes them,
 *     tracker.cpp:435,465,476; the Python reference exposes them, botsort.py:81-85);
 *   - one handle owns n_streams independent trackers advanced by one launch set
 *     (boxmot_hip_botsort_update_batch / _step_device);
 *   - camera-motion compensation: cmc_method = "sof" (configs/trackers/botsort.yaml) or "ecc" (the constructor default) runs that
 *     estimator of the reference on the device inside update (it needs the frame on every call); NULL / "" / "none" estimate
 *     nothing -- a 2x3 warp supplied with boxmot_hip_*_set_warp (what the reference's cmc.apply() returns, e.g. from
 *     boxmot_hip_sof_apply / boxmot_hip_ecc_apply or a host-side estimator) is applied on the device; the feature-matching
 *     estimators ("orb", "sift") are not built (create fails);
 *   - the reference's own symbol names and struct layouts are exported next to these (boxmot_compat.h).
 * There is no CPU fallback: every entry point fails with an error when no HIP
 * device is usable.
 */
#ifndef BOXMOT_HIP_H
#define BOXMOT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* replaces BoxMOTBotSortConfig, c_api.hpp:17-32 */
typedef struct BoxMOTHipBotSortConfig {
    double track_high_thresh;
    double track_low_thresh;
    double new_track_thresh;
    int track_buffer;
    double match_thresh;
    double proximity_thresh;
    double appearance_thresh;
    const char* cmc_method;          /* NULL / "" / "none", or "sof" / "ecc" (estimated on the device inside update) */
    int frame_rate;
    int fuse_first_associate;
    int with_reid;
    int max_obs;                     /* accepted for signature parity; display-only state */
    const char* reid_model_path;     /* OSN1 (OSNet) or CLP1 (CLIP-ReID ViT-B/16) weight blob (boxmot_amd.reid_weights / clip_weights) or NULL */
    const char* reid_preprocess;     /* NULL or "resize" */
    double second_match_thresh;
    double unconfirmed_match_thresh;
    double unconfirmed_emb_scale;
    int removed_stracks_buffer;
    /* capacity / batching */
    int n_streams;                   /* >= 1 independent trackers in this handle */
    int max_tracks;                  /* track slots per stream (live + lost) */
    int max_dets;                    /* detections per frame per stream */
    int emb_dim;                     /* appearance vector length (512 for OSNet) */
    int n_class_lists;               /* 1, or nr_classes when per_class=True */
    int tracker_kind;                /* 0 = BoT-SORT; 1 = ByteTrack (set by boxmot_hip_bytetrack_default_config) */
    int is_obb;                      /* 0 = axis-aligned detections [x1 y1 x2 y2 conf cls] -> rows [x1 y1 x2 y2 id conf cls det_ind];
                                      * 1 = oriented detections [cx cy w h angle conf cls] -> rows [cx cy w h angle id conf cls det_ind]
                                      * (basetracker.py:179-201 decides this from the first frame's column count; botsort.py:120-131,
                                      * bytetrack.py:283-291, KalmanFilterXYWH(ndim=5), iou_batch_obb): the 10-state filter and the
                                      * rotated-rectangle IoU run in the frame step; cmc_method must be none, embeddings come as embs */
} BoxMOTHipBotSortConfig;

typedef struct BoxMOTHipBotSort BoxMOTHipBotSort;
typedef struct BoxMOTHipReID BoxMOTHipReID;

/* fills the constructor defaults of BotSort (botsort.py:66-86) */
void boxmot_hip_botsort_default_config(BoxMOTHipBotSortConfig* config);

/* ByteTrack (boxmot/trackers/bbox/bytetrack/bytetrack.py:201-408; the reference's native twin:
 * boxmot/native/cpp/trackers/bytetrack) runs on the same handle type and entry points: BoT-SORT grew out of it, so the
 * frame step is the same sequence of stages with an (x, y, aspect, height) Kalman state, score fusion in the first and the
 * unconfirmed association, fixed 0.5 / 0.7 thresholds for the second / unconfirmed association, no appearance, no class
 * vote and an unbounded removed list.  This fills ByteTrack's constructor defaults (bytetrack.py:225-233: min_conf 0.1 ->
 * track_low_thresh, track_thresh 0.45 -> track_high_thresh and new_track_thresh, match_thresh 0.8, track_buffer 25,
 * frame_rate 30) and tracker_kind = 1; create / update / ... are the boxmot_hip_botsort_* functions. */
void boxmot_hip_bytetrack_default_config(BoxMOTHipBotSortConfig* config);

/* boxmot_botsort_create / _destroy / _reset, c_api.hpp:36-40 */
BoxMOTHipBotSort* boxmot_hip_botsort_create(const BoxMOTHipBotSortConfig* config);
void boxmot_hip_botsort_destroy(BoxMOTHipBotSort* handle);
int boxmot_hip_botsort_reset(BoxMOTHipBotSort* handle);

/* Capacities.  The reference's track and detection lists are Python lists without a limit (botsort.py:177-250); the device tables are
 * created at max_tracks / max_dets and GROW on demand: a host-API update (update / update_stream / update_batch[_frames]) that would
 * not fit -- more detections than max_dets, or live tracks + this frame's detections > max_tracks -- first re-makes the tables at
 * (at least) twice the size and carries the state over (ids, filters, lists unchanged), so results do not depend on the initial
 * sizes.  The limit is the LDS state of the assignment solver (about 3000 tracks + detections per stream), reported as an error.
 * The device-resident step has no host in the loop to do this: size the handle with _reserve before a burst (it reports an
 * overflow through boxmot_hip_botsort_status as before).  _capacity returns the current sizes and how often the tables grew. */
int boxmot_hip_botsort_reserve(BoxMOTHipBotSort* handle, int max_tracks, int max_dets);
int boxmot_hip_botsort_capacity(BoxMOTHipBotSort* handle, int* max_tracks, int* max_dets, int* n_grows);

/* boxmot_botsort_update, c_api.hpp:42-55: one frame of stream 0, synchronous. */
int boxmot_hip_botsort_update(
    BoxMOTHipBotSort* handle,
    const float* dets, int det_rows, int det_cols,
    const float* embs, int emb_rows, int emb_cols,
    const uint8_t* image, int image_rows, int image_cols, int image_channels,
    float* out_tracks, int out_capacity_rows, int out_cols,
    int* out_rows, int* out_is_obb);

/* Same, for stream `stream` and active-list (class) `class_list`; `frame_count`
 * >= 0 sets the tracker's frame counter before the step (per-class fan-out,
 * basetracker.py:223-263), -1 leaves it alone. */
int boxmot_hip_botsort_update_stream(
    BoxMOTHipBotSort* handle, int stream, int class_list, int frame_count,
    const float* dets, int det_rows, int det_cols,
    const float* embs, int emb_rows, int emb_cols,
    const uint8_t* image, int image_rows, int image_cols, int image_channels,
    float* out_tracks, int out_capacity_rows, int out_cols,
    int* out_rows, int* out_is_obb);

/* Camera-motion compensation, application half (STrack.multi_gmc, botsort_track.py:117-132; called from
 * BotSort._apply_aabb_camera_motion_compensation, botsort.py:134-145).  `warp_2x3` = the row-major 2x3 matrix
 * the reference's cmc.apply(img, dets) returns ([r00 r01 tx; r10 r11 ty]); it is applied to the predicted pool and
 * to the unconfirmed tracks in the NEXT update / update_stream / update_batch of `stream` and then dropped.
 * NULL clears a pending warp.  The estimators are boxmot_hip_ecc_* and boxmot_hip_sof_* below. */
int boxmot_hip_botsort_set_warp(BoxMOTHipBotSort* handle, int stream, const double* warp_2x3);

/* One frame for each of the first n_streams streams in one launch set.  det_rows[s] == -1 leaves stream s untouched
 * in this call (the reference's replay loop does not pass frames without detections to the tracker, replay.py:318-341).
 * dets[s] -> (det_rows[s], 6) fp32; embs[s] -> (det_rows[s], emb_cols) fp32 or embs == NULL;
 * images[s] -> (rows, cols, 3) uint8 BGR or NULL to keep the previously uploaded frame;
 * out[s] -> (out_capacity_rows, 9) fp32. */
int boxmot_hip_botsort_update_batch(
    BoxMOTHipBotSort* handle, int n_streams,
    const float* const* dets, const int* det_rows,
    const float* const* embs, int emb_cols,
    const uint8_t* const* images, int image_rows, int image_cols, int image_channels,
    float* const* out_tracks, int out_capacity_rows, int* out_rows);
/* the same with the frames already on the device (one pointer per stream, e.g. boxmot_hip_ingest_device_frames(ring, slot)):
 * detections in, rows out over PCIe, no frame copy in this call */
int boxmot_hip_botsort_update_batch_frames(
    BoxMOTHipBotSort* handle, int n_streams,
    const float* const* dets, const int* det_rows,
    const float* const* embs, int emb_cols,
    const uint8_t* const* d_frames, int image_rows, int image_cols,
    float* const* out_tracks, int out_capacity_rows, int* out_rows);

/* Device-resident step for all streams (asynchronous on the handle's stream):
 *   d_dets   [n_streams][max_dets][6] fp32, d_det_rows [n_streams] int32,
 *   d_embs   [n_streams][max_dets][emb_dim] fp32 or NULL (NULL + with_reid -> ReID runs on d_frames),
 *   d_frames device array of n_streams device pointers to (rows, cols, 3) uint8 BGR frames, or NULL,
 *   d_out    [n_streams][max_dets][8] fp32, d_out_rows [n_streams] int32.
 * Call boxmot_hip_botsort_synchronize() before reading the outputs. */
int boxmot_hip_botsort_step_device(
    BoxMOTHipBotSort* handle,
    const float* d_dets, const int* d_det_rows, const float* d_embs,
    const uint8_t* const* d_frames, int image_rows, int image_cols,
    float* d_out, int* d_out_rows);
int boxmot_hip_botsort_synchronize(BoxMOTHipBotSort* handle);
/* the hipStream_t the handle launches on (as void*), for event timing by the caller */
void* boxmot_hip_botsort_stream(BoxMOTHipBotSort* handle);
/* HIP-event stopwatch on the handle's stream: start records an event, stop records a second one,
 * synchronises on it and returns the elapsed device milliseconds in between. */
int boxmot_hip_botsort_timer_start(BoxMOTHipBotSort* handle);
int boxmot_hip_botsort_timer_stop_ms(BoxMOTHipBotSort* handle, double* out_ms);
/* accumulated device milliseconds of the ReID forward kernels since the last call (and their count) */
int boxmot_hip_botsort_reid_kernel_ms(BoxMOTHipBotSort* handle, double* out_ms, int* out_launches);
/* profiling aid: shader-clock (100 MHz wall clock) stamps of the phases of the last step of the first
 * stream of the launch: start, det prep, det features, pool lists, predict, cost, assignment, updates,
 * second association, unconfirmed, births, bookkeeping, end */
int boxmot_hip_botsort_phase_clocks(BoxMOTHipBotSort* handle, long long* out16);
/* per-stream status words (0 ok, 1 track capacity, 2 class capacity, 3 assignment stall).  The host-API updates report a
 * non-zero word of their streams as an error once and clear it; the device-resident step leaves the words set until
 * boxmot_hip_botsort_reset -- poll them here. */
int boxmot_hip_botsort_status(BoxMOTHipBotSort* handle, int* out_status, int capacity);

/* ReID weights from memory (same blob format as reid_model_path). */
int boxmot_hip_botsort_set_reid_blob(BoxMOTHipBotSort* handle, const float* blob, long n_floats);
/* precision / kernel family of the ReID engine: 0 = per-layer fp32 kernels, 1 = fused fp16 MFMA kernels */
int boxmot_hip_botsort_set_reid_mode(BoxMOTHipBotSort* handle, int mode);

/* boxmot_botsort_last_reid_*_time_ms, c_api.hpp:57-60 (device time of the last update, HIP events) */
int boxmot_hip_botsort_last_reid_time_ms(BoxMOTHipBotSort* handle, double* out_ms);
int boxmot_hip_botsort_last_reid_preprocess_time_ms(BoxMOTHipBotSort* handle, double* out_ms);
int boxmot_hip_botsort_last_reid_process_time_ms(BoxMOTHipBotSort* handle, double* out_ms);
int boxmot_hip_botsort_last_reid_postprocess_time_ms(BoxMOTHipBotSort* handle, double* out_ms);
int boxmot_hip_botsort_last_track_time_ms(BoxMOTHipBotSort* handle, double* out_ms);

/* Parity/debug: copy the live tracks of one stream to host arrays sized for max_tracks rows.
 * which = 0 active list `class_list`, 1 lost list.  Returns the row count in *out_rows.
 * ints[r] = {id, state, is_activated, frame_id, start_frame, tracklet_len},
 * kf[r] = mean[8] ++ cov[64] fp64, smooth[r] = emb_dim fp32, misc[r] = {conf, cls, det_ind}. */
int boxmot_hip_botsort_state_dump(
    BoxMOTHipBotSort* handle, int stream, int which, int class_list,
    int* ints, double* kf, float* smooth, float* misc, int* out_rows,
    int* out_frame_count, int* out_id_count);

/* Parity/debug, the other half of state_dump: the association cost matrices of the LAST update of one stream, as the reference's
 * functions return them -- (tracks, detections) row-major fp64.  Off by default (the step writes nothing); _enable(handle, 1) makes
 * every following step keep copies.
 *   stage 0: first association     (boxmot/trackers/bbox/botsort/botsort.py:306-317: pool x high-confidence detections)
 *   stage 1: second association    (botsort.py:356: remaining tracked x low-confidence detections, IoU only)
 *   stage 2: unconfirmed tracks    (botsort.py:396-413: unconfirmed x left-over detections)
 *   plane 0: the matrix handed to linear_assignment (after fuse_score, matching.py:139-147, the proximity / appearance gates and
 *            np.minimum); plane 1: iou_distance (matching.py:46-80) before score fusion; plane 2: embedding_distance
 *            (matching.py:85-107) where the step evaluated it -- NaN elsewhere: the pairs behind the IoU gate are never evaluated on
 *            the sparse path, every pair is on the dense fallback (more than 4096 ungated pairs).
 * Row r is the r-th track of the stage's track list, column c the c-th detection of its detection list, in the reference's order. */
int boxmot_hip_botsort_debug_costs_enable(BoxMOTHipBotSort* handle, int on);
int boxmot_hip_botsort_debug_costs(
    BoxMOTHipBotSort* handle, int stream, int stage, int plane,
    double* out, long out_capacity, int* out_rows, int* out_cols);

/* ReID C ABI, replaces boxmot_reid_capi_* (boxmot/native/cpp/trackers/base/include/boxmot/trackers/base/reid_capi.h:36-94) */
BoxMOTHipReID* boxmot_hip_reid_create(const char* model_path, const float* blob, long n_floats, int max_crops);
void boxmot_hip_reid_destroy(BoxMOTHipReID* handle);
int boxmot_hip_reid_feature_dim(BoxMOTHipReID* handle);
/* "resize" (default) or "resize_pad" (reid/core/preprocessing.py:12-45; get_preprocess_fn :56-65) */
int boxmot_hip_reid_set_preprocess(BoxMOTHipReID* handle, const char* name);
int boxmot_hip_reid_set_mode(BoxMOTHipReID* handle, int mode);
/* boxes (n, box_cols>=4) fp32 xyxy -- or, when box_cols is 5, 7 or 9, oriented boxes [cx, cy, w, h, angle, ...] whose rectified crop
   is the reference's _crop_obb (boxmot/reid/backends/base_backend.py:91-122, 157: cv2.getRotationMatrix2D + cv2.warpAffine INTER_LINEAR,
   zero border); out (n, feature_dim) fp32, L2-normalised */
int boxmot_hip_reid_compute_features(
    BoxMOTHipReID* handle, const uint8_t* image, int image_rows, int image_cols, int image_channels,
    const float* boxes, int n_boxes, int box_cols, float* out_features, int out_capacity_rows);
/* intermediate: normalised crops (n, 256, 128, 3) fp32 NHWC, RGB */
int boxmot_hip_reid_preprocess(
    BoxMOTHipReID* handle, const uint8_t* image, int image_rows, int image_cols, int image_channels,
    const float* boxes, int n_boxes, int box_cols, float* out_crops);

/* device milliseconds (HIP events on the handle's stream) of the last compute_features: crop / resize / normalise, and the
 * backbone forward (cf. boxmot_botsort_last_reid_{preprocess,process}_time_ms, c_api.hpp:58-59) */
int boxmot_hip_reid_last_time_ms(BoxMOTHipReID* handle, double* out_preprocess_ms, double* out_process_ms);

const char* boxmot_hip_last_error(void);
/* number of visible HIP devices (0 when none / runtime unusable) */
int boxmot_hip_device_count(void);
/* the HIP device a handle was created on (the device every call on it runs on); -1 for NULL */
int boxmot_hip_botsort_device(BoxMOTHipBotSort* handle);

/* ------------------------------------------------------------------------------------------------
 * Camera-motion estimation: the ECC estimator (boxmot/motion/cmc/ecc.py:14-96; the reference's StrongSORT applies it on every
 * frame, strongsort.py:63,83-86, and BoT-SORT / DeepOCSORT accept it as cmc_method "ecc").  Arguments = the constructor's
 * (scale 0.15, eps 1e-5, max_iter 100; MOTION_TRANSLATION, grayscale, no alignment are fixed).  apply = ECC.apply(img, dets):
 * the first call of a stream stores the frame and returns the identity; later calls return the 2 x 3 warp (row-major doubles,
 * translation in full-resolution pixels) between the previous and this frame, or the identity when the iteration hits one of
 * OpenCV's "did not converge" exits (ecc.py:67-76).  Feed the result to boxmot_hip_*_set_warp.  apply_device takes a frame that
 * already is in HBM (e.g. an ingest-ring slot).  reset(stream < 0: all) forgets the previous frame.
 * ------------------------------------------------------------------------------------------------ */
typedef struct BoxMOTHipEcc BoxMOTHipEcc;
BoxMOTHipEcc* boxmot_hip_ecc_create(int n_streams, int image_rows, int image_cols, double scale, double eps, int max_iter);
void boxmot_hip_ecc_destroy(BoxMOTHipEcc* handle);
int boxmot_hip_ecc_reset(BoxMOTHipEcc* handle, int stream);
int boxmot_hip_ecc_apply(BoxMOTHipEcc* handle, int stream, const uint8_t* image, int image_rows, int image_cols, int image_channels,
                         double* out_warp_2x3, int* out_iterations);
int boxmot_hip_ecc_apply_device(BoxMOTHipEcc* handle, int stream, const uint8_t* d_frame, double* out_warp_2x3, int* out_iterations);

/* ------------------------------------------------------------------------------------------------
 * Camera-motion estimation: the sparse-optical-flow estimator (boxmot/motion/cmc/sof.py:14-147) -- the cmc_method of
 * configs/trackers/botsort.yaml and the estimator DeepOCSORT constructs (deepocsort.py:297).  Arguments = SOF.__init__'s (scale 0.15,
 * min_inliers 8, min_inlier_ratio 0.2, ransac_reproj_threshold 3.0); the OpenCV arguments sof.py fixes (1000 corners at quality 0.01,
 * 21 x 21 window, maxLevel 3 = up to 4 pyramid levels, 30 iterations / 0.01) are fixed here too.  apply = SOF.apply(img, dets): dets = (n, >= 4) fp32 rows whose
 * first four columns are tlbr boxes in frame pixels (masked out of the corner detector, base_cmc.py:63-105), NULL / 0 for none; the
 * first call of a stream detects keypoints and returns the identity; later calls track them, fit the partial-affine 2 x 3 warp
 * (row-major doubles, translation in full-resolution pixels) and refresh the keypoints; too few tracked points or a weak fit return
 * the identity (sof.py:95-115).  out_info8 (optional) = mode (0 initialising, 1 tracked, 2 re-detected), keypoints kept for the next
 * frame, tracked points, RANSAC inliers, RANSAC iterations, estimate accepted, corners detected, initialized.  Feed the warp to
 * boxmot_hip_*_set_warp.  apply_device runs ALL streams of the handle on frames that already are in HBM (d_frames: device table of
 * n_streams pointers; d_dets [n_streams][max_dets][det_stride] and d_ndets [n_streams] on the device, or NULL) and returns
 * n_streams warps (and 8 ints each).  keypoints copies out the points the next frame will track.
 * ------------------------------------------------------------------------------------------------ */
typedef struct BoxMOTHipSof BoxMOTHipSof;
BoxMOTHipSof* boxmot_hip_sof_create(int n_streams, int image_rows, int image_cols, double scale, int min_inliers, double min_inlier_ratio,
                                    double ransac_reproj_threshold);
void boxmot_hip_sof_destroy(BoxMOTHipSof* handle);
int boxmot_hip_sof_reset(BoxMOTHipSof* handle, int stream);
int boxmot_hip_sof_apply(BoxMOTHipSof* handle, int stream, const uint8_t* image, int image_rows, int image_cols, int image_channels,
                         const float* dets, int n_dets, int det_stride, double* out_warp_2x3, int* out_info8);
int boxmot_hip_sof_apply_device(BoxMOTHipSof* handle, const uint8_t* const* d_frames, const float* d_dets, const int* d_ndets, int max_dets,
                                int det_stride, double* out_warps, int* out_info8);
int boxmot_hip_sof_keypoints(BoxMOTHipSof* handle, int stream, float* out_xy, int capacity, int* out_n);
/* test access to the corner detector's images of the last frame: which = 0 minimum-eigenvalue map (fp32 [h][w]), 1 detection mask
 * (uint8 [h][w]), 2 the scaled grayscale frame (uint8 [h][w]) */
int boxmot_hip_sof_debug_map(BoxMOTHipSof* handle, int stream, int which, void* out, int capacity_bytes, int* out_h, int* out_w);

/* ------------------------------------------------------------------------------------------------
 * Frame ingest ring (no counterpart in the reference: its trackers receive a numpy frame per call, basetracker.py:120-147, and
 * its native binding copies it, native/trackers/botsort.py:200-230).  n_slots x n_streams page-locked host frames with device
 * twins: decode frame t + 1 into slot (t + 1) % n_slots while frame t is tracked.
 *   host_ptr(slot, stream)   where the caller writes the rows x cols x 3 uint8 BGR frame of `stream`
 *   submit(slot, n)          asynchronous H2D copy of the first n streams' frames on the ring's copy stream
 *   wait(slot, hip_stream)   makes that stream wait for the slot's upload (no host wait) -- pass the tracker handle's stream
 *   device_frames(slot)      device table of per-stream frame pointers = the d_frames argument of *_step_device[_frames]
 *   release(slot, stream)    marks the slot consumed once the work queued on `stream` so far is done; the next submit waits for it
 *   host_done(slot)          blocks until the slot's upload has left the host buffer (only needed before refilling it)
 * ------------------------------------------------------------------------------------------------ */
typedef struct BoxMOTHipIngest BoxMOTHipIngest;
BoxMOTHipIngest* boxmot_hip_ingest_create(int n_slots, int n_streams, int image_rows, int image_cols);
void boxmot_hip_ingest_destroy(BoxMOTHipIngest* handle);
uint8_t* boxmot_hip_ingest_host_ptr(BoxMOTHipIngest* handle, int slot, int stream);
const uint8_t* const* boxmot_hip_ingest_device_frames(BoxMOTHipIngest* handle, int slot);
int boxmot_hip_ingest_submit(BoxMOTHipIngest* handle, int slot, int n_streams);
int boxmot_hip_ingest_wait(BoxMOTHipIngest* handle, int slot, void* consumer_hip_stream);
int boxmot_hip_ingest_release(BoxMOTHipIngest* handle, int slot, void* consumer_hip_stream);
int boxmot_hip_ingest_host_done(BoxMOTHipIngest* handle, int slot);

/* ------------------------------------------------------------------------------------------------
 * DeepOCSORT (boxmot/trackers/bbox/deepocsort/deepocsort.py:235-492).  The reference has no native backend
 * for this tracker; the entry points follow the BoT-SORT ones above (same buffer, error and ownership
 * conventions, c_api.hpp:36-61).  Fields = the constructor arguments of DeepOcSort (deepocsort.py:263-281) and
 * of BaseTracker (basetracker.py:19-31).
 * ------------------------------------------------------------------------------------------------ */
enum { BOXMOT_HIP_ASSO_IOU = 0, BOXMOT_HIP_ASSO_GIOU = 1, BOXMOT_HIP_ASSO_DIOU = 2, BOXMOT_HIP_ASSO_CIOU = 3,
       BOXMOT_HIP_ASSO_HMIOU = 4, BOXMOT_HIP_ASSO_CENTROID = 5 };
typedef struct BoxMOTHipDeepOcSortConfig {
    double det_thresh;
    int max_age;                     /* <= 45: the reference's filter keeps a 50-entry observation history (xysr.py:18) */
    int max_obs;
    int min_hits;
    double iou_threshold;
    int delta_t;                     /* 1..3 */
    double inertia;
    double w_association_emb;
    double alpha_fixed_emb;
    double aw_param;
    int embedding_off;
    int cmc_off;                     /* informational: a warp is applied only when one was supplied (boxmot_hip_deepocsort_set_warp) */
    int aw_off;
    double Q_xy_scaling;
    double Q_s_scaling;
    const char* reid_model_path;     /* OSN1 / CLP1 blob or NULL when embeddings are supplied */
    int n_streams;
    int max_tracks;
    int max_dets;
    int emb_dim;
    /* OC-SORT (trackers/bbox/ocsort/ocsort.py:334-349) = this step with embedding_off = cmc_off = 1, plus its optional BYTE
     * association of detections with min_conf < score < det_thresh (ocsort.py:393-399, 456-485); use_byte needs embedding_off */
    int use_byte;
    double min_conf;
    /* BaseTracker's asso_func (basetracker.py:28,69-71; AssociationFunction, trackers/association/iou.py:118-423): the function
     * behind every "iou" matrix of the step (association.py:95, deepocsort.py:420, ocsort.py:457,486).  `centroid` normalises by
     * the frame diagonal: frame_w / frame_h = the size the reference reads off its first image (basetracker.py:175-180); left 0
     * they are taken from the first host update that carries an image (device-resident steps need them set). */
    int asso_func;                   /* BOXMOT_HIP_ASSO_* */
    int frame_w, frame_h;
    /* 1: oriented detections [cx cy w h angle conf cls] -> rows [cx cy w h angle id conf cls det_ind] -- OC-SORT with is_obb
     * (ocsort.py:49-87, :121-154, :241-310, :363-555: the 9-state KalmanFilterXYSR(dim_x=9, dim_z=5) incl. the aligned measurement,
     * the interpolated angle of the re-update and the damped angular velocity; iou_batch_obb).  Needs embedding_off = 1 (DeepOcSort
     * has no oriented mode) and asso_func = BOXMOT_HIP_ASSO_IOU; warps are refused; state_dump returns 90 doubles per track */
    int is_obb;
} BoxMOTHipDeepOcSortConfig;

typedef struct BoxMOTHipDeepOcSort BoxMOTHipDeepOcSort;

/* fills the constructor defaults (deepocsort.py:263-281, basetracker.py:19-31); note cmc_off defaults to 0 there */
void boxmot_hip_deepocsort_default_config(BoxMOTHipDeepOcSortConfig* config);
BoxMOTHipDeepOcSort* boxmot_hip_deepocsort_create(const BoxMOTHipDeepOcSortConfig* config);
void boxmot_hip_deepocsort_destroy(BoxMOTHipDeepOcSort* handle);
int boxmot_hip_deepocsort_reset(BoxMOTHipDeepOcSort* handle);
/* capacities grow on demand in the host-API updates (see boxmot_hip_botsort_reserve) */
int boxmot_hip_deepocsort_reserve(BoxMOTHipDeepOcSort* handle, int max_tracks, int max_dets);
int boxmot_hip_deepocsort_capacity(BoxMOTHipDeepOcSort* handle, int* max_tracks, int* max_dets, int* n_grows);
/* KalmanBoxTracker.apply_affine_correction (deepocsort.py:190-209, xysr.py:311-366): the 2x3 warp cmc.apply returned,
 * applied to every track of `stream` at the start of its NEXT update (deepocsort.py:347-351) and then dropped. */
int boxmot_hip_deepocsort_set_warp(BoxMOTHipDeepOcSort* handle, int stream, const double* warp_2x3);
/* DeepOcSort.update for stream 0 (deepocsort.py:302-492): arguments as boxmot_hip_botsort_update; embs == NULL runs the
 * ReID engine on every detection with conf > det_thresh.  Rows [x1,y1,x2,y2,id,conf,cls,det_ind,0]. */
int boxmot_hip_deepocsort_update(
    BoxMOTHipDeepOcSort* handle,
    const float* dets, int det_rows, int det_cols,
    const float* embs, int emb_rows, int emb_cols,
    const uint8_t* image, int image_rows, int image_cols, int image_channels,
    float* out_tracks, int out_capacity_rows, int out_cols,
    int* out_rows, int* out_is_obb);
/* Same for stream `stream`.  frame_count >= 0 sets the tracker's frame counter before the step and *id_count_inout
 * (ids issued so far) is loaded before and stored after it: the per-class fan-out of the reference rewinds the frame
 * counter for every class and shares one id counter (basetracker.py:223-263; deepocsort.py:57,117-118). */
int boxmot_hip_deepocsort_update_stream(
    BoxMOTHipDeepOcSort* handle, int stream, int frame_count, int* id_count_inout,
    const float* dets, int det_rows, int det_cols,
    const float* embs, int emb_rows, int emb_cols,
    const uint8_t* image, int image_rows, int image_cols, int image_channels,
    float* out_tracks, int out_capacity_rows, int out_cols,
    int* out_rows, int* out_is_obb);
/* one frame for each of the first n_streams streams in one launch set (arguments as boxmot_hip_botsort_update_batch) */
int boxmot_hip_deepocsort_update_batch(
    BoxMOTHipDeepOcSort* handle, int n_streams,
    const float* const* dets, const int* det_rows,
    const float* const* embs, int emb_cols,
    const uint8_t* const* images, int image_rows, int image_cols, int image_channels,
    float* const* out_tracks, int out_capacity_rows, int* out_rows);
/* Device-resident step for all n_streams streams (asynchronous on the handle's stream; embeddings supplied):
 * d_dets [S][max_dets][6], d_det_rows [S], d_embs [S][max_dets][emb_dim], d_out [S][max_tracks][8], d_out_rows [S]. */
int boxmot_hip_deepocsort_step_device(BoxMOTHipDeepOcSort* handle, const float* d_dets, const int* d_det_rows,
                                       const float* d_embs, float* d_out, int* d_out_rows);
/* the same with the embeddings computed on the device from frames already resident in HBM (d_frames: one pointer per stream,
 * all image_rows x image_cols x 3 uint8 BGR): crop list -> ReID backbone of the loaded weights -> step; no host copies */
int boxmot_hip_deepocsort_step_device_frames(BoxMOTHipDeepOcSort* handle, const float* d_dets, const int* d_det_rows,
                                              const uint8_t* const* d_frames, int image_rows, int image_cols, float* d_out,
                                              int* d_out_rows);
/* accumulated device milliseconds of the ReID forward region since the last call (and the number of passes) */
int boxmot_hip_deepocsort_reid_kernel_ms(BoxMOTHipDeepOcSort* handle, double* out_ms, int* out_launches);
/* 0 = per-layer fp32 kernels, 1 = fp16 MFMA kernels (fused for OSNet-x0.25, layer-per-launch for osnet_x1_0); CLIP-ReID has one family */
int boxmot_hip_deepocsort_set_reid_mode(BoxMOTHipDeepOcSort* handle, int mode);
void* boxmot_hip_deepocsort_stream(BoxMOTHipDeepOcSort* handle);
/* waits for the handle's stream; throws if a step exceeded the bound below */
int boxmot_hip_deepocsort_synchronize(BoxMOTHipDeepOcSort* handle);
/* Pipelining of device-resident steps (all three trackers; BoT-SORT: when its streams occupy at most half of the device's CUs):
 * consecutive *_step_device / *_step_device_frames calls are asynchronous, and the ReID pass of call t + 1 runs on a second stream
 * of the handle while the frame step of call t is still on *_stream().  Rows of call t are complete after *_synchronize (or after
 * *_stream()'s work).  The inputs of a call (detections, counts, frames) must be complete when it is made -- or be ordered on
 * *_stream(): once that accessor has been called, every ReID pass waits for the work queued there (correct, but such a caller's
 * frames serialise).  BOXMOT_HIP_PIPELINE=0 / 1 (read at create) forces the pipeline off / on. */
/* Host-known upper bound on the ReID crops of the following step_device_frames calls (all streams together; -1 = none, the
 * default; a host call, cheap enough to repeat before every step with that step's detection total).  The backbone families that size their launches on the host (osnet_x0_5 ... x1_0, CLIP-ReID) otherwise read the
 * crop count back -- a stream synchronisation inside every step, during which the GPU waits for the host to queue the pass.
 * With a bound the crop list is filled up to it with copies of its first entry and nothing travels to the host; a step with
 * more crops than the bound is NOT stepped (no rows, tracker state untouched: it would have consumed stale embeddings) and is reported
 * by the next synchronize / host update -- step it again with a sufficient bound.  (OSNet-x0.25's fused kernels take the count on the device.) */
int boxmot_hip_deepocsort_set_crop_bound(BoxMOTHipDeepOcSort* handle, int max_total_crops);
/* parity debugging: live tracks of `stream` in list order -- ints5 (rows,5) = id, age, time_since_update, hit_streak,
 * observed; kf72 (rows,72) = x[8] ++ P[8][8] fp64 (index 7 unused); emb (rows, emb_dim) fp64.  NULL skips an output. */
int boxmot_hip_deepocsort_state_dump(BoxMOTHipDeepOcSort* handle, int stream, int* ints5, double* kf72, double* emb,
                                     int* out_rows, int* out_frame_count, int* out_id_count);

/* parity debugging, cost values: the matrices of the LAST update's `associate` call (boxmot/trackers/association/association.py:61-152)
 * in its own orientation, (detections, tracks) row-major fp64.  plane 0 = final_cost = -(iou_matrix + angle_diff_cost + emb_cost), what
 * linear_assignment was given; 1 = iou_matrix (the association function's return); 2 = the weighted emb_cost.  *out_branch: 0 = no matrix
 * (no detections), 1 = the already-a-permutation early-out (:104-108: planes 0 and 2 are not computed), 2 = the solver ran.  Off by default. */
int boxmot_hip_deepocsort_debug_costs_enable(BoxMOTHipDeepOcSort* handle, int on);
int boxmot_hip_deepocsort_debug_costs(BoxMOTHipDeepOcSort* handle, int stream, int plane, double* out, long out_capacity,
                                      int* out_rows, int* out_cols, int* out_branch);

/* ------------------------------------------------------------------------------------------------
 * StrongSORT (boxmot/trackers/bbox/strongsort/strongsort.py:16-126, sort/tracker.py, sort/track.py,
 * sort/linear_assignment.py).  No native backend exists in the reference; conventions as above.
 * Fields = the constructor arguments of StrongSort (strongsort.py:41-52) + BaseTracker.max_age.
 * ------------------------------------------------------------------------------------------------ */
typedef struct BoxMOTHipStrongSortConfig {
    int max_age;
    double min_conf;
    double max_cos_dist;
    double max_iou_dist;
    int n_init;
    int nn_budget;                   /* 1..1024 samples per track (None is not supported) */
    double mc_lambda;
    double ema_alpha;
    const char* reid_model_path;     /* OSN1 / CLP1 blob or NULL when embeddings are supplied */
    int n_streams;
    int max_tracks;
    int max_dets;
    int emb_dim;
} BoxMOTHipStrongSortConfig;

typedef struct BoxMOTHipStrongSort BoxMOTHipStrongSort;

void boxmot_hip_strongsort_default_config(BoxMOTHipStrongSortConfig* config);
BoxMOTHipStrongSort* boxmot_hip_strongsort_create(const BoxMOTHipStrongSortConfig* config);
void boxmot_hip_strongsort_destroy(BoxMOTHipStrongSort* handle);
int boxmot_hip_strongsort_reset(BoxMOTHipStrongSort* handle);
/* capacities grow on demand in the host-API updates (see boxmot_hip_botsort_reserve) */
int boxmot_hip_strongsort_reserve(BoxMOTHipStrongSort* handle, int max_tracks, int max_dets);
int boxmot_hip_strongsort_capacity(BoxMOTHipStrongSort* handle, int* max_tracks, int* max_dets, int* n_grows);
/* Track.camera_update (sort/track.py:139-148): the 2x3 warp the reference's cmc.apply returned, applied to every track in
 * the NEXT update of `stream` (the reference applies its ECC estimate unconditionally, strongsort.py:83-86); without a
 * pending warp the identity is applied, which is what the reference computes for a static camera. */
int boxmot_hip_strongsort_set_warp(BoxMOTHipStrongSort* handle, int stream, const double* warp_2x3);
/* StrongSort.update for stream 0 (strongsort.py:69-123); embs == NULL runs the ReID engine on every detection with
 * conf >= min_conf.  Rows [x1,y1,x2,y2,id,conf,cls,det_ind,0]. */
int boxmot_hip_strongsort_update(
    BoxMOTHipStrongSort* handle,
    const float* dets, int det_rows, int det_cols,
    const float* embs, int emb_rows, int emb_cols,
    const uint8_t* image, int image_rows, int image_cols, int image_channels,
    float* out_tracks, int out_capacity_rows, int out_cols,
    int* out_rows, int* out_is_obb);
int boxmot_hip_strongsort_update_batch(
    BoxMOTHipStrongSort* handle, int n_streams,
    const float* const* dets, const int* det_rows,
    const float* const* embs, int emb_cols,
    const uint8_t* const* images, int image_rows, int image_cols, int image_channels,
    float* const* out_tracks, int out_capacity_rows, int* out_rows);
/* Device-resident step for all n_streams streams (asynchronous; embeddings supplied, identity camera motion):
 * d_dets [S][max_dets][6], d_det_rows [S], d_embs [S][max_dets][emb_dim], d_out [S][max_tracks][8], d_out_rows [S]. */
int boxmot_hip_strongsort_step_device(BoxMOTHipStrongSort* handle, const float* d_dets, const int* d_det_rows,
                                       const float* d_embs, float* d_out, int* d_out_rows);
int boxmot_hip_strongsort_step_device_frames(BoxMOTHipStrongSort* handle, const float* d_dets, const int* d_det_rows,
                                              const uint8_t* const* d_frames, int image_rows, int image_cols, float* d_out,
                                              int* d_out_rows);
int boxmot_hip_strongsort_reid_kernel_ms(BoxMOTHipStrongSort* handle, double* out_ms, int* out_launches);
int boxmot_hip_strongsort_set_reid_mode(BoxMOTHipStrongSort* handle, int mode);
void* boxmot_hip_strongsort_stream(BoxMOTHipStrongSort* handle);
int boxmot_hip_strongsort_synchronize(BoxMOTHipStrongSort* handle);
/* as boxmot_hip_deepocsort_set_crop_bound */
int boxmot_hip_strongsort_set_crop_bound(BoxMOTHipStrongSort* handle, int max_total_crops);
/* len(self.tracker.tracks) of `stream` (tentative + confirmed): the reference asks its camera-motion estimator for a warp only
 * when this is >= 1 (strongsort.py:83-86), and that estimator is stateful -- a caller that owns one needs the same gate. */
int boxmot_hip_strongsort_track_count(BoxMOTHipStrongSort* handle, int stream, int* out_tracks);
/* parity debugging: tracks of `stream` in list order -- ints6 (rows,6) = id, state (1 tentative / 2 confirmed), hits, age,
 * time_since_update, sample-bank size; kf72 (rows,72) = mean[8] ++ cov[8][8]; feat (rows, emb_dim) fp32. */
int boxmot_hip_strongsort_state_dump(BoxMOTHipStrongSort* handle, int stream, int* ints6, double* kf72, float* feat,
                                     int* out_rows, int* out_frame_count, int* out_next_id);

/* parity debugging, cost values: the matrices of the LAST update's two min_cost_matching calls (sort/linear_assignment.py:14-79),
 * (tracks, detections) row-major fp64.  stage 0 = confirmed tracks x detections under the gated appearance metric (sort/tracker.py:108-122,
 * gate_cost_matrix linear_assignment.py:145-198); stage 1 = the IoU stage (sort/iou_matching.py:49-87).  plane 0 = the metric's matrix as
 * returned, plane 1 = after `cost > max_distance -> max_distance + 1e-5`, what linear_sum_assignment is given.  Off by default. */
int boxmot_hip_strongsort_debug_costs_enable(BoxMOTHipStrongSort* handle, int on);
int boxmot_hip_strongsort_debug_costs(BoxMOTHipStrongSort* handle, int stream, int stage, int plane, double* out, long out_capacity,
                                      int* out_rows, int* out_cols);

#ifdef __cplusplus
}
#endif
#endif /* BOXMOT_HIP_H */
