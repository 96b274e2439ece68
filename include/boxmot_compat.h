/*
 * boxmot_compat.h -- the reference's own native FFI names, exported by libboxmot_hip.so.
 *
 * Every declaration below is the reference's, symbol for symbol and field for field, so that the reference's ctypes
 * bindings (boxmot/native/trackers/{botsort,bytetrack,ocsort}.py, boxmot/native/reid/capi.py) load this library
 * unchanged -- only the library path differs:
 *   boxmot_botsort_*    boxmot/native/cpp/trackers/botsort/include/botsort/c_api.hpp:17-61
 *   boxmot_bytetrack_*  boxmot/native/cpp/trackers/bytetrack/include/bytetrack/c_api.hpp:16-46
 *   boxmot_ocsort_*     boxmot/native/cpp/trackers/ocsort/include/ocsort/c_api.hpp:16-52
 *   boxmot_reid_capi_*  boxmot/native/cpp/trackers/base/include/boxmot/trackers/base/reid_capi.h:36-94
 * They are adapters over the boxmot_hip_* entry points of boxmot_hip.h (which add what the reference ABI cannot
 * express: fp64 thresholds, the three Python-only association knobs, several streams per handle, device-resident steps).
 *
 * Behaviour that differs from the reference library, all loud:
 *   - `reid_model_path` / `model_path` name an OSN1 weight blob (boxmot_amd.reid_weights.save_blob), not an ONNX file;
 *   - `cmc_method`: "sof" (the YAML default) and "ecc" run that estimator of the reference on the device inside update; NULL, "" or
 *     "none" estimate nothing (a caller that has a warp supplies it per frame through boxmot_hip_botsort_set_warp, boxmot_hip.h);
 *     "orb" / "sift" are not built: create fails loudly.  Both estimators restate OpenCV algorithms (cv2 is absent offline):
 *     their numerics are pinned against this repository's restatements only;
 *   - the device tables start at BOXMOT_HIP_MAX_TRACKS (default 1024) live + lost tracks and BOXMOT_HIP_MAX_DETS (default 512)
 *     detections per frame and grow when a frame does not fit (boxmot_hip_botsort_reserve, boxmot_hip.h);
 *     BOXMOT_HIP_REID_MAX_CROPS (default 1024) boxes per standalone ReID call is a limit: exceeding it fails the call with a
 *     message, it never truncates silently;
 *   - the track-id counter is per handle (the reference's is process-global, botsort/src/track.cpp:13);
 *   - there is no CPU fallback: without a HIP device `create` returns NULL / 0 with an error message.
 */
#ifndef BOXMOT_COMPAT_H
#define BOXMOT_COMPAT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- BoT-SORT: c_api.hpp:17-61 ---- */
struct BoxMOTBotSortConfig {
    float track_high_thresh;
    float track_low_thresh;
    float new_track_thresh;
    int track_buffer;
    float match_thresh;
    float proximity_thresh;
    float appearance_thresh;
    const char* cmc_method;
    int frame_rate;
    int fuse_first_associate;
    int with_reid;
    int max_obs;
    const char* reid_model_path;
    const char* reid_preprocess;
};
struct BoxMOTBotSortHandle;

struct BoxMOTBotSortHandle* boxmot_botsort_create(const struct BoxMOTBotSortConfig* config);
void boxmot_botsort_destroy(struct BoxMOTBotSortHandle* handle);
int boxmot_botsort_reset(struct BoxMOTBotSortHandle* handle);
int boxmot_botsort_update(
    struct BoxMOTBotSortHandle* handle,
    const float* dets, int det_rows, int det_cols,
    const float* embs, int emb_rows, int emb_cols,
    const uint8_t* image_data, int image_rows, int image_cols, int image_channels,
    float* out_tracks, int out_capacity_rows, int out_cols,
    int* out_rows, int* out_is_obb);
int boxmot_botsort_last_reid_time_ms(struct BoxMOTBotSortHandle* handle, double* out_reid_time_ms);
int boxmot_botsort_last_reid_preprocess_time_ms(struct BoxMOTBotSortHandle* handle, double* out_time_ms);
int boxmot_botsort_last_reid_process_time_ms(struct BoxMOTBotSortHandle* handle, double* out_time_ms);
int boxmot_botsort_last_reid_postprocess_time_ms(struct BoxMOTBotSortHandle* handle, double* out_time_ms);
const char* boxmot_botsort_last_error(void);

/* ---- ByteTrack: bytetrack/c_api.hpp:16-46 ---- */
struct BoxMOTByteTrackConfig {
    float min_conf;
    float track_thresh;
    float match_thresh;
    int track_buffer;
    int frame_rate;
    int max_obs;
};
struct BoxMOTByteTrackHandle;

struct BoxMOTByteTrackHandle* boxmot_bytetrack_create(const struct BoxMOTByteTrackConfig* config);
void boxmot_bytetrack_destroy(struct BoxMOTByteTrackHandle* handle);
int boxmot_bytetrack_reset(struct BoxMOTByteTrackHandle* handle);
int boxmot_bytetrack_update(
    struct BoxMOTByteTrackHandle* handle,
    const float* dets, int det_rows, int det_cols,
    const uint8_t* image_data, int image_rows, int image_cols, int image_channels,
    float* out_tracks, int out_capacity_rows, int out_cols,
    int* out_rows, int* out_is_obb);
const char* boxmot_bytetrack_last_error(void);

/* ---- OC-SORT: ocsort/c_api.hpp:16-52 ---- */
struct BoxMOTOCSORTConfig {
    float min_conf;
    float det_thresh;
    float iou_threshold;
    int max_age;
    int min_hits;
    int delta_t;
    int use_byte;
    float inertia;
    float q_xy_scaling;
    float q_s_scaling;
    int max_obs;
};
struct BoxMOTOCSORTHandle;

struct BoxMOTOCSORTHandle* boxmot_ocsort_create(const struct BoxMOTOCSORTConfig* config);
void boxmot_ocsort_destroy(struct BoxMOTOCSORTHandle* handle);
int boxmot_ocsort_reset(struct BoxMOTOCSORTHandle* handle);
int boxmot_ocsort_update(
    struct BoxMOTOCSORTHandle* handle,
    const float* dets, int det_rows, int det_cols,
    const uint8_t* image_data, int image_rows, int image_cols, int image_channels,
    float* out_tracks, int out_capacity_rows, int out_cols,
    int* out_rows, int* out_is_obb);
const char* boxmot_ocsort_last_error(void);

/* ---- ReID: reid_capi.h:36-94 ---- */
int boxmot_reid_capi_create(const char* model_path, const char* preprocess, void** out_handle);
void boxmot_reid_capi_destroy(void* handle);
int boxmot_reid_capi_feature_dim(void* handle, int* out_feature_dim);
int boxmot_reid_capi_compute_features(
    void* handle,
    const float* boxes_xyxy, int n_boxes,
    const uint8_t* image_data, int image_rows, int image_cols, int image_channels,
    float* out_features, int out_capacity_floats);
/* staged: preprocess -> process -> postprocess, in this order on one handle (reid_capi.h:61-89) */
int boxmot_reid_capi_preprocess(
    void* handle,
    const float* boxes_xyxy, int n_boxes,
    const uint8_t* image_data, int image_rows, int image_cols, int image_channels);
int boxmot_reid_capi_process(void* handle);
int boxmot_reid_capi_postprocess(void* handle, float* out_features, int out_capacity_floats);
const char* boxmot_reid_capi_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* BOXMOT_COMPAT_H */
