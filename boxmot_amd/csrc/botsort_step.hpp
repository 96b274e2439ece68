// BoT-SORT frame step: ONE workgroup advances ONE stream by ONE frame, entirely
// on the device (no host round trip between detections in and rows out).
//
// Reference semantics restated phase by phase (AABB layout, CMC off):
//   BotSort._update_impl                 boxmot/trackers/bbox/botsort/botsort.py:177-249
//   _split_detections / _create_detections             botsort.py:251-274
//   STrack.multi_predict + KalmanFilterXYWH.multi_predict
//                                         botsort_track.py:96-115, kalman_filters/base.py:311-327, xywh.py:149-160
//   iou_distance / embedding_distance / fuse_score / gating
//                                         association/matching.py:46-147, association/iou.py:133-150, botsort.py:306-317
//   linear_assignment (lap.lapjv, extend_cost + cost_limit)   matching.py:28-43
//   STrack.update / re_activate / activate + KalmanFilterXYWH.update / initiate
//                                         botsort_track.py:232-282, base.py:286-355, xywh.py:136-186
//   second association / unconfirmed / births / ageing   botsort.py:335-476
//   joint / sub / remove_duplicate_stracks + output rows botsort_utils.py:10-82, botsort.py:478-500
//
// Numeric policy: Kalman state, IoU and cost matrices in fp64 with the oracle's
// operation order (compiled with -ffp-contract=off); detection geometry and
// appearance vectors in fp32 exactly where the reference uses fp32.
#pragma once

#include "block_prims.hpp"
#include "botsort_types.hpp"
#include "kernel_macros.hpp"
#include "obb_geometry.hpp"

// phase functions of the frame step: inlined by default.  -DBM_STEP_NOINLINE=1 compiles the three large ones that are instantiated
// several times (cost build, IoU cost, assignment) as calls -- one copy of each, bounded live ranges; =2 the small per-track ones too
#ifndef BM_STEP_NOINLINE
#define BM_STEP_NOINLINE 0
#endif
#if BM_STEP_NOINLINE >= 1
#define BM_STEP_BIG_FN __device__ __noinline__
#else
#define BM_STEP_BIG_FN __device__ inline
#endif
#if BM_STEP_NOINLINE >= 2
#define BM_STEP_FN __device__ __noinline__
#else
#define BM_STEP_FN __device__ inline
#endif

namespace bm {
constexpr int BOX_W = 4;          // floats of a detection's box (x, y, w, h)
constexpr int CONF_COL = 4;       // confidence column of a detection row; the class follows it
#define BM_OBB 0
#include "botsort_step_body.hpp"
#undef BM_OBB
// The same frame step for ORIENTED detections (cx, cy, w, h, theta, conf, cls): BotSort / ByteTrack with is_obb
// (botsort.py:105, 267-271, 306, 357, 396, 495; bytetrack.py:266, 286, 303, 335, 355, 397) -- a second copy of the body with the
// oriented layout and `#if BM_OBB` where the reference branches: the 10-state filter, the rotated IoU, the 9-column rows.
namespace obb {
constexpr int KF_DIM = 10;        // state (cx, cy, w, h, theta) + velocities
constexpr int KF_STRIDE = 110;    // 10 mean + 100 covariance doubles per track
constexpr int DET_COLS = 7;
constexpr int OUT_COLS = 9;
constexpr int BOX_W = 5;
constexpr int CONF_COL = 5;
#define BM_OBB 1
#include "botsort_step_body.hpp"
#undef BM_OBB
}  // namespace obb
}  // namespace bm
