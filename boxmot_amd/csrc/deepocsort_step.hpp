// DeepOCSORT frame step for one stream (one workgroup), gfx950 / wave64.
//
// Reference path (Python, boxmot 21.0.0):
//   DeepOcSort._update_impl            boxmot/trackers/bbox/deepocsort/deepocsort.py:302-492
//   KalmanBoxTracker                   deepocsort.py:51-233
//   KalmanFilterXYSR predict/update/freeze/unfreeze   motion/kalman_filters/xysr.py:368-476
//     over predict_state / update_state (Joseph form) motion/kalman_filters/base.py:366-459, :461-500
//   associate / compute_aw_max_metric / speed_direction_batch   trackers/association/association.py:8-152
//   iou_batch                          trackers/association/iou.py:134-150
//   lap.lapjv(cost, extend_cost=True)  association.py:20-24 (no cost limit: a full assignment of the smaller side)
//
// Data layout: like the BoT-SORT step (botsort_types.hpp) every array is stream-major and a track lives in a
// slot; `list` holds the slots in the order of the reference's `active_tracks` Python list (ids of new tracks,
// the row order of the output and the order of removals all follow it).  The 7-state filter is kept as
// x[8] ++ P[8][8] fp64 (row/column 7 unused, zero) so that lane l of a wavefront owns P[l>>3][l&7]; the whole
// predict / update / observation-centric re-update chain of one track runs in the registers of one wavefront
// (neighbouring elements by shuffles), fp64, contraction off.
//
// The step body (deepocsort_step_body.hpp) is compiled twice, like the BoT-SORT one: in namespace bm for axis-aligned boxes and in
// namespace bm::obb for ORIENTED detections -- OC-SORT with is_obb (ocsort.py:49-87, :121-154, :241-310, :363-555): a 9-state
// KalmanFilterXYSR (x, y, s, r, theta + velocities), the rotated IoU, 7-column detections and 9-column rows; `#if BM_OBB` where the
// reference branches.  The tables below are shared; their per-track widths follow DocsSizes::is_obb.
#pragma once

#include "block_prims.hpp"
#include "lap_jv.hpp"
#include "botsort_types.hpp"
#include "kernel_macros.hpp"
#include "obb_geometry.hpp"

namespace bm {

struct DocsConfigDev {
    float det_thresh_f32;             // `scores > det_thresh` compares fp32 with a weak Python float (NEP 50)
    double det_thresh;
    int max_age, min_hits, delta_t;
    double iou_threshold, inertia, w_emb, alpha_fixed, aw_param, q_xy, q_s;
    int embedding_off, aw_off;
    // OC-SORT's BYTE branch (ocsort.py:393-399, 456-485): detections with min_conf < score < det_thresh are associated,
    // after the first round, with the predicted boxes of the still unmatched tracks
    int use_byte;
    float min_conf_f32;
    // association function (BaseTracker's asso_func, trackers/association/iou.py:408-417): 0 iou, 1 giou, 2 diou, 3 ciou, 4 hmiou,
    // 5 centroid; asso_diag = sqrt(w^2 + h^2) of the frame (centroid's norm_factor, iou.py:266; basetracker.py:175-180)
    int asso_mode;
    double asso_diag;
};

struct DocsState {
    int cap, dim;
    int* frame_count; int* id_count; int* n_tracks; int* status;       // [S]
    int* list;            // [S][cap] ordered slots
    int* slot_used;       // [S][cap]
    double* kf;           // [S][cap][72]   x[8] ++ P[8][8]          (oriented: [90] x[9] ++ P[9][9])
    double* kf_saved;     // [S][cap][72]   filter state at freeze() (first missed frame)
    double* last_obs;     // [S][cap][5]    last_observation (placeholder -1 x5)          (oriented: [6])
    double* last_z;       // [S][cap][4]    the last measurement as the filter stored it (history_obs; never warped)   (oriented: [5])
    double* obs_box;      // [S][cap][3][5] the three most recent observations, oldest first      (oriented: [3][6])
    int* obs_age;         // [S][cap][3]
    int* n_obs;           // [S][cap]
    double* velocity;     // [S][cap][2]    (dy, dx) unit vector
    int* has_vel;
    double* emb;          // [S][cap][dim]
    int* id; int* age; int* tsu; int* hits; int* hit_streak; int* observed; int* has_saved; int* n_miss;
    float* conf; float* cls; float* det_ind;
};

struct DocsScratch {
    int max_dets;
    int* keep;            // [S][nd]   indices of detections with score > det_thresh
    double* alpha;        // [S][nd]   dets_alpha per kept detection
    double* trk_box;      // [S][cap][4]  predicted boxes (list order)               (oriented: [5])
    double* kobs;         // [S][cap][5]  k_previous_obs per track                   (oriented: [6])
    double* iou;          // [S][nd][cap]  det-major
    double* cost;         // [S][nd][cap]
    double* embc;         // [S][nd][cap]
    double* row_w;        // [S][nd]
    double* col_w;        // [S][cap]
    int* row_cnt;         // [S][nd]
    int* col_cnt;         // [S][cap]
    int* m_det; int* m_trk;      // [S][nd]  matched pairs (kept-det index, list position)
    int* un_d; int* un_t;        // [S][nd], [S][cap]
    int* tmp_a; int* tmp_b;      // [S][max(cap,nd)]
    int* lap_x; int* lap_y;      // [S][cap + nd], [S][nd]
    int* flag_t; int* flag_d;    // [S][cap], [S][nd]
};

struct DocsStepArgs {
    DocsConfigDev cfg;
    DocsState st;
    DocsScratch sc;
    const float* dets;        // [S][nd][6]                                       (oriented: [7])
    const int* n_dets;        // [S]
    const float* embs;        // [S][nd][dim] or nullptr
    const double* warp;       // [S][6] camera-motion warp (2x3 row-major) applied before the prediction, or nullptr
    const int* warp_flag;     // [S] non-zero: apply warp[s] in this step
    float* out;               // [S][cap][8]  (a frame can output more rows than detections: new + matched)   (oriented: [9])
    int* out_n;               // [S]
    int stream_base;
    // Parity debugging (boxmot_hip_deepocsort_debug_costs; nullptr = off, the default): the matrices of this step's `associate`
    // (association.py:61-152), [S][DOCS_DBG_PLANES][max_dets][cap] fp64 detection-major (plane[k * cap + t]) and
    // [S][4] = (n detections, n tracks, branch: 0 no matrix / 1 already-a-permutation early-out / 2 solver, 0).
    // plane 0 = final_cost (branch 2 only), 1 = the association function's matrix (iou_matrix), 2 = the weighted emb_cost (branch 2).
    double* dbg_cost;
    int* dbg_shape;
};
constexpr int DOCS_DBG_PLANES = 3;

struct DocsSizes { int S, cap, nd, dim; int is_obb = 0; };
__host__ __device__ inline int docs_kf_stride(int is_obb) { return is_obb ? 90 : KF_STRIDE; }

template <class A>
void docs_allocate(DocsStepArgs& args, const DocsSizes& z, A& a) {
    const size_t S = z.S, cap = z.cap, nd = z.nd, dim = z.dim, big = cap > nd ? cap : nd;
    const size_t kfs = docs_kf_stride(z.is_obb), box = z.is_obb ? 5 : 4, obs = box + 1;      // filter state, box, box + score
    DocsState& st = args.st;
    st.cap = z.cap; st.dim = z.dim;
    st.frame_count = a.template get<int>(S); st.id_count = a.template get<int>(S);
    st.n_tracks = a.template get<int>(S); st.status = a.template get<int>(S);
    st.list = a.template get<int>(S * cap); st.slot_used = a.template get<int>(S * cap);
    st.kf = a.template get<double>(S * cap * kfs); st.kf_saved = a.template get<double>(S * cap * kfs);
    st.last_obs = a.template get<double>(S * cap * obs); st.last_z = a.template get<double>(S * cap * box); st.obs_box = a.template get<double>(S * cap * 3 * obs);
    st.obs_age = a.template get<int>(S * cap * 3); st.n_obs = a.template get<int>(S * cap);
    st.velocity = a.template get<double>(S * cap * 2); st.has_vel = a.template get<int>(S * cap);
    st.emb = a.template get<double>(S * cap * dim);
    st.id = a.template get<int>(S * cap); st.age = a.template get<int>(S * cap); st.tsu = a.template get<int>(S * cap);
    st.hits = a.template get<int>(S * cap); st.hit_streak = a.template get<int>(S * cap);
    st.observed = a.template get<int>(S * cap); st.has_saved = a.template get<int>(S * cap);
    st.n_miss = a.template get<int>(S * cap);
    st.conf = a.template get<float>(S * cap); st.cls = a.template get<float>(S * cap); st.det_ind = a.template get<float>(S * cap);
    DocsScratch& sc = args.sc;
    sc.max_dets = z.nd;
    sc.keep = a.template get<int>(S * nd); sc.alpha = a.template get<double>(S * nd);
    sc.trk_box = a.template get<double>(S * cap * box); sc.kobs = a.template get<double>(S * cap * obs);
    sc.iou = a.template get<double>(S * nd * cap); sc.cost = a.template get<double>(S * nd * cap);
    sc.embc = a.template get<double>(S * nd * cap);
    sc.row_w = a.template get<double>(S * nd); sc.col_w = a.template get<double>(S * cap);
    sc.row_cnt = a.template get<int>(S * nd); sc.col_cnt = a.template get<int>(S * cap);
    sc.m_det = a.template get<int>(S * nd); sc.m_trk = a.template get<int>(S * nd);
    sc.un_d = a.template get<int>(S * nd); sc.un_t = a.template get<int>(S * cap);
    sc.tmp_a = a.template get<int>(S * big); sc.tmp_b = a.template get<int>(S * big);
    sc.lap_x = a.template get<int>(S * (cap + nd)); sc.lap_y = a.template get<int>(S * nd);
    sc.flag_t = a.template get<int>(S * cap); sc.flag_d = a.template get<int>(S * nd);
}

constexpr double DOCS_INF = 1e300;
__host__ __device__ inline long docs_lap_lds_bytes(int cap, int nd) { return jv_lds_bytes(cap + nd); }

// per-track widths of the axis-aligned layout
constexpr int DOCS_KF_STRIDE = KF_STRIDE;     // 72: x[8] ++ P[8][8]
constexpr int DOCS_BOX = 4;                   // doubles of a box; the score follows it in an observation, conf / cls in a detection row
constexpr int DOCS_OBS = 5;
constexpr int DOCS_DET_COLS = DET_COLS;
constexpr int DOCS_OUT_COLS = OUT_COLS;
#define BM_OBB 0
#include "deepocsort_step_body.hpp"
#undef BM_OBB
namespace obb {
constexpr int DOCS_KF_STRIDE = 90;            // x[9] ++ P[9][9]
constexpr int DOCS_BOX = 5;                   // (cx, cy, w, h, theta)
constexpr int DOCS_OBS = 6;
constexpr int DOCS_DET_COLS = 7;
constexpr int DOCS_OUT_COLS = 9;
#define BM_OBB 1
#include "deepocsort_step_body.hpp"
#undef BM_OBB
}  // namespace obb

}  // namespace bm
