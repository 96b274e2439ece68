// Device data layout of the multi-stream BoT-SORT tracker (one "stream" = one
// camera = one reference BotSort instance; boxmot/trackers/bbox/botsort/botsort.py).
//
// HBM layout (DESIGN.md "Data layout"): every array is stream-major,
// [n_streams][...], so one workgroup owns one contiguous slab per array.
//   track record ("slot") s of a stream:
//     kf[s]      : 72 fp64 = mean[8] ++ cov[8][8]  (576 B, AoS so that a
//                  64-lane wavefront reads/writes one track with one coalesced
//                  512-B + 64-B access; lane l <-> cov element (l>>3, l&7))
//     smooth[s]  : dim fp32 (EMA appearance vector, botsort_track.py:58-67)
//     scalars    : id, state, is_activated, frame_id, start_frame, tracklet_len,
//                  conf/cls/det_ind (fp32, copied from the matched detection)
//     class vote : up to KCLS (cls, summed conf) pairs (botsort_track.py:69-82)
//   ordered index lists (slot ids): active[n_lists][cap] (one per class when
//   per_class=True, basetracker.py:223-263), lost[cap]; removed ids live in a
//   ring with deque(maxlen) semantics (botsort.py:93-95).
#pragma once

namespace bm {

constexpr int KF_DIM = 8;
constexpr int KF_STRIDE = 72;      // 8 mean + 64 cov doubles per track
constexpr int KCLS = 8;            // distinct classes remembered per track
constexpr int DET_COLS = 6;        // x1,y1,x2,y2,conf,cls (detection_layout.py:61-71)
constexpr int OUT_COLS = 8;        // x1,y1,x2,y2,id,conf,cls,det_ind (track_results.py:12-31)

// TrackState, basetrack.py:18-22
constexpr int ST_NEW = 0, ST_TRACKED = 1, ST_LOST = 2, ST_LONGLOST = 3, ST_REMOVED = 4;

// per-stream status word written by the step kernel (0 = ok)
constexpr int STATUS_OK = 0;
constexpr int STATUS_TRACK_CAPACITY = 1;   // more live tracks than `cap` slots
constexpr int STATUS_CLASS_CAPACITY = 2;   // a track saw more than KCLS classes
constexpr int STATUS_LAP_STALL = 3;        // assignment loop hit its iteration bound

struct BotSortConfigDev {
    double track_high_thresh, track_low_thresh, new_track_thresh;
    double match_thresh, proximity_thresh, appearance_thresh;
    double second_match_thresh, unconfirmed_match_thresh, unconfirmed_emb_scale;
    float new_track_thresh_f32;   // np.float32 < python-float compares in fp32 (NEP 50)
    int fuse_first_associate, with_reid;
    int max_time_lost;            // int(frame_rate / 30.0 * track_buffer), botsort.py:103-104
    int removed_cap;              // removed_stracks_buffer (deque maxlen)
    // 0 = BoT-SORT; 1 = ByteTrack (bytetrack.py:258-408): the same stages with an (x, y, aspect, height) filter state
    // (kalman_filters/xyah.py), only vh zeroed for non-tracked tracks, score fusion in the first and the unconfirmed
    // association, no class vote, an unbounded removed list (a per-slot flag, kept in `hist_n`)
    int kind;
};

// Persistent tracker state, all pointers device memory, indexed [stream][...].
struct BotSortState {
    int cap, dim, n_lists, removed_alloc;
    int* frame_count;    // [S]
    int* id_count;       // [S]   per-stream BaseTrack._count
    int* n_active;       // [S][n_lists]
    int* n_lost;         // [S]
    int* rm_head;        // [S]
    int* rm_size;        // [S]
    int* stamp;          // [S]   running mark value
    int* status;         // [S]
    int* active_list;    // [S][n_lists][cap]
    int* lost_list;      // [S][cap]
    int* removed_ring;   // [S][removed_alloc]
    double* kf;          // [S][cap][72]
    float* smooth;       // [S][cap][dim]
    int* id;             // [S][cap]
    int* state;
    int* is_activated;
    int* frame_id;
    int* start_frame;
    int* tracklet_len;
    int* slot_used;
    int* mark;           // scratch marks (stamp based)
    float* conf;
    float* cls;
    float* det_ind;
    int* hist_n;         // [S][cap]
    float* hist_cls;     // [S][cap][KCLS]
    float* hist_w;       // [S][cap][KCLS]
};

// Per-stream scratch (global memory, L2 resident), indexed [stream][...].
struct BotSortScratch {
    int max_dets;
    float* det_xywh;     // [S][nd][4]
    float* det_xyxy;     // [S][nd][4]
    float* det_area;     // [S][nd]
    float* det_feat;     // [S][nd][dim]   normalised detection features
    double* det_norm;    // [S][nd]
    double* trk_norm;    // [S][cap]
    int* first_idx;      // [S][nd]
    int* second_idx;     // [S][nd]
    int* left_idx;       // [S][nd]
    int* pool;           // [S][cap]
    int* unconf;         // [S][cap]
    int* remain;         // [S][cap]
    int* list_a;         // [S][cap]  bookkeeping temporaries
    int* list_b;         // [S][cap]
    int* activated;      // [S][cap]
    int* refound;        // [S][cap]
    int* newly_lost;     // [S][cap]
    int* newly_removed;  // [S][cap]
    int* match_slot;     // [S][nd]
    int* match_det;      // [S][nd]
    int* match_flag;     // [S][nd]
    int* drop_a;         // [S][cap]
    int* drop_b;         // [S][cap]
    double* cost;        // [S][nd][cap]  detection-major: cost[c * cap + r]
    int* lap_x;          // [S][cap]   column of row (or -1)
    int* lap_y;          // [S][nd]    row of column (or -1)
    double* lap_u;       // [S][nd]
    double* lap_v;       // [S][cap]
    double* lap_minv;    // [S][cap]
    int* lap_way;        // [S][cap]
    int* lap_used;       // [S][cap]
    double* box_a;       // [S][cap][4]  fp64 xyxy of list rows
    int* pair_list;      // [S][4096]  (row, col) pairs that pass the IoU gate (sparse cosine evaluation)
};

struct BotSortStepArgs {
    BotSortConfigDev cfg;
    BotSortState st;
    BotSortScratch sc;
    const float* dets;        // [S][max_dets][6] fp32
    const int* n_dets;        // [S]
    const float* embs;        // [S][max_dets][dim] raw per-detection features, or nullptr
    const int* list_sel;      // [S] active-list (class) selector, or nullptr (= list 0)
    const int* frame_count_set;  // [S] value to set before the step (per_class), or nullptr
    float* out;               // [S][max_dets][8]
    int* out_n;               // [S]
    const double* warp;       // [S][6] camera-motion warp (2x3 row-major) applied after prediction, or nullptr
    const int* warp_flag;     // [S] non-zero: apply warp[s] in this step
    int stream_base;          // workgroup b advances stream (stream_base + b)
    long long* phase_clock;   // optional [16] shader-clock stamps of stream stream_base's phases (profiling), or nullptr
    // Parity debugging (boxmot_hip_botsort_debug_costs; nullptr = off, the default): copies of the three association cost
    // matrices of this step, [S][DBG_STAGES][DBG_PLANES][max_dets][cap] fp64, detection-major like `cost` (plane[c * cap + r]),
    // and their shapes [S][DBG_STAGES][2] = (n_rows tracks, n_cols detections).  Nothing reads them on the device.
    double* dbg_cost;
    int* dbg_shape;
};

// stages: the first association (botsort.py:285-333), the second, IoU-only one (:335-378), the unconfirmed tracks (:380-431)
constexpr int DBG_STAGES = 3;
// planes: 0 = the matrix the assignment solver was given (matching.py:46-107, 139-147 + the gates of botsort.py:306-317, 396-413);
//         1 = iou_distance before score fusion; 2 = embedding_distance (cosine, clipped at 0) where the step evaluated it, NaN elsewhere
//             (the sparse path evaluates the pairs that pass the IoU gate, the dense fallback every pair)
constexpr int DBG_PLANES = 3;

}  // namespace bm
