// Allocation of the tracker's persistent state and per-stream scratch
// (layout: botsort_types.hpp).  The allocator is a policy object with
// `template <class T> T* get(size_t n)` returning zero-initialised storage, so
// the same sizing logic serves the device arena (boxmot_hip.hip) and the
// host-memory harness used to run the kernels under sanitizers in tests/.
#pragma once

#include <cstddef>

#include "botsort_types.hpp"

namespace bm {

// is_obb: the tables of the oriented-box step (bm::obb): a 10-state filter (110 doubles per track), 5-float boxes
struct BotSortSizes { int S, cap, nd, dim, n_lists, removed_alloc; int is_obb = 0; };

template <class A>
void botsort_allocate(BotSortStepArgs& args, const BotSortSizes& z, A& a) {
    const size_t S = z.S, cap = z.cap, nd = z.nd, dim = z.dim, nl = z.n_lists;
    const size_t kf_stride = z.is_obb ? 110 : KF_STRIDE, box_w = z.is_obb ? 5 : 4;
    BotSortState& st = args.st;
    st.cap = z.cap; st.dim = z.dim; st.n_lists = z.n_lists; st.removed_alloc = z.removed_alloc;
    st.frame_count = a.template get<int>(S); st.id_count = a.template get<int>(S);
    st.n_active = a.template get<int>(S * nl); st.n_lost = a.template get<int>(S);
    st.rm_head = a.template get<int>(S); st.rm_size = a.template get<int>(S);
    st.stamp = a.template get<int>(S); st.status = a.template get<int>(S);
    st.active_list = a.template get<int>(S * nl * cap); st.lost_list = a.template get<int>(S * cap);
    st.removed_ring = a.template get<int>(S * z.removed_alloc);
    st.kf = a.template get<double>(S * cap * kf_stride);
    st.smooth = a.template get<float>(S * cap * dim);
    st.id = a.template get<int>(S * cap); st.state = a.template get<int>(S * cap);
    st.is_activated = a.template get<int>(S * cap); st.frame_id = a.template get<int>(S * cap);
    st.start_frame = a.template get<int>(S * cap); st.tracklet_len = a.template get<int>(S * cap);
    st.slot_used = a.template get<int>(S * cap); st.mark = a.template get<int>(S * cap);
    st.conf = a.template get<float>(S * cap); st.cls = a.template get<float>(S * cap);
    st.det_ind = a.template get<float>(S * cap);
    st.hist_n = a.template get<int>(S * cap);
    st.hist_cls = a.template get<float>(S * cap * KCLS); st.hist_w = a.template get<float>(S * cap * KCLS);
    BotSortScratch& sc = args.sc;
    sc.max_dets = z.nd;
    sc.det_xywh = a.template get<float>(S * nd * box_w); sc.det_xyxy = a.template get<float>(S * nd * 4);
    sc.det_area = a.template get<float>(S * nd); sc.det_feat = a.template get<float>(S * nd * dim);
    sc.det_norm = a.template get<double>(S * nd); sc.trk_norm = a.template get<double>(S * cap);
    sc.first_idx = a.template get<int>(S * nd); sc.second_idx = a.template get<int>(S * nd);
    sc.left_idx = a.template get<int>(S * nd);
    sc.pool = a.template get<int>(S * cap); sc.unconf = a.template get<int>(S * cap);
    sc.remain = a.template get<int>(S * cap);
    sc.list_a = a.template get<int>(S * cap); sc.list_b = a.template get<int>(S * cap);
    sc.activated = a.template get<int>(S * cap); sc.refound = a.template get<int>(S * cap);
    sc.newly_lost = a.template get<int>(S * cap); sc.newly_removed = a.template get<int>(S * cap);
    sc.match_slot = a.template get<int>(S * nd); sc.match_det = a.template get<int>(S * nd);
    sc.match_flag = a.template get<int>(S * nd);
    sc.drop_a = a.template get<int>(S * cap); sc.drop_b = a.template get<int>(S * cap);
    sc.cost = a.template get<double>(S * cap * nd);
    sc.lap_x = a.template get<int>(S * cap); sc.lap_y = a.template get<int>(S * nd);
    sc.lap_u = a.template get<double>(S * nd); sc.lap_v = a.template get<double>(S * cap);
    sc.lap_minv = a.template get<double>(S * cap);
    sc.lap_way = a.template get<int>(S * cap); sc.lap_used = a.template get<int>(S * cap);
    sc.box_a = a.template get<double>(S * cap * box_w);
    sc.pair_list = a.template get<int>(S * 4096);
}

inline BotSortConfigDev make_config_dev(double high, double low, double new_thresh, double match, double prox,
                                        double app, double second, double unc, double unc_scale, int fuse,
                                        int with_reid, int frame_rate, int track_buffer, int removed_cap, int kind = 0) {
    BotSortConfigDev d;
    d.track_high_thresh = high; d.track_low_thresh = low; d.new_track_thresh = new_thresh;
    d.match_thresh = match; d.proximity_thresh = prox; d.appearance_thresh = app;
    d.second_match_thresh = second; d.unconfirmed_match_thresh = unc; d.unconfirmed_emb_scale = unc_scale;
    d.new_track_thresh_f32 = (float)new_thresh;
    d.fuse_first_associate = fuse; d.with_reid = with_reid;
    d.max_time_lost = (int)(frame_rate / 30.0 * track_buffer);   // botsort.py:103-104
    d.removed_cap = removed_cap;
    d.kind = kind;
    return d;
}

}  // namespace bm
