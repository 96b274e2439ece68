"""HOTA / CLEAR (MOTA) / Identity (IDF1) of MOT-challenge result rows against ground truth: the acceptance metrics of the
reference's evaluation flow (``boxmot eval``: boxmot/engine/eval/evaluator.py -> engine/eval/trackeval/runner.py:162-241
``trackeval_aabb`` -> the TrackEval package; summary columns HOTA, MOTA, IDF1, AssA, AssRe, IDSW, IDs,
engine/eval/trackeval/results.py:14).

The arithmetic lives in the third-party **TrackEval** package (JonathonLuiten/TrackEval; the reference runs its
``scripts/run_mot_challenge.py``), which is absent offline: this is a restatement of its published algorithm -- parity
unpinned against the package itself:
  * MOT-challenge 2D box preprocessing (``MotChallenge2DBox.get_preprocessed_seq_data``): evaluate class 1 (pedestrian)
    ground truth with a non-zero consider flag; tracker boxes that match (IoU >= 0.5, Hungarian) a distractor-class ground
    truth (2 person on vehicle, 7 static person, 8 distractor, 12 reflection) are removed first;
  * CLEAR (``metrics/clear.py``): per-frame Hungarian matching on IoU >= 0.5 with a 1000-point bonus for keeping the previous
    frame's partner; MOTA = (TP - FP - IDSW) / (TP + FN), MOTP = mean IoU of the TPs, IDSW / Frag / MT / PT / ML;
  * Identity (``metrics/identity.py``): one global Hungarian assignment of ground-truth ids to tracker ids minimising
    IDFN + IDFP; IDF1 = IDTP / (IDTP + 0.5 IDFP + 0.5 IDFN);
  * HOTA (``metrics/hota.py``): alpha = 0.05 ... 0.95; global alignment scores from soft Jaccard potential matches, per-frame
    Hungarian on alignment * IoU, DetA / AssA / AssRe / AssPr / LocA per alpha, HOTA = mean over alpha of sqrt(DetA AssA).
Host-side evaluation code: NumPy + SciPy's ``linear_sum_assignment`` (exactly what TrackEval calls), nothing on the device.
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import linear_sum_assignment

ALPHAS = np.arange(0.05, 0.99, 0.05)
EPS = np.finfo("float").eps
DISTRACTOR_CLASSES = (2, 7, 8, 12)
PEDESTRIAN = 1


def box_iou_ltwh(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """IoU of (N,4) x (M,4) boxes given as left, top, width, height (TrackEval ``_calculate_box_ious``, box_format x0y0wh)."""
    a, b = np.asarray(a, dtype=float).reshape(-1, 4), np.asarray(b, dtype=float).reshape(-1, 4)
    ax1, ay1, ax2, ay2 = a[:, 0], a[:, 1], a[:, 0] + a[:, 2], a[:, 1] + a[:, 3]
    bx1, by1, bx2, by2 = b[:, 0], b[:, 1], b[:, 0] + b[:, 2], b[:, 1] + b[:, 3]
    iw = np.maximum(np.minimum(ax2[:, None], bx2[None]) - np.maximum(ax1[:, None], bx1[None]), 0)
    ih = np.maximum(np.minimum(ay2[:, None], by2[None]) - np.maximum(ay1[:, None], by1[None]), 0)
    inter = iw * ih
    union = (a[:, 2] * a[:, 3])[:, None] + (b[:, 2] * b[:, 3])[None] - inter
    iou = np.zeros_like(inter)
    ok = union > 0 + EPS
    iou[ok] = inter[ok] / union[ok]
    return iou


def preprocess_mot(gt_rows: np.ndarray, res_rows: np.ndarray, n_frames: int | None = None) -> dict:
    """gt rows [frame, id, l, t, w, h, consider, class, visibility]; result rows [frame, id, l, t, w, h, conf, ...] (MOT text
    format, 1-based frames).  Returns per-frame id / box arrays with contiguous ids, as TrackEval's preprocessed ``data``."""
    gt_rows = np.asarray(gt_rows, dtype=float).reshape(-1, 9) if np.size(gt_rows) else np.zeros((0, 9))
    res_rows = np.asarray(res_rows, dtype=float)
    res_rows = res_rows.reshape(-1, res_rows.shape[-1]) if res_rows.size else np.zeros((0, 7))
    last = int(max(gt_rows[:, 0].max() if len(gt_rows) else 0, res_rows[:, 0].max() if len(res_rows) else 0))
    T = n_frames or last
    gt_ids, tr_ids, gt_boxes, tr_boxes, sims = [], [], [], [], []
    for t in range(1, T + 1):
        g = gt_rows[gt_rows[:, 0] == t]
        r = res_rows[res_rows[:, 0] == t]
        g_cls, g_zero = g[:, 7].astype(int), g[:, 6]
        sim = box_iou_ltwh(g[:, 2:6], r[:, 2:6])
        # tracker boxes matched to a distractor ground truth are dropped
        drop = np.zeros(len(r), bool)
        if len(g) and len(r):
            m = sim.copy()
            m[m < 0.5 - EPS] = 0
            rows, cols = linear_sum_assignment(-m)
            ok = m[rows, cols] > 0 + EPS
            rows, cols = rows[ok], cols[ok]
            drop[cols[np.isin(g_cls[rows], DISTRACTOR_CLASSES)]] = True
        keep_g = (g_zero != 0) & (g_cls == PEDESTRIAN)
        gt_ids.append(g[keep_g, 1].astype(int)); gt_boxes.append(g[keep_g, 2:6])
        tr_ids.append(r[~drop, 1].astype(int)); tr_boxes.append(r[~drop, 2:6])
        sims.append(sim[keep_g][:, ~drop])
    def relabel(lists):
        uniq = np.unique(np.concatenate(lists)) if any(len(x) for x in lists) else np.zeros(0, int)
        lut = {int(v): i for i, v in enumerate(uniq)}
        return [np.array([lut[int(v)] for v in x], dtype=int) for x in lists], len(uniq)
    gt_ids, n_gt = relabel(gt_ids)
    tr_ids, n_tr = relabel(tr_ids)
    return dict(gt_ids=gt_ids, tracker_ids=tr_ids, similarity_scores=sims, num_gt_ids=n_gt, num_tracker_ids=n_tr,
                num_gt_dets=int(sum(len(x) for x in gt_ids)), num_tracker_dets=int(sum(len(x) for x in tr_ids)), num_timesteps=T)


def clear(data: dict, threshold: float = 0.5) -> dict:
    res = dict(CLR_TP=0, CLR_FN=0, CLR_FP=0, IDSW=0, MT=0, PT=0, ML=0, Frag=0, MOTP_sum=0.0)
    if data["num_tracker_dets"] == 0:
        res["CLR_FN"] = data["num_gt_dets"]; res["ML"] = data["num_gt_ids"]
        return _clear_final(res)
    if data["num_gt_dets"] == 0:
        res["CLR_FP"] = data["num_tracker_dets"]
        return _clear_final(res)
    n_gt = data["num_gt_ids"]
    gt_id_count, gt_matched, gt_frag = np.zeros(n_gt), np.zeros(n_gt), np.zeros(n_gt)
    prev_tr = np.full(n_gt, np.nan)
    prev_timestep_tr = np.full(n_gt, np.nan)
    for gt_t, tr_t, sim in zip(data["gt_ids"], data["tracker_ids"], data["similarity_scores"]):
        if len(gt_t) == 0:
            res["CLR_FP"] += len(tr_t)
            continue
        if len(tr_t) == 0:
            res["CLR_FN"] += len(gt_t)
            gt_id_count[gt_t] += 1
            continue
        score = (tr_t[None, :] == prev_timestep_tr[gt_t[:, None]]) * 1000.0 + sim
        score[sim < threshold - EPS] = 0
        rows, cols = linear_sum_assignment(-score)
        ok = score[rows, cols] > 0 + EPS
        rows, cols = rows[ok], cols[ok]
        m_gt, m_tr = gt_t[rows], tr_t[cols]
        prev_matched = prev_tr[m_gt]
        res["IDSW"] += int(np.sum(~np.isnan(prev_matched) & (m_tr != prev_matched)))
        gt_id_count[gt_t] += 1
        gt_matched[m_gt] += 1
        not_prev = np.isnan(prev_timestep_tr)
        prev_tr[m_gt] = m_tr
        prev_timestep_tr[:] = np.nan
        prev_timestep_tr[m_gt] = m_tr
        gt_frag += np.logical_and(not_prev, ~np.isnan(prev_timestep_tr))
        res["CLR_TP"] += len(m_gt)
        res["CLR_FN"] += len(gt_t) - len(m_gt)
        res["CLR_FP"] += len(tr_t) - len(m_gt)
        if len(m_gt):
            res["MOTP_sum"] += float(sim[rows, cols].sum())
    ratio = gt_matched[gt_id_count > 0] / gt_id_count[gt_id_count > 0]
    res["MT"] = int(np.sum(ratio > 0.8)); res["PT"] = int(np.sum(ratio >= 0.2)) - res["MT"]
    res["ML"] = n_gt - res["MT"] - res["PT"]
    res["Frag"] = int(np.sum(np.maximum(gt_frag - 1, 0)))
    return _clear_final(res)


def _clear_final(r: dict) -> dict:
    n_gt = r["CLR_TP"] + r["CLR_FN"]
    r["MOTA"] = (r["CLR_TP"] - r["CLR_FP"] - r["IDSW"]) / max(1.0, n_gt)
    r["MOTP"] = r["MOTP_sum"] / max(1.0, r["CLR_TP"])
    r["MODA"] = (r["CLR_TP"] - r["CLR_FP"]) / max(1.0, n_gt)
    r["CLR_Re"] = r["CLR_TP"] / max(1.0, n_gt)
    r["CLR_Pr"] = r["CLR_TP"] / max(1.0, r["CLR_TP"] + r["CLR_FP"])
    return r


def identity(data: dict, threshold: float = 0.5) -> dict:
    res = dict(IDTP=0, IDFN=0, IDFP=0)
    if data["num_tracker_dets"] == 0:
        res["IDFN"] = data["num_gt_dets"]
        return _id_final(res)
    if data["num_gt_dets"] == 0:
        res["IDFP"] = data["num_tracker_dets"]
        return _id_final(res)
    n_gt, n_tr = data["num_gt_ids"], data["num_tracker_ids"]
    potential = np.zeros((n_gt, n_tr))
    gt_count, tr_count = np.zeros(n_gt), np.zeros(n_tr)
    for gt_t, tr_t, sim in zip(data["gt_ids"], data["tracker_ids"], data["similarity_scores"]):
        m_gt, m_tr = np.nonzero(sim >= threshold)
        potential[gt_t[m_gt], tr_t[m_tr]] += 1
        gt_count[gt_t] += 1
        tr_count[tr_t] += 1
    n = n_gt + n_tr
    fp_mat, fn_mat = np.zeros((n, n)), np.zeros((n, n))
    fp_mat[n_gt:, :n_tr] = 1e10
    fn_mat[:n_gt, n_tr:] = 1e10
    for g in range(n_gt):
        fn_mat[g, :n_tr] = gt_count[g]
        fn_mat[g, n_tr + g] = gt_count[g]
    for t in range(n_tr):
        fp_mat[:n_gt, t] = tr_count[t]
        fp_mat[t + n_gt, t] = tr_count[t]
    fn_mat[:n_gt, :n_tr] -= potential
    fp_mat[:n_gt, :n_tr] -= potential
    rows, cols = linear_sum_assignment(fn_mat + fp_mat)
    res["IDFN"] = int(fn_mat[rows, cols].sum())
    res["IDFP"] = int(fp_mat[rows, cols].sum())
    res["IDTP"] = int(gt_count.sum()) - res["IDFN"]
    return _id_final(res)


def _id_final(r: dict) -> dict:
    r["IDR"] = r["IDTP"] / max(1.0, r["IDTP"] + r["IDFN"])
    r["IDP"] = r["IDTP"] / max(1.0, r["IDTP"] + r["IDFP"])
    r["IDF1"] = r["IDTP"] / max(1.0, r["IDTP"] + 0.5 * r["IDFP"] + 0.5 * r["IDFN"])
    return r


def hota(data: dict) -> dict:
    nA = len(ALPHAS)
    tp, fn, fp, loc = np.zeros(nA), np.zeros(nA), np.zeros(nA), np.zeros(nA)
    assa = np.zeros(nA); assre = np.zeros(nA); asspr = np.zeros(nA)
    if data["num_tracker_dets"] == 0:
        fn[:] = data["num_gt_dets"]
        return _hota_final(tp, fn, fp, loc + 1.0, assa, assre, asspr)
    if data["num_gt_dets"] == 0:
        fp[:] = data["num_tracker_dets"]
        return _hota_final(tp, fn, fp, loc + 1.0, assa, assre, asspr)
    n_gt, n_tr = data["num_gt_ids"], data["num_tracker_ids"]
    potential = np.zeros((n_gt, n_tr))
    gt_count, tr_count = np.zeros((n_gt, 1)), np.zeros((1, n_tr))
    for gt_t, tr_t, sim in zip(data["gt_ids"], data["tracker_ids"], data["similarity_scores"]):
        denom = sim.sum(0)[None, :] + sim.sum(1)[:, None] - sim
        sim_iou = np.zeros_like(sim)
        ok = denom > 0 + EPS
        sim_iou[ok] = sim[ok] / denom[ok]
        potential[gt_t[:, None], tr_t[None, :]] += sim_iou
        gt_count[gt_t] += 1
        tr_count[0, tr_t] += 1
    align = potential / (gt_count + tr_count - potential)
    matches = [np.zeros((n_gt, n_tr)) for _ in ALPHAS]
    for gt_t, tr_t, sim in zip(data["gt_ids"], data["tracker_ids"], data["similarity_scores"]):
        if len(gt_t) == 0:
            fp += len(tr_t)
            continue
        if len(tr_t) == 0:
            fn += len(gt_t)
            continue
        score = align[gt_t[:, None], tr_t[None, :]] * sim
        rows, cols = linear_sum_assignment(-score)
        for a, alpha in enumerate(ALPHAS):
            ok = sim[rows, cols] >= alpha - EPS
            r, c = rows[ok], cols[ok]
            n = len(r)
            tp[a] += n; fn[a] += len(gt_t) - n; fp[a] += len(tr_t) - n
            if n:
                loc[a] += float(sim[r, c].sum())
                matches[a][gt_t[r], tr_t[c]] += 1
    for a in range(nA):
        mc = matches[a]
        ass_a = mc / np.maximum(1, gt_count + tr_count - mc)
        assa[a] = np.sum(mc * ass_a) / np.maximum(1, tp[a])
        assre[a] = np.sum(mc * (mc / np.maximum(1, gt_count))) / np.maximum(1, tp[a])
        asspr[a] = np.sum(mc * (mc / np.maximum(1, tr_count))) / np.maximum(1, tp[a])
    loc = np.maximum(1e-10, loc) / np.maximum(1e-10, tp)
    return _hota_final(tp, fn, fp, loc, assa, assre, asspr)


def _hota_final(tp, fn, fp, loc, assa, assre, asspr) -> dict:
    deta = tp / np.maximum(1, tp + fn + fp)
    detre = tp / np.maximum(1, tp + fn)
    detpr = tp / np.maximum(1, tp + fp)
    h = np.sqrt(deta * assa)
    return dict(HOTA=float(h.mean()), DetA=float(deta.mean()), AssA=float(assa.mean()), DetRe=float(detre.mean()),
                DetPr=float(detpr.mean()), AssRe=float(assre.mean()), AssPr=float(asspr.mean()), LocA=float(loc.mean()),
                HOTA_alpha=h, HOTA_TP=tp, HOTA_FN=fn, HOTA_FP=fp)


def evaluate_mot(gt_rows, res_rows, n_frames: int | None = None) -> dict:
    """All three metric families of one sequence + the reference's summary columns (results.py:14)."""
    data = preprocess_mot(gt_rows, res_rows, n_frames)
    out = {}
    out.update(hota(data)); out.update(clear(data)); out.update(identity(data))
    out["IDs"] = data["num_tracker_ids"]; out["GT_IDs"] = data["num_gt_ids"]
    out["Dets"] = data["num_tracker_dets"]; out["GT_Dets"] = data["num_gt_dets"]
    out["summary"] = {k: out[k] for k in ("HOTA", "MOTA", "IDF1", "AssA", "AssRe", "IDSW", "IDs")}
    return out
