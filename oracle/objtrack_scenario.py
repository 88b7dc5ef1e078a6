"""Closed-loop synthetic detection sequences for the object-association layer.  TEST INFRASTRUCTURE:
imported by oracle/make_golden_objtrack.py (which drives the REFERENCE's LBMObjTracker) and by tests/ (which drive
lbm_amd.object_tracker with the same sequence and compare with the committed outputs).

A scenario is a detector and a point tracker simulated on the host: objects move under a common pan plus their own
drift; every frame the detector reports jittered boxes with scores and labels (with drop-outs, one-frame false
positives, low-score duplicates, a label flip and an empty frame), and the "point tracker" reports each planted point
at its planted position moved by the pan since it was planted (+ noise), so points of drifting objects leave their
boxes and get re-planted.  The tracker under test only sees (frame_id, boxes, scores, labels, point_coords)."""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch

CONFIGS = {
    # lbm/configs/objtrack_tao.yaml:35-49
    "tao": dict(object_points=16, init_score_thr=0.01, obj_score_thr=0.01, match_score_thr=0.01, memo_tracklet_frames=10,
                memo_point_frames=3, distractor_score_thr=0.5, distractor_nms_thr=0.3, rm_distractor=True, max_distance=100.0,
                fps=30, weight_match_with_det_scores=True, weight_match_with_det_labels=True),
    # lbm/configs/objtrack_bft.yaml:35-49
    "bft": dict(object_points=16, init_score_thr=0.3, obj_score_thr=0.2, match_score_thr=0.01, memo_tracklet_frames=10,
                memo_point_frames=2, distractor_score_thr=0.5, distractor_nms_thr=0.3, rm_distractor=True, max_distance=100.0,
                fps=30, weight_match_with_det_scores=True, weight_match_with_det_labels=True),
    # the switches the shipped configs leave on, turned off; a short tracklet memory; few points per object
    "plain": dict(object_points=5, init_score_thr=0.25, obj_score_thr=0.1, match_score_thr=0.3, memo_tracklet_frames=4,
                  memo_point_frames=5, distractor_score_thr=0.5, distractor_nms_thr=0.3, rm_distractor=False, max_distance=-1,
                  fps=10, weight_match_with_det_scores=False, weight_match_with_det_labels=False),
}


def tracker_args(name: str) -> SimpleNamespace:
    return SimpleNamespace(**CONFIGS[name])


class Scenario:
    def __init__(self, seed: int, n_objects: int = 7, T: int = 40, W: int = 854, H: int = 480):
        self.rng = np.random.default_rng(seed)
        r = self.rng
        self.T, self.W, self.H = T, W, H
        self.centre = np.stack([r.uniform(0.15 * W, 0.85 * W, n_objects), r.uniform(0.15 * H, 0.85 * H, n_objects)], 1)
        self.size = np.stack([r.uniform(40, 160, n_objects), r.uniform(40, 140, n_objects)], 1)
        self.drift = r.uniform(-4.0, 4.0, (n_objects, 2)) * (r.uniform(size=(n_objects, 1)) < 0.5)
        self.label = r.integers(0, 4, n_objects)
        self.base_score = r.uniform(0.15, 0.95, n_objects)
        self.first = r.integers(0, 6, n_objects)                      # objects enter over the first frames
        self.gap = [(int(a), int(a + l)) for a, l in zip(r.integers(8, max(T - 14, 9), n_objects), r.choice([0, 2, 5, 13], n_objects))]
        self.pan = np.cumsum(np.concatenate([[[0.0, 0.0]], r.normal([2.0, 0.5], 1.0, (T, 2))]), 0)
        self.planted = {}                                             # (track id, point slot) -> (position, frame)

    def detections(self, t: int):
        r = self.rng
        if t == 17:                                                   # the detector found nothing
            return np.zeros((0, 4), np.float32), np.zeros(0, np.float32), np.zeros(0, np.int64)
        boxes, scores, labels = [], [], []
        for k in range(len(self.label)):
            if t < self.first[k] or self.gap[k][0] <= t < self.gap[k][1]:
                continue
            c = self.centre[k] + self.pan[t] + self.drift[k] * t + r.normal(0, 1.5, 2)
            s = self.size[k] * r.uniform(0.95, 1.05, 2)
            boxes.append([c[0] - s[0] / 2, c[1] - s[1] / 2, c[0] + s[0] / 2, c[1] + s[1] / 2])
            scores.append(float(np.clip(self.base_score[k] + r.normal(0, 0.08), 0.02, 0.99)))
            labels.append(int(self.label[k]) if not (k == 2 and t == 21) else int(self.label[k] + 1))
            if r.uniform() < 0.2:                                     # a low-score duplicate of the same object
                boxes.append(list(np.asarray(boxes[-1]) + r.normal(0, 4, 4)))
                scores.append(float(r.uniform(0.05, 0.45)))
                labels.append(int(r.integers(0, 4)))
        if r.uniform() < 0.3:                                         # a one-frame false positive
            c = np.array([r.uniform(50, self.W - 50), r.uniform(50, self.H - 50)])
            boxes.append([c[0] - 30, c[1] - 25, c[0] + 30, c[1] + 25])
            scores.append(float(r.uniform(0.2, 0.7)))
            labels.append(int(r.integers(0, 4)))
        order = r.permutation(len(boxes))
        return (np.asarray(boxes, np.float32).reshape(-1, 4)[order], np.asarray(scores, np.float32)[order],
                np.asarray(labels, np.int64)[order])

    def point_predictions(self, t: int, memo):
        """What the point tracker would report at frame t for the tracklets' planted points (memo of the tracker under test)."""
        ids, pts = memo[2].tolist(), memo[4].numpy()
        live = {}
        out = np.zeros((len(ids), pts.shape[1], 2), np.float32)
        for i, k in enumerate(ids):
            for j in range(pts.shape[1]):
                pos = pts[i, j]
                rec = self.planted.get((k, j))
                if rec is None or not np.array_equal(rec[0], pos):    # planted (or re-planted) after the previous frame
                    rec = (pos.copy(), t - 1)
                live[(k, j)] = rec
                out[i, j] = pos + (self.pan[t] - self.pan[max(rec[1], 0)]) + self.rng.normal(0, 0.7, 2)
        self.planted = live
        return torch.from_numpy(out.reshape(-1, 2))


def run(tracker, seed: int, **kw):
    """Drives `tracker` (reference or build: same interface) through a scenario; returns per-frame records."""
    sc = Scenario(seed, **kw)
    torch.manual_seed(seed)
    frames = []
    for t in range(sc.T):
        boxes, scores, labels = sc.detections(t)
        coords = vis = None
        memo = tracker.memo
        if memo[4].shape[0] > 0:
            coords = sc.point_predictions(t, memo)
            vis = torch.ones(coords.shape[0], dtype=torch.bool)
        out = tracker.track(t, torch.from_numpy(boxes), torch.from_numpy(scores), torch.from_numpy(labels), coords, vis)
        m = tracker.memo
        frames.append(dict(
            out_bboxes=out["bboxes"].numpy().astype(np.float32), out_labels=out["labels"].numpy().astype(np.int64),
            out_scores=out["scores"].numpy().astype(np.float32), out_ids=out["instances_id"].numpy().astype(np.int64),
            memo_bboxes=m[0].numpy().astype(np.float32), memo_labels=m[1].numpy().astype(np.int64), memo_ids=m[2].numpy().astype(np.int64),
            memo_last=m[3].numpy().astype(np.int64), memo_points=m[4].numpy().astype(np.float32), memo_update=m[5].numpy().astype(bool),
            memo_points_last=m[6].numpy().astype(np.int64), remove_mask=tracker.points_remove_mask.numpy().astype(bool)))
    return frames


KEYS = ("out_bboxes", "out_labels", "out_scores", "out_ids", "memo_bboxes", "memo_labels", "memo_ids", "memo_last", "memo_points",
        "memo_update", "memo_points_last", "remove_mask")


def pack(frames):
    """Ragged per-frame records -> flat arrays + offsets, for one .npz."""
    d = {}
    for key in KEYS:
        arrs = [f[key] for f in frames]
        d[key] = np.concatenate([a.reshape(-1) for a in arrs]) if arrs else np.zeros(0)
        d[key + "_n"] = np.asarray([a.size for a in arrs], np.int64)
    return d


def unpack(d, key, t):
    off = np.concatenate([[0], np.cumsum(d[key + "_n"])])
    return d[key][off[t]:off[t + 1]]
