"""Object-level association on top of the point tracker (reference: lbm/models/lbm/object_tracker.py:12-418 and the
three helpers it takes from lbm/utils/tracker_utils.py:53-239, 727-744).

The reference keeps one dict per tracklet holding small device tensors and walks them in Python, so a frame with M
tracklets costs O(M) tiny device launches and host syncs.  Here the tracklets are a structure of arrays on the HOST
(M x 4 boxes, M x npt x 2 points, ...): association is a few dozen boxes of control flow whose results the caller
reads on the host anyway (tools/eval_objtrack.py:89-93), so the only device traffic per frame is one copy of the
predicted point coordinates; everything that touches the GPU state (adding / dropping points) goes through
``LBM_online.update`` -> ``lbm_hip_update_points`` in one launch pair.

Behaviour follows the reference decision for decision, including its corner cases (documented where they occur):
same thresholds, same order of tracklets (insertion order, survivors keep their order), same order of random draws
(``torch.rand`` twice per sampled box: x then y), so with the same seed on the same device the two produce identical
tracklets.  ``memo``, ``tracks``, ``points_remove_mask``, ``track`` and ``update`` keep the reference's names and shapes.
"""
from __future__ import annotations

from typing import Dict, List

import torch
from torch import Tensor


def bbox_overlaps(b1: Tensor, b2: Tensor, eps: float = 1e-6) -> Tensor:
    """Pairwise IoU of xyxy boxes, (m,4) x (n,4) -> (m,n) (tracker_utils.py:53-239, mode 'iou', not aligned)."""
    if b1.shape[0] * b2.shape[0] == 0:
        return b1.new_zeros((b1.shape[0], b2.shape[0]))
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    wh = (torch.min(b1[:, None, 2:], b2[None, :, 2:]) - torch.max(b1[:, None, :2], b2[None, :, :2])).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = torch.max(a1[:, None] + a2[None, :] - inter, inter.new_tensor([eps]))
    return inter / union


def check_points_in_box(bboxes: Tensor, points: Tensor) -> Tensor:
    """(n,4) xyxy boxes, (p,2) points -> (n,p) bool, borders inclusive (tracker_utils.py:727-733)."""
    points = points.to(bboxes)
    x, y = points[None, :, 0], points[None, :, 1]
    return (x >= bboxes[:, 0:1]) & (x <= bboxes[:, 2:3]) & (y >= bboxes[:, 1:2]) & (y <= bboxes[:, 3:4])


def sample_points_in_box(bbox: Tensor, n: int) -> Tensor:
    """n uniform points in an xyxy box: all x are drawn first, then all y (tracker_utils.py:736-744)."""
    x0, y0, x1, y1 = bbox
    xs = torch.rand(n, device=bbox.device) * (x1 - x0) + x0
    ys = torch.rand(n, device=bbox.device) * (y1 - y0) + y0
    return torch.stack((xs, ys), dim=1)


class LBMObjTracker:
    """Same constructor contract as the reference (an ``args`` namespace, object_tracker.py:36-63)."""

    def __init__(self, args):
        self.object_points = int(args.object_points)
        self.init_score_thr = args.init_score_thr
        self.obj_score_thr = args.obj_score_thr
        self.match_score_thr = args.match_score_thr
        self.memo_tracklet_frames = args.memo_tracklet_frames
        self.memo_point_frames = args.memo_point_frames
        self.distractor_score_thr = args.distractor_score_thr
        self.distractor_nms_thr = args.distractor_nms_thr
        self.rm_distractor = args.rm_distractor
        self.max_distance = args.max_distance
        self.fps = args.fps
        self.growth_factor = self.fps / 6
        self.distance_smoothing_factor = 100 / self.fps
        self.weight_match_with_det_scores = args.weight_match_with_det_scores
        self.weight_match_with_det_labels = args.weight_match_with_det_labels
        self.backdrops = []
        self.points_remove_mask = None
        self.reset()

    # ---- tracklet store: one row per tracklet, insertion order ---------------------------------
    def reset(self) -> None:
        npt = self.object_points
        self.num_tracks = 0
        self._ids: List[int] = []
        self._row: Dict[int, int] = {}
        self._bbox = torch.empty(0, 4)
        self._label = torch.empty(0, dtype=torch.long)
        self._score = torch.empty(0)
        self._last = torch.empty(0, dtype=torch.long)
        self._tentative = torch.empty(0, dtype=torch.bool)
        self._points = torch.empty(0, npt, 2)
        self._points_update = torch.empty(0, npt, dtype=torch.bool)
        self._points_last = torch.empty(0, npt, dtype=torch.long)

    @property
    def empty(self) -> bool:
        return not self._ids

    @property
    def ids(self) -> List[int]:
        return list(self._ids)

    @property
    def tracks(self) -> dict:
        """Read-only per-tracklet view in the reference's layout (object_tracker.py:192-201)."""
        return {k: dict(bbox=self._bbox[i], points=self._points[i], points_update_mask=self._points_update[i],
                        points_last_frame=self._points_last[i], label=self._label[i], score=self._score[i],
                        last_frame=int(self._last[i]), tentative=bool(self._tentative[i]))
                for i, k in enumerate(self._ids)}

    @property
    def memo(self):
        """(bboxes (M,4), labels (M,1), ids (M,), last frames (M,), points (M,npt,2), points_update_mask (M,npt),
        points_last_frame (M,npt)) -- object_tracker.py:65-121."""
        return (self._bbox.clone(), self._label.view(-1, 1).clone(), torch.tensor(self._ids, dtype=torch.long), self._last.clone(),
                self._points.clone(), self._points_update.clone(), self._points_last.clone())

    def _keep_rows(self, keep: Tensor) -> None:
        self._ids = [k for k, f in zip(self._ids, keep.tolist()) if f]
        self._row = {k: i for i, k in enumerate(self._ids)}
        for name in ("_bbox", "_label", "_score", "_last", "_tentative", "_points", "_points_update", "_points_last"):
            setattr(self, name, getattr(self, name)[keep])

    def _append(self, ids: List[int], bbox: Tensor, label: Tensor, score: Tensor, frame_id: int, points: Tensor) -> None:
        n, npt = len(ids), self.object_points
        for k in ids:
            self._row[k] = len(self._ids)
            self._ids.append(k)
        self._bbox = torch.cat([self._bbox, bbox])
        self._label = torch.cat([self._label, label])
        self._score = torch.cat([self._score, score])
        self._last = torch.cat([self._last, torch.full((n,), frame_id, dtype=torch.long)])
        self._tentative = torch.cat([self._tentative, torch.ones(n, dtype=torch.bool)])
        self._points = torch.cat([self._points, points])
        self._points_update = torch.cat([self._points_update, torch.ones(n, npt, dtype=torch.bool)])
        self._points_last = torch.cat([self._points_last, torch.full((n, npt), frame_id, dtype=torch.long)])

    # ---- object_tracker.py:123-146 ----
    def compute_distance_mask(self, bboxes1: Tensor, bboxes2: Tensor, frame_ids1: Tensor, frame_ids2: Tensor) -> Tensor:
        c1 = (bboxes1[:, :2] + bboxes1[:, 2:]) / 2.0
        c2 = (bboxes2[:, :2] + bboxes2[:, 2:]) / 2.0
        dist = torch.cdist(c1, c2)
        gap = torch.abs(frame_ids1[:, None] - frame_ids2[None, :]).float()
        reach = self.max_distance * torch.exp(gap / self.growth_factor)       # a tracklet unseen for longer may have moved further
        return torch.where(dist <= reach, torch.ones_like(dist), torch.exp(-(dist - reach) / self.distance_smoothing_factor))

    # ---- object_tracker.py:148-239 ----
    def update(self, ids: Tensor, bboxes: Tensor, labels: Tensor, scores: Tensor, frame_id: int) -> None:
        npt = self.object_points
        sel = torch.nonzero(ids > -1).flatten().tolist()
        id_list = [int(ids[j]) for j in sel]
        seen = [j for j, k in zip(sel, id_list) if k in self._row]
        fresh = [(j, k) for j, k in zip(sel, id_list) if k not in self._row]
        if seen:
            rows = torch.tensor([self._row[int(ids[j])] for j in seen], dtype=torch.long)
            det = torch.tensor(seen, dtype=torch.long)
            box = bboxes[det]
            self._bbox[rows] = box
            self._label[rows] = labels[det]
            self._score[rows] = scores[det]
            self._last[rows] = frame_id
            # the stored points are the QUERY positions (where the points were planted), not their current predictions
            p = self._points[rows]
            inside = ((p[..., 0] >= box[:, None, 0]) & (p[..., 0] <= box[:, None, 2]) &
                      (p[..., 1] >= box[:, None, 1]) & (p[..., 1] <= box[:, None, 3]))
            last = self._points_last[rows]
            last[inside] = frame_id
            self._points_last[rows] = last
            self._points_update[rows] = False
            self._tentative[rows] = False
        if fresh:
            det = torch.tensor([j for j, _ in fresh], dtype=torch.long)
            pts = torch.stack([sample_points_in_box(bboxes[j], npt) for j, _ in fresh])     # draws in detection order
            self._append([k for _, k in fresh], bboxes[det], labels[det], scores[det], frame_id, pts)
        # drop tracklets unseen for memo_tracklet_frames, and tentative ones (one sighting) not confirmed by the next frame.
        # points_remove_mask describes the rows BEFORE the drop: it is what LBM_online.update applies to its point state.
        if self.empty:
            self.points_remove_mask = torch.ones(0, dtype=torch.bool)
        else:
            drop = ((frame_id - self._last) >= self.memo_tracklet_frames) | (self._tentative & (self._last < frame_id))
            self.points_remove_mask = drop[:, None].expand(-1, npt).clone()
            if bool(drop.any()):
                self._keep_rows(~drop)
        # points that have been outside their object's box for memo_point_frames are re-planted inside the current box
        stale = (frame_id - self._points_last) >= self.memo_point_frames
        for i in torch.nonzero(stale.any(dim=1)).flatten().tolist():
            m = stale[i]
            self._points[i][m] = sample_points_in_box(self._bbox[i], int(m.sum()))
            self._points_last[i][m] = frame_id
            self._points_update[i][m] = True

    # ---- object_tracker.py:242-373 ----
    def track(self, frame_id: int, bboxes: Tensor, scores: Tensor, labels: Tensor, point_coords: Tensor, point_vis: Tensor):
        """frame_id: current frame; bboxes (n,4) xyxy, scores (n,), labels (n,) detections; point_coords (M*npt, 2) the
        point tracker's predictions for the points of the M tracklets in ``memo`` order (point_vis is accepted and, as in
        the reference, unused).  Returns {'bboxes','labels','scores','instances_id'} of the detections that carry an id."""
        bboxes = torch.as_tensor(bboxes, dtype=torch.float32).cpu()
        scores = torch.as_tensor(scores, dtype=torch.float32).cpu()
        labels = torch.as_tensor(labels).long().cpu()
        if frame_id == 0:
            self.reset()
        if self.empty or bboxes.shape[0] == 0:
            keep = scores > self.init_score_thr
            bboxes, scores, labels = bboxes[keep], scores[keep], labels[keep]
            n_new = bboxes.shape[0]
            ids = torch.arange(self.num_tracks, self.num_tracks + n_new)
            self.num_tracks += n_new
        else:
            keep = scores > self.obj_score_thr
            bboxes, scores, labels = bboxes[keep], scores[keep], labels[keep]
            ids = torch.full((bboxes.shape[0],), -1, dtype=torch.long)
            memo_bboxes, memo_labels, memo_ids, memo_last = self._bbox, self._label, list(self._ids), self._last
            if self.rm_distractor:
                bboxes, labels, scores, ids, _, _ = self.remove_distractor(
                    bboxes, labels, scores, ids, nms="inter", distractor_score_thr=self.distractor_score_thr,
                    distractor_nms_thr=self.distractor_nms_thr)
            n, M = bboxes.shape[0], memo_bboxes.shape[0]
            # a detection votes for a tracklet with the number of that tracklet's points predicted inside its box
            coords = torch.as_tensor(point_coords, dtype=torch.float32).cpu()
            counts = check_points_in_box(bboxes, coords).view(n, M, self.object_points).sum(dim=-1)
            det_area = (bboxes[:, 2] - bboxes[:, 0]) * (bboxes[:, 3] - bboxes[:, 1])
            memo_area = (memo_bboxes[:, 2] - memo_bboxes[:, 0]) * (memo_bboxes[:, 3] - memo_bboxes[:, 1])
            match = counts * torch.clamp(memo_area[None, :] / det_area[:, None], min=0.1, max=1.0)   # large boxes swallow points
            if self.weight_match_with_det_scores:
                match = match * scores[:, None]
            if self.weight_match_with_det_labels:
                same = labels[:, None] == memo_labels[None, :]
                match = match * torch.where(same, torch.tensor(1.0), torch.tensor(0.5))
            match = (match.softmax(dim=0) + match.softmax(dim=1)) / 2.0
            if self.max_distance != -1:
                now = torch.full((n,), frame_id, dtype=torch.long)
                match = match * self.compute_distance_mask(bboxes, memo_bboxes, now, memo_last)
            # greedy, in detection order: a detection takes its best tracklet, which is then closed to the others
            for i in range(n):
                conf, j = torch.max(match[i, :], dim=0)
                if conf > self.match_score_thr and memo_ids[int(j)] > -1 and scores[i] > self.obj_score_thr:
                    ids[i] = memo_ids[int(j)]
                    match[:i, j] = 0
                    match[i + 1:, j] = 0
            new = (ids == -1) & (scores > self.init_score_thr)
            n_new = int(new.sum())
            ids[new] = torch.arange(self.num_tracks, self.num_tracks + n_new, dtype=torch.long)
            self.num_tracks += n_new
        self.update(ids, bboxes, labels, scores, frame_id)
        tracked = ids > -1
        return dict(bboxes=bboxes[tracked], labels=labels[tracked], scores=scores[tracked], instances_id=ids[tracked])

    # ---- object_tracker.py:375-418 ----
    def remove_distractor(self, bboxes, labels, scores, ids, track_feats=None, mask_inds=(), distractor_score_thr=0.5,
                          distractor_nms_thr=0.3, nms="inter"):
        """A low-score box that overlaps any EARLIER box (earlier = smaller index, whether or not that one survives) by
        more than the threshold is dropped; 'intra' restricts this to boxes of the same label."""
        valid = labels > -1
        low = torch.nonzero(scores < distractor_score_thr).flatten()
        if nms not in ("inter", "intra"):
            raise NotImplementedError(nms)
        iou = bbox_overlaps(bboxes[low], bboxes)
        if nms == "intra":
            iou = iou * (labels[low][:, None] == labels[None, :])
        if low.numel():
            earlier = torch.arange(bboxes.shape[0])[None, :] < low[:, None]
            valid[low[((iou > distractor_nms_thr) & earlier).any(dim=1)]] = False
        if track_feats is not None:
            track_feats = track_feats[valid]
        if len(mask_inds) > 0:
            mask_inds = mask_inds[valid]
        return bboxes[valid], labels[valid], scores[valid], ids[valid], track_feats, mask_inds
