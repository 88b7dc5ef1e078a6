"""Object-association layer: lbm_amd.object_tracker against the outputs of the REFERENCE's LBMObjTracker
(tests/golden/objtrack_*.npz, written by oracle/make_golden_objtrack.py) on the same closed-loop detection sequences.
Host logic: runs without a GPU.  Bar: ids, masks and frame counters exact; boxes / points / scores bit-equal (both
sides are fp32 torch on the host with the same draws from the same generator)."""
import os

import numpy as np
import pytest
import torch

from lbm_amd.object_tracker import LBMObjTracker, bbox_overlaps, check_points_in_box, sample_points_in_box
from oracle import objtrack_scenario as sc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = {"objtrack_tao": ("tao", 42, {}), "objtrack_bft": ("bft", 43, {}), "objtrack_plain": ("plain", 44, {"n_objects": 5, "T": 30})}


@pytest.mark.parametrize("name", sorted(CASES))
def test_tracklets_match_the_reference_frame_by_frame(name):
    cfg, seed, kw = CASES[name]
    want = np.load(os.path.join(GOLDEN, name + ".npz"))
    frames = sc.run(LBMObjTracker(sc.tracker_args(cfg)), seed, **kw)
    assert len(frames) == len(want["out_ids_n"])
    for t, f in enumerate(frames):
        for key in sc.KEYS:
            ref = sc.unpack(want, key, t)
            got = f[key].reshape(-1)
            assert got.shape == ref.shape, (name, t, key, got.shape, ref.shape)
            assert np.array_equal(got, ref), (name, t, key)


def test_scenarios_exercise_every_branch():
    """The sequences must contain what the association has to cope with: births, tentative drops, lost tracklets,
    re-planted points, an empty frame, id re-use never happening."""
    want = np.load(os.path.join(GOLDEN, "objtrack_plain.npz"))
    T = len(want["out_ids_n"])
    ids_seen = np.concatenate([sc.unpack(want, "memo_ids", t) for t in range(T)])
    assert ids_seen.max() >= 12
    assert sum(int(sc.unpack(want, "remove_mask", t).any()) for t in range(T)) >= 5
    assert any(want["out_ids_n"][t] == 0 for t in range(T))
    tao = np.load(os.path.join(GOLDEN, "objtrack_tao.npz"))
    assert sum(int(sc.unpack(tao, "memo_update", t).sum()) for t in range(1, len(tao["out_ids_n"]))) > 100


def test_helpers():
    b = torch.tensor([[0., 0., 10., 10.], [10., 10., 20., 20.], [32., 32., 38., 42.]])
    c = torch.tensor([[0., 0., 10., 20.], [0., 10., 10., 19.], [10., 10., 20., 20.]])
    iou = bbox_overlaps(b, c)
    assert iou.shape == (3, 3)
    assert torch.allclose(iou[0], torch.tensor([0.5, 0.0, 0.0])) and float(iou[1, 2]) == 1.0
    assert tuple(bbox_overlaps(torch.empty(0, 4), c).shape) == (0, 3)
    inside = check_points_in_box(b[:2], torch.tensor([[10., 10.], [5., 11.], [-1., 3.]]))
    assert inside.tolist() == [[True, False, False], [True, False, False]]          # borders count as inside
    torch.manual_seed(3)
    p = sample_points_in_box(torch.tensor([2., 3., 4., 9.]), 50)
    assert p.shape == (50, 2) and bool(((p[:, 0] >= 2) & (p[:, 0] <= 4) & (p[:, 1] >= 3) & (p[:, 1] <= 9)).all())


def test_frame_zero_resets_and_no_detections_keeps_tracklets():
    tr = LBMObjTracker(sc.tracker_args("plain"))
    box = torch.tensor([[10., 10., 50., 50.]])
    out = tr.track(0, box, torch.tensor([0.9]), torch.tensor([1]), None, None)
    assert out["instances_id"].tolist() == [0] and tr.memo[5].all()
    pts = tr.memo[4].reshape(-1, 2)
    out = tr.track(1, box, torch.tensor([0.9]), torch.tensor([1]), pts, None)
    assert out["instances_id"].tolist() == [0] and not tr.memo[5].any() and not tr.tracks[0]["tentative"]
    out = tr.track(2, torch.empty(0, 4), torch.empty(0), torch.empty(0, dtype=torch.long), pts, None)
    assert out["instances_id"].numel() == 0 and tr.ids == [0] and not tr.points_remove_mask.any()
    out = tr.track(0, box, torch.tensor([0.9]), torch.tensor([1]), pts, None)       # a new video: ids restart
    assert out["instances_id"].tolist() == [0] and tr.memo[5].all()
