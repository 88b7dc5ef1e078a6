"""Generates tests/golden/objtrack_*.npz from the REFERENCE's own LBMObjTracker (lbm/models/lbm/object_tracker.py),
run on the host CPU.  TEST INFRASTRUCTURE ONLY, build container only (needs /root/reference).

Import note: object_tracker.py imports lbm/utils/tracker_utils.py, whose first lines import ``addict`` (not installed
here) for BaseTracker.init_track / BaseTracker.memo -- two methods LBMObjTracker never reaches (it overrides ``memo`` and
keeps plain dicts).  A placeholder module whose ``Dict`` raises on any use lets the import statement pass while proving
that nothing of addict takes part in the outputs recorded here.  Only OUTPUTS are stored; the detections are
regenerated from seeds by oracle/objtrack_scenario.py on both sides.   Run:  python oracle/make_golden_objtrack.py"""
from __future__ import annotations

import importlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
REF = "/root/reference"

from oracle import objtrack_scenario as sc  # noqa: E402

CASES = {"objtrack_tao": ("tao", 42, {}), "objtrack_bft": ("bft", 43, {}), "objtrack_plain": ("plain", 44, {"n_objects": 5, "T": 30})}


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("make_golden_objtrack.py needs /root/reference (build container only)")
    from oracle.make_golden import pin_reference_packages
    pin_reference_packages()
    if "addict" not in sys.modules:
        class Dict:  # noqa: D401
            def __init__(self, *a, **k):
                raise RuntimeError("addict.Dict is a placeholder here: this path must not be reached")
        m = types.ModuleType("addict")
        m.Dict = Dict
        sys.modules["addict"] = m
    ot = importlib.import_module("lbm.models.lbm.object_tracker")
    assert ot.__file__.startswith(REF + "/"), ot.__file__
    return ot


def main():
    ot = import_reference()
    out_dir = os.path.join(os.path.dirname(HERE), "tests", "golden")
    for name, (cfg, seed, kw) in CASES.items():
        tracker = ot.LBMObjTracker(sc.tracker_args(cfg))
        frames = sc.run(tracker, seed, **kw)
        d = sc.pack(frames)
        path = os.path.join(out_dir, name + ".npz")
        np.savez_compressed(path, **d)
        n_tr = [len(f["memo_ids"]) for f in frames]
        print(f"{name}: {len(frames)} frames, tracklets/frame max {max(n_tr)}, ids issued {int(max(f['memo_ids'].max(initial=-1) for f in frames)) + 1}, "
              f"removed rows {int(sum(f['remove_mask'].any(axis=-1).sum() if f['remove_mask'].ndim == 2 else 0 for f in frames))}, "
              f"re-planted points {int(sum(f['memo_update'].sum() for f in frames))}, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
