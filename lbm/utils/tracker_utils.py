"""Reference import path lbm.utils.tracker_utils: the three helpers the object tracker uses (tracker_utils.py:53, 727, 736).
BaseTracker / KalmanFilter (tracker_utils.py:242-725) are not used by any caller of the tracking path and are not provided."""
from lbm_amd.object_tracker import bbox_overlaps, check_points_in_box, sample_points_in_box  # noqa: F401
