"""Import-path twin of lbm/utils/tracker_utils.py.  With the reference checkout and ``addict`` (tracker_utils.py:8) available
the reference's own file is executed here (BaseTracker, KalmanFilter, bbox helpers); otherwise the three helpers the object
tracker uses (tracker_utils.py:53, 727, 736) come from lbm_amd/object_tracker.py."""
from lbm_amd._refshim import delegate as _delegate

if not _delegate(globals()):
    from lbm_amd.object_tracker import bbox_overlaps, check_points_in_box, sample_points_in_box  # noqa: F401
