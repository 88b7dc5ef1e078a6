"""load_video with the reference's contract (lbm/utils/demo_utils.py:8-135):
returns (frames: list of HxWx3 uint8 BGR arrays, video: (1,T,3,H,W) float32 RGB in [0,255], query_points (N,2) xy).

Decoding backends, tried in order: OpenCV when importable (what the reference uses), PIL for image directories,
``.npy`` / ``.pt`` arrays of shape (T,H,W,3) uint8 RGB, and the spec ``synthetic:T:H:W[:seed]`` (the
container has no video decoder, so benchmarks use the seeded synthetic clip).  Grid queries follow the reference
rule x = int(i*W/(g+1)), y = int(j*H/(g+1)) (demo_utils.py:97-107); click queries need an OpenCV GUI."""
import glob
import os

import numpy as np
import torch

from lbm_amd.synth import grid_queries, synthetic_video

_IMG_EXT = ("*.jpg", "*.jpeg", "*.png", "*.bmp", "*.tiff")


def _cv2():
    try:
        import cv2
        return cv2
    except ImportError:
        return None


def _read_rgb_frames(video_path):
    if video_path.startswith("synthetic:"):
        parts = video_path.split(":")
        T, H, W = int(parts[1]), int(parts[2]), int(parts[3])
        seed = int(parts[4]) if len(parts) > 4 else 42
        v = synthetic_video(T, H, W, seed=seed)[0]                              # (T,3,H,W) float
        return [f.permute(1, 2, 0).numpy().astype(np.uint8) for f in v]
    cv2 = _cv2()
    if os.path.isdir(video_path):
        paths = sorted(p for ext in _IMG_EXT for p in glob.glob(os.path.join(video_path, ext)))
        if not paths:
            raise IOError(f"No image files found in directory: {video_path}")
        if cv2 is not None:
            out = [cv2.imread(p) for p in paths]
            return [cv2.cvtColor(f, cv2.COLOR_BGR2RGB) for f in out if f is not None]
        from PIL import Image
        return [np.asarray(Image.open(p).convert("RGB")) for p in paths]
    if os.path.isfile(video_path):
        if video_path.endswith(".npy"):
            return list(np.load(video_path, allow_pickle=False))
        if video_path.endswith(".pt"):
            return list(torch.load(video_path, weights_only=True).numpy())
        if cv2 is None:
            raise IOError(f"cannot decode {video_path}: OpenCV is not installed (use an image directory, .npy/.pt, "
                          f"or synthetic:T:H:W)")
        cap = cv2.VideoCapture(video_path)
        if not cap.isOpened():
            raise IOError(f"Unable to open video file: {video_path}")
        out = []
        while True:
            ok, f = cap.read()
            if not ok:
                break
            out.append(cv2.cvtColor(f, cv2.COLOR_BGR2RGB))
        cap.release()
        if not out:
            raise IOError(f"Could not read any frames from video: {video_path}")
        return out
    raise FileNotFoundError(f"Path is not a valid file or directory: {video_path}")


def load_video(video_path, click_query=False, grid_query=False, grid_size=10):
    rgb = _read_rgb_frames(video_path)
    shape0 = rgb[0].shape
    for i, f in enumerate(rgb):
        if f.shape != shape0:
            raise ValueError(f"Inconsistent frame shapes detected. Frame 0: {shape0}, Frame {i}: {f.shape}")
    H, W = shape0[:2]
    pts = torch.zeros((0, 2), dtype=torch.float32)
    if click_query:
        cv2 = _cv2()
        if cv2 is None:
            raise RuntimeError("click_query needs an OpenCV GUI; pass grid_query=True or supply points yourself")
        picked, title = [], "First Frame - Click to add points, Press 'q' to finish"
        canvas = np.ascontiguousarray(rgb[0][:, :, ::-1])

        def on_click(event, x, y, flags, param):
            if event == cv2.EVENT_LBUTTONDOWN:
                picked.append((x, y))
                cv2.circle(canvas, (x, y), 5, (0, 255, 0), -1)
                cv2.imshow(title, canvas)
        cv2.imshow(title, canvas)
        cv2.setMouseCallback(title, on_click)
        while (cv2.waitKey(1) & 0xFF) != ord("q"):
            pass
        cv2.destroyWindow(title)
        if picked:
            pts = torch.tensor(picked, dtype=torch.float32)
    if grid_query:
        g = grid_queries(H, W, grid_size)
        pts = torch.cat([pts, g], 0) if len(pts) else g
    video = torch.from_numpy(np.stack(rgb, 0)).float().permute(0, 3, 1, 2).unsqueeze(0)
    frames_bgr = [np.ascontiguousarray(f[:, :, ::-1]) for f in rgb]
    return frames_bgr, video, pts
