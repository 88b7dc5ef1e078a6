"""LBM.inference (lbm.py:211-280, the loop tools/eval_pttrack.py:56 drives) at the evaluation memory sizes of
lbm/configs/lbm.yaml:41-52: ms per frame on a TAP-Vid-shaped clip (256x256 frames, T frames, N points queried in the first
frames).  usage: inference_bench.py [T] [N]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from lbm_amd.config import LBMConfig  # noqa: E402
from lbm_amd.lbm import LBM  # noqa: E402
from lbm_amd.synth import synthetic_state_dict, synthetic_video  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
video = synthetic_video(T, 256, 256, seed=7).cuda()
r = np.random.default_rng(0)
q = np.stack([r.integers(0, 8, N), r.uniform(8, 248, N), r.uniform(8, 248, N)], -1).astype(np.float32)
queries = torch.from_numpy(q).unsqueeze(0).cuda()
for M in (12, 20, 36, 48, 60):
    m = LBM(LBMConfig(input_size=(384, 512), memory_size=12).to_namespace())
    m.load_state_dict(synthetic_state_dict(m, seed=42), strict=True)
    m = m.cuda().eval()
    if M != 12:
        m.set_memory_size(M)
    m.extend_queries = False
    with torch.no_grad():
        m.inference(video[:, :12], queries.clone())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = m.inference(video, queries.clone())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert bool(torch.isfinite(out["points"]).all())
    print(f"M {M}: {1e3 * dt / T:.3f} ms per frame, {T / dt:.0f} frames/s, {N * T / dt:.3g} point-frames/s ({T} frames, {N} points)")
