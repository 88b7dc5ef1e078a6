"""Serial vs pipelined frames: where do the outputs first differ?  (debugging aid for the bit-identity test)"""
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from lbm_amd import hip  # noqa: E402
from lbm_amd.config import LBMConfig  # noqa: E402
from lbm_amd.lbm_online import LBM_online  # noqa: E402
from lbm_amd.synth import grid_queries, synthetic_state_dict, synthetic_video  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
hip.load().lbm_hip_debug_set(7, rows)
if len(sys.argv) > 3:
    hip.load().lbm_hip_debug_set(8, int(sys.argv[3]))
if len(sys.argv) > 4:
    hip.load().lbm_hip_debug_set(9, int(sys.argv[4]))
T, H, W = 16, 240, 320
video = synthetic_video(T, H, W, seed=21).cuda()
frames = [video[:, t].contiguous() for t in range(T)]
pts = grid_queries(H, W, 32).cuda()
runs = {}
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 1
MODES = ["serial"] + [f"pipelined{i}" for i in range(REPS)] + ["serial2"]
for mode in MODES:
    model = LBM_online(LBMConfig(input_size=(384, 512), memory_size=12).to_namespace())
    model.load_state_dict(synthetic_state_dict(model, seed=42), strict=True)
    model = model.cuda().eval()
    out = []
    with torch.no_grad():
        model.init(pts, frames[0])
        for t in range(1, T):
            if mode.startswith("pipelined") and t + 1 < T:
                model.prefetch(frames[t + 1])
            c, v, r, f_t = model.online_forward(frames[t])
            out.append((c.clone(), model.last_logits[0][0].clone(), r.clone(), f_t.clone()))
    torch.cuda.synchronize()
    runs[mode] = out
for other in MODES[1:]:
    print("serial vs", other)
    for t, (a, b) in enumerate(zip(runs["serial"], runs[other])):
        d = [float((x.float() - y.float()).abs().max()) for x, y in zip(a, b)]
        n = int(((a[0] - b[0]).abs().amax(-1) > 0).sum())
        if n or max(d) > 0:
            print("   rows", torch.nonzero((a[0] - b[0]).abs().amax(-1) > 0).flatten().tolist(), "vis rows", torch.nonzero((a[1] - b[1]).abs() > 0).flatten().tolist())
            print(f"  frame {t + 1}: max diff coords {d[0]:.3g} vis {d[1]:.3g} rho {d[2]:.3g} feat {d[3]:.3g}; points differing {n}")
