"""Frame pipeline vs serial frames: which stage differs first?  After every frame (device synchronised) the frame-level
products of the step -- correlation maps and value maps, read from the workspace -- and the outputs are compared bit for
bit with the serial run's.  usage: pipe_f.py [reps] [corr variant]"""
import sys

import torch

sys.path.insert(0, ".")
from lbm_amd import hip  # noqa: E402
from lbm_amd.config import LBMConfig  # noqa: E402
from lbm_amd.lbm_online import LBM_online  # noqa: E402
from lbm_amd.synth import grid_queries, synthetic_state_dict, synthetic_video  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
hip.load().lbm_hip_debug_set(9, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
T, H, W = 16, 240, 320
video = synthetic_video(T, H, W, seed=21).cuda()
frames = [video[:, t].contiguous() for t in range(T)]
pts = grid_queries(H, W, 32).cuda()


def run(pipelined):
    model = LBM_online(LBMConfig(input_size=(384, 512), memory_size=12).to_namespace())
    model.load_state_dict(synthetic_state_dict(model, seed=42), strict=True)
    model = model.cuda().eval()
    eng = model.transformer.get_engine()
    cfg = model.transformer.cfg
    P0, L = cfg.num_cells, cfg.num_tokens
    rec = []
    with torch.no_grad():
        model.init(pts, frames[0])
        for t in range(1, T):
            if pipelined and t + 1 < T:
                model.prefetch(frames[t + 1])
            c, v, r, f_t = model.online_forward(frames[t])
            torch.cuda.synchronize()
            F = eng.debug_buffer("corr_maps", (3, 2, P0, 256), torch.float16).clone()
            V = eng.debug_buffer("value", (L, 6 * 256)).clone()
            rec.append((F, V, c.clone(), model.last_logits[0][0].clone()))
    return rec


ref = run(False)
nF = nV = nO = n = 0
for rep in range(reps):
    got = run(True)
    for t, (a, b) in enumerate(zip(ref, got)):
        n += 1
        dF, dV = not torch.equal(a[0], b[0]), not torch.equal(a[1], b[1])
        dO = not (torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]))
        nF += dF; nV += dV; nO += dO
        if dF or dV or dO:
            cells = torch.nonzero((a[0].float() - b[0].float()).abs().amax((0, 1, 3)) > 0).flatten().tolist()
            print(f"rep {rep} frame {t + 2}: F differs {dF} (cells {cells[:10]}{'...' if len(cells) > 10 else ''}, {len(cells)} in all), value differs {dV}, outputs differ {dO}")
print(f"{n} frames: correlation maps differ in {nF}, value maps in {nV}, outputs in {nO}")
