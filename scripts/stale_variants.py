"""Frame pipeline: rate of wrong correlation-map cells per kernel variant, variants interleaved on ONE box (the rate depends on
the box).  usage: stale_variants.py reps v1,v2,...   (debug key 9 values; 0 / 5 = the product 2x2-block kernel (no packed-fp32 instructions), 6 = cell-by-cell kernel, 1 = block kernel built with packed-fp32,
3 = block kernel with every loaded row tied behind its wait, 4 = the same + 16 idle cycles after each wait)"""
import sys

import torch

sys.path.insert(0, ".")
from lbm_amd import hip  # noqa: E402
from lbm_amd.config import LBMConfig  # noqa: E402
from lbm_amd.lbm_online import LBM_online  # noqa: E402
from lbm_amd.synth import grid_queries, synthetic_state_dict, synthetic_video  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
variants = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,3,4").split(",")]
lib = hip.load()
T, H, W = 16, 240, 320
video = synthetic_video(T, H, W, seed=21).cuda()
frames = [video[:, t].contiguous() for t in range(T)]
pts = grid_queries(H, W, 32).cuda()
P0 = 96 * 128


def run(pipelined, variant):
    lib.lbm_hip_debug_set(9, variant)
    model = LBM_online(LBMConfig(input_size=(384, 512), memory_size=12).to_namespace())
    model.load_state_dict(synthetic_state_dict(model, seed=42), strict=True)
    model = model.cuda().eval()
    eng = model.transformer.get_engine()
    rec = []
    with torch.no_grad():
        model.init(pts, frames[0])
        for t in range(1, T):
            if pipelined and t + 1 < T:
                model.prefetch(frames[t + 1])
            model.online_forward(frames[t])
            torch.cuda.synchronize()
            rec.append(eng.debug_buffer("corr_maps", (3, 2, P0, 256), torch.float16).clone())
    return rec


refs = {v: run(False, v) for v in variants}
for v in variants[1:]:
    same = all(torch.equal(a, b) for a, b in zip(refs[variants[0]], refs[v]))
    print(f"serial maps of variant {v} equal variant {variants[0]}'s: {same}")
bad = {v: 0 for v in variants}
n = {v: 0 for v in variants}
for rep in range(reps):
    for v in variants:
        got = run(True, v)
        for a, b in zip(refs[v], got):
            n[v] += 1
            bad[v] += 0 if torch.equal(a, b) else 1
    if rep % 10 == 9:
        print(f"after {rep + 1} reps:", {v: f"{bad[v]}/{n[v]}" for v in variants}, flush=True)
print("TOTAL wrong frames per variant:", {v: f"{bad[v]}/{n[v]}" for v in variants})
