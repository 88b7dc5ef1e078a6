"""Frame pipeline screen: serial vs pipelined OUTPUTS (coordinates, visibility logits, rho, tokens), bit for bit, for several
settings of the debug knobs interleaved on one box.  usage: pipe_screen.py reps name=key:val+key:val [name=...]
e.g.  pipe_screen.py 60 default= claim=12:1 packedcorr=9:1 serial_ln=11:1"""
import sys
import time

import torch

sys.path.insert(0, ".")
from lbm_amd import hip  # noqa: E402
from lbm_amd.config import LBMConfig  # noqa: E402
from lbm_amd.lbm_online import LBM_online  # noqa: E402
from lbm_amd.synth import grid_queries, synthetic_state_dict, synthetic_video  # noqa: E402

reps = int(sys.argv[1])
configs = {}
for spec in sys.argv[2:]:
    name, _, kv = spec.partition("=")
    configs[name] = [(int(x.split(":")[0]), int(x.split(":")[1])) for x in kv.split("+") if x]
DEFAULTS = {9: 0, 11: 0, 12: 0, 7: 2048, 8: 63}
lib = hip.load()
T, H, W = 16, 240, 320
video = synthetic_video(T, H, W, seed=21).cuda()
frames = [video[:, t].contiguous() for t in range(T)]
pts = grid_queries(H, W, 32).cuda()


def run(pipelined, knobs):
    for k, v in DEFAULTS.items():
        lib.lbm_hip_debug_set(k, v)
    for k, v in knobs:
        lib.lbm_hip_debug_set(k, v)
    model = LBM_online(LBMConfig(input_size=(384, 512), memory_size=12).to_namespace())
    model.load_state_dict(synthetic_state_dict(model, seed=42), strict=True)
    model = model.cuda().eval()
    out = []
    with torch.no_grad():
        model.init(pts, frames[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(1, T):
            if pipelined and t + 1 < T:
                model.prefetch(frames[t + 1])
            c, v, r, f_t = model.online_forward(frames[t])
            out.append((c.clone(), model.last_logits[0][0].clone(), r.clone(), f_t.clone()))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return out, dt / (T - 1)


refs = {n: run(False, k)[0] for n, k in configs.items()}
bad = {n: 0 for n in configs}
cnt = {n: 0 for n in configs}
tm = {n: [] for n in configs}
for rep in range(reps):
    for n, k in configs.items():
        got, dt = run(True, k)
        tm[n].append(dt)
        for a, b in zip(refs[n], got):
            cnt[n] += 1
            bad[n] += 0 if all(torch.equal(x, y) for x, y in zip(a, b)) else 1
print("frames whose pipelined outputs differ from the serial run:", {n: f"{bad[n]}/{cnt[n]}" for n in configs})
print("median ms per frame (host-timed, with clones):", {n: round(1e3 * sorted(tm[n])[len(tm[n]) // 2], 3) for n in configs})
