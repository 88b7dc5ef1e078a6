"""Frame pipeline: is the token pyramid final when the encoder stream moves on?  A clone taken on the encoder stream right
after encode_tokens is compared with the same buffer after a device synchronisation.  usage: pipe_snap.py [reps] [variant]"""
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
bad_tok = bad_split = total = 0
for rep in range(reps):
    model = LBM_online(LBMConfig(input_size=(384, 512), memory_size=12).to_namespace())
    model.load_state_dict(synthetic_state_dict(model, seed=42), strict=True)
    model = model.cuda().eval()
    snaps = []
    orig = model.backbone.encode_tokens

    def wrapped(fr, out=None, split_out=None, _orig=orig):
        tok = _orig(fr, out, split_out=split_out)
        if split_out is not None:
            snaps.append((tok, tok.clone(), split_out.planes, split_out.planes.clone()))
        return tok

    model.backbone.encode_tokens = wrapped
    with torch.no_grad():
        model.init(pts, frames[0])
        for t in range(1, T):
            if t + 1 < T:
                model.prefetch(frames[t + 1])
            model.online_forward(frames[t])
            torch.cuda.synchronize()
            for tok, snap, sp, sps in snaps:
                total += 1
                if not torch.equal(tok, snap):
                    bad_tok += 1
                    d = (tok - snap).abs().amax(-1)[0]
                    print(f"rep {rep} frame {t + 1}: tokens changed after the clone in rows", torch.nonzero(d > 0).flatten().tolist()[:12], float(d.max()))
                if not torch.equal(sp, sps):
                    bad_split += 1
            snaps.clear()
print(f"{total} encoder outputs: {bad_tok} fp32 pyramids and {bad_split} split twins differ from the clone taken on the encoder stream")
