"""Offline (CPU) analysis of gpurun_out/r2/stale_dumps.pt written by stale_offline.py: through which pyramid level did the
error of a wrong map cell enter (the three layers mix the levels with different weights), in which channels, and what value
must the wave have used for the normalised p1[2][2] row there?"""
import sys

import torch

D = torch.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r2/stale_dumps.pt", weights_only=True)
for d in D:
    got, want = d["got"].float(), d["want"].float()
    diff = (got[:, 0] + got[:, 1] / 2048) - (want[:, 0] + want[:, 1] / 2048)          # (layer, cell, 256)
    c = 3                                                                               # cell (1,1)
    ch = torch.nonzero(diff[:, c].abs().amax(0) > 0).flatten()
    mix = torch.tensor(d["mix"]).view(3, 4)
    mix[:, 0] += 1
    print(f"frame {d['frame']} block (y0 {d['y0']}, x0 {d['x0']}): {len(ch)} channels differ: {ch.tolist()}")
    for k in range(4):
        r = diff[:, c][:, ch] / mix[:, k:k + 1]
        print(f"   entered through level {k}? implied delta spread over the layers {float((r.max(0).values - r.min(0).values).abs().max()):.2e}")
    i22 = d["names"].index("p1[2][2]")
    n_cur = d["cur"][i22] * d["inv"][i22]
    implied = n_cur + (diff[:, c] / mix[:, 1:2]).mean(0) / 0.0625
    print("   normalised p1[2][2] in those channels, true :", [f"{float(v):+.2e}" for v in n_cur[ch][:6]])
    print("   ... and what the wave must have used        :", [f"{float(v):+.2e}" for v in implied[ch][:6]])
