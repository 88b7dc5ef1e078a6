"""Is lbm_hip_ffn_slabs deterministic while another stream keeps the chip busy?  Bitwise comparison over many launches."""
import sys

import torch

sys.path.insert(0, ".")
from lbm_amd import hip  # noqa: E402

torch.manual_seed(0)
dev = "cuda"
M = 1024
W1 = torch.randn(1024, 256, device=dev) / 16
b1 = torch.randn(1024, device=dev) * 0.1
W2 = torch.randn(256, 1024, device=dev) / 32
s1, s2 = hip.SplitWeight(W1), hip.SplitWeight(W2)
f1, f2 = hip.fragment_major(s1), hip.fragment_major(s2)
x = torch.randn(M, 256, device=dev)
xs = hip.SplitTensor((M, 256), dev)
hip._check(hip.load().lbm_hip_split_planes(x.data_ptr(), xs.hi, xs.planes[1].data_ptr(), x.numel(), hip._stream()), "split")
ref = hip.ffn_slabs(xs, f1, b1, f2, 2).clone()
torch.cuda.synchronize()
# background load on a second stream: big split products (LDS-heavy tiles) and streaming copies
bg = torch.cuda.Stream()
A = torch.randn(16320, 256, device=dev)
As = hip.SplitTensor((16320, 256), dev)
hip._check(hip.load().lbm_hip_split_planes(A.data_ptr(), As.hi, As.planes[1].data_ptr(), A.numel(), hip._stream()), "split")
Wb = hip.SplitWeight(torch.randn(1536, 256, device=dev) / 16)
big = torch.empty(64 << 20, device=dev)
torch.cuda.synchronize()
mode = sys.argv[1] if len(sys.argv) > 1 else "busy"
bad = 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
outs = []
for it in range(N):
    if mode == "busy" and it % 4 == 0:
        with torch.cuda.stream(bg):
            hip.linear_ss(As, Wb, None)
            big.mul_(1.0001)
    outs.append(hip.ffn_slabs(xs, f1, b1, f2, 2))
    if len(outs) == 50:
        for o in outs:
            neq = o != ref
            if bool(neq.any()):
                bad += 1
                idx = torch.nonzero(neq)
                sl, rows, cols = idx[:, 0].unique().tolist(), idx[:, 1].unique().tolist(), idx[:, 2].unique().tolist()
                print(f"mismatch: slabs {sl[:8]} rows {rows[:10]} ({len(rows)}) cols {cols[:10]} ({len(cols)}), max diff {float((o - ref)[neq].abs().max()):.3g}", flush=True)
        outs = []
torch.cuda.synchronize()
print(f"{mode}: {bad} of {N} launches differ from the first result")
