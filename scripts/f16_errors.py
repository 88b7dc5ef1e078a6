"""Errors of the reduced-precision (f16 hi-plane) mode against the CPU oracle on the published geometry."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_hip_parity as T
from oracle import fixtures as fx, lbm_ref_cpu as ref

g = fx.geometry("full256")
inp = fx.step_inputs(g, 2, 300, seed=77)
for prec in ("split", "f16", "f16s"):
    tr, sd = T._transformer(g, precision=prec)
    with torch.no_grad():
        want = ref.transformer_forward(sd, g, inp["query"], inp["feat"], inp["stream"], inp["collision"], inp["vis_mask"],
                                       inp["mem_mask"], inp["queried"], prefix="")
        cu = {k: v.cuda() for k, v in inp.items()}
        got = tr(cu["query"], cu["feat"], cu["stream"], cu["collision"], cu["vis_mask"], cu["mem_mask"], cu["queried"])
    w_out, w_mem, w_off, w_vis, w_rho = want
    g_out, g_mem, g_off, g_vis, g_rho = got
    same = (g_out[-1]["reference_points"].cpu() == w_out[-1]["reference_points"]).all(-1).all(-1)
    allsame = torch.ones_like(same)
    for li in range(len(w_out)):
        a = g_out[li]["reference_points"].cpu(); b = w_out[li]["reference_points"]
        key = lambda t: (t[..., 1] * 1000 + t[..., 0]).sort(-1).values
        eq = (key(a) == key(b)).all(-1)
        print("   layer", li, "same top-k set:", eq.float().mean().item())
        allsame &= eq
    d = lambda x, y: (x.cpu() - y)[allsame].abs().max().item()
    print("   on points whose sets agree in every layer (%.3f): offset %.3g vis %.3g rho %.3g mem %.3g" % (
        allsame.float().mean().item(), d(g_off, w_off), d(g_vis, w_vis), d(g_rho, w_rho), d(g_mem["stream"], w_mem["stream"])))
    print(prec, "same final cell:", same.float().mean().item(),
          "| offset", (g_off.cpu() - w_off).abs().max().item(), "vis", (g_vis.cpu() - w_vis).abs().max().item(),
          "rho", (g_rho.cpu() - w_rho).abs().max().item(), "mem", (g_mem["stream"].cpu() - w_mem["stream"]).abs().max().item(),
          "| on same cells: offset", (g_off.cpu() - w_off)[same].abs().max().item())
