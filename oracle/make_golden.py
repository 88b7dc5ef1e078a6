"""Generates tests/golden/*.npz from the REFERENCE itself.  TEST INFRASTRUCTURE ONLY — runs in the build
container only (needs /root/reference); nothing under tests/, bench.py or smoke() imports this file.

What is imported from the reference (verified importable, nothing is copied into the repo):
  * lbm/models/lbm/msdeformattn.py  — as-is (torch only);
  * lbm/utils/coord_utils.py        — as-is;
  * lbm/models/lbm/transformer.py   — with ``mmcv.ops`` resolved to the reference's own msdeformattn module
    (mmcv is not installed; this is the variant the reference's published PyTorch numbers use,
    assets/tensorrt.md:112).
lbm.py / lbm_online.py / backbone.py need torchvision and are NOT importable: the state roll
(lbm.py:256-259 == lbm_online.py:143-146) is restated in ``_roll`` below, four torch.cat calls.

The fixtures hold only the reference's OUTPUTS; weights and inputs are regenerated from seeds by
oracle/fixtures.py on both sides.  Run:  python oracle/make_golden.py
"""
from __future__ import annotations

import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
REF = "/root/reference"

from oracle import fixtures as fx  # noqa: E402


def pin_reference_packages():
    """The repo carries an import-path twin of the reference's package (``lbm/``, for drop-in callers): make sure the
    names below resolve to the REFERENCE's directories and nothing else."""
    for name in [m for m in sys.modules if m == "lbm" or m.startswith("lbm.")]:
        del sys.modules[name]
    for name, sub in (("lbm", "lbm"), ("lbm.models", "lbm/models"), ("lbm.models.lbm", "lbm/models/lbm"), ("lbm.utils", "lbm/utils")):
        pkg = types.ModuleType(name)
        pkg.__path__ = [os.path.join(REF, sub)]
        sys.modules[name] = pkg


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("make_golden.py needs /root/reference (build container only)")
    pin_reference_packages()
    msda = importlib.import_module("lbm.models.lbm.msdeformattn")
    mmcv = types.ModuleType("mmcv")
    ops = types.ModuleType("mmcv.ops")
    ops.MultiScaleDeformableAttention = msda.MultiScaleDeformableAttention
    mmcv.ops = ops
    sys.modules["mmcv"], sys.modules["mmcv.ops"] = mmcv, ops
    tr = importlib.import_module("lbm.models.lbm.transformer")
    cu = importlib.import_module("lbm.utils.coord_utils")
    for m in (msda, tr, cu):
        assert m.__file__.startswith(REF + "/"), m.__file__
    return msda, tr, cu


def build_ref_transformer(tr_mod, g, seed=42):
    m = tr_mod.LBMTransformer(size=list(g.size), stride=g.stride, n_layer=g.num_layers, hidden_dim=g.embed_dim,
                              n_head=g.num_heads, n_level=g.num_levels, n_neighbor=g.num_neighbors, n_mem=g.memory_size,
                              n_topk=g.top_k_regions)
    sd = fx.transformer_state_dict(g, seed)
    m.load_state_dict(sd, strict=True)
    return m.eval(), sd


def _np(t):
    return t.detach().cpu().numpy()


def _roll(state, memory, vis, queried, thr=0.8):
    """lbm.py:256-259 / lbm_online.py:143-146."""
    coll, stream, mem_mask, vis_mask = state
    coll = torch.cat([coll[:, :, 1:], memory["collision"].unsqueeze(2)], dim=2)
    stream = torch.cat([stream[:, :, 1:], memory["stream"].unsqueeze(2)], dim=2)
    mem_mask = torch.cat([mem_mask[:, :, 1:], ~queried.unsqueeze(-1)], dim=2)
    vis_mask = torch.cat([vis_mask[:, :, 1:], (torch.sigmoid(vis) < thr).unsqueeze(-1)], dim=2)
    return coll, stream, mem_mask, vis_mask


@torch.no_grad()
def gen_step(tr_mod, name, store_corr=True):
    spec = fx.GEOMS[name]
    g = fx.geometry(name)
    model, _ = build_ref_transformer(tr_mod, g)
    inp = fx.step_inputs(g, spec["B"], spec["N"])
    out, memory, offset, vis, rho = model(inp["query"].clone(), inp["feat"], inp["stream"].clone(), inp["collision"].clone(),
                                          inp["vis_mask"], inp["mem_mask"], inp["queried"])
    d = {"offset": _np(offset), "vis": _np(vis), "rho": _np(rho), "mem_stream": _np(memory["stream"]),
         "mem_collision": _np(memory["collision"])}
    for i, o in enumerate(out):
        corr = o["corr"]
        k = o["reference_points"].shape[2]
        top = torch.topk(corr, min(k + 1, corr.shape[-1]), dim=-1)
        d[f"ref{i}"] = _np(o["reference_points"])
        d[f"rho{i}"] = _np(o["reference_rho"])
        d[f"topv{i}"] = _np(top[0])       # k+1 best values: the k-th/k+1-th margin tells how stable the selection is
        d[f"topi{i}"] = _np(top[1]).astype(np.int32)
        if store_corr:
            d[f"corr{i}"] = _np(corr)
        else:
            d[f"corr_s{i}"] = _np(corr[..., ::13])
    return d


@torch.no_grad()
def gen_msda(msda_mod, name):
    """MultiScaleDeformableAttention.forward (P = 9 and P = 4) and the bare sampling core."""
    g = fx.geometry(name)
    d = {}
    for P in (9, 4):
        torch.manual_seed(0)
        m = msda_mod.MultiScaleDeformableAttention(g.embed_dim, g.num_heads, g.num_levels, P, dropout=0.1, batch_first=True).eval()
        from lbm_amd.synth import synthetic_state_dict
        sd = synthetic_state_dict(m, seed=100 + P)
        m.load_state_dict(sd, strict=True)
        inp = fx.msda_inputs(g, 2, 24, seed=13 + P)
        shapes = torch.tensor(g.level_shapes, dtype=torch.long)
        starts = torch.tensor(g.level_starts, dtype=torch.long)
        y = m(inp["query"], inp["value"], inp["value"], reference_points=inp["reference_points"], spatial_shapes=shapes,
              level_start_index=starts)
        d[f"out_p{P}"] = _np(y)
    return d


@torch.no_grad()
def gen_corr(tr_mod, name):
    spec = fx.GEOMS[name]
    g = fx.geometry(name)
    model, _ = build_ref_transformer(tr_mod, g)
    inp = fx.step_inputs(g, spec["B"], spec["N"], seed=21)
    feats = model.get_multiscale_feats(inp["feat"])
    c = model.layers[1].correlation(inp["query"], feats, size=(g.h, g.w))
    return {"corr": _np(c), "pyr1": _np(feats[1]), "pyr3": _np(feats[3])}


@torch.no_grad()
def gen_sequence(tr_mod, name, T=7):
    """T frames of transformer + roll: frame 0 all-masked, memory fills past M, one point queried late (t = 2),
    one never queried within the window."""
    spec = fx.GEOMS[name]
    g = fx.geometry(name)
    B, N, M, C = spec["B"], spec["N"], g.memory_size, g.embed_dim
    model, _ = build_ref_transformer(tr_mod, g)
    feats = fx.sequence_feats(g, B, T)
    inp = fx.step_inputs(g, B, N, seed=33)
    q_init = inp["query"]
    times = torch.zeros(B, N, dtype=torch.long)
    times[:, 5] = 2
    times[:, 6] = T + 3
    state = (torch.zeros(B, N, M, C), torch.zeros(B, N, M, C), torch.ones(B, N, M, dtype=torch.bool),
             torch.ones(B, N, M, dtype=torch.bool))
    coords, viss, rhos = [], [], []
    for t in range(T):
        queried = times <= t
        coll, stream, mem_mask, vis_mask = state
        out, memory, offset, vis, rho = model(q_init.clone(), feats[t], stream.clone(), coll.clone(), vis_mask, mem_mask, queried)
        state = _roll(state, memory, vis, queried)
        coords.append(out[-1]["reference_points"].squeeze(2) + offset)
        viss.append(vis)
        rhos.append(rho)
    coll, stream, mem_mask, vis_mask = state
    return {"coords": _np(torch.stack(coords, 1)), "vis": _np(torch.stack(viss, 1)), "rho": _np(torch.stack(rhos, 1)),
            "final_stream": _np(stream), "final_collision": _np(coll), "final_vis_mask": _np(vis_mask),
            "final_mem_mask": _np(mem_mask), "times": _np(times)}


def gen_coords(cu_mod):
    idx = torch.tensor([[[0, 1, 127, 128, 12287, 6000, 777]]])
    return {"idx": _np(idx), "coords": _np(cu_mod.indices_to_coords(idx, [384, 512], 4))}


def main():
    msda_mod, tr_mod, cu_mod = import_reference()
    out_dir = sys.argv[1] if len(sys.argv) > 1 else fx.GOLDEN_DIR      # another directory: regenerate and compare by hand
    os.makedirs(out_dir, exist_ok=True)
    jobs = {
        "step_tiny64": lambda: gen_step(tr_mod, "tiny64"),
        "step_small256": lambda: gen_step(tr_mod, "small256"),
        "step_full256": lambda: gen_step(tr_mod, "full256", store_corr=False),
        "msda_small256": lambda: gen_msda(msda_mod, "small256"),
        "corr_small256": lambda: gen_corr(tr_mod, "small256"),
        "seq_tiny64": lambda: gen_sequence(tr_mod, "tiny64"),
        "seq_small256": lambda: gen_sequence(tr_mod, "small256"),
        "coords": lambda: gen_coords(cu_mod),
    }
    for name, fn in jobs.items():
        d = fn()
        path = os.path.join(out_dir, name + ".npz")
        np.savez_compressed(path, **d)
        print(f"{name}: {len(d)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
