"""Generates tests/golden/wrap_*.npz from the REFERENCE's own wrapper code -- lbm/models/lbm/backbone.py (encode_frames_online,
sample_queries_online), lbm/models/lbm/lbm.py (set_memory_size, inference) and lbm/models/lbm_online.py (init, update,
online_forward) -- run on the host CPU.  TEST INFRASTRUCTURE ONLY, build container only (needs /root/reference).

What is the reference here and what is a stand-in (declared):
  * backbone.py imports torchvision (absent in this container) for two things: ``transforms.Normalize`` and
    ``models.resnet18``.  Both are replaced by STAND-INS written here: Normalize = (x - mean) / std per channel, and a
    ResNet-18 stem..layer3 with the torchvision module names (conv1, bn1, relu, maxpool, layer1..3 of BasicBlocks:
    conv3x3-BN-ReLU-conv3x3-BN + identity / 1x1-conv-BN down-sample, ReLU after the sum).  So the ResNet BODY stays
    "parity unpinned" (SURVEY section 8c); everything AROUND it -- the /255 + resize + normalise call order, the three
    projections, the up-sampling, concat, fusion, position embedding and token layout of backbone.py:48-65,175-193, the query
    sampling of backbone.py:130-172, and all of lbm.py / lbm_online.py -- is the reference's own code executing.
  * ``mmcv.ops`` resolves to the reference's in-tree msdeformattn.py, ``addict`` to a placeholder that raises on use
    (as in make_golden.py / make_golden_objtrack.py).
Only OUTPUTS are stored; weights, videos and queries are regenerated from seeds on both sides (lbm_amd/synth.py).
Run:  python oracle/make_golden_wrappers.py"""
from __future__ import annotations

import importlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
REF = "/root/reference"
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

from lbm_amd.config import LBMConfig  # noqa: E402
from lbm_amd.synth import synthetic_state_dict, synthetic_video  # noqa: E402
from oracle.make_golden import import_reference  # noqa: E402


# ---- declared stand-ins for torchvision ------------------------------------------------------------------------------------
class _BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, 0, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idn = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + idn)


class _ResNet18Trunk(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = nn.Sequential(_BasicBlock(64, 64, 1), _BasicBlock(64, 64, 1))
        self.layer2 = nn.Sequential(_BasicBlock(64, 128, 2), _BasicBlock(128, 128, 1))
        self.layer3 = nn.Sequential(_BasicBlock(128, 256, 2), _BasicBlock(256, 256, 1))


class _Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = torch.tensor(mean).view(1, -1, 1, 1), torch.tensor(std).view(1, -1, 1, 1)

    def __call__(self, x):
        return (x - self.mean.to(x)) / self.std.to(x)


def install_standins():
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    md = types.ModuleType("torchvision.models")
    tr.Normalize = _Normalize
    md.resnet18 = lambda weights=None: _ResNet18Trunk()
    md.ResNet18_Weights = types.SimpleNamespace(DEFAULT=None)
    tv.transforms, tv.models = tr, md
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tr, "torchvision.models": md})
    if "addict" not in sys.modules:
        class Dict:
            def __init__(self, *a, **k):
                raise RuntimeError("addict.Dict is a placeholder here: this path must not be reached")
        m = types.ModuleType("addict")
        m.Dict = Dict
        sys.modules["addict"] = m


def import_wrappers():
    import_reference()                      # pins lbm.* to /root/reference, mmcv.ops -> the reference's msdeformattn.py
    install_standins()
    mods = [importlib.import_module(n) for n in ("lbm.models.lbm.backbone", "lbm.models.lbm.lbm", "lbm.models.lbm_online")]
    for m in mods:
        assert m.__file__.startswith(REF + "/"), m.__file__
    return mods


# ---- scenarios (inputs are functions of seeds only; tests rebuild them with wrapper_inputs()) -------------------------------
SIZE, FRAME = (96, 128), (120, 160)


def wrapper_inputs():
    H, W = FRAME
    video = synthetic_video(8, H, W, seed=31)                                   # (1, 8, 3, H, W)
    r = np.random.default_rng(5)
    N = 12
    q = np.stack([r.integers(0, 4, N), r.uniform(8, W - 8, N), r.uniform(8, H - 8, N)], -1).astype(np.float32)
    q[0, 0], q[1, 0] = 0, 5                                                     # one point from the start, one very late
    pts0 = torch.from_numpy(np.stack([r.uniform(8, W - 8, 9), r.uniform(8, H - 8, 9)], -1).astype(np.float32))
    pts1 = torch.from_numpy(np.stack([r.uniform(8, W - 8, 11), r.uniform(8, H - 8, 11)], -1).astype(np.float32))
    pts1[:4] = pts0[[0, 2, 3, 7]]                                               # survivors keep their points ...
    pts1[2] = torch.tensor([40.0, 50.0])                                        # ... except one that is re-planted
    remove = torch.tensor([0, 1, 0, 0, 1, 1, 1, 0, 1], dtype=torch.bool)        # 4 of 9 survive
    update = torch.tensor([0, 0, 1, 0] + [1] * 7, dtype=torch.bool)             # survivor 2 re-planted, 7 appended
    return dict(video=video, queries=torch.from_numpy(q).unsqueeze(0), pts0=pts0, pts1=pts1, remove=remove, update=update)


def main():
    bb_mod, lbm_mod, online_mod = import_wrappers()
    cfg = LBMConfig(input_size=SIZE, memory_size=12)
    inp = wrapper_inputs()
    video = inp["video"]
    out = {}
    with torch.no_grad():
        # (1) encoder epilogue around the stand-in body + set_memory_size + inference (lbm.py)
        m = lbm_mod.LBM(cfg.to_namespace()).eval()
        sd = synthetic_state_dict(m, seed=42)
        m.load_state_dict(sd, strict=True)
        tok = m.backbone.encode_frames_online(video[:, 1])                        # (1, hw, C)
        out["enc_tokens"] = tok[0, ::3].numpy()
        q = inp["queries"].clone()
        q[:, :, 2] = q[:, :, 2] / FRAME[0] * SIZE[0]
        q[:, :, 1] = q[:, :, 1] / FRAME[1] * SIZE[1]
        out["sample_queries"] = m.backbone.sample_queries_online(video, q)[0].numpy()
        m.extend_queries = False
        res = m.inference(video[:, :6], inp["queries"].clone())
        out["inf_points"], out["inf_visibility"] = res["points"].numpy(), res["visibility"].numpy()
        for M in (20, 48):
            m2 = lbm_mod.LBM(cfg.to_namespace()).eval()
            m2.load_state_dict(sd, strict=True)
            m2.set_memory_size(M)
            out[f"temb{M}_collision0"] = m2.transformer.layers[0].collision_time_emb.detach().numpy()
            out[f"temb{M}_stream2"] = m2.transformer.layers[2].stream_time_emb.detach().numpy()
            if M == 20:
                m2.extend_queries = False
                res = m2.inference(video[:, :6], inp["queries"].clone())
                out["inf20_points"], out["inf20_visibility"] = res["points"].numpy(), res["visibility"].numpy()
        # (2) LBM_online: init, 4 frames, update (drop 5, re-plant 1, add 7), 3 more frames
        o = online_mod.LBM_online(cfg.to_namespace()).eval()
        o.load_state_dict(sd, strict=True)
        o.init(inp["pts0"], video[:, 0])
        out["on_query0"] = o.query[0].numpy()
        rec = []
        for t in range(1, 5):
            c, v, r, f_t = o.online_forward(video[:, t])
            rec.append((c.numpy(), v.numpy(), r.numpy()))
        o.update(inp["pts1"], f_t, inp["update"], inp["remove"])
        out["on_query1"] = o.query[0].numpy()
        out["on_stream_after_update"] = o.stream_dist[0].numpy()
        out["on_vis_mask_after_update"] = o.vis_mask[0].numpy()
        out["on_mem_mask_after_update"] = o.mem_mask[0].numpy()
        for t in range(5, 8):
            c, v, r, f_t = o.online_forward(video[:, t])
            rec.append((c.numpy(), v.numpy(), r.numpy()))
        for i, (c, v, r) in enumerate(rec):
            out[f"on_coords{i}"], out[f"on_vis{i}"], out[f"on_rho{i}"] = c, v, r
        out["on_stream_final"] = o.stream_dist[0].numpy()
        out["on_collision_final"] = o.collision_dist[0].numpy()
        out["on_vis_mask_final"] = o.vis_mask[0].numpy()
        out["on_mem_mask_final"] = o.mem_mask[0].numpy()
        out["on_t"] = np.array(o.t)
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, "wrap_small.npz")
    np.savez_compressed(path, **out)
    print(path, {k: v.shape for k, v in out.items()}, os.path.getsize(path))


if __name__ == "__main__":
    main()
