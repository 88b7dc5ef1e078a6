"""The CPU oracle's WRAPPER-level restatements (encoder epilogue, query sampling, set_memory_size, LBM.inference,
LBM_online.init / update / online_forward) against outputs of the reference's own backbone.py / lbm.py / lbm_online.py
(tests/golden/wrap_small.npz, written by oracle/make_golden_wrappers.py; the ResNet-18 body inside is a declared stand-in
for torchvision and stays parity-unpinned, everything around it is the reference's code)."""
import os

import numpy as np
import pytest
import torch

from oracle import lbm_ref_cpu as ref
from oracle.make_golden_wrappers import FRAME, SIZE, wrapper_inputs

GOLD = os.path.join(os.path.dirname(__file__), "golden", "wrap_small.npz")


@pytest.fixture(scope="module")
def setup():
    from lbm_amd.config import LBMConfig
    from lbm_amd.lbm_online import LBM_online
    from lbm_amd.synth import synthetic_state_dict
    torch.set_num_threads(8)
    model = LBM_online(LBMConfig(input_size=SIZE, memory_size=12).to_namespace())
    sd = synthetic_state_dict(model, seed=42)
    return sd, dict(np.load(GOLD, allow_pickle=False)), wrapper_inputs()


def _close(a, b, atol):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else a
    np.testing.assert_allclose(a, b, rtol=0, atol=atol)


def test_encoder_epilogue_and_query_sampling(setup):
    sd, gold, inp = setup
    g = ref.Geometry(size=SIZE, memory_size=12)
    with torch.no_grad():
        feat = ref.encode_frames(sd, g, inp["video"][:, 1])                      # (1,C,h,w)
        tok = feat.permute(0, 2, 3, 1).reshape(g.h * g.w, -1)
        _close(tok[::3], gold["enc_tokens"], 2e-5)
        q = inp["queries"][0]
        want = gold["sample_queries"]
        for i in range(q.shape[0]):
            f = ref.encode_frames(sd, g, inp["video"][:, int(q[i, 0])])
            px = torch.stack([q[i, 1] / FRAME[1] * SIZE[1], q[i, 2] / FRAME[0] * SIZE[0]]).view(1, 2)
            _close(ref.sample_queries(f, px, g)[0], want[i], 2e-5)


@pytest.mark.parametrize("M", [20, 48])
def test_set_memory_size(setup, M):
    sd, gold, _ = setup
    g = ref.Geometry(size=SIZE, memory_size=12)
    sd = {k: v.clone() for k, v in sd.items()}
    ref.resize_time_embeddings(sd, g, M)
    _close(sd["transformer.layers.0.collision_time_emb"], gold[f"temb{M}_collision0"], 1e-7)
    _close(sd["transformer.layers.2.stream_time_emb"], gold[f"temb{M}_stream2"], 1e-7)


@pytest.mark.parametrize("M", [12, 20])
def test_inference_with_late_queries(setup, M):
    sd, gold, inp = setup
    g = ref.Geometry(size=SIZE, memory_size=12)
    sd = {k: v.clone() for k, v in sd.items()}
    if M != 12:
        ref.resize_time_embeddings(sd, g, M)
    with torch.no_grad():
        pts, vis = ref.inference(sd, g, inp["video"][:, :6], inp["queries"].clone())
    key = "inf" if M == 12 else "inf20"
    _close(pts, gold[key + "_points"], 2e-3)
    stable = (torch.sigmoid(vis) - 0.8).abs() > 1e-4
    assert ((torch.sigmoid(vis) > 0.8).numpy() == gold[key + "_visibility"])[stable.numpy()].all()


def test_online_init_update_forward(setup):
    sd, gold, inp = setup
    g = ref.Geometry(size=SIZE, memory_size=12)
    video = inp["video"]
    o = ref.RefOnline(sd, g)
    with torch.no_grad():
        o.init(inp["pts0"], video[:, 0])
        _close(o.state.query[0], gold["on_query0"], 2e-5)
        i = 0
        for t in range(1, 5):
            c, v, r, feat, vl, rl = o.online_forward(video[:, t])
            _close(c, gold[f"on_coords{i}"], 2e-3)
            _close(r, gold[f"on_rho{i}"], 1e-4)
            assert (v.numpy() == gold[f"on_vis{i}"])[((torch.sigmoid(vl) - 0.8).abs() > 1e-4).numpy()].all()
            i += 1
        o.update(inp["pts1"], feat, inp["update"], inp["remove"])
        _close(o.state.query[0], gold["on_query1"], 2e-5)
        _close(o.state.stream[0], gold["on_stream_after_update"], 1e-4)
        assert (o.state.vis_mask[0].numpy() == gold["on_vis_mask_after_update"]).all()
        assert (o.state.mem_mask[0].numpy() == gold["on_mem_mask_after_update"]).all()
        for t in range(5, 8):
            c, v, r, feat, vl, rl = o.online_forward(video[:, t])
            _close(c, gold[f"on_coords{i}"], 2e-3)
            _close(r, gold[f"on_rho{i}"], 1e-4)
            i += 1
    _close(o.state.stream[0], gold["on_stream_final"], 2e-4)
    _close(o.state.collision[0], gold["on_collision_final"], 2e-4)
    assert (o.state.vis_mask[0].numpy() == gold["on_vis_mask_final"]).all()
    assert (o.state.mem_mask[0].numpy() == gold["on_mem_mask_final"]).all()
    assert o.state.t == int(gold["on_t"])
