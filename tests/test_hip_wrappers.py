"""The HIP path behind the reference's wrapper API against outputs of the reference's OWN backbone.py / lbm.py / lbm_online.py
(tests/golden/wrap_small.npz; see tests/test_wrapper_golden_cpu.py for what is pinned and what is a stand-in).
Bars: coordinates 0.05 px; probabilities / embeddings 1e-3 with the encoder in the chain."""
import os

import numpy as np
import pytest
import torch

from oracle.make_golden_wrappers import FRAME, SIZE, wrapper_inputs

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "wrap_small.npz")


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(GOLD, allow_pickle=False)), wrapper_inputs()


def _model(cls_name="LBM_online"):
    from lbm_amd.config import LBMConfig
    from lbm_amd.synth import synthetic_state_dict
    import lbm_amd.lbm_online as lo
    import lbm_amd.lbm as lm
    cls = lo.LBM_online if cls_name == "LBM_online" else lm.LBM
    m = cls(LBMConfig(input_size=SIZE, memory_size=12).to_namespace())
    m.load_state_dict(synthetic_state_dict(m, seed=42), strict=True)
    return m.cuda().eval()


def _close(a, b, atol):
    np.testing.assert_allclose(a.detach().float().cpu().numpy(), b, rtol=0, atol=atol)


def test_encode_frames_online_and_sample_queries_online(gold):
    g, inp = gold
    m = _model("LBM")
    with torch.no_grad():
        tok = m.backbone.encode_frames_online(inp["video"][:, 1].cuda())            # (1, hw, C)   backbone.py:175-193
        _close(tok[0, ::3], g["enc_tokens"], 1e-3)
        q = inp["queries"].clone()
        q[:, :, 2] = q[:, :, 2] / FRAME[0] * SIZE[0]
        q[:, :, 1] = q[:, :, 1] / FRAME[1] * SIZE[1]
        _close(m.backbone.sample_queries_online(inp["video"].cuda(), q.cuda())[0], g["sample_queries"], 1e-3)   # backbone.py:130-172


@pytest.mark.parametrize("M", [20, 48])
def test_set_memory_size(gold, M):
    g, _ = gold
    m = _model("LBM")
    m.set_memory_size(M)                                                             # lbm.py:70-87
    _close(m.transformer.layers[0].collision_time_emb, g[f"temb{M}_collision0"], 1e-6)
    _close(m.transformer.layers[2].stream_time_emb, g[f"temb{M}_stream2"], 1e-6)
    assert m.memory_size == M


@pytest.mark.parametrize("M", [12, 20])
def test_inference_with_late_queries(gold, M):
    g, inp = gold
    m = _model("LBM")
    if M != 12:
        m.set_memory_size(M)
    m.extend_queries = False
    with torch.no_grad():
        res = m.inference(inp["video"][:, :6].cuda(), inp["queries"].clone().cuda())  # lbm.py:211-280
    key = "inf" if M == 12 else "inf20"
    _close(res["points"], g[key + "_points"], 0.05)
    assert res["visibility"].dtype == torch.bool
    assert (res["visibility"].cpu().numpy() == g[key + "_visibility"]).mean() > 0.97   # flags within 1e-3 of the 0.8 threshold may flip


def test_online_init_update_forward(gold):
    g, inp = gold
    video = inp["video"].cuda()
    o = _model("LBM_online")
    with torch.no_grad():
        o.init(inp["pts0"].cuda(), video[:, 0])                                      # lbm_online.py:27-46
        _close(o.query[0], g["on_query0"], 1e-3)
        i = 0
        for t in range(1, 5):
            c, v, r, f_t = o.online_forward(video[:, t])                             # lbm_online.py:114-158
            _close(c, g[f"on_coords{i}"], 0.05)
            _close(r, g[f"on_rho{i}"], 1e-3)
            assert f_t.shape == (1, 256, SIZE[0] // 4, SIZE[1] // 4)
            i += 1
        o.update(inp["pts1"].cuda(), f_t, inp["update"].cuda(), inp["remove"].cuda())  # lbm_online.py:48-112
        assert o.N == 11
        _close(o.query[0], g["on_query1"], 1e-3)
        _close(o.stream_dist[0], g["on_stream_after_update"], 2e-3)
        assert (o.vis_mask[0].cpu().numpy() == g["on_vis_mask_after_update"]).all()
        assert (o.mem_mask[0].cpu().numpy() == g["on_mem_mask_after_update"]).all()
        for t in range(5, 8):
            c, v, r, f_t = o.online_forward(video[:, t])
            _close(c, g[f"on_coords{i}"], 0.05)
            _close(r, g[f"on_rho{i}"], 1e-3)
            i += 1
    _close(o.stream_dist[0], g["on_stream_final"], 2e-3)
    _close(o.collision_dist[0], g["on_collision_final"], 2e-3)
    assert (o.mem_mask[0].cpu().numpy() == g["on_mem_mask_final"]).all()
    assert (o.vis_mask[0].cpu().numpy() == g["on_vis_mask_final"]).mean() > 0.99
    assert o.t == int(g["on_t"])
