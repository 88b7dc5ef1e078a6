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


def test_uint8_frames_track_bit_identically_to_float_frames(gold):
    """Video ingest: frames handed over as the decoder's uint8 bytes -- planar through the model API,
    interleaved RGB / BGR at the operator -- give exactly the bits of the reference-style float frames (demo_utils.py:127-131)."""
    from lbm_amd import hip
    _, inp = gold
    video = inp["video"].cuda()                                   # integer-valued floats in [0, 255]
    v8 = video.to(torch.uint8)
    assert torch.equal(v8.float(), video)
    want = hip.preprocess(video[0, :2], SIZE[0], SIZE[1], layout=4)
    assert torch.equal(hip.preprocess(v8[0, :2], SIZE[0], SIZE[1], layout=4), want)
    hwc = v8[0, :2].permute(0, 2, 3, 1).contiguous()
    assert torch.equal(hip.preprocess(hwc, SIZE[0], SIZE[1], layout=4, u8_layout="hwc_rgb"), want)
    assert torch.equal(hip.preprocess(hwc.flip(-1).contiguous(), SIZE[0], SIZE[1], layout=4, u8_layout="hwc_bgr"), want)
    planar = hip.preprocess(video[0, :2], SIZE[0], SIZE[1], layout=0)
    assert torch.equal(hip.preprocess(v8[0, :2], SIZE[0], SIZE[1], layout=0), planar)
    # the pixel-major kernel (one thread per pixel) against the element-wise one: same bits, fourth channel zero
    assert torch.equal(want[..., :3], planar.permute(0, 2, 3, 1)) and not want[..., 3].any()
    a, b = _model("LBM_online"), _model("LBM_online")
    with torch.no_grad():
        a.init(inp["pts0"].cuda(), video[:, 0])
        b.init(inp["pts0"].cuda(), v8[:, 0])
        for t in range(1, 5):
            ca, va, ra, _ = a.online_forward(video[:, t])
            cb, vb, rb, _ = b.online_forward(v8[:, t])
            assert torch.equal(ca, cb) and torch.equal(va, vb) and torch.equal(ra, rb)


def test_interleaved_uint8_frames_through_the_model_api(gold):
    """A video decoder's (H, W, 3) bytes, RGB or BGR, straight into init / prefetch / online_forward (`frame_layout`): bit-equal to the
    reference-style float (3, H, W) frames, frame size taken from the declared layout."""
    _, inp = gold
    video = inp["video"].cuda()
    hwc = video.to(torch.uint8).permute(0, 1, 3, 4, 2).contiguous()              # (1, T, H, W, 3)
    a = _model("LBM_online")
    with torch.no_grad():
        a.init(inp["pts0"].cuda(), video[:, 0])
        want = [a.online_forward(video[:, t])[:3] for t in range(1, 5)]
        for layout, frames in (("hwc_rgb", hwc), ("hwc_bgr", hwc.flip(-1).contiguous())):
            b = _model("LBM_online")
            b.frame_layout = layout
            b.init(inp["pts0"].cuda(), frames[:, 0])
            assert tuple(b.img_size) == tuple(video.shape[-2:])
            for t in range(1, 5):
                if t % 2 == 0:
                    b.prefetch(frames[:, t])                                        # the pipelined path reads the same layout
                cb, vb, rb, f_t = b.online_forward(frames[:, t])
                ca, va, ra = want[t - 1]
                assert torch.equal(ca, cb) and torch.equal(va, vb) and torch.equal(ra, rb), (layout, t)
            assert b.pipelined_frames == 2
        with pytest.raises(ValueError):
            a.frame_layout = "nhwc"


def test_prefetch_and_online_forward_under_inference_mode(gold):
    """torch.inference_mode() tensors have no version counter; prefetch / online_forward / the graph key must work with them and
    give the tracks of the no_grad run."""
    _, inp = gold
    video = inp["video"].cuda()
    a, b = _model("LBM_online"), _model("LBM_online")
    with torch.no_grad():
        a.init(inp["pts0"].cuda(), video[:, 0])
        want = [a.online_forward(video[:, t])[0].clone() for t in range(1, 5)]
    with torch.inference_mode():
        vid = inp["video"].cuda()                                # an inference tensor
        b.init(inp["pts0"].cuda(), vid[:, 0])
        for t in range(1, 5):
            b.prefetch(vid[:, t])
            c = b.online_forward(vid[:, t])[0]
            assert torch.equal(c, want[t - 1]), t
        assert b.pipelined_frames == 4


def test_two_trackers_driven_from_two_host_threads():
    """Two independent LBM_online trackers, each on its own HIP stream, each driven by its own host thread at the same time, against
    their own serial runs: bit for bit.  (The library keeps no state per call; what is shared -- the name table of the profiler, the
    fork / join events of the sub-stream path, the knob table -- is behind locks.)  Also the widest single check of the kernels
    running beside a second queue."""
    import threading
    from lbm_amd.synth import grid_queries, random_queries, synthetic_video
    H, W, T = 240, 320, 10
    vids = [synthetic_video(T, H, W, seed=70 + i).cuda() for i in range(2)]
    pts = [grid_queries(H, W, 12).cuda(), random_queries(300, H, W, 7).cuda()]

    def run(i, stream, out):
        m = _model("LBM_online")
        with torch.no_grad(), torch.cuda.stream(stream):
            m.init(pts[i], vids[i][:, 0])
            for t in range(1, T):
                c, v, r, _ = m.online_forward(vids[i][:, t])
                out.append((c.clone(), m.last_logits[0].clone(), r.clone()))
            out.append((m._stream.clone(), m._coll.clone(), m._vis.clone()))
        stream.synchronize()

    serial = [[], []]
    for i in range(2):
        run(i, torch.cuda.current_stream(), serial[i])
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(2)]
    both = [[], []]
    errs = []

    def guarded(i):
        try:
            run(i, streams[i], both[i])
        except Exception as e:          # surfaces in the main thread below
            errs.append(e)

    th = [threading.Thread(target=guarded, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    assert not errs, errs
    for i in range(2):
        assert len(both[i]) == len(serial[i]) == T
        for k, (a, b) in enumerate(zip(serial[i], both[i])):
            for x, y in zip(a, b):
                assert torch.equal(x, y), f"tracker {i}, entry {k}: threaded run differs from the serial run"


def test_c1_eight_clicked_points_on_a_480p_stream_against_the_oracle():
    """BASELINE config 1 (demo_click-style: 8 clicked points, one 480x854 stream; tools/demo_click.py:40-48, benchmark_torch.py:61-76):
    LBM_online at N = 8 for 14 frames -- the ring of 12 wraps -- every frame against the CPU oracle; same tie rule and bars as
    tests/test_hip_configs.py, on the peaked checkpoint so the arg-max is exercised.  The 8 points are bench.py's."""
    import bench
    from oracle import lbm_ref_cpu as ref
    from lbm_amd.config import LBMConfig
    from lbm_amd.lbm_online import LBM_online
    from lbm_amd.synth import synthetic_state_dict, synthetic_video
    H, W, T = 480, 854, 15
    for preset in ("peaked", "flat"):
        m = LBM_online(LBMConfig(input_size=(384, 512), memory_size=12).to_namespace())
        sd = synthetic_state_dict(m, seed=42, preset=preset)
        m.load_state_dict(sd, strict=True)
        m = m.cuda().eval()
        g = ref.Geometry(size=(384, 512), memory_size=12)
        video = synthetic_video(T, H, W, seed=42)
        pts = bench.clicked_points(8, H, W)
        o = ref.RefOnline(sd, g)
        tie_seen = torch.zeros(8, dtype=torch.bool)
        dropped = torch.zeros(8, dtype=torch.bool)
        with torch.no_grad():
            m.init(pts.cuda(), video[:, 0].cuda())
            o.init(pts, video[:, 0])
            for t in range(1, T):
                c, v, r, _ = m.online_forward(video[:, t].cuda())
                feat = ref.encode_frames(o.sd, g, video[:, t])
                coords, ovl, _, out = ref.track_step(o.sd, g, o.state, feat)          # (rolls the oracle's state)
                for layer, k in zip(out, g.topk_per_layer):
                    tv = torch.topk(layer["corr"][0], k + 1, dim=-1)[0]
                    tie_seen |= (tv[:, k - 1] - tv[:, k]) < 1e-4
                oc, ovl = coords[0].clone(), ovl[0]
                oc[:, 0] = oc[:, 0] / g.size[1] * W
                oc[:, 1] = oc[:, 1] / g.size[0] * H
                ov = torch.sigmoid(ovl) > g.vis_threshold
                off = ((c.cpu() - oc).abs().amax(-1) >= 0.05) | ((m.last_logits[0][0].cpu() - ovl).abs() >= 1e-3)
                assert not bool((off & ~tie_seen & ~dropped).any()), f"{preset} frame {t}: {c.cpu() - oc}"
                dropped |= off
                ok = ~dropped & ((torch.sigmoid(ovl) - 0.8).abs() > 2e-3)
                assert (v.cpu() == ov)[ok].all()
        print(f"c1 {preset}: {int(dropped.sum())} of 8 points dropped by the tie rule ({int(tie_seen.sum())} saw an oracle gap < 1e-4)")
        assert int(dropped.sum()) <= (0 if preset == "peaked" else 1)
        assert m.t == T - 1


@pytest.mark.parametrize("M", [20, 36, 48, 60])
def test_inference_at_the_eval_memory_sizes_against_the_oracle(M):
    """LBM.inference at the memory sizes of lbm/configs/lbm.yaml:41-52: M + 6 frames, so the ring wraps;
    queries start at frames 0..3; every frame against the oracle (tie rule of tests/test_hip_configs.py)."""
    from oracle import lbm_ref_cpu as ref
    from lbm_amd.synth import synthetic_state_dict, synthetic_video
    m = _model("LBM")
    g = ref.Geometry(size=SIZE, memory_size=12)
    sd = {k: v.clone() for k, v in synthetic_state_dict(m, seed=42).items()}
    m.set_memory_size(M)
    ref.resize_time_embeddings(sd, g, M)
    T, (H, W), N = M + 6, FRAME, 24
    video = synthetic_video(T, H, W, seed=50 + M)
    r = np.random.default_rng(M)
    q = np.stack([r.integers(0, 4, N), r.uniform(8, W - 8, N), r.uniform(8, H - 8, N)], -1).astype(np.float32)
    queries = torch.from_numpy(q).unsqueeze(0)
    m.extend_queries = False
    with torch.no_grad():
        got = m.inference(video.cuda(), queries.clone().cuda())
        # the oracle's loop (oracle/lbm_ref_cpu.py: inference) with the gaps of its own selections on the side
        qs = queries.clone()
        qs[:, :, 2] = qs[:, :, 2] / H * SIZE[0]
        qs[:, :, 1] = qs[:, :, 1] / W * SIZE[1]
        times = qs[:, :, 0].long()
        feats = [ref.encode_frames(sd, g, video[:, t]) for t in range(T)]
        q_init = torch.zeros(1, N, 256)
        for t in range(4):
            sel = times[0] == t
            if bool(sel.any()):
                q_init[0, sel] = ref.sample_queries(feats[t], qs[0, sel, 1:], g)
        st = ref.fresh_state(q_init, g)
        tie_seen = torch.zeros(N, dtype=torch.bool)
        dropped = torch.zeros(N, dtype=torch.bool)
        for t in range(T):
            coords, vis, _, out = ref.track_step(sd, g, st, feats[t], times <= t)
            for layer, k in zip(out, g.topk_per_layer):
                tv = torch.topk(layer["corr"][0], k + 1, dim=-1)[0]
                tie_seen |= (tv[:, k - 1] - tv[:, k]) < 1e-4
            c = coords[0].clone()
            c[:, 0] = c[:, 0] / SIZE[1] * W
            c[:, 1] = c[:, 1] / SIZE[0] * H
            live = times[0] <= t                       # before its query frame a point's output is not meaningful
            off = ((got["points"][0, t].cpu() - c).abs().amax(-1) >= 0.05) & live
            assert not bool((off & ~tie_seen & ~dropped).any()), f"M {M} frame {t}: tracks differ with no tie in the oracle's selection"
            dropped |= off
            flag_ok = ((torch.sigmoid(vis[0]) - 0.8).abs() > 2e-3) & live & ~dropped
            assert (got["visibility"][0, t].cpu() == (torch.sigmoid(vis[0]) > 0.8))[flag_ok].all()
        line = f"inference M={M}: {int(dropped.sum())} of {N} points dropped by the tie rule ({int(tie_seen.sum())} saw an oracle gap < 1e-4)"
        print(line)
        import os
        if os.path.isdir("gpurun_out"):
            with open("gpurun_out/parity_drops.txt", "a") as f:
                f.write(line + "\n")
        assert dropped.float().mean() <= 0.05, f"M {M}: {int(dropped.sum())} of {N} points lost to ties"


def test_f16_storage_mode_through_update_and_inference(gold):
    """Precision "f16s" (memories + value maps f16 in HBM) through the parts of the API that touch the state: update() (add / drop /
    re-plant points: the edit goes through fp32 copies and comes back as f16) and LBM.inference (late queries).  Against the
    fp32-grade run of the same model: same tracks within the reduced-precision bars for the points both runs kept on the same cells."""
    g, inp = gold
    video = inp["video"].cuda()
    runs = {}
    for prec in ("split", "f16s"):
        o = _model("LBM_online")
        o.set_precision(prec)
        o = o.cuda()
        out = []
        with torch.no_grad():
            o.init(inp["pts0"].cuda(), video[:, 0])
            for t in range(1, 4):
                c, v, r, f_t = o.online_forward(video[:, t])
                out.append((c.clone(), o.last_logits[0][0].clone()))
            o.update(inp["pts1"].cuda(), f_t, inp["update"].cuda(), inp["remove"].cuda())
            assert o.N == 11 and o._stream.dtype == (torch.float16 if prec == "f16s" else torch.float32)
            assert o.stream_dist.dtype == torch.float32
            for t in range(4, 7):
                c, v, r, f_t = o.online_forward(video[:, t])
                out.append((c.clone(), o.last_logits[0][0].clone()))
        runs[prec] = out
    close = total = 0
    for (ca, la), (cb, lb) in zip(runs["split"], runs["f16s"]):
        same = (ca - cb).abs().amax(-1) < 0.05
        close += int(same.sum()); total += same.numel()
        assert float((la - lb).abs()[same].max()) < 1e-2
    assert close / total > 0.9, f"only {close} of {total} point-frames agree between the f16-storage and the fp32-grade run"
    m = _model("LBM")
    m.set_precision("f16s")
    m = m.cuda()
    m.extend_queries = False
    with torch.no_grad():
        res = m.inference(inp["video"][:, :6].cuda(), inp["queries"].clone().cuda())
    d = np.abs(res["points"].cpu().numpy() - g["inf_points"]).max(-1)
    assert (d < 0.05).mean() > 0.9 and np.isfinite(d).all()
