"""Host-side mirror of the reference interface: import paths, config merge rule, video/query loader, state-dict layout."""
import argparse
import os

import numpy as np
import pytest
import torch


def test_reference_import_paths_resolve():
    from lbm.models import LBM
    from lbm.models.lbm_online import LBM_online, LBM_export
    from lbm.models.lbm.lbm import LBM as LBM2
    from lbm.models.lbm.transformer import LBMTransformer, LBMTransformerLayer, MLP
    from lbm.models.lbm.backbone import Backbone
    from lbm.models.lbm.msdeformattn import MultiScaleDeformableAttention, multi_scale_deformable_attn_pytorch
    from lbm.utils.train_utils import fix_random_seeds
    from lbm.utils.eval_utils import load_config
    from lbm.utils.demo_utils import load_video
    from lbm.utils.coord_utils import indices_to_coords
    assert LBM is LBM2 and issubclass(LBM_online, LBM)
    fix_random_seeds(42)


def test_load_config_merge_rule(tmp_path):
    from lbm.utils.eval_utils import load_config
    p = tmp_path / "demo.yaml"
    p.write_text("input_size: [384, 512]\nmemory_size: 12\nstride: 4\nvideo_path: 'data/demo.mp4'\n")
    args = argparse.Namespace(config_path=str(p), memory_size=20, video_path=None, grid_size=32)
    cfg = load_config(args)
    assert cfg.memory_size == 20 and cfg.video_path == "data/demo.mp4" and cfg.grid_size == 32 and cfg.input_size == [384, 512]
    assert load_config(argparse.Namespace(config_path=str(tmp_path / "missing.yaml"))) == {}


def test_load_video_grid_rule_and_layout(tmp_path):
    from lbm.utils.demo_utils import load_video
    frames, video, pts = load_video("synthetic:3:48:64", grid_query=True, grid_size=3)
    assert len(frames) == 3 and frames[0].shape == (48, 64, 3) and frames[0].dtype == np.uint8
    assert video.shape == (1, 3, 3, 48, 64) and video.dtype == torch.float32
    assert float(video.min()) >= 0 and float(video.max()) <= 255
    # BGR frames, RGB tensor (demo_utils.py:127-131)
    assert np.array_equal(frames[1][:, :, ::-1], video[0, 1].permute(1, 2, 0).numpy().astype(np.uint8))
    want = [(int(i * 64 / 4), int(j * 48 / 4)) for i in range(1, 4) for j in range(1, 4)]
    assert pts.tolist() == [[float(a), float(b)] for a, b in want]
    # no click / grid -> zero points, as benchmark_torch.py:47 gets
    assert load_video("synthetic:2:48:64")[2].shape == (0, 2)
    np.save(tmp_path / "clip.npy", np.zeros((2, 8, 8, 3), dtype=np.uint8))
    assert load_video(str(tmp_path / "clip.npy"))[1].shape == (1, 2, 3, 8, 8)
    with pytest.raises(FileNotFoundError):
        load_video(str(tmp_path / "nope"))


def test_state_dict_has_the_lbm_pt_key_layout():
    """232 transformer keys, the ResNet-18 stem..layer3 keys, proj/fusion/pos-emb; strict load."""
    from lbm_amd.config import LBMConfig
    from lbm_amd.synth import synthetic_state_dict
    from lbm.models.lbm_online import LBM_online
    m = LBM_online(LBMConfig().to_namespace())
    sd = m.state_dict()
    tkeys = [k for k in sd if k.startswith("transformer.")]
    assert len(tkeys) == 232
    assert sd["transformer.layers.0.collision_time_emb"].shape == (1, 13, 256)
    assert sd["transformer.layers.2.lattice_attention.sampling_offsets.weight"].shape == (288, 256)
    assert sd["transformer.collision_mem_layer.attention_weights.weight"].shape == (64, 256)
    assert sd["transformer.layers.1.corr_conv.weight"].shape == (1, 4, 1, 1)
    assert sd["transformer.start_levels"].tolist() == [0, 12288, 15360, 16128]
    assert sd["backbone.layer1.0.weight"].shape == (64, 3, 7, 7)
    assert sd["backbone.layer2.0.downsample.0.weight"].shape == (128, 64, 1, 1)
    assert sd["backbone.layer3.1.bn2.num_batches_tracked"].dtype == torch.int64
    assert sd["backbone.fusion.0.weight"].shape == (256, 192, 1, 1)
    assert sd["backbone.frame_pos_embedding"].shape == (1, 256, 96, 128)
    n_tr = sum(v.numel() for k, v in m.transformer.named_parameters())
    assert n_tr == 11_814_726
    m.load_state_dict(synthetic_state_dict(m, 1), strict=True)
    # ckpt_converter.py:55 layout: {'model': state_dict}
    ckpt = {"model": {("module." + k): v for k, v in sd.items()}}
    stripped = {k[len("module."):]: v for k, v in ckpt["model"].items()}
    m.load_state_dict(stripped, strict=True)


def test_indices_to_coords_host_version(golden):
    from lbm.utils.coord_utils import indices_to_coords
    gold = golden("coords")
    got = indices_to_coords(torch.from_numpy(gold["idx"]), [384, 512], 4)
    assert np.array_equal(got.numpy(), gold["coords"])


def test_packed_weight_container_round_trip_and_rejections(tmp_path):
    """lbm_amd/packed.py container: entries come back bit-equal at 256-byte aligned offsets; truncated,
    foreign and inconsistent files are refused."""
    from collections import OrderedDict

    import numpy as np
    import pytest
    import torch

    from lbm_amd import packed
    g = torch.Generator().manual_seed(1)
    ent = OrderedDict(a=torch.randn(3, 5, generator=g), b=torch.randn(2, 7, 8, generator=g).half(), c=torch.empty(0, 4),
                      d=torch.randn(1, generator=g))
    p = str(tmp_path / "w.lbmw")
    n = packed.write_arena(p, ent, {"k": [1, 2]}, abi=3, source_digest="abc")
    header, views, arena = packed.read_arena(p)
    assert header["abi"] == 3 and header["meta"] == {"k": [1, 2]} and header["source_digest"] == "abc" and arena.numel() == n
    for name, t in ent.items():
        assert header["entries"][name]["offset"] % packed.ALIGN == 0
        assert views[name].dtype == t.dtype and torch.equal(views[name], t)
    raw = open(p, "rb").read()
    for bad, what in ((raw[:-1], "bytes on disk"), (b"NOTLBMW\n" + raw[8:], "bad magic"), (raw[:40], "truncated")):
        q = str(tmp_path / "bad.lbmw")
        open(q, "wb").write(bad)
        with pytest.raises(packed.PackedFormatError, match=what):
            packed.read_header(q)
    sd = {"x": torch.arange(6.).view(2, 3), "y": torch.ones(2)}
    d0 = packed.state_digest(sd)
    sd["y"][1] = float(np.nextafter(np.float32(1), np.float32(2)))
    assert packed.state_digest(sd) != d0 and len(d0) == 64


def test_ctypes_mirrors_match_the_c_header(tmp_path):
    """Every descriptor struct of include/lbm_hip.h against its ctypes mirror in lbm_amd/hip.py: same size, same field names, same
    offsets -- checked by compiling a C program that prints offsetof() for the fields the Python side declares (gcc; no GPU)."""
    import ctypes
    import json
    import subprocess
    from lbm_amd import hip
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pairs = [("lbm_hip_linear_desc", hip.LinearDesc), ("lbm_hip_linear_ss_desc", hip.LinearSSDesc), ("lbm_hip_msda_desc", hip.MsdaDesc),
             ("lbm_hip_step_desc", hip.StepDesc), ("lbm_hip_profile_entry", hip.ProfileEntry)]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "lbm_hip.h"', 'int main(void) {', '  printf("{");']
    for cname, cls in pairs:
        lines.append(f'  printf("\\"{cname}\\": {{\\"sizeof\\": %zu", sizeof({cname}));')
        for fname, *_ in cls._fields_:
            lines.append(f'  printf(", \\"{fname}\\": %zu", offsetof({cname}, {fname}));')
        lines.append('  printf("}%s", "' + ("" if cname == pairs[-1][0] else ", ") + '");')
    lines += ['  printf("}\\n");', '  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    got = json.loads(subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout)
    for cname, cls in pairs:
        assert got[cname]["sizeof"] == ctypes.sizeof(cls), cname
        for fname, *_ in cls._fields_:
            assert got[cname][fname] == getattr(cls, fname).offset, f"{cname}.{fname}"


def test_ctypes_argument_lists_match_the_c_prototypes():
    """Number of parameters of every prototype in include/lbm_hip.h against the argtypes lbm_amd/hip.py declares for it (a parameter
    added on one side only would shift every later argument silently)."""
    import re
    from lbm_amd import hip
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "lbm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    lib = hip.load()
    checked = 0
    for m in re.finditer(r"\b(?:int|int64_t|uint64_t|const char\*)\s+(lbm_hip_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        name, params = m.group(1), m.group(2).strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        fn = getattr(lib, name)
        if fn.argtypes is None:
            continue
        assert len(fn.argtypes) == n, f"{name}: header has {n} parameters, hip.py declares {len(fn.argtypes)}"
        checked += 1
    assert checked >= 30


def test_product_library_refuses_the_timing_only_ablation_knobs():
    """Keys 14 / 16 / 18 (bits 0-7) / 20 and bits 12.. of key 2 switch parts of kernels OFF for timing (results are wrong by design);
    that code exists only in the -DLBM_TUNING build.  The product library must refuse them with an error -- not accept and ignore --
    and lbm_hip_debug_active must list exactly the knobs that differ from their defaults (what bench.py prints in `config.knobs`)."""
    from lbm_amd import hip
    lib = hip.load()
    if lib.lbm_hip_tuning_build():
        pytest.skip("LBM_AMD_LIB points at a tuning build")
    assert hip.debug_active() == {}
    for key, value in ((14, 1), (16, 8), (18, 2), (20, 4), (2, 1 << 12), (2, 4)):
        assert lib.lbm_hip_debug_set(key, value) == -1, (key, value)
        assert b"ablation" in lib.lbm_hip_last_error()
    assert hip.debug_active() == {}
    try:
        assert lib.lbm_hip_debug_set(18, 4 << 8) == 0          # steps per workgroup of conv64_rows: a real tuning knob, results unchanged
        assert lib.lbm_hip_debug_set(2, 256) == 0
        assert lib.lbm_hip_debug_set(21, 0) == 0
        assert lib.lbm_hip_debug_set(17, 1) == 0               # its default: not "active"
        assert hip.debug_active() == {18: 4 << 8, 2: 256, 21: 0}
    finally:
        for key, value in ((18, 0), (2, 0), (21, 1)):
            lib.lbm_hip_debug_set(key, value)
    assert hip.debug_active() == {}
    assert lib.lbm_hip_debug_set(99, 1) == -1


def test_inference_mode_tensors_do_not_break_the_frame_and_query_keys():
    """Tensors created under torch.inference_mode() have no version counter (reading ._version raises); the keys that use it
    (LBM_online._frame_key for prefetch, the engine's per-query cache, the graph key) fall back to None for them."""
    from lbm_amd import hip
    from lbm_amd.lbm_online import LBM_online
    with torch.inference_mode():
        f = torch.zeros(1, 3, 8, 8)
        assert hip.tensor_version(f) is None
        key = LBM_online._frame_key(f)
        assert key[-1] is None and key[1] == (1, 3, 8, 8)
    g = torch.zeros(1, 3, 8, 8)
    v0 = hip.tensor_version(g)
    g.add_(1)
    assert hip.tensor_version(g) == v0 + 1
    assert LBM_online._frame_key(g)[-1] == v0 + 1


def test_frame_geometry_follows_the_declared_layout():
    from lbm_amd import hip
    assert hip.frame_hw(torch.zeros(2, 3, 48, 64)) == (48, 64)
    assert hip.frame_hw(torch.zeros(2, 3, 48, 64, dtype=torch.uint8)) == (48, 64)
    assert hip.frame_hw(torch.zeros(2, 48, 64, 3, dtype=torch.uint8), "hwc_rgb") == (48, 64)
    assert hip.frame_hw(torch.zeros(2, 48, 64, 3, dtype=torch.uint8), "hwc_bgr") == (48, 64)
    assert hip.frame_hw(torch.zeros(2, 3, 48, 64), "hwc_rgb") == (48, 64)      # float frames are always (B,3,H,W)
