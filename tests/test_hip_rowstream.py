"""Row-streaming split product (lbm_amd/csrc/gemm_rs.hip) through the C ABI.

Linear form: bit-identical to the tiled product (gemm16.hip) on every output kind, ragged shapes included.
Correlation + top-k form (lbm_hip_corr_topk; transformer.py:306-308, coord_utils.py:4-29): the selected cells, their order,
scores and centres equal the dense product followed by the dense top-k, bit for bit; exact ties go to the lowest cell
wherever in the partition (other half-wave, other step, other cell range) the equal cells lie."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from lbm_amd import hip as h
    h.load()
    return h


def _split(hip, x):
    t = hip.SplitTensor(tuple(x.shape), "cuda")
    hip.load().lbm_hip_split_planes(x.data_ptr(), t.hi, t.planes[1].data_ptr(), x.numel(), None)
    return t


@pytest.mark.parametrize("ring", [3, 4])           # debug key 25: stages of the row-streaming kernel's LDS ring (steps fetched 2 / 3 ahead)
@pytest.mark.parametrize("M,N", [(4100, 256), (33, 1456), (5000, 432), (40000, 1024), (1, 256), (31, 8), (257, 260)])
def test_linear_is_bit_identical_to_the_tiled_product(hip, M, N, ring):
    lib = hip.load()
    lib.lbm_hip_debug_set(25, ring)
    r = np.random.default_rng(M + N)
    xs = _split(hip, torch.from_numpy(r.normal(size=(M, 256)).astype(np.float32)).cuda())
    ws = hip.SplitWeight(torch.from_numpy((r.normal(size=(N, 256)) * 0.1).astype(np.float32)).cuda())
    b = torch.from_numpy(r.normal(size=(N,)).astype(np.float32)).cuda()
    res = torch.from_numpy(r.normal(size=(M, N)).astype(np.float32)).cuda()
    try:
        for act, use_res, f32, split in [(0, False, True, False), (1, False, False, True), (2, True, True, True), (1, False, True, True)]:
            out = {}
            for flag in (256, 512):            # debug key 2: 256 = tiled kernel only, 512 = row-streaming whenever the shape admits it
                lib.lbm_hip_debug_set(2, flag)
                out[flag] = hip.linear_ss(xs, ws, b, act=act, res=res if use_res else None, res_scale=0.5, want_f32=f32, want_split=split)
            (y0, s0), (y1, s1) = out[256], out[512]
            if f32:
                assert torch.equal(y0.view(torch.int32), y1.view(torch.int32)), (act, use_res)
            if split:
                assert torch.equal(s0.view(torch.int16), s1.view(torch.int16)), (act, use_res)
    finally:
        lib.lbm_hip_debug_set(2, 0)
        lib.lbm_hip_debug_set(25, 3)


def _dense(hip, q, F, k, grid_w):
    """the unfused path: tiled product into an fp32 score matrix, then the dense top-k kernel"""
    lib = hip.load()
    B, N, _ = q.shape
    idx, ref, val = [], [], []
    lib.lbm_hip_debug_set(2, 256)
    try:
        for b in range(B):
            qs, fs = hip.SplitTensor((N, 256), "cuda"), hip.SplitTensor((F.shape[1], 256), "cuda")
            qs.planes.copy_(q.planes[:, b])
            fs.planes.copy_(F.planes[:, b])
            corr, _ = hip.linear_ss(qs, fs)
            i, r, v = hip.topk(corr, k, grid_w)
            idx.append(i); ref.append(r); val.append(v)
    finally:
        lib.lbm_hip_debug_set(2, 0)
    return torch.stack(idx), torch.stack(ref), torch.stack(val)


@pytest.mark.parametrize("B,N,h,w,k", [(1, 1024, 96, 128, 9), (3, 300, 40, 52, 4), (2, 37, 24, 32, 1), (64, 64, 16, 24, 9), (1, 2600, 48, 64, 9)])
@pytest.mark.parametrize("ring", [3, 4])
def test_correlation_top_k_equals_dense_product_and_dense_top_k(hip, B, N, h, w, k, ring):
    g = torch.Generator().manual_seed(B * 1000 + N)
    qf = torch.nn.functional.normalize(torch.randn(B, N, 256, generator=g), dim=-1).cuda()
    Ff = (torch.randn(B, h * w, 256, generator=g) * 0.3).cuda()
    q, F = _split(hip, qf), _split(hip, Ff)
    hip.load().lbm_hip_debug_set(25, ring)
    try:
        idx, ref, val = hip.corr_topk(q, F, k, w)
    finally:
        hip.load().lbm_hip_debug_set(25, 3)
    di, dr, dv = _dense(hip, q, F, k, w)
    assert torch.equal(idx, di.view_as(idx))
    assert torch.equal(val.view(torch.int32), dv.view_as(val).view(torch.int32))
    assert torch.equal(ref, dr.view_as(ref))


@pytest.mark.parametrize("shape", ["ascending", "descending", "bump", "plateaus"])
@pytest.mark.parametrize("k", [9, 4])
def test_correlation_top_k_on_rows_that_keep_the_lists_busy(hip, shape, k):
    """Score rows on which a lane's list changes at every cell (ascending), never after the first k (descending), in runs (a smooth
    bump: what a real correlation row looks like) or among long runs of exactly equal scores: the per-lane lists of the fused kernel
    must end up as the dense product + dense top-k, ties cut at the lowest cell."""
    h, w, N = 96, 128, 520
    P0 = h * w
    g = torch.Generator().manual_seed(17)
    d = torch.nn.functional.normalize(torch.randn(256, generator=g), dim=0)
    qf = torch.nn.functional.normalize(d[None, None] + 0.2 * torch.randn(1, N, 256, generator=g), dim=-1)
    c = torch.arange(P0, dtype=torch.float32)
    if shape == "ascending":
        amp = 0.2 + 0.8 * c / P0
    elif shape == "descending":
        amp = 1.0 - 0.8 * c / P0
    elif shape == "bump":
        yy, xx = c // w, c % w
        amp = 0.2 + torch.exp(-((yy - 40.0) ** 2 + (xx - 77.0) ** 2) / 200.0)
    else:
        amp = 0.25 * (1 + (c // 37) % 4)                 # runs of 37 equal cells, four levels
    Ff = amp[None, :, None] * d[None, None]              # every cell the same direction: scores follow amp exactly, ties included
    if shape != "plateaus":
        Ff = Ff + 0.002 * torch.randn(1, P0, 256, generator=g)
    q, F = _split(hip, qf.cuda()), _split(hip, Ff.cuda())
    idx, ref, val = hip.corr_topk(q, F, k, w)
    di, dr, dv = _dense(hip, q, F, k, w)
    assert torch.equal(idx, di.view_as(idx)), (shape, k)
    assert torch.equal(val.view(torch.int32), dv.view_as(val).view(torch.int32)), (shape, k)
    assert torch.equal(ref, dr.view_as(ref)), (shape, k)


def test_correlation_top_k_breaks_exact_ties_to_the_lowest_cell(hip):
    """Identical map rows give exactly equal scores.  Copies are planted in the same half-wave, the other half-wave (rows 4 apart),
    another step (32 apart), another cell range, and as a run of 12 equal cells that must be cut at k."""
    h, w, k, N = 96, 128, 9, 300
    P0 = h * w
    g = torch.Generator().manual_seed(5)
    qf = torch.nn.functional.normalize(torch.randn(1, N, 256, generator=g), dim=-1)
    Ff = torch.randn(1, P0, 256, generator=g) * 0.05
    # every point's own direction, planted (scaled so it wins) at several cells: exact duplicates of one another
    plant = [7, 8, 11, 7 + 32, 5000, 5000 + 4, 12000, 3, 6001, 9000, 9001, 9002]
    for n in range(0, N, 3):
        for c in plant:
            Ff[0, (c + 13 * n) % P0] = qf[0, n] * 2.0
    q, F = _split(hip, qf.cuda()), _split(hip, Ff.cuda())
    idx, ref, val = hip.corr_topk(q, F, k, w)
    di, dr, dv = _dense(hip, q, F, k, w)
    assert torch.equal(idx, di.view_as(idx))
    for n in range(0, N, 3):
        cells = sorted((c + 13 * n) % P0 for c in plant)
        got = idx[0, n].tolist()
        assert got == cells[:k], (n, got, cells[:k])
        assert len(set(val[0, n].tolist())) == 1            # all k scores exactly equal: the cut was made on the index alone


def test_corr_topk_rejects_bad_arguments(hip):
    q = hip.SplitTensor((1, 8, 256), "cuda")
    F = hip.SplitTensor((1, 64, 256), "cuda")
    q.planes.zero_(); F.planes.zero_()
    with pytest.raises(hip.HipError):
        hip.corr_topk(q, F, 10, 8)                              # k > 9
    lib = hip.load()
    assert lib.lbm_hip_corr_topk_workspace_bytes(1, 8, 64, 0) < 0


def _ffn_operands(hip, M, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    W1 = (torch.randn(1024, 256, generator=g) / 16).cuda()
    b1 = (torch.randn(1024, generator=g) * 0.1).cuda()
    W2 = (torch.randn(256, 1024, generator=g) / 32).cuda()
    x = torch.randn(M, 256, generator=g).cuda()
    f1, f2 = hip.fragment_major(hip.SplitWeight(W1)), hip.fragment_major(hip.SplitWeight(W2))
    return W1, b1, W2, x, _split(hip, x), f1, f2


@pytest.mark.parametrize("M", [1, 33, 1024, 2050, 4100, 9216])
def test_many_rows_feed_forward_leaves_the_same_slabs_as_the_few_rows_kernel(hip, M):
    """ffn_rows_kernel (weights stationary, rows streamed; debug key 15 = 2) against ffn_fused_kernel (key 15 = 1): the eight
    partial outputs must be bit-identical -- same MFMAs in the same order -- and their sum must meet the fp64 bar."""
    lib = hip.load()
    W1, b1, W2, x, xs, f1, f2 = _ffn_operands(hip, M, 100 + M)
    try:
        lib.lbm_hip_debug_set(15, 1)
        a = hip.ffn_slabs(xs, f1, b1, f2)
        lib.lbm_hip_debug_set(15, 2)
        b = hip.ffn_slabs(xs, f1, b1, f2)
    finally:
        lib.lbm_hip_debug_set(15, 0)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    want = torch.nn.functional.gelu(x.double() @ W1.double().t() + b1.double()) @ W2.double().t()
    assert float((b.sum(0).double() - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("B,H,W", [(1, 5, 32), (2, 24, 64), (3, 7, 96), (4, 96, 128), (16, 96, 128)])
@pytest.mark.parametrize("relu,use_res,f32,split", [(True, False, False, True), (True, True, True, True), (False, False, True, False),
                                                    (True, True, True, False), (False, True, False, True)])
def test_patch_streaming_convolution_is_bit_identical_to_the_tiled_one(hip, B, H, W, relu, use_res, f32, split):
    """conv64_rows_kernel (3 x 3, stride 1, 64 -> 64; debug key 4 = 16) against the tiled implicit GEMM (key 4 = 15): same accumulation
    chain, so every output must agree bit for bit -- borders (zero padding), one-row images, every output combination (the residual
    rows are fetched two steps ahead behind a wait that counts the stores of each combination) and 1 to 24 steps per workgroup."""
    lib = hip.load()
    g = torch.Generator(device="cpu").manual_seed(B * 100 + H)
    x = torch.randn(B, H, W, 64, generator=g).cuda()
    w = hip.SplitWeight((torch.randn(64, 576, generator=g) * 0.05).cuda())
    b = torch.randn(64, generator=g).cuda()
    res = torch.randn(B, H, W, 64, generator=g).cuda() if use_res else None
    xs = _split(hip, x)
    outs = {}
    try:
        for mode in (15, 16):
            lib.lbm_hip_debug_set(4, mode)
            hip.profile_begin()
            outs[mode] = hip.conv2d_nhwc_ss(xs, w, b, 3, 1, 1, relu, res, f32, split)
            names = {p["name"] for p in hip.profile_end()}
            assert ("conv64_rows" in names) == (mode == 16), names
    finally:
        lib.lbm_hip_debug_set(4, 0)
    (y0, s0), (y1, s1) = outs[15], outs[16]
    if f32:
        assert torch.equal(y0.view(torch.int32), y1.view(torch.int32))
    if split:
        assert torch.equal(s0.planes.view(torch.int16), s1.planes.view(torch.int16))


@pytest.mark.parametrize("B,H,W,C,tile", [(2, 8, 64, 128, 256), (3, 6, 64, 128, 192), (1, 12, 64, 128, 0), (2, 8, 32, 256, 256), (1, 12, 32, 256, 192),
                                          (3, 24, 32, 256, 0), (1, 4, 64, 256, 0), (5, 6, 32, 128, 0)])
@pytest.mark.parametrize("relu,use_res,f32,split", [(True, False, False, True), (True, True, True, True), (False, True, True, False)])
def test_patch_resident_convolution_matches_the_tiled_one(hip, B, H, W, C, tile, relu, use_res, f32, split):
    """conv_patch_kernel (3 x 3, stride 1, C -> C with C = 128 / 256 on 64- / 32-pixel-wide maps: the input patch of a 256- or 192-pixel tile
    stays in LDS for all nine taps, weights stream; debug key 4 = 22, tile forced by key 26 >> 8) against the tiled implicit GEMM (key 4 = 21)
    and against float64: the same products summed channel chunk by channel chunk instead of tap by tap, so agreement is to fp32
    rounding -- image borders (zero padding above / below / left / right of every tile), tiles at image boundaries of a batch, one to
    several tiles per image, every output combination."""
    lib = hip.load()
    g = torch.Generator(device="cpu").manual_seed(B * 100 + H + C)
    x = torch.randn(B, H, W, C, generator=g).cuda()
    w4 = torch.randn(C, C, 3, 3, generator=g) * 0.03
    w = hip.SplitWeight(w4.permute(0, 2, 3, 1).reshape(C, 9 * C).contiguous().cuda())
    b = torch.randn(C, generator=g).cuda()
    res = torch.randn(B, H, W, C, generator=g).cuda() if use_res else None
    xs = _split(hip, x)
    outs = {}
    try:
        for mode in (21, 22):
            lib.lbm_hip_debug_set(4, mode)
            lib.lbm_hip_debug_set(26, tile << 8)
            hip.profile_begin()
            outs[mode] = hip.conv2d_nhwc_ss(xs, w, b, 3, 1, 1, relu, res, f32, split)
            names = {p["name"] for p in hip.profile_end()}
            assert ("conv_patch" in names) == (mode == 22), names
    finally:
        lib.lbm_hip_debug_set(4, 0)
        lib.lbm_hip_debug_set(26, 0)
    want = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w4.double(), b.double().cpu(), padding=1).permute(0, 2, 3, 1)
    if use_res:
        want = want + res.double().cpu()
    if relu:
        want = torch.relu(want)
    tol = 4e-6 * max(1.0, float(want.abs().max()))
    (y0, s0), (y1, s1) = outs[21], outs[22]
    if f32:
        assert float((y1.double().cpu() - want).abs().max()) < tol
        assert float((y1 - y0).abs().max()) < tol
    if split:
        assert float((s1.float().double().cpu() - want).abs().max()) < tol
        assert float((s1.float() - s0.float()).abs().max()) < tol


@pytest.mark.parametrize("B,H,W", [(1, 8, 8), (2, 36, 124), (1, 96, 128), (3, 64, 256), (1, 384, 512)])
def test_fused_stem_matches_convolution_relu_maxpool(hip, B, H, W):
    """stem_pool (7 x 7 / 2 convolution + ReLU + 3 x 3 / 2 max-pooling in one launch, stem_rs.hip) against the two separate kernels
    and against float64 torch: image borders, strips that end past the image, one to many pooled rows per workgroup."""
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + H + W)
    x = torch.zeros(B, H, W, 4)
    x[..., :3] = torch.randn(B, H, W, 3, generator=g) * 2.0
    w4 = torch.randn(64, 4, 7, 7, generator=g) * 0.05
    w4[:, 3] = 0
    bias = torch.randn(64, generator=g) * 0.5
    flat = w4.permute(0, 2, 3, 1).reshape(64, 196)
    flat = torch.cat([flat, torch.zeros(64, 28)], 1).contiguous()
    ws = hip.SplitWeight(flat.cuda())
    xs = hip.split_tensor(x.cuda())
    y, ys = hip.stem_pool(xs, ws, bias.cuda())
    want = torch.nn.functional.max_pool2d(torch.relu(torch.nn.functional.conv2d(
        x.permute(0, 3, 1, 2).double(), w4.double(), bias.double(), stride=2, padding=3)), 3, 2, 1).permute(0, 2, 3, 1)
    assert y.shape == want.shape
    tol = 3e-6 * max(1.0, float(want.abs().max()))
    assert float((y.double().cpu() - want).abs().max()) < tol
    assert float((ys.float().double().cpu() - want).abs().max()) < tol
    # the two-kernel path on the same operands: same values up to the summation order inside the matrix cores
    c = hip.conv2d_nhwc(x.cuda(), ws, bias.cuda(), 7, 7, 2, 3, True, None)
    y2, ys2 = hip.maxpool3x3s2_nhwc_s(c)
    assert float((y - y2).abs().max()) < tol
    only_split = hip.stem_pool(xs, ws, bias.cuda(), want_f32=False)
    assert only_split[0] is None and torch.equal(only_split[1].planes.view(torch.int16), ys.planes.view(torch.int16))


def test_preprocess_split_is_the_split_of_preprocess(hip):
    g = torch.Generator(device="cpu").manual_seed(5)
    frames = torch.randint(0, 256, (2, 3, 60, 90), generator=g, dtype=torch.uint8).cuda()
    want = hip.split_tensor(hip.preprocess(frames.float(), 48, 64, layout=4))
    for f in (frames, frames.float()):
        got = hip.preprocess_split(f, 48, 64)
        assert torch.equal(got.planes.view(torch.int16), want.planes.view(torch.int16))
    hwc = frames.permute(0, 2, 3, 1).contiguous()
    assert torch.equal(hip.preprocess_split(hwc.flip(-1).contiguous(), 48, 64, "hwc_bgr").planes.view(torch.int16), want.planes.view(torch.int16))


@pytest.mark.parametrize("B,h,w,with_pos", [(1, 8, 8, True), (2, 16, 24, False), (1, 96, 128, True)])
def test_encoder_tail_with_the_level0_product_inside_is_bit_identical(hip, B, h, w, with_pos):
    """lbm_hip_encoder_tail_fused (q1 = f1 . W1^T + b1 formed per 8 x 8 block on the matrix cores) against lbm_hip_linear_ss followed
    by lbm_hip_encoder_tail: tokens of all four levels and their split twin, bit for bit."""
    g = torch.Generator(device="cpu").manual_seed(h * w + B)
    P0 = h * w
    L = P0 + P0 // 4 + P0 // 16 + P0 // 64
    f1 = hip.split_tensor(torch.randn(B, h, w, 64, generator=g).cuda())
    w1 = hip.SplitWeight((torch.randn(256, 64, generator=g) * 0.2).cuda())
    b1 = torch.randn(256, generator=g).cuda()
    q2 = torch.randn(B, P0 // 4, 256, generator=g).cuda()
    q3 = torch.randn(B, P0 // 16, 256, generator=g).cuda()
    pos = torch.randn(P0, 256, generator=g).cuda() if with_pos else None
    outs = []
    for fused in (False, True):
        tok = torch.zeros(B, L, 256, device="cuda")
        toks = hip.SplitTensor((B, L, 256), "cuda")
        if fused:
            hip.encoder_tail_fused(f1, w1, b1, q2, q3, pos, tok, toks, h, w)
        else:
            q1 = hip.linear_ss(f1.reshape(-1, 64), w1, b1)[0]
            hip.encoder_tail(q1, q2, q3, pos, tok, toks, h, w)
        outs.append((tok, toks))
    assert torch.equal(outs[0][0].view(torch.int32), outs[1][0].view(torch.int32))
    assert torch.equal(outs[0][1].planes.view(torch.int16), outs[1][1].planes.view(torch.int16))


@pytest.mark.parametrize("B,h,w,with_pos", [(1, 8, 8, True), (3, 16, 24, False), (8, 16, 8, True), (2, 96, 128, True)])
def test_encoder_tail_patch_form_equals_the_cell_by_cell_taps(hip, B, h, w, with_pos):
    """The encoder tail fetches, per 2 x 2 group of cells, one 3 x 3 patch of the half-resolution level and one 2 x 2 patch of the
    quarter-resolution level (13 rows) where it used to fetch four bilinear taps per cell and level (32 rows); debug key 24 bit 1
    selects the old form.  Tokens of all levels, split twin and row norms must be the same bits, borders included (grids of one
    block, where every group touches an edge), for both entry points."""
    lib = hip.load()
    g = torch.Generator(device="cpu").manual_seed(7 * h + w + B)
    P0 = h * w
    L = P0 + P0 // 4 + P0 // 16 + P0 // 64
    f1 = hip.split_tensor(torch.randn(B, h, w, 64, generator=g).cuda())
    w1 = hip.SplitWeight((torch.randn(256, 64, generator=g) * 0.2).cuda())
    b1 = torch.randn(256, generator=g).cuda()
    q2 = torch.randn(B, P0 // 4, 256, generator=g).cuda()
    q3 = torch.randn(B, P0 // 16, 256, generator=g).cuda()
    pos = torch.randn(P0, 256, generator=g).cuda() if with_pos else None
    q1 = hip.linear_ss(f1.reshape(-1, 64), w1, b1)[0]
    outs = []
    try:
        for key in (3, 1):
            lib.lbm_hip_debug_set(24, key)
            for fused in (False, True):
                tok = torch.zeros(B, L, 256, device="cuda")
                toks = hip.SplitTensor((B, L, 256), "cuda")
                inv = torch.zeros(B, L, device="cuda")
                if fused:
                    hip.encoder_tail_fused(f1, w1, b1, q2, q3, pos, tok, toks, h, w, inv)
                else:
                    hip.encoder_tail(q1, q2, q3, pos, tok, toks, h, w)
                outs.append((tok, toks, inv))
    finally:
        lib.lbm_hip_debug_set(24, 1)
    for other in outs[1:]:
        assert torch.equal(outs[0][0].view(torch.int32), other[0].view(torch.int32))
        assert torch.equal(outs[0][1].planes.view(torch.int16), other[1].planes.view(torch.int16))
    assert torch.equal(outs[1][2].view(torch.int32), outs[3][2].view(torch.int32))
    assert bool(torch.isfinite(outs[3][0]).all()) and float(outs[3][0].abs().max()) > 0


def test_encoder_tail_row_norms_are_the_norm_kernels(hip):
    """The row factors lbm_hip_encoder_tail_fused leaves beside the tokens equal, bit for bit, what the step's own norm pass computes
    (lbm_hip_corr_maps writes them into its inv_norm argument)."""
    B, h, w = 2, 16, 24
    g = torch.Generator(device="cpu").manual_seed(3)
    P0 = h * w
    L = P0 + P0 // 4 + P0 // 16 + P0 // 64
    f1 = hip.split_tensor(torch.randn(B, h, w, 64, generator=g).cuda())
    w1 = hip.SplitWeight((torch.randn(256, 64, generator=g) * 0.2).cuda())
    q2 = torch.randn(B, P0 // 4, 256, generator=g).cuda()
    q3 = torch.randn(B, P0 // 16, 256, generator=g).cuda()
    tok = torch.zeros(B, L, 256, device="cuda")
    inv = torch.zeros(B, L, device="cuda")
    hip.encoder_tail_fused(f1, w1, None, q2, q3, None, tok, None, h, w, inv)
    inv2 = torch.zeros(B, L, device="cuda")
    F = torch.empty(1, B, P0, 256, device="cuda")
    mix = (C.c_float * 4)(0.1, 0.2, 0.3, 0.4)
    assert hip.load().lbm_hip_corr_maps(tok.data_ptr(), inv2.data_ptr(), F.data_ptr(), mix, 1, B, h, w, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(inv.view(torch.int32), inv2.view(torch.int32))
    want = 1.0 / tok.double().norm(dim=-1).clamp_min(1e-12)
    assert float(((inv.double() - want) / want).abs().max()) < 1e-6


@pytest.mark.parametrize("B,N,k,P", [(8, 5, 3, 9), (16, 33, 9, 9), (8, 7, 1, 4), (24, 3, 4, 9)])
def test_msda_xcd_aware_order_gives_the_same_rows(hip, B, N, k, P):
    """msda_sample with a batch of streams re-deals its workgroups over the XCDs (debug key 21): a bijection of the grid, so every
    output row must equal the dispatch-order run bit for bit -- stream counts 8, 16, 24, odd point counts, k = 1, 3, 4, 9."""
    lib = hip.load()
    g = torch.Generator(device="cpu").manual_seed(B * 100 + N * 10 + k)
    shapes = [(16, 24), (8, 12), (4, 6), (2, 3)]
    L = sum(h * w for h, w in shapes)
    value = torch.randn(B, L, 256, generator=g).cuda()
    offlog = (torch.randn(B * N, 48 * P, generator=g) * 2.0).cuda()
    ref = (torch.rand(B * N, k, 2, generator=g) * torch.tensor([96.0, 64.0])).cuda()
    outs = []
    try:
        for mode in (0, 1):
            lib.lbm_hip_debug_set(21, mode)
            outs.append(hip.msda_sample(value, shapes, offlog, ref, B, N, k, P, 96.0, 64.0))
    finally:
        lib.lbm_hip_debug_set(21, 1)
    assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
    assert bool(torch.isfinite(outs[1]).all()) and float(outs[1].abs().max()) > 0


@pytest.mark.parametrize("B,N,k,spread,off_scale,shapes", [
    (1, 70, 9, "block", 1.0, [(96, 128), (48, 64), (24, 32), (12, 16)]),        # 3 x 3 neighbourhood, small offsets: everything in the window
    (1, 70, 9, "block", 6.0, [(96, 128), (48, 64), (24, 32), (12, 16)]),        # large learned offsets: part of the taps outside the window
    (1, 64, 9, "scattered", 2.0, [(96, 128), (48, 64), (24, 32), (12, 16)]),    # flat correlation row: cells anywhere, nothing staged
    (1, 64, 4, "block", 2.0, [(96, 128), (48, 64), (24, 32), (12, 16)]),        # layer 1: 2 x 2 cells
    (1, 33, 4, "mixed", 3.0, [(16, 24), (8, 12), (4, 6), (2, 3)]),              # tiny maps: the window is the whole coarse level
    (16, 9, 9, "mixed", 2.0, [(16, 24), (8, 12), (4, 6), (2, 3)]),              # a batch of streams: XCD-aware order of the point workgroups
    (1, 40, 9, "edge", 2.5, [(96, 128), (48, 64), (24, 32), (12, 16)]),         # reference cells on the map border: zero-padded taps
    (1, 77, 1, "scattered", 2.0, [(96, 128), (48, 64), (24, 32), (12, 16)]),    # k = 1 (last layer, heads): the tap-table kernel
    (8, 21, 1, "edge", 4.0, [(16, 24), (8, 12), (4, 6), (2, 3)]),               # k = 1, a batch of streams, borders
])
def test_msda_window_kernel_gives_the_direct_kernels_rows(hip, B, N, k, spread, off_scale, shapes):
    """The LDS-staged form of msda_sample (k = 9 / 4 lattice launches: one workgroup per point, window of value cells staged per level
    and head, taps worked out once per (cell, sample)) against the direct gather (debug key 22 = 0): every output row bit for bit --
    clustered reference cells, scattered ones, offsets that leave the window, map borders."""
    lib = hip.load()
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + N * 10 + k + int(off_scale * 7))
    h0, w0 = shapes[0]
    net_w, net_h = 4.0 * w0, 4.0 * h0
    L = sum(h * w for h, w in shapes)
    P = 9
    value = torch.randn(B, L, 256, generator=g).cuda()
    offlog = torch.randn(B * N, 48 * P, generator=g)
    offlog[:, :32 * P] *= off_scale
    offlog = offlog.cuda()
    R = B * N
    centre = torch.rand(R, 1, 2, generator=g) * torch.tensor([net_w, net_h])
    if spread == "edge":
        centre = centre * torch.tensor([1.0, 0.02])                          # top rows of the map
        centre[::2] = torch.tensor([net_w - 1.0, net_h - 1.0]) - centre[::2] * torch.tensor([0.02, 1.0])
    side = 3 if k == 9 else 2
    cells = torch.stack([torch.tensor([float(i % side), float(i // side)]) for i in range(k)], 0) * 4.0     # neighbouring cells, 4 px apart
    block = (centre // 4.0) * 4.0 + 2.0 + cells.unsqueeze(0)
    scattered = torch.rand(R, k, 2, generator=g) * torch.tensor([net_w, net_h])
    if spread == "scattered":
        ref = scattered
    elif spread == "mixed":
        ref = torch.where((torch.arange(R) % 3 == 0).view(R, 1, 1), scattered, block)
        ref[1::3, -1] = scattered[1::3, -1]                                  # one outlier cell beside a neighbourhood
    else:
        ref = block
    ref = ref.cuda().contiguous()
    outs = []
    try:
        for mode in (0, 3):
            lib.lbm_hip_debug_set(22, mode)
            hip.profile_begin()
            outs.append(hip.msda_sample(value, shapes, offlog, ref, B, N, k, P, net_w, net_h))
            names = {p["name"] for p in hip.profile_end()}
            assert names == (({"msda_taps"} if k == 1 else {"msda_window"}) if mode else {"msda_sample"}), names
    finally:
        lib.lbm_hip_debug_set(22, 3)
    assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
    assert bool(torch.isfinite(outs[1]).all()) and float(outs[1].abs().max()) > 0


def test_step_with_the_window_and_tap_table_kernels_equals_the_direct_gather(hip):
    """Whole frames through LBM_online with every deformable sampling launch on the direct kernel (debug key 22 = 0) and on the
    LDS-window / tap-table kernels (3, the default) -- k = 9, 4, 1, the paired offset + visibility heads (a two-module launch) and
    the 4-point collision-memory head: tracks, logits and both memories bit for bit."""
    from lbm_amd.config import LBMConfig
    from lbm_amd.lbm_online import LBM_online
    from lbm_amd.synth import random_queries, synthetic_state_dict, synthetic_video
    lib = hip.load()
    H, W, T = 120, 160, 5
    video = synthetic_video(T, H, W, seed=5).cuda()
    pts = random_queries(37, H, W, 9).cuda()
    runs = []
    try:
        for mode in (0, 3):
            lib.lbm_hip_debug_set(22, mode)
            m = LBM_online(LBMConfig(input_size=(96, 128), memory_size=4).to_namespace())
            m.load_state_dict(synthetic_state_dict(m, seed=42, preset="peaked"), strict=True)
            m = m.cuda().eval()
            got = []
            with torch.no_grad():
                m.init(pts, video[:, 0])
                for t in range(1, T):
                    if t == 2:
                        hip.profile_begin()
                    c, v, r, _ = m.online_forward(video[:, t])
                    if t == 2:
                        names = {p["name"] for p in hip.profile_end() if p["name"].startswith("msda")}
                        assert names == ({"msda_window", "msda_taps"} if mode else {"msda_sample"}), names
                    got += [c.clone(), m.last_logits[0].clone(), m.last_logits[1].clone()]
                got += [m._stream.clone(), m._coll.clone()]
            runs.append(got)
    finally:
        lib.lbm_hip_debug_set(22, 3)
    for i, (x, y) in enumerate(zip(*runs)):
        assert torch.equal(x.view(torch.int32), y.view(torch.int32)), f"entry {i}"


# ---- f16-storage mode (step precision 3): operators on f16 data equal the fp32 operators on the widened data, bit for bit ----

def test_f16_value_maps_give_the_rows_of_the_widened_maps(hip):
    """msda_window / msda_taps reading f16 value maps (the f16-storage mode) against the fp32 kernels on the same values widened to fp32:
    the widening is exact and the arithmetic identical, so every row must match bit for bit."""
    g = torch.Generator(device="cpu").manual_seed(77)
    shapes = [(96, 128), (48, 64), (24, 32), (12, 16)]
    L = sum(h * w for h, w in shapes)
    for B, N, k in ((1, 50, 9), (8, 12, 4), (1, 90, 1)):
        v16 = torch.randn(B, L, 512, generator=g).cuda().half()              # a 512-wide interleaved map: columns 256..511 are used
        offlog = torch.randn(B * N, 48 * 9, generator=g)
        offlog[:, :32 * 9] *= 2.5
        centre = torch.rand(B * N, 1, 2, generator=g) * torch.tensor([512.0, 384.0])
        side = 3 if k == 9 else 2 if k == 4 else 1
        cells = torch.stack([torch.tensor([float(i % side), float(i // side)]) for i in range(k)], 0) * 4.0
        ref = ((centre // 4.0) * 4.0 + 2.0 + cells.unsqueeze(0)).cuda().contiguous()
        a = hip.msda_sample(v16, shapes, offlog.cuda(), ref, B, N, k, 9, 512.0, 384.0, vcol=256)
        b = hip.msda_sample(v16.float(), shapes, offlog.cuda(), ref, B, N, k, 9, 512.0, 384.0, vcol=256)
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (B, N, k)
        assert float(a.abs().max()) > 0


@pytest.mark.parametrize("R,S,head", [(40, 12, 5), (3000, 48, 17), (9, 70, 0)])
def test_f16_memories_read_out_like_the_widened_memories(hip, R, S, head):
    """lbm_hip_attn_readout_h16 (memories stored as f16) against lbm_hip_attn_readout on the widened sequence: bit for bit, both
    wave mappings, masks and ring order included."""
    g = torch.Generator(device="cpu").manual_seed(R + S)
    U = (torch.randn(R, 4, 256, generator=g) * 0.15).cuda()
    seq = torch.randn(R, S, 256, generator=g).cuda().half()
    temb = (torch.randn(S, 256, generator=g) * 0.5).cuda()
    ma = (torch.rand(R, S, generator=g) < 0.3).cuda().to(torch.uint8)
    mb = (torch.rand(R, S, generator=g) < 0.2).cuda().to(torch.uint8)
    ma[0] = 1
    a, va = hip.attn_readout(U, seq, S, ring_head=head, temb=temb, mask_a=ma, mask_b=mb, want_valid=True)
    b, vb = hip.attn_readout(U, seq.float(), S, ring_head=head, temb=temb, mask_a=ma, mask_b=mb, want_valid=True)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32)) and torch.equal(va, vb)


@pytest.mark.parametrize("M,N", [(40000, 1536), (33000, 256), (70001, 520)])
def test_single_plane_row_streaming_product_with_f16_output(hip, M, N):
    """The value projection of the f16-storage mode: hi planes only, plain f16 output (ys_plane = 0, no fp32 copy).  The row-streaming
    kernel (debug key 2 = 512) against the tiled one (256): same MFMA order over k, so the same f16 bits; and both against fp64."""
    lib = hip.load()
    r = np.random.default_rng(M + N)
    x = torch.from_numpy(r.normal(size=(M, 256)).astype(np.float32)).cuda()
    w = torch.from_numpy((r.normal(size=(N, 256)) * 0.1).astype(np.float32)).cuda()
    b = torch.from_numpy(r.normal(size=(N,)).astype(np.float32)).cuda()
    xs, ws = _split(hip, x), hip.SplitWeight(w)
    outs = {}
    try:
        for flag in (256, 512):
            lib.lbm_hip_debug_set(2, flag)
            y = torch.zeros(M, N, device="cuda", dtype=torch.float16)
            hip.profile_begin()
            hip.linear_ss_raw(A=xs.hi, a_plane=xs.plane, B=ws.hi, b_plane=0, bias=b.data_ptr(), Ys=y.data_ptr(), ys_plane=0, M=M, N=N, K=256,
                              lda=256, ldb=256, ldys=N, batch=1, planes=1, res_scale=1.0)
            names = {p["name"] for p in hip.profile_end()}
            assert names == ({"gemm_rh"} if flag == 512 else {"gemm_sh<128,128,2,4,2>"}), names
            outs[flag] = y
    finally:
        lib.lbm_hip_debug_set(2, 0)
    assert torch.equal(outs[256].view(torch.int16), outs[512].view(torch.int16))
    want = x.half().double() @ w.half().double().t() + b.double()
    err = (outs[512].double() - want).abs().max().item()
    assert err < 2e-2 * float(want.abs().max()) / 10 + 5e-3, err          # f16 output: 2^-11 relative


@pytest.mark.parametrize("M,N", [(33000, 1536), (5000, 1024), (256, 32), (40001, 96), (70000, 2048)])
def test_rows_stationary_product_equals_the_tiled_product(hip, M, N):
    """gemm_rs EPI 2 (activation rows stationary in registers, weights streamed, 16-byte row stores straight from the accumulators:
    the value projection's form) against the tiled kernel: same split arithmetic in the same order, so the same bits."""
    lib = hip.load()
    r = np.random.default_rng(M * 7 + N)
    xs = _split(hip, torch.from_numpy(r.normal(size=(M, 256)).astype(np.float32)).cuda())
    ws = hip.SplitWeight(torch.from_numpy((r.normal(size=(N, 256)) * 0.1).astype(np.float32)).cuda())
    b = torch.from_numpy(r.normal(size=(N,)).astype(np.float32)).cuda()
    outs = {}
    try:
        for flag, want in ((256, None), (1024, "gemm_rs_rows")):
            lib.lbm_hip_debug_set(2, flag)
            for bias in (b, None):
                hip.profile_begin()
                y, _ = hip.linear_ss(xs, ws, bias)
                names = {p["name"] for p in hip.profile_end()}
                assert want is None or names == {want}, names
                outs[(flag, bias is None)] = y
    finally:
        lib.lbm_hip_debug_set(2, 0)
    for nb in (False, True):
        assert torch.equal(outs[(256, nb)].view(torch.int32), outs[(1024, nb)].view(torch.int32)), (M, N, nb)
