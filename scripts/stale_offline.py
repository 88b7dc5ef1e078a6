"""Frame pipeline + block correlation kernel (variant 1, UNMODIFIED): what exactly is in a wrong map cell?

The traced variant of the kernel (stale_trace.py) does not show the effect, so this script leaves the kernel alone and
analyses the damage afterwards.  After every pipelined frame (device synchronised) the stored maps are compared with the
serial run's.  For every wrong 2x2 block the SAME kernel is then run serially on modified inputs, one hypothesis at a time:
for each of the block's 21 input rows, the token row, its norm, or both taken from the slot's PREVIOUS generation (the frame
that used the slot two frames earlier).  Every arithmetic step of the kernel is per channel, so hypotheses are compared per
128-byte line (32 channels) of each (layer, plane, cell):
    'stale-input(row k, tok|norm|both)'   the stored bits equal the kernel's output for that hypothesis,
    'old-output'                           the stored bits equal the map of the previous generation (the store never landed),
    'unexplained'                          neither.
usage: stale_offline.py [reps]"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from lbm_amd import hip  # noqa: E402
from lbm_amd.config import LBMConfig  # noqa: E402
from lbm_amd.lbm_online import LBM_online  # noqa: E402
from lbm_amd.synth import grid_queries, synthetic_state_dict, synthetic_video  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lib = hip.load()
lib.lbm_hip_debug_set(9, 1)
T, H, W = 16, 240, 320
video = synthetic_video(T, H, W, seed=21).cuda()
frames = [video[:, t].contiguous() for t in range(T)]
pts = grid_queries(H, W, 32).cuda()
h, w = 96, 128
P0 = h * w
L = P0 + P0 // 4 + P0 // 16 + P0 // 64


def bits(x):
    return x.contiguous().view(torch.int16)


def up_pair(dst, lvl, n):
    src = max((dst + 0.5) / (1 << lvl) - 0.5, 0.0)
    i0 = int(src)
    return i0, i0 + (1 if i0 < n - 1 else 0)


def block_rows(y0, x0):
    """the 21 token rows the block kernel loads for the 2x2 block at (y0, x0), with the name of each"""
    rows = [((y0 + dy) * w + x0 + dx, f"c0[{dy}][{dx}]") for dy in range(2) for dx in range(2)]
    hl, wl, st = h // 2, w // 2, P0
    iy, ix = y0 // 2, x0 // 2
    ry = [max(iy - 1, 0), iy, min(iy + 1, hl - 1)]
    rx = [max(ix - 1, 0), ix, min(ix + 1, wl - 1)]
    rows += [(st + ry[u] * wl + rx[v], f"p1[{u}][{v}]") for u in range(3) for v in range(3)]
    st = P0 + hl * wl
    for l in (2, 3):
        hl, wl = h >> l, w >> l
        ya, yb = up_pair(y0, l, hl)
        xa, xb = up_pair(x0, l, wl)
        rows += [(st + yy * wl + xx, f"p{l}[{a}][{b}]") for a, yy in enumerate((ya, yb)) for b, xx in enumerate((xa, xb))]
        st += hl * wl
    return rows


def kernel_maps(tok, inv, mix):
    """the block kernel on (tok, inv) as given -> split planes (3, 2, P0, 256) f16, exactly what the step would store"""
    F = torch.empty(3, 1, P0, 256, device="cuda")
    m = (C.c_float * 12)(*mix)
    lib.lbm_hip_debug_set(10, 1)
    hip._check(lib.lbm_hip_corr_maps(tok.data_ptr(), inv.data_ptr(), F.data_ptr(), m, 3, 1, h, w, hip._stream()), "corr_maps")
    lib.lbm_hip_debug_set(10, 0)
    o = F[:, 0].clamp(-65000.0, 65000.0)
    hi = o.half()
    lo = ((o - hi.float()) * 2048.0).half()
    return torch.stack((hi, lo), 1)


def run(pipelined, ref=None):
    model = LBM_online(LBMConfig(input_size=(384, 512), memory_size=12).to_namespace())
    model.load_state_dict(synthetic_state_dict(model, seed=42), strict=True)
    model = model.cuda().eval()
    eng = model.transformer.get_engine()
    mix = eng.w.corr_mix
    rec, hist, slot_of = [], {}, {}
    n = dict(frames=0, wrong_frames=0, wrong_blocks=0)
    with torch.no_grad():
        model.init(pts, frames[0])
        for t in range(1, T):
            if pipelined and t + 1 < T:
                slot_of[t + 1] = 0 if model._pipe is None else model._pipe["next"]
                model.prefetch(frames[t + 1])
            model.online_forward(frames[t])
            torch.cuda.synchronize()
            F = eng.debug_buffer("corr_maps", (3, 2, P0, 256), torch.float16)
            if ref is None:
                rec.append(F.clone())
                continue
            slot = slot_of.get(t)
            if slot is None:
                continue
            n["frames"] += 1
            tok = model._pipe["tok"][slot]                       # (1, L, 256), final (pipe_snap.py)
            inv = eng.debug_buffer("inv_norm", (L,))
            wrong = (bits(F) != bits(ref[t - 1])).any(3).any(1).any(0)
            if bool(wrong.any()):
                n["wrong_frames"] += 1
                cells = torch.nonzero(wrong).flatten().tolist()
                blocks = sorted({((p // w) // 2, (p % w) // 2) for p in cells})
                n["wrong_blocks"] += len(blocks)
                base = kernel_maps(tok, inv, mix)
                base_ok = bool((bits(base) == bits(ref[t - 1])).all())
                prev = hist.get(slot)
                print(f"frame {t} slot {slot}: {len(cells)} wrong cells in {len(blocks)} blocks; the kernel re-run serially on the final tokens "
                      f"and norms reproduces the serial maps: {base_ok}; previous generation known: {prev is not None}", flush=True)
                for by, bx in blocks[:6]:
                    y0, x0 = by * 2, bx * 2
                    pcs = [(y0 + dy) * w + x0 + dx for dy in range(2) for dx in range(2)]
                    got = F[:, :, pcs].reshape(3, 2, 4, 8, 32)                      # layer, plane, cell, line, 32 channels
                    want = ref[t - 1][:, :, pcs].reshape(3, 2, 4, 8, 32)
                    bad = (bits(got) != bits(want)).view(3, 2, 4, 8, 32).any(-1)       # (layer, plane, cell, line)
                    where = [(int(a), int(b), (int(c) // 2, int(c) % 2), int(d)) for a, b, c, d in torch.nonzero(bad).tolist()]
                    print(f"  block (y0 {y0}, x0 {x0}) = workgroup {(by * (w // 2) + bx) // 4} of {P0 // 16}: wrong (layer, plane, cell, 128-B line): {where}", flush=True)
                    if prev is None:
                        continue
                    ptok, pinv, pF = prev
                    rws = [r for r, _ in block_rows(y0, x0)]
                    other = model._pipe["tok"][1 - slot]
                    dumps.append(dict(frame=t, slot=slot, y0=y0, x0=x0, rows=rws, names=[nm for _, nm in block_rows(y0, x0)],
                                      cur=tok[0, rws].cpu(), prev=ptok[0, rws].cpu(), other=other[0, rws].cpu(),
                                      inv=inv[rws].cpu(), pinv=pinv[rws].cpu(), got=F[:, :, pcs].cpu(), want=ref[t - 1][:, :, pcs].cpu(),
                                      old=pF[:, :, pcs].cpu(), mix=list(mix)))
                    expl = {}
                    old = pF[:, :, pcs].reshape(3, 2, 4, 8, 32)
                    for key in where:
                        a, b, c, d = key[0], key[1], key[2][0] * 2 + key[2][1], key[3]
                        if bool((bits(got[a, b, c, d]) == bits(old[a, b, c, d])).all()):
                            expl.setdefault(key, []).append("old-output")
                    for row, name in block_rows(y0, x0):
                        for kind in ("tok", "norm", "both"):
                            tk, iv = tok, inv
                            if kind in ("tok", "both"):
                                tk = tok.clone()
                                tk[0, row] = ptok[0, row]
                            if kind in ("norm", "both"):
                                iv = inv.clone()
                                iv[row] = pinv[row]
                            hyp = kernel_maps(tk, iv, mix)[:, :, pcs].reshape(3, 2, 4, 8, 32)
                            for key in where:
                                a, b, c, d = key[0], key[1], key[2][0] * 2 + key[2][1], key[3]
                                if bool((bits(got[a, b, c, d]) == bits(hyp[a, b, c, d])).all()):
                                    expl.setdefault(key, []).append(f"stale-input({name} row {row}, {kind})")
                    for key in where:
                        print(f"    {key}: {expl.get(key, ['unexplained'])}", flush=True)
            hist[slot] = (tok.clone(), inv.clone(), F.clone())
    return rec, n


dumps = []
ref, _ = run(False)
tot = {}
for rep in range(reps):
    _, st = run(True, ref)
    for k, v in st.items():
        tot[k] = tot.get(k, 0) + v
print("TOTAL", tot)
if dumps:
    import os
    os.makedirs("gpurun_out/r2", exist_ok=True)
    torch.save(dumps, "gpurun_out/r2/stale_dumps.pt")
