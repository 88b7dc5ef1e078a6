"""Frame pipeline + block correlation kernel: is a wrong map cell a STALE INPUT or a LOST STORE?

Correlation-map variant 2 (debug key 9 = 2) is the 2x2-block kernel with a trace: every wave records the token row and norm
it actually loaded for p1[2][2] (the one input only cell (1,1) consumes), the layer-0 value it computed for that cell, and
the XCC it ran on.  After every frame (device synchronised) the host compares
  (a) the traced input with the slot's final tokens / norms            -> mismatch = the wave loaded something else,
                                                                          and whether that is the slot's content two frames ago,
  (b) the stored map cell with the split of the traced value           -> mismatch = the store did not land (or was undone),
  (c) the stored maps with the serial run's                            -> which cells are wrong at all.
usage: stale_trace.py [reps] [poison 0|1]     poison: tokens, split twin, norms and maps of the slot are overwritten with
NaN patterns on the encoder stream right before every prefetch (a consumer that sees NaNs reads pre-kernel content)."""
import sys

import torch

sys.path.insert(0, ".")
from lbm_amd import hip  # noqa: E402
from lbm_amd.config import LBMConfig  # noqa: E402
from lbm_amd.lbm_online import LBM_online  # noqa: E402
from lbm_amd.synth import grid_queries, synthetic_state_dict, synthetic_video  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
poison = len(sys.argv) > 2 and sys.argv[2] == "1"
lib = hip.load()
lib.lbm_hip_debug_set(9, 2)
T, H, W = 16, 240, 320
video = synthetic_video(T, H, W, seed=21).cuda()
frames = [video[:, t].contiguous() for t in range(T)]
pts = grid_queries(H, W, 32).cuda()
h, w = 96, 128
P0 = h * w
NB = P0 // 4
bw = w // 2
blk = torch.arange(NB, device="cuda")
cell11 = ((blk // bw) * 2 + 1) * w + (blk % bw) * 2 + 1
traces = [torch.zeros(NB, 768, device="cuda") for _ in range(3)]


def bits(x):
    return x.contiguous().view(torch.int32) if x.dtype == torch.float32 else x.contiguous().view(torch.int16)


def run(pipelined, ref=None):
    model = LBM_online(LBMConfig(input_size=(384, 512), memory_size=12).to_namespace())
    model.load_state_dict(synthetic_state_dict(model, seed=42), strict=True)
    model = model.cuda().eval()
    eng = model.transformer.get_engine()
    L = model.transformer.cfg.num_tokens
    rec = []
    hist = {}          # slot -> (tok, inv, F) of the frame that used the slot before
    stats = dict(frames=0, wrong_frames=0, wrong_cells=0, in_bad=0, s_bad=0, st_bad=0, nan_in=0)
    slot_of = {}
    with torch.no_grad():
        lib.lbm_hip_debug_ptr(1, traces[2].data_ptr())
        model.init(pts, frames[0])
        for t in range(1, T):
            if pipelined and t + 1 < T:
                s = 0 if model._pipe is None else model._pipe["next"]
                if poison and model._pipe is not None and model._pipe["tok_s"][s] is not None and s in eng._ws:
                    with torch.cuda.stream(model._pipe["stream"]):
                        model._pipe["tok"][s].fill_(float("nan"))
                        model._pipe["tok_s"][s].planes.fill_(float("nan"))
                        for name, n in (("inv_norm", L * 4), ("corr_maps", 3 * P0 * 256 * 4)):
                            off = int(lib.lbm_hip_step_workspace_offset(eng._last_desc, name.encode()))
                            eng._ws[s][off:off + n].fill_(255)
                lib.lbm_hip_debug_ptr(1, traces[s].data_ptr())
                model.prefetch(frames[t + 1])
                slot_of[t + 1] = s
            lib.lbm_hip_debug_ptr(1, traces[2].data_ptr())      # an unpipelined frame's kernel must not touch a slot's trace
            c, v, r, f_t = model.online_forward(frames[t])
            torch.cuda.synchronize()
            slot = slot_of.get(t)
            ws_slot = slot if slot is not None else eng._last_slot
            F = eng.debug_buffer("corr_maps", (3, 2, P0, 256), torch.float16)
            if ref is None:
                rec.append(F.clone())
                continue
            if slot is None:
                continue
            stats["frames"] += 1
            tok = model._pipe["tok"][slot][0]
            inv = eng.debug_buffer("inv_norm", (L,))
            tr = traces[slot]
            t22 = tr[:, 514].long()
            in_bad = (bits(tr[:, :256]) != bits(tok[t22])).any(1)
            s_bad = bits(tr[:, 512]) != bits(inv[t22])
            o = tr[:, 256:512].clamp(-65000.0, 65000.0)
            hi = o.half()
            lo = ((o - hi.float()) * 2048.0).half()
            st_bad = (bits(F[0, 0, cell11]) != bits(hi)).any(1) | (bits(F[0, 1, cell11]) != bits(lo)).any(1)
            nan_in = torch.isnan(tr[:, :513]).any(1)
            hw = tr[:, 516:519].contiguous().view(torch.int32)
            moved = hw[:, 0] != hw[:, 1]
            stats["moved"] = stats.get("moved", 0) + int(moved.sum())
            stats["max_ticks"] = max(stats.get("max_ticks", 0), int(hw[:, 2].max()))
            stats["over_100us"] = stats.get("over_100us", 0) + int((hw[:, 2] > 10000).sum())
            if bool(moved.any()):
                b = int(torch.nonzero(moved)[0])
                print(f"  frame {t}: {int(moved.sum())} waves ended on another HW_ID than they started on, e.g. block {b}: {int(hw[b, 0]):#x} -> {int(hw[b, 1]):#x}, {int(hw[b, 2])} ticks", flush=True)
            wrong = (bits(F) != bits(ref[t - 1])).any(3).any(1).any(0)          # (P0,) cells that differ from the serial run
            stats["in_bad"] += int(in_bad.sum()); stats["s_bad"] += int(s_bad.sum()); stats["st_bad"] += int(st_bad.sum())
            stats["nan_in"] += int(nan_in.sum())
            cells = torch.nonzero(wrong).flatten().tolist()
            if cells or bool(in_bad.any()) or bool(s_bad.any()) or bool(st_bad.any()):
                stats["wrong_frames"] += 1 if cells else 0
                stats["wrong_cells"] += len(cells)
                flagged = sorted(set(torch.nonzero(in_bad | s_bad | st_bad).flatten().tolist()) |
                                 {(p // w // 2) * bw + (p % w) // 2 for p in cells})
                prev = hist.get(slot)
                for b in flagged[:12]:
                    p11 = int(cell11[b])
                    y0, x0 = (b // bw) * 2, (b % bw) * 2
                    wrong_here = [(int(p) // w - y0, int(p) % w - x0) for p in cells if (int(p) // w) // 2 == b // bw and (int(p) % w) // 2 == b % bw]
                    msg = (f"  frame {t} slot {slot} block {b} (y0 {y0}, x0 {x0}; workgroup {int(tr[b, 515])} of {NB // 4}, XCC {int(tr[b, 513])}): "
                           f"cells wrong vs serial {wrong_here}; input row {int(t22[b])} loaded-vs-final {'DIFFERS' if in_bad[b] else 'equal'}, "
                           f"norm {'DIFFERS' if s_bad[b] else 'equal'}, stored-vs-computed {'DIFFERS' if st_bad[b] else 'equal'}")
                    if in_bad[b]:
                        d = bits(tr[b, :256]) != bits(tok[t22[b]])
                        msg += f"; {int(d.sum())} of 256 channels differ (first {int(torch.nonzero(d)[0])}, last {int(torch.nonzero(d)[-1])})"
                        if prev is not None:
                            msg += f"; equals the slot's previous generation: {bool((bits(tr[b, :256]) == bits(prev[0][t22[b]])).all())}"
                        msg += f"; NaN in it: {bool(torch.isnan(tr[b, :256]).any())}"
                    if s_bad[b] and prev is not None:
                        msg += f"; norm equals previous generation: {bool(bits(tr[b, 512]) == bits(prev[1][t22[b]]))}, NaN: {bool(torch.isnan(tr[b, 512]))}"
                    if st_bad[b] and prev is not None:
                        msg += f"; stored cell equals previous generation: {bool((bits(F[0, :, p11]) == bits(prev[2][0, :, p11])).all())}"
                    print(msg, flush=True)
            hist[slot] = (tok.clone(), inv.clone(), F.clone())
    return rec, stats


ref, _ = run(False)
tot = {}
for rep in range(reps):
    _, st = run(True, ref)
    for k, v in st.items():
        tot[k] = max(tot.get(k, 0), v) if k == "max_ticks" else tot.get(k, 0) + v
    print(f"rep {rep}: {st}", flush=True)
print("TOTAL", "poison" if poison else "plain", tot)
