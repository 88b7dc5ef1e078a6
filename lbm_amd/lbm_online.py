"""Stateful online tracker (reference: lbm/models/lbm_online.py:10-200), generalised to B streams.

``init`` / ``update`` / ``online_forward`` keep the reference signatures and return values.  Differences on purpose
(MI355X-first, semantics preserved):

* the two per-point memories are **ring buffers** in HBM: a frame overwrites the oldest slot in place instead of
  ``torch.cat``-shifting 2 x (N, M, C) floats and cloning the whole state five times per frame
  (lbm_online.py:133-146).  ``collision_dist`` / ``stream_dist`` / ``vis_mask`` / ``mem_mask`` are exposed as
  properties that materialise the reference's logical (oldest-first) order on demand;
* no host synchronisation inside a frame (the reference's ``.item()`` / ``nonzero`` / ``assert`` syncs are gone);
* ``init_streams`` / ``online_forward_batch`` run B independent streams in one step (BASELINE configs 3-4).
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import hip
from .lbm import LBM


class LBM_online(LBM):
    def __init__(self, args, with_obj_tracker: bool = False):
        super().__init__(args=args)
        self.num_query = None
        self.t = 0
        self.N = 0
        self.B = 1
        self.query = None           # (B, N, C)
        self.query_points = None
        self._stream = None         # (B, N, M, C) physical ring order
        self._coll = None
        self._vis = None            # (B, N, M) uint8
        self._mem = None
        self._head = 0
        self.img_size = None
        self.device = "cuda" if torch.cuda.is_available() else "cpu"
        # Optional HIP-graph replay of the whole frame step (encoder + tracker, one graph per ring-buffer head): takes the
        # ~180 launches per frame off the host.  Off by default (the step is GPU-bound; measured 1.99 vs 1.98 ms at C2);
        # LBM_AMD_GRAPH=1 enables it.
        import os
        self.use_graph = os.environ.get("LBM_AMD_GRAPH", "0") == "1"
        self.graph_max_rows = 4096
        self._graphs = {}
        self._gkey = None
        self._g_in = None
        self._g_tok = None
        self._eager_done = None
        self._g_streams = []
        self._g_fork = None
        self._g_join = []
        # frame pipeline (prefetch()): encoder stream, two token / workspace slots, hand-over events
        self._pipe = None
        self._pf = {}               # frame key -> slot, for the (at most two) frames encoded ahead
        self.pipelined_frames = 0   # frames whose encoder ran ahead on the encoder stream
        if with_obj_tracker:
            from .object_tracker import LBMObjTracker
            self.objtracker = LBMObjTracker(args)

    @property
    def frame_layout(self) -> str:
        """Layout of uint8 frames: "chw" (B,3,H,W) (default), "hwc_rgb" or "hwc_bgr" (B,H,W,3) -- a decoder's bytes as they are,
        a quarter of the float frame's bytes; float frames are always (B,3,H,W) as in the reference."""
        return self.backbone.frame_layout

    @frame_layout.setter
    def frame_layout(self, v: str) -> None:
        if v not in hip.U8_LAYOUTS:
            raise ValueError(f"frame_layout must be one of {sorted(hip.U8_LAYOUTS)}")
        self.backbone.frame_layout = v

    def _hw(self, frames: torch.Tensor):
        return hip.frame_hw(frames, self.backbone.frame_layout)

    # ---- logical (oldest-first) views of the ring state, reference layout (1|B, N, M, ...) ----
    def _logical(self, x: Optional[torch.Tensor]):
        if x is None:
            return None
        M = x.shape[2]
        order = (torch.arange(M, device=x.device) + self._head) % M
        return x.index_select(2, order)

    @property
    def stream_dist(self):
        v = self._logical(self._stream)
        return None if v is None else v.float()

    @property
    def collision_dist(self):
        v = self._logical(self._coll)
        return None if v is None else v.float()

    @property
    def vis_mask(self):
        v = self._logical(self._vis)
        return None if v is None else v.bool()

    @property
    def mem_mask(self):
        v = self._logical(self._mem)
        return None if v is None else v.bool()

    def _state_dtype(self) -> torch.dtype:
        from .engine import state_dtype
        return state_dtype(self.transformer.precision)

    def _alloc_state(self, B: int, N: int) -> None:
        dev, M, Cc = self.device, self.memory_size, self.embed_dim
        dt = self._state_dtype()             # fp32, or f16 in the f16-storage mode (precision "f16s")
        self._stream = torch.zeros(B, N, M, Cc, device=dev, dtype=dt)
        self._coll = torch.zeros(B, N, M, Cc, device=dev, dtype=dt)
        self._vis = torch.ones(B, N, M, device=dev, dtype=torch.uint8)
        self._mem = torch.ones(B, N, M, device=dev, dtype=torch.uint8)
        self._head = 0

    def _to_net_px(self, pts: torch.Tensor) -> torch.Tensor:
        H, W = self.img_size
        out = pts.clone().float()
        out[..., 1] = (out[..., 1] / H) * self.size[0]
        out[..., 0] = (out[..., 0] / W) * self.size[1]
        return out

    # ---- reference API (B = 1) ----------------------------------------------------------------
    @torch.no_grad()
    def init(self, query_points: torch.Tensor, first_frame: torch.Tensor) -> None:
        """lbm_online.py:27-46 — query_points (N,2) xy in frame pixels; first_frame (1,3,H,W) in [0,255]."""
        self.query_points = query_points
        self.N = query_points.shape[0]
        self.B = 1
        self.img_size = self._hw(first_frame)
        self.device = first_frame.device
        self.t = 0
        self.query = None
        tok = self.backbone.encode_tokens(first_frame)
        update_mask = torch.ones(self.N, device=self.device, dtype=torch.bool)
        remove_mask = torch.zeros(self.N, device=self.device, dtype=torch.bool)
        self.update(self.query_points, tok, update_mask, remove_mask)

    @torch.no_grad()
    def update(self, query_points, frame_feat, update_mask, remove_mask) -> None:
        """lbm_online.py:48-112 — add / remove tracked points.  ``frame_feat`` is either the (1,L,C) token buffer
        of the current frame or the (1,C,h,w) map ``online_forward`` returned."""
        N = query_points.shape[0]
        if N > 0:
            if frame_feat.dim() == 4:       # (1,C,h,w) view -> token-major
                frame_feat = frame_feat.permute(0, 2, 3, 1).reshape(1, self.h * self.w, self.embed_dim)
            dev = frame_feat.device
            self.device = dev
            # every point is sampled (N bilinear reads) and the device picks the rows it needs: no boolean indexing,
            # no .item(), nothing the host has to wait for (the reference syncs three times here, lbm_online.py:57-83)
            pts = self._to_net_px(query_points.to(dev)).unsqueeze(0).contiguous()
            new_q = self.backbone.sample_points(frame_feat, pts)            # (1, N, C)
            old = None
            if self.query is not None and self.N > 0:
                old = (self.query, self._stream, self._coll, self._vis, self._mem)
                remove = remove_mask[: self.N].to(dev)
            else:
                remove = None
                self._head = 0
            # the ring head is untouched: kept rows move in their physical slot order
            dt = self._state_dtype()
            if old is not None and dt != torch.float32:      # f16-storage mode: the (rare) point-set edit goes through fp32 copies
                old = (old[0], old[1].float(), old[2].float(), old[3], old[4])
            self.query, self._stream, self._coll, self._vis, self._mem = hip.update_points(
                remove, update_mask.to(dev), old, new_q, self.memory_size)
            if dt != torch.float32:
                self._stream, self._coll = self._stream.to(dt), self._coll.to(dt)
            self.query_points = query_points
        self.N = N
        self.B = 1

    @torch.no_grad()
    def online_forward(self, frame: torch.Tensor):
        """lbm_online.py:114-158 — frame (1,3,H,W) in [0,255] -> (coord (N,2) frame px, vis (N,) bool, rho (N,), f_t (1,C,h,w))."""
        if self.query is None:
            raise RuntimeError("online_forward called before init()/update() registered any query point")
        coords, vis, rho, tok = self.online_forward_batch(frame)
        f_t = tok[:, : self.h * self.w].view(-1, self.h, self.w, self.embed_dim).permute(0, 3, 1, 2)
        return coords[0], vis[0], rho[0], f_t

    # ---- multi-stream API ---------------------------------------------------------------------
    @torch.no_grad()
    def init_streams(self, query_points: torch.Tensor, first_frames: torch.Tensor) -> None:
        """query_points (B,N,2) xy frame pixels; first_frames (B,3,H,W).  B independent streams, one state."""
        B, N = query_points.shape[:2]
        self.B, self.N = B, N
        self.img_size = self._hw(first_frames)
        self.device = first_frames.device
        self.t = 0
        self.query_points = query_points
        tok = self.backbone.encode_tokens(first_frames)
        self.query = self.backbone.sample_points(tok, self._to_net_px(query_points.to(self.device)).contiguous())
        self._alloc_state(B, N)

    # ---- frame pipeline ------------------------------------------------------------------------
    @staticmethod
    def _frame_key(frames: torch.Tensor):
        # address, geometry and the tensor's version counter: an in-place write to the frame after prefetch() changes the key
        # (tensors created under torch.inference_mode() have no version counter: edits of those are not detected)
        return (frames.data_ptr(), tuple(frames.shape), tuple(frames.stride()), frames.dtype, hip.tensor_version(frames))

    @torch.no_grad()
    def prefetch(self, frames: torch.Tensor) -> None:
        """Start on the frame the NEXT online_forward / online_forward_batch call will be given.

        Everything that does not depend on the tracked points -- the encoder, the six value maps and the correlation
        maps, about half of a frame's GPU time with one stream of ~1000 points -- is enqueued on a second HIP stream
        into the other of two token / workspace slots, so it overlaps the per-point chain of the current frame, whose
        ~120 small dependent launches leave most of the 256 CUs idle.  Results are identical to the unpipelined call
        (same kernels on the same data); only the time at which they run changes.  Measured on MI355X boxes: -10 % step
        time at 1024 points on most, +30 % on one (two queues of one process getting in each other's way), so a caller
        that cares should time both forms on its machine, as bench.py does.  The caller must not modify
        ``frames`` until the matching online_forward call has been made.  No counterpart in the reference, whose
        frame loop is strictly serial (lbm_online.py:114-158, tools/benchmark_torch.py:71-76)."""
        if self.query is None or not frames.is_cuda:
            return
        main = torch.cuda.current_stream()
        NS = 2
        key = (frames.shape[0], self.N, self.memory_size, frames.device)
        if self._pipe is None or self._pipe["key"] != key:
            L = self.transformer.cfg.num_tokens
            dev = frames.device
            self._pipe = {
                "key": key, "stream": torch.cuda.Stream(device=dev), "next": 0,
                "tok": [torch.empty(frames.shape[0], L, self.embed_dim, device=dev) for _ in range(NS)],
                "tok_s": [None] * NS,
                "enc_done": [torch.cuda.Event() for _ in range(NS)],
            }
        p = self._pipe
        slot = p["next"]
        p["next"] = (slot + 1) % NS
        enc = p["stream"]
        # the frame was produced on the caller's stream, and the slot's previous reader (the per-point phase two
        # frames back) was enqueued there as well: one wait covers both
        enc.wait_stream(main)
        frames.record_stream(enc)
        H, W = self._hw(frames)
        with torch.cuda.stream(enc):
            if p["tok_s"][slot] is None and self.backbone.packed().precision in ("split", "f16"):
                p["tok_s"][slot] = hip.SplitTensor(tuple(p["tok"][slot].shape), frames.device)
            tok = self.backbone.encode_tokens(frames, p["tok"][slot], split_out=p["tok_s"][slot])
            self.transformer.get_engine().step(tok, self.query, self._stream, self._coll, self._vis, self._mem,
                                               ring_head=0, update_state=False, img_hw=(H, W),
                                               tokens_split=p["tok_s"][slot], phase=1, ws_slot=slot)
            p["enc_done"][slot].record(enc)
        self._pf = {k: v for k, v in self._pf.items() if v != slot}
        self._pf[self._frame_key(frames)] = slot

    @torch.no_grad()
    def online_forward_batch(self, frames: torch.Tensor, tokens: Optional[torch.Tensor] = None, out=None):
        """frames (B,3,H,W) -> coords (B,N,2) frame px, vis (B,N) bool, rho (B,N) in (0,1), tokens (B,L,C).

        ``out``: optional dict of caller-owned result tensors ("coords" (B,N,2), "vis", "rho", "rho_prob" (B,N) fp32, "offset"
        (B,N,2), "vis_flag" (B,N) bool) the step writes instead of allocating -- e.g. slices of a whole-run track log.

        Lifetime of the returned tokens: for a frame that was encoded ahead (prefetch) they are a VIEW of one of the two
        pipeline slots and are overwritten by the prefetch after next; a caller that keeps them longer (e.g. for a later
        update()) passes ``tokens=`` and gets a copy there -- the reference returns a fresh tensor every frame
        (lbm_online.py:125-126).  Unpipelined frames return a fresh tensor (or ``tokens`` itself)."""
        H, W = self._hw(frames)
        if self._pf:
            slot = self._pf.pop(self._frame_key(frames), None)
            if slot is None and len(self._pf) > 1:
                # frames encoded ahead and never asked for: only the most recent prefetch can still be the next frame
                last = next(reversed(self._pf))
                self._pf = {last: self._pf[last]}
            p = self._pipe
            if slot is not None and p is not None and p["key"] == (frames.shape[0], self.N, self.memory_size, frames.device):
                self.pipelined_frames += 1
                torch.cuda.current_stream().wait_event(p["enc_done"][slot])
                tok = p["tok"][slot]
                out = self.transformer.get_engine().step(tok, self.query, self._stream, self._coll, self._vis, self._mem,
                                                         ring_head=self._head, queried=None, update_state=True, img_hw=(H, W),
                                                         vis_threshold=self.visibility_treshold,
                                                         tokens_split=p["tok_s"][slot], phase=2, ws_slot=slot, out=out)
                self._head = (self._head + 1) % self.memory_size
                self.t += 1
                self.last_logits = (out["vis"], out["rho"])
                if tokens is not None:
                    tokens.copy_(tok)
                    tok = tokens
                return out["coords"], out["vis_flag"], out["rho_prob"], tok
        if self._pipe is not None:      # the encoder's scratch is shared: never run it on two streams at once
            torch.cuda.current_stream().wait_stream(self._pipe["stream"])
        if self.use_graph and out is None and frames.is_cuda and self.B * self.N <= self.graph_max_rows and self.N > 0:
            res = self._forward_graph(frames)
            if res is not None:
                return res
        # a frame encoded ahead owns its workspace slot until it is consumed: an unpipelined frame in between uses a third
        busy = set(self._pf.values())
        ws_slot = next(i for i in range(8) if i not in busy)
        tok = self.backbone.encode_tokens(frames, tokens)
        out = self.transformer.get_engine().step(tok, self.query, self._stream, self._coll, self._vis, self._mem,
                                                 ring_head=self._head, queried=None, update_state=True, img_hw=(H, W),
                                                 vis_threshold=self.visibility_treshold,
                                                 tokens_split=self.backbone.last_tokens_split, ws_slot=ws_slot, out=out)
        self._head = (self._head + 1) % self.memory_size
        self.t += 1
        self.last_logits = (out["vis"], out["rho"])
        return out["coords"], out["vis_flag"], out["rho_prob"], tok

    def _forward_graph(self, frames: torch.Tensor):
        """Replay (capturing on first use) the HIP graph of this frame step; None = not available, run eagerly."""
        key = (tuple(frames.shape), self.N, self.memory_size, self._stream.data_ptr(), self.query.data_ptr(), hip.tensor_version(self.query),
               float(self.visibility_treshold), self.transformer.precision, self.backbone.precision)
        if self._gkey != key:
            self._graphs, self._gkey, self._eager_done = {}, key, False
            self._g_in = torch.empty_like(frames)            # same dtype and layout: uint8 frames stay uint8
            L = self.transformer.cfg.num_tokens
            self._g_tok = torch.empty(frames.shape[0], L, self.embed_dim, device=frames.device, dtype=torch.float32)
            self._g_fork = torch.cuda.Event()
            self._g_join = [torch.cuda.Event() for _ in range(3)]
        if not self._eager_done:
            # the first frame of a configuration runs eagerly: it packs weights and sizes every persistent buffer
            self._eager_done = True
            return None
        H, W = self._hw(frames)
        self._g_in.copy_(frames)
        head = self._head
        if head not in self._graphs:
            try:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    tok = self.backbone.encode_tokens(self._g_in, self._g_tok)
                    out = self.transformer.get_engine().step(tok, self.query, self._stream, self._coll, self._vis, self._mem,
                                                             ring_head=head, queried=None, update_state=True, img_hw=(H, W),
                                                             vis_threshold=self.visibility_treshold,
                                                             tokens_split=self.backbone.last_tokens_split)
                self._graphs[head] = (g, out, self.transformer.get_engine()._qc)    # the cache entry keeps u0 / the split twin alive
            except Exception as e:          # capture not possible here: say so once and stay eager
                import warnings
                warnings.warn(f"lbm_amd: HIP graph capture failed ({e}); running eagerly")
                self.use_graph = False
                torch.cuda.synchronize()
                return None
        g, out, qc = self._graphs[head]
        if qc is not self.transformer.get_engine()._qc:
            # another step replaced the engine's per-query cache (its tensors are baked into the captured graphs): start over
            self._graphs, self._eager_done = {}, False
            return None
        g.replay()
        self._head = (self._head + 1) % self.memory_size
        self.t += 1
        vis, rho = out["vis"].clone(), out["rho"].clone()
        self.last_logits = (vis, rho)
        return out["coords"].clone(), torch.sigmoid(vis) > self.visibility_treshold, torch.sigmoid(rho), self._g_tok

    @torch.no_grad()
    def online_forward_obj(self, frame, bboxes, scores, labels):
        """lbm_online.py:160-200 — one frame of object tracking: track the points planted in the known objects, associate
        this frame's detections (bboxes (n,4) xyxy frame px, scores (n,), labels (n,); None = no detector run on this
        frame) with the tracklets through those points, then plant / drop points as the tracklets changed.
        Returns (coords, visibility, pred_track_instances) -- the point predictions are those made BEFORE the update."""
        coords = visibility = tok = pred_track_instances = None
        if self.N > 0:
            coords, visibility, _, tok = self.online_forward_batch(frame)
            coords, visibility = coords[0], visibility[0]
        if bboxes is not None:
            # association runs on the host (object_tracker.py header): one copy of the predictions per frame
            host_coords = None if coords is None else coords.cpu()
            host_vis = None if visibility is None else visibility.cpu()
            pred_track_instances = self.objtracker.track(self.t, bboxes, scores, labels, host_coords, host_vis)
            memo = self.objtracker.memo
            update_mask = memo[5].flatten()                                   # (M * npt,)  tracklets after the update
            remove_mask = self.objtracker.points_remove_mask.flatten()        # (M' * npt,) tracklets before the drop
            query_points = memo[4].reshape(-1, 2)
            if self.N == 0:          # nothing was tracked yet: this frame has not been encoded
                self.img_size = self._hw(frame)
                self.device = frame.device
                tok = self.backbone.encode_tokens(frame)
            self.update(query_points, tok, update_mask, remove_mask)
            self.query_points = query_points
        return coords, visibility, pred_track_instances


class LBM_export(LBM):
    """Placeholder for import compatibility: the ONNX/TensorRT export variant (lbm_online.py:203-353) is out of
    scope (and broken in the reference at this commit)."""

    def forward(self, *a, **k):
        raise NotImplementedError("ONNX/TensorRT export is out of scope")
