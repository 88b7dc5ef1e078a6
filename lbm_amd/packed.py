"""On-disk form of the packed kernel weights: ``lbm.pt`` -> one arena file -> one host-to-device copy.

Turning the reference checkpoint (``lbm.pt``, the state dict of lbm/models/lbm/lbm.py:12-68) into what the kernels read
takes float64 folding on the host (BatchNorm into the convolutions, query/key and value/output projections of every
attention into one matrix each, the encoder's projection o fusion products, merged products), ~250 small uploads and a
split into f16 planes per matrix.  The result only depends on the checkpoint, so it is kept on disk as ONE arena:

    bytes 0..7     magic  b"LBMWPK1\\n"
    bytes 8..15    little-endian u64: length H of the header
    bytes 16..16+H UTF-8 JSON: {"abi": LBM_HIP_ABI_VERSION, "format": 1, "source_digest": sha256 of the state dict,
                   "meta": {...geometry, correlation mix, conv shapes...},
                   "entries": {name: {"dtype": "f32"|"f16", "shape": [...], "offset": o, "nbytes": n}, ...}}
    then zero padding to a multiple of 4096, then the payload; every entry starts on a 256-byte boundary (the
    alignment the kernels' 16-byte vector loads and the LDS-DMA rows assume, with room to spare).

Loading reads the payload into one host buffer, copies it to HBM once and hands out views: no arithmetic, no per-tensor
allocation, and the f16 planes are exactly the bits the packer produced, so a model restored from the file computes
bit-identical results (tests/test_hip_parity.py).  The file is tied to the ABI version of the library that wrote it and,
when asked, to the digest of the checkpoint it came from.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import json
import os
import struct
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import numpy as np
import torch

MAGIC = b"LBMWPK1\n"
FORMAT = 1
ALIGN = 256
PAGE = 4096
_DTYPES = {"f32": torch.float32, "f16": torch.float16}
_NAMES = {v: k for k, v in _DTYPES.items()}


class PackedFormatError(RuntimeError):
    pass


def state_digest(sd: Dict[str, torch.Tensor]) -> str:
    """sha256 over (name, dtype, shape, bytes) of a state dict in key order: ties a packed file to its checkpoint."""
    h = hashlib.sha256()
    for name in sorted(sd):
        t = sd[name].detach().cpu().contiguous()
        h.update(name.encode())
        h.update(str(t.dtype).encode())
        h.update(str(tuple(t.shape)).encode())
        h.update(t.numpy().tobytes() if t.numel() else b"")
    return h.hexdigest()


# ---- the container: independent of the model (and of the GPU) ------------------------------------

def write_arena(path: str, entries: "OrderedDict[str, torch.Tensor]", meta: dict, abi: int, source_digest: str = "") -> int:
    """Writes the container; returns the payload size.  Atomic: the file appears under its name only when complete."""
    index, off = OrderedDict(), 0
    for name, t in entries.items():
        if t.dtype not in _NAMES:
            raise PackedFormatError(f"{name}: dtype {t.dtype} has no place in the container")
        nbytes = t.numel() * t.element_size()
        index[name] = {"dtype": _NAMES[t.dtype], "shape": list(t.shape), "offset": off, "nbytes": nbytes}
        off = (off + nbytes + ALIGN - 1) // ALIGN * ALIGN
    header = json.dumps({"abi": abi, "format": FORMAT, "source_digest": source_digest, "meta": meta, "entries": index,
                         "payload_bytes": off}).encode()
    start = (16 + len(header) + PAGE - 1) // PAGE * PAGE
    tmp = f"{path}.tmp{os.getpid()}"
    with open(tmp, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<Q", len(header)))
        f.write(header)
        f.write(b"\0" * (start - 16 - len(header)))
        for name, t in entries.items():
            e = index[name]
            f.seek(start + e["offset"])
            f.write(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes() if t.numel() else b"")
        f.truncate(start + off)
    os.replace(tmp, path)
    return off


def read_header(path: str) -> Tuple[dict, int]:
    """(header, payload start).  Raises PackedFormatError on anything that is not a complete file of this format."""
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        head = f.read(16)
        if len(head) < 16 or head[:8] != MAGIC:
            raise PackedFormatError(f"{path}: not a packed-weights file (bad magic)")
        (hlen,) = struct.unpack("<Q", head[8:])
        if hlen > size - 16:
            raise PackedFormatError(f"{path}: truncated header")
        try:
            header = json.loads(f.read(hlen).decode())
        except ValueError as e:
            raise PackedFormatError(f"{path}: unreadable header ({e})") from None
    if header.get("format") != FORMAT:
        raise PackedFormatError(f"{path}: container format {header.get('format')}, this build reads {FORMAT}")
    start = (16 + hlen + PAGE - 1) // PAGE * PAGE
    if size != start + header["payload_bytes"]:
        raise PackedFormatError(f"{path}: {size} bytes on disk, header describes {start + header['payload_bytes']}")
    for name, e in header["entries"].items():
        n = int(np.prod(e["shape"])) * (4 if e["dtype"] == "f32" else 2)
        if e["dtype"] not in _DTYPES or e["offset"] % ALIGN or n != e["nbytes"] or e["offset"] + n > header["payload_bytes"]:
            raise PackedFormatError(f"{path}: entry {name} is inconsistent")
    return header, start


def read_arena(path: str, device="cpu") -> Tuple[dict, Dict[str, torch.Tensor], torch.Tensor]:
    """(header, {name: view}, arena).  The payload is read into one buffer and moved to `device` in one copy."""
    header, start = read_header(path)
    n = header["payload_bytes"]
    host = torch.from_numpy(np.fromfile(path, dtype=np.uint8, count=n, offset=start)) if n else torch.empty(0, dtype=torch.uint8)
    arena = host.to(device) if str(device) != "cpu" else host
    views = {}
    for name, e in header["entries"].items():
        views[name] = arena[e["offset"]: e["offset"] + e["nbytes"]].view(_DTYPES[e["dtype"]]).view(*e["shape"])
    return header, views, arena


# ---- the model's packed weights <-> container entries ---------------------------------------------

def _conv_entries(out, prefix, conv):
    out[prefix + ".w"] = conv.w
    out[prefix + ".b"] = conv.b
    out[prefix + ".planes"] = conv.ws.planes


def collect(model) -> Tuple["OrderedDict[str, torch.Tensor]", dict]:
    """Everything the kernels read, from a model whose weights are packed (packs them now if they are not)."""
    enc = model.backbone.packed()
    trk = model.transformer.get_engine().w
    cfg = model.transformer.cfg
    ent: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    convs = {"stem": [enc.stem.kh, enc.stem.stride, enc.stem.pad]}
    _conv_entries(ent, "enc.stem", enc.stem)
    for i, (c1, c2, down) in enumerate(enc.blocks):
        for tag, c in (("c1", c1), ("c2", c2), ("down", down)):
            if c is not None:
                _conv_entries(ent, f"enc.blk{i}.{tag}", c)
                convs[f"blk{i}.{tag}"] = [c.kh, c.stride, c.pad]
    for i, (w, ws) in enumerate(zip(enc.lvl_w, enc.lvl_ws)):
        ent[f"enc.lvl{i}.w"] = w
        ent[f"enc.lvl{i}.planes"] = ws.planes
    ent["enc.lvl_b"] = enc.lvl_b
    ent["enc.pos"] = enc.pos
    for i, t in enumerate(trk.tensors):
        if t is not None:
            ent[f"trk.{i}"] = t
            if trk.split[i] is not None:
                ent[f"trk.{i}.planes"] = trk.split[i].planes
    meta = {"input_size": list(enc.size), "n_blocks": len(enc.blocks), "convs": convs, "n_layers": trk.n_layers, "n_table": trk.n,
            "memory_size": trk.memory_size, "corr_mix": trk.corr_mix, "corr_bias": trk.corr_bias,
            "topk": list(cfg.topk_per_layer), "embed_dim": cfg.embed_dim, "num_heads": cfg.num_heads,
            "num_neighbors": cfg.num_neighbors, "stride": cfg.stride}
    return ent, meta


def save_packed(model, path: str, source_digest: Optional[str] = None) -> int:
    """Packs (if needed) and writes the arena file.  source_digest: state_digest() of the checkpoint, computed here from
    the model's own state dict when not given."""
    from . import hip
    ent, meta = collect(model)
    if source_digest is None:
        source_digest = state_digest(model.state_dict())
    return write_arena(path, ent, meta, hip.header_constants()["LBM_HIP_ABI_VERSION"], source_digest)


def load_packed(model, path: str, expect_digest: Optional[str] = None) -> dict:
    """Installs the packed weights of `path` into `model` (already on its GPU) without folding anything: one read, one
    host-to-device copy.  The model must have the file's geometry; expect_digest (state_digest of the checkpoint) makes a
    stale file an error instead of a wrong answer.  Returns the header."""
    from . import engine, hip
    hip._require_gpu()
    header, _ = read_header(path)
    abi = hip.header_constants()["LBM_HIP_ABI_VERSION"]
    if header["abi"] != abi:
        raise PackedFormatError(f"{path}: packed for ABI {header['abi']}, this library is ABI {abi}; re-pack from the checkpoint")
    if expect_digest is not None and header["source_digest"] != expect_digest:
        raise PackedFormatError(f"{path}: packed from another checkpoint (digest {header['source_digest'][:12]}..., expected {expect_digest[:12]}...)")
    meta = header["meta"]
    cfg = model.transformer.cfg
    cfg.validate_for_hip()
    mine = {"input_size": list(model.backbone.size), "n_layers": len(model.transformer.layers), "topk": list(cfg.topk_per_layer),
            "embed_dim": cfg.embed_dim, "num_heads": cfg.num_heads, "num_neighbors": cfg.num_neighbors, "stride": cfg.stride,
            "memory_size": model.transformer.layers[0].collision_time_emb.shape[1] - 1}
    for key, v in mine.items():
        if meta[key] != v:
            raise PackedFormatError(f"{path}: packed for {key} = {meta[key]}, the model has {v}")
    dev = model.backbone.frame_pos_embedding.device
    if dev.type != "cuda":
        raise hip.HipError("load_packed: move the model to its GPU first")
    _, v, arena = read_arena(path, dev)

    def split_weight(name):
        sw = hip.SplitWeight.__new__(hip.SplitWeight)
        sw.planes = v[name]
        sw.shape = tuple(sw.planes.shape[1:])
        return sw

    def conv(prefix, key):
        c = engine._Conv.__new__(engine._Conv)
        c.w, c.b, c.ws = v[prefix + ".w"], v[prefix + ".b"], split_weight(prefix + ".planes")
        c.kh, c.stride, c.pad = meta["convs"][key]
        return c

    bb = model.backbone
    enc = engine.EncoderWeights.__new__(engine.EncoderWeights)
    enc.precision = bb.precision or os.environ.get("LBM_AMD_PRECISION", "split")
    if enc.precision not in engine.PRECISIONS:
        raise ValueError(f"precision must be one of {engine.PRECISIONS}")
    enc.planes = 1 if enc.precision == "f16" else 2
    enc.size, enc.h, enc.w, enc.dev = tuple(bb.size), bb.H_prime, bb.W_prime, dev
    enc.P0 = enc.h * enc.w
    enc.L = enc.P0 + enc.P0 // 4 + enc.P0 // 16 + enc.P0 // 64
    enc.stem = conv("enc.stem", "stem")
    enc.blocks = [(conv(f"enc.blk{i}.c1", f"blk{i}.c1"), conv(f"enc.blk{i}.c2", f"blk{i}.c2"),
                   conv(f"enc.blk{i}.down", f"blk{i}.down") if f"blk{i}.down" in meta["convs"] else None)
                  for i in range(meta["n_blocks"])]
    enc.lvl_w = [v[f"enc.lvl{i}.w"] for i in range(3)]
    enc.lvl_ws = [split_weight(f"enc.lvl{i}.planes") for i in range(3)]
    enc.lvl_b, enc.pos = v["enc.lvl_b"], v["enc.pos"]
    enc._scratch = enc.tokens_split = enc._own_split = enc._split_target = None
    enc._arena = arena

    trk = engine.TrackerWeights.__new__(engine.TrackerWeights)
    n = meta["n_table"]
    trk.cfg = cfg
    trk.tensors = [v.get(f"trk.{i}") for i in range(n)]
    trk.split = [split_weight(f"trk.{i}.planes") if f"trk.{i}.planes" in v else None for i in range(n)]
    trk.memory_size, trk.corr_mix, trk.corr_bias = meta["memory_size"], list(meta["corr_mix"]), list(meta["corr_bias"])
    trk.n, trk.n_layers, trk.device = n, meta["n_layers"], dev
    trk._arena = arena
    trk.build_tables()
    cfg.memory_size = trk.memory_size
    bb._packed = enc
    model.transformer._engine = engine.TrackEngine(cfg, trk, model.transformer.precision)
    return header
