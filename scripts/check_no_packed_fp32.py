#!/usr/bin/env python3
"""Guard of the packed-fp32 mitigation: disassemble the gfx950 code objects inside a built
liblbm_hip.so (or .o) and fail if a packed-fp32 instruction appears in any kernel other than the reproducer's
(`cb_pk::` = corr_block_pk.hip, which is compiled WITH them on purpose).

The library is built with `-target-feature -packed-fp32-ops`; that flag can be lost silently (another hipcc, a changed
Makefile, hand-written asm), and the fault it avoids is a rate-type one that a run-time screen may miss.  This check is
deterministic: it reads the instructions that ship.

usage: check_no_packed_fp32.py lbm_amd/liblbm_hip.so      (exit status 0 = clean)
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = os.environ.get("LBM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
# the instruction class the compiler flag removes: 64-bit packed fp32 arithmetic and the packed move it pairs loads with
PACKED = re.compile(r"\bv_pk_(mov_b32|mul_f32|fma_f32|add_f32)\b")
ALLOWED_SYMBOL = re.compile(r"cb_pk")            # mangled names of namespace lbm::cb_pk (corr_block_pk.hip)
SYMBOL = re.compile(r"^[0-9a-f]+ <([^>]+)>:")


def code_objects(path: str, tmp: str) -> list[str]:
    """The gfx950 code objects bundled in `path` (a .so or .o), extracted into `tmp`."""
    local = os.path.join(tmp, os.path.basename(path))
    shutil.copy(path, local)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, cwd=tmp,
                   stdout=subprocess.DEVNULL)
    out = sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "amdgcn" in f and os.path.getsize(os.path.join(tmp, f)) > 0)
    if not out:
        raise RuntimeError(f"{path}: no gfx950 code object found")
    return out


def scan(path: str) -> tuple[dict[str, int], int, int]:
    """-> ({offending symbol: count}, kernels seen, packed instructions inside the allowed reproducer)."""
    bad: dict[str, int] = {}
    n_sym = 0
    n_allowed = 0
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(path, tmp):
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co], check=True,
                                 capture_output=True, text=True).stdout
            sym = "?"
            for line in dis.splitlines():
                m = SYMBOL.match(line)
                if m:
                    sym = m.group(1)
                    n_sym += 1
                    continue
                if PACKED.search(line):
                    if ALLOWED_SYMBOL.search(sym):
                        n_allowed += 1
                    else:
                        bad[sym] = bad.get(sym, 0) + 1
    return bad, n_sym, n_allowed


def main(argv: list[str]) -> int:
    if len(argv) != 2:
        print(__doc__)
        return 2
    bad, n_sym, n_allowed = scan(argv[1])
    if bad:
        for sym, n in sorted(bad.items()):
            print(f"packed-fp32 instructions ({n}) in {sym}")
        print(f"FAIL: {len(bad)} of {n_sym} symbols of {argv[1]} contain v_pk_(mov_b32|mul_f32|fma_f32|add_f32)")
        return 1
    print(f"check-nopk: {n_sym} symbols of {os.path.basename(argv[1])} clean "
          f"({n_allowed} packed instructions inside the cb_pk reproducer, as intended)")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
