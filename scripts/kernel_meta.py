"""Registers, LDS and spills of every kernel in a built object / library (from the code object's metadata notes).
usage: kernel_meta.py <file.o | lib.so> [name substring]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
src = os.path.abspath(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
tmp = tempfile.mkdtemp()
try:
    local = os.path.join(tmp, os.path.basename(src))
    shutil.copy(src, local)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, cwd=tmp, stdout=subprocess.DEVNULL)
    for f in sorted(os.listdir(tmp)):
        if "gfx950" not in f:
            continue
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], check=True, stdout=subprocess.PIPE, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
            name = subprocess.run(["c++filt", g("name")], stdout=subprocess.PIPE, text=True).stdout.strip()
            if pat in name:
                agpr = re.match(r"\s*(\d+)", blk).group(1)
                print(f"vgpr {g('vgpr_count'):>4} agpr {agpr:>3} sgpr {g('sgpr_count'):>3} lds {g('group_segment_fixed_size'):>6} scratch {g('private_segment_fixed_size'):>4} "
                      f"spill v{g('vgpr_spill_count')} s{g('sgpr_spill_count')}  {name[:150]}")
finally:
    shutil.rmtree(tmp)
