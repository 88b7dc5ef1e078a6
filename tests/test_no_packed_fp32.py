"""The product library must not contain packed-fp32 instructions outside the reproducer.

Beside a second HIP queue, code in which hipcc had paired fp32 work into v_pk_mov_b32 / v_pk_*_f32 occasionally computed a zero
in one quarter-wave; the library is therefore built with `-target-feature -packed-fp32-ops` (lbm_amd/csrc/Makefile).  That
fault is a rate-type one -- a run-time screen can miss it -- so correctness rests on THIS deterministic check of the
instructions that ship: the gfx950 code objects inside lbm_amd/liblbm_hip.so are disassembled with llvm-objdump and no kernel
other than lbm::cb_pk::* (corr_block_pk.hip, compiled WITH them on purpose, reached only through debug key 9 = 1..4) may hold one.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
LIB = os.path.join(ROOT, "lbm_amd", "liblbm_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain not present")
def test_no_packed_fp32_instruction_outside_the_reproducer():
    import check_no_packed_fp32 as chk
    assert os.path.exists(LIB), "build the library first (python -c 'import __graft_entry__ as g; g.build()')"
    bad, n_sym, n_allowed = chk.scan(LIB)
    assert n_sym >= 100, f"only {n_sym} symbols disassembled: the extraction is not seeing the kernels"
    assert not bad, f"packed-fp32 instructions in {sorted(bad)[:5]} ({sum(bad.values())} in all): the -packed-fp32-ops flag was lost"
    assert n_allowed > 0, "the reproducer (corr_block_pk.hip) no longer holds packed instructions: the check may be matching nothing"


def test_the_pattern_matches_what_it_should():
    import check_no_packed_fp32 as chk
    for line in ("\tv_pk_mov_b32 v[84:85], v[66:67], v[68:69] op_sel:[1,0]", "  v_pk_fma_f32 v[6:7], v[60:61], s[54:55], -0.5",
                 "v_pk_mul_f32 v[0:1], v[2:3], v[4:5]", "v_pk_add_f32 v[60:61], v[0:1], 0.5"):
        assert chk.PACKED.search(line)
    for line in ("v_pk_add_f16 v0, v1, v2", "v_mov_b64_e32 v[66:67], v[70:71]", "v_pk_fma_f16 v1, v2, v3, v4", "v_mfma_f32_32x32x16_f16"):
        assert not chk.PACKED.search(line)
    assert chk.ALLOWED_SYMBOL.search("_ZN3lbm5cb_pk22corr_maps_block_kernelILi1EEEvNS0_11CorrMapArgsE")
    assert not chk.ALLOWED_SYMBOL.search("_ZN3lbm22corr_maps_block_kernelILi0EEEvNS_11CorrMapArgsE")
