"""tools/isa_lint.py over the shading kernels' ISA: the two hazards the compiler cannot see inside shade.hip's asm statements (a scalar
load's destination touched before its s_waitcnt; a transcendental's result read by the very next VALU instruction).  CPU-only: hipcc
cross-compiles.  The first test proves the lint sees both on hand-made snippets, the second runs it on the real kernels."""
import importlib.util
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "tools", "isa_lint.py"))
isa_lint = importlib.util.module_from_spec(spec)
spec.loader.exec_module(isa_lint)

HEAD = "_ZN6arctic12_GLOBAL__N_110k_materialILi2ELb0ELb0EEEvNS_11ShadeParamsE:\n"
TAIL = "\t.end_amdhsa_kernel\n"


def _lint(tmp_path, body):
    p = tmp_path / "k.s"
    p.write_text(HEAD + body + TAIL)
    return isa_lint.lint(str(p))


def test_lint_sees_the_hazards(tmp_path):
    clean = ("\t;;#ASMSTART\n\ts_load_dwordx4 s[8:11], s[2:3], 0x0\n\t;;#ASMEND\n\tv_mov_b32_e32 v1, v2\n\ts_waitcnt lgkmcnt(0)\n"
             "\tv_pk_add_f32 v[4:5], s[8:9], v[6:7]\n\tv_rsq_f32_e32 v8, v9\n\tv_mul_f32_e32 v10, v11, v12\n\tv_mul_f32_e32 v13, v8, v8\n")
    assert _lint(tmp_path, clean) == []
    early = clean.replace("\tv_mov_b32_e32 v1, v2\n", "\tv_pk_add_f32 v[4:5], s[8:9], v[6:7]\n")
    assert len(_lint(tmp_path, early)) == 1 and "before s_waitcnt" in _lint(tmp_path, early)[0]
    hot = clean.replace("\tv_mul_f32_e32 v10, v11, v12\n", "")
    assert len(_lint(tmp_path, hot)) == 1 and "transcendental" in _lint(tmp_path, hot)[0]
    nop = clean.replace("\tv_mul_f32_e32 v10, v11, v12\n", "\ts_nop 0\n")
    assert _lint(tmp_path, nop) == []


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_shading_kernels_are_clean():
    csrc = os.path.join(ROOT, "arctic-renderer_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "asm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    found = isa_lint.lint("/tmp/shade-hip-amdgcn-amd-amdhsa-gfx950.s")
    assert found == [], "\n".join(found)
