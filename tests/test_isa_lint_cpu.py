"""-m "not gpu": the hazard lint (tools/isa_lint.py) over the device assembly the library is built from.  Inline asm is invisible
to the compiler's hazard recogniser and wait-count pass, so the few raw instructions the kernels keep are checked in the final
instruction stream; packed-fp32 instructions must not appear at all."""
import glob
import os

import pytest
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint  # noqa: E402


def run(tmp_path, body):
    p = tmp_path / "k.s"
    p.write_text("kern:\n" + body + "\ts_endpgm\n")
    return isa_lint.lint(str(p))[0]


def test_lint_flags_the_hazards_it_is_there_for(tmp_path):
    # an MFMA result read by a raw max with too few wait states (12 needed for the 8-pass bf16 tile on gfx950)
    f = run(tmp_path, "\tv_mfma_f32_32x32x16_bf16 v[0:15], v[20:23], v[24:27], v[0:15]\n\ts_nop 7\n\tv_max_f32 v0, v0, v8\n")
    assert len(f) == 1 and "needs 12" in f[0]
    assert not run(tmp_path, "\tv_mfma_f32_32x32x16_bf16 v[0:15], v[20:23], v[24:27], v[0:15]\n\ts_nop 11\n\tv_max_f32 v0, v0, v8\n")
    # the f32-input tile: 16 passes + 2
    f = run(tmp_path, "\tv_mfma_f32_32x32x2_f32 v[0:15], v20, v24, v[0:15]\n\ts_nop 15\n\tv_min_f32_e32 v0, v0, v8\n")
    assert len(f) == 1 and "needs 18" in f[0]
    # transcendental -> raw VALU
    assert len(run(tmp_path, "\tv_exp_f32_e32 v1, v2\n\tv_max_f32_e32 v3, v1, v1\n")) == 1
    assert not run(tmp_path, "\tv_exp_f32_e32 v1, v2\n\ts_nop 0\n\tv_max_f32_e32 v3, v1, v1\n")
    # the hazard is found along a loop back edge too
    loop = (".LBB0_1:\n\tv_max_f32_e32 v30, v1, v1\n\tv_add_f32_e32 v31, v30, v30\n"
            "\tv_mfma_f32_32x32x16_bf16 v[0:15], v[20:23], v[24:27], v[0:15]\n\ts_nop 3\n\ts_cbranch_scc1 .LBB0_1\n")
    assert len(run(tmp_path, loop)) == 1
    # raw result -> v_readlane / permlane swap
    assert len(run(tmp_path, "\tv_max_f32 v1, v2, v3\n\tv_readlane_b32 s0, v1, 63\n")) == 1
    assert len(run(tmp_path, "\tv_max_f32 v1, v2, v3\n\ts_nop 0\n\tv_permlane16_swap_b32_e32 v1, v5\n")) == 1
    # inline-asm DPP without its own wait states
    f = run(tmp_path, "\tv_add_f32_e32 v1, v2, v3\n\t;;#ASMSTART\n\tv_max_f32_dpp v4, v1, v1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t;;#ASMEND\n")
    assert len(f) == 1 and "needs 2" in f[0]
    # LDS atomic before a barrier without a wait
    assert len(run(tmp_path, "\tds_max_u64 v2, v[20:21]\n\ts_barrier\n")) == 1
    assert not run(tmp_path, "\tds_max_u64 v2, v[20:21]\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n")
    # packed fp32 is refused outright
    assert len(run(tmp_path, "\tv_pk_add_f32 v[0:1], v[0:1], v[2:3]\n")) == 1


def test_shipped_device_code_passes_the_lint():
    import shutil
    if not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")):
        pytest.skip("no hipcc: the device assembly cannot be generated on this machine")
    isa = os.path.join(ROOT, "mocopci_amd", "csrc", "isa")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "mocopci_amd", "csrc"), "-j4", "-s", "isa"])
    files = sorted(glob.glob(os.path.join(isa, "*.s")))
    assert len(files) >= 15
    total, bad = 0, []
    for p in files:
        f, n = isa_lint.lint(p)
        total += n
        bad += f
    assert total > 100000, "the assembly was not parsed"
    assert not bad, "\n".join(bad[:20])
    # no inline-asm LDS instruction is left: the sampling loops use the builtin atomic, which the wait-count pass tracks
    for p in files:
        in_asm = False
        for line in open(p):
            if "#ASMSTART" in line:
                in_asm = True
            elif "#ASMEND" in line:
                in_asm = False
            elif in_asm:
                assert not line.strip().startswith(("ds_", "global_", "buffer_", "flat_")), (p, line)
