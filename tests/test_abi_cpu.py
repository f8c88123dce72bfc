"""-m "not gpu": the C-ABI library loads without a GPU and exports every symbol include/mocopci_hip.h
declares; the ctypes signature table matches the header's argument counts.  No compute calls."""
import os
import re

import pytest

from mocopci_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "mocopci_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"(?:int|size_t|const char \*)\s*(mcp_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return out


def test_every_declared_symbol_is_exported_and_bound():
    decl = header_functions()
    assert len(decl) >= 18
    lib = _lib.load()
    for name, nargs in decl.items():
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
        assert len(_lib.SIGNATURES[name]) == nargs, (name, len(_lib.SIGNATURES[name]), nargs)
    assert set(_lib.SIGNATURES) == set(decl)


def test_version_and_error_strings():
    lib = _lib.load()
    assert lib.mcp_abi_version() == 1
    assert lib.mcp_error_string(0) == b"ok"
    assert b"bad argument" in lib.mcp_error_string(10001)


def test_argument_validation_needs_no_gpu():
    lib = _lib.load()
    assert lib.mcp_furthest_point_sampling(0, 0, 0, None, None, None, None) == 10001
    assert lib.mcp_knn(1, 1, 1, 64, 0, None, None, None, None, None) == 10001
    assert lib.mcp_furthest_point_sampling_ws(1, 0, 1, None, None, None, None, 0, None) == 10001


def test_fps_workspace_query():
    """Only the tiled kernel's range (16384 < N <= 65536) uses scratch: 20 bytes per point of the padded cloud."""
    lib = _lib.load()
    assert lib.mcp_fps_workspace_bytes(8, 8192, 2048) == 0 and lib.mcp_fps_workspace_bytes(8, 16384, 2048) == 0
    assert lib.mcp_fps_workspace_bytes(8, 65536, 2048) == 8 * 65536 * 20
    assert lib.mcp_fps_workspace_bytes(2, 20000, 16) == 2 * 20032 * 20
    assert lib.mcp_fps_workspace_bytes(1, 65537, 16) == 0


def test_library_device_code_has_no_packed_fp32_instructions(tmp_path):
    """The device code that ships must not contain v_pk_*_f32: on MI355X kernels using them returned wrong values while an
    MFMA-heavy kernel of another stream ran (mocopci_amd/csrc/common.h: mcp_f2; tools/fps_under_load.py).  The library is built
    with -fno-slp-vectorize -fno-vectorize and uses scalar pairs; this disassembles every gfx950 code object in the .so."""
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]
    if not all(os.path.exists(t) for t in tools):
        pytest.skip("ROCm LLVM tools not installed")
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mocopci_amd", "libmocopci_hip.so")
    fat = tmp_path / "fat.bin"
    subprocess.check_call([tools[0], f"--dump-section=.hip_fatbin={fat}", so, str(tmp_path / "discard.so")])
    blob = fat.read_bytes()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [i for i in range(len(blob)) if blob.startswith(magic, i)]
    assert len(starts) >= 10, "one bundle per translation unit expected"
    packed = mfma = 0
    for k, (a, b) in enumerate(zip(starts, starts[1:] + [len(blob)])):
        piece, co = tmp_path / f"b{k}.bin", tmp_path / f"b{k}.co"
        piece.write_bytes(blob[a:b])
        subprocess.check_call([tools[1], "--unbundle", "--type=o", f"--input={piece}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        asm = subprocess.run([tools[2], "-d", str(co)], capture_output=True, text=True, check=True).stdout
        packed += sum(1 for line in asm.splitlines() if "v_pk_" in line and "_f32" in line)
        mfma += asm.count("v_mfma")
    assert mfma > 1000, "disassembly did not see the kernels"
    assert packed == 0, f"{packed} packed-fp32 instructions in the shipped device code"
