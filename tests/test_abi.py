"""C-ABI checks that need no GPU: the hipcc-built library loads, exports every symbol that
include/mds.h declares, and the ctypes structs generated from the header have the sizes the C
compiler gives them."""
import ctypes
import os
import subprocess
import tempfile

from mds import cabi
from conftest import ROOT


def test_hip_library_exports_every_declared_symbol():
    csrc = os.path.join(ROOT, "ball-action-spotting_amd", "csrc")
    subprocess.run(["make", "-s", "-j8", "all"], cwd=csrc, check=True)      # hipcc cross-compiles gfx950
    lib = cabi.Lib(cabi.HIP_LIB)
    assert lib.missing == []
    declared = {n for n, _ in cabi.FUNCS}
    assert {"mds_pw_fwd", "mds_conv_fwd", "mds_dw_bwd", "mds_gem_bwd", "mds_pack_weights"} <= declared
    assert len(declared) == 40      # + mds_last_error (returns const char*); grows with include/mds.h, never silently
    assert lib.dll.mds_version() == cabi.MDS_VERSION


def test_every_entry_point_is_typed_from_the_header():
    """an untyped ctypes call passes Python ints as 32-bit C ints: a hipStream_t other than the null stream would be truncated
    (round 3: mds_focal_fwd_bwd / mds_multi_* / mds_aug_pass / mds_frame_luma crashed on a non-default stream)"""
    lib = cabi.Lib(cabi.HIP_LIB)
    for name, params in cabi.FUNCS:
        if name == "mds_version":
            continue
        f = lib.fn[name[4:]]
        assert f.argtypes is not None, name
        assert len(f.argtypes) == (0 if params.strip() == "void" else len(params.split(","))), name
        if "mds_stream_t" in params:
            assert f.argtypes[-1] is ctypes.c_void_p, name


def test_struct_sizes_match_the_c_compiler():
    names = [n for n in cabi.STRUCTS]
    src = '#include <stdio.h>\n#include "mds.h"\nint main(void){\n' + "".join(
        f'  printf("{n} %zu\\n", sizeof({n}));\n' for n in names) + "  return 0; }\n"
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "sz.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "sz")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    sizes = dict(line.split() for line in out.strip().split("\n"))
    for n in names:
        assert ctypes.sizeof(cabi.STRUCTS[n]) == int(sizes[n]), n


def test_errors_are_codes_not_aborts():
    """bad arguments come back as negative codes + message through the simulator build too"""
    from hipemu.loader import load_emulator
    import torch
    lib = load_emulator()
    a = cabi.make("mds_pw_fwd_args", dtype=7, M=4, K=8, N=16, x=torch.zeros(32), w=torch.zeros(128), y=torch.zeros(64),
                  pro=cabi.pro(0), residual=None, stats=None)
    rc = lib.fn["pw_fwd"](ctypes.byref(a), 0)
    assert rc == cabi.MDS_ERR_UNSUPPORTED and b"dtype" in lib.dll.mds_last_error()
    a = cabi.make("mds_pw_fwd_args", dtype=0, M=4, K=7, N=16, x=torch.zeros(32), w=torch.zeros(128), y=torch.zeros(64),
                  pro=cabi.pro(0), residual=None, stats=None)
    assert lib.fn["pw_fwd"](ctypes.byref(a), 0) == cabi.MDS_ERR_BAD_ARG


def test_product_never_touches_the_oracle_or_the_simulator():
    """the oracle and the kernel simulator are test infrastructure: nothing under ball-action-spotting_amd/ may import, open or
    name them (a product path routed through either would void every parity claim)"""
    import re
    pkg = os.path.join(ROOT, "ball-action-spotting_amd")
    bad = []
    for base, _, files in os.walk(pkg):
        if "build" in base.split(os.sep) or "__pycache__" in base:
            continue
        for f in files:
            if not f.endswith((".py", ".hip", ".h", "Makefile")):
                continue
            text = open(os.path.join(base, f), errors="ignore").read()
            for pat in (r"^\s*(from|import)\s+oracle", r"libmds_emu", r"from\s+hipemu|import\s+hipemu", r"#\s*if(n)?def\s+MDS_EMU"):
                if re.search(pat, text, flags=re.M) and not (f == "Makefile" and pat == r"libmds_emu"):
                    bad.append((os.path.relpath(os.path.join(base, f), ROOT), pat))
    assert not bad, bad
