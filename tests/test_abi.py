"""CPU: the C-ABI library builds/loads and exports every symbol include/mumpy_hip.h declares, with the arity the
ctypes binding assumes.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

from conftest import PKG, ROOT


def header_decls():
    src = open(os.path.join(ROOT, "include", "mumpy_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(?:int64_t|int|const char\*)\s+(mumpy_\w+)\s*\(([^)]*)\)\s*;", src):
        args = [a.strip() for a in m.group(2).split(",") if a.strip() and a.strip() != "void"]
        decls[m.group(1)] = args
    return decls


def test_header_declares_expected_entry_points():
    d = header_decls()
    for name in ("mumpy_window_attention_fwd", "mumpy_deform_sample_fwd", "mumpy_deform_attention_fwd", "mumpy_faf_fwd",
                 "mumpy_patch_embed_fwd", "mumpy_linear_fwd", "mumpy_layernorm_fwd", "mumpy_last_error"):
        assert name in d


def test_library_exports_every_declared_symbol():
    from mumpy_hip.lib import SIGNATURES, library_path, load_library
    assert os.path.exists(library_path()), "build with `python __graft_entry__.py`"
    lib = load_library()
    decls = header_decls()
    for name, args in decls.items():
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
        if name in SIGNATURES:
            assert len(SIGNATURES[name]) == len(args), f"{name}: binding has {len(SIGNATURES[name])} args, header {len(args)}"
    for name in SIGNATURES:
        assert name in decls, f"{name} bound but not declared in include/mumpy_hip.h"
    from mumpy_hip.lib import ABI_VERSION
    assert lib.mumpy_abi_version() == ABI_VERSION == 2          # bumped when mumpy_layernorm_bwd grew dx_add / accumulate


def test_binding_argument_types_match_the_header():
    """Every bound argument has the ctypes kind of its C declaration (a float/double or int/int64 slip corrupts the call)."""
    from mumpy_hip.lib import SIGNATURES
    decls = header_decls()

    def kind(carg):
        carg = carg.strip()
        if "*" in carg:
            return "ptr"
        base = carg.rsplit(None, 1)[0] if " " in carg else carg
        return {"int": "int", "int64_t": "i64", "float": "f32", "double": "f64"}[base.replace("const ", "").strip()]

    def ckind(t):
        if t is ctypes.c_void_p or (isinstance(t, type) and issubclass(t, ctypes._Pointer)):
            return "ptr"
        return {ctypes.c_int: "int", ctypes.c_int64: "i64", ctypes.c_float: "f32", ctypes.c_double: "f64"}[t]

    for name, args in SIGNATURES.items():
        got = [ckind(t) for t in args]
        want = [kind(a) for a in decls[name]]
        assert got == want, f"{name}: binding {got} vs header {want}"


def test_argument_validation_without_gpu():
    """Rejected arguments return negative codes before anything is launched, so this is safe on a CPU box."""
    from mumpy_hip.lib import load_library
    lib = load_library()
    rc = lib.mumpy_layernorm_fwd(None, None, None, None, 4, 96, 1e-5, None)
    assert rc == -3 and b"null" in lib.mumpy_last_error()
    rc = lib.mumpy_linear_fwd(16, 16, None, None, 16, 4, 100, 96, 0, None)        # N % 32 != 0
    assert rc == -1
    rc = lib.mumpy_window_attention_fwd(16, 16, 16, None, None, 0, 1, 10, 14, 96, 0, 0.1, None)   # grid not /7
    assert rc == -1
    rc = lib.mumpy_temporal_attention_fwd(16, 16, 4, 17, 768, 12, 0.125, None)    # T > 16
    assert rc == -4


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import mumpy_hip.lib as L
    monkeypatch.setattr(L, "_LIB", None)
    monkeypatch.setenv("MUMPY_HIP_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU/torch fallback"):
        L.load_library()


@pytest.mark.parametrize("src,dst", [(432, 224), (240, 224), (1920, 224), (1080, 224), (224, 224), (37, 224), (854, 448)])
def test_resize_nearest_table_is_pillows(src, dst):
    """The host helper behind mumpy_resize_normalize_u8_fwd reproduces PIL's NEAREST source indices (incl. the exact ties
    that Pillow's double accumulator decides, e.g. 1920 -> 224): checked against PIL resizing an index ramp."""
    import numpy as np
    from PIL import Image
    from mumpy_hip.lib import load_library
    lib = load_library()
    tab = (ctypes.c_int32 * dst)()
    assert lib.mumpy_resize_nearest_table(src, dst, tab) == 0
    ramp = np.arange(src, dtype=np.int32).reshape(1, src)                     # mode "I": pixel value = source column
    pil = np.array(Image.fromarray(ramp, mode="I").resize((dst, 1), Image.NEAREST)).reshape(-1)
    assert list(tab) == pil.tolist()


def test_shipped_library_reads_no_environment_and_the_tuning_build_exists():
    """The shipped libmumpy_hip.so is built without the tuning hooks (mumpy_tuning_build() == 0: no getenv in the library, the
    planner is a pure function of the shape); the diagnostics build that tools/ and the variant tests select with MUMPY_HIP_LIB is
    a separate file with the same exported symbols."""
    import os
    import subprocess
    from mumpy_hip.lib import library_path, load_library, tuning_library_path
    assert load_library().mumpy_tuning_build() == 0
    nm = subprocess.run(["nm", "-D", "--undefined-only", library_path()], capture_output=True, text=True).stdout
    assert "getenv" not in nm
    assert os.path.exists(tuning_library_path()), "run `python __graft_entry__.py` (builds both libraries)"
    t = ctypes.CDLL(tuning_library_path())
    assert t.mumpy_tuning_build() == 1 and t.mumpy_abi_version() == load_library().mumpy_abi_version()
