"""The C-ABI library loads and exports every function the header declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mistitch.h")
LIB = os.path.join(ROOT, "image-stitcher_amd", "libmistitch.so")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mistitch_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_path():
    names = declared_functions()
    for need in ("mistitch_register_pairs", "mistitch_pcm", "mistitch_pcm_topk", "mistitch_ncc",
                 "mistitch_edge_variance", "mistitch_apply_flatfield", "mistitch_fuse", "mistitch_mip"):
        assert need in names


def test_library_exports_every_declared_symbol():
    if not os.path.exists(LIB):
        pytest.fail(f"{LIB} is not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    import torch  # noqa: F401  (the library links the HIP runtime torch already maps)

    lib = ctypes.CDLL(LIB)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing
    lib.mistitch_version.restype = ctypes.c_int
    assert lib.mistitch_version() >= 1


def test_binding_lists_every_declared_symbol():
    from image_stitcher_amd import _lib

    assert sorted(_lib.EXPORTS) == declared_functions()


def test_calls_without_a_context_fail_cleanly():
    """NULL context -> MISTITCH_ERR_ARG with a message, never a crash (host-side check only)."""
    from image_stitcher_amd import _lib

    rc = _lib.lib.mistitch_set_pair_chunk(None, 4)
    assert rc == _lib.ERR_ARG
    assert _lib.lib.mistitch_last_error()


def test_context_needs_a_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from image_stitcher_amd import _lib

    with pytest.raises(RuntimeError):
        _lib.Context()
