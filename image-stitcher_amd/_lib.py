"""ctypes binding of libmistitch.so (see include/mistitch.h).

There is no CPU fallback: importing this module without the built library, or
creating a Context without a GPU, raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

# torch ships its own HIP runtime; it must be the one already mapped when
# libmistitch.so is loaded, or the process ends up with two runtimes and the
# second one finds no device.
import torch  # noqa: F401  (import order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MISTITCH_LIB") or os.path.join(_HERE, "libmistitch.so")  # override: kernel experiments

OK = 0
ERR_ARG, ERR_UNSUPPORTED, ERR_HIP, ERR_NOMEM = -1, -2, -3, -4
PAIR_OK, PAIR_FEATURELESS, PAIR_NO_CANDIDATE = 0, 1, 2
FFT_F64, FFT_F32 = 0, 1
REG_SUBPIXEL = 2
REG_PEAKS10 = 4
REG_WINDOW_HANN = 8
REG_CONSENSUS = 16
FUSE_OVERWRITE, FUSE_FEATHER = 0, 1
FF_NONE, FF_F32, FF_F64 = 0, 1, 2

EXPORTS = [
    "mistitch_version", "mistitch_last_error", "mistitch_create", "mistitch_destroy",
    "mistitch_set_pair_chunk", "mistitch_set_staging_limit", "mistitch_reserve_registration", "mistitch_register_pairs",
    "mistitch_pcm", "mistitch_pcm_topk", "mistitch_ncc", "mistitch_edge_variance",
    "mistitch_apply_flatfield", "mistitch_fuse", "mistitch_mip", "mistitch_synchronize",
    "mistitch_profile_enable", "mistitch_profile_collect", "mistitch_subpixel_shift",
    "mistitch_set_upsample_factor", "mistitch_pyramid", "mistitch_chunk_layout", "mistitch_refine_pairs",
    "mistitch_register_pairs_flatfield", "mistitch_fuse_static", "mistitch_register_pairs_candidates",
    "mistitch_bitshuffle", "mistitch_fuse_chunked", "mistitch_pyramid_chunked", "mistitch_set_pixel_max",
    "mistitch_pyramid_to_chunks",
]


class MistitchError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libmistitch error {code}: {message}")
        self.code = code


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C image-stitcher_amd/csrc` "
            "(or __graft_entry__.build()); this package has no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, u32 = C.c_void_p, C.c_int, C.c_int64, C.c_uint
    lib.mistitch_version.restype = i32
    lib.mistitch_last_error.restype = C.c_char_p
    sigs = {
        "mistitch_create": [i32, vp, C.POINTER(vp)],
        "mistitch_destroy": [vp],
        "mistitch_set_pair_chunk": [vp, i32],
        "mistitch_set_staging_limit": [vp, i64],
        "mistitch_reserve_registration": [vp, i32, i32, i32, i32, u32],
        "mistitch_register_pairs": [vp, vp, i32, i32, i32, i64, vp, vp, i32, i32, u32, vp, vp],
        "mistitch_register_pairs_flatfield": [vp, vp, i32, i32, i32, i64, vp, i32, vp, vp, i32, i32, u32, vp, vp],
        "mistitch_register_pairs_candidates": [vp, vp, i32, i32, i32, i64, vp, i32, vp, vp, i32, i32, u32, vp, vp, vp],
        "mistitch_refine_pairs": [vp, vp, i32, i32, i32, i64, vp, vp, i32, i32, vp, C.c_double, vp, vp],
        "mistitch_pcm": [vp, vp, i64, vp, i64, i32, i32, u32, vp],
        "mistitch_pcm_topk": [vp, vp, i64, vp, i64, i32, i32, u32, i32, vp, vp, vp],
        "mistitch_ncc": [vp, vp, i64, vp, i64, i32, i32, i32, vp],
        "mistitch_subpixel_shift": [vp, vp, i64, vp, i64, i32, i32, u32, i32, vp],
        "mistitch_set_upsample_factor": [vp, i32],
        "mistitch_set_pixel_max": [vp, i32],
        "mistitch_edge_variance": [vp, vp, i64, i32, i32, vp],
        "mistitch_apply_flatfield": [vp, vp, vp, i32, vp, i32, i32, i32],
        "mistitch_fuse": [vp, vp, i32, i32, i32, i64, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32],
        "mistitch_fuse_static": [vp, vp, i32, i32, i32, i64, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32],
        "mistitch_fuse_chunked": [vp, vp, i32, i32, i32, i64, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32],
        "mistitch_mip": [vp, vp, i32, i64, i64, vp],
        "mistitch_pyramid": [vp, vp, i32, i32, i32, i32, C.POINTER(vp)],
        "mistitch_pyramid_chunked": [vp, vp, i32, i32, i32, i32, i32, i32, C.POINTER(vp)],
        "mistitch_pyramid_to_chunks": [vp, vp, i32, i32, i32, i32, i32, i32, i32, C.POINTER(vp)],
        "mistitch_chunk_layout": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
        "mistitch_bitshuffle": [vp, vp, vp, i64, i32, i64],
        "mistitch_synchronize": [vp],
        "mistitch_profile_enable": [vp, i32],
        "mistitch_profile_collect": [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(i32), i32,
                                     C.POINTER(i32)],
    }
    for name, argtypes in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = i32
    return lib


lib = _load()


def check(code: int) -> None:
    if code != OK:
        raise MistitchError(code, lib.mistitch_last_error().decode("utf-8", "replace"))


class Context:
    """One library context bound to a device and a stream (torch's current ones by default)."""

    def __init__(self, device: Optional[int] = None, stream: Optional[int] = None, pair_chunk: Optional[int] = None):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("libmistitch needs a ROCm GPU: torch.cuda.is_available() is False")
        if device is None:
            device = torch.cuda.current_device()
        if stream is None:
            stream = torch.cuda.current_stream(device).cuda_stream
        self.device = int(device)
        self.stream = int(stream)
        handle = C.c_void_p()
        check(lib.mistitch_create(self.device, C.c_void_p(self.stream), C.byref(handle)))
        self._h = handle
        if pair_chunk is not None:
            self.set_pair_chunk(pair_chunk)

    @property
    def handle(self) -> C.c_void_p:
        if self._h is None:
            raise RuntimeError("context already destroyed")
        return self._h

    def set_pair_chunk(self, n: int) -> None:
        check(lib.mistitch_set_pair_chunk(self.handle, int(n)))

    def set_staging_limit(self, n_bytes: int) -> None:
        check(lib.mistitch_set_staging_limit(self.handle, int(n_bytes)))

    def set_upsample_factor(self, u: int) -> None:
        check(lib.mistitch_set_upsample_factor(self.handle, int(u)))

    def pixel_max(self, value: int):
        """``with ctx.pixel_max(255): ...``: calls inside clip flatfield-corrected pixels to ``value`` (8-bit tiles);
        back to 65535 afterwards."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            check(lib.mistitch_set_pixel_max(self.handle, int(value)))
            try:
                yield self
            finally:
                check(lib.mistitch_set_pixel_max(self.handle, 65535))
        return scope()

    def synchronize(self) -> None:
        check(lib.mistitch_synchronize(self.handle))

    def profile(self, on: bool) -> None:
        """Start (or stop) recording HIP events around every kernel this context launches."""
        check(lib.mistitch_profile_enable(self.handle, 1 if on else 0))

    def profile_collect(self):
        """{kernel name: (total milliseconds, launches)} since profiling was enabled / last collected."""
        cap = 64
        names = C.create_string_buffer(4096)
        ms = (C.c_double * cap)()
        cnt = (C.c_int * cap)()
        n = C.c_int(0)
        check(lib.mistitch_profile_collect(self.handle, names, 4096, ms, cnt, cap, C.byref(n)))
        keys = names.value.decode().split("\n") if n.value else []
        return {k: (ms[i], cnt[i]) for i, k in enumerate(keys)}

    def close(self) -> None:
        if getattr(self, "_h", None) is not None:
            lib.mistitch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context() -> Context:
    """A cached context for torch's current device and current stream."""
    import torch

    dev = torch.cuda.current_device() if torch.cuda.is_available() else -1
    stream = torch.cuda.current_stream(dev).cuda_stream if dev >= 0 else 0
    key = (dev, stream)
    if key not in _default_ctx:
        _default_ctx[key] = Context(dev if dev >= 0 else None, stream)
    return _default_ctx[key]


def ptr(t) -> C.c_void_p:
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr())


def ptr_of(t) -> C.c_void_p:
    """Device pointer of a (possibly strided) tensor view's first element."""
    return C.c_void_p(t.data_ptr())


def ptr_at(t, i: int) -> C.c_void_p:
    """Device pointer of element ``i`` of a contiguous 1-D tensor."""
    return C.c_void_p(t.data_ptr() + int(i) * t.element_size())
