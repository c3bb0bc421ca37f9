"""ctypes binding of libc3prop.so (the C ABI declared in include/c3prop.h).

The library is the product: there is no CPU fallback.  If the shared object is
missing or a call is made without a visible GPU, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# (C3P_LIB: an alternative build of the library -- A/B timing of kernel variants, tools/ab_build*.sh; never set in production)
LIB_PATH = os.path.abspath(os.environ["C3P_LIB"]) if os.environ.get("C3P_LIB") else os.path.join(_HERE, "libc3prop.so")

# flags (mirror include/c3prop.h)
HOST_PTRS = 0x1
PER_SLICE_H = 0x2
ORDER_RIGHT = 0x4
FORCE_GENERIC = 0x8
HERMITIAN_H = 0x10  # c3p_pwc_lindblad: the caller declares h0 / hks Hermitian (D = 2, 3: real arithmetic in the Hermitian basis)

KERNEL_NAMES = {0: "none", 1: "generic_lds", 2: "generic_global", 3: "smalld", 4: "mfma", 5: "ode_wg", 6: "ode_row", 7: "ode_mfma", 8: "ode_row_or_wg"}

SOLVERS = {"rk4": 0, "rk38": 1, "rk5": 2, "tsit5": 3}
STEPS = {"schrodinger": 0, "von_neumann": 1, "lindblad": 2}

_vp = C.c_void_p
_i = C.c_int
_i64 = C.c_int64
_d = C.c_double

# name -> (restype, argtypes); every symbol include/c3prop.h declares
SIGNATURES = {
    "c3p_version": (_i, []),
    "c3p_device_count": (_i, []),
    "c3p_last_error": (C.c_char_p, []),
    "c3p_last_kernel": (_i, []),
    "c3p_last_kernel_detail": (_i, [C.c_char_p, _i]),
    "c3p_set_profiling": (_i, [_i]),
    "c3p_set_option": (_i, [C.c_char_p, C.c_char_p]),
    "c3p_get_option": (C.c_long, [C.c_char_p]),
    "c3p_reserve": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "c3p_workspace_generation": (C.c_long, []),
    "c3p_last_kernel_ms": (_d, []),
    "c3p_shutdown": (None, []),
    "c3p_pwc_unitary": (_i, [_vp, _i64, _vp, _i64, _vp, _d, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "c3p_pwc_lindblad": (_i, [_vp, _i64, _vp, _i64, _vp, _vp, _i, _d, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "c3p_expm": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "c3p_matmul_chain": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "c3p_superop": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "c3p_kron": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "c3p_rk4_unitary": (_i, [_vp, _vp, _vp, _vp, _i64, _d, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "c3p_gate_overlap": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _vp, _vp]),
    "c3p_gate_infid": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "c3p_synth_signals_vjp": (_i, [_vp, _vp, _vp, _d, _d, _d, _d, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "c3p_pwc_unitary_vjp": (_i, [_vp, _i64, _vp, _i64, _vp, _d, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "c3p_pwc_unitary_goal_vjp": (_i, [_vp, _i64, _vp, _i64, _vp, _d, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "c3p_pwc_lindblad_vjp": (_i, [_vp, _i64, _vp, _i64, _vp, _vp, _i, _d, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "c3p_pwc_lindblad_tape_bytes": (C.c_size_t, [_i, _i, _i, _i, C.POINTER(C.c_int)]),
    "c3p_pwc_lindblad_taped": (_i, [_vp, _i64, _vp, _i64, _vp, _vp, _i, _d, _i, _i, _i, _i, _i, _vp, _vp, _vp, C.c_size_t, _i, _vp]),
    "c3p_pwc_lindblad_vjp_taped": (_i, [_vp, C.c_size_t, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "c3p_synth_signals": (_i, [_vp, _vp, _vp, _d, _d, _d, _d, _i, _i, _i, _i, _vp, _vp, _vp]),
    "c3p_ode_solve": (_i, [_vp, _vp, _vp, _vp, _i, _d, _i, _i, _i, _i, _i, _i, _vp, _i64, _i, _i, _vp, _vp]),
}

_lib: Optional[C.CDLL] = None


class C3PropError(Exception):
    """Raised with the reference's `C3:Error` prefix (c3/experiment.py:465-468)."""


def load() -> C.CDLL:
    """Load libc3prop.so and bind every declared symbol.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise C3PropError(
            f"C3:Error: {LIB_PATH} is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the propagator path."
        )
    # PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Import torch
    # first so that the process holds ONE HIP runtime and torch device pointers / streams
    # are valid inside libc3prop.so.
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().c3p_last_error().decode("utf-8", "replace")
        raise C3PropError(f"C3:Error: {msg}")


def require_gpu() -> None:
    lib = load()
    if lib.c3p_device_count() <= 0:
        raise C3PropError("C3:Error: no HIP device visible; the propagator path has no CPU fallback.")


def last_kernel() -> str:
    return KERNEL_NAMES.get(load().c3p_last_kernel(), "?")


def last_kernel_detail() -> str:
    """"file: kernel<...> xN; ..." -- every kernel the last compute call of this thread launched (c3p_last_kernel_detail)."""
    lib = load()
    n = lib.c3p_last_kernel_detail(None, 0)
    buf = C.create_string_buffer(n + 1)
    lib.c3p_last_kernel_detail(buf, n + 1)
    return buf.value.decode()


def set_option(name: str, value) -> None:
    """c3p_set_option: `value` an int, "all", or None (back to the default)."""
    check(load().c3p_set_option(name.encode(), None if value is None else str(value).encode()))


def get_option(name: str) -> int:
    return int(load().c3p_get_option(name.encode()))


class options:
    """Context manager for A/B switches of the library's option table (include/c3prop.h: c3p_set_option):

        with _lib.options(no_regd=1): ...

    restores the previous values on exit."""

    def __init__(self, **kw):
        self.kw = kw
        self.old = {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.old[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, None if v < 0 else v)
        return False
