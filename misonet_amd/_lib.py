"""ctypes binding of libmisonet_hip.so (C ABI: include/misonet.h).

There is no CPU fallback: if the shared library is missing or does not load, every compute entry point raises.
Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C misonet_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MISONET_LIB_PATH: A/B measurements of two builds on one box (tools/gpu_ab_lib.sh); normal use never sets it
LIB_PATH = os.environ.get("MISONET_LIB_PATH") or os.path.join(_HERE, "libmisonet_hip.so")

OK, EINVAL, ESTATE, EHIP, ENOMEM, ENAN = 0, -1, -2, -3, -4, -5


class MisonetError(RuntimeError):
    def __init__(self, code, detail):
        super().__init__(f"libmisonet_hip error {code}: {detail}")
        self.code = code


class Cfg(C.Structure):
    _fields_ = [("in_ch", C.c_int), ("out_ch", C.c_int), ("en_ch", C.c_int * 7), ("de_ch", C.c_int * 7),
                ("n_freq", C.c_int), ("tcn_norm", C.c_int)]       # tcn_norm: 0 IN, 1 gLN, 2 cLN, 3 BatchNorm1d (ABI 400)


# name -> (restype, argtypes); every symbol declared in include/misonet.h
SIGNATURES = {
    "misonet_strerror": (C.c_char_p, [C.c_int]),
    "misonet_last_error": (C.c_char_p, []),
    "misonet_version": (C.c_int, []),
    "misonet_net_create": (C.c_int, [C.POINTER(Cfg), C.POINTER(C.c_void_p)]),
    "misonet_net_destroy": (C.c_int, [C.c_void_p]),
    "misonet_net_num_tensors": (C.c_int, [C.c_void_p]),
    "misonet_net_tensor_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "misonet_net_tensor_numel": (C.c_longlong, [C.c_void_p, C.c_int]),
    "misonet_net_set_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_longlong]),
    "misonet_net_commit": (C.c_int, [C.c_void_p]),
    "misonet_net_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "misonet_net_get_precision": (C.c_int, [C.c_void_p]),
    "misonet_net_keep_activations": (C.c_int, [C.c_void_p, C.c_int]),
    "misonet_net_buffer_plan": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong),
                                          C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "misonet_net_workspace_bytes": (C.c_longlong, [C.c_void_p, C.c_int, C.c_int]),
    "misonet_net_forward": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    "misonet_net_check": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "misonet_net_tap_shape": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "misonet_net_tap": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "misonet_mvdr_workspace_bytes": (C.c_longlong, [C.c_int, C.c_int, C.c_int]),
    "misonet_mvdr": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                               C.c_void_p, C.c_longlong, C.c_void_p]),
    "misonet_mvdr_debug": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "misonet_pit_scratch_bytes": (C.c_longlong, [C.c_int, C.c_int, C.c_int]),
    "misonet_pit_select": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_longlong, C.c_void_p]),
    "misonet_pipeline_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                          C.POINTER(C.c_void_p)]),
    "misonet_pipeline_destroy": (C.c_int, [C.c_void_p]),
    "misonet_pipeline_workspace_bytes": (C.c_longlong, [C.c_void_p, C.c_int, C.c_int]),
    "misonet_pipeline_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    "misonet_pipeline_run_wav": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    "misonet_stft_frames": (C.c_int, [C.c_int]),
    "misonet_stft_workspace_bytes": (C.c_longlong, [C.c_int, C.c_int, C.c_int]),
    "misonet_stft": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    "misonet_frontend_init": (C.c_int, []),
    "misonet_istft": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "misonet_pipeline_check": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "misonet_profile_begin": (C.c_int, [C.c_int]),
    "misonet_profile_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    "misonet_event_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "misonet_event_record": (C.c_int, [C.c_void_p, C.c_void_p]),
    "misonet_event_elapsed_ms": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    "misonet_event_destroy": (C.c_int, [C.c_void_p]),
}

_lib = None


def lib():
    """The loaded library; raises ImportError (loudly) when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build the HIP extension first (make -C misonet_amd/csrc). "
                              "misonet_amd has no CPU fallback.")
        # torch first: it ships its own libamdhip64; loading this library BEFORE torch makes the process hold two HIP runtimes
        # (ours resolved against /opt/rocm), and the second one to initialise sees "no ROCm-capable device" (found by running
        # __graft_entry__.build() and smoke() in one process)
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(code):
    if code != OK:
        L = lib()
        detail = L.misonet_last_error().decode() or L.misonet_strerror(code).decode()
        if code == ENAN:
            raise FloatingPointError(f"libmisonet_hip: {detail}")
        raise MisonetError(code, detail)


def stream_ptr(device=None):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
