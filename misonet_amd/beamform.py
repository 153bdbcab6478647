"""Host-side mirror of the reference's array-processing calls on the inference path.

``Apply_Beamforming(source_stft, mix_stft, epsi)``  -- reference tester.py:1071-1136 (MVDR per frequency bin)
``pit_select(anchor, cand)``                        -- reference tester.py:1043-1065 / 889-915 (PIT over all S! permutations)

Both run as HIP kernels through the C ABI (misonet_mvdr / misonet_pit_select in include/misonet.h).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def _dev_c64(x, device):
    was_numpy = not isinstance(x, torch.Tensor)
    t = torch.as_tensor(x)
    if not t.is_complex():
        raise TypeError("expected a complex array")
    if t.device.type != "cuda":
        t = t.to(device)
    return t.to(torch.complex64).contiguous(), was_numpy


def Apply_Beamforming(source_stft, mix_stft, epsi=1e-6, device=None, return_debug=False):
    """MVDR beamforming, same arguments and result as Tester_Enhance.Apply_Beamforming (tester.py:1071-1136).

    source_stft, mix_stft: complex [B, F, Ch, T] (np.ndarray as in the reference, or torch tensors; permuted
    views are fine).  Returns a torch complex64 tensor [B, T, F]: on the CPU when the inputs were ndarrays (the
    reference returns torch.from_numpy(...)), on the device when they were device tensors.  Inputs are not modified.
    """
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    src, np_in = _dev_c64(source_stft, device)
    mix, _ = _dev_c64(mix_stft, device)
    if src.shape != mix.shape or src.dim() != 4:
        raise ValueError(f"source_stft {tuple(src.shape)} and mix_stft {tuple(mix.shape)} must both be [B, F, Ch, T]")
    B, F, M, T = src.shape
    L = _lib.lib()
    nws = L.misonet_mvdr_workspace_bytes(B, F, M)
    ws = torch.empty(max(int(nws), 8), dtype=torch.uint8, device=src.device)
    out = torch.empty((B, T, F), dtype=torch.complex64, device=src.device)
    with torch.cuda.device(src.device):
        st = _lib.stream_ptr(src.device)
        _lib.check(L.misonet_mvdr(src.data_ptr(), mix.data_ptr(), B, F, M, T, float(epsi), out.data_ptr(), ws.data_ptr(),
                                  ws.numel(), st))
        dbg = None
        if return_debug:
            steer = torch.empty((B, F, M), dtype=torch.complex128, device=src.device)
            w = torch.empty((B, F, M), dtype=torch.complex128, device=src.device)
            _lib.check(L.misonet_mvdr_debug(ws.data_ptr(), B, F, M, steer.data_ptr(), w.data_ptr(), st))
            dbg = dict(steer1=steer, w=w)
    if np_in:
        out = out.cpu()
    return (out, dbg) if return_debug else out


def pit_select(anchor, cand, return_dist=False):
    """Speaker alignment by the reference's PIT rule (tester.py:1053-1065, 902-915): all S! permutations in
    itertools.permutations order, first minimum of the summed magnitude distance (1 <= S <= 4).

    anchor, cand: complex [B, S, T, F] device tensors.  Returns int32 [B, S] ``sel`` with aligned speaker i =
    cand[:, sel[:, i]] (and the distance matrix float64 [B, S, S] if asked)."""
    a, _ = _dev_c64(anchor, None if isinstance(anchor, torch.Tensor) and anchor.is_cuda else torch.device("cuda"))
    c, _ = _dev_c64(cand, a.device)
    if a.shape != c.shape or a.dim() != 4:
        raise ValueError("anchor and cand must both be [B, S, T, F]")
    B, S, T, F = a.shape
    sel = torch.empty((B, S), dtype=torch.int32, device=a.device)
    L = _lib.lib()
    nd = L.misonet_pit_scratch_bytes(B, S, F) // 8                                    # result [B,S,S] + per-bin partials
    dist = torch.empty((nd // (S * S), S, S), dtype=torch.float64, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(L.misonet_pit_select(a.data_ptr(), c.data_ptr(), B, S, T, F, sel.data_ptr(), dist.data_ptr(),
                                        dist.numel() * 8, _lib.stream_ptr(a.device)))
    return (sel, dist[:B]) if return_dist else sel
