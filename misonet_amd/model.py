"""Host-side mirror of the reference model classes for the inference path.

``MISO_1`` / ``MISO_3`` keep the constructor signature, the module protocol used by the reference harness
(``.cuda(idx)``, ``.eval()``, ``.load_state_dict(sd)``, ``.state_dict()``, ``print(model)``; reference
run.py:121-151) and the ``forward`` call surface (reference model.py:76-111, 350-395; called from
tester.py:1035,1051,1242), but own no torch parameters: the weights live in the HIP library's packed layout and
every forward is a sequence of hand-written gfx950 kernels launched through the C ABI (include/misonet.h).

Differences from the reference, on purpose:
  * the constructor does not mutate the channel lists passed in (model.py:16-17 does);
  * a NaN in the output raises FloatingPointError instead of dropping into pdb (model.py:109-110);
  * inference only (no autograd); norm_type must be "IN" (config/NN_BSS.yml:123).
"""
from __future__ import annotations

import os

import ctypes as C
from collections import OrderedDict
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import _lib
from . import weights as W


class _Trunk:
    _extra_in = 0          # extra complex input channels besides the microphones (MISO_3: BF + MISO1 = 2)

    def __init__(self, num_spks, num_ch, num_bottleneck, en_bottleneck_channels, de_bottleneck_channels, norm_type):
        if num_bottleneck != 7 or len(en_bottleneck_channels) != 7 or len(de_bottleneck_channels) != 7:
            raise ValueError("misonet_amd supports the reference geometry num_bottleneck = 7 (model.py:40-73)")
        if not isinstance(norm_type, str):
            raise TypeError("norm_type must be a string (chose_norm, model.py:570-581)")
        # norm_type selects the two OUTER norms of every TemporalBlock (model.py:530,535): "IN" (the committed config,
        # config/NN_BSS.yml:123), "gLN", "cLN", anything else = BatchNorm1d (eval mode: running statistics)
        self.norm_type = norm_type
        self.num_spks, self.num_ch, self.num_bottleneck = int(num_spks), int(num_ch), 7
        self.en_ch = tuple(int(c) for c in en_bottleneck_channels)
        self.de_ch = tuple(int(c) for c in de_bottleneck_channels)
        self.in_ch = 2 * (self.num_ch + self._extra_in)
        self.out_ch = 2 * self.num_spks
        self.spec = W.tensor_spec(self.in_ch, self.out_ch, self.en_ch, self.de_ch, norm_type)
        self._sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
        self._net = C.c_void_p()
        self._committed = False
        self._device: Optional[torch.device] = None
        self._ws: Dict[tuple, torch.Tensor] = {}
        self.training = True
        self._keep = False
        self.precision = "bf16x6"      # the library's default (misonet_net.precision in csrc/net.hip): fp32-faithful, the bench's mode
        L = _lib.lib()
        cfg = _lib.Cfg(self.in_ch, self.out_ch, (C.c_int * 7)(*self.en_ch), (C.c_int * 7)(*self.de_ch), W.N_FREQ,
                       W.norm_kind(norm_type))
        _lib.check(L.misonet_net_create(C.byref(cfg), C.byref(self._net)))
        names = [L.misonet_net_tensor_name(self._net, i).decode() for i in range(L.misonet_net_num_tensors(self._net))]
        # (BatchNorm1d's int64 counter num_batches_tracked lives in the state_dict only: eval mode does not read it)
        if names != [k for k in self.spec.keys() if not k.endswith(".num_batches_tracked")]:
            raise RuntimeError("library tensor list differs from weights.tensor_spec")
        default_prec = os.environ.get("MISONET_PRECISION")      # e.g. MISONET_PRECISION=bf16x3 for an unmodified harness
        if default_prec:
            self.set_precision(default_prec)

    def __del__(self):
        try:
            if self._net:
                _lib.lib().misonet_net_destroy(self._net)
                self._net = C.c_void_p()
        except Exception:
            pass

    # ---- module protocol (run.py:68,76-79,134-151) --------------------------------------------------------------
    def cuda(self, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("misonet_amd needs a ROCm device (no CPU fallback)")
        if device is None:
            device = torch.cuda.current_device()
        d = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        if d.type != "cuda":
            raise RuntimeError(f"misonet_amd runs on ROCm devices only (got {d})")
        if d.index is None:            # "cuda" -> "cuda:<current>": torch.device("cuda") != torch.device("cuda:0")
            d = torch.device("cuda", torch.cuda.current_device())
        if d != self._device:
            self._committed = False    # weights are uploaded to the device that is current at commit
            self._ws.clear()
        self._device = d
        return self

    def to(self, device):
        return self.cuda(device)

    def eval(self):
        self.training = False
        return self

    PRECISIONS = {"f32": 0, "bf16x3p": 1, "bf16x3": 2, "bf16x6": 3, "f16x3": 4, "f32w": 5, "bf16x6w": 6}
    PRODUCT_PRECISIONS = ("f32", "f32w", "bf16x6")     # the others exist only in the experiment build (csrc: make exp)

    def set_precision(self, mode: str):
        """Arithmetic of the 3x3 convolutions (99.4 % of the FLOPs; the reference computes them in float32, model.py:77-80):
        "f32w" -- float32 matrix cores with the DenseBlock convs (94 % of the MACs) in Winograd F(2x2, 3x3) form: float32
        products and sums throughout, 2.25x fewer matrix instructions (conv_wino.hip); "f32" -- the same matrix cores in the
        direct form (bitwise an fmaf chain, conv.hip); "bf16x6" (the default of a new handle) -- fp32-faithful on the bf16 matrix
        cores: both operands split EXACTLY into three bf16 pieces, the six leading partial products accumulated in float32
        (conv_bf16x6.hip).  The measured alternatives "bf16x3" / "bf16x3p" (16-bit operands), "f16x3" (22-bit operands) and
        "bf16x6w" (Winograd in the bf16x6 arithmetic: correct, slower than "bf16x6") are not product modes: they are compiled
        only into the experiment library (`make exp`, loaded through MISONET_LIB_PATH)."""
        if mode not in self.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(self.PRECISIONS)}")
        rc = _lib.lib().misonet_net_set_precision(self._net, self.PRECISIONS[mode])
        if rc != 0 and mode not in self.PRODUCT_PRECISIONS:
            raise ValueError(f"precision {mode!r} exists only in the experiment build of the library (csrc: `make exp`, "
                             f"MISONET_LIB_PATH=.../libmisonet_hip_exp.so); product modes: {self.PRODUCT_PRECISIONS}")
        _lib.check(rc)
        if mode != self.precision:
            self._ws.clear()           # the workspace layout (and size) depends on the arithmetic mode
        self.precision = mode
        return self

    def keep_activations(self, keep: bool = True):
        """Diagnostics: give every activation buffer its own memory so that :meth:`tap` can read any stage after a
        forward.  By default buffers with disjoint lifetimes share memory (0.26 instead of 0.55 GB per forward-sample)."""
        _lib.check(_lib.lib().misonet_net_keep_activations(self._net, 1 if keep else 0))
        self._ws.clear()
        self._keep = bool(keep)
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("misonet_amd implements the inference path only")
        return self.eval()

    def state_dict(self):
        return OrderedDict((k, torch.from_numpy(v.copy())) for k, v in self._sd.items())

    def load_state_dict(self, state_dict, strict=True):
        """Accepts the reference's ``package['model_state_dict']`` (run.py:139-151): same key names and shapes."""
        missing = [k for k in self.spec if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self.spec]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing keys {missing[:4]}..., "
                               f"unexpected keys {unexpected[:4]}...")
        L = _lib.lib()
        for k, shape in self.spec.items():
            if k not in state_dict:
                continue
            v = state_dict[k]
            if isinstance(v, torch.Tensor):
                v = v.detach().cpu().numpy()
            if k.endswith(".num_batches_tracked"):                       # BatchNorm1d bookkeeping: kept, not used (eval mode)
                self._sd[k] = np.asarray(v, dtype=np.int64)
                continue
            v = np.ascontiguousarray(v, dtype=np.float32)
            if tuple(v.shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {k}: copying a param with shape {tuple(v.shape)}, "
                                   f"expected {tuple(shape)}")
            self._sd[k] = v
            _lib.check(L.misonet_net_set_tensor(self._net, k.encode(), v.ctypes.data_as(C.c_void_p), v.size))
        self._committed = False
        return self

    def __repr__(self):
        return (f"{type(self).__name__}(num_spks={self.num_spks}, num_ch={self.num_ch}, en={list(self.en_ch)}, "
                f"de={list(self.de_ch)}, TCN(2,7,128), backend=libmisonet_hip gfx950, "
                f"params={sum(int(np.prod(s)) for s in self.spec.values())})")

    # ---- internals ----------------------------------------------------------------------------------------------
    def _commit(self):
        if self._committed:
            return
        if self._device is None:
            self.cuda()
        with torch.cuda.device(self._device):
            _lib.check(_lib.lib().misonet_net_commit(self._net))
        self._committed = True

    def _workspace(self, B, T):
        key = (B, T)
        ws = self._ws.get(key)
        if ws is None:
            self._ws.clear()
            n = _lib.lib().misonet_net_workspace_bytes(self._net, B, T)
            ws = torch.empty(n, dtype=torch.uint8, device=self._device)
            self._ws[key] = ws
        return ws

    def _forward_segments(self, segs: Sequence[torch.Tensor], check_nan=True):
        self._commit()
        segs = [self._as_c64(s) for s in segs]
        B, _, T, F = segs[0].shape
        if F != W.N_FREQ:
            raise ValueError(f"the network is defined for F = {W.N_FREQ} frequency bins (nperseg 256); got {F}")
        for s in segs[1:]:
            if s.shape[0] != B or s.shape[2] != T or s.shape[3] != F:
                raise ValueError("input segments disagree in B/T/F")
        if T < 2:
            # the reference fails here too: nn.InstanceNorm2d at the F = 1 bottleneck sees ONE element per (sample, channel)
            # and torch raises this ValueError (model.py:89, 413) -- a variance of one value is not a statistic
            raise ValueError(f"Expected more than 1 spatial element when training, got input size "
                             f"torch.Size([{B}, 128, {T}, 1]) (the network needs T >= 2 frames)")
        if sum(s.shape[1] for s in segs) * 2 != self.in_ch:
            raise ValueError(f"expected {self.in_ch // 2} complex input channels, got {sum(s.shape[1] for s in segs)}")
        ws = self._workspace(B, T)
        out = torch.empty((B, self.num_spks, T, F), dtype=torch.complex64, device=self._device)
        ptrs = (C.c_void_p * len(segs))(*[s.data_ptr() for s in segs])
        chans = (C.c_int * len(segs))(*[s.shape[1] for s in segs])
        L = _lib.lib()
        with torch.cuda.device(self._device):
            st = _lib.stream_ptr(self._device)
            _lib.check(L.misonet_net_forward(self._net, len(segs), ptrs, chans, B, T, out.data_ptr(), ws.data_ptr(),
                                             ws.numel(), st))
            if check_nan:
                _lib.check(L.misonet_net_check(self._net, ws.data_ptr(), st))
        return out

    def _as_c64(self, x):
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(x)
        if not x.is_complex():
            raise TypeError("expected a complex STFT tensor [B, ch, T, F]")
        if x.dim() != 4:
            raise ValueError("expected a 4-D tensor [B, ch, T, F]")
        if self._device is None:
            self.cuda()
        if x.device != self._device:
            raise RuntimeError(f"Expected all tensors to be on the same device, but found {x.device} and {self._device}")
        return x.to(torch.complex64).contiguous()

    def tap(self, name, B, T):
        """Diagnostic: an intermediate activation of the last forward as float32 [B, C, T, F] (reference layout)."""
        L = _lib.lib()
        c, f = C.c_int(), C.c_int()
        _lib.check(L.misonet_net_tap_shape(self._net, name.encode(), C.byref(c), C.byref(f)))
        dst = torch.empty((B, c.value, T, f.value), dtype=torch.float32, device=self._device)
        ws = self._ws[(B, T)]
        with torch.cuda.device(self._device):
            _lib.check(L.misonet_net_tap(self._net, name.encode(), ws.data_ptr(), B, T, dst.data_ptr(),
                                         _lib.stream_ptr(self._device)))
        return dst


class MISO_1(_Trunk):
    """Drop-in for reference model.MISO_1 (model.py:8-111): complex [B,M,T,129] -> complex64 [B,num_spks,T,129];
    microphone 0 of the input is the reference microphone."""
    _extra_in = 0

    def forward(self, mixture, check_nan=True):
        return self._forward_segments([mixture], check_nan)

    __call__ = forward


class MISO_3(_Trunk):
    """Drop-in for reference model.MISO_3 (model.py:282-395): forward(mixture [B,M,T,F], a [B,1,T,F], b [B,1,T,F]).
    Channel order is cat(mixture, a, b) for the real parts, then the imaginary parts (model.py:360-364); the
    reference harness passes a = beamformer output, b = MISO1 estimate (tester.py:1242)."""
    _extra_in = 2

    def forward(self, mixture, MISO1, BF, check_nan=True):
        return self._forward_segments([mixture, MISO1, BF], check_nan)

    __call__ = forward
