"""On-device MISO1 -> MVDR -> MISO3 pipeline: the body of the reference's Tester_Enhance.inference
(reference tester.py:846-975) with every stage kept in HBM, plus the utterance sharding used for multi-GPU runs.

``Enhancer`` mirrors the part of ``Tester_Enhance`` that does arithmetic:
    __init__(model_sep, model, num_spks, ref_ch, ...)   tester.py:799-825
    MISO1_Inference / alignment / Apply_Beamforming / MISO3_inference   tester.py:874-939  -> ``enhance``
    ISTFT + int16 + gap trim + chunk concat                               tester.py:949-969  -> ``to_wav_int16``
Unlike the reference it is correct for batch > 1 (the reference broadcasts the last batch row, tester.py:1065):
every utterance of the batch gets its own alignment, i.e. results equal the reference run utterance by utterance.

Multi-GPU: utterances are independent, so the batch is block-split over ranks with no collective on the data
path; ``gather_outputs`` is the optional all_gather of the results (RCCL on GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from . import stft as S
from .model import MISO_1, MISO_3
from .weights import N_FREQ


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block split of ``n_items`` utterances: rank r owns [lo, hi).
    The first ``n_items % world`` ranks get one extra item."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_outputs(local: torch.Tensor, n_items: int, group=None) -> torch.Tensor:
    """all_gather of per-rank results (first dim = utterances of this rank's shard) back into global order."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = [shard_range(n_items, r, world) for r in range(world)]
    cap = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    if pad.is_complex():
        bufs = [torch.empty_like(torch.view_as_real(pad)) for _ in range(world)]
        dist.all_gather(bufs, torch.view_as_real(pad).contiguous(), group=group)
        bufs = [torch.view_as_complex(b) for b in bufs]
    else:
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=0)


def run_sharded(process: Callable[[int, int], torch.Tensor], n_items: int, rank: int, world: int, gather: bool,
                group=None):
    """Run ``process(lo, hi)`` on this rank's shard; optionally gather every rank's result in global order."""
    lo, hi = shard_range(n_items, rank, world)
    local = process(lo, hi)
    if gather and world > 1:
        return gather_outputs(local, n_items, group)
    return local


class CapturedPass:
    """One :meth:`Enhancer.enhance` pass captured in a HIP graph, together with everything the graph's kernels point at.

    The graph bakes raw device addresses (workspace, inputs, result), so this object OWNS references to all of them: the
    workspace stays alive for as long as the graph does, whatever other batch sizes / lengths / precision modes the
    Enhancer is used with in between (its own cache may drop and re-allocate workspaces freely).  ``replay()`` re-runs the
    pass; new inputs are copied INTO ``mix`` / ``clean`` first (``load``).  Unpacks as ``(graph, out)``."""

    def __init__(self, graph, out, ws, mix, clean, key, owner=None):
        self.graph, self.out, self.mix, self.clean = graph, out, mix, clean
        self._ws, self.key = ws, key
        # a STRONG reference: the graph's kernels read the Enhancer's pipeline handle and the nets' device weights, so they must
        # outlive the graph -- and check() must stay usable for as long as replay() is (the Enhancer keeps only a WeakSet of its
        # captured passes: no cycle)
        self._owner = owner

    def check(self):
        """Synchronise and raise FloatingPointError if the last replay produced a NaN (the pass's flag word lives in the
        workspace this object owns, not in the Enhancer's eager workspace)."""
        if self._owner is None:
            raise RuntimeError("this pass was captured without an owning Enhancer")
        self._owner.check(self.key[0], self.key[1], captured=self)

    def load(self, mix, clean=None):
        self.mix.copy_(mix)
        if self.clean is not None and clean is not None:
            self.clean.copy_(clean)

    def replay(self):
        self.graph.replay()
        return self.out

    def __iter__(self):                       # ``g, out = enh.capture_graph(...)``
        yield self
        yield self.out


class Enhancer:
    """Fused on-device MISO1 -> (alignment) -> MVDR -> MISO3 for batches of 4 s chunks."""

    def __init__(self, model_sep: MISO_1, model: Optional[MISO_3], num_spks: int = 2, ref_ch: int = 0, epsi: float = 1e-6):
        """``model = None``: a separation-only Enhancer (:meth:`separate`, :meth:`beamform_utterance`,
        :meth:`beamform_chunks` -- what the reference's ``Tester_Beamforming`` needs: it has no MISO_3, tester.py:259-262)."""
        if not isinstance(model_sep, MISO_1) or not (model is None or isinstance(model, MISO_3)):
            raise TypeError("Enhancer needs misonet_amd.MISO_1 and misonet_amd.MISO_3 (or None) instances")
        self.model_sep, self.model = model_sep, model
        self.num_spks, self.ref_ch, self.num_ch = int(num_spks), int(ref_ch), model_sep.num_ch
        if model_sep._device is None:
            model_sep.cuda()
        if model is not None and model._device is None:
            model.cuda(model_sep._device)
        self.device = model_sep._device
        if model is not None and model._device != self.device:
            raise RuntimeError(f"MISO_1 is on {self.device} but MISO_3 on {model._device}")
        model_sep._commit()
        if model is not None:
            model._commit()
        self._pipe = C.c_void_p()
        _lib.check(_lib.lib().misonet_pipeline_create(model_sep._net, model._net if model is not None else None, self.num_ch,
                                                      self.num_spks, self.ref_ch, float(epsi), C.byref(self._pipe)))
        self._ws: Dict[tuple, torch.Tensor] = {}
        self._captured = weakref.WeakSet()     # live CapturedPass objects: check(B, T) must not silently look elsewhere

    def __del__(self):
        try:
            if self._pipe:
                _lib.lib().misonet_pipeline_destroy(self._pipe)
                self._pipe = C.c_void_p()
        except Exception:
            pass

    def _ready(self):
        """Re-commit after a later ``load_state_dict`` / ``.cuda()`` on either model (a checkpoint loaded after the
        Enhancer was built clears the committed state) and follow a device move."""
        if self.model is not None and self.model_sep._device != self.model._device:
            raise RuntimeError(f"MISO_1 is on {self.model_sep._device} but MISO_3 on {self.model._device}")
        if self.model_sep._device != self.device:
            self.device = self.model_sep._device
            self._ws.clear()
        self.model_sep._commit()
        if self.model is not None:
            self.model._commit()

    def _check_c64(self, x, name, shape=None):
        """shared argument check of enhance / separate: complex tensor on this Enhancer's device, optional shape"""
        if not isinstance(x, torch.Tensor) or not x.is_complex():
            raise TypeError(f"{name} must be a complex torch tensor")
        if x.device != self.device:
            raise RuntimeError(f"Expected all tensors to be on the same device, but found {name} on {x.device} and the "
                               f"models on {self.device}")
        if shape is not None and tuple(x.shape) != tuple(shape):
            raise ValueError(f"{name} must be {list(shape)}, got {list(x.shape)}")
        return x.to(torch.complex64).contiguous()

    def _ws_key(self, B, T):
        # the layout depends on the arithmetic modes and on whether buffers may share memory
        m3 = self.model
        return (B, T, self.model_sep.precision, m3.precision if m3 is not None else None, self.model_sep._keep,
                m3._keep if m3 is not None else None)

    def workspace(self, B, T):
        key = self._ws_key(B, T)
        ws = self._ws.get(key)
        if ws is None:
            self._ws.clear()
            n = _lib.lib().misonet_pipeline_workspace_bytes(self._pipe, B, T)
            ws = torch.empty(n, dtype=torch.uint8, device=self.device)
            ws[:16].zero_()                    # the NaN flag word: check() on a workspace no pass has run in reads "clean"
            self._ws[key] = ws
        return ws

    def enhance(self, mix: torch.Tensor, clean: Optional[torch.Tensor] = None, want_bf=False, want_miso1=False,
                check_nan=True, out: Optional[torch.Tensor] = None):
        """mix complex [B,M,T,F] (device); clean complex [B,S,T,F] = clean sources at ref_ch (tester.py:889-891) or
        None to skip the clean-reference re-ordering.  Returns MISO3 output complex64 [B,S,T,F]
        (and a dict with 'bf' [B,S,T,F] / 'miso1' [B,S,M,T,F] when requested)."""
        self._ready()
        if self.model is None:
            raise RuntimeError("this Enhancer was built without MISO_3 (separation only): use separate() / beamform_*()")
        if not isinstance(mix, torch.Tensor) or mix.dim() != 4:
            raise ValueError("mix must be a 4-D complex tensor [B, M, T, F]")
        mix = self._check_c64(mix, "mix")
        B, M, T, F = mix.shape
        if M != self.num_ch:
            raise ValueError(f"expected {self.num_ch} microphones, got {M}")
        if F != N_FREQ:
            raise ValueError(f"the networks are defined for F = {N_FREQ} frequency bins, got {F}")
        if T < 2:                              # as the reference: InstanceNorm over one element (model.py:89, 413)
            raise ValueError(f"Expected more than 1 spatial element when training, got T = {T} frame(s)")
        if clean is not None:
            clean = self._check_c64(clean, "clean", (B, self.num_spks, T, F))
        ws = self.workspace(B, T)
        if out is None:
            out = torch.empty((B, self.num_spks, T, F), dtype=torch.complex64, device=self.device)
        elif (not isinstance(out, torch.Tensor) or out.dtype != torch.complex64 or out.device != self.device
              or tuple(out.shape) != (B, self.num_spks, T, F) or not out.is_contiguous()):
            raise ValueError(f"out must be a contiguous complex64 tensor [{B}, {self.num_spks}, {T}, {F}] on {self.device}")
        bf = torch.empty_like(out) if want_bf else None
        m1 = torch.empty((B, self.num_spks, M, T, F), dtype=torch.complex64, device=self.device) if want_miso1 else None
        L = _lib.lib()
        with torch.cuda.device(self.device):
            st = _lib.stream_ptr(self.device)
            _lib.check(L.misonet_pipeline_run(self._pipe, mix.data_ptr(), clean.data_ptr() if clean is not None else None,
                                              B, T, out.data_ptr(), bf.data_ptr() if want_bf else None,
                                              m1.data_ptr() if want_miso1 else None, ws.data_ptr(), ws.numel(), st))
            if check_nan:
                _lib.check(L.misonet_pipeline_check(self._pipe, ws.data_ptr(), st))
        if want_bf or want_miso1:
            return out, dict(bf=bf, miso1=m1)
        return out

    def capture_graph(self, mix: torch.Tensor, clean: Optional[torch.Tensor] = None) -> CapturedPass:
        """Capture one :meth:`enhance` pass over the given (static) input tensors into a HIP graph and return a
        :class:`CapturedPass` (unpacks as ``(graph, out)``): ``graph.replay()`` re-runs the ≈ 450 kernel launches of the
        pass with one host call and leaves the result in ``out``; new inputs are copied INTO ``mix`` / ``clean`` before a
        replay.  Every entry point of the C ABI is asynchronous on the caller's stream and allocates nothing, so the whole
        pass is capturable.  The graph holds raw addresses: ``mix`` / ``clean`` must ALREADY be contiguous complex64
        device tensors (a converted copy would be what the graph reads, not the caller's tensor), and the returned object
        keeps the workspace, the inputs and the result alive -- the Enhancer's own workspace cache may be re-used for other
        shapes / modes meanwhile.  NaN checking is the caller's (``misonet_pipeline_check`` synchronises, so it is not part
        of the graph): call ``captured.check()`` (or ``enh.check(B, T, captured=captured)``) after a replay -- the flag word
        lives in the workspace the CapturedPass owns; ``enh.check(B, T)`` without it refuses to answer while a captured pass
        of that shape is alive (it would read the eager workspace, which a replay never touches)."""
        self._ready()
        for name, x in (("mix", mix), ("clean", clean)):
            if x is None:
                continue
            if (not isinstance(x, torch.Tensor) or x.dtype != torch.complex64 or not x.is_contiguous()
                    or x.device != self.device):
                raise ValueError(f"capture_graph: {name} must be a contiguous complex64 tensor on {self.device} (the graph "
                                 "records its address)")
        B, M, T, F = mix.shape
        out = torch.empty((B, self.num_spks, T, F), dtype=torch.complex64, device=self.device)
        self.enhance(mix, clean, check_nan=False, out=out)               # warm-up: workspace allocation, lazy init
        torch.cuda.synchronize(self.device)
        key = self._ws_key(B, T)
        ws = self._ws[key]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.device(self.device), torch.cuda.graph(g):
            self.enhance(mix, clean, check_nan=False, out=out)
        if self._ws.get(key) is not ws:                                  # (cannot happen: same key as the warm-up)
            raise RuntimeError("workspace changed during capture")
        # from now on this workspace belongs to the graph: the cache hands out a fresh one for the same key, so an eager
        # pass between two replays cannot clobber activations a replay is about to read
        del self._ws[key]
        cp = CapturedPass(g, out, ws, mix, clean, key, owner=self)
        self._captured.add(cp)
        return cp

    def check(self, B: int, T: int, captured: Optional[CapturedPass] = None):
        """Synchronise and raise FloatingPointError if the last pass on the (B, T) workspace produced a NaN
        (``captured``: the workspace of that captured pass instead of the eager one)."""
        if captured is None:
            key = self._ws_key(B, T)
            live = [c for c in self._captured if c.key == key]
            if live and key not in self._ws:
                # after capture_graph the (B, T) workspace belongs to the graph; a fresh eager workspace always reads
                # clean, so `g.replay(); enh.check(B, T)` would silently check nothing
                raise RuntimeError(f"check({B}, {T}): a captured pass owns the workspace of this shape and no eager pass has run "
                                   "since; call captured.check() / check(B, T, captured=...) for the replay's NaN flag")
        ws = captured._ws if captured is not None else self.workspace(B, T)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().misonet_pipeline_check(self._pipe, ws.data_ptr(), _lib.stream_ptr(self.device)))

    def enhance_wav(self, wav: torch.Tensor, clean_wav: Optional[torch.Tensor] = None, want_bf=False, want_miso1=False,
                    check_nan=True):
        """Waveform entry: wav float32 [B, n_samples, M] on the device (one 4 s chunk per row,
        time-major as ``librosa.load(...).T``), clean_wav float32 [B, n_samples, S] = the clean sources at ref_ch, or
        None.  The STFT front-end runs as a HIP kernel straight into the network's layout; returns what
        :meth:`enhance` returns (T = n_samples // 64 + 1 frames)."""
        self._ready()
        if self.model is None:
            raise RuntimeError("this Enhancer was built without MISO_3 (separation only): use separate() / beamform_*()")
        if not isinstance(wav, torch.Tensor) or wav.dim() != 3 or wav.is_complex():
            raise ValueError("wav must be a real 3-D tensor [B, n_samples, M]")
        if wav.device != self.device:
            raise RuntimeError(f"Expected all tensors to be on the same device, but found wav on {wav.device} and the "
                               f"models on {self.device}")
        wav = wav.to(torch.float32).contiguous()
        B, Ls, M = wav.shape
        if M != self.num_ch:
            raise ValueError(f"expected {self.num_ch} microphones, got {M}")
        if clean_wav is not None:
            if not isinstance(clean_wav, torch.Tensor) or clean_wav.device != self.device:
                raise RuntimeError(f"clean_wav must be a tensor on {self.device}")
            clean_wav = clean_wav.to(torch.float32).contiguous()
            if tuple(clean_wav.shape) != (B, Ls, self.num_spks):
                raise ValueError("clean_wav must be [B, n_samples, num_spks]")
        L = _lib.lib()
        T = L.misonet_stft_frames(Ls)
        if T < 2:                              # n_samples < 64: one frame -- the reference's instance norms raise (model.py:89, 413)
            raise ValueError(f"Expected more than 1 spatial element when training, got T = {T} frame(s) ({Ls} samples)")
        ws = self.workspace(B, T)
        out = torch.empty((B, self.num_spks, T, 129), dtype=torch.complex64, device=self.device)
        bf = torch.empty_like(out) if want_bf else None
        m1 = torch.empty((B, self.num_spks, M, T, 129), dtype=torch.complex64, device=self.device) if want_miso1 else None
        with torch.cuda.device(self.device):
            st = _lib.stream_ptr(self.device)
            _lib.check(L.misonet_pipeline_run_wav(self._pipe, wav.data_ptr(),
                                                  clean_wav.data_ptr() if clean_wav is not None else None, B, Ls,
                                                  out.data_ptr(), bf.data_ptr() if want_bf else None,
                                                  m1.data_ptr() if want_miso1 else None, ws.data_ptr(), ws.numel(), st))
            if check_nan:
                _lib.check(L.misonet_pipeline_check(self._pipe, ws.data_ptr(), st))
        if want_bf or want_miso1:
            return out, dict(bf=bf, miso1=m1)
        return out

    def enhance_wav_int16(self, wav: torch.Tensor, clean_wav: Optional[torch.Tensor] = None, check_nan=True) -> torch.Tensor:
        """The reference's unit of work on the device: wav in -> int16 wav out (tester.py:865-867 H2D ... 949-952 iSTFT,
        x 32767, int16).  wav float32 [B, n_samples, M] (device) -> int16 [B, S, 64 * (n_samples // 64)] (device; = n_samples
        when that is a multiple of the hop, as the reference's 4 s chunks are): HIP STFT front-end,
        the fused pipeline, ONE ``istft_k`` launch over the B * S enhanced spectrograms incl. the truncating cast --
        nothing leaves the device and nothing synchronises (with ``check_nan=False``)."""
        return S.istft_int16(self.enhance_wav(wav, clean_wav, check_nan=check_nan))

    def stream_wav(self, batches, depth: int = 2, check_nan: bool = True):
        """HOST-resident batches in, int16 waves out, copies overlapped with compute.

        ``batches`` yields ``wav`` or ``(wav, clean_wav)``: CPU float32 tensors [B, n_samples, M] / [B, n_samples, S] (pinned
        or not; un-pinned ones are staged through a pinned buffer of this generator).  For every batch, in order, yields
        int16 ndarray [B, S, n_samples].  Three streams: H2D copies of batch i + 1 (copy-in stream) and the D2H copy of
        batch i - 1's int16 result (copy-out stream) run beside the pipeline of batch i (current stream); ``depth`` device
        input / pinned output slots.  The host only blocks on the result of the OLDEST batch in flight, so the GPU always
        has the next batch queued.  A NaN in batch i raises FloatingPointError when that batch is handed out (the
        pipeline's flag word travels with the result instead of a synchronising check).

        Contracts: the compute stream is whatever stream is CURRENT when the generator resumes for a batch
        (re-read every iteration; the events that protect the slot buffers are recorded on and waited by that stream).  A
        PINNED input is the direct source of an asynchronous H2D copy: leave it untouched until the result of ITS batch has
        been yielded (un-pinned inputs are copied into the generator's own staging buffer before ``next()`` returns).  A
        CUDA-resident input is read on the copy-in stream after the work queued on the stream that was current when it
        was handed over."""
        import collections
        self._ready()
        dev = self.device
        depth = max(1, int(depth))
        with torch.cuda.device(dev):
            s_in, s_out = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
            slots = [dict() for _ in range(depth)]
            pending = collections.deque()

            def finish(rec):
                rec["ev_out"].synchronize()
                if check_nan and int(rec["flag_h"][0]) != 0:
                    raise FloatingPointError("libmisonet_hip: NaN in pipeline output")
                return rec["pcm_h"].numpy().copy()

            def staged(sl, name, host):
                """pinned source for the H2D copy of `host` (itself if already pinned, or already on a device)"""
                host = host.to(torch.float32).contiguous() if host.dtype != torch.float32 or not host.is_contiguous() else host
                if host.is_cuda or host.is_pinned():
                    return host
                pin = sl.get("pin_" + name)
                if pin is None or pin.shape != host.shape:
                    pin = sl["pin_" + name] = torch.empty(host.shape, dtype=torch.float32, pin_memory=True)
                elif "ev_in" in sl:
                    sl["ev_in"].synchronize()                 # the previous H2D out of this staging buffer is done
                pin.copy_(host)
                return pin

            for i, item in enumerate(batches):
                wav_h, clean_h = item if isinstance(item, (tuple, list)) else (item, None)
                while len(pending) >= depth:                  # slot i % depth still belongs to batch i - depth
                    yield finish(pending.popleft())
                sl = slots[i % depth]
                cur = torch.cuda.current_stream(dev)          # the consumer may have changed streams between two next() calls
                src_w = staged(sl, "wav", wav_h)
                src_c = staged(sl, "clean", clean_h) if clean_h is not None else None
                fresh = False
                if sl.get("wav_d") is None or sl["wav_d"].shape != src_w.shape:
                    sl["wav_d"] = torch.empty(src_w.shape, dtype=torch.float32, device=dev)
                    fresh = True
                if src_c is not None and (sl.get("clean_d") is None or sl["clean_d"].shape != src_c.shape):
                    sl["clean_d"] = torch.empty(src_c.shape, dtype=torch.float32, device=dev)
                    fresh = True
                if src_w.is_cuda or (src_c is not None and src_c.is_cuda):
                    s_in.wait_stream(cur)                     # a device-resident input: after its producer on the current stream
                if fresh:
                    # a new slot buffer comes out of the CURRENT stream's allocator pool: its block may be one that work
                    # already queued on the current stream still writes (e.g. the previous batch's spectrogram, freed on the
                    # host the moment its iSTFT was enqueued).  The copy-in stream must not touch it before that work is
                    # done.  (Found when the iSTFT became a HIP kernel: torch.istft synchronises the host, which had hidden it.)
                    s_in.wait_stream(cur)
                with torch.cuda.stream(s_in):
                    if "ev_free" in sl:
                        s_in.wait_event(sl["ev_free"])        # the pass that read this slot's device buffers is done
                    sl["wav_d"].copy_(src_w, non_blocking=True)
                    if src_c is not None:
                        sl["clean_d"].copy_(src_c, non_blocking=True)
                    # device-resident inputs are read by the copy-in stream: the caching allocator must not hand their blocks out
                    # on the current stream before that copy is done, whatever the caller does with them after next() returns
                    if src_w.is_cuda:
                        src_w.record_stream(s_in)
                    if src_c is not None and src_c.is_cuda:
                        src_c.record_stream(s_in)
                    sl["ev_in"] = torch.cuda.Event()
                    sl["ev_in"].record(s_in)
                cur.wait_event(sl["ev_in"])
                B, Ls, _ = sl["wav_d"].shape
                pcm = self.enhance_wav_int16(sl["wav_d"], sl["clean_d"] if src_c is not None else None, check_nan=False)
                T = _lib.lib().misonet_stft_frames(Ls)
                flag = self.workspace(B, T)[:4].view(torch.int32).clone()      # the pass's NaN flag word (ws[0])
                sl["ev_free"] = torch.cuda.Event()
                sl["ev_free"].record(cur)
                rec = {"pcm_h": sl.get("pcm_h"), "flag_h": sl.get("flag_h")}
                if rec["pcm_h"] is None or rec["pcm_h"].shape != pcm.shape:
                    rec["pcm_h"] = sl["pcm_h"] = torch.empty(pcm.shape, dtype=torch.int16, pin_memory=True)
                    rec["flag_h"] = sl["flag_h"] = torch.zeros(1, dtype=torch.int32, pin_memory=True)
                with torch.cuda.stream(s_out):
                    s_out.wait_event(sl["ev_free"])
                    rec["pcm_h"].copy_(pcm, non_blocking=True)
                    rec["flag_h"].copy_(flag, non_blocking=True)
                    pcm.record_stream(s_out)
                    flag.record_stream(s_out)
                    rec["ev_out"] = torch.cuda.Event()
                    rec["ev_out"].record(s_out)
                pending.append(rec)
            while pending:
                yield finish(pending.popleft())

    def separate(self, mix: torch.Tensor, clean: Optional[torch.Tensor] = None, check_nan=True) -> torch.Tensor:
        """Separation stage only: MISO1_Inference over the circular shifts + alignments (tester.py:1014-1068, 889-915).
        mix complex [B,M,T,F] -> aligned estimates complex64 [B,S,M,T,F] (speaker, microphone)."""
        self._ready()
        if not isinstance(mix, torch.Tensor) or mix.dim() != 4:
            raise ValueError("mix must be a 4-D complex tensor [B, M, T, F]")
        mix = self._check_c64(mix, "mix")
        B, M, T, F = mix.shape
        if M != self.num_ch:
            raise ValueError(f"expected {self.num_ch} microphones, got {M}")
        if F != N_FREQ:
            raise ValueError(f"the networks are defined for F = {N_FREQ} frequency bins, got {F}")
        if T < 2:                              # as the reference: InstanceNorm over one element (model.py:89, 413)
            raise ValueError(f"Expected more than 1 spatial element when training, got T = {T} frame(s)")
        if clean is not None:
            clean = self._check_c64(clean, "clean", (B, self.num_spks, T, F))
        ws = self.workspace(B, T)
        m1 = torch.empty((B, self.num_spks, M, T, F), dtype=torch.complex64, device=self.device)
        L = _lib.lib()
        with torch.cuda.device(self.device):
            st = _lib.stream_ptr(self.device)
            _lib.check(L.misonet_pipeline_run(self._pipe, mix.data_ptr(), clean.data_ptr() if clean is not None else None,
                                              B, T, None, None, m1.data_ptr(), ws.data_ptr(), ws.numel(), st))
            if check_nan:
                _lib.check(L.misonet_pipeline_check(self._pipe, ws.data_ptr(), st))
        return m1

    def beamform_utterance(self, obs_splits: List[torch.Tensor], clean_splits: List[torch.Tensor], gap: int,
                           epsi: float = 1e-6, max_batch: int = 16, to_host: bool = True):
        """Utterance-wise MVDR of the reference's Tester_Beamforming (tester.py:340-449, ``utterance_flag``) for ONE
        recording: its splits are separated as ONE batch (:meth:`separate`; the reference runs them one by one), all
        (speaker, mic) estimates and the observation go back to the time domain with one batched iSTFT, the splits are
        stitched (last one trimmed by ``gap``), the whole recording is re-analysed by the HIP STFT front-end and ONE MVDR per
        speaker is solved over all its frames (spatial covariances accumulated over the full utterance instead of per 4 s
        chunk).  obs_splits: list of complex [M,T,F]; clean_splits: list of complex [S,T,F].  Returns int16 [S, n] (an ndarray;
        ``to_host=False``: the device tensor, nothing synchronises)."""
        from .beamform import Apply_Beamforming
        K = len(obs_splits)
        if K < 1 or len(clean_splits) != K:
            raise ValueError("obs_splits / clean_splits must be non-empty lists of equal length")
        obs = torch.stack([torch.as_tensor(o) for o in obs_splits]).to(self.device, non_blocking=True)   # [K,M,T,F]
        cl = torch.stack([torch.as_tensor(c) for c in clean_splits]).to(self.device, non_blocking=True)  # [K,S,T,F]
        est = torch.cat([self.separate(obs[lo:lo + max_batch], cl[lo:lo + max_batch])          # [K,S,M,T,F], in groups of
                         for lo in range(0, K, max_batch)])                                   # <= max_batch splits (workspace)
        e = S.istft(est)                                                                      # [K,S,M,chunk]
        o = S.istft(obs)                                                                      # [K,M,chunk]
        n = e.shape[-1]
        keep = [n] * (K - 1) + [n - int(gap)]
        est_t = torch.cat([e[k, ..., : keep[k]] for k in range(K)], dim=-1)                   # [S,M,L]
        obs_t = torch.cat([o[k, ..., : keep[k]] for k in range(K)], dim=-1)                   # [M,L]
        Ls = obs_t.shape[-1]
        pad = (-Ls) % S.HOP                                                   # scipy's padded=True: whole hops
        sig = torch.cat([obs_t[None], est_t], dim=0)                          # [1+S, M, L]
        sig = torch.nn.functional.pad(sig, (0, pad)).permute(0, 2, 1).contiguous()      # [1+S, Lp, M]
        spec = S.stft_hip(sig)                                                # [1+S, M, Tt, F]
        mix_bf = spec[0].permute(2, 0, 1)[None]                               # [1,F,M,Tt]
        bf = torch.stack([Apply_Beamforming(spec[1 + s].permute(2, 0, 1)[None], mix_bf, epsi)[0]
                          for s in range(self.num_spks)])                     # [S,Tt,F]
        pcm = S.istft_int16(bf)
        return pcm.cpu().numpy() if to_host else pcm

    def beamform_chunks(self, mix: torch.Tensor, clean: Optional[torch.Tensor] = None, epsi: float = 1e-6) -> torch.Tensor:
        """Chunk-wise MVDR (BASELINE configs[2]: MISO1 -> MVDR; the ``utterance_flag = False`` branch of the reference's
        Tester_Beamforming, tester.py:452-535): separation of a batch of 4 s chunks, then one MVDR per (chunk, speaker) over
        the chunk's own frames.  mix complex [B,M,T,F], clean complex [B,S,T,F] or None -> beamformer outputs complex64
        [B,S,T,F].  (With MISO_3 attached, ``enhance(..., want_bf=True)`` returns the same tensor from the fused pass.)"""
        from .beamform import Apply_Beamforming
        est = self.separate(mix, clean)                                                       # [B,S,M,T,F]
        mix_bf = self._check_c64(mix, "mix").permute(0, 3, 1, 2)                              # [B,F,M,T]
        return torch.stack([Apply_Beamforming(est[:, s].permute(0, 3, 1, 2), mix_bf, epsi)
                            for s in range(self.num_spks)], dim=1)                            # [B,S,T,F]

    def enhance_recording(self, wav_observe, wav_clean=None, num_ch_utilize: Optional[int] = None, chunk_size: int = 64000,
                          max_batch: int = 16, save_path: Optional[str] = None, fs: int = 16000) -> np.ndarray:
        """Recording in -> enhanced int16 waves out: the reference's loader item AND its tester body as one device-side
        object (``AudioDataset_Test.__getitem__``, dataloader/data.py:524-597, + ``Tester_Enhance.inference``,
        tester.py:846-975), without host STFT dicts.

        wav_observe: float32 [L, M_all] (ndarray or CPU tensor, time-major as ``read_wav`` returns it); wav_clean: the
        clean sources, a sequence of S arrays [L, M_all] (``<name>_0.wav``, ``<name>_1.wav``, data.py:527-528) or None.
        Steps, each as the reference does it: microphone sub-sampling ``[0:M:M // num_ch_utilize]`` (data.py:544; default:
        the network's microphone count), 4 s chunks with the last one zero-padded by ``gap`` (data.py:536-595;
        ``L == chunk_size``, which the reference leaves unhandled, is one chunk), the clean sources at ``ref_ch`` of the
        sub-sampled array (tester.py:889-890), per chunk STFT -> MISO1 x M -> alignments -> MVDR x S -> MISO3 x S -> iSTFT ->
        x 32767 -> int16 (HIP kernels end to end, the chunks of the recording as batches of <= ``max_batch`` through
        :meth:`stream_wav`: H2D of the next batch and D2H of the previous one beside the compute), chunks stitched with the
        padded tail dropped (tester.py:960-969).  Returns int16 [S, L]; ``save_path`` = "<dir>/<wav_name>" also writes
        ``<save_path>_{s}.wav`` as 24-bit PCM (tester.py:972-974)."""
        self._ready()
        if self.model is None:
            raise RuntimeError("this Enhancer was built without MISO_3 (separation only): use separate() / beamform_*()")
        obs = np.asarray(wav_observe, dtype=np.float32)
        if obs.ndim != 2 or obs.shape[0] <= obs.shape[1]:
            raise ValueError("wav_observe must be [n_samples, n_mics] with n_samples > n_mics (data.py:510)")
        L, M_all = obs.shape
        n_use = self.num_ch if num_ch_utilize is None else int(num_ch_utilize)
        if n_use < 1 or n_use > M_all:
            raise ValueError(f"num_ch_utilize must be in [1, {M_all}]")
        mics = list(range(0, M_all, M_all // n_use))                       # data.py:544: [0:M:M // num_ch_utilize]
        if len(mics) != self.num_ch:
            raise ValueError(f"[0:{M_all}:{M_all // n_use}] selects {len(mics)} microphones, the networks take {self.num_ch}")
        if int(chunk_size) < 2 * S.HOP or int(chunk_size) % S.HOP:
            raise ValueError(f"chunk_size must be a multiple of the hop ({S.HOP}) and at least two hops")
        obs = obs[:, mics]
        cl = None
        if wav_clean is not None:
            if len(wav_clean) != self.num_spks:
                raise ValueError(f"wav_clean must hold {self.num_spks} source recordings")
            srcs = []
            for c in wav_clean:
                c = np.asarray(c, dtype=np.float32)
                if c.shape != (L, M_all):
                    raise ValueError("every clean source must have the shape of wav_observe")
                srcs.append(c[:, mics][:, self.ref_ch])
            cl = np.stack(srcs, axis=1)                                    # [L, S]
        pieces, gap = S.split_chunks(obs, int(chunk_size))
        cpieces = S.split_chunks(cl, int(chunk_size))[0] if cl is not None else None
        K = len(pieces)

        def batches():
            for lo in range(0, K, max_batch):
                w = torch.from_numpy(np.stack(pieces[lo:lo + max_batch]))
                yield (w, torch.from_numpy(np.stack(cpieces[lo:lo + max_batch]))) if cpieces is not None else w

        pcm = np.concatenate(list(self.stream_wav(batches())), axis=0)     # [K, S, chunk]
        out = np.stack([S.stitch_int16([pcm[k, s] for k in range(K)], gap) for s in range(self.num_spks)])
        if save_path is not None:
            import os
            os.makedirs(os.path.dirname(os.path.abspath(save_path)), exist_ok=True)
            for s in range(self.num_spks):
                S.write_wav_pcm24(f"{save_path}_{s}.wav", out[s], fs)
        return out

    def inference(self, data_loader, saveDir, fs=16000, write=True, max_batch=32):
        """Drop-in for ``Tester_Enhance.inference(data_loader, saveDir)`` (tester.py:846-975): the loader yields
        ``(split_observe_dict, split_clean_s0_dict, split_clean_s1_dict, gap, wav_name)`` with dict values complex
        ``[B, Ch, T, F]`` keyed '0', '1', ... (dataloader/data.py:524-597).  Every split runs through
        :meth:`enhance`; the int16 waves of the splits are stitched (last split trimmed by ``gap``) and written as
        ``<saveDir>/<wav_name>_{0,1}.wav`` (PCM-24).  Returns {wav_name: int16 [2, n_samples]}."""
        import os
        os.makedirs(saveDir, exist_ok=True)
        results = {}
        dev = self.device

        def h2d(x):
            """one asynchronous copy per tensor (pinned staging), instead of a synchronous ``.to(device)`` per split"""
            x = torch.as_tensor(x)
            if x.device == dev:
                return x
            if x.device.type == "cpu" and not x.is_pinned():
                x = x.contiguous().pin_memory()
            return x.to(dev, non_blocking=True)

        def finalize(rec):
            """the previous loader item: wait for ITS D2H only, stitch, write"""
            rec["ev"].synchronize()
            pcm = rec["pcm_h"].numpy()                                                        # [n_split, B, S, n]
            for b in range(pcm.shape[1]):
                gp = rec["gap"]
                g = int(gp[b]) if hasattr(gp, "__len__") else int(gp)
                wav = np.stack([S.stitch_int16([pcm[k, b, s] for k in range(pcm.shape[0])], g)
                                for s in range(self.num_spks)])                               # [S, n_total]
                name = rec["name"][b] if not isinstance(rec["name"], str) else rec["name"]
                results[name] = wav
                if write:
                    for s in range(self.num_spks):
                        S.write_wav_pcm24(os.path.join(saveDir, f"{name}_{s}.wav"), wav[s], fs)

        with torch.cuda.device(dev):
            s_out = torch.cuda.Stream(dev)
            prev = None
            for (obs_d, s0_d, s1_d, gap, wav_name) in data_loader:
                n_split = len(obs_d)
                obs, cl = [], []
                for k in range(n_split):
                    obs.append(h2d(obs_d[str(k)]))
                    s0 = h2d(torch.as_tensor(s0_d[str(k)])[:, self.ref_ch])                   # tester.py:889-890
                    s1 = h2d(torch.as_tensor(s1_d[str(k)])[:, self.ref_ch])
                    cl.append(torch.stack((s0, s1), dim=1))
                # the 4 s splits of a loader item are independent chunks of equal length (dataloader/data.py:558-595): they run
                # as ONE batch of n_split * B utterances (the reference: one pass per split), in groups of at most
                # ``max_batch``.  Bit-identical to split-by-split (results do not depend on the batch).
                Bk = obs[0].shape[0]
                obs, cl = torch.cat(obs), torch.cat(cl)                                        # [n_split * B, ...]
                out = torch.empty((obs.shape[0], self.num_spks) + tuple(obs.shape[2:]), dtype=torch.complex64, device=dev)
                for lo in range(0, obs.shape[0], max_batch):
                    self.enhance(obs[lo:lo + max_batch], cl[lo:lo + max_batch], out=out[lo:lo + max_batch])
                outs = out.reshape((n_split, Bk) + tuple(out.shape[1:]))                       # [n_split,B,S,T,F]
                pcm = S.istft_int16(outs)                        # ONE batched iSTFT over n_split * B * S spectrograms
                ev_done = torch.cuda.Event()
                ev_done.record(torch.cuda.current_stream(dev))
                rec = {"pcm_h": torch.empty(pcm.shape, dtype=torch.int16, pin_memory=True), "gap": gap, "name": wav_name}
                with torch.cuda.stream(s_out):                   # ONE D2H per loader item, beside the next item's compute
                    s_out.wait_event(ev_done)
                    rec["pcm_h"].copy_(pcm, non_blocking=True)
                    pcm.record_stream(s_out)
                    rec["ev"] = torch.cuda.Event()
                    rec["ev"].record(s_out)
                if prev is not None:
                    finalize(prev)
                prev = rec
            if prev is not None:
                finalize(prev)
        return results

    def to_wav_int16(self, enhanced_chunks: List[torch.Tensor], gap: int) -> np.ndarray:
        """tester.py:949-969 for one recording: list over 4 s splits of complex [S,T,F] -> int16 [S, n_samples].
        One batched iSTFT over all (split, speaker) spectrograms and one device-to-host copy."""
        pcm = S.istft_int16(torch.stack([torch.as_tensor(ch) for ch in enhanced_chunks])).cpu().numpy()   # [n_split, S, n]
        return np.stack([S.stitch_int16([pcm[k, s] for k in range(pcm.shape[0])], gap) for s in range(self.num_spks)])
