"""Drop-in for the reference's ``Tester_Enhance`` (reference tester.py:798-975) at the HARNESS level: same constructor
arguments as ``run.py:272-274`` passes, same ``test()`` / ``inference(data_loader, saveDir)`` surface, same outputs
(``<saveDir>/<wav_name>_{0,1}.wav`` PCM-24; ``cv_dev93`` / ``test_eval92`` sub-directories).  The arithmetic is
:class:`misonet_amd.pipeline.Enhancer`: every stage of ``inference`` stays on the MI355X.

    # run.py:12   from tester import Tester_Enhance
    from misonet_amd.tester import Tester_Enhance

Checked at construction (the reference would fail later or silently): the STFT geometry must be the one the network is
defined for (hann, 256 / 192: F = 129), ``model_sep`` / ``model`` must be misonet_amd networks.
"""
from __future__ import annotations

import os
from pathlib import Path

from . import stft as S
from .model import MISO_1, MISO_3
from .pipeline import Enhancer


class Tester_Enhance(object):
    def __init__(self, dataset, enhance_mode, dt_loader, test_loader, model_sep, model, num_ch_utilize,
                 device, num_spks, chunk_time, save_rootDir, ref_ch, cuda_flag, **ISTFT_args):
        if not isinstance(model_sep, MISO_1) or not isinstance(model, MISO_3):
            raise TypeError("Tester_Enhance needs misonet_amd.MISO_1 / misonet_amd.MISO_3 (INTEGRATION.md section 2)")
        if enhance_mode != "MISO3":                                                # tester.py:935-945: anything else runs
            raise ValueError(f"enhance_mode = {enhance_mode!r}: only 'MISO3' is implemented (the reference's other "
                             "branch, MISO2_inference, needs model.MISO_2, which is out of scope: SURVEY.md section 2)")
        self.dataset, self.enhance_mode = dataset, enhance_mode                    # tester.py:802,810
        self.dt_loader, self.test_loader = dt_loader, test_loader
        self.model_sep, self.model = model_sep, model
        self.num_ch_utilize, self.device, self.num_spks = num_ch_utilize, device, int(num_spks)
        self.fs = int(ISTFT_args["fs"])                                            # tester.py:813-816
        self.window, self.nperseg, self.noverlap = ISTFT_args["window"], int(ISTFT_args["length"]), int(ISTFT_args["overlap"])
        if self.window != "hann" or self.nperseg != S.NPERSEG or self.nperseg - self.noverlap != S.HOP:
            raise ValueError(f"the networks are defined for a hann window of {S.NPERSEG} samples with hop {S.HOP} "
                             f"(config/NN_BSS.yml:72-88); got {self.window}/{self.nperseg}/{self.noverlap}")
        if int(num_ch_utilize) != model_sep.num_ch:
            raise ValueError(f"num_ch_utilize = {num_ch_utilize} but MISO_1 was built for {model_sep.num_ch} microphones")
        self.chunk_size = int(chunk_time * self.fs)                                # tester.py:822
        self.save_rootDir, self.ref_ch, self.cuda_flag = save_rootDir, int(ref_ch), cuda_flag
        if not cuda_flag:
            raise RuntimeError("misonet_amd has no CPU path (cuda_flag must be true)")
        if isinstance(device, int):                                                # run.py passes config['gpu_num']
            model_sep.cuda(device)
            model.cuda(device)
        self._enh = Enhancer(model_sep.eval(), model.eval(), num_spks=self.num_spks, ref_ch=self.ref_ch)

    def test(self):
        """tester.py:827-844: development set into ``cv_dev93``, test set into ``test_eval92``."""
        out = {}
        for loader, sub in ((self.dt_loader, "cv_dev93"), (self.test_loader, "test_eval92")):
            save_dir = os.path.join(self.save_rootDir, sub)
            Path(save_dir).mkdir(exist_ok=True, parents=True)
            out[sub] = self.inference(loader, save_dir)
        return out

    def inference(self, data_loader, saveDir):
        """tester.py:846-975; returns {wav_name: int16 [num_spks, n_samples]} besides writing the files."""
        return self._enh.inference(data_loader, saveDir, fs=self.fs)


class Tester_Beamforming(object):
    """Drop-in for the reference's ``Tester_Beamforming`` (reference tester.py:259-449) at the harness level: same constructor
    arguments (``run.py:255-257``), same ``test()`` / ``inference(data_loader, saveDir)`` surface, same files
    (``<saveDir>/<wav_name>_{0,1}.wav`` PCM-24; ``train_si284`` or ``cv_dev93`` + ``test_eval92``).  Only MISO_1 is needed.

    ``utterance_flag = True`` (tester.py:340-449): every recording's splits are separated, taken back to the time domain,
    stitched, re-analysed as ONE long STFT and ONE MVDR per speaker is solved over the whole recording
    (:meth:`Enhancer.beamform_utterance`; golden G9 comes from this branch of the real class).
    ``utterance_flag = False`` (tester.py:452-535): one MVDR per (4 s chunk, speaker), int16 per chunk, chunks stitched
    (:meth:`Enhancer.beamform_chunks`).  That branch of the reference cannot run as committed (``self.MaxInt16`` is never
    defined -- the attribute is ``MaxINT16``, tester.py:283 -- and ``observe`` is permuted inside the per-recording loop); what is
    built here is what it evidently means: the same per-chunk arithmetic as Tester_Enhance up to the beamformer output.
    """

    def __init__(self, dataset, tr_loader, dt_loader, test_loader, model, num_ch_utilize, device, num_spks, chunk_time,
                 save_rootDir, ref_ch, cuda_flag, tr_inference_flag, utterance_flag, **ISTFT_args):
        if not isinstance(model, MISO_1):
            raise TypeError("Tester_Beamforming needs a misonet_amd.MISO_1 (INTEGRATION.md section 2)")
        self.dataset = dataset                                                     # tester.py:263-270
        self.tr_loader, self.dt_loader, self.test_loader = tr_loader, dt_loader, test_loader
        self.model, self.num_ch_utilize, self.device, self.num_spks = model, num_ch_utilize, device, int(num_spks)
        self.fs = int(ISTFT_args["fs"])                                            # tester.py:273-276
        self.window, self.nperseg, self.noverlap = ISTFT_args["window"], int(ISTFT_args["length"]), int(ISTFT_args["overlap"])
        if self.window != "hann" or self.nperseg != S.NPERSEG or self.nperseg - self.noverlap != S.HOP:
            raise ValueError(f"the network is defined for a hann window of {S.NPERSEG} samples with hop {S.HOP} "
                             f"(config/NN_BSS.yml:72-88); got {self.window}/{self.nperseg}/{self.noverlap}")
        if int(num_ch_utilize) != model.num_ch:
            raise ValueError(f"num_ch_utilize = {num_ch_utilize} but MISO_1 was built for {model.num_ch} microphones")
        self.chunk_size = int(chunk_time * self.fs)                                # tester.py:282
        self.save_rootDir, self.ref_ch, self.cuda_flag = save_rootDir, int(ref_ch), cuda_flag
        self.tr_inference_flag, self.utterance_flag = bool(tr_inference_flag), bool(utterance_flag)
        if not cuda_flag:
            raise RuntimeError("misonet_amd has no CPU path (cuda_flag must be true)")
        if isinstance(device, int):
            model.cuda(device)
        self._enh = Enhancer(model.eval(), None, num_spks=self.num_spks, ref_ch=self.ref_ch)

    def test(self):
        """tester.py:289-325: the training set into ``train_si284`` when ``tr_inference_flag``, else the development set into
        ``cv_dev93`` and the test set into ``test_eval92``."""
        sets = ((self.tr_loader, "train_si284"),) if self.tr_inference_flag else \
            ((self.dt_loader, "cv_dev93"), (self.test_loader, "test_eval92"))
        out = {}
        for loader, sub in sets:
            save_dir = os.path.join(self.save_rootDir, sub)
            Path(save_dir).mkdir(exist_ok=True, parents=True)
            out[sub] = self.inference(loader, save_dir)
        return out

    def inference(self, data_loader, saveDir, write=True):
        """tester.py:327-535.  The loader yields ``(split_observe_dict, split_clean_s0_dict, split_clean_s1_dict, gap,
        wav_name)`` with dict values complex ``[B, Ch, T, F]`` keyed '0', '1', ... (dataloader/data.py:524-597).  Returns
        {wav_name: int16 [num_spks, n_samples]} besides writing the files."""
        import numpy as np
        import torch
        os.makedirs(saveDir, exist_ok=True)
        results = {}
        dev = self._enh.device

        def h2d(x):
            """one asynchronous copy per tensor out of pinned memory instead of a synchronous ``.to(device)`` per split"""
            x = torch.as_tensor(x)
            if x.device == dev:
                return x
            if x.device.type == "cpu" and not x.is_pinned():
                x = x.contiguous().pin_memory()
            return x.to(dev, non_blocking=True)

        def finalize(rec):
            """the PREVIOUS loader item: wait for its one D2H only, stitch, write -- beside the current item's kernels"""
            rec["ev"].synchronize()
            for b, name in enumerate(rec["names"]):
                if rec["utterance"]:
                    wav = rec["host"][b].numpy().copy()                                              # [S, n]
                else:
                    pcm = rec["host"][0].numpy()                                                     # [K,B,S,n]
                    wav = np.stack([S.stitch_int16([pcm[k, b, s] for k in range(pcm.shape[0])], rec["gaps"][b])
                                    for s in range(self.num_spks)])
                results[name] = wav
                if write:
                    for s in range(self.num_spks):
                        S.write_wav_pcm24(os.path.join(saveDir, f"{name}_{s}.wav"), wav[s], self.fs)

        with torch.cuda.device(dev):
            s_out = torch.cuda.Stream(dev)
            prev = None
            for (obs_d, s0_d, s1_d, gap, wav_name) in data_loader:
                K = len(obs_d)
                obs = [h2d(obs_d[str(k)]) for k in range(K)]                                         # K x [B,M,T,F]
                clean = [torch.stack((h2d(torch.as_tensor(s0_d[str(k)])[:, self.ref_ch]),
                                      h2d(torch.as_tensor(s1_d[str(k)])[:, self.ref_ch])), dim=1) for k in range(K)]
                B = obs[0].shape[0]
                gaps = [int(gap[b]) if hasattr(gap, "__len__") else int(gap) for b in range(B)]
                names = [wav_name] * B if isinstance(wav_name, str) else list(wav_name)
                if self.utterance_flag:
                    dev_out = [self._enh.beamform_utterance([o[b] for o in obs], [c[b] for c in clean], gaps[b], to_host=False)
                               for b in range(B)]                                                    # B x int16 [S, n_b] (device)
                else:
                    dev_out = [torch.stack([S.istft_int16(self._enh.beamform_chunks(obs[k], clean[k])) for k in range(K)])]
                ev_done = torch.cuda.Event()
                ev_done.record(torch.cuda.current_stream(dev))
                rec = {"names": names, "gaps": gaps, "utterance": self.utterance_flag,
                       "host": [torch.empty(t.shape, dtype=torch.int16, pin_memory=True) for t in dev_out]}
                with torch.cuda.stream(s_out):                   # the item's D2H beside the next item's compute
                    s_out.wait_event(ev_done)
                    for h, t in zip(rec["host"], dev_out):
                        h.copy_(t, non_blocking=True)
                        t.record_stream(s_out)
                    rec["ev"] = torch.cuda.Event()
                    rec["ev"].record(s_out)
                if prev is not None:
                    finalize(prev)
                prev = rec
            if prev is not None:
                finalize(prev)
        return results
