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
