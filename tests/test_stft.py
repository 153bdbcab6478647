"""The torch STFT / iSTFT used on the product side equals the reference's SciPy contract
(dataloader/data.py:505-522,542; tester.py:949-952,979-990) -- checked on CPU against the oracle's SciPy form."""
import numpy as np
import torch

from misonet_amd import stft as S
from misonet_amd.weights import synthetic_utterance
from oracle import pipeline_oracle


def test_stft_matches_scipy_contract():
    obs, _, _ = synthetic_utterance(2, 63 * 64, 3)
    ref = pipeline_oracle.stft_chunk(obs)                                     # [M,T,F]
    got = S.stft(torch.from_numpy(obs.T.copy())).numpy()
    assert got.shape == ref.shape == (3, 64, 129)
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-6
    full, _, _ = synthetic_utterance(2, 64000, 1)
    assert S.stft(torch.from_numpy(full.T.copy())).shape == (1, 1001, 129)    # 16 kHz, 4 s -> T = 1001


def test_istft_int16_matches_scipy_contract():
    r = np.random.default_rng(0)
    wav = (0.2 * r.standard_normal(63 * 64)).astype(np.float32)
    spec = pipeline_oracle.stft_chunk(wav[:, None])[0]                         # [T,F]
    ref = pipeline_oracle.istft_int16(spec)
    got = S.istft_int16(torch.from_numpy(spec)).numpy()
    assert got.shape == ref.shape == (63 * 64,)
    assert np.max(np.abs(got.astype(np.int32) - ref.astype(np.int32))) <= 1
    back = S.istft(torch.from_numpy(spec)).numpy()                             # round trip
    assert np.max(np.abs(back - wav)) < 1e-5


def test_chunking_and_stitching():
    wav = np.arange(10 * 2, dtype=np.float32).reshape(10, 2)
    chunks, gap = S.split_chunks(wav, 4)
    assert len(chunks) == 3 and gap == 2 and all(c.shape == (4, 2) for c in chunks)
    assert np.array_equal(chunks[2][2:], np.zeros((2, 2)))
    chunks1, gap1 = S.split_chunks(wav[:3], 4)                                 # shorter than one chunk (data.py:558-565)
    assert len(chunks1) == 1 and gap1 == 1
    chunks2, gap2 = S.split_chunks(wav[:8], 4)
    assert len(chunks2) == 2 and gap2 == 0
    pcs = [np.arange(4, dtype=np.int16), np.arange(4, dtype=np.int16)]
    assert S.stitch_int16(pcs, 3).tolist() == [0, 1, 2, 3, 0]


def test_pcm24_writer_roundtrip(tmp_path):
    """tester.py:972-974 writes int16 data with subtype PCM_24: samples become int16 << 8."""
    x = np.array([[0, 1], [-1, 32767], [-32768, 1234], [77, -77]], dtype=np.int16)
    p = str(tmp_path / "a_0.wav")
    S.write_wav_pcm24(p, x, 8000)
    v, fs = S.read_wav_pcm24(p)
    assert fs == 8000 and v.shape == (4, 2)
    assert np.array_equal(v, x.astype(np.int32) << 8)
    S.write_wav_pcm24(p, x[:, 0], 16000)
    v, fs = S.read_wav_pcm24(p)
    assert v.shape == (4, 1) and fs == 16000
    import pytest
    with pytest.raises(TypeError):
        S.write_wav_pcm24(p, x.astype(np.float32), 8000)


def test_pcm24_writer_bytes_match_the_reference_writer(tmp_path):
    """tester.py:972-974 writes through libsndfile: sf.write(path, int16, fs, 'PCM_24').  Expected file assembled by hand
    from the layout of the reference's OWN outputs (sample/MISO3/3_441c040w_445c040o_0.wav, read in the build container:
    'RIFF' 192214 'WAVE' 'fmt ' 16 | tag 1, 1 ch, 8000 Hz, 24000 B/s, align 3, 24 bit | 'data' 192177, file 192222 bytes
    = 44 + 64059 * 3 + 1 pad byte, low byte of every sample 0 = int16 << 8)."""
    x = np.array([1, -2, 32767], dtype=np.int16)                       # odd data length: pad byte
    p = str(tmp_path / "utt_0.wav")
    S.write_wav_pcm24(p, x, 8000)
    want = (b"RIFF" + (36 + 9 + 1).to_bytes(4, "little") + b"WAVE" +
            b"fmt " + (16).to_bytes(4, "little") + (1).to_bytes(2, "little") + (1).to_bytes(2, "little") +
            (8000).to_bytes(4, "little") + (24000).to_bytes(4, "little") + (3).to_bytes(2, "little") + (24).to_bytes(2, "little") +
            b"data" + (9).to_bytes(4, "little") +
            bytes([0x00, 0x01, 0x00]) +                                # 1 << 8
            bytes([0x00, 0xFE, 0xFF]) +                                # -2 << 8 = 0xFFFE00
            bytes([0x00, 0xFF, 0x7F]) +                                # 32767 << 8
            b"\x00")                                                   # pad to an even chunk length
    assert open(p, "rb").read() == want
    # two channels, 16 kHz, even length: no pad
    y = np.array([[-32768, 256], [0, -1]], dtype=np.int16)
    S.write_wav_pcm24(p, y, 16000)
    want2 = (b"RIFF" + (36 + 12).to_bytes(4, "little") + b"WAVEfmt " + (16).to_bytes(4, "little") +
             (1).to_bytes(2, "little") + (2).to_bytes(2, "little") + (16000).to_bytes(4, "little") +
             (96000).to_bytes(4, "little") + (6).to_bytes(2, "little") + (24).to_bytes(2, "little") +
             b"data" + (12).to_bytes(4, "little") +
             bytes([0x00, 0x00, 0x80, 0x00, 0x00, 0x01, 0x00, 0x00, 0x00, 0x00, 0xFF, 0xFF]))
    assert open(p, "rb").read() == want2
    v, fs = S.read_wav_pcm24(p)
    assert fs == 16000 and np.array_equal(v, y.astype(np.int32) << 8)
