"""ORACLE (test infrastructure only -- never imported by the product path).

NumPy restatement of the reference's inference-time array processing:
  * Tester_Enhance.Apply_Beamforming          reference tester.py:1071-1136
  * get_spatial_covariance_matrix             tester.py:1138-1152
  * PhaseCorrection                           tester.py:1154-1167
  * get_mvdr_beamformer / apply_beamformer    tester.py:1211-1228
  * the 2-permutation PIT alignment used in   tester.py:1043-1065 (shift alignment)
                                              tester.py:889-915   (clean-reference alignment)

Pinned against goldens produced by the real reference (oracle/gen_golden.py).  The
reference relies on NumPy-1.x behaviours (np.complex, vector-RHS linalg.solve); the
restatement states the intended arithmetic directly.
"""
from __future__ import annotations

from itertools import permutations

import numpy as np


def spatial_covariance(x):
    """x [B,F,C,T] complex -> [B,F,C,C], R = x x^H / T  (tester.py:1147-1152)."""
    T = x.shape[-1]
    return np.einsum("...dt,...et->...de", x, x.conj()) / T


def phase_correction(w):
    """Sequential-in-f phase alignment (tester.py:1161-1167)."""
    w = w.copy()
    B, Fq, _ = w.shape
    for b in range(B):
        for f in range(1, Fq):
            z = np.sum(w[b, f, :] * w[b, f - 1, :].conj())
            w[b, f, :] = w[b, f, :] * np.exp(-1j * np.angle(z))
    return w


def mvdr_parts(source, mix, epsi=1e-6, dtype=np.complex64):
    """All intermediates of Apply_Beamforming; source/mix [B,F,M,T] complex."""
    source = np.asarray(source).astype(dtype)
    mix = np.asarray(mix).astype(dtype)
    B, Fq, M, T = source.shape
    scm_s = spatial_covariance(source)
    scm_s = 0.5 * (scm_s + np.conj(scm_s.swapaxes(-1, -2)))              # tester.py:1092
    noise = mix - source                                                  # tester.py:1095
    scm_n = spatial_covariance(noise)
    scm_n = 0.5 * (scm_n + np.conj(scm_n.swapaxes(-1, -2)))              # tester.py:1100
    vals, vecs = np.linalg.eigh(scm_s.reshape(-1, M, M))                  # tester.py:1107-1108
    idx = np.argmax(vals, axis=-1)
    steer = np.stack([vecs[i, :, idx[i]] for i in range(vals.shape[0])]).reshape(B, Fq, M)
    steer = steer / steer[:, :, :1]                                       # tester.py:1119
    nrm = np.linalg.norm(steer, axis=-1, keepdims=True)
    steer0 = steer * np.sqrt(M / nrm)                                     # tester.py:1123 (sqrt of M over the NORM)
    steer1 = phase_correction(steer0)                                     # tester.py:1128
    rn = scm_n + epsi * np.eye(M)[None, None]                             # tester.py:1086-1088,1221
    numer = np.linalg.solve(rn, steer1[..., None])[..., 0]                # tester.py:1222 (vector right-hand side)
    denom = np.einsum("...d,...d->...", steer1.conj(), numer)
    w = numer / denom[..., None]                                          # tester.py:1224
    out = np.einsum("...a,...at->...t", w.conj(), mix)                    # tester.py:1228  [B,F,T]
    return dict(scm_s=scm_s, scm_n=scm_n, steer0=steer0, steer1=steer1, w=w,
                out=np.transpose(out, (0, 2, 1)))                         # tester.py:1134 -> [B,T,F]


def apply_beamforming(source, mix, epsi=1e-6, dtype=np.complex64):
    """-> [B,T,F] complex (tester.py:1071-1136)."""
    return mvdr_parts(source, mix, epsi, dtype)["out"].astype(np.complex64)


def pit_select(ref_mag_src, cand):
    """Generic form of the reference's PIT alignment.

    ref_mag_src [B,S,T,F] complex (or magnitudes) -- the anchors (shift-0 estimates, or clean sources)
    cand        [B,S,T,F] complex                -- the speakers to be re-ordered
    Returns int array sel [B,S]: aligned speaker i is cand[:, sel[b,i]].
    dist[b,i,j] = sum_{t,f} | |ref_i| - |cand_j| |, best permutation by einsum('bij,pij->bp') + argmin
    (first minimum on ties), tester.py:1053-1065 / 902-915.
    """
    ref = np.abs(np.asarray(ref_mag_src))
    c = np.abs(np.asarray(cand))
    B, S = ref.shape[:2]
    dist = np.abs(ref[:, :, None].astype(np.float64) - c[:, None].astype(np.float64)).sum(axis=(-1, -2))   # [B,S,S]
    perms = list(permutations(range(S)))
    cost = np.stack([sum(dist[:, i, p[i]] for i in range(S)) for p in perms], axis=1)    # [B,P]
    best = np.argmin(cost, axis=1)
    return np.array([perms[k] for k in best], dtype=np.int64), dist
