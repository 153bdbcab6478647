"""CPU-side checks of the C-ABI shared library: it loads, exports every symbol include/misonet.h declares, and its
host-only entry points (plan construction, tensor registry, workspace sizing, argument validation) behave.
No kernels are launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from misonet_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib


def test_header_symbols_exported():
    L = _lib()
    hdr = open(os.path.join(ROOT, "include", "misonet.h")).read()
    declared = set(re.findall(r"\b(misonet_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    assert declared == set(L.SIGNATURES.keys()), declared ^ set(L.SIGNATURES.keys())
    lib = L.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.misonet_version() >= 100
    assert lib.misonet_strerror(-4).decode() == "workspace too small"


def test_product_library_has_no_experiment_switches():
    """the kernel-variant / ablation / timeline switches (MISONET_X6_*, MISONET_DMA*, MISONET_SUBBATCH, ...)
    exist only in the experiment build (`make exp`, -DMISONET_EXPERIMENTS).  The product library reads NO environment
    variable: not one MISONET_* name (nor a getenv import) is in the binary, and its ABI version says at least 450 (product modes 0 / 3 / 5 only)."""
    L = _lib()
    assert os.path.realpath(L.LIB_PATH).endswith("libmisonet_hip.so")
    blob = open(L.LIB_PATH, "rb").read()
    names = set(re.findall(rb"MISONET_[A-Z0-9_]{3,}", blob))
    assert not names, names
    assert b"getenv" not in blob
    assert L.lib().misonet_version() >= 450
    src = "".join(open(f).read() for f in __import__("glob").glob(os.path.join(ROOT, "misonet_amd", "csrc", "*.hip")))
    assert "getenv" not in src                                   # every switch goes through kernels.hpp exp_env()


def _make(in_ch=12, out_ch=4, en=(24, 32, 32, 32, 32, 64, 128), de=(128, 64, 32, 32, 32, 32, 24), nf=129, tcn_norm=0):
    L = _lib()
    cfg = L.Cfg(in_ch, out_ch, (C.c_int * 7)(*en), (C.c_int * 7)(*de), nf, tcn_norm)
    h = C.c_void_p()
    return L, L.lib().misonet_net_create(C.byref(cfg), C.byref(h)), h


def test_tensor_registry_matches_reference_keys():
    from misonet_amd import weights as W
    L, rc, h = _make()
    assert rc == 0
    lib = L.lib()
    names = [lib.misonet_net_tensor_name(h, i).decode() for i in range(lib.misonet_net_num_tensors(h))]
    spec = W.miso1_spec()
    assert names == list(spec.keys()) and len(names) == 268
    for i, k in enumerate(names):
        assert lib.misonet_net_tensor_numel(h, i) == int(np.prod(spec[k]))
    # MISO_3 geometry
    L3, rc3, h3 = _make(in_ch=16, out_ch=2)
    assert rc3 == 0
    names3 = [lib.misonet_net_tensor_name(h3, i).decode() for i in range(lib.misonet_net_num_tensors(h3))]
    assert names3 == list(W.miso3_spec().keys())
    lib.misonet_net_destroy(h)
    lib.misonet_net_destroy(h3)


def test_workspace_and_validation():
    L, rc, h = _make()
    lib = L.lib()
    # default (bf16x6, 6 bytes per element in the dense-block buffers): buffers with disjoint lifetimes share memory
    w1 = lib.misonet_net_workspace_bytes(h, 1, 1001)
    w2 = lib.misonet_net_workspace_bytes(h, 2, 1001)
    assert 200e6 < w1 < 300e6 and 1.9 < w2 / w1 < 2.1           # 0.27 GB per forward-sample at T = 1001 ...
    assert lib.misonet_net_keep_activations(h, 1) == 0
    wk = lib.misonet_net_workspace_bytes(h, 1, 1001)
    assert 500e6 < wk < 600e6                                    # ... 0.55 GB when every tap must stay readable
    assert lib.misonet_net_keep_activations(h, 0) == 0 and lib.misonet_net_workspace_bytes(h, 1, 1001) == w1
    assert lib.misonet_net_set_precision(h, 0) == 0              # exact f32: 4 bytes per element
    assert 120e6 < lib.misonet_net_workspace_bytes(h, 1, 1001) < 200e6
    assert lib.misonet_net_set_precision(h, 3) == 0
    assert lib.misonet_net_workspace_bytes(h, 0, 10) == -1
    # set_tensor validation
    v = np.zeros(10, np.float32)
    assert lib.misonet_net_set_tensor(h, b"nope", v.ctypes.data_as(C.c_void_p), 10) == L.EINVAL
    assert b"nope" in lib.misonet_last_error()
    assert lib.misonet_net_set_tensor(h, b"encoders.0.0.conv2d.bias", v.ctypes.data_as(C.c_void_p), 10) == L.EINVAL
    # commit before all tensors are set -> state error naming the missing key
    assert lib.misonet_net_commit(h) == L.ESTATE
    assert b"missing state_dict key" in lib.misonet_last_error()
    lib.misonet_net_destroy(h)
    # unsupported geometries are rejected at create time
    assert _make(nf=257)[1] == L.EINVAL                          # only F = 129 works
    assert _make(en=(24, 32, 32, 32, 32, 64, 256))[1] == L.EINVAL
    assert _make(de=(128, 64, 32, 32, 32, 32, 32))[1] == L.EINVAL  # skip-concat mismatch
    assert _make(in_ch=13)[1] == L.EINVAL
    assert lib.misonet_mvdr_workspace_bytes(2, 129, 6) > 0


def test_host_mirror_requires_library_and_validates():
    import misonet_amd as mz
    from misonet_amd import weights as W
    with pytest.raises(ValueError):
        mz.MISO_1(2, 6, 8, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN")
    # norm_type variants are accepted since round 4 (the outer norms of the TemporalBlocks, model.py:530,535,570-581):
    # "BN" = BatchNorm1d, whose state_dict carries weight / bias / running statistics and the int64 counter
    mb = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "BN")
    sdb = W.make_state_dict(W.miso1_spec(norm_type="BN"), 3)
    mb.load_state_dict(sdb)
    assert list(mb.state_dict().keys()) == list(sdb.keys())
    assert mb.state_dict()["TCN.temporal_conv_net.0.0.net.0.num_batches_tracked"].dtype == torch.int64
    with pytest.raises(RuntimeError):                           # an "IN" checkpoint does not fit a "BN" network
        mb.load_state_dict(W.make_state_dict(W.miso1_spec(), 0))
    with pytest.raises(TypeError):
        mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), None)
    en = list(W.DEFAULT_EN_CH)
    m = mz.MISO_1(2, 6, 7, en, list(W.DEFAULT_DE_CH), "IN")
    assert en == list(W.DEFAULT_EN_CH)                           # no in-place mutation (model.py:16-17 mutates)
    sd = W.make_state_dict(W.miso1_spec(), 0)
    m.load_state_dict(sd)
    back = m.state_dict()
    assert list(back.keys()) == list(sd.keys())
    assert np.array_equal(back["decoders.6.1.deconv2d.bias"].numpy(), sd["decoders.6.1.deconv2d.bias"])
    bad = dict(sd)
    bad["encoders.0.0.conv2d.weight"] = np.zeros((24, 16, 3, 3), np.float32)
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad)
    with pytest.raises(RuntimeError):
        m.load_state_dict({k: v for k, v in list(sd.items())[:10]})
    assert "MISO_1" in repr(m) and "2587384" in repr(m)
    with pytest.raises(NotImplementedError):
        m.train()


def test_product_does_not_import_oracle():
    """The product path must never route through the oracle or a CPU fallback."""
    pkg = os.path.join(ROOT, "misonet_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+\.*oracle", src, flags=re.M), fn
            assert "import oracle" not in src and "from oracle" not in src, fn


def test_precision_switch_host_side():
    L, rc, h = _make()
    lib = L.lib()
    assert lib.misonet_net_get_precision(h) == 3                 # bf16x6 (fp32-faithful, the bench's mode) by default
    assert lib.misonet_net_set_precision(h, 0) == 0 and lib.misonet_net_get_precision(h) == 0   # f32
    assert lib.misonet_net_set_precision(h, 5) == 0 and lib.misonet_net_get_precision(h) == 5   # f32w
    # the product library has exactly three arithmetic modes; bf16x3 (1, 2), f16x3 (4), bf16x6w (6) are experiment-build modes
    for alt in (1, 2, 4, 6):
        assert lib.misonet_net_set_precision(h, alt) == L.EINVAL and lib.misonet_net_get_precision(h) == 5, alt
    assert lib.misonet_net_set_precision(h, 7) == L.EINVAL and lib.misonet_net_set_precision(h, -1) == L.EINVAL
    lib.misonet_net_destroy(h)
    import misonet_amd as mz
    from misonet_amd import weights as W
    m = mz.MISO_3(1, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN")
    assert m.precision == "bf16x6"
    m.set_precision("f32w")
    assert m.precision == "f32w"
    with pytest.raises(ValueError, match="experiment build"):
        m.set_precision("bf16x3")
    assert m.precision == "f32w"
    with pytest.raises(ValueError):
        m.set_precision("fp8")


def test_reference_checkpoint_format_roundtrip(tmp_path):
    """run.py:139-151 loads torch.load(path)['model_state_dict'] (format written by trainer.py:340-347)."""
    import torch
    import misonet_amd as mz
    from misonet_amd import weights as W
    sd = {k: torch.from_numpy(v) for k, v in W.make_state_dict(W.miso1_spec(), 3).items()}
    path = str(tmp_path / "ckpt.pt")
    torch.save({"model_state_dict": sd, "optimizer": {}, "epoch": 7, "tr_avg_loss": 0.1, "val_avg_loss": 0.2}, path)
    package = torch.load(path, map_location="cpu")
    m = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN")
    m.load_state_dict(package["model_state_dict"])
    back = m.state_dict()
    for k in sd:
        assert torch.equal(back[k], sd[k])


@pytest.mark.parametrize("prec", [0, 3, 5])
def test_workspace_plan_never_overlaps_live_buffers(prec):
    """The lifetime-shared activation arena: for every arithmetic mode and several utterance lengths, two
    buffers that are alive at the same step of a forward never share a byte, the skip buffers D[i] live from their encoder
    to their decoder (reference model.py:84-99), and keep_activations gives every buffer its own memory."""
    L, rc, h = _make()
    lib = L.lib()
    assert rc == 0 and lib.misonet_net_set_precision(h, prec) == 0
    for keep in (0, 1):
        assert lib.misonet_net_keep_activations(h, keep) == 0
        for T in (5, 501, 1001, 2500):
            n = 32
            off = (C.c_longlong * n)(); size = (C.c_longlong * n)(); t0 = (C.c_int * n)(); t1 = (C.c_int * n)()
            k = lib.misonet_net_buffer_plan(h, T, n, off, size, t0, t1)
            assert k == 19                                     # IN, E0-E4, D0-D6, X2-X6, OUT
            rects = [(off[i], off[i] + size[i], t0[i], t1[i]) for i in range(k)]
            assert all(a % 256 == 0 and b > a and 0 <= s <= e <= 16 for a, b, s, e in rects)
            for i in range(k):
                for j in range(i + 1, k):
                    a0, a1, s0, e0 = rects[i]; b0, b1, s1, e1 = rects[j]
                    alive_together = s0 <= e1 and s1 <= e0
                    if alive_together:
                        assert a1 <= b0 or b1 <= a0, (prec, keep, T, i, j, rects[i], rects[j])
            if keep:
                assert all(s == 0 and e == 16 for _, _, s, e in rects)
            else:
                # D[i] (plan index 6 + i): alive from encoder 6 - i (step 7 - i) to decoder i (step 9 + i)
                for i in range(7):
                    assert (rects[6 + i][2], rects[6 + i][3]) == (7 - i, 9 + i)
                block = max(b for _, b, _, _ in rects)
                assert block < sum(b - a for a, b, _, _ in rects)      # something is shared
    lib.misonet_net_destroy(h)


@pytest.mark.parametrize("nt,kind,extra", [("gLN", 1, 56), ("cLN", 2, 56), ("BN", 3, 112)])
def test_tensor_registry_norm_type_variants(nt, kind, extra):
    """norm_type of the constructors (model.py:9, 283) = the outer norms of the 14 TemporalBlocks (model.py:530,535,570-581):
    the library's tensor list follows the reference's state_dict order for every variant (the float tensors of it; the int64
    num_batches_tracked of BatchNorm1d stays on the Python side)."""
    from misonet_amd import weights as W
    L, rc, h = _make(tcn_norm=kind)
    assert rc == 0
    lib = L.lib()
    names = [lib.misonet_net_tensor_name(h, i).decode() for i in range(lib.misonet_net_num_tensors(h))]
    spec = W.miso1_spec(norm_type=nt)
    want = [k for k in spec if not k.endswith(".num_batches_tracked")]
    assert names == want and len(names) == 268 + extra
    assert W.norm_kind(nt) == kind and W.norm_kind("IN") == 0 and W.norm_kind("whatever") == 3
    lib.misonet_net_destroy(h)
    assert _make(tcn_norm=4)[1] == L.EINVAL and _make(tcn_norm=-1)[1] == L.EINVAL


def test_library_load_pulls_torch_in_first():
    """Loading libmisonet_hip.so before torch leaves the process with two HIP runtimes (torch ships its own libamdhip64) and the
    second one to initialise reports "no ROCm-capable device" -- seen when __graft_entry__.build() and smoke() ran in one
    process.  _lib.lib() therefore imports torch first; checked in a fresh interpreter."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from misonet_amd import _lib; assert 'torch' not in sys.modules; "
            "_lib.lib(); assert 'torch' in sys.modules; print('ok')" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-500:]
