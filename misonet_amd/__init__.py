"""misonet_amd -- MI355X-native MISO1 -> MVDR -> MISO3 inference path (hand-written HIP behind a C ABI).

Host-side mirror of the reference call surface:
  MISO_1, MISO_3            (reference model.py:8-111, 282-395)
  Apply_Beamforming         (reference tester.py:1071-1136)
  Enhancer                  (reference tester.py:846-975, the Tester_Enhance hot loop, kept on-device)
  tester.Tester_Enhance     (reference tester.py:798-975: the harness class itself, same constructor / test / inference)
The compute lives in csrc/ (libmisonet_hip.so); importing a compute symbol without the built
library raises -- there is no CPU fallback.
"""
__all__ = ["MISO_1", "MISO_3", "Apply_Beamforming", "Enhancer", "weights"]


def __getattr__(name):
    if name in ("MISO_1", "MISO_3"):
        from . import model
        return getattr(model, name)
    if name == "Apply_Beamforming":
        from .beamform import Apply_Beamforming
        return Apply_Beamforming
    if name == "Enhancer":
        from .pipeline import Enhancer
        return Enhancer
    if name == "weights":
        import importlib
        return importlib.import_module(".weights", __name__)
    raise AttributeError(name)
