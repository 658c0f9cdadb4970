"""MI355X-native implementation of MAX-GRNet's per-frame pose/mesh inference path."""
import importlib as _importlib

from . import accounting, netspec, synth  # noqa: F401


def __getattr__(name):
    # the model classes need torch + the HIP library; keep `import pkg` light for CPU-only tools
    if name in ("GRNet", "build_synthetic_model"):
        return getattr(_importlib.import_module(__name__ + ".grnet"), name)
    if name in ("_lib", "grnet", "harness", "pipeline"):
        return _importlib.import_module(__name__ + "." + name)
    raise AttributeError(name)
