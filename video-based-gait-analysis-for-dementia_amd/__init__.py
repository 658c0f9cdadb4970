"""MI355X-native implementation of MAX-GRNet's per-frame pose/mesh inference path."""
from . import netspec, synth  # noqa: F401
