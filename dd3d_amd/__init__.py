"""dd3d_amd -- MI355X-native gradient-matching distillation step of kghandour/dd3d."""
from . import _lib  # noqa: F401  (fails loudly when libdd3d_hip.so is missing)
from .sparse import CoordinateManager, SparseTensor, TensorField  # noqa: F401

__version__ = "0.1.0"
