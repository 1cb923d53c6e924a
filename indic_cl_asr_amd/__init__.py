"""MI355X-native Conformer hybrid RNNT-CTC + continual-learning training path."""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
