"""MI355X-native kernels (libsparse_hip.so) and their host-side orchestration."""
from . import lib  # noqa: F401
