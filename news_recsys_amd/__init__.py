"""news_recsys_amd -- MI355X-native embedding / pooling / feature-interaction path for
News_Recsys-style rankers.  Python host code over a C-ABI HIP library."""
__version__ = "0.1.0"

from . import _lib  # noqa: F401


def lib_available() -> bool:
    """True when the in-tree HIP library has been built (it is required: no CPU fallback)."""
    return _lib.is_available()
