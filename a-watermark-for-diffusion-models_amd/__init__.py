"""MI355X-native Gaussian-Shading watermark hot path (embed / DDIM step / extract) behind the reference's own
Python interface.  The directory name is fixed by the build contract and is not an importable identifier; import it
through the `gswm_amd` loader module at the repo root:

    import gswm_amd
    from gswm_amd import gs_insert, extract          # drop-in twins of the reference's modules
    from gswm_amd import codec                        # batch-first device API over the C ABI (include/gswm.h)
"""
from . import _native  # noqa: F401
from . import codec  # noqa: F401

__all__ = ["codec", "gs_insert", "extract", "comfy", "ddim", "dist"]


def __getattr__(name):  # lazy sub-modules (keep `import gswm_amd` light)
    if name in __all__:
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
