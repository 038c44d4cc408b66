"""imp-release_amd: MI355X-native (gfx950) implementation of the IMP / EIMP matching hot path.

Python host code mirroring the reference's ``GM`` / ``DGNNS`` / ``AdaGMN`` surface
(nets/gm.py, nets/gms.py, nets/adgm.py) over hand-written HIP kernels reached through the C-ABI
library ``csrc/libimp_hip.so`` (declared in ``include/imp_hip.h``).  There is no CPU fallback:
every compute entry point raises if the HIP library is missing.

Import as ``imp_release_amd`` (alias package next to this directory).
"""
__version__ = '0.1.0'

from . import synthetic  # noqa: F401,E402  (numpy only)


def __getattr__(name):
    # torch-dependent modules are imported lazily so that `synthetic` stays usable without them
    if name in ('GM', 'DGNNS', 'AdaGMN', 'AttentionHandle', 'normalize_keypoints'):
        from . import modules
        return getattr(modules, name)
    if name in ('matching_iterative', 'matching_iterative_uncertainty'):
        from . import matching
        return getattr(matching, name)
    if name in ('modules', 'matching', 'dist', '_lib', 'eval_loop', 'metrics', 'pipeline', 'data', 'pose'):
        import importlib
        return importlib.import_module('.' + name, __name__)
    raise AttributeError(name)
