"""Importable alias for the ``imp-release_amd/`` package directory (a hyphen is not a valid
Python identifier).  All code lives in ``imp-release_amd/``; this module only redirects the
package search path there."""
import os as _os

_real = _os.path.abspath(_os.path.join(_os.path.dirname(__file__), '..', 'imp-release_amd'))
__path__.insert(0, _real)
with open(_os.path.join(_real, '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(_real, '__init__.py'), 'exec'))
del _f
