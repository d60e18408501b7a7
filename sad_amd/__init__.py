"""Importable alias for the package directory
``non-parallel-text-style-transfer-using-self-attn-discriminator_amd/`` (a hyphenated name cannot be
an ``import`` target).  ``import sad_amd`` == that package; submodules resolve from its directory."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "non-parallel-text-style-transfer-using-self-attn-discriminator_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
