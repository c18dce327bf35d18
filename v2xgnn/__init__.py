"""Importable alias of the product package.

The package directory required by the build contract is
`globecom2020-resourceallocationgnn_amd/`, whose name is not a valid Python identifier;
`import v2xgnn` loads that directory as the package `v2xgnn` (all code lives there).
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "globecom2020-resourceallocationgnn_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
