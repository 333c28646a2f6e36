"""Importable alias of the `celo-bls-snark-rs_amd/` package directory (a hyphen is not a legal
Python identifier, so this shim points its __path__ at the real directory and runs its __init__)."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "celo-bls-snark-rs_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
