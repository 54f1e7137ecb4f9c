"""Importable name of the package that lives in `heatnet-pub_amd/` (a hyphen cannot be imported).

`import heatnet_amd` == the code under /heatnet-pub_amd; submodules resolve from that directory.
"""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "heatnet-pub_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
