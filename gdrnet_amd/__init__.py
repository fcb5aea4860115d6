"""Import alias for the product package.

The product lives in the directory ``gdr-net_amd/`` (the name the build contract
asks for).  A hyphen is not a legal Python identifier, so this tiny package gives it
an importable name: ``import gdrnet_amd`` executes ``gdr-net_amd/__init__.py`` and
resolves every submodule (``gdrnet_amd.GDRN``, ``gdrnet_amd.engine`` ...) from that
directory.
"""
import os as _os

_here = _os.path.dirname(_os.path.abspath(__file__))
_real = _os.path.join(_os.path.dirname(_here), "gdr-net_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
