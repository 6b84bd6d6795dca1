"""Import shim: the product package lives in ``acousticswarms-speech_amd/`` (the
directory name the build contract asks for, which is not a valid Python
identifier).  This shim makes it importable as ``acousticswarms_speech_amd``."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "acousticswarms-speech_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
