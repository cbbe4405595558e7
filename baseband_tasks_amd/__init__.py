"""Import shim: the package sources live in ``baseband-tasks_amd/`` (the
directory name the project layout prescribes, which is not a valid Python
identifier).  ``import baseband_tasks_amd`` resolves here and continues
there."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      'baseband-tasks_amd')
__path__.insert(0, _real)
__file__ = _os.path.join(_real, '__init__.py')
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, 'exec'))
del _f, _os
