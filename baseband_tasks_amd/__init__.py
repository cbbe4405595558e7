"""Import shim: the package sources live in ``baseband-tasks_amd/`` (the
directory name the project layout prescribes, which is not a valid Python
identifier).  ``import baseband_tasks_amd`` loads that directory as this
package with importlib's documented recipe for importing a source file
directly: a module spec whose origin is ``baseband-tasks_amd/__init__.py`` and
whose submodule search path is that directory replaces this stub in
``sys.modules``."""
import importlib.util as _util
import os as _os
import sys as _sys

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      'baseband-tasks_amd')
_spec = _util.spec_from_file_location(__name__, _os.path.join(_real, '__init__.py'),
                                      submodule_search_locations=[_real])
_module = _util.module_from_spec(_spec)
_sys.modules[__name__] = _module
_spec.loader.exec_module(_module)
