"""`disprcnn` import names for the MI355X implementation (SURVEY 7 / 8b "Python operator API to preserve").

The reference's drivers import `disprcnn.modeling.*`, `disprcnn.layers.*`, `disprcnn.structures.*`, `disprcnn.utils.*`
(tools/train_net.py:10-20, tools/test_net.py, train_idispnet_fa.py:53-75).  This package makes every such name resolve to the module of the
same relative name under `disprcnn_amd` -- the SAME module object, not a copy -- so those drivers run on the HIP path with their
imports unchanged:

    from disprcnn.modeling.psmnet.stackhourglass import PSMNet        # -> disprcnn_amd.modeling.psmnet.stackhourglass.PSMNet
    from disprcnn.layers import ROIAlign, nms                          # -> disprcnn_amd.layers
    from disprcnn.modeling.detector import build_detection_model
    from disprcnn.utils.loss_utils import PSMLoss

A name `disprcnn_amd` does not provide raises ImportError, as a missing reference module would.
"""
import importlib
import importlib.abc
import importlib.util
import sys

import disprcnn_amd as _impl

_PREFIX = __name__ + "."
_TARGET = _impl.__name__ + "."


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_PREFIX):
            return None
        real = _TARGET + fullname[len(_PREFIX):]
        try:
            if importlib.util.find_spec(real) is None:
                return None
        except (ImportError, ValueError):
            return None
        return importlib.util.spec_from_loader(fullname, self, is_package=True)

    def create_module(self, spec):
        # the implementation's own module object.  importlib's module_from_spec stamps the alias spec (and, for `is_package`, an empty
        # __path__) onto whatever create_module returns: what the real module had is put back in exec_module, so that its relative imports
        # (`from .. import _lib`) keep resolving through ITS package
        real = importlib.import_module(_TARGET + spec.name[len(_PREFIX):])
        _SAVED[spec.name] = (real, real.__spec__, getattr(real, "__package__", None), hasattr(real, "__path__"))
        return real

    def exec_module(self, module):
        alias = getattr(module, "__spec__", None)
        saved = _SAVED.pop(getattr(alias, "name", None), None)
        if saved is None or saved[0] is not module:
            return
        _, spec, package, had_path = saved
        module.__spec__ = spec
        if package is not None:
            module.__package__ = package
        if not had_path and hasattr(module, "__path__"):
            del module.__path__


_SAVED = {}

if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())

__path__ = []          # a namespace of aliases: sub-modules come from the finder above
