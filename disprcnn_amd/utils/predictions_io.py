"""`predictions.pth` -- the on-disk / wire format of the reference's inference results.

Reference: disprcnn/engine/inference.py:125-133 -- `torch.save({'left': [BoxList, ...], 'right': [BoxList, ...]}, 'predictions.pth')`
(a plain list of BoxLists for the mono detectors), re-read with `torch.load(..., 'cpu')` by engine/inference.py:101-103 and
data/datasets/kitti_car.py:51-57,102-106.  A BoxList is pickled by class path + instance dict: `bbox [R,4] f32`, `size (w, h)`, `mode`,
`extra_fields {scores [R], labels [R], mask [R,1,28,28], disparity [R,224,224], ...}`, `PixelWise_map {name: DisparityMap}`,
`mask_thresh` (disprcnn/structures/bounding_box.py:20-41).

The file written here names the REFERENCE's classes (`disprcnn.structures.bounding_box.BoxList`, `disprcnn.structures.disparity.DisparityMap`)
so the reference's own `torch.load` reads it unchanged; the reader maps those class paths onto this package's classes and refuses every
other global that is not part of torch's tensor serialisation -- a predictions file is data, not code."""
import pickle
import sys
import types

import torch

from ..structures.bounding_box import BoxList
from ..structures.disparity import DisparityMap

_REF = {BoxList: ("disprcnn.structures.bounding_box", "BoxList"), DisparityMap: ("disprcnn.structures.disparity", "DisparityMap")}
_OURS = {v: k for k, v in _REF.items()}
# what a tensor-bearing pickle legitimately references (torch.save's own rebuild helpers and containers)
_ALLOWED = {("collections", "OrderedDict"), ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_parameter"),
            ("torch", "Size"), ("torch", "device"), ("copyreg", "_reconstructor"), ("copy_reg", "_reconstructor"), ("builtins", "object"),
            ("__builtin__", "object"), ("builtins", "tuple"), ("builtins", "list"), ("builtins", "dict"), ("builtins", "set")}
_STORAGES = {"FloatStorage", "DoubleStorage", "HalfStorage", "BFloat16Storage", "LongStorage", "IntStorage", "ShortStorage", "CharStorage",
             "ByteStorage", "BoolStorage", "UntypedStorage"}


def _stub(module, name):
    """A placeholder class that pickles as `module.name` (the module is registered for the duration of the save only)."""
    return type(name, (object,), {"__module__": module, "__qualname__": name})


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _OURS:
            return _OURS[(module, name)]
        if (module, name) in _ALLOWED:
            return super().find_class(module, name)
        if module in ("torch", "torch.storage") and name in _STORAGES:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"predictions file references {module}.{name}: not part of the predictions format")


class _PickleModule:
    """The `pickle_module` interface torch.save / torch.load expect."""
    __name__ = "disprcnn_amd.utils.predictions_io"
    Pickler, Unpickler = pickle.Pickler, _Unpickler
    HIGHEST_PROTOCOL, DEFAULT_PROTOCOL = pickle.HIGHEST_PROTOCOL, pickle.DEFAULT_PROTOCOL
    PicklingError, UnpicklingError = pickle.PicklingError, pickle.UnpicklingError

    @staticmethod
    def dump(obj, f, protocol=2):
        pickle.Pickler(f, protocol=protocol).dump(obj)

    @staticmethod
    def load(f, **kw):
        return _Unpickler(f, **kw).load()


def _export(obj, stubs):
    """The tree with detached CPU copies of every tensor (the reference gathers predictions to the CPU before saving, inference.py:44-50)
    and every BoxList / DisparityMap replaced by an instance of the stand-in class carrying the reference's class path and the same
    instance dict: pickling it emits byte for byte what pickling the reference's object emits (NEWOBJ + BUILD)."""
    if torch.is_tensor(obj):
        return obj.detach().cpu()
    ref = _REF.get(type(obj))
    if ref is not None:
        out = object.__new__(stubs[ref])
        state = {k: _export(v, stubs) for k, v in obj.__dict__.items()}
        if isinstance(obj, BoxList):
            state.setdefault("mask_thresh", 0.5)                       # the reference's constructor sets it (bounding_box.py:40)
            state["size"] = tuple(int(v) for v in obj.size)
        out.__dict__.update(state)
        return out
    if isinstance(obj, dict):
        return type(obj)((k, _export(v, stubs)) for k, v in obj.items())
    if isinstance(obj, (list, tuple)):
        return type(obj)(_export(v, stubs) for v in obj)
    return obj


def save_predictions(predictions, path):
    """Write `predictions` -- {'left': [BoxList], 'right': [BoxList]} or a list of BoxLists -- in the reference's format."""
    saved, stubs = {}, {}
    try:
        # pickle writes a class by its dotted path and checks that the path resolves to that very class: for the duration of the save the
        # reference's module paths hold stand-in modules with stand-in classes.  Whatever sys.modules had under those names (nothing; the
        # real reference, /root/reference on the path; or the `disprcnn` alias package of this repo) is put back afterwards.
        for module, name in _REF.values():
            parts = module.split(".")
            for i in range(1, len(parts) + 1):                         # parent packages too: pickle imports the dotted path
                m = ".".join(parts[:i])
                if m not in saved:
                    saved[m] = sys.modules.get(m)
                    sys.modules[m] = types.ModuleType(m)
            cls = _stub(module, name)
            setattr(sys.modules[module], name, cls)
            stubs[(module, name)] = cls
        torch.save(_export(predictions, stubs), path, pickle_module=_PickleModule, pickle_protocol=2)
    finally:
        for m, old in saved.items():
            if old is None:
                sys.modules.pop(m, None)
            else:
                sys.modules[m] = old


def load_predictions(path, map_location="cpu"):
    """Read a predictions file written by the reference or by save_predictions into this package's BoxList / DisparityMap."""
    out = torch.load(path, map_location=map_location, pickle_module=_PickleModule, weights_only=False)
    for bl in _boxlists(out):
        bl.size = tuple(bl.size)
        bl.__dict__.setdefault("extra_fields", {})
        bl.__dict__.setdefault("PixelWise_map", {})
    return out


def _boxlists(obj):
    if isinstance(obj, BoxList):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _boxlists(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _boxlists(v)
