"""vdetlib -- the reference's import root, served by the MI355X build.

T-CNN and the reference's own modules import ``vdetlib.utils.protocol``, ``vdetlib.utils.cython_nms``,
``vdetlib.vdet.track`` ... (``/root/reference/vdet/track.py:13``, ``vdet/video_det.py:11``, ``vdet/image_det.py:9``).
With this package on ``sys.path`` (it sits next to ``vdetlib_amd``) those imports resolve, unchanged, to the modules
of ``vdetlib_amd``: ``vdetlib.X.Y`` IS ``vdetlib_amd.X.Y`` (one module object under two names, so monkey-patching
either -- e.g. ``video_det.imread = ...`` -- is seen through both).
"""
import importlib
import importlib.abc
import importlib.util
import sys

import vdetlib_amd as _impl

__version__ = _impl.__version__
_PREFIX, _REAL = __name__ + ".", _impl.__name__ + "."


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """``vdetlib.a.b`` -> the module object of ``vdetlib_amd.a.b``"""

    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_PREFIX):
            return None
        real = _REAL + fullname[len(_PREFIX):]
        try:
            spec = importlib.util.find_spec(real)
        except (ImportError, ValueError):
            return None
        if spec is None:
            return None
        return importlib.util.spec_from_loader(fullname, self, is_package=spec.submodule_search_locations is not None)

    def create_module(self, spec):
        module = importlib.import_module(_REAL + spec.name[len(_PREFIX):])
        # importlib assigns module.__spec__ = <alias spec> after this returns; the real module must keep its own spec
        # (importlib.reload, __package__ == __spec__.parent, the package's submodule_search_locations): restored below
        spec.loader_state = getattr(module, "__spec__", None)
        return module

    def exec_module(self, module):
        alias = getattr(module, "__spec__", None)
        real = getattr(alias, "loader_state", None)
        if real is not None:
            module.__spec__ = real
        return None


sys.meta_path.insert(0, _AliasFinder())


def __getattr__(name):
    # ``import vdetlib; vdetlib.utils`` and ``from vdetlib import utils``
    try:
        return importlib.import_module(_PREFIX + name)
    except ModuleNotFoundError as e:
        if e.name in (_PREFIX + name, _REAL + name):     # only "there is no such submodule"; a missing dependency propagates
            raise AttributeError(name)
        raise
