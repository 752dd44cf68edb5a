"""Drop-in aliases: expose this package's modules under the reference's import names.

After ``install_reference_aliases()`` the reference's callers (sample/conditional_synthesis.py:9-18,
sample/edit.py, sample/synthesize.py) resolve

    from utils.model_util import create_model_and_diffusion, load_saved_model
    from model.cfg_sampler import ClassifierFreeSampleModel
    from diffusion.respace import SpacedDiffusion
    from utils import dist_util ; from utils.fixseed import fixseed

to the MI355X implementations.  Only hot-path modules are aliased; anything else (data loaders,
parsers, plotting) is still imported from wherever the caller's sys.path finds it.  A name an aliased module
does not define (e.g. ``utils.editing_util.load_fixed_dataset``, a fixture loader) falls through to the
caller's own file of that module name: ``from utils.editing_util import get_keyframes_mask, load_fixed_dataset``
(sample/conditional_synthesis.py:21) takes the first from here and the second from the reference tree.
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import sys
import types

_ALIASES = {
    "diffusion.gaussian_diffusion": "diffusion.gaussian_diffusion",
    "diffusion.respace": "diffusion.respace",
    "model.mdm": "model.mdm",
    "model.cfg_sampler": "model.cfg_sampler",
    "model.rotation2xyz": "model.rotation2xyz",
    "utils.model_util": "utils.model_util",
    "utils.editing_util": "utils.editing_util",
    "utils.dist_util": "utils.dist_util",
    "utils.fixseed": "utils.fixseed",
}
# (the device-side recover_from_ric / sample_to_xyz of data_loaders.humanml.scripts.motion_process is NOT
# aliased: the reference's module of that name also holds data-preparation code; import it explicitly)


_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_fallbacks: dict = {}


class _CallerPath:
    """``__path__`` of a synthetic parent package (``utils``, ``model``, ``diffusion``): every ``<sys.path entry>/<name>``
    directory outside this package, recomputed on each use like a namespace package's path — so the caller's tree may be put
    on sys.path before OR after install_reference_aliases(), and its non-aliased submodules (``utils.parser_util``,
    ``utils.misc``, ``model.mdm_unet`` ...) stay importable."""

    def __init__(self, name):
        self._name = name

    def _dirs(self):
        out = []
        for base in sys.path:
            cand = os.path.abspath(os.path.join(base or ".", self._name))
            if os.path.isdir(cand) and not (cand + os.sep).startswith(_PKG_DIR + os.sep) and cand not in out:
                out.append(cand)
        return out

    def __iter__(self):
        return iter(self._dirs())

    def __len__(self):
        return len(self._dirs())

    def __getitem__(self, i):
        return self._dirs()[i]

    def __contains__(self, item):
        return item in self._dirs()

    def __repr__(self):
        return f"_CallerPath({self._name!r}, {self._dirs()!r})"

    def append(self, item):   # importlib appends to a namespace path it extends; nothing to keep here
        pass


def _callers_module(ref_name: str):
    """The caller tree's own ``<ref_name>.py`` (first hit on sys.path outside this package), loaded once under a
    private name; None when there is none (this repo used standalone)."""
    if ref_name in _fallbacks:
        return _fallbacks[ref_name]
    rel = os.path.join(*ref_name.split(".")) + ".py"
    mod = None
    for base in sys.path:
        cand = os.path.abspath(os.path.join(base or ".", rel))
        if os.path.isfile(cand) and not cand.startswith(_PKG_DIR + os.sep):
            spec = importlib.util.spec_from_file_location("_condmdi_caller_." + ref_name, cand)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            break
    _fallbacks[ref_name] = mod
    return mod


def _fall_through(ref_name: str, ours):
    def __getattr__(name):   # PEP 562: only consulted for names `ours` does not define
        if name.startswith("__"):
            raise AttributeError(name)
        other = _callers_module(ref_name)
        if other is None or not hasattr(other, name):
            raise AttributeError(f"module '{ref_name}' (MI355X path: {ours.__name__}) has no attribute '{name}'")
        return getattr(other, name)
    return __getattr__


def install_reference_aliases(overwrite: bool = False):
    pkg = __name__.rsplit(".", 1)[0]
    installed = []
    for ref_name, ours in _ALIASES.items():
        if ref_name in sys.modules and not overwrite:
            continue
        mod = importlib.import_module(f"{pkg}.{ours}")
        parent_name = ref_name.split(".")[0]
        parent = sys.modules.get(parent_name)
        if parent is None:
            # the caller's own package of that name (the reference's are namespace packages: no __init__.py) if its tree
            # is already on sys.path; else a synthetic parent whose __path__ tracks sys.path
            try:
                parent = importlib.import_module(parent_name)
                if (getattr(parent, "__file__", None) or "").startswith(_PKG_DIR + os.sep):
                    raise ImportError
            except ImportError:
                parent = types.ModuleType(parent_name)
                parent.__path__ = _CallerPath(parent_name)
                sys.modules[parent_name] = parent
        if "__getattr__" not in vars(mod):
            mod.__getattr__ = _fall_through(ref_name, mod)
        sys.modules[ref_name] = mod
        setattr(parent, ref_name.split(".")[1], mod)
        installed.append(ref_name)
    return installed
