"""Drop-in aliases: expose this package's modules under the reference's import names.

After ``install_reference_aliases()`` the reference's callers (sample/conditional_synthesis.py:9-18,
sample/edit.py, sample/synthesize.py) resolve

    from utils.model_util import create_model_and_diffusion, load_saved_model
    from model.cfg_sampler import ClassifierFreeSampleModel
    from diffusion.respace import SpacedDiffusion
    from utils import dist_util ; from utils.fixseed import fixseed

to the MI355X implementations.  Only hot-path modules are aliased; anything else (data loaders,
parsers, plotting) is still imported from wherever the caller's sys.path finds it.
"""
from __future__ import annotations

import importlib
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
            parent = types.ModuleType(parent_name)
            parent.__path__ = []  # namespace-like: lets `import utils.x` fall through for non-aliased x
            sys.modules[parent_name] = parent
        sys.modules[ref_name] = mod
        setattr(parent, ref_name.split(".")[1], mod)
        installed.append(ref_name)
    return installed
