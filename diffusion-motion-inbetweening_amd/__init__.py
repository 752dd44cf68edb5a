"""MI355X-native sampling path for CondMDI (setarehc/diffusion-motion-inbetweening).

The directory name carries a hyphen (it is fixed by the project layout), so import the package with

    import importlib
    condmdi = importlib.import_module("diffusion-motion-inbetweening_amd")

Sub-packages mirror the reference's module layout for the hot path only:
``diffusion.{gaussian_diffusion,respace}``, ``model.{mdm,cfg_sampler,rotation2xyz}``,
``utils.{model_util,editing_util,dist_util,fixseed}``.  ``compat.install_reference_aliases()``
registers them under the reference's top-level names so its sample scripts import them unchanged.
"""
from . import _native  # noqa: F401
from .build import build_native  # noqa: F401

__all__ = ["_native", "build_native"]
