"""Import the REAL reference (/root/reference) on CPU — available in the build container only.

Used by tests/golden/make_golden.py to generate the committed golden vectors and by the
`reference`-marked tests to pin the numpy oracle.  Nothing here runs on the GPU box (the reference
does not exist there).  Shim list = SURVEY.md §8c:
  1. np.float / np.int aliases (numpy >= 1.24 removed them; quaternion.py:13, motion_process.py:72)
  2. stub `clip` module whose text tower returns embeddings registered by the test
  3. stub `smplx`, identity Rotation2xyz (SMPL files absent)
  4. stub `spacy` (pulled in by utils/model_util.py -> data_loaders/humanml/data/dataset.py:9)
  5. model.keyframe_conditioned = False before ClassifierFreeSampleModel(model)
  6. data = SimpleNamespace(dataset=SimpleNamespace())
  7. noise injection: torch.randn / torch.randn_like patched to replay a prepared stream
"""
from __future__ import annotations

import contextlib
import os
import sys
import types
from types import SimpleNamespace

REFERENCE_ROOT = os.environ.get("CONDMDI_REFERENCE", "/root/reference")
_state = {"text_embed": None}


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "diffusion"))


def set_text_embedding(t):
    """Tensor [B, 512] the stub CLIP returns from encode_text (our boundary is enc_text)."""
    _state["text_embed"] = t


def _install_stubs():
    import numpy as np
    import torch

    if not hasattr(np, "float"):
        np.float = float  # type: ignore[attr-defined]
    if not hasattr(np, "int"):
        np.int = int  # type: ignore[attr-defined]
    if not hasattr(np, "bool"):
        np.bool = bool  # type: ignore[attr-defined]

    clip = types.ModuleType("clip")

    class _FakeClip(torch.nn.Module):
        def encode_text(self, tokens):
            emb = _state["text_embed"]
            assert emb is not None, "call ref_shims.set_text_embedding first"
            return emb.to(tokens.device)

    clip.load = lambda *a, **k: (_FakeClip(), None)
    clip.tokenize = lambda texts, context_length=77, truncate=False: torch.zeros(
        (len(texts), context_length), dtype=torch.long)
    clip.model = types.ModuleType("clip.model")
    clip.model.convert_weights = lambda m: None
    sys.modules.setdefault("clip", clip)
    sys.modules.setdefault("clip.model", clip.model)

    for name in ("smplx", "smplx.lbs", "spacy"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            sys.modules[name] = m
    sys.modules["smplx"].SMPLLayer = type("SMPLLayer", (torch.nn.Module,), {})
    sys.modules["smplx.lbs"].vertices2joints = lambda *a, **k: None
    sys.modules["smplx"].lbs = sys.modules["smplx.lbs"]


def import_reference():
    """Returns a namespace of the reference's hot-path modules (imported under their own names)."""
    if not available():
        raise RuntimeError(f"reference checkout not found at {REFERENCE_ROOT}")
    _install_stubs()
    clash = [n for n in ("diffusion", "model", "utils", "data_loaders")
             if n in sys.modules and REFERENCE_ROOT not in (getattr(sys.modules[n], "__file__", "") or "")
             and not hasattr(sys.modules[n], "__path__")]
    if clash:
        raise RuntimeError(f"modules {clash} already imported from elsewhere")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import torch
    import model.mdm as ref_mdm  # noqa

    class _IdentityRot2xyz:
        def __init__(self, *a, **k):
            self.smpl_model = torch.nn.Identity()

        def __call__(self, x, *a, **k):
            return x

    ref_mdm.Rotation2xyz = _IdentityRot2xyz
    import diffusion.gaussian_diffusion as ref_gd  # noqa
    import diffusion.respace as ref_respace  # noqa
    import model.cfg_sampler as ref_cfg  # noqa
    import utils.editing_util as ref_edit  # noqa
    import utils.model_util as ref_model_util  # noqa
    return SimpleNamespace(gd=ref_gd, respace=ref_respace, mdm=ref_mdm, cfg=ref_cfg,
                           editing=ref_edit, model_util=ref_model_util)


def default_args(**over):
    """The attributes create_model_and_diffusion reads, with the parser defaults
    (utils/parser_util.py:10-120)."""
    a = dict(arch='trans_enc', emb_trans_dec=False, layers=8, latent_dim=512, ff_size=1024,
             dim_mults=(2, 2, 2, 2), unet_adagn=True, unet_zero=True, out_mult=1, cond_mask_prob=.1,
             keyframe_mask_prob=.1, lambda_rcxyz=0., lambda_vel=0., lambda_fc=0.,
             unconstrained=False, keyframe_conditioned=False,
             keyframe_selection_scheme='random_frames', zero_keyframe_loss=False,
             dataset='humanml', abs_3d=False, traj_only=False, xz_only=False,
             use_random_proj=False, drop_redundant=False, noise_schedule='cosine',
             diffusion_steps=1000, sigma_small=True, predict_xstart=True, use_ddim=False,
             clip_range=6.0, use_fp16=False, apply_zero_mask=False, traj_extra_weight=1.,
             time_weighted_loss=False, train_x0_as_eps=False)
    a.update(over)
    return SimpleNamespace(**a)


def make_reference_model(ref, args, state_dict_torch, cfg=False):
    """Reference MDM (+ optional CFG wrapper) on CPU, eval mode, weights from `state_dict_torch`."""
    data = SimpleNamespace(dataset=SimpleNamespace())
    model, diffusion = ref.model_util.create_model_and_diffusion(args, data)
    missing, unexpected = model.load_state_dict(state_dict_torch, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("clip_model.") for k in missing), missing
    model.eval()
    if cfg:
        model.keyframe_conditioned = False
        model = ref.cfg.ClassifierFreeSampleModel(model)
        model.eval()
    return model, diffusion


@contextlib.contextmanager
def injected_noise(stream):
    """Replay `stream` (iterable of tensors) through torch.randn / torch.randn_like."""
    import torch
    it = iter(stream)
    real_randn, real_like = torch.randn, torch.randn_like

    def fake_randn(*shape, **kw):
        t = next(it)
        shp = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else tuple(shape)
        assert tuple(t.shape) == shp, (t.shape, shp)
        return t.clone()

    def fake_like(x, **kw):
        t = next(it)
        assert t.shape == x.shape
        return t.clone().to(x.device)

    torch.randn, torch.randn_like = fake_randn, fake_like
    try:
        yield
    finally:
        torch.randn, torch.randn_like = real_randn, real_like
