"""MDM ``trans_enc`` denoiser, MI355X-native.

Mirror of the reference's ``model/mdm.py`` for the path north_star names: ``arch='trans_enc'``,
``data_rep='hml_vec'`` (also 'rot6d'/'xyz', which share the code path), cond_mode 'no_cond' or
'text'.  The module holds parameters with the reference's state-dict names and shapes (SURVEY.md
§5.4) — so ``load_state_dict`` / ``.to()`` / ``.parameters()`` / checkpoints work unchanged — but
``forward`` does no torch arithmetic: it hands the weights to libcondmdi_hip.so once and runs the
whole denoiser (model/mdm.py:239-306) as hand-written gfx950 kernels.

Not provided (reference-only): trans_dec / gru, the 'better_cond' keypoint
variant, OutputProcessLarge, rot_vel, action embeddings.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .. import _native as N
from .rotation2xyz import Rotation2xyz


class PositionalEncoding(nn.Module):
    """Sinusoidal table pe[pos, 2i] = sin(pos * w_i), pe[pos, 2i+1] = cos(pos * w_i),
    w_i = exp(-2i ln(1e4) / d) (reference :317-335); kept as the `pe` buffer [max_len, 1, d]."""

    def __init__(self, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        freq = torch.exp(torch.arange(0, d_model, 2).float() * (-np.log(10000.0) / d_model))
        pe = torch.zeros(max_len, d_model)
        pe[:, 0::2] = torch.sin(pos * freq)
        pe[:, 1::2] = torch.cos(pos * freq)
        self.register_buffer('pe', pe.unsqueeze(1))


class TimestepEmbedder(nn.Module):
    """Parameter holder for pe[t] -> Linear -> SiLU -> Linear (reference :338-353)."""

    def __init__(self, latent_dim, sequence_pos_encoder):
        super().__init__()
        self.latent_dim = latent_dim
        self.sequence_pos_encoder = sequence_pos_encoder
        self.time_embed = nn.Sequential(nn.Linear(latent_dim, latent_dim), nn.SiLU(),
                                        nn.Linear(latent_dim, latent_dim))


class InputProcess(nn.Module):
    """Parameter holder for the pose embedding Linear(n_feats -> d) (reference :356-380)."""

    def __init__(self, data_rep, input_feats, latent_dim):
        super().__init__()
        self.data_rep, self.input_feats, self.latent_dim = data_rep, input_feats, latent_dim
        self.poseEmbedding = nn.Linear(input_feats, latent_dim)


class OutputProcess(nn.Module):
    """Parameter holder for the final Linear(d -> n_feats) (reference :397-423)."""

    def __init__(self, data_rep, input_feats, latent_dim, njoints, nfeats):
        super().__init__()
        self.data_rep, self.input_feats, self.latent_dim = data_rep, input_feats, latent_dim
        self.njoints, self.nfeats = njoints, nfeats
        self.poseFinal = nn.Linear(latent_dim, input_feats)


class _NativeDenoise(torch.autograd.Function):
    """Differentiable (w.r.t. x) call of the native denoiser: forward = cmdi_mdm_forward with the activation
    stash kept, backward = cmdi_mdm_vjp on that stash.  One stash per engine: the backward must run before the
    engine's next forward (checked through a per-engine call counter)."""

    @staticmethod
    def forward(ctx, x, xin, t, eng):
        out = eng.mdm_forward(xin, t)
        eng._autograd_serial = getattr(eng, "_autograd_serial", 0) + 1
        ctx.eng, ctx.serial, ctx.x_dtype = eng, eng._autograd_serial, x.dtype
        return out

    @staticmethod
    def backward(ctx, gout):
        eng = ctx.eng
        if getattr(eng, "_autograd_serial", 0) != ctx.serial:
            raise RuntimeError("the native denoiser keeps ONE activation stash per engine: call backward / "
                               "torch.autograd.grad before the next model(...) call with requires_grad input")
        gx = eng.mdm_vjp(gout.detach().to(dtype=torch.float32).contiguous())
        return gx.to(ctx.x_dtype), None, None, None


class MDM(nn.Module):
    def __init__(self, modeltype='', njoints=263, nfeats=1, num_actions=1, translation=True,
                 pose_rep='rot6d', glob=True, glob_rot=True, latent_dim=256, ff_size=1024,
                 num_layers=8, num_heads=4, dropout=0.1, ablation=None, activation="gelu",
                 legacy=False, data_rep='rot6d', dataset='amass', clip_dim=512, arch='trans_enc',
                 emb_trans_dec=False, clip_version=None, tf_out_mult=None,
                 train_keypoint_mask='none', **kargs):
        super().__init__()
        if not arch.startswith('trans_enc') or arch.endswith('_large'):
            raise ValueError("the MI355X engine implements arch='trans_enc' only "
                             f"(got {arch!r}; trans_dec / gru / *_large are reference-only)")
        if data_rep not in ('rot6d', 'xyz', 'hml_vec'):
            raise ValueError(f"unsupported data_rep {data_rep!r}")
        if 'better_cond' in train_keypoint_mask or train_keypoint_mask in ('keypoints', 'keyposes'):
            raise ValueError("train_keypoint_mask variants are reference-only")
        if activation != "gelu":
            raise ValueError("only activation='gelu' is implemented")
        self.legacy, self.modeltype = legacy, modeltype
        self.njoints, self.nfeats, self.num_actions = njoints, nfeats, num_actions
        self.data_rep, self.dataset = data_rep, dataset
        self.pose_rep, self.glob, self.glob_rot, self.translation = pose_rep, glob, glob_rot, translation
        self.latent_dim, self.ff_size = latent_dim, ff_size
        self.num_layers, self.num_heads, self.dropout = num_layers, num_heads, dropout
        self.ablation, self.activation, self.clip_dim = ablation, activation, clip_dim
        self.action_emb = kargs.get('action_emb', None)
        self.input_feats = njoints * nfeats
        self.train_keypoint_mask = train_keypoint_mask
        self.normalize_output = kargs.get('normalize_encoder_output', False)
        self.cond_mode = kargs.get('cond_mode', 'no_cond')
        self.cond_mask_prob = kargs.get('cond_mask_prob', 0.)
        # The reference never sets this on MDM, so ClassifierFreeSampleModel(MDM) raises
        # AttributeError there; the transformer ignores keyframes anyway.
        self.keyframe_conditioned = False
        self.arch = arch
        self.emb_trans_dec = emb_trans_dec
        if 'action' in self.cond_mode:
            raise ValueError("action conditioning is reference-only")
        if clip_dim != 512:
            raise ValueError("clip_dim must be 512 (CLIP ViT-B/32 text width)")

        # same construction order as the reference => same default init under the same torch seed
        self.input_process = InputProcess(data_rep, self.input_feats, latent_dim)
        self.sequence_pos_encoder = PositionalEncoding(latent_dim, dropout)
        layer = nn.TransformerEncoderLayer(d_model=latent_dim, nhead=num_heads,
                                           dim_feedforward=ff_size, dropout=dropout,
                                           activation=activation)
        self.seqTransEncoder = nn.TransformerEncoder(layer, num_layers=num_layers,
                                                     enable_nested_tensor=False)
        self.embed_timestep = TimestepEmbedder(latent_dim, self.sequence_pos_encoder)
        self.clip_version = clip_version
        self.clip_model = None
        if 'text' in self.cond_mode:
            self.embed_text = nn.Linear(clip_dim, latent_dim)
            self.clip_model = self.load_and_freeze_clip(clip_version)
        self.output_process = OutputProcess(data_rep, self.input_feats, latent_dim, njoints, nfeats)
        self.rot2xyz = Rotation2xyz(device='cpu', dataset=dataset)
        self._engine = None
        self._engine_key = None

    # ---- reference API ------------------------------------------------------------------------
    def parameters_wo_clip(self):
        return [p for name, p in self.named_parameters() if not name.startswith('clip_model.')]

    def load_and_freeze_clip(self, clip_version):
        """CLIP text tower (third-party, reference :173-186).  Optional here: without the `clip`
        package the caller supplies embeddings through y['text_embed'] ([B, 512])."""
        try:
            import clip  # type: ignore
        except Exception:
            return None
        clip_model, _ = clip.load(clip_version, device='cpu', jit=False)
        clip.model.convert_weights(clip_model)
        clip_model.eval()
        for p in clip_model.parameters():
            p.requires_grad = False
        return clip_model

    def mask_cond(self, cond, force_mask=False):
        if force_mask:
            return torch.zeros_like(cond)
        if self.training and self.cond_mask_prob > 0.:
            raise NotImplementedError("training-time condition dropout is not part of the sampling path")
        return cond

    def encode_text(self, raw_text):
        """CLIP encoding of prompts (reference :211-237): 20 tokens + start/end, zero-padded to 77.  `clip_model` is the
        third-party clip package's model if that package is importable, or a native model.clip_text.CLIPTextTower
        (attach it: ``model.clip_model = CLIPTextTower.from_state_dict(sd, bpe_path=...)``)."""
        if self.clip_model is None:
            raise N.NativeError("no CLIP model available: pass precomputed embeddings in "
                                "model_kwargs['y']['text_embed'] ([B, 512]) or attach a model.clip_text.CLIPTextTower")
        from .clip_text import CLIPTextTower
        device = next(self.parameters()).device
        if isinstance(self.clip_model, CLIPTextTower):
            tok = self.clip_model.tokenize
        else:
            import clip  # type: ignore
            tok = clip.tokenize
        if self.dataset in ('humanml', 'kit'):
            ctx = 20 + 2
            texts = tok(raw_text, context_length=ctx, truncate=True).to(device)
            texts = torch.cat([texts, torch.zeros([texts.shape[0], 77 - ctx], dtype=texts.dtype,
                                                  device=device)], dim=1)
        else:
            texts = tok(raw_text, truncate=True).to(device)
        return self.clip_model.encode_text(texts).float()

    def text_embedding(self, y, batch, device):
        """enc_text [B, 512] for this call: y['text_embed'] if given, else CLIP(y['text']) —
        computed ONCE per sampling call (the reference re-runs CLIP twice per step, Appendix B.5)."""
        if 'text_embed' in y:
            emb = y['text_embed']
        else:
            emb = self.encode_text(y['text'])
        emb = emb.to(device=device, dtype=torch.float32)
        assert emb.shape == (batch, self.clip_dim), emb.shape
        return emb.contiguous()

    # ---- native engine ------------------------------------------------------------------------
    def engine(self, device, max_batch, max_frames, want_grad=False, n_time_rows=None):
        """The native engine holding this module's weights on `device` (built / grown lazily).  The time-embedding
        table is always finalised for every row of `pe` (5000 x d floats, one small GEMM at load time), so forward calls
        and sampling loops with any respacing share ONE engine (`n_time_rows` is accepted for compatibility)."""
        from ..engine import Engine
        device = torch.device(device)
        if device.type == 'cuda' and device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        pe_rows = self.sequence_pos_encoder.pe.shape[0]
        n_time_rows = pe_rows
        eng = self._engine
        precision = getattr(self, "native_precision", None)  # None = library default (f16x3)
        if precision is None and getattr(self, "_range_fallback", False):
            precision = "bf16x6"   # the weights or an earlier run left the f16 range: stay on the unrestricted mode
        need_new = (eng is None or eng.device != device or eng.max_batch < max_batch
                    or eng.max_frames < max_frames or (want_grad and not eng.want_grad)
                    or (precision is not None and eng.precision != precision)
                    or self._engine_key != self._weights_key(n_time_rows))
        if need_new:
            if eng is not None:
                max_batch = max(max_batch, eng.max_batch)
                max_frames = max(max_frames, eng.max_frames)
                want_grad = want_grad or eng.want_grad
                eng.close()
            sd = {k: v for k, v in self.state_dict().items() if not k.startswith('clip_model.')}

            def build(prec):
                e = Engine(n_layers=self.num_layers, d_model=self.latent_dim, d_ff=self.ff_size,
                           n_heads=self.num_heads, n_feats=self.input_feats, max_frames=max_frames,
                           max_batch=max_batch, pe_rows=pe_rows, text_cond='text' in self.cond_mode,
                           want_grad=want_grad, precision=prec, device=device)
                try:
                    e.load_state_dict(sd, n_time_rows=n_time_rows)
                except Exception:
                    e.close()
                    raise
                return e

            try:
                eng = build(precision)
            except N.RangeError:
                if precision is not None:
                    raise              # the caller asked for f16x3 explicitly: report, do not switch silently
                self._range_fallback = True
                eng = build("bf16x6")  # a weight beyond the f16 range: the default falls back to exact bf16 planes
            self._engine = eng
            self._engine_key = self._weights_key(n_time_rows)
        return eng

    def _weights_key(self, n_time_rows):
        # parameter versions change on in-place updates (load_state_dict, optimizer steps, .to())
        return (n_time_rows,) + tuple((p.data_ptr(), p._version) for p in self.parameters_wo_clip())

    def invalidate_engine(self):
        if self._engine is not None:
            self._engine.close()
        self._engine, self._engine_key = None, None

    def check_range(self):
        """forward() launches asynchronously and never reads the device status flag.  Call this after one or more forward calls
        to learn whether any of them left the f16 range of the default precision (RangeError: results invalid — re-run after
        ``range_fallback()`` or with ``native_precision = 'bf16x6'``) or fed an out-of-range device timestep (IndexError, as the
        reference's pe[timesteps] raises).  One 4-byte read-back: synchronises the stream.  The sampling loops and the
        single-step samplers of GaussianDiffusion do this themselves."""
        if self._engine is not None:
            self._engine.check_range()

    def range_certificate(self, **assumptions) -> dict:
        """Weight-only bounds on every tensor the default precision carries as split f16 (utils/range_certificate.py): whether
        ANY input inside the stated assumptions (x_bound, text_l2_bound, n_frames) can reach the f16 range guard, and by how many
        bits not.  trans_enc only; host-side, no device work."""
        if self.arch != 'trans_enc':
            raise NotImplementedError("range_certificate covers arch='trans_enc' (see utils/range_certificate.py for why)")
        from ..utils.range_certificate import trans_enc_range_certificate
        sd = {k: v for k, v in self.state_dict().items() if not k.startswith('clip_model.')}
        return trans_enc_range_certificate(sd, **assumptions)

    def range_fallback(self) -> bool:
        """After a RangeError of the default (f16x3) engine: switch this module to bf16x6 for good and report whether a
        retry makes sense (False if the caller pinned a precision or the fallback is already active)."""
        if getattr(self, "native_precision", None) is not None or getattr(self, "_range_fallback", False):
            return False
        self._range_fallback = True
        self.invalidate_engine()
        return True

    def _forward_native(self, x, timesteps, y, cfg):
        if y is None:
            raise TypeError("MDM.forward needs y (a dict), as in the reference (mdm.py:247)")
        if self.training:
            raise NotImplementedError("the native denoiser is inference-only: call model.eval()")
        device = next(self.parameters()).device
        if device.type != 'cuda':
            raise N.NativeError("MDM runs on a HIP device only (no CPU path): call model.to('cuda')")
        B, J, F, T = x.shape
        assert J * F == self.input_feats
        need_grad = torch.is_grad_enabled() and x.requires_grad
        if timesteps.device.type == 'cpu':   # host tensor: check like the reference's pe[timesteps] would (no sync)
            lo, hi = int(timesteps.min()), int(timesteps.max())
            if lo < 0 or hi >= self.sequence_pos_encoder.pe.shape[0]:
                raise IndexError(f"timesteps out of range for the positional table: [{lo}, {hi}]")
        eng = self.engine(device, max_batch=B, max_frames=T, want_grad=need_grad)
        cond = dict(batch=B, n_frames=T, cfg=cfg)
        if 'text' in self.cond_mode and not y.get('uncond', False):
            cond['enc_text'] = self.text_embedding(y, B, device)
        if cfg:
            cond['text_scale'] = torch.as_tensor(y['text_scale'], dtype=torch.float32).reshape(-1)
        xin = x.detach().to(device=device, dtype=torch.float32).contiguous()
        eng.set_condition(**cond)
        if need_grad:
            # torch.autograd.grad(loss(model(z, t, **kw)), z) — the reference's reconstruction-guidance and
            # cond_fn pattern (gaussian_diffusion.py:411-416, utils/editing_util.py:276-296) — is served by the
            # native input-VJP (cmdi_mdm_vjp): gradients flow to x only, the weights are constants.
            return _NativeDenoise.apply(x, xin, timesteps.to(device), eng)
        return eng.mdm_forward(xin, timesteps.to(device))

    def forward(self, x, timesteps, y=None, obs_x0=None, obs_mask=None, cond_val=None,
                cond_mask=None, **kwargs):
        """x [B, njoints, nfeats, T], timesteps [B] (int, ORIGINAL scale) -> [B, njoints, nfeats, T].
        obs_x0 / obs_mask are accepted and ignored, as the reference's transformer does when called
        through ClassifierFreeSampleModel."""
        return self._forward_native(x, timesteps, y, cfg=False)

    def train(self, mode=True):
        super().train(mode)
        return self
