"""MDM_UNET denoiser ("Diffuser-style" temporal U-Net), MI355X-native.

Mirror of the reference's ``model/mdm_unet.py`` for the configuration CondMDI trains and releases
(``configs/model.py:27-36,65-67``: arch='unet', latent_dim 512, unet_adagn, unet_zero, dim_mults with equal
entries, e.g. (2, 2, 2, 2); hml_vec data; optional keyframe conditioning that concatenates the keyframe mask).
The module holds parameters under the reference's state-dict names and shapes, so checkpoints load unchanged,
but ``forward`` does no torch arithmetic: the whole network (reference :766-849, TemporalUnet :214-358) runs as
hand-written gfx950 kernels (csrc/unet.hip: every 1-D convolution is a split-f16 GEMM over tap-shifted rows).

``attention=True`` (Residual(PreNorm(LinearAttention)) sites, reference :102-156) is built as well, forward and input-VJP.
Not provided (reference-only): adagn=False, 'unet_large', xz_only / traj models, train_keypoint_mask variants, action
conditioning.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _native as N
from .mdm import MDM, PositionalEncoding, TimestepEmbedder
from .rotation2xyz import Rotation2xyz


def _conv_gn(cin, cout, k, with_mish):
    mods = [nn.Conv1d(cin, cout, k, padding=k // 2), nn.Identity(), nn.GroupNorm(8, cout), nn.Identity()]
    if with_mish:
        mods.append(nn.Mish())
    return nn.Sequential(*mods)


class _Conv1dAdaGNBlock(nn.Module):   # reference :70-101 (parameter holder)
    def __init__(self, cin, cout, k):
        super().__init__()
        self.block1 = _conv_gn(cin, cout, k, with_mish=False)
        self.block2 = nn.Mish()


class _Conv1dBlock(nn.Module):        # reference :34-67
    def __init__(self, cin, cout, k, zero=False):
        super().__init__()
        self.block = _conv_gn(cin, cout, k, with_mish=True)
        if zero:
            nn.init.zeros_(self.block[0].weight)
            nn.init.zeros_(self.block[0].bias)


class _ResidualTemporalBlock(nn.Module):   # reference :163-212
    def __init__(self, cin, cout, embed_dim, zero):
        super().__init__()
        self.blocks = nn.ModuleList([_Conv1dAdaGNBlock(cin, cout, 5), _Conv1dBlock(cout, cout, 5, zero=zero)])
        self.time_mlp = nn.Sequential(nn.Mish(), nn.Linear(embed_dim, cout * 2), nn.Identity())
        nn.init.zeros_(self.time_mlp[1].weight)
        nn.init.zeros_(self.time_mlp[1].bias)
        self.residual_conv = nn.Conv1d(cin, cout, 1) if cin != cout else nn.Identity()


class _Down(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.conv = nn.Conv1d(dim, dim, 3, 2, 1)


class _Up(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.conv = nn.ConvTranspose1d(dim, dim, 4, 2, 1)


class _ChannelLayerNorm(nn.Module):   # reference :111-121
    def __init__(self, dim):
        super().__init__()
        self.g = nn.Parameter(torch.ones(1, dim, 1))
        self.b = nn.Parameter(torch.zeros(1, dim, 1))


class _LinearAttention(nn.Module):    # reference :135-156
    def __init__(self, dim, heads=4, dim_head=32):
        super().__init__()
        self.heads = heads
        self.to_qkv = nn.Conv1d(dim, heads * dim_head * 3, 1, bias=False)
        self.to_out = nn.Conv1d(heads * dim_head, dim, 1)


class _PreNorm(nn.Module):            # reference :124-132
    def __init__(self, dim, fn):
        super().__init__()
        self.fn = fn
        self.norm = _ChannelLayerNorm(dim)


class _Residual(nn.Module):           # reference :102-108
    def __init__(self, fn):
        super().__init__()
        self.fn = fn


def _attn_site(dim, attention):
    return _Residual(_PreNorm(dim, _LinearAttention(dim))) if attention else nn.Identity()


class TemporalUnet(nn.Module):   # reference :214-358 (parameter holder; the arithmetic lives in csrc/unet.hip)
    def __init__(self, input_dim, cond_dim, dim, dim_mults, zero, added_input_channels, attention=False):
        super().__init__()
        dims = [input_dim, *[int(dim * m) for m in dim_mults]]
        in_out = list(zip(dims[:-1], dims[1:]))
        self.time_mlp = nn.Sequential(nn.Linear(cond_dim, dim * 4), nn.Mish(), nn.Linear(dim * 4, dim))
        self.downs, self.ups = nn.ModuleList([]), nn.ModuleList([])
        n_res = len(in_out)
        for ind, (din, dout) in enumerate(in_out):
            last = ind >= n_res - 1
            self.downs.append(nn.ModuleList([
                _ResidualTemporalBlock(din + added_input_channels * (ind == 0), dout, dim, zero),
                _ResidualTemporalBlock(dout, dout, dim, zero), _attn_site(dout, attention),
                _Down(dout) if not last else nn.Identity()]))
        mid = dims[-1]
        self.mid_block1 = _ResidualTemporalBlock(mid, mid, dim, zero)
        self.mid_attn = _attn_site(mid, attention)
        self.mid_block2 = _ResidualTemporalBlock(mid, mid, dim, zero)
        for ind, (din, dout) in enumerate(reversed(in_out[1:])):
            self.ups.append(nn.ModuleList([
                _ResidualTemporalBlock(dout * 2, din, dim, zero), _ResidualTemporalBlock(din, din, dim, zero),
                _attn_site(din, attention), _Up(din)]))
        self.final_conv = nn.Sequential(_Conv1dBlock(din, din, 5), nn.Conv1d(din, input_dim, 1))
        if zero:
            nn.init.zeros_(self.final_conv[1].weight)
            nn.init.zeros_(self.final_conv[1].bias)


class MDM_UNET(nn.Module):
    def __init__(self, modeltype='', njoints=263, nfeats=1, num_actions=1, translation=True, pose_rep='rot6d',
                 glob=True, glob_rot=True, latent_dim=512, dim_mults=(2, 2, 2, 2), attention=False, ablation=None,
                 legacy=False, data_rep='hml_vec', dataset='humanml', clip_dim=512, emb_trans_dec=False,
                 clip_version=None, adagn=True, zero=True, arch='unet', unet_out_mult=8, xz_only=False,
                 train_keypoint_mask='none', keyframe_conditioned=False, keyframe_selection_scheme='in-between',
                 zero_keyframe_loss=False, **kwargs):
        super().__init__()
        if arch != 'unet' or not adagn or xz_only or train_keypoint_mask != 'none':
            raise ValueError("the MI355X engine implements arch='unet' with adagn=True, xz_only=False, "
                             "train_keypoint_mask='none' (the CondMDI configuration; attention=False or True)")
        if latent_dim != 512 or len(dim_mults) != 4 or len(set(dim_mults)) != 1 or int(latent_dim * dim_mults[0]) % 256:
            raise ValueError("the MI355X engine needs latent_dim=512 and four equal dim_mults (e.g. (2, 2, 2, 2))")
        if dataset != 'humanml' or data_rep != 'hml_vec':
            raise ValueError("only the humanml / hml_vec data representation is implemented")
        self.legacy, self.modeltype = legacy, modeltype
        self.njoints, self.nfeats, self.num_actions = njoints, nfeats, num_actions
        self.data_rep, self.dataset = data_rep, dataset
        self.pose_rep, self.glob, self.glob_rot, self.translation = pose_rep, glob, glob_rot, translation
        self.latent_dim, self.dim_mults, self.attention = latent_dim, tuple(dim_mults), attention
        self.ablation, self.clip_dim = ablation, clip_dim
        self.keyframe_conditioned = keyframe_conditioned
        self.zero_keyframe_loss = zero_keyframe_loss
        self.train_keypoint_mask, self.xz_only = train_keypoint_mask, xz_only
        self.input_feats = njoints * nfeats
        self.cond_mode = kwargs.get('cond_mode', 'no_cond')
        self.cond_mask_prob = kwargs.get('cond_mask_prob', 0.)
        self.arch = arch
        if 'action' in self.cond_mode:
            raise ValueError("action conditioning is reference-only")
        added = self.input_feats if keyframe_conditioned else 0
        self.added_channels = added
        self.sequence_pos_encoder = PositionalEncoding(latent_dim, dropout=0)
        self.unet = TemporalUnet(self.input_feats, latent_dim, latent_dim, self.dim_mults, zero, added, attention=attention)
        self.embed_timestep = TimestepEmbedder(latent_dim, self.sequence_pos_encoder)
        self.clip_version = clip_version
        self.clip_model = None
        if 'text' in self.cond_mode:
            self.embed_text = nn.Linear(clip_dim, latent_dim)
            self.clip_model = self.load_and_freeze_clip(clip_version)
        self.rot2xyz = Rotation2xyz(device='cpu', dataset=dataset)
        self._engine = None
        self._engine_key = None

    # shared with the transformer denoiser (same semantics in the reference: mdm_unet.py:706-764)
    parameters_wo_clip = MDM.parameters_wo_clip
    load_and_freeze_clip = MDM.load_and_freeze_clip
    mask_cond = MDM.mask_cond
    encode_text = MDM.encode_text
    text_embedding = MDM.text_embedding
    _weights_key = MDM._weights_key
    invalidate_engine = MDM.invalidate_engine
    check_range = MDM.check_range

    def range_fallback(self) -> bool:
        """After a RangeError of the default (f16x3) engine: bf16x6 for good (round 5) — except for attention=True, whose
        LinearAttention sites exist in f16x3 only: no fallback is taken and no state changes, so the caller's RangeError
        propagates and the module keeps working on in-range inputs."""
        if self.attention:
            return False
        return MDM.range_fallback(self)

    def engine(self, device, max_batch, max_frames, want_grad=False, n_time_rows=1000):
        """The native engine holding this module's weights on `device` (built / grown lazily)."""
        from ..engine import Engine
        device = torch.device(device)
        if device.type == 'cuda' and device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        pe_rows = self.sequence_pos_encoder.pe.shape[0]
        n_time_rows = pe_rows   # one table for forward calls and every respacing (see MDM.engine)
        eng = self._engine
        # precisions of the native U-Net: 'f16x3' (library default; 22-bit split operands, |x| < 65504) and 'bf16x6' (exact
        # three-plane operands on every convolution, fp32's range — the reference U-Net is plain fp32 at any activation scale,
        # model/mdm_unet.py:561-849).  As MDM.engine: None = the default, which falls back to bf16x6 when a weight (here) or an
        # activation (GaussianDiffusion._range_probe / _with_range_fallback) leaves the f16 range; a pinned precision raises.
        precision = getattr(self, "native_precision", None)
        if precision is None and getattr(self, "_range_fallback", False):
            precision = "bf16x6"
        if precision == "bf16x6" and self.attention:
            raise N.NativeError("MDM_UNET(attention=True) is built for the f16x3 precision only")
        need_new = (eng is None or eng.device != device or eng.max_batch < max_batch or eng.max_frames < max_frames
                    or (want_grad and not eng.want_grad) or (precision is not None and eng.precision != precision)
                    or self._engine_key != self._weights_key(n_time_rows))
        if need_new:
            if eng is not None:
                max_batch, max_frames = max(max_batch, eng.max_batch), max(max_frames, eng.max_frames)
                want_grad = want_grad or eng.want_grad
                eng.close()
            sd = {k: v for k, v in self.state_dict().items() if not k.startswith('clip_model.')}

            def build(prec):
                e = Engine(n_layers=0, d_model=self.latent_dim, d_ff=0, n_heads=0, n_feats=self.input_feats,
                           max_frames=max_frames, max_batch=max_batch, pe_rows=pe_rows,
                           text_cond='text' in self.cond_mode, want_grad=want_grad, precision=prec, arch="unet",
                           unet_added=self.added_channels, unet_mults=self.dim_mults, unet_attention=self.attention,
                           device=device)
                try:
                    e.load_state_dict(sd, n_time_rows=n_time_rows)
                except Exception:
                    e.close()
                    raise
                return e

            try:
                eng = build(precision)
            except N.RangeError:
                if precision is not None or self.attention:
                    raise              # pinned to f16x3 (or no wider mode for this geometry): report, do not switch silently
                self._range_fallback = True
                eng = build("bf16x6")  # a weight beyond the f16 range
            self._engine = eng
            self._engine_key = self._weights_key(n_time_rows)
        return eng

    def _forward_native(self, x, timesteps, y, cfg, obs_x0=None, obs_mask=None):
        if y is None:
            raise TypeError("MDM_UNET.forward needs y (a dict), as in the reference (mdm_unet.py:794)")
        if self.training:
            raise NotImplementedError("the native denoiser is inference-only: call model.eval()")
        assert (obs_x0 is None) == (obs_mask is None), \
            'with spatial-conditioning, both obs_x0 and obs_mask must be provided'
        device = next(self.parameters()).device
        if device.type != 'cuda':
            raise N.NativeError("MDM_UNET runs on a HIP device only (no CPU path): call model.to('cuda')")
        B, J, F, T = x.shape
        assert J * F == self.input_feats
        need_grad = torch.is_grad_enabled() and x.requires_grad
        eng = self.engine(device, max_batch=B, max_frames=T, want_grad=need_grad,
                          n_time_rows=self.sequence_pos_encoder.pe.shape[0])
        cond = dict(batch=B, n_frames=T, cfg=cfg)
        if 'text' in self.cond_mode and not y.get('uncond', False):
            cond['enc_text'] = self.text_embedding(y, B, device)
        if cfg:
            cond['text_scale'] = torch.as_tensor(y['text_scale'], dtype=torch.float32).reshape(-1)
        if self.keyframe_conditioned:
            cond['obs_x0'], cond['obs_mask'] = obs_x0, obs_mask
        eng.set_condition(**cond)
        xin = x.detach().to(device=device, dtype=torch.float32).contiguous()
        if need_grad:   # torch.autograd.grad(loss(model(z, ...)), z) through the native input-VJP (see model/mdm.py)
            from .mdm import _NativeDenoise
            return _NativeDenoise.apply(x, xin, timesteps.to(device), eng)
        return eng.mdm_forward(xin, timesteps.to(device))

    def forward(self, x, timesteps, y=None, obs_x0=None, obs_mask=None, **kwargs):
        """x [B, njoints, nfeats, T], timesteps [B] (ORIGINAL scale), obs_x0 / obs_mask [B, njoints, nfeats, T]
        (used iff keyframe_conditioned) -> [B, njoints, nfeats, T]   (reference :766-849)."""
        return self._forward_native(x, timesteps, y, cfg=False, obs_x0=obs_x0, obs_mask=obs_mask)

    def train(self, mode=True):
        super().train(mode)
        return self
