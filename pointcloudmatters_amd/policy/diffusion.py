"""Diffusion-Policy branch of the BC path: point-cloud observation encoder, conditional 1-D U-Net,
DDPM forward process and the epsilon-prediction loss.

Behavioural counterparts (same constructor arguments, same parameter names, same maths):
  PCDObsEncoder           /root/reference/src/models/components/diffusion_policy/vision/pcd_obs_encoder.py:14-296
  ConditionalUnet1D       .../diffusion/conditional_unet1d.py:17-297, conv1d_components.py:8-45,
                          positional_embedding.py:7-19
  LowdimMaskGenerator     .../diffusion/mask_generator.py:41-105
  DiffusionUnetPcdPolicy  .../diffusion_unet_image_policy.py:23-313 (training path: compute_loss)
  LinearNormalizer        /root/reference/src/utils/diffusion_policy/normalizer.py:14-195 +
                          src/utils/normalize_utils.py:7-21 (range normaliser)
  DDPMSchedule            diffusers==0.29.0 DDPMScheduler (requirements.txt:31; third-party, absent
                          here): `squaredcos_cap_v2` betas and add_noise restated from the published
                          algorithm -- parity unpinned.
The set-abstraction layer is the same code path as ACT's (policy/sa_layer.py), fed by the HIP
pointops; everything dense goes through hipBLASLt / MIOpen via torch.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .rows_linear import linear_rows
from .sa_layer import set_abstraction
from .unet_ops import conv1d_cl, conv_transpose1d_cl, gn_mish_cl
from .._lib import raw_stream as _raw_stream
from .rows_linear import RowsLinear


# the projector of PCDObsEncoder in row layout through csrc/bnrelu.hip + csrc/bnact.hip (round 5).  OPT-IN since round 6 (PCM_PROJECTOR_ROWS=1):
# its last BatchNorm runs the pcm_bnact_* kernels, device code no GPU has executed yet, and the shipped default launches only kernels that
# have a green hardware run (profiles/r06_summary.md).  Off: the module path of round 4 (framework BatchNorm / SyncBatchNorm for the
# projector; data-parallel Diffusion-Policy runs then take the hybrid mode instead of the fully captured chain).
PROJECTOR_ROWS = os.environ.get("PCM_PROJECTOR_ROWS", "0") != "0"


# ----------------------------------------------------------------------------- U-Net pieces
def conv1d_gemm(x, conv):
    """``conv(x)`` for an ``nn.Conv1d`` as ONE im2col copy + ONE GEMM over the whole batch.

    The U-Net works on horizons of 16 / 8 / 4 steps with 512-4096 channels: MIOpen serves these shapes with
    per-sample im2col + GEMM loops (~900 Im2Col and ~200 small GEMM launches per training step at B=64);
    here every convolution is a single (B*L_out, C_in*K) x (C_in*K, C_out) hipBLASLt GEMM (bf16 under
    autocast), its backward two GEMMs.  Same arithmetic up to summation order."""
    k, stride, pad = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    b, cin, _ = x.shape
    w = conv.weight  # (C_out, C_in, K)
    if k == 1 and stride == 1 and pad == 0:
        y = linear_rows(x.transpose(1, 2), w[:, :, 0], conv.bias)  # (B, L, C_out)
        return y.transpose(1, 2)
    xp = F.pad(x, (pad, pad)) if pad else x
    cols = xp.unfold(2, k, stride)  # (B, C_in, L_out, K) view
    lout = cols.shape[2]
    cols = cols.permute(0, 2, 1, 3).reshape(b * lout, cin * k)  # the one im2col copy
    y = F.linear(cols, w.reshape(w.shape[0], cin * k), conv.bias)
    return y.view(b, lout, -1).transpose(1, 2)


class SinusoidalPosEmb(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        half = self.dim // 2
        freq = torch.exp(torch.arange(half, device=x.device) * -(math.log(10000) / (half - 1)))
        ang = x[:, None] * freq[None, :]
        return torch.cat((ang.sin(), ang.cos()), dim=-1)


class Downsample1d(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.conv = nn.Conv1d(dim, dim, 3, 2, 1)

    def forward_cl(self, x):
        return conv1d_cl(x, self.conv)

    def forward(self, x):  # (B, C, T) like the reference module
        return self.forward_cl(x.transpose(1, 2)).transpose(1, 2)


class Upsample1d(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.conv = nn.ConvTranspose1d(dim, dim, 4, 2, 1)

    def forward_cl(self, x):
        return conv_transpose1d_cl(x, self.conv)

    def forward(self, x):
        return self.forward_cl(x.transpose(1, 2)).transpose(1, 2)


class Conv1dBlock(nn.Module):
    """Conv1d -> GroupNorm -> Mish.  `forward_cl` is the channels-last form the U-Net runs: im2col + GEMM, then ONE
    fused GroupNorm + Mish (+ FiLM, + residual) launch (csrc/gnmish.hip)."""

    def __init__(self, inp_channels, out_channels, kernel_size, n_groups=8):
        super().__init__()
        self.block = nn.Sequential(
            nn.Conv1d(inp_channels, out_channels, kernel_size, padding=kernel_size // 2),
            nn.GroupNorm(n_groups, out_channels),
            nn.Mish(),
        )

    def forward_cl(self, x, film=None, film_mode=0, res=None):
        conv = self.block[0]  # its bias rides in the fused launch: added in front of the norm, gradient from the same backward
        return gn_mish_cl(conv1d_cl(x, conv, with_bias=False), self.block[1], film=film, film_mode=film_mode, res=res,
                          conv_bias=conv.bias)

    def forward(self, x):
        return self.forward_cl(x.transpose(1, 2)).transpose(1, 2)


class _AddTrailingDim(nn.Module):  # stands where the reference has einops Rearrange("batch t -> batch t 1")
    def forward(self, x):
        return x.unsqueeze(-1)


class ConditionalResidualBlock1D(nn.Module):
    """Two conv blocks with FiLM conditioning in between and a (1x1-conv) residual."""

    def __init__(self, in_channels, out_channels, cond_dim, kernel_size=3, n_groups=8, cond_predict_scale=False):
        super().__init__()
        self.blocks = nn.ModuleList([
            Conv1dBlock(in_channels, out_channels, kernel_size, n_groups=n_groups),
            Conv1dBlock(out_channels, out_channels, kernel_size, n_groups=n_groups),
        ])
        self.cond_predict_scale = cond_predict_scale
        self.out_channels = out_channels
        film = out_channels * 2 if cond_predict_scale else out_channels
        self.cond_encoder = nn.Sequential(nn.Mish(), RowsLinear(cond_dim, film), _AddTrailingDim())
        self.residual_conv = nn.Conv1d(in_channels, out_channels, 1) if in_channels != out_channels else nn.Identity()

    def forward_cl(self, x, mish_cond):
        """x (B, T, C_in); mish_cond = Mish(cond), shared by every block of the U-Net (the reference recomputes it in
        each cond_encoder).  FiLM rides in the first block's fused launch, the residual add in the second's."""
        film = self.cond_encoder[1](mish_cond)  # (B, 2*C_out) = scale | bias, or (B, C_out) = bias
        out = self.blocks[0].forward_cl(x, film=film, film_mode=1 if self.cond_predict_scale else 2)
        res = conv1d_cl(x, self.residual_conv) if isinstance(self.residual_conv, nn.Conv1d) else x
        return self.blocks[1].forward_cl(out, res=res)

    def forward(self, x, cond):
        return self.forward_cl(x.transpose(1, 2), F.mish(cond)).transpose(1, 2)


class ConditionalUnet1D(nn.Module):
    def __init__(self, input_dim, local_cond_dim=None, global_cond_dim=None, diffusion_step_embed_dim=256,
                 down_dims=(256, 512, 1024), kernel_size=3, n_groups=8, cond_predict_scale=False):
        super().__init__()
        if local_cond_dim is not None:
            raise NotImplementedError("local conditioning is not used by any point-cloud config")
        dims = [input_dim] + list(down_dims)
        dsed = diffusion_step_embed_dim
        self.diffusion_step_encoder = nn.Sequential(
            SinusoidalPosEmb(dsed), RowsLinear(dsed, dsed * 4), nn.Mish(), RowsLinear(dsed * 4, dsed))
        cond_dim = dsed + (global_cond_dim or 0)
        kw = dict(cond_dim=cond_dim, kernel_size=kernel_size, n_groups=n_groups, cond_predict_scale=cond_predict_scale)
        pairs = list(zip(dims[:-1], dims[1:]))
        self.local_cond_encoder = None
        self.mid_modules = nn.ModuleList([ConditionalResidualBlock1D(dims[-1], dims[-1], **kw) for _ in range(2)])
        self.down_modules = nn.ModuleList([
            nn.ModuleList([ConditionalResidualBlock1D(cin, cout, **kw), ConditionalResidualBlock1D(cout, cout, **kw),
                           Downsample1d(cout) if i < len(pairs) - 1 else nn.Identity()])
            for i, (cin, cout) in enumerate(pairs)])
        # conditional_unet1d.py:182-208: `is_last` compares against len(in_out)-1 while iterating over
        # len(in_out)-1 entries, so every up stage keeps its Upsample1d
        self.up_modules = nn.ModuleList([
            nn.ModuleList([ConditionalResidualBlock1D(cout * 2, cin, **kw), ConditionalResidualBlock1D(cin, cin, **kw),
                           Upsample1d(cin)])
            for (cin, cout) in reversed(pairs[1:])])
        self.final_conv = nn.Sequential(Conv1dBlock(down_dims[0], down_dims[0], kernel_size=kernel_size),
                                        nn.Conv1d(down_dims[0], input_dim, 1))

    def forward(self, sample, timestep, local_cond=None, global_cond=None, **kwargs):
        """sample (B, T, input_dim), timestep (B,) -> (B, T, input_dim).  The reference moves to (B, C, T) and back
        (conditional_unet1d.py:236, 295); here the activations stay (B, T, C) from end to end."""
        x = sample
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.long, device=sample.device)
        elif t.dim() == 0:
            t = t[None].to(sample.device)
        t = t.expand(sample.shape[0])
        cond = self.diffusion_step_encoder(t)
        if global_cond is not None:
            cond = torch.cat([cond, global_cond], dim=-1)
        mish_cond = F.mish(cond)
        if mish_cond.is_cuda and torch.is_autocast_enabled("cuda"):
            mish_cond = mish_cond.to(torch.get_autocast_dtype("cuda"))  # one cast instead of one per cond_encoder
        from . import staging  # backward-stage boundaries for the overlapped gradient exchange (identity otherwise)

        skips = []
        for res1, res2, down in self.down_modules:
            x = res2.forward_cl(res1.forward_cl(x, mish_cond), mish_cond)
            skips.append(x)
            if not isinstance(down, nn.Identity):
                x = down.forward_cl(x)
        x, mish_cond, *skips = staging.cut("unet.mid", x, mish_cond, *skips)
        for mid in self.mid_modules:
            x = mid.forward_cl(x, mish_cond)
        x, mish_cond, *skips = staging.cut("unet.up", x, mish_cond, *skips)
        for res1, res2, up in self.up_modules:
            x = torch.cat((x, skips.pop()), dim=-1)
            x = up.forward_cl(res2.forward_cl(res1.forward_cl(x, mish_cond), mish_cond))
        x = self.final_conv[0].forward_cl(x)
        return conv1d_cl(x, self.final_conv[1])


# ----------------------------------------------------------------------------- DDPM forward process
class DDPMSchedule(nn.Module):
    """squaredcos_cap_v2 betas (alpha_bar(t) = cos^2((t+0.008)/1.008 * pi/2), beta capped at 0.999),
    float64 on the host then float32, and x_t = sqrt(abar_t) x_0 + sqrt(1-abar_t) eps."""

    def __init__(self, num_train_timesteps=100, beta_schedule="squaredcos_cap_v2", prediction_type="epsilon",
                 beta_start=0.0001, beta_end=0.02, clip_sample=True, clip_sample_range=1.0,
                 variance_type="fixed_small", **unused):
        super().__init__()
        n = num_train_timesteps
        if beta_schedule == "squaredcos_cap_v2":
            abar = lambda u: math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2
            betas = [min(1 - abar((i + 1) / n) / abar(i / n), 0.999) for i in range(n)]
            betas = torch.tensor(betas, dtype=torch.float32)
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, n, dtype=torch.float32)
        else:
            raise NotImplementedError(beta_schedule)
        if prediction_type != "epsilon" or variance_type != "fixed_small":
            raise NotImplementedError("only epsilon prediction with fixed_small variance is configured by the reference")
        self.num_train_timesteps = n
        self.prediction_type = prediction_type
        self.clip_sample, self.clip_sample_range = bool(clip_sample), float(clip_sample_range)
        self.register_buffer("alphas_cumprod", torch.cumprod(1.0 - betas, dim=0), persistent=False)
        self.set_timesteps(n)

    def add_noise(self, original, noise, timesteps):
        ac = self.alphas_cumprod.to(device=original.device, dtype=original.dtype)[timesteps]
        a = ac.sqrt().reshape(-1, *([1] * (original.dim() - 1)))
        s = (1 - ac).sqrt().reshape(-1, *([1] * (original.dim() - 1)))
        return a * original + s * noise

    # ---- reverse process (rollout): DDPMScheduler.set_timesteps / step of diffusers 0.29, "leading" spacing
    def set_timesteps(self, num_inference_steps):
        n = self.num_train_timesteps
        if not 0 < num_inference_steps <= n:
            raise ValueError(f"num_inference_steps must be in 1..{n}")
        self.num_inference_steps = int(num_inference_steps)
        ratio = n // self.num_inference_steps
        self.timesteps = [int(round(i * ratio)) for i in range(self.num_inference_steps)][::-1]
        self._coef_cache, self._t_dev = {}, {}
        for t in self.timesteps:  # all host-side scalars now: nothing left to compute (or copy) inside a graph capture
            self.step_coefficients(t)

    def timesteps_on(self, device):
        """The timestep sequence as a device int64 tensor (cached): the sampler slices it instead of building a
        one-element tensor from a Python int in every iteration (a pageable H2D copy, illegal under capture)."""
        key = str(device)
        if key not in self._t_dev:
            self._t_dev[key] = torch.tensor(self.timesteps, dtype=torch.long, device=device)
        return self._t_dev[key]

    def step_coefficients(self, t):
        """(sqrt(abar_t), sqrt(1-abar_t), coef_x0, coef_xt, sigma) as python floats holding fp32 values, computed with
        fp32 tensor arithmetic in the order of DDPMScheduler.step / _get_variance."""
        t = int(t)
        hit = self._coef_cache.get(t)
        if hit is not None:
            return hit
        ac = self.alphas_cumprod.detach().to("cpu", torch.float32)
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = ac[t]
        a_prev = ac[prev_t] if prev_t >= 0 else torch.tensor(1.0)
        b_t, b_prev = 1 - a_t, 1 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1 - cur_a
        coef_x0 = (a_prev ** 0.5 * cur_b) / b_t
        coef_xt = cur_a ** 0.5 * b_prev / b_t
        sigma = torch.tensor(0.0)
        if t > 0:
            sigma = torch.clamp(b_prev / b_t * cur_b, min=1e-20) ** 0.5
        out = tuple(float(v) for v in (a_t ** 0.5, b_t ** 0.5, coef_x0, coef_xt, sigma))
        self._coef_cache[t] = out
        return out

    def step(self, model_output, t, sample, noise=None, generator=None, cond_mask=None, cond=None, out=None):
        """x_t -> x_{t-1}.  On the GPU this is ONE launch of pcm_ddpm_step_hip (csrc/ddpm.hip); host tensors take the
        same fp32 chain through torch ops.  `noise` (variance noise, drawn here when None and t > 0) and the optional
        conditioning are fused into the same launch."""
        sa, sb, c0, ct, sigma = self.step_coefficients(t)
        clip = self.clip_sample_range if self.clip_sample else 0.0
        if noise is None and sigma != 0.0:
            noise = torch.randn(sample.shape, device=sample.device, dtype=torch.float32, generator=generator)
        if sample.is_cuda:
            from .. import _lib

            L = _lib.load()
            eps = model_output.contiguous()
            if eps.dtype not in (torch.float32, torch.bfloat16):
                eps = eps.float()
            xt = sample.contiguous()
            assert xt.dtype == torch.float32 and eps.shape == xt.shape
            prev = torch.empty_like(xt) if out is None else out
            nz = noise.contiguous() if (noise is not None and sigma != 0.0) else None
            mask8 = cond_mask.contiguous().view(torch.uint8) if cond_mask is not None else None
            cnd = cond.contiguous() if cond_mask is not None else None
            with torch.cuda.device(xt.device):
                rc = L.pcm_ddpm_step_hip(xt.numel(), int(eps.dtype == torch.bfloat16), eps.data_ptr(), xt.data_ptr(),
                                         nz.data_ptr() if nz is not None else 0,
                                         mask8.data_ptr() if mask8 is not None else 0,
                                         cnd.data_ptr() if cnd is not None else 0, sa, sb, c0, ct, sigma, clip,
                                         prev.data_ptr(), _raw_stream())
            _lib.check(rc, "pcm_ddpm_step_hip")
            return prev
        x0 = (sample - sb * model_output.float()) / sa
        if clip > 0:
            x0 = x0.clamp(-clip, clip)
        prev = c0 * x0 + ct * sample
        if noise is not None and sigma != 0.0:
            prev = prev + sigma * noise
        if cond_mask is not None:
            prev = torch.where(cond_mask, cond, prev)
        return prev


# ----------------------------------------------------------------------------- small helpers
class _AttrMixin(nn.Module):
    """The reference's ModuleAttrMixin owns an empty `_dummy_variable` parameter (module_attr_mixin.py:4-7)
    that never receives a gradient; kept (frozen) so state-dict keys match."""

    def __init__(self):
        super().__init__()
        self._dummy_variable = nn.Parameter(torch.empty(0), requires_grad=False)

    @property
    def device(self):
        return next(iter(self.parameters())).device


class LowdimMaskGenerator(_AttrMixin):
    def __init__(self, action_dim, obs_dim, max_n_obs_steps=2, fix_obs_steps=True, action_visible=False):
        super().__init__()
        self.action_dim, self.obs_dim = action_dim, obs_dim
        self.max_n_obs_steps, self.fix_obs_steps, self.action_visible = max_n_obs_steps, fix_obs_steps, action_visible

    @torch.no_grad()
    def forward(self, shape, seed=None, device=None):
        b, t, d = shape
        assert d == self.action_dim + self.obs_dim
        device = device if device is not None else self.device
        is_action = torch.zeros(shape, dtype=torch.bool, device=device)
        is_action[..., : self.action_dim] = True
        if self.fix_obs_steps:
            obs_steps = torch.full((b,), self.max_n_obs_steps, device=device)
        else:
            g = torch.Generator(device=device)
            if seed is not None:
                g.manual_seed(seed)
            obs_steps = torch.randint(1, self.max_n_obs_steps + 1, (b,), generator=g, device=device)
        steps = torch.arange(t, device=device)[None, :].expand(b, t)
        mask = (steps < obs_steps[:, None])[..., None].expand(b, t, d) & ~is_action
        if self.action_visible:
            act_steps = torch.clamp(obs_steps - 1, min=0)
            mask = mask | ((steps < act_steps[:, None])[..., None].expand(b, t, d) & is_action)
        return mask


class LinearNormalizer(nn.Module):
    """x -> x * scale + offset per key; parameters nested like the reference
    (``params_dict.<key>.{scale,offset,input_stats.{min,max,mean,std}}``), all frozen."""

    def __init__(self):
        super().__init__()
        self.params_dict = nn.ParameterDict()

    @staticmethod
    def _frozen(t):
        return nn.Parameter(torch.as_tensor(t, dtype=torch.float32).clone(), requires_grad=False)

    def set_range(self, key, data_min, data_max, data_mean=None, data_std=None, output_max=1.0, output_min=-1.0, range_eps=1e-4):
        """normalize_utils.py:7-21: map [min, max] to [-1, 1]; near-constant dims keep scale 1."""
        mn = torch.as_tensor(data_min, dtype=torch.float32).flatten().clone()
        mx = torch.as_tensor(data_max, dtype=torch.float32).flatten().clone()
        rng = mx - mn
        ignore = rng < range_eps
        rng[ignore] = output_max - output_min
        scale = (output_max - output_min) / rng
        offset = output_min - scale * mn
        offset[ignore] = (output_max + output_min) / 2 - mn[ignore]
        stats = {"min": mn, "max": mx, "mean": mn * 0 if data_mean is None else data_mean, "std": mn * 0 + 1 if data_std is None else data_std}
        self.params_dict[key] = nn.ParameterDict({
            "scale": self._frozen(scale), "offset": self._frozen(offset),
            "input_stats": nn.ParameterDict({k: self._frozen(v) for k, v in stats.items()}),
        })

    def fit(self, data, **kw):
        for k, v in data.items():
            flat = torch.as_tensor(v, dtype=torch.float32).reshape(-1, v.shape[-1])
            self.set_range(k, flat.min(0).values, flat.max(0).values, flat.mean(0), flat.std(0), **kw)

    def _affine(self, x, key, forward=True):
        p = self.params_dict[key]
        scale, offset = p["scale"], p["offset"]
        shape = x.shape
        x = x.to(device=scale.device, dtype=scale.dtype).reshape(-1, scale.shape[0])
        x = x * scale + offset if forward else (x - offset) / scale
        return x.reshape(shape)

    def normalize(self, x, key=None):
        if isinstance(x, dict):
            return {k: self._affine(v, k) for k, v in x.items()}
        return self._affine(x, key)

    def unnormalize(self, x, key=None):
        if isinstance(x, dict):
            return {k: self._affine(v, k, forward=False) for k, v in x.items()}
        return self._affine(x, key, forward=False)

    def __getitem__(self, key):
        return _Field(self, key)


class _Field:
    def __init__(self, owner, key):
        self.owner, self.key = owner, key

    def normalize(self, x):
        return self.owner.normalize(x, self.key)

    def unnormalize(self, x):
        return self.owner.unnormalize(x, self.key)


# ----------------------------------------------------------------------------- observation encoder
class PCDObsEncoder(_AttrMixin):
    def __init__(self, shape_meta, pcd_model, share_pcd_model=True, n_obs_step=2, pcd_nsample=16, pcd_npoints=1024,
                 use_mask=False, bg_ratio=0.0, pcd_hidden_dim=128, projector_layers=2, projector_channels=(128, 128, 128),
                 pre_sample=False, in_channel=6, pointops=None, sa_impl="reference", overlap_sampling=True, **kwargs):
        super().__init__()
        if not 0.0 <= bg_ratio < 1.0:
            raise ValueError("bg_ratio must be in [0, 1)")
        self.use_mask, self.bg_ratio = use_mask, bg_ratio
        if pointops is None:
            from .. import pointops as _hip_pointops

            pointops = _hip_pointops
        self._pointops = [pointops]
        self.sa_impl, self.overlap_sampling = sa_impl, overlap_sampling
        obs_meta = shape_meta["obs"]
        self.pcd_keys = sorted(k for k, a in obs_meta.items() if a.get("type", "low_dim") == "pcd")
        self.low_dim_keys = sorted(k for k, a in obs_meta.items() if a.get("type", "low_dim") == "low_dim")
        for k, a in obs_meta.items():
            if a.get("type", "low_dim") not in ("pcd", "low_dim"):
                raise RuntimeError(f"Unsupported obs type: {a['type']}")  # pcd_obs_encoder.py:67
        # pcd_obs_encoder.py:38-63: one shared model under the key "pcd", or one model per point-cloud key (given as a dict,
        # or deep copies of the one module)
        key_model_map = nn.ModuleDict()
        if share_pcd_model:
            assert isinstance(pcd_model, nn.Module)
            key_model_map["pcd"] = pcd_model
        else:
            import copy

            for k in obs_meta:  # insertion order of shape_meta, like the reference's loop
                if obs_meta[k].get("type", "low_dim") != "pcd":
                    continue
                if isinstance(pcd_model, dict):
                    key_model_map[k] = pcd_model[k]
                else:
                    assert isinstance(pcd_model, nn.Module)
                    key_model_map[k] = copy.deepcopy(pcd_model)
        self.key_model_map = key_model_map
        any_model = pcd_model if isinstance(pcd_model, nn.Module) else next(iter(pcd_model.values()))
        self.key_shape_map = {k: tuple(a["shape"]) for k, a in obs_meta.items()}
        self.shape_meta, self.share_pcd_model, self.n_obs_step = shape_meta, share_pcd_model, n_obs_step
        self.pcd_nsample, self.pcd_npoints = pcd_nsample, pcd_npoints
        self.pre_sample = pre_sample
        if not pre_sample:
            self.linear = nn.Linear(3 + any_model.num_channels, pcd_hidden_dim, bias=False)
            self.bn = nn.BatchNorm1d(pcd_hidden_dim)
        else:
            # pcd_obs_encoder.py:91-93: the set-abstraction layer runs on the RAW features, in front of the point-cloud model
            # (configs/exp_maniskill2_diffusion_policy/maniskill2_model/scratch_pointnet_pcd_presample{,_wo_rgb,_wo_xyz}.yaml)
            self.linear = nn.Linear(3 + in_channel, in_channel, bias=False)
            self.bn = nn.BatchNorm1d(in_channel)
        self.pool = nn.MaxPool1d(pcd_nsample)
        self.relu = nn.ReLU(inplace=True)
        ch = list(projector_channels)
        proj = []
        for i in range(projector_layers):
            # :103-112: with pre_sample the projector's first convolution reads the model's output width
            cin = pcd_hidden_dim if (i > 0 or not pre_sample) else any_model.num_channels
            proj += [nn.Conv1d(cin, ch[i], kernel_size=1), nn.BatchNorm1d(ch[i]), nn.ReLU(inplace=True)]
        proj += [nn.MaxPool1d(pcd_npoints), nn.Conv1d(ch[projector_layers - 1], ch[projector_layers], kernel_size=1),
                 nn.BatchNorm1d(ch[projector_layers])]
        self.projector = nn.Sequential(*proj)
        self.projector_channels = ch
        self._out_channels = ch[projector_layers]

    @property
    def pointops(self):
        return self._pointops[0]

    def _new_offsets(self, o):
        b = int(o.shape[0])
        cache = self.__dict__.setdefault("_n_o_cache", {})
        key = (b, o.device)
        if key not in cache:
            host = [self.pcd_npoints * (i + 1) for i in range(b)]
            t = torch.tensor(host, dtype=torch.int32, device=o.device)
            t._pcm_host = host
            cache[key] = t
        return cache[key]

    def prefetch_sampling(self, pcd_dict):
        coord, offset = pcd_dict["coord"], pcd_dict["offset"]
        set_abstraction.prefetch_sampling(self, self.pointops, coord, offset, self._new_offsets(offset),
                                          mask=self._mask_of(pcd_dict))

    def sampling_for(self, pcd_dict, overlap=True):
        """FPS / kNN / index statistics of these clouds: the prefetched result if `prefetch_sampling` saw them, else
        computed now (on the side stream with `overlap`)."""
        coord, offset = pcd_dict["coord"], pcd_dict["offset"]
        return set_abstraction.sample_and_query(self, self.pointops, coord, offset, self._new_offsets(offset), overlap=overlap,
                                                mask=self._mask_of(pcd_dict))

    def _mask_of(self, pcd_dict):
        """pcd_obs_encoder.py:203-207: with ``use_mask`` the clouds must carry the per-point foreground mask."""
        return pcd_dict["mask"] if self.use_mask else None

    def install_static_sampling(self, pcd_dict, pre):
        set_abstraction.install_static(self, pcd_dict["coord"], pcd_dict["offset"], pre)

    def load_static_sampling(self, pre):
        set_abstraction.load_static(self, pre)

    def fused_batchnorms(self):
        """BatchNorm layers owned by fused kernels: the SA layer's, and the projector's when its row-layout path applies (PROJECTOR_ROWS on
        and every layer statically eligible, `_projector_plan`) -- then no torch.nn.SyncBatchNorm module is left under data parallelism
        (`BCTrainer.all_batchnorms_fused`); otherwise the projector's BatchNorms stay with the framework (SyncBatchNorm conversion)."""
        if self.sa_impl != "fused":
            return []
        plan = self._projector_plan() if PROJECTOR_ROWS else None
        return [self.bn] + ([m for m in self.projector if isinstance(m, nn.BatchNorm1d)] if plan is not None else [])

    def _projector_plan(self):
        """The projector as a list of row-layout steps -- ("linear", conv) | ("bn", bn, relu) | ("pool",) -- or None when ANY layer does not
        qualify.  Decided from the modules alone (kernel sizes, widths), BEFORE anything runs: a BatchNorm executed by the row path has
        updated its running statistics, so finding a non-qualifying layer half way and re-running the module path would update them twice
        for one batch (round-5 ADVICE)."""
        from . import bn_relu as fused

        layers, steps, C, pooled, i = list(self.projector), [], None, False, 0
        while i < len(layers):
            layer = layers[i]
            if isinstance(layer, nn.Conv1d):
                if layer.kernel_size[0] != 1 or layer.stride[0] != 1 or layer.padding[0] != 0 or layer.groups != 1 or (C is not None and layer.in_channels != C):
                    return None
                C = layer.out_channels
                steps.append(("linear", layer))
            elif isinstance(layer, nn.BatchNorm1d):
                relu = i + 1 < len(layers) and isinstance(layers[i + 1], nn.ReLU)
                if C is None or not fused.supported_layer(layer, C):
                    return None
                steps.append(("bn", layer, relu))
                i += 1 if relu else 0
            elif isinstance(layer, nn.MaxPool1d):
                if pooled or not (layer.kernel_size == self.pcd_npoints and layer.stride == layer.kernel_size and layer.padding == 0
                                  and layer.dilation == 1 and not layer.ceil_mode):
                    return None
                pooled = True
                steps.append(("pool",))
            else:
                return None
            i += 1
        return steps

    def _projector_rows(self, x):
        """The projector (pcd_obs_encoder.py:100-120: [Conv1d(k=1) -> BatchNorm1d -> ReLU] x layers -> MaxPool1d(M) -> Conv1d(k=1) ->
        BatchNorm1d) in ROW layout: tokens stay (b * M, C) -- a 1x1 convolution is a per-row product, BatchNorm1d over (b, C, M) is a
        BatchNorm over the b * M rows, the pool a maximum over each cloud's M rows -- so the BatchNorms run in csrc/bnrelu.hip (the last
        one without its ReLU: csrc/bnact.hip, pcm_bn_act_*), exchange their statistics themselves when synchronised, and the transposes
        disappear.  Returns (b, C_out), or None when the input or a layer does not qualify (host tensors, odd widths: the module path) --
        decided before any layer runs."""
        from . import bn_relu as fused

        if self.sa_impl != "fused" or not x.is_cuda or x.dim() != 2 or x.shape[0] % self.pcd_npoints or x.shape[0] == 0 \
                or x.dtype not in (torch.float32, torch.bfloat16):
            return None
        steps = self._projector_plan()
        if steps is None or steps[0][0] != "linear" or steps[0][1].in_channels != x.shape[1]:
            return None
        for step in steps:
            if step[0] == "linear":
                x = linear_rows(x, step[1].weight[:, :, 0], step[1].bias)
            elif step[0] == "bn":
                x = fused.bn_relu(x, step[1], relu=step[2])
            else:  # .max(dim): like the pooling kernel the gradient goes to ONE position, the first maximum
                x = x.view(-1, self.pcd_npoints, x.shape[-1]).max(dim=1).values
        return x

    def pcd_sampling(self, pxo, mask=None, return_index=False):
        """pcd_obs_encoder.py:123-198: (p (n,3), x (n,c), o (b)) -> x (m,H), or (n_p, x, n_o, idx) with `return_index`."""
        p, x, o = pxo
        n_o = self._new_offsets(o)
        pre = set_abstraction.sample_and_query(self, self.pointops, p, o, n_o, mask=mask)
        n_p, feat, idx = set_abstraction(self, self.pointops, p, x, o, n_o, impl=self.sa_impl, pre=pre)
        if return_index:
            return n_p, feat, n_o, idx
        return feat

    def sa_tokens(self, pcd_model, pcd_dict):
        """The ragged half: PointNet + set abstraction -> (b*M, C) tokens (fixed shape whatever the cloud sizes)."""
        coord, offset = pcd_dict["coord"], pcd_dict["offset"]
        n_o = self._new_offsets(offset)
        pre = set_abstraction.sample_and_query(self, self.pointops, coord, offset, n_o,
                                               overlap=self.overlap_sampling and coord.is_cuda, mask=self._mask_of(pcd_dict))
        if self.pre_sample:
            # pcd_obs_encoder.py:201-218: sample on the raw features, rewrite the cloud dict to the sampled cloud (as the
            # reference does), then the point-cloud model on the m sampled points
            n_p, feat, fps_idx = set_abstraction(self, self.pointops, coord, pcd_dict["feat"], offset, n_o, impl=self.sa_impl, pre=pre)
            pcd_dict["coord"], pcd_dict["feat"], pcd_dict["offset"] = n_p, feat, n_o
            pcd_dict["grid_coord"] = pcd_dict["grid_coord"][fps_idx.long()]
            return pcd_model(pcd_dict)
        features = pcd_model(pcd_dict)
        return set_abstraction(self, self.pointops, coord, features, offset, n_o, impl=self.sa_impl, pre=pre)[1]

    tokenizer_fp32 = True  # policy/precision.py: PointNet + SA layer + projector stay in fp32 under bf16 autocast

    def tokenizer_modules(self):
        return [self]  # everything this encoder owns (point-cloud models, SA linear / bn, projector)

    def encode_pcd(self, pcd_model, pcd_dict):
        from .precision import tokenizer_autocast

        with tokenizer_autocast(self, pcd_dict["coord"] if "coord" in pcd_dict else pcd_dict["sa_tokens"]):
            return self._encode_pcd(pcd_model, pcd_dict)

    def _encode_pcd(self, pcd_model, pcd_dict):
        x = pcd_dict["sa_tokens"] if "sa_tokens" in pcd_dict else self.sa_tokens(pcd_model, pcd_dict)
        if PROJECTOR_ROWS and self.training:
            rows = self._projector_rows(x)
            if rows is not None:
                return rows
        if self.training and any(getattr(m, "_pcm_sync", False) for m in self.projector if isinstance(m, nn.BatchNorm1d)):
            raise RuntimeError("the projector's BatchNorms are marked for synchronised statistics, which only the fused row-layout path "
                               "implements (policy/sync_bn.py); its input does not qualify for that path")
        x = x.view(-1, self.pcd_npoints, x.shape[-1]).transpose(1, 2)  # "(b n) c -> b c n"
        for layer in self.projector:  # 1x1 convolutions as GEMMs (MIOpen falls back to naive bf16 kernels here)
            if isinstance(layer, nn.Conv1d):
                x = conv1d_gemm(x, layer)
            elif isinstance(layer, nn.MaxPool1d) and x.is_cuda and layer.kernel_size == x.shape[-1] and layer.stride == layer.kernel_size \
                    and layer.padding == 0 and layer.dilation == 1 and not layer.ceil_mode:
                # MaxPool1d over the whole token axis == a row maximum: the framework's max_pool_forward_nchw walks the
                # 2048-wide window with one thread per output (282 us at C5 against 9 us for the reduction kernel)
                # (.max(dim), not amax: like the pooling kernel it routes the gradient to ONE position -- the first maximum --
                # where amax would split it among bf16 ties)
                x = x.max(dim=-1, keepdim=True).values
            else:
                x = layer(x.contiguous() if isinstance(layer, nn.BatchNorm1d) else x)
        return x.squeeze(-1)

    def _model_of(self, key):
        return self.key_model_map["pcd"] if self.share_pcd_model else self.key_model_map[key]

    def pcd_features(self, pcd_dict, key=None):
        """packed clouds -> (b, C) features: the eager half of the hybrid trainer mode."""
        return self.encode_pcd(self._model_of(key if key is not None else self.pcd_keys[0]), pcd_dict)

    def forward(self, obs_dict):
        feats, batch = [], None
        for key in self.pcd_keys:
            pcd = obs_dict[key]
            if "pcd_feat" in pcd:  # the cloud features computed by an earlier stage (BCTrainer mode="hybrid")
                batch = pcd["pcd_feat"].shape[0]
                feats.append(pcd["pcd_feat"].reshape(batch, -1))
                continue
            if "sa_tokens" in pcd:
                batch = pcd["sa_tokens"].shape[0] // self.pcd_npoints
            else:
                assert len(pcd["offset"]) % self.n_obs_step == 0
                batch = len(pcd["offset"])
                assert pcd["feat"].shape[1:] == self.key_shape_map[key]
            feats.append(self.encode_pcd(self._model_of(key), pcd).reshape(batch, -1))
        for key in self.low_dim_keys:
            data = obs_dict[key]
            assert batch is None or batch == data.shape[0], (key, batch, data.shape)
            batch = data.shape[0]
            feats.append(data)
        return torch.cat(feats, dim=-1)

    def output_shape(self):
        return (self._out_channels + sum(self.key_shape_map[k][0] for k in self.low_dim_keys),)


# ----------------------------------------------------------------------------- the policy
class DiffusionUnetPcdPolicy(_AttrMixin):
    """Training-side counterpart of DiffusionUnetImagePolicy with a PCDObsEncoder."""

    def __init__(self, shape_meta, noise_scheduler, obs_encoder, horizon, n_action_steps, n_obs_steps,
                 num_inference_steps=None, obs_as_global_cond=True, diffusion_step_embed_dim=256,
                 down_dims=(256, 512, 1024), kernel_size=5, n_groups=8, cond_predict_scale=True, **kwargs):
        super().__init__()
        if not obs_as_global_cond:
            raise NotImplementedError("obs_as_global_cond=False is not used by any point-cloud config")
        action_dim = shape_meta["action"]["shape"][0]
        feat_dim = obs_encoder.output_shape()[0]
        global_cond_dim = feat_dim * n_obs_steps
        goal_meta = shape_meta.get("goal")
        if goal_meta is not None:
            # diffusion_unet_image_policy.py:58-68: a task embedding (goal_pos in the PickCube / Fill / Hang / Excavate
            # configs) widens the global condition; it is concatenated in compute_loss AND predict_action (:197-201, :262-266)
            if "task_emb" not in goal_meta:
                raise NotImplementedError("image goals (agentview_rgb / depth) belong to the image policy, not the point-cloud path")
            global_cond_dim += int(goal_meta["task_emb"]["shape"][0])
        self.goal_dim = global_cond_dim - feat_dim * n_obs_steps
        self.model = ConditionalUnet1D(input_dim=action_dim, local_cond_dim=None, global_cond_dim=global_cond_dim,
                                       diffusion_step_embed_dim=diffusion_step_embed_dim, down_dims=down_dims,
                                       kernel_size=kernel_size, n_groups=n_groups, cond_predict_scale=cond_predict_scale)
        self.obs_encoder = obs_encoder
        self.noise_scheduler = noise_scheduler
        self.mask_generator = LowdimMaskGenerator(action_dim=action_dim, obs_dim=0, max_n_obs_steps=n_obs_steps,
                                                  fix_obs_steps=True, action_visible=False)
        self.normalizer = LinearNormalizer()
        self.horizon, self.obs_feature_dim, self.action_dim = horizon, feat_dim, action_dim
        self.n_action_steps, self.n_obs_steps = n_action_steps, n_obs_steps
        self.num_inference_steps = num_inference_steps or noise_scheduler.num_train_timesteps

    def _with_goal(self, global_cond, goal):
        """Append goal["task_emb"] exactly when the model was built with one (shape_meta["goal"]); anything else is a
        configuration error caught here instead of as a shape mismatch inside the U-Net's condition encoder."""
        has = goal is not None and "task_emb" in goal
        if has != (self.goal_dim > 0):
            raise ValueError("goal conditioning mismatch: model built with goal_dim=%d, batch %s a task_emb"
                             % (self.goal_dim, "carries" if has else "lacks"))
        if not has:
            return global_cond
        emb = goal["task_emb"].to(global_cond.dtype)
        return torch.cat([global_cond, emb.reshape(global_cond.shape[0], -1)], dim=-1)

    def set_normalizer(self, normalizer):
        self.normalizer.load_state_dict(normalizer.state_dict())

    def reset(self):  # the reference module calls policy.reset() at the start of each rollout episode
        pass

    # ---- inference (diffusion_unet_image_policy.py:106-229)
    def conditional_sample(self, condition_data, condition_mask, local_cond=None, global_cond=None, generator=None,
                           noises=None, **kwargs):
        """DDPM ancestral sampling of an action trajectory.  `noises` (optional list: initial trajectory, then one
        variance-noise tensor per iteration) injects the random draws for parity tests."""
        sched = self.noise_scheduler
        if sched.num_inference_steps != self.num_inference_steps:
            sched.set_timesteps(self.num_inference_steps)
        dev = condition_data.device
        if noises is not None:
            trajectory = noises[0].to(dev, torch.float32).clone()
        else:
            trajectory = torch.randn(condition_data.shape, dtype=torch.float32, device=dev, generator=generator)
        mask = condition_mask  # None = nothing is conditioned (no host sync to find out)
        if mask is not None:
            trajectory = torch.where(mask, condition_data, trajectory)
        t_dev = sched.timesteps_on(dev)
        for i, t in enumerate(sched.timesteps):
            eps = self.model(trajectory, t_dev[i : i + 1], local_cond=local_cond, global_cond=global_cond)
            noise = noises[1 + i].to(dev, torch.float32) if noises is not None else None
            # conditioning is re-imposed inside the same launch (the reference does it at the top of the next iteration)
            trajectory = sched.step(eps, t, trajectory, noise=noise, generator=generator, cond_mask=mask, cond=condition_data)
        return trajectory

    @torch.no_grad()
    def predict_action(self, obs_dict, noises=None, generator=None):
        """obs_dict = {"obs": {"pcds": packed clouds (B*To), <low-dim keys> (B, To, d)}} or the flat form; returns
        {"action": (B, n_action_steps, Da), "action_pred": (B, T, Da)}."""
        assert "past_action" not in obs_dict
        obs = dict(obs_dict["obs"]) if "obs" in obs_dict else {k: v for k, v in obs_dict.items() if k != "goal"}
        pcds = obs.pop("pcds", None)
        nobs = self.normalizer.normalize(obs)
        B = next(iter(nobs.values())).shape[0] if nobs else len(pcds["offset"]) // self.n_obs_steps
        To, T, Da = self.n_obs_steps, self.horizon, self.action_dim
        this_nobs = {k: v[:, :To].reshape(-1, *v.shape[2:]) for k, v in nobs.items()}
        if pcds is not None:
            this_nobs["pcds"] = pcds
        global_cond = self.obs_encoder(this_nobs).reshape(B, -1)
        global_cond = self._with_goal(global_cond, obs_dict.get("goal", None))
        dev = global_cond.device
        # obs_as_global_cond: the reference's condition mask is all False (diffusion_unet_image_policy.py:196-198)
        cond_data = torch.zeros(B, T, Da, device=dev, dtype=torch.float32)
        nsample = self.conditional_sample(cond_data, None, global_cond=global_cond, generator=generator, noises=noises)
        action_pred = self.normalizer["action"].unnormalize(nsample[..., :Da])
        start = To - 1
        return {"action": action_pred[:, start:start + self.n_action_steps], "action_pred": action_pred}

    def compute_loss(self, batch):
        """batch = {"obs": {"pcds": packed clouds (B*To of them), <low-dim keys> (B, T, d)}, "action": (B, T, Da)}.
        Optional "noise" / "timesteps" entries inject the random draws (parity tests)."""
        obs = dict(batch["obs"])
        pcds = obs.pop("pcds", None)
        nobs = self.normalizer.normalize(obs)  # point clouds are not normalised
        nactions = self.normalizer["action"].normalize(batch["action"])
        bsz = nactions.shape[0]
        this_nobs = {k: v[:, : self.n_obs_steps].reshape(-1, *v.shape[2:]) for k, v in nobs.items()}
        if pcds is not None:
            this_nobs["pcds"] = pcds
        global_cond = self._with_goal(self.obs_encoder(this_nobs).reshape(bsz, -1), batch.get("goal", None))
        from . import staging

        global_cond = staging.cut("unet.in", global_cond)
        trajectory = nactions
        cond_mask = self.mask_generator(trajectory.shape, device=trajectory.device)  # all False for obs_dim == 0
        noise = batch.get("noise", None)
        if noise is None:
            noise = torch.randn(trajectory.shape, device=trajectory.device)
        timesteps = batch.get("timesteps", None)
        if timesteps is None:
            timesteps = torch.randint(0, self.noise_scheduler.num_train_timesteps, (bsz,), device=trajectory.device).long()
        noisy = self.noise_scheduler.add_noise(trajectory, noise, timesteps)
        noisy = torch.where(cond_mask, trajectory, noisy)
        pred = self.model(noisy, timesteps, local_cond=None, global_cond=global_cond)
        target = noise if self.noise_scheduler.prediction_type == "epsilon" else trajectory
        loss = F.mse_loss(pred.float(), target, reduction="none") * (~cond_mask).to(pred.dtype if pred.dtype == torch.float32 else torch.float32)
        loss = loss.reshape(bsz, -1).mean(dim=1).mean()
        return dict(loss=loss)

    def backward_stages(self):
        """See ACTPCD.backward_stages: U-Net up path (+ final conv) | middle | down path and condition encoders | observation
        encoder -- the 1 GB gradient of the 255.6 M-parameter U-Net leaves in three slabs while backward continues."""
        m = self.model
        up = list(m.final_conv.parameters()) + list(m.up_modules.parameters())
        mid = list(m.mid_modules.parameters())
        seen = {id(p) for p in up + mid}
        down = [p for p in m.parameters() if id(p) not in seen]
        seen |= {id(p) for p in down}
        rest = [p for p in self.parameters() if id(p) not in seen]
        return [("unet.up", up), ("unet.mid", mid), ("unet.in", down), (None, rest)]

    def tokenizer_parameters(self):
        """The eager half of mode="hybrid": the whole observation encoder (PointNet, SA layer, projector) -- every
        BatchNorm of the policy lives here, outside the captured graphs, so synchronised statistics stay plain collectives."""
        return list(self.obs_encoder.parameters())

    @staticmethod
    def hybrid_split(batch):
        obs = batch["obs"]
        rest = dict(batch, obs={k: v for k, v in obs.items() if k != "pcds"})
        return {"obs": {"pcds": obs["pcds"]}}, rest

    @staticmethod
    def hybrid_merge(rest, boundary):
        return dict(rest, obs=dict(rest["obs"], pcds={"pcd_feat": boundary[0]}))

    def forward(self, batch, stage=None):
        if stage == "tokenize":  # packed clouds -> per-cloud features (b, C): the part whose shapes follow the cloud sizes
            return (self.obs_encoder.pcd_features(batch["obs"]["pcds"]),)
        out = self.compute_loss(batch)
        out.setdefault("action_loss", out["loss"])
        out.setdefault("kl_loss", out["loss"].new_zeros(()))
        return out
