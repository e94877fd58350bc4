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

import torch
import torch.nn as nn
import torch.nn.functional as F

from .sa_layer import set_abstraction


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
        y = F.linear(x.transpose(1, 2), w[:, :, 0], conv.bias)  # (B, L, C_out)
        return y.transpose(1, 2)
    xp = F.pad(x, (pad, pad)) if pad else x
    cols = xp.unfold(2, k, stride)  # (B, C_in, L_out, K) view
    lout = cols.shape[2]
    cols = cols.permute(0, 2, 1, 3).reshape(b * lout, cin * k)  # the one im2col copy
    y = F.linear(cols, w.reshape(w.shape[0], cin * k), conv.bias)
    return y.view(b, lout, -1).transpose(1, 2)


def conv_transpose1d_gemm(x, conv):
    """``conv(x)`` for ``nn.ConvTranspose1d(C, C, kernel_size=4, stride=2, padding=1)`` (the U-Net's Upsample1d):
    one GEMM producing the (B, L, C_out, 4) taps, then the overlap-add of the two tap pairs."""
    assert conv.kernel_size[0] == 4 and conv.stride[0] == 2 and conv.padding[0] == 1 and conv.output_padding[0] == 0
    b, cin, l = x.shape
    w = conv.weight  # (C_in, C_out, 4)
    cout = w.shape[1]
    taps = F.linear(x.transpose(1, 2).reshape(b * l, cin), w.reshape(cin, cout * 4).t()).view(b, l, cout, 4)
    # full[o], o = 2*l + k: taps 0,1 land on rows (l, 0..1), taps 2,3 on rows (l+1, 0..1) of a (L+1, 2) grid
    lo = F.pad(taps[..., 0:2], (0, 0, 0, 0, 0, 1))
    hi = F.pad(taps[..., 2:4], (0, 0, 0, 0, 1, 0))
    full = (lo + hi).permute(0, 2, 1, 3).reshape(b, cout, 2 * l + 2)  # (B, C_out, 2L+2)
    y = full[:, :, 1:-1]
    return y + conv.bias[None, :, None] if conv.bias is not None else y


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

    def forward(self, x):
        return conv1d_gemm(x, self.conv)


class Upsample1d(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.conv = nn.ConvTranspose1d(dim, dim, 4, 2, 1)

    def forward(self, x):
        return conv_transpose1d_gemm(x, self.conv)


class Conv1dBlock(nn.Module):
    """Conv1d -> GroupNorm -> Mish."""

    def __init__(self, inp_channels, out_channels, kernel_size, n_groups=8):
        super().__init__()
        self.block = nn.Sequential(
            nn.Conv1d(inp_channels, out_channels, kernel_size, padding=kernel_size // 2),
            nn.GroupNorm(n_groups, out_channels),
            nn.Mish(),
        )

    def forward(self, x):
        conv, norm, act = self.block[0], self.block[1], self.block[2]
        return act(norm(conv1d_gemm(x, conv).contiguous()))


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
        self.cond_encoder = nn.Sequential(nn.Mish(), nn.Linear(cond_dim, film), _AddTrailingDim())
        self.residual_conv = nn.Conv1d(in_channels, out_channels, 1) if in_channels != out_channels else nn.Identity()

    def forward(self, x, cond):
        out = self.blocks[0](x)
        embed = self.cond_encoder(cond)
        if self.cond_predict_scale:
            embed = embed.reshape(embed.shape[0], 2, self.out_channels, 1)
            out = embed[:, 0] * out + embed[:, 1]
        else:
            out = out + embed
        out = self.blocks[1](out)
        res = conv1d_gemm(x, self.residual_conv) if isinstance(self.residual_conv, nn.Conv1d) else x
        return out + res


class ConditionalUnet1D(nn.Module):
    def __init__(self, input_dim, local_cond_dim=None, global_cond_dim=None, diffusion_step_embed_dim=256,
                 down_dims=(256, 512, 1024), kernel_size=3, n_groups=8, cond_predict_scale=False):
        super().__init__()
        if local_cond_dim is not None:
            raise NotImplementedError("local conditioning is not used by any point-cloud config")
        dims = [input_dim] + list(down_dims)
        dsed = diffusion_step_embed_dim
        self.diffusion_step_encoder = nn.Sequential(
            SinusoidalPosEmb(dsed), nn.Linear(dsed, dsed * 4), nn.Mish(), nn.Linear(dsed * 4, dsed))
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
        """sample (B, T, input_dim), timestep (B,) -> (B, T, input_dim)."""
        x = sample.transpose(1, 2)  # (B, C, T)
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.long, device=sample.device)
        elif t.dim() == 0:
            t = t[None].to(sample.device)
        t = t.expand(sample.shape[0])
        cond = self.diffusion_step_encoder(t)
        if global_cond is not None:
            cond = torch.cat([cond, global_cond], dim=-1)
        skips = []
        for res1, res2, down in self.down_modules:
            x = res2(res1(x, cond), cond)
            skips.append(x)
            x = down(x)
        for mid in self.mid_modules:
            x = mid(x, cond)
        for res1, res2, up in self.up_modules:
            x = torch.cat((x, skips.pop()), dim=1)
            x = up(res2(res1(x, cond), cond))
        x = self.final_conv[0](x)
        return conv1d_gemm(x, self.final_conv[1]).transpose(1, 2)


# ----------------------------------------------------------------------------- DDPM forward process
class DDPMSchedule(nn.Module):
    """squaredcos_cap_v2 betas (alpha_bar(t) = cos^2((t+0.008)/1.008 * pi/2), beta capped at 0.999),
    float64 on the host then float32, and x_t = sqrt(abar_t) x_0 + sqrt(1-abar_t) eps."""

    def __init__(self, num_train_timesteps=100, beta_schedule="squaredcos_cap_v2", prediction_type="epsilon",
                 beta_start=0.0001, beta_end=0.02, **unused):
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
        self.num_train_timesteps = n
        self.prediction_type = prediction_type
        self.register_buffer("alphas_cumprod", torch.cumprod(1.0 - betas, dim=0), persistent=False)

    def add_noise(self, original, noise, timesteps):
        ac = self.alphas_cumprod.to(device=original.device, dtype=original.dtype)[timesteps]
        a = ac.sqrt().reshape(-1, *([1] * (original.dim() - 1)))
        s = (1 - ac).sqrt().reshape(-1, *([1] * (original.dim() - 1)))
        return a * original + s * noise


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
        if use_mask or pre_sample or not share_pcd_model:
            raise NotImplementedError("use_mask / pre_sample / per-key pcd models are not used by any shipped config")
        if pointops is None:
            from .. import pointops as _hip_pointops

            pointops = _hip_pointops
        self._pointops = [pointops]
        self.sa_impl, self.overlap_sampling = sa_impl, overlap_sampling
        self.key_model_map = nn.ModuleDict({"pcd": pcd_model})
        obs_meta = shape_meta["obs"]
        self.pcd_keys = sorted(k for k, a in obs_meta.items() if a.get("type", "low_dim") == "pcd")
        self.low_dim_keys = sorted(k for k, a in obs_meta.items() if a.get("type", "low_dim") == "low_dim")
        self.key_shape_map = {k: tuple(a["shape"]) for k, a in obs_meta.items()}
        self.shape_meta, self.share_pcd_model, self.n_obs_step = shape_meta, share_pcd_model, n_obs_step
        self.pcd_nsample, self.pcd_npoints = pcd_nsample, pcd_npoints
        self.linear = nn.Linear(3 + pcd_model.num_channels, pcd_hidden_dim, bias=False)
        self.bn = nn.BatchNorm1d(pcd_hidden_dim)
        self.pool = nn.MaxPool1d(pcd_nsample)
        self.relu = nn.ReLU(inplace=True)
        ch = list(projector_channels)
        proj = []
        for i in range(projector_layers):
            proj += [nn.Conv1d(pcd_hidden_dim, ch[i], kernel_size=1), nn.BatchNorm1d(ch[i]), nn.ReLU(inplace=True)]
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

    def encode_pcd(self, pcd_model, pcd_dict):
        coord, offset = pcd_dict["coord"], pcd_dict["offset"]
        n_o = self._new_offsets(offset)
        pre = set_abstraction.sample_and_query(self, self.pointops, coord, offset, n_o,
                                               overlap=self.overlap_sampling and coord.is_cuda)
        features = pcd_model(pcd_dict)
        _, x, _ = set_abstraction(self, self.pointops, coord, features, offset, n_o, impl=self.sa_impl, pre=pre)
        x = x.view(offset.shape[0], self.pcd_npoints, -1).transpose(1, 2)  # "(b n) c -> b c n"
        for layer in self.projector:  # 1x1 convolutions as GEMMs (MIOpen falls back to naive bf16 kernels here)
            x = conv1d_gemm(x, layer) if isinstance(layer, nn.Conv1d) else layer(x.contiguous() if isinstance(layer, nn.BatchNorm1d) else x)
        return x.squeeze(-1)

    def forward(self, obs_dict):
        feats, batch = [], None
        for key in self.pcd_keys:
            pcd = obs_dict[key]
            assert len(pcd["offset"]) % self.n_obs_step == 0
            batch = len(pcd["offset"])
            assert pcd["feat"].shape[1:] == self.key_shape_map[key]
            feats.append(self.encode_pcd(self.key_model_map["pcd"], pcd).reshape(batch, -1))
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
        self.model = ConditionalUnet1D(input_dim=action_dim, local_cond_dim=None, global_cond_dim=feat_dim * n_obs_steps,
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

    def set_normalizer(self, normalizer):
        self.normalizer.load_state_dict(normalizer.state_dict())

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
        global_cond = self.obs_encoder(this_nobs).reshape(bsz, -1)
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

    def forward(self, batch):
        out = self.compute_loss(batch)
        out.setdefault("action_loss", out["loss"])
        out.setdefault("kl_loss", out["loss"].new_zeros(()))
        return out
