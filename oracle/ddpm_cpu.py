"""CPU restatement of the reverse-diffusion update used by the Diffusion-Policy sampler.  TEST INFRASTRUCTURE ONLY.

The algorithm lives in a third-party dependency that is absent from /root/reference: diffusers==0.29.0
(requirements.txt:31) `DDPMScheduler` (`scheduling_ddpm.py`: `betas_for_alpha_bar`, `set_timesteps`, `step`,
`_get_variance`), configured by configs/model/maniskill2_diffusion_policy_model.yaml:30-38 (100 train steps,
squaredcos_cap_v2, epsilon prediction, fixed_small variance, clip_sample with range 1) and called from
src/models/components/diffusion_policy/diffusion_unet_image_policy.py:106-146.  diffusers is not installed in this
image and the reference holds no golden vector for the sampler => **parity unpinned**: this file restates the
published algorithm in numpy fp32 and the GPU kernel (csrc/ddpm.hip) is tested against it bit for bit.
"""
import math

import numpy as np

f32 = np.float32


def betas_squaredcos_cap_v2(n, max_beta=0.999):
    """betas_for_alpha_bar: beta_i = min(1 - abar((i+1)/n)/abar(i/n), max_beta), abar(u)=cos^2((u+.008)/1.008*pi/2);
    python floats (fp64) then cast to fp32."""
    abar = lambda u: math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2
    return np.array([min(1 - abar((i + 1) / n) / abar(i / n), max_beta) for i in range(n)], dtype=f32)


def alphas_cumprod(n):
    """torch.cumprod(1 - betas) on CPU fp32 tensors, as diffusers computes it at construction: ATen accumulates fp32
    CPU scans in double (acc_type<float> = double on the host) and rounds each prefix to fp32."""
    alphas = f32(1.0) - betas_squaredcos_cap_v2(n)
    return np.cumprod(alphas.astype(np.float64)).astype(f32)


def timesteps(num_train, num_inference):
    """'leading' spacing: (arange(num_inference) * (num_train // num_inference)).round()[::-1]."""
    ratio = num_train // num_inference
    return (np.arange(0, num_inference) * ratio).round()[::-1].astype(np.int64)


def step_coefficients(ac, t, num_inference):
    """fp32 scalars of DDPMScheduler.step for timestep t: (sqrt(abar_t), sqrt(1-abar_t), coef_x0, coef_xt, sigma)."""
    n = ac.shape[0]
    prev_t = t - n // num_inference
    a_t = ac[t]
    a_prev = ac[prev_t] if prev_t >= 0 else f32(1.0)
    b_t = f32(1.0) - a_t
    b_prev = f32(1.0) - a_prev
    cur_a = f32(a_t / a_prev)
    cur_b = f32(1.0) - cur_a
    coef_x0 = f32(f32(np.sqrt(a_prev)) * cur_b) / b_t
    coef_xt = f32(f32(np.sqrt(cur_a)) * b_prev) / b_t
    sigma = f32(0.0)
    if t > 0:
        var = f32(f32(b_prev / b_t) * cur_b)  # fixed_small
        var = max(var, f32(1e-20))
        sigma = f32(np.sqrt(var))
    return f32(np.sqrt(a_t)), f32(np.sqrt(b_t)), f32(coef_x0), f32(coef_xt), sigma


def ddpm_step(eps, xt, noise, coef, clip=1.0, cond_mask=None, cond=None):
    """One x_t -> x_{t-1} update in fp32, un-fused, in the order of DDPMScheduler.step."""
    sa, sb, c0, ct, sigma = coef
    eps = eps.astype(f32)
    xt = xt.astype(f32)
    x0 = (xt - sb * eps) / sa
    if clip and clip > 0:
        x0 = np.clip(x0, f32(-clip), f32(clip))
    prev = c0 * x0 + ct * xt
    if noise is not None and sigma != 0:
        prev = prev + sigma * noise.astype(f32)
    if cond_mask is not None:
        prev = np.where(cond_mask, cond.astype(f32), prev)
    return prev.astype(f32)


def sample(model, shape, noises, num_train=100, num_inference=100, clip=1.0, cond_mask=None, cond=None):
    """conditional_sample: `model(x, t) -> eps`; noises[0] is the initial trajectory, noises[1+i] the variance noise of
    iteration i.  Returns the final trajectory."""
    ac = alphas_cumprod(num_train)
    x = noises[0].astype(f32)
    for i, t in enumerate(timesteps(num_train, num_inference)):
        if cond_mask is not None:
            x = np.where(cond_mask, cond, x)
        eps = model(x, int(t))
        x = ddpm_step(eps, x, noises[1 + i], step_coefficients(ac, int(t), num_inference), clip)
    if cond_mask is not None:
        x = np.where(cond_mask, cond, x)
    return x
