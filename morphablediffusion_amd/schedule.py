"""DDPM schedule + DDIM tables on the host (pure scalar plumbing, identical arithmetic to the reference:
ldm/models/diffusion/morphable_diffusion.py:428-450 and :658-672; ldm/modules/diffusionmodules/util.py:46-60)."""
import numpy as np
import torch


def make_ddim_timesteps(num_ddim_timesteps, num_ddpm_timesteps=1000, method="uniform"):
    if method != "uniform":
        raise NotImplementedError(f'There is no ddim discretization method called "{method}"')
    c = num_ddpm_timesteps // num_ddim_timesteps
    return np.asarray(list(range(0, num_ddpm_timesteps, c))) + 1


class DDIMSchedule:
    def __init__(self, ddim_num_steps=50, ddim_eta=1.0, num_timesteps=1000, linear_start=0.00085, linear_end=0.0120):
        betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, num_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.ddim_timesteps = make_ddim_timesteps(ddim_num_steps, num_timesteps)
        ts = torch.from_numpy(self.ddim_timesteps.astype(np.int64))
        ac = self.alphas_cumprod
        a = ac[ts].double()
        a_prev = torch.cat([ac[0:1], ac[ts[:-1]]], 0)
        sig = ddim_eta * torch.sqrt((1 - a_prev) / (1 - a) * (1 - a / a_prev))
        self.ddim_alphas = a.float()
        self.ddim_alphas_prev = a_prev.float()
        self.ddim_sigmas = sig.float()
        self.ddim_sqrt_one_minus_alphas = torch.sqrt(1.0 - self.ddim_alphas).float()
        self.eta = ddim_eta

    def coefficients(self, index):
        """(sqrt(1-a_t), sqrt(a_t), sqrt(a_prev), sqrt(clamp(1-a_prev-sigma^2, 1e-7)), sigma) as fp32 scalars,
        evaluated with the same fp32 tensor ops as denoise_apply_impl (morphable_diffusion.py:687-694)."""
        a_t = self.ddim_alphas[index]
        a_prev = self.ddim_alphas_prev[index]
        sig = self.ddim_sigmas[index]
        s1m = self.ddim_sqrt_one_minus_alphas[index]
        dir_coef = torch.clamp(1.0 - a_prev - sig ** 2, min=1e-7).sqrt()
        return (float(s1m), float(a_t.sqrt()), float(a_prev.sqrt()), float(dir_coef), float(sig))
