"""TEST INFRASTRUCTURE (oracle): CPU restatement of the optimiser side of the reference's cached-feature training step,
with torch's own autograd / AdamW / clip_grad_norm_ as the arithmetic (train.py:326-359, perceptual_loss.py:68-100,
checkpoint_manager.py:349-356).  Only tests may import this."""
import torch


def l1_clamp_loss_and_grad(sr, hr, accumulation_steps=1):
    """sr, hr [B,C,H,W] -> (loss, d loss / d sr) of ``L1Loss()(sr.clamp(0, 1), hr) / accumulation_steps``"""
    x = sr.detach().clone().requires_grad_(True)
    loss = torch.abs(x.clamp(0, 1) - hr).mean() / accumulation_steps
    loss.backward()
    return loss.detach(), x.grad


class Trainer:
    """params: dict name -> tensor.  step(grads) = clip_grad_norm_ + AdamW.step + EMAModel.update of the reference."""

    def __init__(self, params, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4, max_norm=1.0, ema_decay=0.999):
        self.params = {k: torch.nn.Parameter(v.detach().clone().float()) for k, v in params.items()}
        self.opt = torch.optim.AdamW(list(self.params.values()), lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.max_norm, self.decay = max_norm, ema_decay
        self.shadow = {k: p.data.clone() for k, p in self.params.items()}

    def step(self, grads):
        for k, p in self.params.items():
            p.grad = grads[k].detach().clone().float()
        norm = None
        if self.max_norm and self.max_norm > 0:
            norm = torch.nn.utils.clip_grad_norm_(list(self.params.values()), self.max_norm)
        self.opt.step()
        self.opt.zero_grad()
        for k, p in self.params.items():
            self.shadow[k] = self.decay * self.shadow[k] + (1.0 - self.decay) * p.data
        return norm
