"""One SRADSGAN training iteration on the HIP path: the body of the reference's batch loop
(SRADSGAN/model/sradsgan.py:829-892) with identical arithmetic and update order.

Differences from the reference that do not change results (DESIGN.md "restructured, same numbers"):
  * during the G step the discriminator / VGG parameters do not require grad, so the weight
    gradients the reference computes and then throws away (:857 -> :865) are never computed;
  * the gradient penalty's double backward runs once with weight (1 + lambda_gp) instead of twice
    (once inside gradient_penalty() :639, once inside loss_D.backward() :886) -- same sum;
  * losses stay on the device; nothing calls .item() inside the step (the reference syncs 4x, :898).
"""
import torch

from . import ops


class TrainStep:
    def __init__(self, generator, discriminator, feature_extractor, lr=2e-4, b1=0.9, b2=0.999,
                 weight_content=1e-2, weight_gan=1e-3, lambda_gp=10.0, clip_value=0.01, use_gp=True,
                 grad_sync=None):
        self.G, self.D, self.F = generator, discriminator, feature_extractor
        self.weight_content, self.weight_gan = weight_content, weight_gan
        self.lambda_gp, self.clip_value, self.use_gp = lambda_gp, clip_value, use_gp
        self.opt_G = torch.optim.Adam(self.G.parameters(), lr=lr, betas=(b1, b2))      # sradsgan.py:724
        self.opt_D = torch.optim.Adam(self.D.parameters(), lr=lr, betas=(b1, b2))      # sradsgan.py:725
        self.grad_sync = grad_sync            # data-parallel hook: callable(list_of_params)
        self._d_params = [p for p in self.D.parameters()]
        self._g_params = [p for p in self.G.parameters()]
        for p in self.F.parameters():
            p.requires_grad_(False)           # never in an optimiser (sradsgan.py:724-725)

    def _set_d_grad(self, flag):
        for p in self._d_params:
            p.requires_grad_(flag)

    def gradient_penalty(self, real, fake, alpha):
        """sradsgan.py:595-641 ('L2' norm over channels => per-pixel, 'LS' penalty); returns the
        penalty with its double-backward graph attached (the caller backpropagates it)."""
        interp = (alpha * real + (1 - alpha) * fake).requires_grad_(True)
        d_out = self.D(interp)
        with ops.no_param_grads():
            (grads,) = torch.autograd.grad(d_out, interp, torch.ones_like(d_out), create_graph=True)
        return ops.gp_penalty(grads)

    def __call__(self, imgs_lr, imgs_hr, alpha):
        G, D, F = self.G, self.D, self.F
        # ------------------ generator (sradsgan.py:829-858) ------------------
        self._set_d_grad(False)
        self.opt_G.zero_grad(set_to_none=True)
        gen_hr = G(imgs_lr)
        pixel = ops.l1_mean(gen_hr, imgs_hr)
        with torch.no_grad():
            real_feat = F(imgs_hr)
        content = ops.l1_mean(F(gen_hr), real_feat)
        loss_gan = -D(gen_hr).mean()
        loss_G = pixel + self.weight_content * content + self.weight_gan * loss_gan
        loss_G.backward()
        if self.grad_sync is not None:
            self.grad_sync(self._g_params)
        self.opt_G.step()
        # ---------------- discriminator (sradsgan.py:865-892) ----------------
        self._set_d_grad(True)
        self.opt_D.zero_grad(set_to_none=True)
        fake = gen_hr.detach()
        loss_D = -D(imgs_hr).mean() + D(fake).mean()
        if self.use_gp:
            gp = self.gradient_penalty(imgs_hr, fake, alpha)
            total = loss_D + (1.0 + self.lambda_gp) * gp          # :639 + :884-886 => 1 + lambda
            loss_D = loss_D + self.lambda_gp * gp
        else:
            gp = torch.zeros((), device=imgs_hr.device)
            total = loss_D
        total.backward()
        if self.grad_sync is not None:
            self.grad_sync(self._d_params)
        self.opt_D.step()
        with torch.no_grad():
            torch._foreach_clamp_min_(self._d_params, -self.clip_value)                # :891-892
            torch._foreach_clamp_max_(self._d_params, self.clip_value)
        ops.bump_weight_epoch()
        return dict(loss_G=loss_G.detach(), loss_D=loss_D.detach(), pixel=pixel.detach(),
                    content=content.detach(), loss_gan=loss_gan.detach(), gp=gp.detach(), gen_hr=fake)
