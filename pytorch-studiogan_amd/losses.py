"""Adversarial losses with the reference's names and call signature (reference src/utils/losses.py:197-239,
wired by src/config.py:411-433 as cfgs.LOSS.{d_loss,g_loss}); forward + gradient in one small kernel each."""
import torch
import torch.distributed as dist
from torch import autograd

from . import functional as F
from . import _lib as L


def d_hinge(d_logit_real, d_logit_fake, DDP=False):
    return F.DLossFn.apply(d_logit_real, d_logit_fake, 0)


def g_hinge(d_logit_fake, DDP=False):
    return F.GLossFn.apply(d_logit_fake, 0)


def d_wasserstein(d_logit_real, d_logit_fake, DDP=False):
    return F.DLossFn.apply(d_logit_real, d_logit_fake, 1)


def g_wasserstein(d_logit_fake, DDP=False):
    return F.GLossFn.apply(d_logit_fake, 1)


def d_vanilla(d_logit_real, d_logit_fake, DDP=False):
    return F.DLossFn.apply(d_logit_real, d_logit_fake, 2)


def g_vanilla(d_logit_fake, DDP=False):
    return F.GLossFn.apply(d_logit_fake, 2)


def d_logistic(d_logit_real, d_logit_fake, DDP=False):
    """reference src/utils/losses.py:207-209: mean(softplus(-r) + softplus(f)) -- the vanilla loss with the two means taken as one"""
    return F.DLossFn.apply(d_logit_real, d_logit_fake, 2)


def g_logistic(d_logit_fake, DDP=False):
    return F.GLossFn.apply(d_logit_fake, 2)


def d_ls(d_logit_real, d_logit_fake, DDP=False):
    """reference src/utils/losses.py:216-218 (LSGAN)"""
    return F.DLossFn.apply(d_logit_real, d_logit_fake, 3)


def g_ls(d_logit_fake, DDP=False):
    """reference src/utils/losses.py:221-223"""
    return F.GLossFn.apply(d_logit_fake, 3)


def latent_optimise(zs, fake_labels, generator, discriminator, batch_size, lo_rate, lo_steps, lo_alpha, lo_beta, eval, cal_trsp_cost, device):
    """Latent optimisation of LOGAN, reference src/utils/losses.py:278-298 (called from src/utils/sample.py:123-135 when LOSS.apply_lo): one natural-gradient-like
    step of z along d D(G(z)) / dz, taken WITH a graph (cal_deriv: create_graph=True) so that the transport cost and the images of the moved latents back-propagate
    through it -- the create_graph pass runs through the generator's and the discriminator's differentiable data-gradient operators (functional.LinearDgradFn,
    ConvDgradFn, BNBwdFn, TanhGradFn, ...). The arithmetic on the [B, z_dim] latents is torch's, as in the reference. Like the reference, the function returns from
    inside its loop: one step is taken whatever lo_steps says (>= 2)."""
    for step in range(lo_steps - 1):
        drop_mask = (torch.FloatTensor(batch_size, 1).uniform_() > 1 - lo_rate).to(device)
        zs = zs.detach().requires_grad_(True)
        fake_images = generator(zs, fake_labels, eval=eval)
        fake_dict = discriminator(fake_images, fake_labels, eval=eval)
        z_grads = cal_deriv(inputs=zs, outputs=fake_dict["adv_output"], device=device)
        z_grads_norm = torch.unsqueeze((z_grads.norm(2, dim=1) ** 2), dim=1)
        delta_z = lo_alpha * z_grads / (lo_beta + z_grads_norm)
        zs = torch.clamp(zs + drop_mask * delta_z, -1.0, 1.0)
        trsf_cost = (delta_z.norm(2, dim=1) ** 2).mean() if cal_trsp_cost else None
        return zs, trsf_cost
    return zs, None


def langevin_sampling(zs, z_dim, fake_labels, generator, discriminator, batch_size, langevin_rate, langevin_noise_std, langevin_decay, langevin_decay_steps,
                      langevin_steps, device):
    """Langevin dynamics on the latents at evaluation time, reference src/utils/sample.py:195-216 (RUN.langevin_sampling): `langevin_steps` steps down the energy
    -log N(z; 0, I) - D(G(z)) with Gaussian noise of covariance langevin_noise_std * I, the rate (and the noise scale) decayed every `langevin_decay_steps` steps. The
    gradient runs through the generator and the discriminator in eval mode (the reference takes it with cal_deriv, i.e. with a graph nothing ever differentiates: here
    it is a plain first-order gradient, so conditional generators -- whose batch norm has no second-order pass -- work too); the prior and the noise come from the same
    torch.distributions objects as in the reference, so a seeded run consumes the generator identically."""
    from torch.distributions import multivariate_normal as MN
    scaler = 1.0
    decaying = langevin_decay > 0 and langevin_decay_steps > 0
    loc, eye = torch.zeros(z_dim, device=device), torch.eye(z_dim, device=device)
    prior = MN.MultivariateNormal(loc=loc, covariance_matrix=eye)
    noise = MN.MultivariateNormal(loc=loc, covariance_matrix=eye * langevin_noise_std)
    for i in range(langevin_steps):
        zs = zs.detach().requires_grad_(True)
        adv = discriminator(generator(zs, fake_labels, eval=True), fake_labels, eval=True)["adv_output"]
        energy = -prior.log_prob(zs) - adv
        z_grads = autograd.grad(outputs=energy, inputs=zs, grad_outputs=torch.ones(energy.size(), device=energy.device))[0]
        zs = zs - 0.5 * langevin_rate * z_grads + (langevin_rate ** 0.5) * noise.sample([batch_size]) * scaler
        if decaying and (i + 1) % langevin_decay_steps == 0:
            langevin_rate *= langevin_decay
            scaler *= langevin_decay
    return zs


def normal_nll_loss(x, mu, var):
    """reference src/utils/losses.py:369-375 (InfoGAN's continuous codes)"""
    return F.NormalNllFn.apply(x, mu, var)


def feature_matching_loss(real_embed, fake_embed):
    """reference src/utils/losses.py:254-259 (LOSS.apply_fm, src/worker.py:588-596)"""
    return F.FeatureMatchingFn.apply(real_embed, fake_embed)


class GatherLayer(autograd.Function):
    """All-gather with a backward pass, reference src/utils/losses.py:19-37 (used under DDP by the LeCam regulariser,
    src/worker.py:396-399, and the contrastive heads): forward returns the rank-ordered concatenation of every rank's tensor
    (== torch.cat(reference GatherLayer.apply(x), dim=0)) from ONE collective into one buffer; backward hands each rank the slice of
    the upstream gradient that belongs to its own samples, exactly like the reference (no reduction across ranks)."""

    @staticmethod
    def forward(ctx, x, group=None):
        world, ctx.rank, ctx.n = dist.get_world_size(group), dist.get_rank(group), x.shape[0]
        x = x.contiguous()
        out = torch.empty((world * ctx.n,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        try:
            dist.all_gather_into_tensor(out, x, group=group)
        except (RuntimeError, NotImplementedError):      # a backend without the flat collective
            parts = [torch.empty_like(x) for _ in range(world)]
            dist.all_gather(parts, x, group=group)
            out = torch.cat(parts, dim=0)
        return out

    @staticmethod
    def backward(ctx, g):
        return g[ctx.rank * ctx.n:(ctx.rank + 1) * ctx.n].clone(), None


def gather_logits(x, group=None):
    """Logits of the global batch under data parallelism (identity in a single process)."""
    if group is None or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return x
    return GatherLayer.apply(x, group)


# reference src/config.py:411-433 (define_losses): the names LOSS.adv_loss takes
G_LOSSES = {"vanilla": g_vanilla, "logistic": g_logistic, "least_square": g_ls, "hinge": g_hinge, "wasserstein": g_wasserstein}
D_LOSSES = {"vanilla": d_vanilla, "logistic": d_logistic, "least_square": d_ls, "hinge": d_hinge, "wasserstein": d_wasserstein}


def cal_deriv(inputs, outputs, device):
    """reference src/utils/losses.py:268-275: d(sum outputs)/d(inputs) with a graph for the second-order pass."""
    grads = autograd.grad(outputs=outputs, inputs=inputs, grad_outputs=torch.ones(outputs.size(), device=outputs.device),
                          create_graph=True, retain_graph=True, only_inputs=True)[0]
    return grads


def cal_grad_penalty(real_images, real_labels, fake_images, discriminator, device):
    """WGAN-GP gradient penalty, reference src/utils/losses.py:301-316 (alpha is drawn on the host RNG like the reference
    does, so identical seeds give identical interpolates). The discriminator's backward runs once more as differentiable
    ops (functional.ConvDgradFn / BNBwdFn / ...), and penalty.backward() then walks the second-order graph."""
    batch_size = real_images.shape[0]
    alpha = torch.rand(batch_size, 1).to(real_images.device)
    interpolates = F.interpolate_rows(real_images, fake_images.detach(), alpha).requires_grad_(True)
    fake_dict = discriminator(interpolates, real_labels, eval=False)
    grads = cal_deriv(inputs=interpolates, outputs=fake_dict["adv_output"], device=device)
    return F.GradPenaltyFn.apply(grads)


def cal_maxgrad_penalty(real_images, real_labels, fake_images, discriminator, device):
    """max-gradient penalty of Lipschitz GANs, reference src/utils/losses.py:338-352: same interpolates and double backward as
    cal_grad_penalty, penalty = max_b ||grad_b||^2."""
    batch_size = real_images.shape[0]
    alpha = torch.rand(batch_size, 1).to(real_images.device)
    interpolates = F.interpolate_rows(real_images, fake_images.detach(), alpha).requires_grad_(True)
    fake_dict = discriminator(interpolates, real_labels, eval=False)
    grads = cal_deriv(inputs=interpolates, outputs=fake_dict["adv_output"], device=device)
    return F.GradPenaltyFn.apply(grads, 2)


def cal_dra_penalty(real_images, real_labels, discriminator, device):
    """DRAGAN penalty, reference src/utils/losses.py:319-335: the WGAN-GP functional at real + alpha * 0.5 * std(real) * U(0,1)
    (alpha and U drawn on the host RNG like the reference: identical seeds give identical perturbations)."""
    batch_size = real_images.shape[0]
    alpha = torch.rand(batch_size, 1, 1, 1)
    noise = torch.rand(real_images.size())
    real = real_images.detach().float().contiguous()
    n = real.numel()
    stats = torch.zeros(2, dtype=torch.float32, device=real.device)
    ones = torch.ones_like(real)
    L.call("sg_dot", L.dt(real), L.ptr(real), L.ptr(real), n, L.ptr(stats), 1.0, None, L.stream())
    L.call("sg_dot", L.dt(real), L.ptr(real), L.ptr(ones), n, L.ptr(stats) + 4, 1.0, None, L.stream())
    sq, sm = (float(v) for v in stats.cpu())
    std = max((sq - sm * sm / n) / (n - 1), 0.0) ** 0.5          # torch.Tensor.std(): unbiased, over all elements
    interpolates = (alpha * noise).to(real.device).contiguous()
    L.call("sg_axpby", L.dt(real), L.ptr(real), L.ptr(interpolates), n, 1.0, 0.5 * std, L.stream())   # y = 1 * real + (0.5 std) * y
    interpolates.requires_grad_(True)
    fake_dict = discriminator(interpolates, real_labels, eval=False)
    grads = cal_deriv(inputs=interpolates, outputs=fake_dict["adv_output"], device=device)
    return F.GradPenaltyFn.apply(grads, 0)


def lecam_reg(d_logit_real, d_logit_fake, ema):
    """reference src/utils/losses.py:262-265 (ema: ops.LeCamEMA)."""
    return F.LeCamFn.apply(d_logit_real, d_logit_fake, float(ema.D_real), float(ema.D_fake))


def cal_r1_reg(adv_output, images, device):
    """R1 regulariser, reference src/utils/losses.py:355-361: 0.5 * mean_b ||d sum(D(x)) / d x_b||^2 on the REAL batch; `images`
    must have requires_grad=True before the discriminator forward that produced `adv_output` (src/worker.py:260-261)."""
    grad_dout = cal_deriv(inputs=images, outputs=adv_output, device=device)
    return F.GradPenaltyFn.apply(grad_dout, 1)


def l2_loss(a, b):
    """torch.nn.MSELoss() as the reference's worker builds it (src/worker.py:116): the consistency terms of CR / bCR / zCR
    (src/worker.py:326-361) and the generator's latent-consistency repulsion (:601-603)."""
    return F.MseFn.apply(a, b)


def adjust_k(current_k, topk_gamma, inf_k):
    """reference src/utils/losses.py:364-366."""
    current_k = max(current_k * topk_gamma, inf_k)
    return current_k


def topk_values(d_logit_fake, k):
    """torch.topk(d_logit_fake, k).values of reference src/worker.py:565-566 (one small kernel each way)."""
    return F.TopkFn.apply(d_logit_fake, int(k))


# ---------------------------------------------------------------------------------------------------------
# class-conditioning losses (reference src/utils/losses.py:40-165,242-252; chosen by src/worker.py:141-157)
# ---------------------------------------------------------------------------------------------------------
def _gather_cat(t, group, DDP):
    """torch.cat(GatherLayer.apply(t), dim=0) of the reference under DDP; labels (no gradient) take a plain all-gather."""
    if not DDP or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return t
    if t.is_floating_point():
        return GatherLayer.apply(t, group)
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, t.contiguous(), group=group)
    return torch.cat(parts, dim=0)


class CrossEntropyLoss(torch.nn.Module):
    """reference losses.py:40-47 (ACGAN / TAC / ADC auxiliary classifier): forward(cls_output, label, **_)."""

    def forward(self, cls_output, label, **_):
        return F.ClassLossFn.apply(cls_output, label, 0)


class _EmbedProxyLoss(torch.nn.Module):
    kind = 0

    def __init__(self, num_classes, temperature, master_rank="cuda", DDP=False, m_p=0.0, group=None):
        super().__init__()
        self.num_classes, self.temperature, self.master_rank, self.DDP, self.m_p, self.group = num_classes, temperature, master_rank, DDP, m_p, group

    def forward(self, embed, proxy, label, **_):
        embed, proxy, label = (_gather_cat(t, self.group, self.DDP) for t in (embed, proxy, label))
        en = F.RowNormalizeFn.apply(embed, 1e-8)           # torch.nn.CosineSimilarity(dim=-1, eps=1e-8): x.y / (|x| |y|)
        pn = F.RowNormalizeFn.apply(proxy, 1e-8)
        S = F.MatmulNTFn.apply(en, en)                     # cosine similarity of every pair of embeddings
        p = F.RowDotFn.apply(en, pn)                       # ... of every embedding with the proxy of its own class
        return F.ContrastiveLossFn.apply(S, p, label, self.kind, self.temperature, self.m_p)


class ConditionalContrastiveLoss(_EmbedProxyLoss):
    """ContraGAN's 2C loss, reference losses.py:50-97: ConditionalContrastiveLoss(num_classes, temperature, master_rank, DDP)."""
    kind = 0

    def __init__(self, num_classes, temperature, master_rank="cuda", DDP=False, group=None):
        super().__init__(num_classes, temperature, master_rank, DDP, 0.0, group)


class Data2DataCrossEntropyLoss(_EmbedProxyLoss):
    """ReACGAN's D2D-CE loss, reference losses.py:100-165: Data2DataCrossEntropyLoss(num_classes, temperature, m_p, master_rank, DDP)."""
    kind = 1

    def __init__(self, num_classes, temperature, m_p, master_rank="cuda", DDP=False, group=None):
        super().__init__(num_classes, temperature, master_rank, DDP, m_p, group)


def crammer_singer_loss(adv_output, label, DDP=False, **_):
    """multi-hinge criterion, reference losses.py:242-252: mean relu(1 + max_{c != label} adv[c] - adv[label])."""
    return F.ClassLossFn.apply(adv_output, label, 1)
