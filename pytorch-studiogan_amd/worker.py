"""Training-step driver with the structure of the reference's WORKER (reference src/worker.py:213-497
train_discriminator, :502-681 train_generator, loop body src/loader.py:392-405), minus everything §8 marks out of scope
(ADA / APA / SimCLR augmentation, logging, checkpointing, StyleGAN paths). DiffAugment and the consistency regularisers
(CR, bCR, zCR) are in: studiogan_amd.diffaug / .cr in the roles of cfgs.AUG.series_augment / parallel_augment.
bench.py, __graft_entry__.smoke() and the step-parity tests run this; on a StudioGAN checkout the unmodified reference worker drives the same modules through the backbone seam
(INTEGRATION.md).
"""
import copy

import torch

from . import losses as sg_losses
from . import ops
from .optim import FusedAdam, Ema, sync_replicas


def toggle_grad(model, grad, num_freeze_layers=-1):
    """reference src/utils/misc.py:190-216. num_freeze_layers = N >= 0 (RUN.freezeD, src/worker.py:219): every parameter trainable except those of the first N
    entries of `model.blocks` (fine-tuning with the discriminator's early blocks frozen; the weight bank folds gradients of the trainable layers only)."""
    if num_freeze_layers == -1:
        for p in model.parameters():
            p.requires_grad = grad
        return
    assert grad, "cannot freeze the model when grad is False"
    if hasattr(model, "in_dims"):
        assert num_freeze_layers < len(model.in_dims), f"cannot freeze the {num_freeze_layers}th block > total {len(model.in_dims)} blocks."
    for name, p in model.named_parameters():
        p.requires_grad = True
        for layer in range(num_freeze_layers):
            if "blocks.{layer}".format(layer=layer) in name:
                p.requires_grad = False


def untrack_bn_statistics(m):
    if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
        m.track_running_stats = False


def track_bn_statistics(m):
    if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
        m.track_running_stats = True


def set_deterministic_op_trainable(m):
    """reference src/utils/misc.py:254-262: conv/linear/embedding stay in .train() so spectral norm keeps iterating."""
    if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d, torch.nn.Linear, torch.nn.Embedding)):
        m.train()


def make_GAN_trainable(Gen, Gen_ema, Dis):
    Gen.train()
    Gen.apply(track_bn_statistics)
    if Gen_ema is not None:
        Gen_ema.train()
        Gen_ema.apply(track_bn_statistics)
    Dis.train()
    Dis.apply(track_bn_statistics)


def make_GAN_untrainable(Gen, Gen_ema, Dis):
    Gen.eval()
    Gen.apply(set_deterministic_op_trainable)
    if Gen_ema is not None:
        Gen_ema.eval()
        Gen_ema.apply(set_deterministic_op_trainable)
    Dis.eval()
    Dis.apply(set_deterministic_op_trainable)


def set_bn_trainable(m):
    """reference src/utils/misc.py:236-238"""
    if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
        m.train()


def reset_bn_statistics(m):
    """reference src/utils/misc.py:264-266"""
    if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
        m.reset_running_stats()


def sample_latents(batch_size, z_dim, num_classes, device, z_prior="gaussian", truncation_factor=-1.0, MODEL=None):
    """(zs, fake_labels) of one evaluation / standing-statistics batch in the reference's draw order (src/utils/sample.py:69-118 with y_sampler "totally_random"):
    labels on the device; latents -- N(0, I) on the device, the truncated normal of scipy on the host when truncation_factor > 0 (sample.py:27-40), U(-1, 1) on the
    host for the uniform prior; then InfoGAN's discrete (one-hot) and continuous codes behind z."""
    ys = torch.randint(low=0, high=max(num_classes, 1), size=(batch_size,), dtype=torch.long, device=device)
    if z_prior == "gaussian":
        if truncation_factor == -1.0:
            zs = torch.randn(batch_size, z_dim, device=device)
        elif truncation_factor > 0:
            from scipy.stats import truncnorm
            zs = torch.FloatTensor(truncnorm.rvs(-truncation_factor, truncation_factor, size=[batch_size, z_dim])).to(device)
        else:
            raise ValueError("truncated_factor must be positive.")
    elif z_prior == "uniform":
        zs = torch.FloatTensor(batch_size, z_dim).uniform_(-1.0, 1.0).to(device)
    else:
        raise NotImplementedError(z_prior)
    info_type = getattr(MODEL, "info_type", "N/A")
    if info_type in ("discrete", "both"):
        disc = torch.randint(MODEL.info_dim_discrete_c, (batch_size, MODEL.info_num_discrete_c), device=device)
        zs = torch.cat((zs, torch.nn.functional.one_hot(disc, MODEL.info_dim_discrete_c).view(batch_size, -1)), dim=1)
    if info_type in ("continuous", "both"):
        zs = torch.cat((zs, torch.rand(batch_size, MODEL.info_num_conti_c, device=device) * 2 - 1), dim=1)
    return zs, ys


def apply_standing_statistics(generator, standing_max_batch, standing_step, z_dim, num_classes, device, z_prior="gaussian", world_size=1,
                              distributed_data_parallel=False, MODEL=None):
    """reference src/utils/misc.py:301-334: reset every batch norm's running statistics, then `standing_step` training-mode generator forwards over batches of a
    random size in [1, standing_max_batch / world_size] (python `random`, like the reference) with 'totally_random' labels -- the running statistics (momentum 0.1)
    the evaluation forwards then normalise with. Leaves the generator in eval mode."""
    import random
    generator.train()
    generator.apply(reset_bn_statistics)
    with torch.no_grad():
        for _ in range(standing_step):
            per_gpu = standing_max_batch // world_size
            b = random.randint(1, per_gpu) if distributed_data_parallel else random.randint(1, per_gpu) * world_size
            zs, ys = sample_latents(b, z_dim, num_classes, device, z_prior=z_prior, MODEL=MODEL)
            generator(zs, ys, eval=False)
    generator.eval()


class GeneratorController:
    """reference src/utils/misc.py:63-107: puts the generator (the EMA twin when there is one, src/worker.py:159) into the state the evaluation forwards of
    src/metrics/features.py:17-65 expect. standing_statistics: accumulate them on the first call (std_stat_counter <= 1), plain eval mode afterwards;
    batch_statistics: batch norm stays in training mode without tracking; convolution / linear / embedding layers always stay in .train() so that spectral norm
    keeps iterating (src/utils/misc.py:254-262)."""

    def __init__(self, generator, batch_statistics, standing_statistics, standing_max_batch, standing_step, z_dim, num_classes, device, z_prior="gaussian",
                 world_size=1, distributed_data_parallel=False, MODEL=None, std_stat_counter=0):
        self.generator, self.batch_statistics, self.standing_statistics = generator, batch_statistics, standing_statistics
        self.standing_max_batch, self.standing_step, self.std_stat_counter = standing_max_batch, standing_step, std_stat_counter
        self.sample = dict(z_dim=z_dim, num_classes=num_classes, device=device, z_prior=z_prior, world_size=world_size,
                           distributed_data_parallel=distributed_data_parallel, MODEL=MODEL)

    def prepare_generator(self):
        if self.standing_statistics:
            if self.std_stat_counter > 1:
                self.generator.eval()
            else:
                self.generator.train()
                apply_standing_statistics(self.generator, self.standing_max_batch, self.standing_step, **self.sample)
                self.generator.eval()
            self.generator.apply(set_deterministic_op_trainable)
        else:
            self.generator.eval()
            if self.batch_statistics:
                self.generator.apply(set_bn_trainable)
                self.generator.apply(untrack_bn_statistics)
            self.generator.apply(set_deterministic_op_trainable)
        return self.generator, None, None          # (generator, generator_mapping, generator_synthesis): the StyleGAN halves do not exist here


def sample_zy(batch_size, z_dim, num_classes, device, generator=None, z_prior="gaussian"):
    """reference src/utils/sample.py:69-76 ('totally_random' labels): the LABELS are drawn first, then the latents -- N(0, I) on the device, U(-1, 1) on the
    host generator for the uniform prior (nothing else is drawn for it) -- so a seeded run consumes the generators in the reference's order."""
    ys = torch.randint(low=0, high=max(num_classes, 1), size=(batch_size,), dtype=torch.long, device=device, generator=generator)
    if z_prior == "gaussian":
        zs = torch.randn(batch_size, z_dim, device=device, generator=generator)
    elif z_prior == "uniform":
        zs = torch.FloatTensor(batch_size, z_dim).uniform_(-1.0, 1.0).to(device)
    else:
        raise NotImplementedError(z_prior)
    return zs, ys


def adapt_aa_p(aa_p, sign_sum, count, aa_target, aa_kimg):
    """The ADA / APA overfitting heuristic (reference src/worker.py:478-482): the mean sign of the real logits over the interval is pushed towards aa_target by
    moving the augmentation probability count / (aa_kimg * 1000) up or down, clipped to [0, 1]. sign_sum / count: the accumulated (and, under data
    parallelism, all-reduced) statistic of functional.sign_count_."""
    import numpy as np
    heuristic = sign_sum / count
    adjust = float(np.sign(heuristic - aa_target)) * count / (aa_kimg * 1000)
    return min(1.0, max(aa_p + adjust, 0.0))


class Worker:
    def __init__(self, Gen, Dis, z_dim, num_classes, batch_size, adv_loss="hinge", g_lr=2e-4, d_lr=2e-4, beta1=0.5, beta2=0.999,
                 d_updates_per_step=5, g_updates_per_step=1, acml_steps=1, apply_g_ema=False, g_ema_decay=0.9999, g_ema_start=0,
                 group=None, apply_gp=False, gp_lambda=10.0, apply_topk=False, topk_gamma=0.99, topk_nu=0.5,
                 apply_r1_reg=False, r1_lambda=10.0, apply_maxgp=False, maxgp_lambda=1.0, apply_dra=False, dra_lambda=10.0,
                 apply_lecam=False, lecam_lambda=0.3, lecam_ema_start_iter=1000, lecam_ema_decay=0.99,
                 d_cond_mtd="W/O", aux_cls_type="W/O", cond_lambda=1.0, temperature=1.0, m_p=1.0, tac_dis_lambda=1.0, tac_gen_lambda=1.0,
                 mh_lambda=1.0, apply_diffaug=False, diffaug_type="diffaug", apply_cr=False, cr_aug_type="cr", cr_lambda=10.0,
                 apply_bcr=False, bcr_aug_type="bcr", real_lambda=10.0, fake_lambda=10.0, apply_zcr=False, radius=0.05, g_lambda=0.5, d_lambda=20.0,
                 apply_fm=False, fm_lambda=1.0, apply_wc=False, wc_bound=0.01,
                 apply_apa=False, apa_initial_augment_p=0.0, apa_target=0.6, apa_kimg=500, apa_interval=4,
                 apply_ada=False, ada_aug_type="bgc", ada_initial_augment_p=0.0, ada_target=0.6, ada_kimg=500, ada_interval=4,
                 info_type="N/A", info_num_discrete_c=0, info_dim_discrete_c=0, info_num_conti_c=0, infoGAN_loss_discrete_lambda=1.0,
                 infoGAN_loss_conti_lambda=1.0, freezeD=-1, apply_lo=False, lo_rate=0.8, lo_steps4train=2, lo_alpha=0.9, lo_beta=0.1, lo_lambda=0.1,
                 z_prior="gaussian"):
        self.Gen, self.Dis = Gen, Dis
        # latent optimisation (LOGAN; reference src/utils/sample.py:123-135, src/worker.py:319-321,598-599): z takes one step along d D(G(z)) / dz before the
        # images are generated; the transport cost of that step joins both losses
        self.apply_lo, self.lo_rate, self.lo_steps, self.lo_alpha, self.lo_beta, self.lo_lambda = apply_lo, lo_rate, lo_steps4train, lo_alpha, lo_beta, lo_lambda
        self.z_prior = z_prior
        self.freezeD = freezeD          # RUN.freezeD (src/worker.py:219): the discriminator's first blocks stay fixed
        # InfoGAN (reference src/worker.py:220-224,508-512,607-618; src/utils/sample.py:113-118; src/config.py:501-512)
        self.info_type, self.info_num_discrete_c, self.info_dim_discrete_c, self.info_num_conti_c = info_type, info_num_discrete_c, info_dim_discrete_c, info_num_conti_c
        self.info_discrete_lambda, self.info_conti_lambda = infoGAN_loss_discrete_lambda, infoGAN_loss_conti_lambda
        self.apply_wc, self.wc_bound = apply_wc, wc_bound              # weight clipping after every discriminator update (src/worker.py:489-492)
        # adaptive pseudo augmentation (src/worker.py:82,127-134,273-274,285-289,478-487): real images swapped for fakes with probability aa_p, which follows
        # the sign statistic of the real logits towards aa_target
        self.apply_apa, self.aa_p, self.aa_target, self.aa_kimg, self.aa_interval = apply_apa, float(apa_initial_augment_p), apa_target, apa_kimg, apa_interval
        self.dis_sign_real = torch.zeros(2, dtype=torch.float32, device=next(Gen.parameters()).device) if (apply_apa or apply_ada) else None
        self.apply_fm, self.fm_lambda = apply_fm, fm_lambda          # feature matching in the generator update (src/worker.py:588-596)
        # augmentations in front of the discriminator (reference src/config.py:582-626): series_augment runs on every real / fake batch
        # (src/worker.py:276-278,549-550), parallel_augment makes the second view of the consistency regularisers (:326-354)
        self.series_augment = self._augmenter(diffaug_type, "diffaug_type") if apply_diffaug else (lambda x: x)
        # adaptive discriminator augmentation (src/config.py:590-591, src/worker.py:127-134): the series augmentation is the AdaAugment pipeline, its strength p
        # follows the same heuristic
        self.apply_ada = apply_ada
        assert not (apply_ada and apply_diffaug), "AUG.apply_ada and AUG.apply_diffaug both set cfgs.AUG.series_augment: one of them (reference src/config.py:582-594)"
        if apply_ada:
            from . import ada_aug
            if ada_aug_type not in ada_aug.AUGPIPE:
                raise NotImplementedError(f"ada_aug_type = {ada_aug_type}")
            self.series_augment = ada_aug.AdaAugment(**ada_aug.AUGPIPE[ada_aug_type]).train().to(next(Gen.parameters()).device).requires_grad_(False)
            self.aa_p, self.aa_target, self.aa_kimg, self.aa_interval = float(ada_initial_augment_p), ada_target, ada_kimg, ada_interval
            self.series_augment.p.copy_(torch.as_tensor(self.aa_p))
        self.apply_cr, self.cr_lambda = apply_cr, cr_lambda
        self.apply_bcr, self.real_lambda, self.fake_lambda = apply_bcr, real_lambda, fake_lambda
        self.apply_zcr, self.radius, self.g_lambda, self.d_lambda = apply_zcr, radius, g_lambda, d_lambda
        if apply_zcr and info_type != "N/A":          # the reference's zs_eps (src/utils/sample.py:79-83) keeps width z_dim while zs gets the codes appended (:113-118)
            raise NotImplementedError("apply_zcr with InfoGAN codes: the perturbed latents would lack the codes (the reference fails there as well)")
        assert not (apply_cr and apply_bcr), "CR and bCR share cfgs.AUG.parallel_augment: one of them (reference src/config.py:596-626)"
        self.parallel_augment = self._augmenter(cr_aug_type if apply_cr else bcr_aug_type, "cr_aug_type / bcr_aug_type") if (apply_cr or apply_bcr) else None
        self.apply_dra, self.dra_lambda = apply_dra, dra_lambda
        self.apply_lecam, self.lecam_lambda, self.lecam_ema_start_iter = apply_lecam, lecam_lambda, lecam_ema_start_iter
        self.lecam_ema = ops.LeCamEMA(decay=lecam_ema_decay, start_iter=lecam_ema_start_iter) if apply_lecam else None   # src/worker.py:139-140
        self.apply_r1_reg, self.r1_lambda, self.apply_maxgp, self.maxgp_lambda = apply_r1_reg, r1_lambda, apply_maxgp, maxgp_lambda
        # top-k training of the generator (reference src/worker.py:117-121,565-566; k decays by topk_gamma per epoch down to nu * batch)
        self.apply_topk, self.topk_gamma, self.topk_nu = apply_topk, topk_gamma, topk_nu
        self.topk = batch_size
        self.apply_gp, self.gp_lambda = apply_gp, gp_lambda
        self.z_dim, self.num_classes, self.batch_size = z_dim, num_classes, batch_size
        # class conditioning of the discriminator (reference src/worker.py:123-157): classifier-based GANs get a conditioning loss on top of
        # the adversarial one; the multi-hinge loss ("MH") replaces it
        self.adv_loss, self.d_cond_mtd, self.aux_cls_type = adv_loss, d_cond_mtd, aux_cls_type
        self.cond_lambda, self.tac_dis_lambda, self.tac_gen_lambda, self.mh_lambda = cond_lambda, tac_dis_lambda, tac_gen_lambda, mh_lambda
        self.adc_fake = aux_cls_type == "ADC"
        nc = num_classes * 2 if self.adc_fake else num_classes
        DDP = group is not None
        self.cond_loss = None
        if d_cond_mtd == "AC":
            self.cond_loss = sg_losses.CrossEntropyLoss()
        elif d_cond_mtd == "2C":
            self.cond_loss = sg_losses.ConditionalContrastiveLoss(num_classes=nc, temperature=temperature, master_rank="cuda", DDP=DDP, group=group)
        elif d_cond_mtd == "D2DCE":
            self.cond_loss = sg_losses.Data2DataCrossEntropyLoss(num_classes=nc, temperature=temperature, m_p=m_p, master_rank="cuda", DDP=DDP, group=group)
        self.cond_loss_mi = copy.deepcopy(self.cond_loss) if aux_cls_type == "TAC" else None
        if adv_loss == "MH":
            self.d_loss = self.g_loss = sg_losses.crammer_singer_loss
            self.lossy = torch.full((batch_size,), num_classes, dtype=torch.long, device=next(Gen.parameters()).device)
        else:
            self.d_loss, self.g_loss = sg_losses.D_LOSSES[adv_loss], sg_losses.G_LOSSES[adv_loss]
        self.n_d, self.n_g, self.acml = d_updates_per_step, g_updates_per_step, acml_steps
        self.device = next(Gen.parameters()).device
        if group is not None:
            # what DDP's constructor broadcast does for the reference (src/models/model.py:171-180, seeds differ per rank: src/loader.py:99)
            sync_replicas(Gen, group)
            sync_replicas(Dis, group)
        self.g_optimizer = FusedAdam(Gen.parameters(), lr=g_lr, betas=(beta1, beta2), eps=1e-6)
        self.d_optimizer = FusedAdam(Dis.parameters(), lr=d_lr, betas=(beta1, beta2), eps=1e-6)
        # the Q heads live in the discriminator module but are trained by the GENERATOR's optimiser settings in the generator update (src/config.py:501-512):
        # a small Adam of their own with the generator's hyper-parameters and step count. The discriminator's fused Adam still walks over them, with gradients
        # that are exactly zero in its updates (they do not require grad there) and moments that therefore stay zero: an exact no-op.
        self.info_modules, self.info_optimizer = [], None
        if info_type != "N/A":
            from .backbones.heads import INFO_PARAMS
            from .optim import SmallAdam
            self.info_modules = [getattr(Dis, n) for n in INFO_PARAMS if hasattr(Dis, n)]
            assert self.info_modules, "info_type is set but the discriminator has no Q heads (build it with MODEL.info_type)"
            self.info_optimizer = SmallAdam([p for m in self.info_modules for p in m.parameters()], lr=g_lr, betas=(beta1, beta2), eps=1e-6)
        import os as _os
        self._xchg = group is not None or _os.environ.get("SG_EXCHANGE_SELFTEST") == "1"
        if self._xchg:
            # data parallelism: the networks' block boundaries start the gradient all-reduce of finished arena ranges during backward
            # (optim.ExchangePlan: what DDP's buckets do for the reference, src/models/model.py:171-180)
            self.g_optimizer.attach(Gen)
            self.d_optimizer.attach(Dis)
        # update procedures with a create_graph pass inside keep the whole exchange in step()
        self._plain_d_update = not (apply_gp or apply_r1_reg or apply_maxgp or apply_dra or apply_lo)
        self.Gen_ema, self.ema = None, None
        if apply_g_ema:
            self.Gen_ema = copy.deepcopy(Gen)
            self.ema = Ema(source=Gen, target=self.Gen_ema, decay=g_ema_decay, start_iter=g_ema_start)
        self.group = group

    @staticmethod
    def _augmenter(kind, what):
        from . import diffaug, cr
        if kind == "diffaug":
            return diffaug.apply_diffaug                      # src/config.py:586-587,605-606,619-620
        if kind in ("cr", "bcr"):
            return cr.apply_cr_aug                            # src/config.py:584-585,603-604,617-618
        raise NotImplementedError(f"{what} = {kind}: SimCLR / BYOL / ADA augmentation pipelines are outside the hot path (SURVEY.md §8f)")

    def _consistency(self, a, b):
        """l2 between the two views' logits, plus their class logits (AC) or embeddings (2C / D2D-CE): src/worker.py:329-335,344-353,358-364"""
        loss = sg_losses.l2_loss(a["adv_output"], b["adv_output"])
        if self.d_cond_mtd == "AC":
            loss = loss + sg_losses.l2_loss(a["cls_output"], b["cls_output"])
        elif self.d_cond_mtd in ("2C", "D2DCE"):
            loss = loss + sg_losses.l2_loss(a["embed"], b["embed"])
        return loss

    def _sample(self, injected, k):
        """(zs, fake_labels, zs_eps): reference src/utils/sample.py:69-88 -- zs_eps = zs + radius * N(0, I) when the latent consistency term is on.
        injected entries are (z, y) or (z, y, z_eps)."""
        info = None
        if injected is not None:
            ent = injected[k]
            zs, ys = ent[0], ent[1]
            eps = ent[2] if len(ent) > 2 else None
            info = ent[3] if len(ent) > 3 else None
        else:
            zs, ys = sample_zy(self.batch_size, self.z_dim, self.num_classes, self.device, z_prior=self.z_prior)
            eps = None
        if self.apply_zcr and eps is None:          # src/utils/sample.py:79-83: the perturbation follows the prior
            if self.z_prior == "uniform":
                eps = zs + self.radius * torch.FloatTensor(zs.shape[0], self.z_dim).uniform_(-1.0, 1.0).to(zs.device)
            else:
                eps = zs + self.radius * torch.randn(zs.shape[0], self.z_dim, device=zs.device)
        self.info_codes = (None, None)
        if self.info_type != "N/A":          # src/utils/sample.py:113-118: the codes ride behind z
            B = zs.shape[0]
            disc, conti = info if info is not None else (None, None)
            if self.info_type in ("discrete", "both"):
                if disc is None:
                    disc = torch.randint(self.info_dim_discrete_c, (B, self.info_num_discrete_c), device=zs.device)
                zs = torch.cat((zs, torch.nn.functional.one_hot(disc, self.info_dim_discrete_c).view(B, -1)), dim=1)
            if self.info_type in ("continuous", "both"):
                if conti is None:
                    conti = torch.rand(B, self.info_num_conti_c, device=zs.device) * 2 - 1
                zs = torch.cat((zs, conti), dim=1)
            self.info_codes = (disc, conti)
        self.trsp_cost = None
        if self.apply_lo:                    # src/utils/sample.py:123-135
            zs, self.trsp_cost = sg_losses.latent_optimise(zs=zs, fake_labels=ys, generator=self.Gen, discriminator=self.Dis, batch_size=zs.shape[0],
                                                           lo_rate=self.lo_rate, lo_steps=self.lo_steps, lo_alpha=self.lo_alpha, lo_beta=self.lo_beta, eval=False,
                                                           cal_trsp_cost=True, device=self.device)
        return zs, ys, (eps if self.apply_zcr else None)

    # -- src/worker.py:213-497 ------------------------------------------------------------------------------------
    def train_discriminator(self, current_step, real_batches, injected=None):
        """real_batches: list (n_d * acml) of (images NCHW fp32 in [-1,1], labels). injected: optional list of (z, y)."""
        make_GAN_trainable(self.Gen, self.Gen_ema, self.Dis)
        toggle_grad(self.Gen, False)
        toggle_grad(self.Dis, True, self.freezeD)
        for m in self.info_modules:          # src/worker.py:220-224: the Q heads are not the discriminator's to train
            toggle_grad(m, False)
        self.Gen.apply(untrack_bn_statistics)
        k = 0
        dis_acml_loss = None
        for _ in range(self.n_d):
            self.d_optimizer.zero_grad()
            for micro in range(self.acml):
                real_images, real_labels = real_batches[k]
                zs, fake_labels, zs_eps = self._sample(injected, k)
                k += 1
                fake_images = self.Gen(zs, fake_labels)
                fake_images_eps = self.Gen(zs_eps, fake_labels) if zs_eps is not None else None      # src/utils/sample.py:162-176
                if self.apply_r1_reg:    # src/worker.py:260-261
                    real_images = real_images.detach().requires_grad_(True)
                if self.apply_apa:       # src/worker.py:273-274
                    from . import apa_aug
                    real_images = apa_aug.apply_apa_aug(real_images, fake_images.detach(), self.aa_p, self.device)
                real_images_ = self.series_augment(real_images)          # src/worker.py:276-278
                fake_images_ = self.series_augment(fake_images)
                real_dict = self.Dis(real_images_, real_labels)
                fake_dict = self.Dis(fake_images_, fake_labels, adc_fake=self.adc_fake)
                self.last_d = (fake_images.detach(), real_dict["adv_output"].detach(), fake_dict["adv_output"].detach())
                if self.apply_apa or self.apply_ada:       # src/worker.py:285-289 (the sum stays on the device until the heuristic reads it)
                    from . import functional as _F
                    _F.sign_count_(self.dis_sign_real, real_dict["adv_output"])
                if self.adv_loss == "MH":          # src/worker.py:300-302
                    dis_acml_loss = self.d_loss(DDP=self.group is not None, **real_dict)
                    dis_acml_loss = dis_acml_loss + self.d_loss(fake_dict["adv_output"], self.lossy, DDP=self.group is not None)
                else:
                    dis_acml_loss = self.d_loss(real_dict["adv_output"], fake_dict["adv_output"], DDP=self.group is not None)
                if self.cond_loss is not None:     # src/worker.py:307-317
                    real_cond_loss = self.cond_loss(**real_dict)
                    self.last_cond = real_cond_loss.detach()
                    dis_acml_loss = dis_acml_loss + self.cond_lambda * real_cond_loss
                    if self.aux_cls_type == "TAC":
                        # (the reference hands the WHOLE dictionary to the twin loss, whose forward picks cls_output / embed / proxy -- not the
                        #  mi_* entries, src/worker.py:311 with src/utils/losses.py:45,78,139; mirrored as is)
                        dis_acml_loss = dis_acml_loss + self.tac_dis_lambda * self.cond_loss_mi(**fake_dict)
                    elif self.aux_cls_type == "ADC":
                        dis_acml_loss = dis_acml_loss + self.cond_lambda * self.cond_loss(**fake_dict)
                if self.apply_lo:        # src/worker.py:319-321
                    dis_acml_loss = dis_acml_loss + self.lo_lambda * self.trsp_cost
                if self.apply_cr:        # src/worker.py:325-336: the real batch's second view must score like the first
                    real_prl_dict = self.Dis(self.parallel_augment(real_images), real_labels)
                    dis_acml_loss = dis_acml_loss + self.cr_lambda * self._consistency(real_dict, real_prl_dict)
                if self.apply_bcr:       # src/worker.py:339-354: balanced CR (ICRGAN), real and fake batch
                    real_prl_images = self.parallel_augment(real_images)
                    fake_prl_images = self.parallel_augment(fake_images)
                    real_prl_dict = self.Dis(real_prl_images, real_labels)
                    fake_prl_dict = self.Dis(fake_prl_images, fake_labels, adc_fake=self.adc_fake)
                    dis_acml_loss = dis_acml_loss + self.real_lambda * self._consistency(real_dict, real_prl_dict) \
                        + self.fake_lambda * self._consistency(fake_dict, fake_prl_dict)
                if self.apply_zcr:       # src/worker.py:357-365: latent CR, D's side: G(z) and G(z + eps) score alike
                    fake_eps_dict = self.Dis(fake_images_eps, fake_labels, adc_fake=self.adc_fake)
                    dis_acml_loss = dis_acml_loss + self.d_lambda * self._consistency(fake_dict, fake_eps_dict)
                if self.apply_gp:   # src/worker.py:369-375
                    gp_loss = sg_losses.cal_grad_penalty(real_images=real_images, real_labels=real_labels, fake_images=fake_images,
                                                         discriminator=self.Dis, device=self.device)
                    self.last_gp = gp_loss.detach()
                    dis_acml_loss = dis_acml_loss + self.gp_lambda * gp_loss
                if self.apply_dra:       # src/worker.py:378-383
                    dra = sg_losses.cal_dra_penalty(real_images=real_images.detach(), real_labels=real_labels, discriminator=self.Dis, device=self.device)
                    dis_acml_loss = dis_acml_loss + self.dra_lambda * dra
                if self.apply_lecam:     # src/worker.py:395-407: under data parallelism over the logits of the GLOBAL batch
                    real_adv = sg_losses.gather_logits(real_dict["adv_output"], self.group)
                    fake_adv = sg_losses.gather_logits(fake_dict["adv_output"], self.group)
                    self.lecam_ema.update(float(real_adv.detach().mean()), "D_real", current_step)
                    self.lecam_ema.update(float(fake_adv.detach().mean()), "D_fake", current_step)
                    if current_step > self.lecam_ema_start_iter:
                        dis_acml_loss = dis_acml_loss + self.lecam_lambda * sg_losses.lecam_reg(real_adv, fake_adv, self.lecam_ema)
                if self.apply_maxgp:     # src/worker.py:386-392
                    mg = sg_losses.cal_maxgrad_penalty(real_images=real_images.detach(), real_labels=real_labels, fake_images=fake_images,
                                                       discriminator=self.Dis, device=self.device)
                    dis_acml_loss = dis_acml_loss + self.maxgp_lambda * mg
                if self.apply_r1_reg:    # src/worker.py:410-412
                    self.r1_penalty = sg_losses.cal_r1_reg(adv_output=real_dict["adv_output"], images=real_images, device=self.device)
                    dis_acml_loss = dis_acml_loss + self.r1_lambda * self.r1_penalty
                dis_acml_loss = dis_acml_loss / self.acml
                if self._xchg and self._plain_d_update and micro == self.acml - 1:
                    self.d_optimizer.arm_exchange(self.group)      # last micro-step: finished gradient ranges go on the wire during backward
                dis_acml_loss.backward()
                dis_acml_loss = dis_acml_loss.detach()     # drop the graph now: its weight-bank slots become reusable (bank._free_graph_slot)
            self.d_optimizer.step(group=self.group)
            if (self.apply_apa or self.apply_ada) and self.aa_target is not None and current_step % self.aa_interval == 0:      # src/worker.py:478-487
                import torch.distributed as _dist
                if self.group is not None:
                    _dist.all_reduce(self.dis_sign_real, op=_dist.ReduceOp.SUM, group=self.group)
                s_sum, s_cnt = (float(v) for v in self.dis_sign_real.tolist())
                self.aa_p = adapt_aa_p(self.aa_p, s_sum, s_cnt, self.aa_target, self.aa_kimg)
                if self.apply_ada:
                    self.series_augment.p.copy_(torch.as_tensor(self.aa_p))
                self.dis_sign_real.zero_()
            if self.apply_wc:            # src/worker.py:489-492
                self.d_optimizer.clamp_(self.wc_bound)
        return dis_acml_loss

    # -- src/worker.py:502-681 ------------------------------------------------------------------------------------
    def train_generator(self, current_step, injected=None, real_batches=None):
        """real_batches: list (n_g * acml) of (images, labels), only read by the feature-matching term (LOSS.apply_fm samples a real batch per micro-step,
        src/worker.py:589-591)"""
        make_GAN_trainable(self.Gen, self.Gen_ema, self.Dis)
        toggle_grad(self.Dis, False)
        toggle_grad(self.Gen, True)
        for m in self.info_modules:          # src/worker.py:508-512
            toggle_grad(m, True)
        self.Gen.apply(track_bn_statistics)
        k = 0
        gen_acml_loss = None
        for _ in range(self.n_g):
            self.g_optimizer.zero_grad()
            if self.info_optimizer is not None:
                self.info_optimizer.zero_grad()
            for micro in range(self.acml):
                zs, fake_labels, zs_eps = self._sample(injected, k)
                k += 1
                fake_images = self.Gen(zs, fake_labels)
                fake_images_eps = self.Gen(zs_eps, fake_labels) if zs_eps is not None else None
                fake_images_ = self.series_augment(fake_images)          # src/worker.py:549-550
                fake_dict = self.Dis(fake_images_, fake_labels)
                self.last_g = (fake_images.detach(), fake_dict["adv_output"].detach())
                if self.apply_topk:      # src/worker.py:565-566
                    fake_dict["adv_output"] = sg_losses.topk_values(fake_dict["adv_output"], int(self.topk))
                if self.adv_loss == "MH":          # src/worker.py:569-572
                    gen_acml_loss = self.mh_lambda * self.g_loss(DDP=self.group is not None, **fake_dict)
                else:
                    gen_acml_loss = self.g_loss(fake_dict["adv_output"], DDP=self.group is not None)
                if self.cond_loss is not None:     # src/worker.py:575-585
                    gen_acml_loss = gen_acml_loss + self.cond_lambda * self.cond_loss(**fake_dict)
                    if self.aux_cls_type == "TAC":
                        gen_acml_loss = gen_acml_loss - self.tac_gen_lambda * self.cond_loss_mi(**fake_dict)      # src/worker.py:579
                    elif self.aux_cls_type == "ADC":
                        adc_fake_dict = self.Dis(fake_images_, fake_labels, adc_fake=self.adc_fake)       # (the augmented batch: src/worker.py:583)
                        gen_acml_loss = gen_acml_loss - self.cond_lambda * self.cond_loss(**adc_fake_dict)
                if self.apply_fm:        # src/worker.py:588-596: pooled features of a real batch (detached) against the fake batch's
                    real_images, real_labels = real_batches[k - 1]
                    real_dict = self.Dis(self.series_augment(real_images), real_labels)
                    gen_acml_loss = gen_acml_loss + self.fm_lambda * sg_losses.feature_matching_loss(real_dict["h"].detach(), fake_dict["h"])
                if self.apply_lo:        # src/worker.py:598-599
                    gen_acml_loss = gen_acml_loss + self.lo_lambda * self.trsp_cost
                if self.info_type in ("discrete", "both"):      # src/worker.py:607-615
                    dim, disc = self.info_dim_discrete_c, self.info_codes[0]
                    info_discrete_loss = None
                    for c in range(self.info_num_discrete_c):
                        term = sg_losses.CrossEntropyLoss()(fake_dict["info_discrete_c_logits"][:, c * dim: dim * (c + 1)], disc[:, c])
                        info_discrete_loss = term if info_discrete_loss is None else info_discrete_loss + term
                    self.info_discrete_loss = info_discrete_loss.detach()
                    gen_acml_loss = gen_acml_loss + self.info_discrete_lambda * info_discrete_loss
                if self.info_type in ("continuous", "both"):    # src/worker.py:616-618
                    info_conti_loss = sg_losses.normal_nll_loss(self.info_codes[1], fake_dict["info_conti_mu"], fake_dict["info_conti_var"])
                    self.info_conti_loss = info_conti_loss.detach()
                    gen_acml_loss = gen_acml_loss + self.info_conti_lambda * info_conti_loss
                if self.apply_zcr:       # src/worker.py:601-603: G's side of the latent CR pushes G(z) and G(z + eps) apart
                    gen_acml_loss = gen_acml_loss - self.g_lambda * sg_losses.l2_loss(fake_images, fake_images_eps)
                gen_acml_loss = gen_acml_loss / self.acml
                if self._xchg and not self.apply_lo and micro == self.acml - 1:      # (a create_graph pass inside the update keeps the whole exchange in step())
                    self.g_optimizer.arm_exchange(self.group)
                gen_acml_loss.backward()
                gen_acml_loss = gen_acml_loss.detach()
            # Adam and the EMA of the generator copy (src/worker.py:630-634,675-676) in one launch
            self.g_optimizer.step(ema=self.ema, iteration=current_step, group=self.group)
            if self.info_optimizer is not None:
                self.info_optimizer.step(group=self.group)
        return gen_acml_loss

    def adjust_topk(self):
        """once per epoch (reference src/loader.py: `worker.topk = losses.adjust_k(...)` under LOSS.apply_topk)."""
        self.topk = sg_losses.adjust_k(current_k=self.topk, topk_gamma=self.topk_gamma, inf_k=int(self.batch_size * self.topk_nu))
        return self.topk

    # -- src/loader.py:392-405 ------------------------------------------------------------------------------------
    def step(self, current_step, real_batches, injected_d=None, injected_g=None):
        d = self.train_discriminator(current_step, real_batches, injected_d)
        g = self.train_generator(current_step, injected_g, real_batches if self.apply_fm else None)
        return d, g
