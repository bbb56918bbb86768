"""Adversarial losses with the reference's names and call signature (reference src/utils/losses.py:197-239,
wired by src/config.py:411-433 as cfgs.LOSS.{d_loss,g_loss}); forward + gradient in one small kernel each."""
from . import functional as F


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


G_LOSSES = {"vanilla": g_vanilla, "hinge": g_hinge, "wasserstein": g_wasserstein}
D_LOSSES = {"vanilla": d_vanilla, "hinge": d_hinge, "wasserstein": d_wasserstein}
