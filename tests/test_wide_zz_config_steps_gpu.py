"""GPU: one training step of the non-StyleGAN CIFAR10 configuration files of the reference (width 8) against what the reference's OWN worker produced for that file:
tests/golden/config_steps.npz holds, per file, the random draws the reference's unmodified WORKER.train_discriminator / train_generator (src/worker.py:213-681) consumed
when tools/config_worker_parity_emulated.py --emit ran them on the CPU, and the resulting losses / gradient norms. The networks are rebuilt from torch.manual_seed(0)
(config_map.build reproduces the reference's initialisation bit for bit: tests/test_host_cpu.py), the draws replayed on the device in the reference's order.
Written after the round's last GPU second: green on the CPU interpreter (tests/test_aug_cpu.py::test_emulated_config_steps_against_the_references_worker and the tool's
own comparison, profiles/r05_config_worker_parity_emulated.txt); first GPU run = the driver's. Last-sorted on purpose."""
import os

import pytest
import torch

import aug_checks as AC

pytestmark = pytest.mark.gpu


# the default GPU suite runs one file per feature combination (28 of the 55: the suite's time limit); SG_SLOW=1 runs all of them
SUBSET = ["ACGAN-Mod-ADC", "ACGAN-Mod-Big-TAC", "BigGAN", "BigGAN-ADA", "BigGAN-APA", "BigGAN-CR", "BigGAN-Deep", "BigGAN-Deep-StudioGAN", "BigGAN-DiffAug-LeCam", "BigGAN-ICR",
          "BigGAN-Info", "ContraGAN-ADC", "ContraGAN", "DCGAN-Info", "LGAN", "LOGAN", "LSGAN", "MDGAN", "MHGAN", "ProjGAN", "ReACGAN-ADC-DiffAug", "ReACGAN-TAC", "SAGAN", "SNGAN",
          "SNGAN-Info", "WGAN-DRA", "WGAN-GP", "WGAN-WC"]
NAMES = [n for n in AC.config_step_names() if os.environ.get("SG_SLOW") == "1" or n in SUBSET]


@pytest.mark.parametrize("name", NAMES)
def test_config_step_against_the_references_worker(sg, name):
    AC.config_step_case(name, torch.device("cuda:0"))
