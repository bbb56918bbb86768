"""GPU: InfoGAN (reference configs/CIFAR10/{BigGAN,SNGAN,DCGAN,ReACGAN}-Info.yaml): the generators' code injection ("cBN" / "concat"; models/big_resnet.py:81-92,
models/resnet.py:95-107, models/deep_conv.py:56-68), the discriminators' Q heads (models/big_resnet.py:337-344,373-377), the information losses and toggling of
src/worker.py:220-224,508-512,607-618 and the Q heads' Adam with the generator's settings (src/config.py:499-517), against vectors the REAL reference wrote
(tests/golden/info.npz, oracle/make_golden_info.py): loss and every gradient of one discriminator and one generator update, the Q heads after the step.
Written when the round's GPU minutes were nearly spent: green on the CPU interpreter first (tests/test_aug_cpu.py::test_emulated_infogan_updates_match_reference_vectors),
then on the GPU in the round's last seconds (profiles/r05_pytest_wide_r.txt)."""
import pytest
import torch

import aug_checks as AC

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", AC.INFO_CASES)
def test_infogan_updates_match_reference_vectors(sg, tag):
    AC.info_case(tag, torch.device("cuda:0"))


@pytest.mark.parametrize("name", ["sngan32", "biggan32"])
def test_freeze_d(sg, name):
    """RUN.freezeD (reference src/utils/misc.py:190-216): frozen blocks untouched, the rest as in the unfrozen update's golden vectors"""
    AC.freeze_d_case(name, torch.device("cuda:0"), 2)


def test_logan_latent_optimisation(sg):
    """LOGAN (reference configs/CIFAR10/LOGAN.yaml; src/utils/losses.py:278-298): against the REAL reference's vectors; the discriminator side's bounds are one flipped
    ReLU unit wide (see aug_checks.logan_case; profiles/r06_logan_tie.txt), the tight bounds sit in the oracle test below"""
    AC.logan_case(torch.device("cuda:0"))


@pytest.mark.parametrize("scale", [0.9, 1.0001])
def test_logan_discriminator_side_against_the_oracle(sg, scale):
    """the same update on latents clear of the fixture's ReLU tie, against the fp64 oracle's double backward at rounding-level bounds (1e-4 of the largest gradient)"""
    AC.logan_oracle_case(torch.device("cuda:0"), scale)


@pytest.mark.parametrize("name", ["md", "ac", "2c", "mh"])
def test_r1_with_classifier_heads(sg, name):
    """R1 through attention + the linear adversarial heads (reference configs/*/MDGAN.yaml uses it with the multi-discriminator head); interpreter-verified,
    first GPU run = the driver's"""
    AC.r1_with_heads_case(name, torch.device("cuda:0"))


@pytest.mark.parametrize("name", AC.STANDING_CASES)
def test_generator_preparation_for_evaluation(sg, name):
    """worker.GeneratorController.prepare_generator (reference src/utils/misc.py:63-107,301-334: standing statistics / batch statistics / plain evaluation in front of
    the FID / IS feature extraction) against tests/golden/standing.npz; interpreter-verified, first GPU run = the driver's"""
    AC.standing_case(name, torch.device("cuda:0"))
