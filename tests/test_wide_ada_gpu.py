"""GPU: the remaining stages of the ADA pipeline -- image-space filtering (per-image 4-band amplification filter over the reflect-padded image), additive noise,
cutout ('filter', 'noise', 'cutout', 'bgcfnc' of the reference's ada_augpipe; csrc/ext/ada.hip sg_fir_reflect / sg_ada_noise_cutout) -- fed the draws the REAL
reference's AdaAugment made, against its output and image gradient (tests/golden/ada.npz), and the operators against their adjoints at the benchmark's image size.
Written when the round's GPU minutes were nearly spent: green on the CPU interpreter first (tests/test_aug_cpu.py::test_emulated_ada_pipeline_matches_reference_vectors),
then on the GPU in the round's last seconds (profiles/r05_pytest_wide_r.txt)."""
import pytest
import torch

import aug_checks as AC
from oracle import make_golden_ada as MGD

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.mark.parametrize("case", MGD.CASES[7:], ids=[c[0] for c in MGD.CASES[7:]])
def test_ada_filter_noise_cutout_match_reference_vectors(sg, case):
    AC.ada_case(case, DEV)


def test_fir_reflect_and_cutout_adjoints_at_benchmark_size(sg):
    AC.ada_filter_adjoint_case((64, 3, 128, 128), DEV, 1)
    AC.ada_filter_adjoint_case((5, 1, 33, 47), DEV, 2)
