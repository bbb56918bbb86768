"""Import the REAL reference (/root/reference/src) on CPU, with MagicMock stubs for third-party modules that are
absent from this image (recipe of SURVEY.md §8c). Works only in the authoring container -- /root/reference does not
exist on the GPU box, so nothing in tests -m gpu / smoke() / bench.py may import this module.

TEST INFRASTRUCTURE ONLY (used by oracle/make_golden.py and the CPU-side oracle-vs-reference tests).
"""
import importlib
import os
import sys
import tempfile
from unittest.mock import MagicMock

import yaml

REF_SRC = "/root/reference/src"
_STUBS = ["torchvision", "torchvision.datasets", "torchvision.utils", "torchvision.transforms", "torchvision.transforms.functional",
          "torchvision.models", "torchvision.models.utils", "seaborn", "wandb", "h5py", "kornia", "kornia.filters"]


def available():
    return os.path.isdir(REF_SRC)


def _prepare():
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    for name in _STUBS:
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = MagicMock()


def load_cfgs(overrides):
    """reference config.Configurations from a dict {SECTION: {key: value}} (written to a temporary yaml)."""
    _prepare()
    import config  # reference src/config.py
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
        yaml.safe_dump(overrides, f)
        path = f.name
    try:
        cfgs = config.Configurations(path)
    finally:
        os.unlink(path)
    return cfgs


def build_models(cfgs, mixed_precision=False):
    """Generator / Discriminator exactly as reference src/models/model.py:103-133 builds them (CPU)."""
    _prepare()
    module = importlib.import_module("models." + cfgs.MODEL.backbone)
    M, D = cfgs.MODEL, cfgs.DATA
    Gen = module.Generator(z_dim=M.z_dim, g_shared_dim=M.g_shared_dim, img_size=D.img_size, g_conv_dim=M.g_conv_dim, apply_attn=M.apply_attn,
                           attn_g_loc=M.attn_g_loc, g_cond_mtd=M.g_cond_mtd, num_classes=D.num_classes, g_init=M.g_init, g_depth=M.g_depth,
                           mixed_precision=mixed_precision, MODULES=cfgs.MODULES, MODEL=M)
    Dis = module.Discriminator(img_size=D.img_size, d_conv_dim=M.d_conv_dim, apply_d_sn=M.apply_d_sn, apply_attn=M.apply_attn,
                               attn_d_loc=M.attn_d_loc, d_cond_mtd=M.d_cond_mtd, aux_cls_type=M.aux_cls_type, d_embed_dim=M.d_embed_dim,
                               num_classes=D.num_classes, normalize_d_embed=M.normalize_d_embed, d_init=M.d_init, d_depth=M.d_depth,
                               mixed_precision=mixed_precision, MODULES=cfgs.MODULES, MODEL=M)
    return Gen, Dis


def split_state(module):
    """(params, buffers) dicts keyed by the reference's names."""
    P = {k: v.detach().clone() for k, v in module.named_parameters()}
    B = {k: v.detach().clone() for k, v in module.named_buffers()}
    return P, B
