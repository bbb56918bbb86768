"""Discriminator heads shared by every backbone mirror (reference src/models/big_resnet.py:307-333,358-413, identical blocks in
resnet.py / deep_conv.py / big_resnet_deep_*.py): adversarial linear1 (1, 1 + classes for multi-hinge, classes for multi-discriminator),
projection embedding (PD), auxiliary classifier (AC), embedding + proxy pairs of ContraGAN / ReACGAN (2C, D2DCE), and the twin
(TAC) / discriminative (ADC) auxiliary-classifier variants. Attribute names are the reference's (linear1, linear2, embedding,
linear_mi, embedding_mi), so state_dicts interchange."""
import torch

from .. import functional as F

HEADS = ("W/O", "PD", "AC", "2C", "D2DCE", "MH", "MD")


def build_heads(self, MODULES, feat_dim, d_cond_mtd, aux_cls_type, d_embed_dim, num_classes, MODEL):
    if d_cond_mtd not in HEADS or aux_cls_type not in ("W/O", "N/A", "TAC", "ADC"):
        raise NotImplementedError(f"d_cond_mtd {d_cond_mtd} / aux_cls_type {aux_cls_type}")
    if d_cond_mtd == "MH":
        self.linear1 = MODULES.d_linear(in_features=feat_dim, out_features=1 + num_classes, bias=True)
    elif d_cond_mtd == "MD":
        self.linear1 = MODULES.d_linear(in_features=feat_dim, out_features=num_classes, bias=True)
    else:
        self.linear1 = MODULES.d_linear(in_features=feat_dim, out_features=1, bias=True)
    if aux_cls_type == "ADC":
        num_classes = num_classes * 2
    if d_cond_mtd == "AC":
        self.linear2 = MODULES.d_linear(in_features=feat_dim, out_features=num_classes, bias=False)
    elif d_cond_mtd == "PD":
        self.embedding = MODULES.d_embedding(num_classes, feat_dim)
    elif d_cond_mtd in ("2C", "D2DCE"):
        self.linear2 = MODULES.d_linear(in_features=feat_dim, out_features=d_embed_dim, bias=True)
        self.embedding = MODULES.d_embedding(num_classes, d_embed_dim)
    if aux_cls_type == "TAC":
        if d_cond_mtd == "AC":
            self.linear_mi = MODULES.d_linear(in_features=feat_dim, out_features=num_classes, bias=False)
        elif d_cond_mtd in ("2C", "D2DCE"):
            self.linear_mi = MODULES.d_linear(in_features=feat_dim, out_features=d_embed_dim, bias=True)
            self.embedding_mi = MODULES.d_embedding(num_classes, d_embed_dim)
        else:
            raise NotImplementedError
    # Q head network of InfoGAN, built LAST like the reference's (src/models/big_resnet.py:337-344): a seeded construction then draws every layer's initial weights in
    # the reference's order (tests/aug_checks.py config_step_case relies on it); parameters owned by the GENERATOR's optimiser (src/config.py:501-512)
    info_type = getattr(MODEL, "info_type", "N/A")
    if info_type in ("discrete", "both"):
        self.info_discrete_linear = MODULES.d_linear(in_features=feat_dim, out_features=MODEL.info_num_discrete_c * MODEL.info_dim_discrete_c, bias=False)
    if info_type in ("continuous", "both"):
        self.info_conti_mu_linear = MODULES.d_linear(in_features=feat_dim, out_features=MODEL.info_num_conti_c, bias=False)
        self.info_conti_var_linear = MODULES.d_linear(in_features=feat_dim, out_features=MODEL.info_num_conti_c, bias=False)


def _embed(module, label, slot):
    if module._sg_sn:
        return F.SNEmbeddingFn.apply(module.weight_orig, label.reshape(-1), module._sg_rt, slot)
    return F.EmbeddingFn.apply(module.weight, label.reshape(-1))


INFO_PARAMS = ("info_discrete_linear", "info_conti_mu_linear", "info_conti_var_linear")      # reference src/config.py:346 MISC.info_params


def apply_heads(self, h, label, slot, adc_fake=False, hw=None):
    """h: [B, C] = sum_hw relu(features); hw: the number of positions summed (the Q heads read the MEAN feature, big_resnet.py:374-377).
    Returns the reference's 12-key dictionary."""
    mtd, aux = self.d_cond_mtd, self.aux_cls_type
    embed = proxy = cls_output = mi_embed = mi_proxy = mi_cls_output = None
    info_discrete_c_logits = info_conti_mu = info_conti_var = None
    info_type = getattr(self.MODEL, "info_type", "N/A")
    if info_type != "N/A":
        if hw is None:
            raise RuntimeError("apply_heads: the Q heads need the number of pooled positions (hw)")
        cached = getattr(self, "_sg_info_inv", None)          # keyed on the Python ints: no device read-back per forward
        if cached is None or cached[0] != (int(hw), h.device):
            cached = ((int(hw), h.device), torch.full((1,), 1.0 / hw, dtype=torch.float32, device=h.device))
            self.__dict__["_sg_info_inv"] = cached
        inv = cached[1]
        hm = F.ScalePtrFn.apply(h, inv)                          # h / (bottom_h * bottom_w)
        if info_type in ("discrete", "both"):
            info_discrete_c_logits = self.info_discrete_linear.forward_rt(hm, slot)
        if info_type in ("continuous", "both"):
            info_conti_mu = self.info_conti_mu_linear.forward_rt(hm, slot)
            info_conti_var = F.ExpFn.apply(self.info_conti_var_linear.forward_rt(hm, slot))
    if mtd in ("W/O", "PD"):
        pd = mtd == "PD"
        adv_output = F.PDHeadFn.apply(h, self.linear1.master_weight, self.linear1.bias, self.embedding.master_weight if pd else None,
                                      label if pd else None, self.linear1._sg_rt, self.embedding._sg_rt if pd else None, slot)
    else:
        adv_output = torch.squeeze(self.linear1.forward_rt(h, slot))
    if aux == "ADC":                         # odd labels for fake, even for real (big_resnet.py:366-371)
        label = label * 2 + 1 if adc_fake else label * 2
    if mtd == "AC":
        if self.normalize_d_embed:           # (the reference's loop over linear2.parameters() rebinds a local name: the weight is NOT normalised)
            h = F.RowNormalizeFn.apply(h, 1e-12)
        cls_output = self.linear2.forward_rt(h, slot)
    elif mtd in ("2C", "D2DCE"):
        embed = self.linear2.forward_rt(h, slot)
        proxy = _embed(self.embedding, label, slot)
        if self.normalize_d_embed:
            embed, proxy = F.RowNormalizeFn.apply(embed, 1e-12), F.RowNormalizeFn.apply(proxy, 1e-12)
    elif mtd == "MD":
        adv_output = F.GatherColsFn.apply(adv_output, label)
    if aux == "TAC":
        if mtd == "AC":
            mi_cls_output = self.linear_mi.forward_rt(h, slot)
        elif mtd in ("2C", "D2DCE"):
            mi_embed = self.linear_mi.forward_rt(h, slot)
            mi_proxy = _embed(self.embedding_mi, label, slot)
            if self.normalize_d_embed:
                mi_embed, mi_proxy = F.RowNormalizeFn.apply(mi_embed, 1e-12), F.RowNormalizeFn.apply(mi_proxy, 1e-12)
    return {"h": h, "adv_output": adv_output, "embed": embed, "proxy": proxy, "cls_output": cls_output, "label": label,
            "mi_embed": mi_embed, "mi_proxy": mi_proxy, "mi_cls_output": mi_cls_output,
            "info_discrete_c_logits": info_discrete_c_logits, "info_conti_mu": info_conti_mu, "info_conti_var": info_conti_var}
