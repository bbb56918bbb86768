"""From a StudioGAN configuration (the dict a `src/configs/<DATA>/<NAME>.yaml` file loads to, i.e. the overrides `config.Configurations` applies to the defaults of
reference src/config.py:56-330) to this package's constructors: `model_args(y)` = the keyword arguments of `backbones.<backbone>.Generator / Discriminator` and of
`ops.Modules`, `worker_kwargs(y)` = the keyword arguments of `worker.Worker`. Pure dictionary logic (no device, no library): a configuration that asks for something
this package does not mirror raises NotImplementedError naming the key -- tests/test_host_cpu.py runs it over EVERY non-StyleGAN configuration file of the reference."""

_N = "N/A"
# reference src/config.py defaults of the keys that matter on the training-step path
_MODEL = dict(backbone="resnet", g_cond_mtd="W/O", d_cond_mtd="W/O", aux_cls_type="W/O", normalize_d_embed=False, d_embed_dim=_N, apply_g_sn=False, apply_d_sn=False,
              g_act_fn="ReLU", d_act_fn="ReLU", apply_attn=False, attn_g_loc=[_N], attn_d_loc=[_N], z_prior="gaussian", z_dim=128, g_shared_dim=_N, g_conv_dim=64,
              d_conv_dim=64, g_depth=_N, d_depth=_N, apply_g_ema=False, g_ema_decay=_N, g_ema_start=_N, g_init="ortho", d_init="ortho", info_type=_N,
              g_info_injection=_N, info_num_discrete_c=_N, info_num_conti_c=_N, info_dim_discrete_c=_N)
_LOSS = dict(adv_loss="vanilla", cond_lambda=_N, tac_gen_lambda=_N, tac_dis_lambda=_N, mh_lambda=_N, apply_fm=False, fm_lambda=_N, apply_r1_reg=False, r1_place=_N,
             r1_lambda=_N, m_p=_N, temperature=_N, apply_wc=False, wc_bound=_N, apply_gp=False, gp_lambda=_N, apply_dra=False, dra_lambda=_N, apply_maxgp=False,
             maxgp_lambda=_N, apply_cr=False, cr_lambda=_N, apply_bcr=False, real_lambda=_N, fake_lambda=_N, apply_zcr=False, radius=_N, g_lambda=_N, d_lambda=_N,
             apply_lo=False, lo_alpha=_N, lo_beta=_N, lo_rate=_N, lo_lambda=_N, lo_steps4train=_N, lo_steps4eval=_N, apply_topk=False, topk_gamma=_N, topk_nu=_N,
             infoGAN_loss_discrete_lambda=_N, infoGAN_loss_conti_lambda=_N, apply_lecam=False, lecam_lambda=_N, lecam_ema_start_iter=_N, lecam_ema_decay=_N)
_OPT = dict(type_="Adam", batch_size=64, acml_steps=1, g_lr=0.0002, d_lr=0.0002, g_weight_decay=0.0, d_weight_decay=0.0, beta1=0.5, beta2=0.999, d_first=True,
            g_updates_per_step=1, d_updates_per_step=5)
_AUG = dict(apply_diffaug=False, apply_ada=False, ada_initial_augment_p=_N, ada_target=_N, ada_kimg=_N, ada_interval=_N, apply_apa=False, apa_initial_augment_p=_N,
            apa_target=_N, apa_kimg=_N, apa_interval=_N, cr_aug_type="W/O", bcr_aug_type="W/O", diffaug_type="W/O", ada_aug_type="W/O")
BACKBONES = ("resnet", "big_resnet", "big_resnet_deep_legacy", "big_resnet_deep_studiogan", "deep_conv")


def _section(y, name, defaults):
    got = dict(y.get(name) or {})
    unknown = [k for k in got if k not in defaults and k not in ("total_steps", "world_size")]      # (bookkeeping keys that do not reach the step)
    if unknown:
        raise NotImplementedError(f"{name}.{unknown[0]}: not a key of the training-step path this package mirrors")
    return {**defaults, **got}


def _num(v, default):
    return default if v == _N else v


def sections(y):
    return _section(y, "MODEL", _MODEL), _section(y, "LOSS", _LOSS), _section(y, "OPTIMIZATION", _OPT), _section(y, "AUG", _AUG)


def model_args(y):
    """(backbone module name, Modules kwargs, Generator kwargs, Discriminator kwargs); the MODEL namespace the backbones read is `model_namespace(y)`"""
    M, _, _, _ = sections(y)
    D = y.get("DATA") or {}
    if M["backbone"] not in BACKBONES:
        raise NotImplementedError(f"MODEL.backbone = {M['backbone']} (the StyleGAN backbones are outside SURVEY.md §8)")
    if M["g_act_fn"] != "ReLU" or M["d_act_fn"] != "ReLU":
        raise NotImplementedError("MODEL.g_act_fn / d_act_fn: only ReLU (what every non-StyleGAN configuration uses)")
    img, ncls = D.get("img_size", 32), D.get("num_classes", 10)
    modules = dict(apply_g_sn=M["apply_g_sn"], apply_d_sn=M["apply_d_sn"], g_cond_mtd=M["g_cond_mtd"], backbone=M["backbone"], g_info_injection=M["g_info_injection"])
    gen = dict(z_dim=M["z_dim"], g_shared_dim=M["g_shared_dim"], img_size=img, g_conv_dim=M["g_conv_dim"], apply_attn=M["apply_attn"], attn_g_loc=M["attn_g_loc"],
               g_cond_mtd=M["g_cond_mtd"], num_classes=ncls, g_init=M["g_init"], g_depth=M["g_depth"])
    dis = dict(img_size=img, d_conv_dim=M["d_conv_dim"], apply_d_sn=M["apply_d_sn"], apply_attn=M["apply_attn"], attn_d_loc=M["attn_d_loc"], d_cond_mtd=M["d_cond_mtd"],
               aux_cls_type=M["aux_cls_type"], d_embed_dim=M["d_embed_dim"], normalize_d_embed=M["normalize_d_embed"], num_classes=ncls, d_init=M["d_init"],
               d_depth=M["d_depth"])
    return M["backbone"], modules, gen, dis


def model_namespace(y):
    import types
    M = sections(y)[0]
    return types.SimpleNamespace(**{k: M[k] for k in ("info_type", "g_info_injection", "info_num_discrete_c", "info_num_conti_c", "info_dim_discrete_c", "backbone")})


def worker_kwargs(y):
    """keyword arguments of worker.Worker(Gen, Dis, ...) for this configuration (z_dim, num_classes, batch_size included)"""
    M, Ls, O, A = sections(y)
    D = y.get("DATA") or {}
    if O["type_"] != "Adam":
        raise NotImplementedError(f"OPTIMIZATION.type_ = {O['type_']}: the fused optimiser is Adam (every configuration file uses it)")
    if O["g_weight_decay"] or O["d_weight_decay"]:
        raise NotImplementedError("OPTIMIZATION.*_weight_decay != 0")
    if Ls["apply_r1_reg"] and Ls["r1_place"] not in (_N, "inside_loop"):
        raise NotImplementedError(f"LOSS.r1_place = {Ls['r1_place']}: lazy regularisation is a StyleGAN path")
    if Ls["adv_loss"] not in ("vanilla", "logistic", "least_square", "hinge", "wasserstein", "MH"):
        raise NotImplementedError(f"LOSS.adv_loss = {Ls['adv_loss']}")
    for flag, typ, ok in (("apply_diffaug", "diffaug_type", ("diffaug",)), ("apply_ada", "ada_aug_type", ("blit", "geom", "color", "filter", "noise", "cutout", "bg", "bgc",
                                                                                                        "bgcf", "bgcfn", "bgcfnc"))):
        if A[flag] and A[typ] not in ok:
            raise NotImplementedError(f"AUG.{typ} = {A[typ]}")
    if Ls["apply_cr"] and A["cr_aug_type"] not in ("cr", "diffaug"):
        raise NotImplementedError(f"AUG.cr_aug_type = {A['cr_aug_type']}")
    if Ls["apply_bcr"] and A["bcr_aug_type"] not in ("bcr", "diffaug"):
        raise NotImplementedError(f"AUG.bcr_aug_type = {A['bcr_aug_type']}")
    kw = dict(z_dim=M["z_dim"], num_classes=D.get("num_classes", 10), batch_size=O["batch_size"], adv_loss=Ls["adv_loss"], g_lr=O["g_lr"], d_lr=O["d_lr"],
              beta1=O["beta1"], beta2=O["beta2"], d_updates_per_step=O["d_updates_per_step"], g_updates_per_step=O["g_updates_per_step"], acml_steps=O["acml_steps"],
              apply_g_ema=M["apply_g_ema"], g_ema_decay=_num(M["g_ema_decay"], 0.9999), g_ema_start=_num(M["g_ema_start"], 0),
              apply_gp=Ls["apply_gp"], gp_lambda=_num(Ls["gp_lambda"], 10.0), apply_topk=Ls["apply_topk"], topk_gamma=_num(Ls["topk_gamma"], 0.99),
              topk_nu=_num(Ls["topk_nu"], 0.5), apply_r1_reg=Ls["apply_r1_reg"], r1_lambda=_num(Ls["r1_lambda"], 10.0), apply_maxgp=Ls["apply_maxgp"],
              maxgp_lambda=_num(Ls["maxgp_lambda"], 1.0), apply_dra=Ls["apply_dra"], dra_lambda=_num(Ls["dra_lambda"], 10.0), apply_lecam=Ls["apply_lecam"],
              lecam_lambda=_num(Ls["lecam_lambda"], 0.3), lecam_ema_start_iter=_num(Ls["lecam_ema_start_iter"], 1000), lecam_ema_decay=_num(Ls["lecam_ema_decay"], 0.99),
              d_cond_mtd=M["d_cond_mtd"], aux_cls_type=M["aux_cls_type"], cond_lambda=_num(Ls["cond_lambda"], 1.0), temperature=_num(Ls["temperature"], 1.0),
              m_p=_num(Ls["m_p"], 1.0), tac_dis_lambda=_num(Ls["tac_dis_lambda"], 1.0), tac_gen_lambda=_num(Ls["tac_gen_lambda"], 1.0), mh_lambda=_num(Ls["mh_lambda"], 1.0),
              apply_diffaug=A["apply_diffaug"], diffaug_type=A["diffaug_type"] if A["apply_diffaug"] else "diffaug",
              apply_cr=Ls["apply_cr"], cr_aug_type=A["cr_aug_type"] if Ls["apply_cr"] else "cr", cr_lambda=_num(Ls["cr_lambda"], 10.0),
              apply_bcr=Ls["apply_bcr"], bcr_aug_type=A["bcr_aug_type"] if Ls["apply_bcr"] else "bcr", real_lambda=_num(Ls["real_lambda"], 10.0),
              fake_lambda=_num(Ls["fake_lambda"], 10.0), apply_zcr=Ls["apply_zcr"], radius=_num(Ls["radius"], 0.05), g_lambda=_num(Ls["g_lambda"], 0.5),
              d_lambda=_num(Ls["d_lambda"], 20.0), apply_fm=Ls["apply_fm"], fm_lambda=_num(Ls["fm_lambda"], 1.0), apply_wc=Ls["apply_wc"], wc_bound=_num(Ls["wc_bound"], 0.01),
              apply_apa=A["apply_apa"], apa_initial_augment_p=_num(A["apa_initial_augment_p"], 0.0), apa_target=_num(A["apa_target"], 0.6), apa_kimg=_num(A["apa_kimg"], 500),
              apa_interval=_num(A["apa_interval"], 4), apply_ada=A["apply_ada"], ada_aug_type=A["ada_aug_type"] if A["apply_ada"] else "bgc",
              ada_initial_augment_p=_num(A["ada_initial_augment_p"], 0.0), ada_target=_num(A["ada_target"], 0.6), ada_kimg=_num(A["ada_kimg"], 500),
              ada_interval=_num(A["ada_interval"], 4), info_type=M["info_type"], info_num_discrete_c=_num(M["info_num_discrete_c"], 0),
              info_dim_discrete_c=_num(M["info_dim_discrete_c"], 0), info_num_conti_c=_num(M["info_num_conti_c"], 0),
              infoGAN_loss_discrete_lambda=_num(Ls["infoGAN_loss_discrete_lambda"], 1.0), infoGAN_loss_conti_lambda=_num(Ls["infoGAN_loss_conti_lambda"], 1.0),
              apply_lo=Ls["apply_lo"], lo_rate=_num(Ls["lo_rate"], 0.8), lo_steps4train=_num(Ls["lo_steps4train"], 2), lo_alpha=_num(Ls["lo_alpha"], 0.9),
              lo_beta=_num(Ls["lo_beta"], 0.1), lo_lambda=_num(Ls["lo_lambda"], 0.1), z_prior=M["z_prior"])
    return kw


def build(y, device, mixed_precision=False, group=None):
    """(Generator, Discriminator, Worker) of one configuration on `device`: what reference src/models/model.py:90-140 (load_generator_discriminator) followed by
    src/worker.py:38-230 (WORKER.__init__) assemble from a Configurations object. Imports the backbones (and through them libsgamd.so) only here."""
    import importlib
    from . import ops
    from .worker import Worker
    backbone, modules, gen, dis = model_args(y)
    mod = importlib.import_module(__package__ + ".backbones." + backbone)
    MOD, MODEL = ops.Modules(**modules), model_namespace(y)
    G = mod.Generator(mixed_precision=mixed_precision, MODULES=MOD, MODEL=MODEL, **gen).to(device)
    D = mod.Discriminator(mixed_precision=mixed_precision, MODULES=MOD, MODEL=MODEL, **dis).to(device)
    return G, D, Worker(G, D, group=group, **worker_kwargs(y))
