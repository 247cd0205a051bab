"""Model factory with the reference's names and signatures (model/utils.py:13-45, model/prompt_encoder.py:22-33,
model/prompt_learners/__init__.py:6-24): what ``VLSAHandler.func_load_model`` reaches through
``load_model(cfg['arch'], **arch_cfg)`` (runner/vlsa_handler.py:112-120).

    load_model('VLSA', text_encoder_cfg=..., image_encoder_cfg=..., prompt_learner_cfg=..., pretrained_prompt_learner_cfg=...,
               vlsa_api=..., path_clip_model=...)  ->  vlsa_amd.vlsa.VLSA

``patch_reference()`` is the one-line swap for a process running the reference's code: it points the reference's factory
(and its MIL-encoder name lookup) at the classes of this package, after which ``runner/vlsa_handler.py`` runs unmodified --
construction, freezing (126-149), ``.cuda()``, the per-bag training / evaluation loops (260-345) and checkpoint loading.
"""
from __future__ import annotations

from typing import List, Optional

__all__ = ["load_model", "Deep_VLSA", "get_prompt_encoder", "load_prompt_learner", "load_prompt_adapter", "patch_reference",
           "arch_cfg_from_run_cfg", "func_load_model"]


def load_model(arch: str, dims: Optional[List] = None, **kws):
    """model/utils.py:13-38.  Only the 'VLSA' arch is this package's business; the plain MIL baselines
    (``arch='DeepMIL'``: ABMIL / TransMIL / DSMIL ... classifiers of sa_handler) are out of scope."""
    if arch == "VLSA":
        return Deep_VLSA(**kws)
    if arch == "DeepMIL":
        raise NotImplementedError("arch='DeepMIL' (the non-VL survival baselines of model/utils.py:14-33) is outside the "
                                  "language-guided aggregation path this package implements; use the reference's own factory")
    raise NotImplementedError("Backbone {} cannot be recognized".format(arch))


def Deep_VLSA(**kws):
    """model/utils.py:40-45"""
    for need in ("text_encoder_cfg", "image_encoder_cfg", "prompt_learner_cfg"):
        assert need in kws
    from .vlsa import VLSA
    return VLSA(**kws)


def get_prompt_encoder(vl_model, api):
    """model/prompt_encoder.py:22-33.  CONCH: the HIP text tower adopting ``vl_model.text``.  The OpenAI-CLIP / HuggingFace
    towers are alternative backbones no shipped config uses (cfg_vlsa_conch.yaml:39) and are not built."""
    if api == "CONCH":
        from .prompt_encoder import CONCHPromptEncoder
        return CONCHPromptEncoder(vl_model)
    if api in ("CLIP", "HF"):
        raise NotImplementedError(f"vlsa_api={api!r}: only the CONCH text tower has a HIP implementation")
    raise ValueError(f"Got an invalid api ({api}).")


def load_prompt_learner(learner_name: str, cfg: dict):
    """model/prompt_learners/__init__.py:6-18: 'plain' | 'rank'; anything else yields None there too."""
    from .prompt_learner import PlainPromptLearner, RankPromptLearner
    if learner_name == "plain":
        return PlainPromptLearner(**cfg)
    if learner_name == "rank":
        return RankPromptLearner(**cfg)
    return None


def load_prompt_adapter(prompt_encoder, cfg: dict):
    """model/prompt_learners/__init__.py:20-24"""
    from .prompt_adapter import PromptAdapter
    return PromptAdapter(prompt_encoder, **cfg)


def _sub_cfg(cfg: dict, prefix: str) -> dict:
    """the keys ``<prefix>_<name>`` of a flat run config as ``{name: value}`` (what utils/func.py:136-147 ``fetch_kws`` hands the
    handler; names shorter than two characters are not picked up there either)"""
    cut = len(prefix) + 1
    return {k[cut:]: v for k, v in cfg.items() if k.startswith(prefix) and len(k) > cut and k[len(prefix)] == "_"}


def arch_cfg_from_run_cfg(cfg: dict) -> dict:
    """A run's flat config (cfg_vlsa_conch.yaml / a run directory's config.yaml) -> the keyword arguments of
    ``load_model('VLSA', **arch_cfg)``, as ``VLSAHandler.func_load_model`` assembles them (runner/vlsa_handler.py:88-120)."""
    arch = cfg["arch"].lower()
    assert f"{arch}_api" in cfg, "Please specify the API for VLSA models."
    learner = cfg["vlsa_pmt_learner_name"]
    learner_cfg = _sub_cfg(cfg, f"{arch}_pmt_learner_{learner.lower()}")
    pretrained = bool(cfg.get("vlsa_pmt_learner_pretrained", False))
    learner_cfg.update(name=learner, pretrained=pretrained)
    coop_cfg = None
    if pretrained:          # text prompts pre-trained by CoOp: the checkpoint path is a template over (split seed, method)
        coop_cfg = _sub_cfg(cfg, "vlsa_pmt_learner_coop")
        assert coop_cfg.get("ckpt") is not None, "Found null ckpt path."
        coop_cfg["ckpt"] = coop_cfg["ckpt"].format(cfg["data_split_seed"], coop_cfg["method"])
    return dict(vlsa_api=cfg[f"{arch}_api"], text_encoder_cfg=_sub_cfg(cfg, f"{arch}_txt_encoder"),
                image_encoder_cfg=_sub_cfg(cfg, f"{arch}_img_encoder"), prompt_learner_cfg=learner_cfg,
                pretrained_prompt_learner_cfg=coop_cfg, path_clip_model=cfg["path_clip_model"])


def func_load_model(cfg: dict):
    """``VLSAHandler.func_load_model`` (runner/vlsa_handler.py:88-151) without the handler: build the model from a run config and
    freeze what the config freezes (prompt embeddings, MIL encoder, text tower, logit scale).  ``init_wt`` (a generic re-init of
    every Linear, off in every shipped config) is refused rather than silently ignored."""
    if cfg.get("init_wt"):
        raise NotImplementedError("init_wt: True re-initialises the model with utils/func.py:general_init_weight; run it on the result if wanted")
    arch_cfg = arch_cfg_from_run_cfg(cfg)
    model = load_model(cfg["arch"], **arch_cfg)
    arch = cfg["arch"].lower()
    learner = cfg["vlsa_pmt_learner_name"]
    freeze = []
    if learner == "CoOp":
        freeze += [(model.prompt_learner.context_embeds, arch_cfg["prompt_learner_cfg"]["frozen_context_embeds"]),
                   (model.prompt_learner.rank_embeds, arch_cfg["prompt_learner_cfg"]["frozen_rank_embeds"])]
    if learner in ("CoOp", "Adapter"):
        tower = model.prompt_encoder if hasattr(model, "prompt_encoder") else getattr(model, "text_encoder", None)
        freeze += [(model.mil_encoder, arch_cfg["image_encoder_cfg"]["frozen"]), (tower, arch_cfg["text_encoder_cfg"]["frozen"]),
                   (model.logit_scale, cfg[f"{arch}_frozen_logit_scale"])]
    for obj, frozen in freeze:
        if frozen and obj is not None:
            for p in (obj.parameters() if hasattr(obj, "parameters") else [obj]):
                p.requires_grad = False
    return model


def patch_reference(resident_bags: bool = False, defer_training_calls: bool = True, **resident_kw):
    """Point the reference's factory at this package (call once, before the handler is built):

        import vlsa_amd.model_utils; vlsa_amd.model_utils.patch_reference()

    * ``model.utils.VLSA`` -> ``vlsa_amd.vlsa.VLSA``: ``load_model('VLSA', **arch_cfg)`` (model/utils.py:36,44) then builds the HIP
      model from the handler's unmodified ``arch_cfg``;
    * ``model.deepmil.{VLFAN, FeatMIL, DeepMIL, logit_pooling}`` -> this package's: the name lookup of
      model/utils_vl.py:129-138 and ``utils/model_inference.py`` see the same classes.
    * ``resident_bags=True``: ``dataset.utils.prepare_surv_dataset`` (and the name ``runner.sa_handler`` imported from it, if already
      loaded) wraps what it returns in ``vlsa_amd.ingest.ResidentBags`` -- every bag is read and uploaded ONCE, later epochs find it in
      HBM.  Stored as **fp32** here, bit for bit what dataset/PatchWSI.py:214 hands the handler (an "unmodified" reference run must
      not be fed rounded features silently); ``dtype=torch.bfloat16`` in ``resident_kw`` halves the footprint and the streaming
      time at ~1e-2 on the logits.  Needs ``num_workers: 0`` in the run's config (a worker process has no device: items pass
      through unchanged there, with a one-time warning).
    * ``defer_training_calls`` (default True): ``VLSA.defer_training_calls`` -- the handler's ``net(xs[i])`` calls of a training batch run
      as ONE batched forward the moment ``torch.cat(y_hat)`` looks at them (vlsa_amd/deferred.py): 1.9 instead of 4.5 ms per 32-bag step.
    Returns the patched reference modules (for un-patching in tests)."""
    import model.deepmil as ref_mil
    import model.utils as ref_utils
    import model.vlsa as ref_vlsa
    from . import deepmil as fast
    from .vlsa import VLSA
    saved = dict(VLSA_utils=ref_utils.VLSA, VLSA_vlsa=ref_vlsa.VLSA, VLFAN=ref_mil.VLFAN, FeatMIL=ref_mil.FeatMIL,
                 DeepMIL=ref_mil.DeepMIL, logit_pooling=ref_mil.logit_pooling)
    ref_utils.VLSA = VLSA
    ref_vlsa.VLSA = VLSA
    # the handler's bag-by-bag training loop (runner/vlsa_handler.py:260-289) at the batched step's speed: its net(X) calls are recorded
    # and run as ONE forward_bags when torch.cat first looks at a prediction (vlsa_amd/deferred.py)
    saved["defer_training_calls"] = VLSA.defer_training_calls
    VLSA.defer_training_calls = defer_training_calls
    ref_mil.VLFAN, ref_mil.FeatMIL, ref_mil.DeepMIL = fast.VLFAN, fast.FeatMIL, fast.DeepMIL
    ref_mil.logit_pooling = ref_vlsa.logit_pooling = fast.logit_pooling
    if resident_bags:
        import sys
        import dataset.utils as ref_ds
        from .ingest import ResidentBags
        original = ref_ds.prepare_surv_dataset
        if not getattr(original, "_vlsa_resident", False):
            import torch
            resident_kw.setdefault("dtype", torch.float32)

            def prepare_surv_dataset(*args, **kwargs):
                return ResidentBags(original(*args, **kwargs), **resident_kw)
            prepare_surv_dataset._vlsa_resident = True
            prepare_surv_dataset.__wrapped__ = original
            saved["prepare_surv_dataset"] = original
            ref_ds.prepare_surv_dataset = prepare_surv_dataset
            for name in ("runner.sa_handler", "runner.vlsa_handler", "runner.base_handler"):
                mod = sys.modules.get(name)
                if mod is not None and getattr(mod, "prepare_surv_dataset", None) is original:
                    mod.prepare_surv_dataset = prepare_surv_dataset
    return saved


def unpatch_reference(saved) -> None:
    """Undo ``patch_reference`` (``saved`` = what it returned): the reference's own classes, dataset factory and the class-wide
    ``VLSA.defer_training_calls`` flag are back as they were."""
    import model.deepmil as ref_mil
    import model.utils as ref_utils
    import model.vlsa as ref_vlsa
    from .vlsa import VLSA
    ref_utils.VLSA, ref_vlsa.VLSA = saved["VLSA_utils"], saved["VLSA_vlsa"]
    ref_mil.VLFAN, ref_mil.FeatMIL, ref_mil.DeepMIL = saved["VLFAN"], saved["FeatMIL"], saved["DeepMIL"]
    ref_mil.logit_pooling = ref_vlsa.logit_pooling = saved["logit_pooling"]
    VLSA.defer_training_calls = saved.get("defer_training_calls", False)
    if "prepare_surv_dataset" in saved:
        import sys
        import dataset.utils as ref_ds
        patched = ref_ds.prepare_surv_dataset
        ref_ds.prepare_surv_dataset = saved["prepare_surv_dataset"]
        for name in ("runner.sa_handler", "runner.vlsa_handler", "runner.base_handler"):
            mod = sys.modules.get(name)
            if mod is not None and getattr(mod, "prepare_surv_dataset", None) is patched:
                mod.prepare_surv_dataset = saved["prepare_surv_dataset"]
