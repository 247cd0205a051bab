"""The reference's construction surface: ``load_model('VLSA', text_encoder_cfg=..., image_encoder_cfg=...,
prompt_learner_cfg=..., pretrained_prompt_learner_cfg=..., vlsa_api=..., path_clip_model=...)`` exactly as
``VLSAHandler.func_load_model`` calls it (runner/vlsa_handler.py:88-151, model/utils.py:13-45, model/vlsa.py:22-147), from a
cfg dict with the key surface of cfg_vlsa_conch.yaml.  No GPU: construction, state-dict keys, freezing, checkpoint loading.
The tokenizer and the pretrained VL model come through the package's hooks (synthetic stand-ins, tests/golden/handler_cases.py).
"""
import json
import os
import tempfile

import pytest
import torch
import torch.nn as nn

import _ref_import
import handler_cases as HC
import handler_loop as HL

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture()
def hooks_installed():
    from vlsa_amd import hooks
    prev_t = hooks.set_tokenizer_factory(HC.make_tokenizer)
    prev_m = hooks.set_vl_model_loader(lambda **kw: HC.make_coca_stub())
    yield
    hooks.set_tokenizer_factory(prev_t)
    hooks.set_vl_model_loader(prev_m)


def _build(**overrides):
    from vlsa_amd.model_utils import load_model
    with tempfile.TemporaryDirectory() as tmp:
        p_init, p_proto = HC.write_prompt_files(tmp)
        cfg = HC.make_cfg(p_init, p_proto, **overrides)
        return HL.build_model(cfg, load_model), cfg


def test_model_from_cfg_has_the_reference_key_set(hooks_installed):
    """Same saved key set, shapes and trainable-parameter set as the model the REFERENCE's func_load_model builds from the
    same cfg (tests/golden/handler_keys.json, written by make_golden_handler.py)."""
    from vlsa_amd.vlsa import VLSA
    ref = json.load(open(os.path.join(GOLDEN, "handler_keys.json")))
    model, cfg = _build()
    assert isinstance(model, VLSA) and model.pmt_learner_name == "CoOp"
    saved = {k: list(v.shape) for k, v in model.state_dict().items() if cfg["model_saver_module_filter"] not in k}
    assert saved == ref["saved"]
    assert sorted(n for n, p in model.named_parameters() if p.requires_grad) == ref["requires_grad"]
    # what the handler touches on the model (runner/vlsa_handler.py:126-149,186,249,303; utils/model_inference.py:93-129)
    for attr in ("prompt_learner", "prompt_encoder", "mil_encoder", "logit_scale", "text_tokenizer", "get_logit_scale",
                 "forward_text_only", "encode_instances", "text_encoder_cfg", "image_encoder_cfg", "prompt_learner_cfg"):
        assert hasattr(model, attr), attr
    assert isinstance(model.logit_scale, nn.Parameter) and model.logit_scale.dim() == 0
    assert all(not p.requires_grad for p in model.prompt_encoder.parameters())        # vlsa_txt_encoder_frozen: True
    assert callable(model.mil_encoder.query_div_loss) and model.mil_encoder.Q.method == "TaskRes"
    # checkpoints load the way the handler loads them (strict=False, runner/vlsa_handler.py:317-318)
    missing, unexpected = model.load_state_dict(HC.mil_state(), strict=False)
    assert not unexpected
    assert torch.equal(model.mil_encoder.visual_adapter.bias, HC.mil_state()["mil_encoder.visual_adapter.bias"])


def test_conch_sized_model_matches_the_shipped_checkpoint_keys():
    """With a CONCH-sized tower (768 wide, 128 positions) the saved keys AND shapes are those of the reference's shipped
    run directory (assert/blca-train-VLSA/train_model-last.pth, pinned in ckpt_keys.json): P = 12 queries, 4 base ranks."""
    import types
    from vlsa_amd import hooks
    from vlsa_amd.model_utils import load_model
    from vlsa_amd.prompt_encoder import CONCHPromptEncoder
    import text_cases as TC

    def conch_stub(**kw):
        enc = CONCHPromptEncoder(width=768, heads=12, layers=1, context_length=128, vocab_size=64, output_dim=512)
        tower = types.SimpleNamespace(pad_id=0, heads=12, positional_embedding=enc.positional_embedding, transformer=enc.transformer,
                                      ln_final=enc.ln_final, cls_emb=enc.cls_emb, text_projection=enc.text_projection,
                                      token_embedding=enc.token_embedding)
        return types.SimpleNamespace(text=tower, logit_scale=nn.Parameter(torch.tensor(2.0)))

    prev_t = hooks.set_tokenizer_factory(HC.make_tokenizer)
    prev_m = hooks.set_vl_model_loader(conch_stub)
    try:
        with tempfile.TemporaryDirectory() as tmp:
            p_init, p_proto = HC.write_prompt_files(tmp)
            with open(p_proto, "w") as f:
                json.dump({"synth_0": [f"proto{i % 5}" for i in range(12)]}, f)
            cfg = HC.make_cfg(p_init, p_proto, vlsa_img_encoder_num_query=12, vlsa_pmt_learner_coop_num_ranks=12, time_bins=12)
            model = HL.build_model(cfg, load_model)
    finally:
        hooks.set_tokenizer_factory(prev_t)
        hooks.set_vl_model_loader(prev_m)
    ck = json.load(open(os.path.join(GOLDEN, "ckpt_keys.json")))["model"]
    saved = {k: list(v.shape) for k, v in model.state_dict().items() if "prompt_encoder" not in k}
    assert saved == ck
    assert float(model.logit_scale) == 2.0          # the VL model's own parameter is adopted (model/vlsa.py:105)


def test_pretrained_frozen_coop_prompts_short_circuit(hooks_installed):
    """pretrained + frozen context / rank embeddings: the learner's embeddings come from the checkpoint and the model has
    the reference's ``pretrained_text_features`` switch (model/vlsa.py:57-60,117-122,160-161)."""
    model0, _ = _build()
    with tempfile.TemporaryDirectory() as tmp:
        ck = os.path.join(tmp, "fold{}-{}.pth")
        sd = {"prompt_learner.context_embeds": torch.full_like(model0.prompt_learner.context_embeds.detach(), 0.25),
              "prompt_learner.rank_embeds": torch.full_like(model0.prompt_learner.rank_embeds.detach(), -0.5)}
        torch.save({"model": sd}, ck.format(0, "rank"))
        model, cfg = _build(vlsa_pmt_learner_pretrained=True, vlsa_pmt_learner_coop_ckpt=ck,
                            vlsa_pmt_learner_coop_frozen_context_embeds=True, vlsa_pmt_learner_coop_frozen_rank_embeds=True)
    assert hasattr(model, "pretrained_text_features")                 # the reference's switch
    assert "pretrained_text_features" not in model.state_dict()       # non-persistent there too (model/vlsa.py:60)
    assert torch.all(model.prompt_learner.context_embeds == 0.25) and torch.all(model.prompt_learner.rank_embeds == -0.5)
    assert not model.prompt_learner.context_embeds.requires_grad and not model.prompt_learner.rank_embeds.requires_grad
    # the tower pass behind the switch needs the device: on the CPU it refuses instead of falling back
    from vlsa_amd import VlsaNativeError
    with pytest.raises(VlsaNativeError):
        model.forward_text_only()


def test_adapter_prompt_learner_route(hooks_installed):
    """vlsa_pmt_learner_name: Adapter (model/vlsa.py:65-66,124-147): a text-side PromptAdapter over the rank sentences
    (context template with CLASSNAME replaced), registered as ``prompt_adapter``; no CoOp learner."""
    with tempfile.TemporaryDirectory() as tmp:
        p_init, p_proto = HC.write_prompt_files(tmp)
        with open(p_init, "w") as f:      # 6 rank sentences = num_ranks; every sentence must be in the replay table
            json.dump({"context_templates": ["CLASSNAME"], "class_names": {str(i): [f"rank{i % 4}"] for i in range(HC.K)}}, f)
        from vlsa_amd.model_utils import load_model
        cfg = HC.make_cfg(p_init, p_proto, vlsa_pmt_learner_name="Adapter", vlsa_pmt_learner_adapter_method="TaskRes")
        model = HL.build_model(cfg, load_model)
    assert model.pmt_learner_name == "Adapter" and not hasattr(model, "prompt_learner")
    keys = {k for k in model.state_dict() if "prompt_encoder" not in k}
    assert keys == {"logit_scale", "mil_encoder.Q.residual_features", "mil_encoder.visual_adapter.weight",
                    "mil_encoder.visual_adapter.bias", "prompt_adapter.residual_features"}
    assert tuple(model.prompt_adapter.residual_features.shape) == (HC.K, 512)


def test_invalid_names_raise_like_the_reference(hooks_installed):
    from vlsa_amd.model_utils import get_prompt_encoder, load_model
    with pytest.raises(NotImplementedError):
        load_model("ResNet")
    with pytest.raises(ValueError):
        get_prompt_encoder(None, "OpenCLIP")
    with tempfile.TemporaryDirectory() as tmp:
        p_init, p_proto = HC.write_prompt_files(tmp)
        with pytest.raises(ValueError):
            HL.build_model(HC.make_cfg(p_init, p_proto, vlsa_pmt_learner_name="LoRA"), load_model)
        with pytest.raises(AssertionError):
            a = HL.arch_cfg_of(HC.make_cfg(p_init, p_proto))
            a.pop("vlsa_api")
            load_model("VLSA", **a)


@pytest.mark.skipif(not _ref_import.reference_available(), reason="needs the reference checkout (build container only)")
def test_reference_handler_builds_the_hip_model_after_one_line_patch(hooks_installed):
    """``patch_reference()`` and then the reference's REAL, unmodified ``VLSAHandler.func_load_model(cfg)``: it returns this
    package's VLSA with the reference model's saved key set."""
    _ref_import.import_reference()
    cwd = os.getcwd()
    os.chdir(_ref_import.REF_ROOT)
    import model.deepmil as ref_mil
    import model.utils as ref_utils
    import model.vlsa as ref_vlsa
    from runner.vlsa_handler import VLSAHandler
    from vlsa_amd.model_utils import patch_reference, unpatch_reference
    from vlsa_amd.vlsa import VLSA
    saved = patch_reference()
    try:
        with tempfile.TemporaryDirectory() as tmp:
            p_init, p_proto = HC.write_prompt_files(tmp)
            model = VLSAHandler.func_load_model(HC.make_cfg(p_init, p_proto))
    finally:
        assert VLSA.defer_training_calls is True                      # (patch_reference switches the handler's training loop to deferred calls)
        unpatch_reference(saved)
        assert VLSA.defer_training_calls is False and ref_utils.VLSA is saved["VLSA_utils"] and ref_mil.VLFAN is saved["VLFAN"]
        os.chdir(cwd)
    assert isinstance(model, VLSA)
    ref = json.load(open(os.path.join(GOLDEN, "handler_keys.json")))
    assert {k: list(v.shape) for k, v in model.state_dict().items() if "prompt_encoder" not in k} == ref["saved"]
    assert sorted(n for n, p in model.named_parameters() if p.requires_grad) == ref["requires_grad"]


@pytest.mark.skipif(not _ref_import.reference_available(), reason="needs the reference checkout (build container only)")
def test_patch_reference_can_make_the_handlers_datasets_resident(hooks_installed):
    """``patch_reference(resident_bags=True)``: the dataset factory the reference's handlers call
    (runner/sa_handler.py:111-116 -> dataset/utils.py prepare_surv_dataset) hands back its dataset wrapped in ResidentBags."""
    _ref_import.import_reference()
    cwd = os.getcwd()
    os.chdir(_ref_import.REF_ROOT)
    import dataset.utils as ref_ds
    import model.deepmil as ref_mil
    import model.utils as ref_utils
    import model.vlsa as ref_vlsa
    import runner.sa_handler as ref_sa
    from vlsa_amd.ingest import ResidentBags
    from vlsa_amd.model_utils import patch_reference
    original = ref_ds.prepare_surv_dataset
    assert ref_sa.prepare_surv_dataset is original
    ref_ds.prepare_surv_dataset = lambda *a, **k: ("a dataset", a, k)      # stands in for the file-reading factory
    ref_sa.prepare_surv_dataset = ref_ds.prepare_surv_dataset
    saved = patch_reference(resident_bags=True, dtype=torch.float32)
    try:
        out = ref_sa.prepare_surv_dataset(["p1"], {"cfg": 1}, meta_data=None)
        assert isinstance(out, ResidentBags) and out.dataset == ("a dataset", (["p1"], {"cfg": 1}), {"meta_data": None})
        assert out._dtype == torch.float32
        assert ref_ds.prepare_surv_dataset is ref_sa.prepare_surv_dataset
        patch_reference(resident_bags=True)                                # idempotent: not wrapped twice
        assert not isinstance(ref_sa.prepare_surv_dataset(["p1"], {}).dataset, ResidentBags)
    finally:
        from vlsa_amd.model_utils import unpatch_reference
        unpatch_reference(saved)
        ref_ds.prepare_surv_dataset = ref_sa.prepare_surv_dataset = original
        os.chdir(cwd)


def test_func_load_model_and_the_run_directory_loader(hooks_installed):
    """``vlsa_amd.model_utils.func_load_model`` = the handler's static method without the handler (runner/vlsa_handler.py:88-151):
    same arch_cfg, same model, same frozen set as the handler-shaped test builder; ``vlsa_amd.inference._read_run_cfg`` reads a run
    directory's config.yaml or print_config.txt (utils/func.py:219-241).  (``load_vlsa_model`` itself needs the device: GPU test.)"""
    import yaml
    from vlsa_amd.inference import _read_run_cfg
    from vlsa_amd.model_utils import arch_cfg_from_run_cfg, func_load_model
    with tempfile.TemporaryDirectory() as tmp:
        p_init, p_proto = HC.write_prompt_files(tmp)
        cfg = HC.make_cfg(p_init, p_proto)
        assert arch_cfg_from_run_cfg(cfg) == HL.arch_cfg_of(cfg)
        a, (b, _) = func_load_model(cfg), _build()
        assert [n for n, p in a.named_parameters() if p.requires_grad] == [n for n, p in b.named_parameters() if p.requires_grad]
        assert {k: tuple(v.shape) for k, v in a.state_dict().items()} == {k: tuple(v.shape) for k, v in b.state_dict().items()}
        with pytest.raises(NotImplementedError):
            func_load_model(dict(cfg, init_wt=True))
        run = os.path.join(tmp, "run")
        os.makedirs(run)
        with open(os.path.join(run, "config.yaml"), "w") as f:
            yaml.safe_dump(cfg, f)
        assert _read_run_cfg(run) == cfg
        os.remove(os.path.join(run, "config.yaml"))
        with open(os.path.join(run, "print_config.txt"), "w") as f:
            f.write("header line\n" + "".join(f"{k} --> {v!r}\n" for k, v in cfg.items()))
        assert _read_run_cfg(run) == cfg
        with pytest.raises(RuntimeError):
            _read_run_cfg(tmp)


def test_provider_key_sees_reassigned_parameters_without_walking_per_call(hooks_installed):
    """The text-feature cache key reads kept module / tensor lists (no walk per ``net(X)`` call); the lists are exact because every
    parameter / buffer / submodule registration in the process bumps a structure epoch (torch's global registration hooks).  A
    re-assigned learner parameter -- same ``_version`` as the old one -- must change the key; an unrelated module built elsewhere must not."""
    model, _ = _build()
    walks = []
    orig = type(model)._walk_module
    type(model)._walk_module = staticmethod(lambda m: (walks.append(type(m).__name__), orig(m))[1])
    try:
        k0 = model._provider_key()
        n0 = len(walks)
        assert model._provider_key() == k0 and len(walks) == n0                       # steady state: no walk at all
        nn.Linear(3, 3)                                                                # somebody builds a module: lists are re-walked ...
        assert model._provider_key() == k0 and len(walks) > n0                        # ... found unchanged: same key
        old = model.prompt_learner.context_embeds
        model.prompt_learner.context_embeds = nn.Parameter(old.detach().clone() + 1.0)
        assert model.prompt_learner.context_embeds._version == old._version
        k1 = model._provider_key()
        assert k1 != k0                                                                # the new object is seen
        with torch.no_grad():
            model.prompt_learner.context_embeds.add_(1.0)
        assert model._provider_key() != k1                                             # in-place change: version
        model.prompt_encoder.eval() if model.prompt_encoder.training else model.prompt_encoder.train()
        assert model._provider_key() != k1                                             # train / eval flag
    finally:
        type(model)._walk_module = staticmethod(orig)
