"""Shared inputs of the handler-seam fixture (tests/golden/handler_loop.npz): a cfg dict with the key surface of the
reference's cfg_vlsa_conch.yaml (config/IFMLE/tcga_blca/cfg_vlsa_conch.yaml:39-118, every value synthetic), the stand-ins for
the two host-side loaders (tokenizer, pretrained VL model) and the seeded bags / labels / MIL parameters.  Used by the
generator (tests/golden/make_golden_handler.py, which runs the REFERENCE's handler code on them) and by the tests (which run
this package's model through the same call pattern).  Everything is synthetic: made-up token ids, a random small text
tower; no reference text or weights."""
from __future__ import annotations

import json
import math
import os
import types

import torch
import torch.nn as nn

import cases
import text_cases as TC

TOWER, TOWER_SEED = "train", 9101            # width 128, 2 layers, text features in the [*, 512] CONCH space
P, K, BASE_RANKS = 5, 6, 4                   # prototype queries, ordinal ranks (= time bins), base rank prompts
PARAM_SEED = 9102
SIZES = (300, 517, 900, 64)
LABELS_T, LABELS_E = (0, 3, 5, 2), (1, 0, 1, 0)
STEPS = 3
EVAL_SIZES = (257, 64, 1000)


def prompt_table():
    """text -> made-up token ids (< vocab 64; 0 pad, 1 bos, 2 eos): "X.", a context, 4 rank names, 5 prototype sentences."""
    table, ctx_key, rank_keys = TC.synthetic_prompt_table(TC.TOWERS[TOWER]["vocab"], TOWER_SEED, n_ctx=4, rank_lens=(4, 3, 3, 4))
    g = torch.Generator().manual_seed(TOWER_SEED + 7)
    protos = []
    for i, n in enumerate((9, 14, 3, 20, 6)):
        table[f"proto{i}"] = torch.randint(3, TC.TOWERS[TOWER]["vocab"], (n,), generator=g).tolist()
        protos.append(f"proto{i}")
    return table, ctx_key, rank_keys, protos


def write_prompt_files(tmpdir):
    """The two JSON files the cfg points at (same schema as tools/survival_prompts.json / survival_text_prototypes.json)."""
    table, ctx_key, rank_keys, protos = prompt_table()
    p_init = os.path.join(tmpdir, "init_prompts.json")
    p_proto = os.path.join(tmpdir, "prototypes.json")
    with open(p_init, "w") as f:
        json.dump({"context_templates": [ctx_key], "class_names": {str(i): [k] for i, k in enumerate(rank_keys)}}, f)
    with open(p_proto, "w") as f:
        json.dump({"synth_0": protos}, f)
    return p_init, p_proto


def make_tokenizer(**_):
    table, *_rest = prompt_table()
    return TC.ReplayTokenizer(table, bos=1, eos=2, pad=0)


def make_cfg(p_init, p_proto, **overrides):
    """cfg_vlsa_conch.yaml's model / loss / optimizer keys (placeholders already filled, list-valued sweep keys collapsed)."""
    cfg = {
        "task": "vlsa", "arch": "VLSA", "path_clip_model": "/nonexistent/pretrained-models", "init_wt": False,
        "net_output_converter": "softmax", "model_saver_module_filter": "prompt_encoder", "data_split_seed": 0,
        "time_bins": K,
        "vlsa_api": "CONCH", "vlsa_frozen_logit_scale": False,
        "vlsa_img_encoder_name": "VLFAN", "vlsa_img_encoder_frozen": False, "vlsa_img_encoder_dim_in": 512,
        "vlsa_img_encoder_dim_hid": 256, "vlsa_img_encoder_use_feat_proj": False, "vlsa_img_encoder_drop_rate": 0.25,
        "vlsa_img_encoder_pred_head": "default", "vlsa_img_encoder_dim_reduction": 4, "vlsa_img_encoder_keep_ratio": 0.8,
        "vlsa_img_encoder_query": "Text", "vlsa_img_encoder_num_query": P, "vlsa_img_encoder_query_pooling": "mean",
        "vlsa_img_encoder_gated_query": False, "vlsa_img_encoder_query_text_method": "TaskRes",
        "vlsa_img_encoder_query_text_res_ratio": 0.5, "vlsa_img_encoder_query_text_dim_reduction": 4,
        "vlsa_img_encoder_query_text_keep_ratio": 0.8, "vlsa_img_encoder_query_text_load_path": p_proto,
        "vlsa_img_encoder_query_text_load_idx": "synth_0",
        "vlsa_txt_encoder_name": "mahmoodlab/conch", "vlsa_txt_encoder_frozen": True,
        "vlsa_pmt_learner_name": "CoOp", "vlsa_pmt_learner_pretrained": False,
        "vlsa_pmt_learner_coop_ckpt": None, "vlsa_pmt_learner_coop_method": "rank", "vlsa_pmt_learner_coop_num_ranks": K,
        "vlsa_pmt_learner_coop_num_base_ranks": BASE_RANKS, "vlsa_pmt_learner_coop_num_tokens_per_rank": 4,
        "vlsa_pmt_learner_coop_num_context_tokens": 8, "vlsa_pmt_learner_coop_rank_tokens_position": "tail",
        "vlsa_pmt_learner_coop_init_prompt_path": p_init, "vlsa_pmt_learner_coop_init_prompt_rank_idx": 0,
        "vlsa_pmt_learner_coop_init_prompt_context_idx": 0, "vlsa_pmt_learner_coop_rank_specific_context": False,
        "vlsa_pmt_learner_coop_frozen_context_embeds": False, "vlsa_pmt_learner_coop_frozen_rank_embeds": False,
        "vlsa_pmt_learner_adapter_method": "default", "vlsa_pmt_learner_adapter_num_ranks": K,
        "vlsa_pmt_learner_adapter_res_ratio": 0.5, "vlsa_pmt_learner_adapter_dim_reduction": 4,
        "vlsa_pmt_learner_adapter_keep_ratio": 0.8, "vlsa_pmt_learner_adapter_init_prompt_path": p_init,
        "vlsa_pmt_learner_adapter_init_prompt_rank_idx": 0, "vlsa_pmt_learner_adapter_init_prompt_context_idx": 0,
        "loss_type": "SurvIFMLE-SurvEMD", "loss_survifmle_weight": 1.0, "loss_survemd_weight": 1.0, "loss_survemd_p": 2,
        "evaluator": "VL-IF", "opt_name": "adam", "opt_lr": 0.0002, "opt_weight_decay": 0.00001, "bp_every_batch": 32,
    }
    cfg.update(overrides)
    return cfg


def tower_state():
    return TC.make_tower_weights(TOWER, TOWER_SEED)


def make_coca_stub(build_text_tower=None):
    """What the VL-model loader returns: ``.text`` = a CoCa text tower carrying the seeded weights, ``.logit_scale``.
    ``build_text_tower(out_dim, text_cfg)``: the reference's own ``_build_text_tower`` (generator) -- None: this package's
    tower container with the same attribute names."""
    c = TC.TOWERS[TOWER]
    W = tower_state()
    if build_text_tower is not None:
        text_cfg = dict(context_length=c["ctx"], vocab_size=c["vocab"], width=c["width"], heads=c["heads"], layers=c["layers"],
                        embed_cls=True, output_tokens=True)
        tower = build_text_tower(c["out_dim"], text_cfg)
        missing, unexpected = tower.load_state_dict(W, strict=False)
        assert not unexpected and all(k == "attn_mask" for k in missing), (missing, unexpected)
    else:
        from vlsa_amd.prompt_encoder import CONCHPromptEncoder
        enc = CONCHPromptEncoder(width=c["width"], heads=c["heads"], layers=c["layers"], context_length=c["ctx"],
                                 vocab_size=c["vocab"], output_dim=c["out_dim"])
        enc.load_state_dict(W, strict=True)
        tower = types.SimpleNamespace(pad_id=0, heads=c["heads"], positional_embedding=enc.positional_embedding,
                                      transformer=enc.transformer, ln_final=enc.ln_final, cls_emb=enc.cls_emb,
                                      text_projection=enc.text_projection, token_embedding=enc.token_embedding)
    return types.SimpleNamespace(text=tower, visual=None, logit_scale=nn.Parameter(torch.ones([]) * math.log(1 / 0.07)))


def mil_state():
    """Seeded values for the MIL side, loaded with ``load_state_dict(strict=False)`` as the handler loads checkpoints
    (runner/vlsa_handler.py:317-318)."""
    p = cases.make_params(P, K, PARAM_SEED)
    return {"mil_encoder.Q.residual_features": p["resid"], "mil_encoder.visual_adapter.weight": p["W"],
            "mil_encoder.visual_adapter.bias": p["b"], "logit_scale": torch.tensor(cases.LOGIT_SCALE)}


def train_batch():
    xs = [cases.make_bag(n, PARAM_SEED + 10 + i, "clustered" if i % 2 else "iid")[None] for i, n in enumerate(SIZES)]
    ys = [torch.tensor([[float(t), float(e)]]) for t, e in zip(LABELS_T, LABELS_E)]
    return xs, ys


def eval_loader():
    """[(idx [1, 1], (X [1, N, 512],), y [1, 2])] -- what DataLoader(batch_size=1) over WSIPatchSurv yields
    (dataset/PatchWSI.py:197-215)."""
    out = []
    for i, n in enumerate(EVAL_SIZES):
        X = cases.make_bag(n, PARAM_SEED + 50 + i, "iid" if i % 2 else "clustered")[None]
        out.append((torch.tensor([[i]]), (X,), torch.tensor([[float(i % K), float(i % 2)]])))
    return out
