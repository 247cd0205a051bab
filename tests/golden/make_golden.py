"""Generate the golden vectors by RUNNING THE REFERENCE (container-only; /root/reference must exist).

    python tests/golden/make_golden.py

Writes tests/golden/*.npz (+ ckpt_keys.json).  Each fixture holds the parameters that were used
(so a test does not depend on torch's module-init RNG), a checksum of the seeded bag and the outputs
of the reference's own ``VLSA.forward`` / ``VLFAN`` / ``FeatMIL`` / ``DeepMIL`` / ``logit_pooling`` /
``calc_text_img_similarity`` and of torch.autograd through them.  Only synthetic tensors are stored:
no reference source, weights or prompt text.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402
import cases  # noqa: E402

torch.set_num_threads(4)


def _np(t):
    return t.detach().cpu().numpy().copy()


def fake_vlsa(ref, encoder, T, logit_scale, pooling=None, text_requires_grad=False):
    """Real VLSA.forward without CONCH."""
    VLSA = ref.vlsa.VLSA
    m = VLSA.__new__(VLSA)
    nn.Module.__init__(m)
    m.mil_encoder = encoder
    m.logit_scale = nn.Parameter(torch.tensor(float(logit_scale)))
    m.image_encoder_cfg = {"pooling": pooling}
    m.pmt_learner_name = "CoOp"
    if text_requires_grad:
        m.text_param = nn.Parameter(T.clone())
        m.forward_text_only = lambda: m.text_param
    else:
        m.register_buffer("pretrained_text_features", T.clone(), persistent=False)
    return m


def build_vlfan(ref, P, pooling, head, gated, params, seed):
    torch.manual_seed(seed)
    VLFAN = ref.deepmil.VLFAN
    if gated:
        enc = VLFAN(dim_in=cases.D, dim_hid=256, use_feat_proj=False, drop_rate=0.25, query="Parameter",
                    num_query=P, gated_query=True, query_pooling=pooling, pred_head=head)
        with torch.no_grad():
            enc.Q.copy_(0.5 * params["resid"] + params["prompt"])
    else:
        enc = VLFAN(dim_in=cases.D, dim_hid=256, use_feat_proj=False, drop_rate=0.25, query="Text",
                    num_query=P, gated_query=False, query_pooling=pooling, pred_head=head)
        qnet = ref.prompt_adapter.PromptAdapter(None, method="TaskRes", num_prompts=P,
                                                pretrained_prompt_features=params["prompt"].clone(),
                                                res_ratio=0.5)
        with torch.no_grad():
            qnet.residual_features.copy_(params["resid"])
        enc.reset_query(qnet)
    if head != "Identity":
        with torch.no_grad():
            enc.visual_adapter.weight.copy_(params["W"])
            enc.visual_adapter.bias.copy_(params["b"])
    load_pool_params(enc.query_pooling, cases.make_pool_params(pooling, seed + 3000), P)
    enc.eval()  # dropout in Gated_Attention_Pooling off
    return enc


def load_pool_params(mod, pp, P=None):
    """Copy seeded tensors into the reference's pooling module (names: model/layers.py:90-101,131-135)."""
    with torch.no_grad():
        if isinstance(mod, nn.Parameter):
            mod.copy_(pp["weight"][:, :P])
        elif type(mod).__name__ == "Attention_Pooling":
            mod.attention[0].weight.copy_(pp["w1"]); mod.attention[0].bias.copy_(pp["b1"])
            mod.attention[2].weight.copy_(pp["w2"]); mod.attention[2].bias.copy_(pp["b2"])
        elif type(mod).__name__ == "Gated_Attention_Pooling":
            mod.fc1[0].weight.copy_(pp["wa"]); mod.fc1[0].bias.copy_(pp["ba"])
            mod.score[0].weight.copy_(pp["wg"]); mod.score[0].bias.copy_(pp["bg"])
            mod.fc2.weight.copy_(pp["w2"]); mod.fc2.bias.copy_(pp["b2"])


def pool_grads(mod):
    """Gradients of the pooling module under the oracle's parameter names."""
    if isinstance(mod, nn.Parameter):
        return {"weight": mod.grad}
    if type(mod).__name__ == "Attention_Pooling":
        return {"w1": mod.attention[0].weight.grad, "b1": mod.attention[0].bias.grad,
                "w2": mod.attention[2].weight.grad, "b2": mod.attention[2].bias.grad}
    if type(mod).__name__ == "Gated_Attention_Pooling":
        return {"wa": mod.fc1[0].weight.grad, "ba": mod.fc1[0].bias.grad,
                "wg": mod.score[0].weight.grad, "bg": mod.score[0].bias.grad,
                "w2": mod.fc2.weight.grad, "b2": mod.fc2.bias.grad}
    return {}


def gen_vlfan(ref):
    for (name, N, P, K, pooling, head, gated, kind, seed, grads) in cases.VLFAN_CASES:
        X = cases.bag_for_case(N, kind, seed)
        params = cases.make_params(P, K, seed + 1000, gated, aligned_to=X if kind == "adversarial" else None)
        enc = build_vlfan(ref, P, pooling, head, gated, params, seed)
        model = fake_vlsa(ref, enc, params["T"], cases.LOGIT_SCALE, text_requires_grad=True)
        model.eval()
        Xb = X[None]
        logits, img, txt = model(Xb)
        with torch.no_grad():
            v, attn = enc(Xb, ret_with_attn=True)
        out = dict(x_checksum=np.array(cases.checksum(X)), logits=_np(logits), image_features=_np(img),
                   text_features=_np(txt), v=_np(v))
        if isinstance(attn, tuple):
            out["A"] = _np(attn[0][0])
            out["pool_ext"] = _np(attn[1])
        else:
            out["A"] = _np(attn[0])
        out["param_checksum"] = np.array([float(params["W"].double().sum()), float(params["T"].double().sum())])
        if grads:
            g = cases.gen(seed + 2000)
            G = torch.randn(1, K, generator=g)
            out["G"] = _np(G)
            model.zero_grad()
            (logits * G).sum().backward()
            out["grad.logit_scale"] = _np(model.logit_scale.grad)
            out["grad.T"] = _np(model.text_param.grad)
            if head != "Identity":
                cases.pack_big(out, "grad.W", enc.visual_adapter.weight.grad)
                out["grad.b"] = _np(enc.visual_adapter.bias.grad)
            if gated:
                out["grad.Q"] = _np(enc.Q.grad)
            else:
                out["grad.resid"] = _np(enc.Q.residual_features.grad)
            for k, gr in pool_grads(enc.query_pooling).items():
                cases.pack_big(out, "grad.pool." + k, gr)
        np.savez_compressed(os.path.join(HERE, f"vlfan_{name}.npz"), **out)
        print("vlfan", name, "logits", out["logits"].ravel()[:4])


def gen_zeroshot(ref):
    for (name, N, K, pooling, seed) in cases.ZEROSHOT_CASES:
        X = cases.make_bag(N, seed)
        params = cases.make_params(1, K, seed + 1000)
        enc = ref.deepmil.FeatMIL(pooling=pooling)
        model = fake_vlsa(ref, enc, params["T"], cases.LOGIT_SCALE, pooling=pooling).eval()
        with torch.no_grad():
            logits, img, txt = model(X[None])
        out = dict(x_checksum=np.array(cases.checksum(X)), logits=_np(logits), text_features=_np(txt))
        if img.shape[0] == 1:
            out["image_features"] = _np(img)
        else:
            out["image_features_rows"] = _np(img[:8])
        np.savez_compressed(os.path.join(HERE, f"zeroshot_{name}.npz"), **out)
        print("zeroshot", name, out["logits"].ravel()[:4])


def gen_deepmil(ref):
    for (name, N, K, pooling, seed) in cases.DEEPMIL_CASES:
        X = cases.make_bag(N, seed)
        params = cases.make_params(1, K, seed + 1000)
        torch.manual_seed(seed)
        enc = ref.deepmil.DeepMIL(dim_in=cases.D, dim_hid=256, num_cls=cases.D, use_feat_proj=False,
                                  drop_rate=0.25, pooling=pooling, pred_head="Adapter", dim_reduction=4,
                                  keep_ratio=0.8).eval()
        load_pool_params(enc.sigma, cases.make_pool_params(pooling, seed + 3000))
        ad = cases.make_adapter_params(seed + 4000)
        with torch.no_grad():
            enc.visual_adapter.fc[0].weight.copy_(ad["down"])
            enc.visual_adapter.fc[2].weight.copy_(ad["up"])
        model = fake_vlsa(ref, enc, params["T"], cases.LOGIT_SCALE, text_requires_grad=True).eval()
        logits, img, txt = model(X[None])
        out = dict(x_checksum=np.array(cases.checksum(X)), logits=_np(logits),
                   image_features=_np(img), text_features=_np(txt))
        if pooling in ("attention", "gated_attention"):
            with torch.no_grad():
                v, attn = enc(X[None], ret_with_attn=True)
            out["v"] = _np(v)
            out["attn"] = _np(attn)
        g = cases.gen(seed + 2000)
        G = torch.randn(1, K, generator=g)
        out["G"] = _np(G)
        model.zero_grad()
        (logits * G).sum().backward()
        for k, gr in pool_grads(enc.sigma).items():
            cases.pack_big(out, "grad.pool." + k, gr)
        cases.pack_big(out, "grad.adapter.down", enc.visual_adapter.fc[0].weight.grad)
        cases.pack_big(out, "grad.adapter.up", enc.visual_adapter.fc[2].weight.grad)
        out["grad.logit_scale"] = _np(model.logit_scale.grad)
        out["grad.T"] = _np(model.text_param.grad)
        np.savez_compressed(os.path.join(HERE, f"deepmil_{name}.npz"), **out)
        print("deepmil", name, out["logits"].ravel()[:4])


def gen_interpretation(ref):
    """calc_text_img_similarity (utils/model_inference.py:81-144) on one seeded model, both softmax axes."""
    if ref.model_inference is None:
        print("model_inference not importable:", ref.model_inference_error)
        return
    N, P, K, seed = 512, 8, 8, 401
    X = cases.make_bag(N, seed, "clustered")
    params = cases.make_params(P, K, seed + 1000)
    enc = build_vlfan(ref, P, "mean", "default", False, params, seed)
    model = fake_vlsa(ref, enc, params["T"], cases.LOGIT_SCALE).eval()
    out = dict(x_checksum=np.array(cases.checksum(X)))
    for axis in ("V", "L"):
        _, A, cottn, probs, probs2, dec_imp, shap = ref.model_inference.calc_text_img_similarity(
            model, X[None], axis_softmax=axis)
        out[f"{axis}.A"] = _np(A)[:, ::8]      # every 8th patch column
        out[f"{axis}.cottn"] = _np(cottn)[:, ::8]
        out[f"{axis}.probs"] = _np(probs)
        out[f"{axis}.probs2"] = _np(probs2)
        out[f"{axis}.decoupled_imp"] = _np(dec_imp)
        out[f"{axis}.shap"] = _np(shap)
    # SHAP on a seeded [P,K] similarity matrix on its own
    g = cases.gen(402)
    sim = torch.rand(P, K, generator=g) * 2 - 1
    out["shap_in"] = _np(sim)
    out["shap_out"] = _np(ref.model_inference.evaluate_prototype_shap_imp(sim, 56.31))
    np.savez_compressed(os.path.join(HERE, "interpretation.npz"), **out)
    print("interpretation shap", out["V.shap"][:4])


def gen_misc(ref):
    """query_div_loss and the shipped checkpoint's key set / shapes (values are NOT stored)."""
    out = {}
    for gated in (False, True):
        P = 6
        params = cases.make_params(P, 4, 501, gated)
        Q = 0.5 * params["resid"] + params["prompt"]
        enc = ref.deepmil.VLFAN(dim_in=cases.D, use_feat_proj=False, query="Parameter", num_query=P,
                                gated_query=gated)
        with torch.no_grad():
            enc.Q.copy_(Q)
        tag = "gated" if gated else "plain"
        out[f"{tag}.Q"] = _np(Q)
        out[f"{tag}.loss_last_div"] = _np(enc.query_div_loss(last_div=True))
        out[f"{tag}.loss_all"] = _np(enc.query_div_loss(last_div=False))
    np.savez_compressed(os.path.join(HERE, "query_div.npz"), **out)
    ck = os.path.join(_ref_import.REF_ROOT, "assert", "blca-train-VLSA", "train_model-last.pth")
    if os.path.exists(ck):
        sd = torch.load(ck, map_location="cpu", weights_only=False)
        keys = {k: list(v.shape) for k, v in sd["model"].items()}
        with open(os.path.join(HERE, "ckpt_keys.json"), "w") as f:
            json.dump({"model": keys, "top": sorted(sd.keys())}, f, indent=1, sort_keys=True)
        print("ckpt keys", keys)


def gen_train_step(ref):
    """The reference's own training step: VLSA.forward per bag, SurvIFMLE + SurvEMD, Adam with weight decay on the
    >= 2-D parameters only (runner/vlsa_handler.py:241-289, optim/optim_factory.py:25-37)."""
    import loss.loss_surv as LS
    import loss.loss_surv_ext as LE
    cfg = cases.TRAIN
    P, K = cfg["P"], cfg["K"]
    params = cases.make_params(P, K, cfg["seed"] + 1000)
    enc = build_vlfan(ref, P, "mean", "default", False, params, cfg["seed"])
    enc.train()
    model = fake_vlsa(ref, enc, params["T"], cases.LOGIT_SCALE, text_requires_grad=True)
    bags = cases.train_bags()
    t = torch.tensor(cfg["t"]).view(-1, 1)
    e = torch.tensor(cfg["e"]).view(-1, 1).float()
    decay = [p for n, p in model.named_parameters() if p.requires_grad and not (p.dim() <= 1 or n.endswith(".bias"))]
    no_decay = [p for n, p in model.named_parameters() if p.requires_grad and (p.dim() <= 1 or n.endswith(".bias"))]
    opt = torch.optim.Adam([{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": cfg["wd"]}], lr=cfg["lr"])
    ifmle, emd = LS.SurvIFMLE(), LE.SurvEMD(p=2)
    out = {}
    for step in range(cfg["steps"]):
        preds = torch.cat([model(x[None])[0] for x in bags], dim=0)
        inc = torch.softmax(preds, dim=-1)
        l1, l2 = ifmle(inc, t, e), emd(inc, t, e, model.get_logit_scale())
        loss = 1.0 * l1 + 1.0 * l2
        opt.zero_grad()
        loss.backward()
        if step == 0:
            out["grad0.resid"] = _np(enc.Q.residual_features.grad)
            out["grad0.logit_scale"] = _np(model.logit_scale.grad)
            out["logits0"] = _np(preds)
        opt.step()
        out[f"loss{step}"] = np.array([float(loss), float(l1), float(l2)])
        if step in (0, cfg["steps"] - 1):
            out[f"resid@{step}"] = _np(enc.Q.residual_features)
            out[f"b@{step}"] = _np(enc.visual_adapter.bias)
            out[f"logit_scale@{step}"] = _np(model.logit_scale)
            out[f"T@{step}"] = _np(model.text_param)
            cases.pack_big(out, f"W@{step}", enc.visual_adapter.weight)
    np.savez_compressed(os.path.join(HERE, "train_step.npz"), **out)
    print("train", [out[f"loss{i}"] for i in range(cfg["steps"])])


def gen_cindex(ref):
    """The reference's c-index on discrete predictions (eval/cindex.py:6-43, type_pred='incidence')."""
    from eval.cindex import concordance_index
    out = {}
    for (n, K, seed) in cases.CINDEX_CASES:
        y, inc = cases.make_cindex_case(n, K, seed)
        out[f"c{seed}"] = np.array([concordance_index(y.clone(), inc.clone(), type_pred="incidence")], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "cindex.npz"), **out)
    print("cindex", {k: float(v[0]) for k, v in out.items()})


if __name__ == "__main__":
    ref = _ref_import.import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "cindex":
        gen_cindex(ref)
        sys.exit(0)
    gen_vlfan(ref)
    gen_zeroshot(ref)
    gen_deepmil(ref)
    gen_interpretation(ref)
    gen_misc(ref)
    gen_train_step(ref)
    gen_cindex(ref)
    print("done")
