"""Interpretation utilities over a trained VLSA model -- counterparts of utils/model_inference.py:11-178.

``calc_text_img_similarity`` decouples the prediction over the P text prototypes.  The reference re-encodes all N
patches through the visual adapter (an [N,512]x[512,512] GEMM) and contracts with the attention weights
(utils/model_inference.py:129-132); algebraically  A @ (((X W^T + b)/L) T^^T) = ((out W^T + b * sum_n A)/L) T^^T
and sum_n A_pn = 1 (softmax over the patches), so the decoupled [P, K] similarities follow from the P aggregated rows
``out`` the forward pass already produced: no second pass over the bag is needed.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import functional as VF
from .deepmil import VLFAN


def evaluate_prototype_shap_imp(decoupled_similarity, logit_scale, verbose=False):
    """Exact Shapley values of the P prototypes for the survival risk sum_k (K - k) softmax(ls * mean_p sim)[k];
    the empty coalition is worth 1 (utils/model_inference.py:23-79).  O(P 2^P) on the host, vectorised."""
    if isinstance(decoupled_similarity, torch.Tensor) and decoupled_similarity.is_cuda:
        # on the device: the value of every coalition by one thread each, then one workgroup per prototype (vlsa_prototype_shapley)
        from . import _native as nat
        sim = decoupled_similarity.detach().float().contiguous()
        num_p, num_cls = sim.shape
        lib = nat.load()
        V = torch.empty(2 ** num_p, dtype=torch.float32, device=sim.device)
        shap = torch.empty(num_p, dtype=torch.float32, device=sim.device)
        nat.check(lib.vlsa_prototype_shapley(sim.data_ptr(), num_p, num_cls, float(logit_scale), V.data_ptr(), shap.data_ptr(),
                                             torch.cuda.current_stream(sim.device).cuda_stream), "vlsa_prototype_shapley")
        if verbose:
            print("[SHAP] base", V[0].item(), "full", V[-1].item(), "sum", shap.sum().item())
        return shap.cpu()
    # numpy on purpose: the arrays are tiny ([2^P, P]); torch's CPU ops wake its whole intra-op thread pool for them, which
    # on a many-core host costs milliseconds per call and is erratic (measured 1.7 ... 19 ms for the same input)
    import numpy as np
    sim = np.asarray(torch.as_tensor(decoupled_similarity, dtype=torch.float32).cpu().numpy(), dtype=np.float32)
    num_p, num_cls = sim.shape
    n_cases = 2 ** num_p
    idx = np.arange(n_cases)
    masks = ((idx[:, None] >> np.arange(num_p)[None, :]) & 1).astype(np.float32)          # [2^P, P]
    cnt = masks.sum(axis=1)
    mean_sim = (masks[:, :, None] * sim[None, :, :]).sum(axis=1) / np.maximum(cnt, 1.0)[:, None]
    z = np.float32(logit_scale) * mean_sim
    z = z - z.max(axis=1, keepdims=True)
    e = np.exp(z)
    prob = e / e.sum(axis=1, keepdims=True)
    wts = (num_cls - np.arange(0, num_cls)).astype(np.float32)
    V = (prob * wts).sum(axis=1).astype(np.float32)
    V[0] = 1.0
    fac = [math.factorial(i) for i in range(num_p + 1)]
    Wt = np.array([fac[i] * fac[num_p - i - 1] / fac[num_p] for i in range(num_p)], dtype=np.float32)
    shap_np = np.zeros(num_p, dtype=np.float32)
    cnt_i = cnt.astype(np.int64)
    for i in range(num_p):
        without = idx[((idx >> i) & 1) == 0]
        shap_np[i] = (Wt[cnt_i[without]] * (V[without + (1 << i)] - V[without])).sum()
    shap = torch.from_numpy(shap_np)
    V = torch.from_numpy(V)
    if verbose:
        print("[SHAP] base", V[0].item(), "full", V[-1].item(), "sum", shap.sum().item())
    return shap


@torch.no_grad()
def calc_text_img_similarity(model, X_feats, axis_softmax="V", verbose=False):
    """Same return tuple as the reference: (None, A_softmax[P,N], cottn_score[P,N], probs[1,K], probs_2[1,K],
    decoupled_imp[P,K], shap[P]) -- for a VLSA model whose encoder is a VLFAN with mean query pooling and a Linear head."""
    assert axis_softmax in ["L", "V"]
    model.eval()
    enc = model.mil_encoder
    assert isinstance(enc, VLFAN)
    X = X_feats if X_feats.dim() == 3 else X_feats[None]
    X = X.to(next(model.parameters()).device)
    logit_scale = float(model.get_logit_scale())
    That = F.normalize(model.forward_text_only(), dim=-1)
    # ONE pass over the bag: the P aggregated rows and the attention weights; pooling + head on the P rows (as VLFAN.forward)
    Xp = enc.project(X)
    out, cottn = VF.vlfan_cross_attention(Xp, enc.get_query(), gated=enc.gated_query,
                                          coattn_scale=float(enc.coattn_logit_scale.exp()), want_attn=True)   # [P, D], [P, N]
    image_feature = enc.visual_adapter(enc.forward_query_pooling(out.unsqueeze(0))[0])
    if axis_softmax == "V":
        A = cottn
    else:  # softmax over the prototypes of the same scaled cosine scores: recover them from the patch-softmax weights
        qp = VF.prepare_queries(enc.get_query(), enc.gated_query, float(enc.coattn_logit_scale.exp()))
        _, _, _, scores = VF.vlfan_partial(Xp, qp, want_scores=True)  # log2-domain scaled scores
        A = F.softmax(scores / 1.4426950408889634, dim=0)
    L = image_feature.norm(dim=-1)
    probs = F.softmax(logit_scale * (image_feature / L) @ That.t(), dim=-1)
    # decoupled similarities from the aggregated rows (see module docstring)
    dec = (enc.visual_adapter(out) / L) @ That.t()                # [P, K]
    decoupled_imp = F.softmax(logit_scale * dec, dim=0)
    probs_2 = F.softmax(logit_scale * dec.mean(dim=0, keepdim=True), dim=-1)
    shap = evaluate_prototype_shap_imp(dec, logit_scale, verbose=verbose)
    cottn_h = cottn.cpu()
    A_h = cottn_h.clone() if A is cottn else A.cpu()              # axis 'V': the same matrix -- one device-to-host copy, two tensors
    return None, A_h, cottn_h, probs.cpu(), probs_2.cpu(), decoupled_imp.cpu(), shap


@torch.no_grad()
def calc_abmil_text_img_similarity(model, X_feats, verbose=False, **kws):
    """utils/model_inference.py:146-178 for a VLSA model whose MIL encoder is the ABMIL-style ``DeepMIL``: ->
    (attention weights over the patches [1, N] (softmax of the encoder's RAW scores: ``ret_with_attn`` hands those out,
    model/layers.py:118-122,149-153), predicted incidence [1, K]), both on the host.  Scores, pooling and head run in the HIP
    kernels of ``DeepMIL.forward``."""
    model.eval()
    X = X_feats if X_feats.dim() == 3 else X_feats[None]
    assert X.shape[0] == 1
    X = X.to(next(model.parameters()).device)
    scale = float(model.get_logit_scale())
    That = F.normalize(model.forward_text_only(), dim=-1)
    if verbose:
        print("pred_logit_scale:", scale)
    feature, raw_scores = model.mil_encoder(X, ret_with_attn=True)
    attn = F.softmax(raw_scores, dim=-1)
    unit = feature / feature.norm(dim=-1)
    probs = F.softmax(scale * unit @ That.t(), dim=-1)
    return attn.cpu(), probs.cpu()


def _read_run_cfg(run_path: str) -> dict:
    """config.yaml of a run directory, else its print_config.txt ('key --> value' lines); utils/func.py:219-241"""
    import ast
    import os
    import yaml
    path = os.path.join(run_path, "config.yaml")
    if os.path.exists(path):
        with open(path) as f:
            return yaml.load(f, Loader=yaml.FullLoader)
    path = os.path.join(run_path, "print_config.txt")
    if os.path.exists(path):
        cfg = {}
        for line in open(path):
            if "-->" in line:
                k, v = (t.strip() for t in line.split("-->", 1))
                try:
                    cfg[k] = ast.literal_eval(v)
                except (ValueError, SyntaxError):
                    cfg[k] = v
        return cfg
    raise RuntimeError(f"[Model CFG] Model configuration is not found in {run_path}.")


def load_vlsa_model(run_path, cuda_id=0, return_cfg=False):
    """utils/model_inference.py:11-21: a run directory (config + ``train_model-last.pth``) -> the model on ``cuda:<cuda_id>`` with the
    trained tensors loaded (``strict=False``: keys of the frozen text tower are not in the reference's checkpoints)."""
    import os
    from .model_utils import func_load_model
    cfg = _read_run_cfg(run_path)
    cfg["cuda_id"] = cuda_id
    model = func_load_model(cfg).cuda(cuda_id)
    ckpt = torch.load(os.path.join(run_path, "train_model-last.pth"), map_location=f"cuda:{cuda_id}", weights_only=False)
    model.load_state_dict(ckpt["model"], strict=False)
    return (model, cfg) if return_cfg else model
