"""Query network of VLFAN / text-side 'Adapter' prompt learner: frozen text-prototype features + a learnable adaptation
(reference: model/prompt_learners/prompt_adapter.py:11-149).  Same constructor keywords as the reference
(``prompt_encoder, tokenizer, method, load_path, load_idx, load_negative_prompts, load_negative_idx, num_prompts,
init_prompt_path, init_prompt_context_idx, init_prompt_rank_idx, pretrained_prompt_features, dim_reduction, keep_ratio,
res_ratio``), same state-dict keys (``residual_features``, ``neg_residual_features``, ``adapter.fc.*``, ``fc.0.weight``);
``prompt_features`` / ``neg_prompt_features`` are non-persistent buffers there too.

The frozen prototype features are text-tower outputs of fixed sentences (prompt_adapter.py:60-63,74-79).  The tower here is
the HIP one (``vlsa_amd.prompt_encoder.CONCHPromptEncoder``), which runs on the MI355X only, while the reference's handler
builds the model on the CPU and moves it afterwards (``func_load_model(cfg).cuda()``, runner/base_handler.py:114).  So when
the encoder is not on the device yet, the sentences are tokenised at construction and the tower pass is DEFERRED to the first
use (``forward`` / ``get_raw_prompt_features``), by which time ``.cuda()`` has happened; the result is the same tensor the
reference computes in its constructor.
"""
from __future__ import annotations

import json
from typing import Optional, Sequence, Union

import torch
import torch.nn as nn

from .layers import Adapter


def _read_json(path):
    with open(path, "r") as f:
        return json.load(f)


def _rank_sentences(path, context_idx, rank_idx):
    """context template with CLASSNAME replaced by each class' rank name (utils/io.py:151-173 with replace=True)."""
    spec = _read_json(path)
    template = spec["context_templates"][context_idx]
    return [template.replace("CLASSNAME", names[rank_idx]) for names in spec["class_names"].values()]


def _runs_on_host(prompt_encoder) -> bool:
    """True for this package's HIP tower while its weights are still on the CPU (the pass must wait for ``.cuda()``)."""
    tensors = getattr(prompt_encoder, "_tower_tensors", None)
    return tensors is not None and not tensors()[0].is_cuda


class _DeferredTowerPass:
    """A tower pass over tokenised sentences that has to wait for the device (the HIP tower has no CPU route).  A plain
    object instead of a closure: a model with pending passes can be pickled, and ``copy.deepcopy`` of the model binds the
    copy's pass to the copy's encoder (shared through the deepcopy memo)."""

    def __init__(self, prompt_encoder, token_ids, mean_row):
        self.prompt_encoder, self.token_ids, self.mean_row = prompt_encoder, token_ids, mean_row

    def __call__(self):
        return PromptAdapter._tower_pass(self.prompt_encoder, self.token_ids, self.mean_row)


class PromptAdapter(nn.Module):
    def __init__(self, prompt_encoder=None, tokenizer=None, method: str = "default", load_path: Optional[str] = None,
                 load_idx: Union[int, str] = 0, load_negative_prompts: bool = False, load_negative_idx: str = "prompt_normal_tissue",
                 num_prompts: int = 4, init_prompt_path: Optional[str] = None, init_prompt_context_idx: int = 0,
                 init_prompt_rank_idx: int = 0, pretrained_prompt_features=None, dim_reduction: int = 4, keep_ratio: float = 0.8,
                 res_ratio: float = 0.5, init_texts: Optional[Sequence[str]] = None, neg_texts: Optional[Sequence[str]] = None,
                 pretrained_neg_prompt_features=None, **kwargs):
        """Beyond the reference's keywords: ``init_texts`` / ``neg_texts`` (the sentences themselves instead of a JSON path)
        and ``pretrained_neg_prompt_features`` (a ready [1, D] negative prototype)."""
        super().__init__()
        assert method in ["default", "FC", "Adapter", "TaskRes"]
        self.method = method
        self.num_prompts = num_prompts
        self.__dict__["_pending"] = {}          # buffer name -> zero-argument callable producing it (deferred tower pass)
        dim = None
        if pretrained_prompt_features is None:
            if init_texts is None:
                if init_prompt_path is not None:
                    init_texts = _rank_sentences(init_prompt_path, init_prompt_context_idx, init_prompt_rank_idx)
                elif load_path is not None:
                    init_texts = _read_json(load_path)[str(load_idx)]
                else:
                    raise RuntimeError("Please specify `init_prompt_path` or `load_path` to load initial prompts or texts.")
            if prompt_encoder is None or tokenizer is None:
                raise RuntimeError("give `pretrained_prompt_features`, or `prompt_encoder` + `tokenizer` + texts")
            assert len(init_texts) == num_prompts, f"Expected {num_prompts} initial texts, but got {len(init_texts)}."
            prompt_features = self._encode_or_defer("prompt_features", prompt_encoder, tokenizer, list(init_texts), False)
        elif callable(pretrained_prompt_features) and not isinstance(pretrained_prompt_features, torch.Tensor):
            # features that need the device-side tower (a CoOp-pretrained learner's text features, model/vlsa.py:129-137):
            # evaluated at first use, see the module docstring
            self._pending["prompt_features"] = pretrained_prompt_features
            prompt_features = None
        else:
            assert len(pretrained_prompt_features) == num_prompts, \
                f"Expected {num_prompts} initial texts, but got {len(pretrained_prompt_features)}."
            prompt_features = pretrained_prompt_features.detach().clone()
        self.register_buffer("prompt_features", prompt_features, persistent=False)
        if load_negative_prompts:
            if pretrained_neg_prompt_features is not None:
                neg = pretrained_neg_prompt_features.detach().clone().reshape(1, -1)
            else:
                if neg_texts is None:
                    assert load_path is not None, "Found null `load_path`."
                    neg_texts = _read_json(load_path)[str(load_negative_idx)]
                assert prompt_encoder is not None and tokenizer is not None, "negative prompts need texts + encoder + tokenizer"
                neg = self._encode_or_defer("neg_prompt_features", prompt_encoder, tokenizer, list(neg_texts), True)
            self.register_buffer("neg_prompt_features", neg, persistent=False)
        if prompt_features is not None:
            dim, dtype = prompt_features.shape[-1], prompt_features.dtype
        else:
            dim, dtype = int(prompt_encoder.output_dim), torch.float32
        if method == "Adapter":
            self.adapter = Adapter(dim, dim_reduction).to(dtype)
            assert 0 <= keep_ratio <= 1.0
            self.keep_ratio = keep_ratio
        elif method == "TaskRes":
            self.residual_features = nn.Parameter(torch.randn(num_prompts, dim))
            self.neg_residual_features = nn.Parameter(torch.randn(1, dim)) if load_negative_prompts else None
            self.res_ratio = res_ratio
        elif method == "FC":
            self.fc = nn.Sequential(nn.Linear(dim, dim, bias=False), nn.Dropout(0.25))

    # -- frozen prototype features: now, or once the tower is on the device ----------------------------------------------
    @staticmethod
    def _tower_pass(prompt_encoder, token_ids, mean_row):
        tensors = getattr(prompt_encoder, "_tower_tensors", None)
        if tensors is not None:
            dev = tensors()[0].device
            if dev.type != "cuda":
                from ._native import VlsaNativeError
                raise VlsaNativeError("PromptAdapter: the text tower is still on the CPU -- move the model to the MI355X first "
                                      "(`func_load_model(cfg).cuda()`); there is no CPU fallback for the tower")
            token_ids = token_ids.to(dev)
        with torch.no_grad():
            feats = prompt_encoder(prompts_text=token_ids)
            if mean_row:
                feats = feats.mean(0, keepdims=True)
        return feats.detach().clone()

    def _encode_or_defer(self, name, prompt_encoder, tokenizer, texts, mean_row):
        token_ids = tokenizer(texts, return_raw_tokens=False, return_num_tokens=False)      # [n, ctx_length]
        if _runs_on_host(prompt_encoder):
            self._pending[name] = _DeferredTowerPass(prompt_encoder, token_ids, mean_row)
            return None
        return self._tower_pass(prompt_encoder, token_ids, mean_row)

    def _materialise(self):
        """Run the deferred tower passes (the encoder has been moved to the device by now)."""
        for name in list(self._pending):
            feats = self._pending[name]().detach().clone()
            if name == "prompt_features":      # the constructor's check for ready-made features (prompt_adapter.py:66), deferred too
                assert len(feats) == self.num_prompts, f"Expected {self.num_prompts} initial texts, but got {len(feats)}."
            like = next((p for p in self.parameters()), None)
            self._buffers[name] = feats if like is None else feats.to(like.device)
            del self._pending[name]

    def pending_on_device(self) -> bool:
        """True when deferred tower passes exist and could run now (their encoder has reached the GPU)."""
        if not self._pending:
            return False
        return not any(_runs_on_host(getattr(f, "prompt_encoder", None)) for f in self._pending.values()
                       if getattr(f, "prompt_encoder", None) is not None)

    def _feature(self, name):
        if self._pending:
            self._materialise()
        return getattr(self, name)

    def _has_neg(self):
        return "neg_prompt_features" in self._buffers

    def get_raw_prompt_features(self):
        raw = self._feature("prompt_features").clone()
        if self._has_neg():
            raw = torch.cat([raw, self._feature("neg_prompt_features").clone()], dim=0)
        return raw

    def forward(self):
        pf = self._feature("prompt_features")       # (read only below: every method is out of place; only the pass-through hands out a copy)
        has_neg = self._has_neg()
        if self.method == "Adapter":
            return (1 - self.keep_ratio) * self.adapter(pf) + self.keep_ratio * pf
        if self.method == "TaskRes":
            # (one launch each way: pf + ratio * residual -- with the shipped ratio 0.5 the product is exact, so the fused form equals
            # the reference's `res_ratio * residual + features` bit for bit; otherwise within one ulp)
            out = torch.add(pf, self.residual_features, alpha=self.res_ratio)
            if has_neg:
                neg = self.neg_prompt_features.clone()
                if self.neg_residual_features is not None:
                    neg = self.res_ratio * self.neg_residual_features + neg
                out = torch.cat([out, neg], dim=0)
            return out
        if self.method == "FC":
            src = torch.cat([pf, self.neg_prompt_features.clone()], dim=0) if has_neg else pf
            return self.fc(src)
        return pf.clone()
