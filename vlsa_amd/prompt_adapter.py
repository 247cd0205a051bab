"""Query network of VLFAN: text-prototype features + a learnable adaptation (reference:
model/prompt_learners/prompt_adapter.py:11-149).  The frozen prototype features come from the VL text tower,
which is outside this package's scope: pass them in as ``pretrained_prompt_features`` (what
the reference itself supports, prompt_adapter.py:65-68) or give a ``prompt_encoder`` + ``tokenizer`` + texts.
State-dict keys match the reference (``residual_features``, ``neg_residual_features``, ``adapter.fc.*``, ``fc.0.weight``);
``prompt_features`` / ``neg_prompt_features`` are non-persistent buffers there too.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.nn as nn

from .layers import Adapter


class PromptAdapter(nn.Module):
    def __init__(self, prompt_encoder=None, tokenizer=None, method: str = "default", init_texts: Optional[Sequence[str]] = None,
                 neg_texts: Optional[Sequence[str]] = None, num_prompts: int = 4, pretrained_prompt_features=None,
                 pretrained_neg_prompt_features=None, load_negative_prompts: bool = False, dim_reduction: int = 4,
                 keep_ratio: float = 0.8, res_ratio: float = 0.5, **kwargs):
        super().__init__()
        assert method in ["default", "FC", "Adapter", "TaskRes"]
        self.method = method
        if pretrained_prompt_features is None:
            if prompt_encoder is None or tokenizer is None or init_texts is None:
                raise RuntimeError("give `pretrained_prompt_features`, or `prompt_encoder` + `tokenizer` + `init_texts`")
            assert len(init_texts) == num_prompts, f"Expected {num_prompts} initial texts, but got {len(init_texts)}."
            with torch.no_grad():
                prompt_features = prompt_encoder(prompts_text=tokenizer(list(init_texts), return_raw_tokens=False,
                                                                        return_num_tokens=False))
        else:
            assert len(pretrained_prompt_features) == num_prompts, \
                f"Expected {num_prompts} initial texts, but got {len(pretrained_prompt_features)}."
            prompt_features = pretrained_prompt_features
        self.register_buffer("prompt_features", prompt_features.detach().clone(), persistent=False)
        if load_negative_prompts:
            if pretrained_neg_prompt_features is not None:
                neg = pretrained_neg_prompt_features.detach().clone().reshape(1, -1)
            else:
                assert neg_texts is not None and prompt_encoder is not None, "negative prompts need texts + encoder"
                with torch.no_grad():
                    neg = prompt_encoder(prompts_text=tokenizer(list(neg_texts), return_raw_tokens=False,
                                                                return_num_tokens=False)).mean(0, keepdims=True)
            self.register_buffer("neg_prompt_features", neg, persistent=False)
        dim = prompt_features.shape[-1]
        if method == "Adapter":
            self.adapter = Adapter(dim, dim_reduction).to(prompt_features.dtype)
            assert 0 <= keep_ratio <= 1.0
            self.keep_ratio = keep_ratio
        elif method == "TaskRes":
            self.residual_features = nn.Parameter(torch.randn(num_prompts, dim))
            self.neg_residual_features = nn.Parameter(torch.randn(1, dim)) if load_negative_prompts else None
            self.res_ratio = res_ratio
        elif method == "FC":
            self.fc = nn.Sequential(nn.Linear(dim, dim, bias=False), nn.Dropout(0.25))

    def get_raw_prompt_features(self):
        raw = self.prompt_features.clone()
        if hasattr(self, "neg_prompt_features"):
            raw = torch.cat([raw, self.neg_prompt_features.clone()], dim=0)
        return raw

    def forward(self):
        pf = self.prompt_features.clone()
        has_neg = hasattr(self, "neg_prompt_features")
        if self.method == "Adapter":
            return (1 - self.keep_ratio) * self.adapter(pf) + self.keep_ratio * pf
        if self.method == "TaskRes":
            out = self.res_ratio * self.residual_features + pf
            if has_neg:
                neg = self.neg_prompt_features.clone()
                if self.neg_residual_features is not None:
                    neg = self.res_ratio * self.neg_residual_features + neg
                out = torch.cat([out, neg], dim=0)
            return out
        if self.method == "FC":
            src = torch.cat([pf, self.neg_prompt_features.clone()], dim=0) if has_neg else pf
            return self.fc(src)
        return pf
