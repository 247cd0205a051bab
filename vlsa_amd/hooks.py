"""Host-side hooks of the model factory: where the tokenizer and the pretrained VL model come from.

The reference builds both inside ``VLSA.__init__`` (model/vlsa.py:38-49: ``Tokenizer(root, name, api)`` from
model/utils_vl.py:19-75 and ``load_vl_model_to_cpu`` from model/utils_vl.py:78-149).  Tokenisation and weight download /
deserialisation are host glue outside the hot path, so this package does not re-implement them: it
calls a *hook*.

    set_tokenizer_factory(fn)    fn(root=..., name=..., api=...) -> tokenizer with the reference wrapper's contract:
                                 ``tok(text | [texts], return_raw_tokens, return_num_tokens)`` and the attributes
                                 ``bos_token_id / eos_token_id / pad_token_id``
    set_vl_model_loader(fn)      fn(text_encoder_cfg=..., root=..., api=...) -> object with ``.text`` (a CoCa text tower: the
                                 attributes CONCHPromptEncoder adopts, model/prompt_encoder.py:213-243) and ``.logit_scale``

Defaults (no hook installed): the host application's own loaders when they are importable -- running inside the reference's
tree, ``model.utils_vl.Tokenizer`` and ``model.conch.create_model_from_pretrained`` (the calls model/utils_vl.py:27-41,116-123
make); otherwise the ``conch`` pip package of mahmoodlab/CONCH.  Neither being importable raises with instructions; nothing
here falls back to a CPU computation of the hot path.
"""
from __future__ import annotations

import os.path as osp
from typing import Callable, Optional

_tokenizer_factory: Optional[Callable] = None
_vl_model_loader: Optional[Callable] = None


def set_tokenizer_factory(fn: Optional[Callable]) -> Optional[Callable]:
    """Install (or, with None, remove) the tokenizer hook; returns the previous one."""
    global _tokenizer_factory
    prev, _tokenizer_factory = _tokenizer_factory, fn
    return prev


def set_vl_model_loader(fn: Optional[Callable]) -> Optional[Callable]:
    """Install (or, with None, remove) the VL-model hook; returns the previous one."""
    global _vl_model_loader
    prev, _vl_model_loader = _vl_model_loader, fn
    return prev


def make_tokenizer(root, name, api):
    if _tokenizer_factory is not None:
        return _tokenizer_factory(root=root, name=name, api=api)
    try:
        from model.utils_vl import Tokenizer            # the host application's wrapper (reference tree on sys.path)
    except Exception as exc:
        raise RuntimeError("no tokenizer: call vlsa_amd.hooks.set_tokenizer_factory(fn) -- fn(root, name, api) must return an "
                           "object with the contract of the reference's model.utils_vl.Tokenizer -- or run inside the "
                           "reference's tree, where that class is importable") from exc
    return Tokenizer(root=root, name=name, api=api)


def load_vl_model(text_encoder_cfg, root, api):
    if _vl_model_loader is not None:
        return _vl_model_loader(text_encoder_cfg=text_encoder_cfg, root=root, api=api)
    if api != "CONCH":
        raise NotImplementedError(f"vlsa_api={api!r}: the HIP text tower implements the CONCH (CoCa) text transformer only "
                                  "(cfg_vlsa_conch.yaml:39); install a loader with vlsa_amd.hooks.set_vl_model_loader for other towers")
    ckpt = osp.join(root, text_encoder_cfg["name"], "pytorch_model.bin")
    create = None
    try:
        from model.conch import create_model_from_pretrained as create          # reference tree
    except Exception:
        try:
            from conch.open_clip_custom import create_model_from_pretrained as create   # mahmoodlab/CONCH package
        except Exception as exc:
            raise RuntimeError("no CONCH model loader: call vlsa_amd.hooks.set_vl_model_loader(fn), or make the reference's "
                               "`model.conch` / the `conch` package importable") from exc
    return create("conch_ViT-B-16", checkpoint_path=ckpt, return_transform=False)
