"""Import shim for the upstream reference (liupei101/VLSA at /root/reference).

TEST INFRASTRUCTURE, container-only: the reference's Python never travels to the GPU box, and nothing
in the product package imports this file.  It is used solely by ``make_golden.py`` to generate the
committed golden vectors and by ``tests/test_oracle_vs_reference.py`` (skipped when /root/reference
is absent).  Recipe follows SURVEY.md Appendix A.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

REF_ROOT = os.environ.get("VLSA_REFERENCE_ROOT", "/root/reference")
_STUB_ROOTS = ("nystrom_attention", "torch_geometric", "h5py", "ftfy", "torchvision", "timm", "wandb")


class _Anything(type):
    def __getattr__(cls, name):
        return cls


class _Dummy(metaclass=_Anything):
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, name):
        return _Dummy


class _StubModule(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Dummy


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def __init__(self, roots):
        self.roots = tuple(roots)

    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in self.roots:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        pass


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "model"))


_done = False


def import_reference():
    """Make ``model``, ``utils`` ... of the reference importable; returns a namespace of handy symbols."""
    global _done
    if not reference_available():
        raise RuntimeError(f"reference not found at {REF_ROOT}")
    if not _done:
        # real heavy deps first, so transformers never sees a fake torchvision
        from transformers import CLIPModel, AutoTokenizer  # noqa: F401
        from transformers import PreTrainedTokenizerFast
        try:
            from transformers.modeling_attn_mask_utils import (  # noqa: F401
                _create_4d_causal_attention_mask, _prepare_4d_attention_mask)
        except Exception:
            pass
        missing = []
        for root in _STUB_ROOTS:
            try:
                __import__(root)
            except Exception:
                missing.append(root)
                for k in [k for k in sys.modules if k.split(".")[0] == root]:
                    del sys.modules[k]
        sys.meta_path.append(_StubFinder(missing))
        if not hasattr(PreTrainedTokenizerFast, "batch_encode_plus"):
            PreTrainedTokenizerFast.batch_encode_plus = lambda self, texts, **kw: self(texts, **kw)
        sys.path.insert(0, REF_ROOT)
        _done = True
    import model  # noqa: F401
    from model import deepmil, layers, vlsa
    from model.prompt_learners import prompt_adapter
    ns = types.SimpleNamespace(deepmil=deepmil, layers=layers, vlsa=vlsa, prompt_adapter=prompt_adapter)
    try:
        from utils import model_inference
        ns.model_inference = model_inference
    except Exception as exc:  # pragma: no cover
        ns.model_inference = None
        ns.model_inference_error = exc
    return ns
