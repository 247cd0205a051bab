"""CPU side of the safety net: the environment switches are parsed once at import, the look-ahead / deferral state
follows re-assigned encoder parameters (structure epoch) and plain scalar attributes, the structure hooks are installed by the first
model and not at import, and the PARANOID comparison raises on a mismatch (stubbed per-bag route: no device here)."""
import subprocess
import sys

import pytest
import torch


def test_env_switches_are_read_at_import():
    code = ("import vlsa_amd.vlsa as V; print(int(V.ENV_NO_DEFER), int(V.ENV_NO_LOOKAHEAD), int(V.ENV_PARANOID), "
            "int(getattr(V._install_structure_hooks, 'done', False)))")
    import os
    env = dict(os.environ, VLSA_AMD_NO_DEFER="1", VLSA_AMD_NO_LOOKAHEAD="0", VLSA_AMD_PARANOID="yes")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(__file__)))
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ["1", "0", "1", "0"]      # ... and importing the module installed no process-global hook


def _net(P=4, K=3):
    from vlsa_amd.vlsa import VLSA
    cfg = dict(name="VLFAN", dim_in=512, use_feat_proj=False, query="Parameter", num_query=P, query_pooling="mean")
    return VLSA.from_modules(cfg, pretrained_text_features=torch.randn(K, 512, generator=torch.Generator().manual_seed(1)))


def test_first_model_installs_the_structure_hooks_and_state_follows_reassignment():
    from vlsa_amd import vlsa as V
    net = _net()
    assert V._HOOKS_INSTALLED[0]
    T = net._text_features()
    s0 = net._eval_state(T)
    assert net._same_state(s0, net._eval_state(T))
    enc = net.mil_encoder
    enc.Q = torch.nn.Parameter(enc.Q.detach().clone())                       # a new object with the same version (0) as the old one
    s1 = net._eval_state(T)
    assert not net._same_state(s0, s1)                                        # the kept lists were re-walked (structure epoch)
    enc.query_pooling = "max"                                                 # a plain attribute
    s2 = net._eval_state(T)
    assert not net._same_state(s1, s2) and net._same_state(s2, net._eval_state(T))
    k0 = net._defer_key()
    enc.query_pooling = "mean"
    assert net._defer_key() != k0


def test_paranoid_check_raises_on_a_mismatch(monkeypatch):
    from vlsa_amd import vlsa as V
    net = _net()
    ref = torch.tensor([[1.0, 2.0, 3.0]])
    monkeypatch.setattr(net, "forward", lambda X: (ref.clone(), None, None))
    X = torch.zeros(1, 5, 512)
    assert net._paranoid_check(X, ref[0] + 5e-5, "stub") < V.PARANOID_TOLERANCE
    with pytest.raises(V.ParanoidMismatch, match="per-bag route"):
        net._paranoid_check(X, ref[0] + 1e-3, "stub")
    assert net._paranoid_checks == 2 and net._materialising is False


def test_identity_featmil_is_never_deferred():
    from vlsa_amd.vlsa import VLSA
    net = VLSA.from_modules(dict(name="FeatMIL", dim_in=512, pooling="logit_top10"),
                            pretrained_text_features=torch.randn(3, 512, generator=torch.Generator().manual_seed(2)))
    net.train()
    net.defer_training_calls = True

    class FakeCuda(torch.Tensor):          # `takes` wants a device tensor: stand in for one on the CPU
        is_cuda = True
    X = torch.zeros(1, 7, 512).as_subclass(FakeCuda)
    assert net._defer_call(X, torch.zeros(3, 512)) is None
