"""TEST INFRASTRUCTURE: the call pattern of the reference's VLSA handler, restated so that the drop-in model can be driven
exactly the way ``runner/vlsa_handler.py`` drives it on a box that has no reference checkout (the GPU box).

    build_model(cfg)            runner/vlsa_handler.py:88-151   cfg -> arch_cfg -> load_model('VLSA', **arch_cfg), freezing
    make_optimizer(model, cfg)  runner/base_handler.py:181-186 + optim/optim_factory.py:25-37,78-79 (Adam; weight decay on
                                every parameter that is not 1-D and whose name does not end in '.bias' -- the 0-dim logit_scale included)
    update_network(...)         runner/vlsa_handler.py:260-289  per-bag forward, cat, objective, backward, step
    test_model(...)             runner/vlsa_handler.py:315-345  eval mode, per-bag forward under no_grad, softmax

tests/test_construction_cpu.py additionally runs the reference's REAL ``func_load_model`` against the patched factory when
/root/reference is present."""
from __future__ import annotations

import torch


def strip_prefix(cfg: dict, prefix: str) -> dict:
    """keys 'prefix_x' -> 'x' (utils/func.py:136-147; one-character remainders are skipped there as well)"""
    out = {}
    for k, v in cfg.items():
        if k.startswith(prefix):
            rest = k.split(prefix)[1]
            if len(rest) >= 2:
                out[rest[1:]] = v
    return out


def arch_cfg_of(cfg: dict) -> dict:
    arch = cfg["arch"].lower()
    name = cfg["vlsa_pmt_learner_name"]
    pl = strip_prefix(cfg, f"{arch}_pmt_learner_{name.lower()}")
    pretrained = cfg.get("vlsa_pmt_learner_pretrained", False)
    pl.update(name=name, pretrained=pretrained)
    pre = None
    if pretrained:
        pre = strip_prefix(cfg, "vlsa_pmt_learner_coop")
        assert pre.get("ckpt") is not None, "Found null ckpt path."
        pre["ckpt"] = pre["ckpt"].format(cfg["data_split_seed"], pre["method"])
    return dict(vlsa_api=cfg[f"{arch}_api"], text_encoder_cfg=strip_prefix(cfg, arch + "_txt_encoder"),
                image_encoder_cfg=strip_prefix(cfg, arch + "_img_encoder"), prompt_learner_cfg=pl,
                pretrained_prompt_learner_cfg=pre, path_clip_model=cfg["path_clip_model"])


def _freeze(obj):
    params = obj.parameters() if isinstance(obj, torch.nn.Module) else [obj]
    for p in params:
        p.requires_grad = False


def build_model(cfg: dict, load_model):
    a = arch_cfg_of(cfg)
    model = load_model(cfg["arch"], **a)
    name = cfg["vlsa_pmt_learner_name"]
    plan = []
    if name == "CoOp":
        plan += [(model.prompt_learner.context_embeds, a["prompt_learner_cfg"]["frozen_context_embeds"]),
                 (model.prompt_learner.rank_embeds, a["prompt_learner_cfg"]["frozen_rank_embeds"])]
    if name in ("CoOp", "Adapter"):
        plan += [(model.mil_encoder, a["image_encoder_cfg"]["frozen"]), (model.prompt_encoder, a["text_encoder_cfg"]["frozen"]),
                 (model.logit_scale, cfg["vlsa_frozen_logit_scale"])]
    for obj, frozen in plan:
        if frozen:
            _freeze(obj)
    return model


def make_optimizer(model, cfg):
    decay, plain = [], []
    for n, p in model.named_parameters():
        if p.requires_grad:
            (plain if (p.dim() == 1 or n.endswith(".bias")) else decay).append(p)      # 0-dim logit_scale IS decayed there
    return torch.optim.Adam([{"params": plain, "weight_decay": 0.0}, {"params": decay, "weight_decay": cfg["opt_weight_decay"]}],
                            lr=cfg["opt_lr"])


def update_network(net, optimizer, objective, xs, ys):
    preds = torch.cat([net(x)[0] for x in xs], dim=0)
    optimizer.zero_grad()
    label = torch.cat(ys, dim=0)
    loss = objective(preds, label[:, 0], label[:, 1], net.get_logit_scale())
    loss.backward()
    optimizer.step()
    return loss.item(), preds.detach().cpu()


def test_model(model, loader, state_dict=None):
    if state_dict is not None:
        model.load_state_dict(state_dict, strict=False)
    model.eval()
    raw, ys = [], []
    for _idx, data_x, y in loader:
        X = data_x[0].cuda()
        with torch.no_grad():
            r, *_ = model(X)
        raw.append(r.detach().cpu())
        ys.append(y)
    raw = torch.cat(raw, dim=0)
    return dict(raw_y_hat=raw, y_hat=torch.softmax(raw, dim=-1), y=torch.cat(ys, dim=0))


test_model.__test__ = False      # not a pytest test
