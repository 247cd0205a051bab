"""One training epoch + one evaluation pass of the reference's handler loops, end to end on one MI355X (BASELINE configs[4]'s loop shape:
cfg_vlsa_conch.yaml -- 32 bags per optimizer step, VLFAN + TaskRes text queries, rank prompt learner through the CONCH-size tower,
IF-MLE + EMD, Adam), over synthetic TCGA-like patients (2k-12k patches each) held in `vlsa_amd.ingest.ResidentBags`:

  * training: `_train_each_epoch` / `_update_network` (runner/vlsa_handler.py:192-289) as written -- shuffled DataLoader(batch_size=1),
    `data_x[0].cuda()`, one `net(x)` per bag, `torch.cat`, loss, backward, step -- with and without the deferred training calls
    (`patch_reference()` sets them);
  * evaluation: `test_model` (315-345) as written -- `net(X)` per bag under no_grad, softmax, two `.cpu()` per bag -- with and without the
    look-ahead windows.
Epoch 1 (disk -> HBM in the real loop, here host -> HBM) is reported separately.  `python tools/bench_epoch.py [patients] [fp32|bf16]`"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import text_cases as TC
from vlsa_amd.ingest import ResidentBags
from vlsa_amd.losses import SurvObjective
from vlsa_amd.prompt_adapter import PromptAdapter
from vlsa_amd.prompt_encoder import CONCHPromptEncoder
from vlsa_amd.prompt_learner import RankPromptLearner
from vlsa_amd.vlsa import VLSA
import gc; gc.collect(); gc.freeze()

dev = "cuda"
NP = int(sys.argv[1]) if len(sys.argv) > 1 else 384
DT = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float32       # fp32 = what patch_reference() stores
K = P = 12
BATCH = 32
EMPTY_CACHE = False        # (set for the timed epochs: see train_epoch)
c = TC.TOWERS["conch"]
enc = CONCHPromptEncoder(width=c["width"], heads=c["heads"], layers=c["layers"], vocab_size=c["vocab"], output_dim=c["out_dim"])
for p_ in enc.parameters():
    p_.requires_grad_(False)
enc = enc.to(dev)
table, ctx_key, names = TC.synthetic_prompt_table(c["vocab"], 1)
pl = RankPromptLearner(dict(max_num_tokens=127, embedding_dim=768, embedding_dtype=torch.float32), TC.ReplayTokenizer(table),
                       enc.token_embedding, num_base_ranks=4, num_ranks=K, num_tokens_per_rank=4, num_context_tokens=8,
                       init_context=ctx_key, init_rank_names=names)
qnet = PromptAdapter(method="TaskRes", num_prompts=P, pretrained_prompt_features=torch.randn(P, 512), res_ratio=0.5)
cfg = dict(name="VLFAN", dim_in=512, use_feat_proj=False, num_query=P, query="Text", query_pooling="mean", pred_head="default")
net = VLSA.from_modules(cfg, prompt_learner=pl, prompt_encoder=enc, query_network=qnet).to(dev)
opt = torch.optim.Adam([p_ for p_ in net.parameters() if p_.requires_grad], lr=2e-4)
objective = SurvObjective()


class Patients(torch.utils.data.Dataset):
    """items shaped like WSIPatchSurv's 'patch' mode (dataset/PatchWSI.py:197-215): (index, (feats fp32 [N, 512], coords), (t, e))"""

    def __init__(self, n):
        g = torch.Generator().manual_seed(11)
        self.sizes = [int(x) for x in torch.randint(2000, 12000, (n,), generator=g)]
        self.pool = torch.randn(max(self.sizes) + 64 * n, 512, generator=g)
        self.y = torch.stack([torch.randint(0, K, (n,), generator=g).float(), (torch.rand(n, generator=g) < 0.45).float()], dim=1)

    def __len__(self):
        return len(self.sizes)

    def __getitem__(self, i):
        return torch.tensor([i], dtype=torch.int), (self.pool[64 * i:64 * i + self.sizes[i]], torch.zeros(1)), self.y[i]


ds = Patients(NP)
rb = ResidentBags(ds, dtype=DT)
train_loader = torch.utils.data.DataLoader(rb, batch_size=1, shuffle=True, generator=torch.Generator().manual_seed(3), num_workers=0)
eval_loader = torch.utils.data.DataLoader(rb, batch_size=1, shuffle=False, num_workers=0)
patches = sum(ds.sizes)


def update_network(xs, ys):                      # runner/vlsa_handler.py:260-289
    y_hat = []
    for i in range(len(xs)):
        pred, *_ = net(xs[i])
        y_hat.append(pred)
    opt.zero_grad()
    bag_preds = torch.cat(y_hat, dim=0)
    bag_label = torch.cat(ys, dim=0)
    loss = objective(bag_preds, bag_label[:, 0].long(), bag_label[:, 1], net.get_logit_scale())
    loss.backward()
    opt.step()
    return loss.item(), bag_preds.detach().cpu()


def train_epoch():                               # runner/vlsa_handler.py:192-236
    net.train()
    xs, ys, preds = [], [], []
    n = len(train_loader)
    for i_batch, (data_idx, data_x, data_y) in enumerate(train_loader, 1):
        xs.append(data_x[0].cuda())
        ys.append(data_y.cuda())
        if i_batch % BATCH == 0 or i_batch == n:
            _, p_ = update_network(xs, ys)
            preds.append(p_)
            xs, ys = [], []
            if EMPTY_CACHE:
                torch.cuda.empty_cache()             # runner/vlsa_handler.py:222: the reference does this after every mini-batch
    return torch.cat(preds)


def eval_pass():                                 # runner/vlsa_handler.py:315-345
    net.eval()
    raw, pred = [], []
    for data_idx, data_x, data_y in eval_loader:
        X = data_x[0].cuda()
        with torch.no_grad():
            r, *_ = net(X)
            p_ = torch.softmax(r, dim=-1)
        raw.append(r.detach().cpu()); pred.append(p_.detach().cpu())
    return torch.cat(raw), torch.cat(pred)


def timed(fn, reps=3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


torch.cuda.synchronize(); t0 = time.perf_counter()
train_epoch()
torch.cuda.synchronize(); first = time.perf_counter() - t0
EMPTY_CACHE = True
train_epoch()
steps = (NP + BATCH - 1) // BATCH
print(f"{NP} patients, {patches} patches ({patches * 512 * (2 if DT == torch.bfloat16 else 4) / 1e9:.2f} GB resident as {str(DT)[6:]}), {steps} optimizer steps of {BATCH} bags per epoch")
print(f"epoch 1 (uploads every bag: host -> HBM): {first * 1e3:.0f} ms")
res = {}
for defer, ec in ((False, True), (True, True), (True, False)):
    net.defer_training_calls = defer
    EMPTY_CACHE = ec
    train_epoch()
    res[defer] = timed(train_epoch)
    print(f"training epoch, handler loops as written, defer_training_calls = {defer}, torch.cuda.empty_cache() per mini-batch {'as the reference does' if ec else 'REMOVED'}: "
          f"{res[defer] * 1e3:7.1f} ms = {res[defer] / steps * 1e3:5.2f} ms per step ({NP / res[defer]:7.0f} patients/s, {patches / res[defer] / 1e9:.2f} G patches/s trained)")
for la in (0, 64):
    net.lookahead_bags = la
    eval_pass()
    t = timed(eval_pass)
    print(f"evaluation pass, test_model as written, look-ahead {'on ' if la else 'off'}: {t * 1e3:7.1f} ms = {t / NP * 1e6:6.1f} us per patient "
          f"({patches / t / 1e9:.2f} G patches/s)")
