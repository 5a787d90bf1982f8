import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "da-sac_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn as nn
import models
from oracle import nets_ref as N
from conftest import rel_err
CRIT = nn.CrossEntropyLoss(ignore_index=255, reduction="none")
sd = N.resnet101_state(seed=5, randomize_bn=True, he_init=True, residual_gain=0.25, aspp_gain=0.2)
g = torch.Generator().manual_seed(5)
x = torch.randn(2, 3, 41, 57, generator=g)
y = torch.randint(0, 19, (2, 41, 57), generator=g); y[:, :3] = 255
def oracle(dt):
    ref = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    for k in N.trainable_keys(ref): ref[k].requires_grad_(True)
    losses, _ = N.segnet_forward("deeplabv2_resnet101", ref, x.to(dt), y)
    losses["loss_ce"].sum().backward()
    return ref
r32, r64 = oracle(torch.float32), oracle(torch.float64)
net = models.DeepLabV2_ResNet101(num_classes=19, criterion=CRIT, freeze_bn=True)
net.load_state_dict(sd, strict=True); net.cuda().train()
l2, _ = net(x.cuda(), y.cuda()); l2["loss_ce"].mean().backward()
rows = []
for k, p in net.named_parameters():
    rows.append((rel_err(p.grad, r64[k].grad), rel_err(r32[k].grad, r64[k].grad), rel_err(p.grad, r32[k].grad), k))
rows.sort(reverse=True)
print("hip-vs-f64   cpu32-vs-f64   hip-vs-cpu32   param")
for r in rows[:10]: print("%.2e     %.2e       %.2e    %s" % r)
print("max cpu32-vs-f64:", max(r[1] for r in rows), " max hip-vs-f64:", max(r[0] for r in rows))
print("---- in network order (from the output backwards), hip-vs-f64")
order = [k for k, _ in net.named_parameters()]
errs = {r[3]: r[0] for r in rows}
for k in reversed(order):
    if "layer4" in k or "layer5" in k or "layer3.22" in k or "layer3.21" in k:
        print("%.2e  %s" % (errs[k], k))
