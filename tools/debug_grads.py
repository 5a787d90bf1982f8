import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "da-sac_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn as nn
import models
from oracle import nets_ref as N
from conftest import rel_err
B, S = int(sys.argv[1]), int(sys.argv[2])
CRIT = nn.CrossEntropyLoss(ignore_index=255, reduction="none")
sd = N.resnet101_state(seed=5, randomize_bn=True, he_init=True, residual_gain=0.25, aspp_gain=0.04)
g = torch.Generator().manual_seed(5)
x = torch.randn(B, 3, S, S, generator=g)
y = torch.randint(0, 19, (B, S, S), generator=g); y[:, :3] = 255
ref = {k: v.clone() for k, v in sd.items()}
for k in N.trainable_keys(ref): ref[k].requires_grad_(True)
t = time.time()
losses, _ = N.segnet_forward("deeplabv2_resnet101", ref, x, y)
losses["loss_ce"].sum().backward()
print("oracle", time.time() - t, float(losses["loss_ce"]))
net = models.DeepLabV2_ResNet101(num_classes=19, criterion=CRIT, freeze_bn=True)
net.load_state_dict(sd, strict=True); net.cuda().train()
l2, _ = net(x.cuda(), y.cuda()); l2["loss_ce"].mean().backward()
print("hip loss", float(l2["loss_ce"]))
errs = sorted(((rel_err(p.grad, ref[k].grad), k, float(p.grad.norm()), float(ref[k].grad.norm())) for k, p in net.named_parameters()), reverse=True)
for e in errs[:12]: print(e)
