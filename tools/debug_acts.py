import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "da-sac_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn as nn, torch.nn.functional as F
import models
from dasac_hip import engine as E, ops
from oracle import nets_ref as N, head_ref as H
from conftest import rel_err
CRIT = nn.CrossEntropyLoss(ignore_index=255, reduction="none")
sd = N.resnet101_state(seed=5, randomize_bn=True, he_init=True, residual_gain=0.25, aspp_gain=0.2)
g = torch.Generator().manual_seed(5)
x = torch.randn(2, 3, 41, 57, generator=g)
y = torch.randint(0, 19, (2, 41, 57), generator=g); y[:, :3] = 255
dt = torch.float64
ref = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
outs = []
a = F.conv2d(x.to(dt), ref["model.conv1.weight"], stride=2, padding=3)
a = F.relu(N.batchnorm(ref, "model.bn1", a, False)); a = F.max_pool2d(a, 3, 2, 1, ceil_mode=True)
for li, (planes, blocks, stride, dil) in enumerate(N.RESNET101_STAGES, start=1):
    for bi in range(blocks):
        a = N.bottleneck(ref, "model.layer{}.{}".format(li, bi), a, stride if bi == 0 else 1, dil, False)
        a.requires_grad_(True) if not a.requires_grad else None
        a.retain_grad(); outs.append(("layer{}.{}".format(li, bi), a))
logits = N.aspp_sum(ref, "model.layer5.conv2d_list", a)
up = H.upsample_bilinear_ac(logits, 41, 57)
loss = H.ce_mean_all_pixels(up, y); loss.sum().backward()
net = models.DeepLabV2_ResNet101(num_classes=19, criterion=CRIT, freeze_bn=True)
net.load_state_dict(sd, strict=True); net.cuda().train()
eng = E.Engine(net._plan())
out, saved = eng.forward(x.cuda(), keep=True)
acts = dict(saved["acts"])
# block output slots: ops with res != None
blk_slots = [op.dst for op in eng.plan.ops if op.kind == "conv" and op.res is not None]
dl = ops.upsample_bwd(ops.ce_loss(ops.upsample_softmax(out, (41, 57))[0], y.cuda(), want_grad=True)[1], out.shape[2:])
trace = {}
eng.backward(saved, dl, [True] * len(eng.params), trace=trace)
print("block          fwd err     grad err   (vs f64 oracle)")
for (name, t), slot in list(zip(outs, blk_slots))[-8:]:
    gr = t.grad * (t.detach() > 0)         # engine stores the masked gradient (dz) of a ReLU output
    print("%-12s  %.2e   %.2e" % (name, rel_err(acts[slot], t), rel_err(trace[slot], gr)))
    d = (trace[slot].cpu().double() - gr).abs()
    bad = (d > 1e-4 * gr.abs().max())
    if bad.any():
        idx = bad.nonzero()
        print("     bad elements:", int(bad.sum()), "of", bad.numel(), " e.g.", idx[:5].tolist(), " act there:", [float(t[tuple(i)]) for i in idx[:5]])
