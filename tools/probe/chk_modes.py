import sys, os
sys.path.insert(0, os.getcwd())
import torch
from argparse import Namespace
from unmore_amd import ops, synth
from unmore_amd.objectness_net import ObjectnessNet
dev = torch.device("cuda:0")
net = ObjectnessNet(dev, 128, "dpt_base", Namespace(use_bg_sdf=True, sdf_activation="tanh")).to(dev)
spec = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.peak_edited_state_dict(spec, "base").items()}, strict=True)
net = net.to(dev); net.eval()
for p in net.parameters(): p.requires_grad = False
net.set_compute_dtype(torch.float32)
x = torch.rand(20, 3, 128, 128, device=dev)
def run():
    with torch.no_grad():
        o = net.get_prediction(x)
    return {k: v.clone() for k, v in o.items()}
a = run()
ops.set_f32_mode("x3_fast"); b = run(); ops.set_f32_mode("x3")
c = run()
net.set_sdf_head_mode("factored"); d = run(); net.set_sdf_head_mode("auto")
for name, (p, q) in {"6term vs 3term": (a, b), "6term again": (a, c), "collapsed vs factored": (c, d)}.items():
    print(name, {k: float((p[k] - q[k]).abs().max()) for k in p})
print("graph mode", getattr(net, "graph_mode", None), "f32 mode", ops.get_f32_mode())
