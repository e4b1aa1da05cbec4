import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from argparse import Namespace
from oracle import objectness_oracle as orc
from unmore_amd.hashrng import hash_init, uniform01
from unmore_amd.objectness_net import ObjectnessNet
from unmore_amd.loss import objectness_loss
sys.path.insert(0, 'tests')
from test_model_gpu import _labels, _net
B,H,W=2,64,64
net, sd = _net("dpt_tiny","tiny")
net.train()
x = torch.from_numpy(uniform01(f"img:tiny{H}x{W}", (B,3,H,W)))
gc,gs,sal=_labels(B,H,W,0)
res={}
for name,dt in (("f32",torch.float32),("f64",torch.float64)):
    sdo={k:v.clone().to(dt).requires_grad_(True) for k,v in sd.items()}
    out=orc.forward(sdo,x.to(dt),orc.CONFIGS["dpt_tiny"])
    l,_=orc.loss_terms(out,gc.to(dt),gs.to(dt),sal.to(dt)); l.backward()
    res[name]={k:v.grad for k,v in sdo.items()}
out=net(images=x.cuda()); loss=objectness_loss(out,gc.cuda(),gs.cuda(),sal.cuda()); loss.backward()
rows=[]
for n,p in net.named_parameters():
    if p.grad is None: continue
    r64=res["f64"][n]; sc=r64.abs().max().item()+1e-12
    e_m=(p.grad.cpu().double()-r64).abs().max().item()/sc
    e_o=(res["f32"][n].double()-r64).abs().max().item()/sc
    rows.append((e_m,e_o,n))
rows.sort(reverse=True)
for r in rows[:15]: print("mine %.2e  oracle32 %.2e  %s"%r)
