import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, torch
from helpers import *
from multipitch_architectures_amd.losses import BCELoss, PolyphonyLoss
from multipitch_architectures_amd.nn_models.layers import Dropout
name,B,T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
g=load_golden(name,B,T)
dev=torch.device('cuda:0')
model=build_model(name,dev)
for m in model.modules():
    if isinstance(m,Dropout): m.p=0.0
model.train()
x,y=synth_batch(B,T); x,y=x.to(dev),y.to(dev)
res=model(x)
loss=PolyphonyLoss()(res[0],res[1],y) if isinstance(res,tuple) else BCELoss()(res,y)
loss.backward()
print('loss',float(loss), g['train.losses'][0], float(g['train.loss64']))
rows=[]
for k,p in model.named_parameters():
    mine=p.grad.detach().cpu().numpy().ravel()[sample_idx(p.numel(),16)].astype(np.float64)
    r64=g[f'grad64.{k}.samples']; r32=g[f'grad.{k}.samples'].astype(np.float64); sc=float(g[f'grad64.{k}.absmax'])
    rows.append((np.abs(mine-r64).max()/max(sc,1e-30), np.abs(r32-r64).max()/max(sc,1e-30), sc, k))
print('in registration order (output side last):')
for r in rows: print('%.2e ref %.2e scale %.2e %s'%r)
