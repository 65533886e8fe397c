import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch
from adapter4rec_amd import _lib as L
dev='cuda:0'
sizes, groups, lrs = [1000003, 5, 64, 2 ** 20 + 1, 18], [0, 2, 1, 3, 1], [5e-5, 1e-4, 1.5e-4, 2e-4]
n=sum(sizes)
seg_end = torch.tensor([sum(sizes[:i + 1]) for i in range(len(sizes))], dtype=torch.int32, device=dev)
seg_group = torch.tensor(groups, dtype=torch.int32, device=dev)
glr = torch.tensor(lrs, device=dev)
g0=torch.Generator().manual_seed(1)
p0=torch.randn(n,generator=g0).to(dev); gr=torch.randn(n,generator=g0).to(dev)
out=[]
for off in (0,1):
    p,m,v,g=[torch.zeros(n+4,device=dev)[off:off+n] for _ in range(4)]
    p.copy_(p0); g.copy_(gr)
    L.adam_step(p,g,m,v,seg_end,seg_group,glr,1)
    out.append((p.clone(),m.clone(),v.clone()))
for nm,a,b in zip('pmv',*out):
    d=(a-b).abs(); idx=d.nonzero().flatten()
    print(nm, 'ndiff',idx.numel(), 'max',float(d.max()), idx[:10].tolist(), idx[-5:].tolist() if idx.numel() else '')
