#!/bin/bash
timeout 900 python -m pytest tests/test_event_sampler.py -m gpu -x -q 2>&1 | tail -4 | cut -c1-200
python - <<'PY'
import torch, time, numpy as np, sys
sys.path.insert(0,'.')
from enerf_amd.event_sampler import build_event_tables, event_pair_batch
rng=np.random.default_rng(0); n=2_000_000
ev=np.stack([rng.integers(0,346,n), rng.integers(0,260,n), rng.permutation(n*2)[:n].astype(np.float64), rng.choice([-1.,1.],n)],1).astype(np.float32)
ev=torch.from_numpy(ev).cuda()
torch.cuda.synchronize(); t0=time.perf_counter(); t=build_event_tables(ev); torch.cuda.synchronize(); print('build tables (2M events): %.1f ms'%((time.perf_counter()-t0)*1e3), t['events'].shape)
N=t['events'].shape[0]; poses=torch.eye(4,device='cuda')[:3].repeat(N,1,1)
g=torch.Generator(device='cuda').manual_seed(0)
for _ in range(3): event_pair_batch(t,poses,(320.,320.,173.,130.),4096,True,8,generator=g)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(100): event_pair_batch(t,poses,(320.,320.,173.,130.),4096,True,8,generator=g)
torch.cuda.synchronize(); print('event_pair_batch(4096): %.3f ms'%((time.perf_counter()-t0)*10))
PY
