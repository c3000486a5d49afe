"""cProfile of the host side of the training step (python bench.py --workload train path)."""
import cProfile, pstats, sys
import numpy as np, torch
sys.path.insert(0, '.')
from tsdiff_amd import synth
from tsdiff_amd.distributed import dp_backward
from tsdiff_amd.epsnet import get_model
from tsdiff_amd.utils import AttrDict
dev = torch.device('cuda:0')
cfg = synth.DEFAULT_MODEL_CONFIG
model = get_model(AttrDict(cfg))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(cfg, 0).items()}, strict=False)
model = model.to(dev).train()
from types import SimpleNamespace
from tsdiff_amd import optim
opt = optim.get_optimizer(SimpleNamespace(type='adam', lr=5e-4, weight_decay=0.0, beta1=0.95, beta2=0.999), model)
batches = []
for k in range(4):
    b = synth.wb97xd3_like_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 200, seed=2000 + k)
    g = {kk: torch.from_numpy(v).to(dev) for kk, v in b.items() if isinstance(v, np.ndarray)}
    g['pos'] = (g['pos'] * 1.5).contiguous()
    batches.append(g)
def step(i):
    g = batches[i % 4]
    model._batches.clear()
    opt.zero_grad()
    loss = model.get_loss(g['atom_type'], g['r_feat'], g['p_feat'], g['pos'], g['bond_index'], g['bond_type'], g['batch'], g["num_nodes_per_graph"], int(sys.argv[1]) if len(sys.argv) > 1 else 200)
    dp_backward(model, loss)
    optim.clip_grad_norm_(model.parameters(), 3000.0)
    opt.step()
for i in range(5): step(i)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(20): step(i)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(60)
