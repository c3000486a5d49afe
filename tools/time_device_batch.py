import time, numpy as np, torch, sys
sys.path.insert(0, '.')
from tsdiff_amd import synth, engine
from tsdiff_amd.utils import AttrDict
dev = torch.device('cuda:0')
cfg = engine.make_cfg(synth.DEFAULT_MODEL_CONFIG)
b = synth.wb97xd3_like_batch(200, seed=1)
g = {k: torch.from_numpy(v).to(dev) for k, v in b.items() if isinstance(v, np.ndarray)}
for defer in (False, True):
    for _ in range(5):
        db = engine.DeviceBatch(cfg, g['atom_type'], g['r_feat'], g['p_feat'], g['bond_index'], g['bond_type'], g['batch'], g['num_nodes_per_graph'], defer_status=defer)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        db = engine.DeviceBatch(cfg, g['atom_type'], g['r_feat'], g['p_feat'], g['bond_index'], g['bond_type'], g['batch'], g['num_nodes_per_graph'], defer_status=defer)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('defer', defer, 'host ms per build', (t1 - t0) / 50 * 1e3, 'incl gpu', (t2 - t0) / 50 * 1e3)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(50):
    db = engine.DeviceBatch(cfg, g['atom_type'], g['r_feat'], g['p_feat'], g['bond_index'], g['bond_type'], g['batch'], g['num_nodes_per_graph'], defer_status=True)
pr.disable(); pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
