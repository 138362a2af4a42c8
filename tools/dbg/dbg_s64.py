import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden, golden_meta, golden_weights, scene_from_golden, region_kwargs
from pstl_diffusion_policy_amd.engine import PackedWeights, SceneBatch, Sampler
from pstl_diffusion_policy_amd.synthetic import default_hparams
dev = torch.device('cuda:0')
d = load_golden('e7_s64_guid'); meta = golden_meta(d); hp = default_hparams()
sm = Sampler(PackedWeights(golden_weights(), dev), hp)
sb = SceneBatch({k: torch.from_numpy(v) for k, v in scene_from_golden(d).items()}, meta['S'], hp, dev)
out = sm.sampling_region(sb, meta['steps'], torch.from_numpy(d['x_T']).to(dev), torch.from_numpy(d['z']).to(dev), full_list=True, **region_kwargs(meta))
N = sb.N
cl = out['controls_list'].reshape(meta['steps'], N, 40).cpu().numpy()
ref = d['controls_list'].reshape(meta['steps'], N, 40)
err = np.abs(cl - ref)
for s in range(meta['steps']):
    e = err[s]
    print('step', s, 'max %.3g' % e.max(), 'n>2e-5:', int((e > 2e-5).sum()), 'n>1e-4:', int((e > 1e-4).sum()), 'argmax', np.unravel_index(e.argmax(), e.shape))
s = int(np.argmax(err.reshape(meta['steps'], -1).max(1)))
r, f = np.unravel_index(err[s].argmax(), err[s].shape)
print('worst: step', s, 'row', r, 'elem', f, 'got', cl[s, r, f], 'ref', ref[s, r, f], 'mode', r % 3, 'valid', float(sb.valid[r]))
print('row errs at that step:', np.round(err[s, r] * 1e5, 2))
