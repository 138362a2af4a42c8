"""The in-kernel noise stream (csrc/rng.hpp: Philox4x32 with kPhiloxRounds = 7 rounds + Box-Muller; reference: torch.randn_like,
nusc_train.py:563,584) held to what a DDPM sampler needs of it: standard-normal moments and tails, and no dependence between the
values of neighbouring counters -- rows, column quads, reverse steps, seeds -- which is where a counter-based generator with
too few rounds shows first (Salmon et al., SC'11: Philox4x32-7 passes BigCrush, fewer rounds fail on exactly such patterns)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _draw(sm, sb, step, seed, steps=50):
    return sm.fill_normal(sb, steps, step, seed).cpu().numpy().astype(np.float64)


@pytest.fixture(scope="module")
def setup():
    from pstl_diffusion_policy_amd.engine import PackedWeights, Sampler, SceneBatch
    from pstl_diffusion_policy_amd.nusc_model import init_state_dict
    from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch
    dev = torch.device("cuda:0")
    hp = default_hparams()
    sm = Sampler(PackedWeights(init_state_dict(1007), dev), hp)
    sb = SceneBatch(make_scene_batch(512, K=2, S=64, seed=3), 64, hp, dev)      # 98 304 rows x 40 = 3.9 M values per draw
    return sm, sb


def test_moments_and_tails(setup):
    sm, sb = setup
    z = _draw(sm, sb, 17, 12345)
    n = z.size
    assert np.isfinite(z).all()
    assert abs(z.mean()) < 5.0 / np.sqrt(n)
    assert abs(z.var() - 1.0) < 5.0 * np.sqrt(2.0 / n)
    assert abs((z ** 3).mean()) < 5.0 * np.sqrt(15.0 / n)
    assert abs((z ** 4).mean() - 3.0) < 5.0 * np.sqrt(96.0 / n)
    # tails: P(|z| > 3) = 2.6998e-3, P(|z| > 4) = 6.334e-5 (binomial five-sigma bands)
    for thr, p in ((3.0, 2.6997960632601866e-3), (4.0, 6.334248366623973e-5)):
        k = int((np.abs(z) > thr).sum())
        assert abs(k - n * p) < 5.0 * np.sqrt(n * p) + 1, (thr, k, n * p)
    # every column (= counter word / Box-Muller branch) by itself
    assert np.abs(z.mean(axis=0)).max() < 5.5 / np.sqrt(z.shape[0])
    assert np.abs(z.var(axis=0) - 1.0).max() < 5.5 * np.sqrt(2.0 / z.shape[0])


def test_neighbouring_counters_are_independent(setup):
    sm, sb = setup
    a = _draw(sm, sb, 17, 12345)
    N = a.shape[0]
    band = 5.0 / np.sqrt(a.size)

    def corr(x, y):
        return float((x * y).mean())

    assert abs(corr(a[:-1], a[1:])) < band                      # consecutive rows (counter + 10)
    assert abs(corr(a[:, :36], a[:, 4:])) < band                # consecutive column quads (counter + 1)
    assert abs(corr(a[:, 0::4], a[:, 1::4])) < band             # the two Box-Muller branches of one pair
    assert abs(corr(a[:, 0::4], a[:, 2::4])) < band             # the two pairs of one Philox block
    assert abs(corr(a, _draw(sm, sb, 18, 12345))) < band        # consecutive reverse steps (third counter word + 1)
    assert abs(corr(a, _draw(sm, sb, 17, 12346))) < band        # consecutive seeds (key + 1)
    assert abs(corr(a, _draw(sm, sb, 17, 12345 + (1 << 32)))) < band   # second key word + 1
    # squares too (dependence that leaves the linear correlation alone)
    assert abs(corr(a[:-1] ** 2 - 1, a[1:] ** 2 - 1)) < 2 * band * np.sqrt(2.0) * 2
    assert abs(corr(a ** 2 - 1, _draw(sm, sb, 18, 12345) ** 2 - 1)) < 2 * band * np.sqrt(2.0) * 2
    # uniformity of the underlying 32-bit words through the normal CDF: 64 equiprobable bins, chi-square
    from math import erf
    u = 0.5 * (1.0 + np.vectorize(erf)(a[::4].ravel()[:400000] / np.sqrt(2.0)))
    cnt = np.bincount(np.minimum((u * 64).astype(int), 63), minlength=64)
    chi2 = float(((cnt - cnt.mean()) ** 2 / cnt.mean()).sum())
    assert chi2 < 63 + 5.0 * np.sqrt(2 * 63), chi2
    assert N == sb.N
