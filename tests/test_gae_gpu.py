"""GPU: GAE scan kernel vs golden (reference) and oracle, both conventions."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests import golden_util as gu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_gae_matches_reference_golden_bit_exact():
    from gennbv_amd import gae
    fx = gu.load("F8_gae")
    adv, ret = gae.compute_returns_and_advantage(T(fx["rewards"]), T(fx["values"]), T(fx["episode_starts"]),
                                                 T(fx["last_values"]), T(fx["dones"]), 0.99, 0.95)
    assert adv.cpu().numpy().tobytes() == fx["sb3_advantages"].tobytes()
    assert ret.cpu().numpy().tobytes() == fx["sb3_returns"].tobytes()
    rret, radv = gae.compute_returns_rsl(T(fx["rewards"]), T(fx["values"]), T(fx["rsl_dones"]), T(fx["last_values"]),
                                         0.99, 0.95, normalize=False)
    assert rret.cpu().numpy().tobytes() == fx["rsl_returns"].tobytes()
    _, nadv = gae.compute_returns_rsl(T(fx["rewards"]), T(fx["values"]), T(fx["rsl_dones"]), T(fx["last_values"]),
                                      0.99, 0.95, normalize=True)
    np.testing.assert_allclose(nadv.cpu().numpy(), fx["rsl_advantages_normalized"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("t,n", [(1, 1), (7, 33), (128, 256), (300, 70), (513, 2048)])
def test_gae_vs_oracle_shapes(t, n):
    from gennbv_amd import gae
    rs = np.random.RandomState(t * 1000 + n)
    r = rs.randn(t, n).astype(np.float32); v = rs.randn(t, n).astype(np.float32)
    es = (rs.rand(t, n) < 0.1).astype(np.uint8); lv = rs.randn(n).astype(np.float32)
    dn = (rs.rand(n) < 0.5).astype(np.uint8)
    a_o, r_o = orc.gae_sb3(r, v, es, lv, dn, 0.99, 0.95)
    a, rt = gae.compute_returns_and_advantage(T(r).view(t, n, 1), T(v).view(t, n, 1), T(es).view(t, n, 1), T(lv).view(n, 1), T(dn), 0.99, 0.95)
    assert a.shape == (t, n, 1)
    assert a.cpu().numpy().tobytes() == a_o.tobytes() and rt.cpu().numpy().tobytes() == r_o.tobytes()
    rr_o, ra_o = orc.gae_rsl(r, v, es, lv, 0.998, 0.9)
    rr, ra = gae.compute_returns_rsl(T(r), T(v), T(es), T(lv), 0.998, 0.9, normalize=False)
    assert rr.cpu().numpy().tobytes() == rr_o.tobytes() and ra.cpu().numpy().tobytes() == ra_o.tobytes()


def test_gae_all_terminal_and_no_terminal():
    from gennbv_amd import gae
    t, n = 64, 40
    r = np.ones((t, n), np.float32); v = np.zeros((t, n), np.float32); lv = np.zeros(n, np.float32)
    for fill in (0, 1):
        es = np.full((t, n), fill, np.uint8); dn = np.full(n, fill, np.uint8)
        a_o, r_o = orc.gae_sb3(r, v, es, lv, dn)
        a, rt = gae.compute_returns_and_advantage(T(r), T(v), T(es), T(lv), T(dn), 0.99, 0.95)
        assert a.cpu().numpy().tobytes() == a_o.tobytes()
        if fill:  # every step terminal: advantage == reward
            assert np.all(a_o == 1.0)
