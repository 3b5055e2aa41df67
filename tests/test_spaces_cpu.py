"""CPU tests of the gym-shaped spaces of the per-agent CrowdEnv view (crowd_ppo/crowd_env_2f.py:49-51)."""
import numpy as np

from egogen_amd.spaces import Box, Dict, crowd_env_spaces


def test_spaces_match_the_reference_declaration():
    act, obs = crowd_env_spaces()
    assert isinstance(act, Box) and act.shape == (128,) and act.dtype == np.float32
    assert np.all(act.low == -6.0) and np.all(act.high == 6.0)
    assert isinstance(obs, Dict) and list(obs.spaces.keys()) == ["state", "egosensing", "dist", "time"]
    for k, (lo, hi, shape) in {"state": (-2.0, 2.0, (2, 402)), "egosensing": (-1.0, 1.0, (2, 32)), "dist": (0.0, 1.0, (1,)),
                               "time": (0.0, 1.0, (1,))}.items():
        sp = obs[k]
        assert sp.shape == shape and np.all(sp.low == lo) and np.all(sp.high == hi), k


def test_box_sample_contains_seed():
    act, obs = crowd_env_spaces()
    act.seed(3)
    a = act.sample()
    act.seed(3)
    b = act.sample()
    assert a.shape == (128,) and a.dtype == np.float32 and np.array_equal(a, b)
    assert act.contains(a) and a in act
    assert not act.contains(a[:64]) and not act.contains(np.full(128, 6.5, np.float32))
    s = obs.sample()
    assert list(s.keys()) == ["state", "egosensing", "dist", "time"] and obs.contains(s)
    s["dist"] = np.array([2.0], np.float32)
    assert not obs.contains(s)
    assert not obs.contains({"state": s["state"]})
