"""Observation / action spaces of the per-agent `CrowdEnv` view (crowd_ppo/crowd_env_2f.py:49-51):

    self.action_space = Box(-6., 6., (128,))
    self.observation_space = Dict({"state": Box(-2., 2., (2, 402)), "egosensing": Box(-1., 1., (2, 32)),
                                   "dist": Box(0., 1.), "time": Box(0., 1.)})

`gymnasium.spaces` when it is installed (the reference's dependency; tianshou's `DummyVectorEnv` / `Collector` read
`.shape`, `.sample()`, `.contains()` and iterate `Dict.spaces`); otherwise the two small classes below, which carry the part
of that interface a vector-env wrapper touches: `shape`, `dtype`, `low`, `high`, `sample()`, `contains()`, `seed()`, and for
`Dict`: `spaces`, `keys()`, `items()`, `__getitem__`, `sample()`, `contains()`.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional, Sequence

import numpy as np

try:  # the real thing, when present
    from gymnasium.spaces import Box, Dict  # type: ignore  # noqa: F401
    HAVE_GYMNASIUM = True
except Exception:  # ModuleNotFoundError in this image
    HAVE_GYMNASIUM = False

    class Box:
        """gymnasium.spaces.Box(low, high, shape=None, dtype=np.float32): a closed box in R^shape, scalar or array bounds."""

        def __init__(self, low, high, shape: Optional[Sequence[int]] = None, dtype=np.float32, seed: Optional[int] = None):
            self.dtype = np.dtype(dtype)
            if shape is None:
                shape = np.broadcast(np.asarray(low), np.asarray(high)).shape or (1,)   # gymnasium: scalar bounds -> shape (1,)
            self.shape = tuple(int(s) for s in shape)
            self.low = np.broadcast_to(np.asarray(low, self.dtype), self.shape).copy()
            self.high = np.broadcast_to(np.asarray(high, self.dtype), self.shape).copy()
            if np.any(self.low > self.high):
                raise ValueError("Box: low > high")
            self._rng = np.random.default_rng(seed)

        def seed(self, seed: Optional[int] = None):
            self._rng = np.random.default_rng(seed)
            return [seed]

        def sample(self) -> np.ndarray:
            return self._rng.uniform(self.low, self.high, self.shape).astype(self.dtype)

        def contains(self, x) -> bool:
            x = np.asarray(x)
            if not np.can_cast(x.dtype, self.dtype, casting="same_kind") and x.dtype.kind not in "fiu":
                return False
            return bool(x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high))

        __contains__ = contains

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

        def __eq__(self, other):
            return (isinstance(other, Box) and self.shape == other.shape and self.dtype == other.dtype
                    and np.array_equal(self.low, other.low) and np.array_equal(self.high, other.high))

    class Dict:
        """gymnasium.spaces.Dict: an ordered mapping name -> space."""

        def __init__(self, spaces=None, seed: Optional[int] = None, **kw):
            self.spaces = OrderedDict(spaces or {})
            self.spaces.update(kw)
            self.shape = None
            self.dtype = None
            if seed is not None:
                self.seed(seed)

        def seed(self, seed: Optional[int] = None):
            return [sp.seed(None if seed is None else seed + i) for i, sp in enumerate(self.spaces.values())]

        def sample(self):
            return OrderedDict((k, sp.sample()) for k, sp in self.spaces.items())

        def contains(self, x) -> bool:
            return (isinstance(x, dict) and set(x.keys()) == set(self.spaces.keys())
                    and all(self.spaces[k].contains(x[k]) for k in self.spaces))

        __contains__ = contains

        def __getitem__(self, k):
            return self.spaces[k]

        def __iter__(self):
            return iter(self.spaces)

        def __len__(self):
            return len(self.spaces)

        def keys(self):
            return self.spaces.keys()

        def values(self):
            return self.spaces.values()

        def items(self):
            return self.spaces.items()

        def __repr__(self):
            return "Dict(" + ", ".join(f"{k!r}: {v!r}" for k, v in self.spaces.items()) + ")"

        def __eq__(self, other):
            return isinstance(other, Dict) and list(self.spaces.items()) == list(other.spaces.items())


def crowd_env_spaces():
    """(action_space, observation_space) exactly as crowd_env_2f.py:49-51 declares them."""
    action = Box(-6.0, 6.0, (128,))
    observation = Dict({"state": Box(-2.0, 2.0, (2, 402)), "egosensing": Box(-1.0, 1.0, (2, 32)),
                        "dist": Box(0.0, 1.0), "time": Box(0.0, 1.0)})
    return action, observation
