"""The seeding chain of the reference's gym wrapper, restated.

The reference pins gym==0.9.4 (setup.py:13) and uses ``gym.utils.seeding.np_random`` /
``hash_seed`` in ``simglucose/envs/simglucose_gym_env.py:38,54,62-64``.  gym is not a dependency
here; its published algorithm is: ``hash_seed(s)`` = the first 8 bytes of SHA-512(str(s)) read as a
little-endian integer (32-bit words), and ``np_random(s)`` seeds ``numpy.random.RandomState`` with
the 32-bit words of ``hash_seed(s mod 2**64)``.  Pinned by the reference's own known answers
(tests/test_seed.py:19,23: seed 0 -> 23:00 start after reset, seed 1000 -> 14:00) in
tests/test_host_surface.py.
"""
import hashlib
import os
import struct

import numpy as np


def _bigint_from_bytes(raw):
    raw = raw + b"\0" * (4 - len(raw) % 4)          # gym pads even when already aligned
    words = struct.unpack("%dI" % (len(raw) // 4), raw)
    return sum(w << (32 * i) for i, w in enumerate(words))


def hash_seed(seed=None, max_bytes=8):
    if seed is None:
        seed = create_seed(max_bytes=max_bytes)
    digest = hashlib.sha512(str(seed).encode("utf8")).digest()
    return _bigint_from_bytes(digest[:max_bytes])


def create_seed(a=None, max_bytes=8):
    if a is None:
        return _bigint_from_bytes(os.urandom(max_bytes))
    if isinstance(a, (int, np.integer)):
        return int(a) % 2 ** (8 * max_bytes)
    raise ValueError("Invalid type for seed: %r" % (type(a),))


def np_random(seed=None):
    if seed is not None and not (isinstance(seed, (int, np.integer)) and 0 <= seed):
        raise ValueError("Seed must be a non-negative integer or omitted, not %r" % (seed,))
    seed = create_seed(seed)
    h = hash_seed(seed)
    words = []
    while h > 0:
        h, w = divmod(h, 2 ** 32)
        words.append(w)
    rng = np.random.RandomState()
    rng.seed(words or [0])
    return rng, seed


def derive_episode(rng):
    """simglucose_gym_env.py:62-66 -> (sensor seed, scenario seed, patient seed, start hour)."""
    seed2 = hash_seed(rng.randint(0, 1000)) % 2 ** 31
    seed3 = hash_seed(seed2 + 1) % 2 ** 31
    seed4 = hash_seed(seed3 + 1) % 2 ** 31
    hour = rng.randint(low=0.0, high=24.0)
    return seed2, seed3, seed4, int(hour)
