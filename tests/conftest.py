import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ia():
    import ieache_amd
    if not os.path.exists(ieache_amd.library_path()):
        ieache_amd.build_library()
    return ieache_amd


@pytest.fixture(scope="session")
def O():
    from oracle import oracle
    oracle.lib()
    return oracle


class KeyBundle:
    """Product-generated key material + the oracle's view of the same cloud key."""

    def __init__(self, ia, O, n, N, seed=(1, 2, 3), **kw):
        from ieache_amd import tools
        self.p = ia.default_params().copy(n=n, N=N, **kw)
        k = tools.keygen_raw(self.p, seed)
        self.lwe_key, self.tlwe_key, self.bk, self.ksk = k["lwe_key"], k["tlwe_key"], k["bk"], k["ksk"]
        p = self.p
        self.ck = O.CloudKey(p.n, p.N, p.k, p.l, p.Bgbit, p.ks_t, p.ks_basebit, self.bk, self.ksk)
        self.tools = tools

    def enc(self, bits, seed):
        return self.tools.encrypt_bits(self.p, self.lwe_key, np.asarray(bits, dtype=np.uint8), seed)

    def dec(self, samples):
        return self.tools.decrypt_bits(self.p, self.lwe_key, samples)


@pytest.fixture(scope="session")
def make_keys(ia, O):
    cache = {}

    def make(n, N, seed=(1, 2, 3), **kw):
        key = (n, N, seed, tuple(sorted(kw.items())))
        if key not in cache:
            cache[key] = KeyBundle(ia, O, n, N, seed, **kw)
        return cache[key]

    return make


@pytest.fixture(scope="session")
def gpu_ctx(ia, make_keys):
    """Context factory for -m gpu tests; fails loudly when no GPU is present."""
    made = {}

    def make(n, N, **kw):
        kb = make_keys(n, N, **kw)
        key = (n, N, tuple(sorted(kw.items())))
        if key not in made:
            assert ia.device_count() > 0, "gpu-marked test needs a HIP device"
            made[key] = ia.Context.from_arrays(kb.p, kb.bk, kb.ksk)
        return kb, made[key]

    yield make
    for c in made.values():
        c.close()
