import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def fqref():
    """The CPU oracle (oracle/libfqref.so) — the checker, never the thing under test."""
    from oracle import fqref as m
    m.lib()
    return m


@pytest.fixture(scope="session", autouse=True)
def poisoned_device_memory():
    """Before the first test, fill most of the free device memory with a byte (0xA5; FQH_TEST_POISON=<byte> picks another,
    FQH_TEST_POISON=off skips it) and give it back to the driver — whatever the library allocates afterwards (workspaces, line
    buffers, lists) starts out as garbage instead of the zeros a fresh box hands out.  A result that depends on memory nobody
    wrote shows up as a failure here; on by default since round 5, so the driver's own `pytest -m gpu` starts from garbage too
    (round 4's stream race was invisible on zero-filled memory)."""
    v = os.environ.get("FQH_TEST_POISON", "0xA5")
    if v.lower() not in ("off", "no", "none", ""):
        import torch
        if torch.cuda.is_available():
            free, _ = torch.cuda.mem_get_info()
            blocks = []
            left = int(free * 0.9)
            while left > (1 << 30):
                n = min(left, 8 << 30)
                blocks.append(torch.full((n,), int(v, 0) & 0xFF, dtype=torch.uint8, device="cuda:0"))
                left -= n
            torch.cuda.synchronize()
            del blocks
            torch.cuda.empty_cache()
    yield
