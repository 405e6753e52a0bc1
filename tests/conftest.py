import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

FIX = os.path.join(ROOT, "tests", "fixtures")
GOLD = os.path.join(ROOT, "tests", "golden")
SMOKE = os.path.join(FIX, "smoke.brick")
HDR = os.path.join(FIX, "table_mountain_2_puresky_1k.hdr")
LUT = os.path.join(FIX, "lut.txt")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Without -m gpu on a GPU-less machine the gpu tests are skipped instead of failing at vr_create."""
    if "gpu" in (config.getoption("-m") or ""):
        return
    try:
        import volren_amd
        have = volren_amd.load().vr_device_count() > 0
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="no HIP device")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def paths():
    return {"smoke": SMOKE, "hdr": HDR, "lut": LUT, "gold": GOLD, "root": ROOT}
