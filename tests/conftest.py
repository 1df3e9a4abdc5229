import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure; oracle/liboracle.so)."""
    import oracle_py

    oracle_py.set_math_mode(False)
    return oracle_py


@pytest.fixture(scope="session")
def pt():
    """The product binding (libpt_amd.so); importing needs no GPU."""
    return importlib.import_module("thu-acg-f2024-path-tracer_amd")


@pytest.fixture(scope="session")
def ctx(pt):
    """One GPU context for the whole session; fails loudly (no CPU fallback) without a device."""
    c = pt.Context(0)
    yield c
    c.close()


@pytest.fixture()
def det(orc):
    """Oracle in deterministic-math mode (bit-exact parity with the kernels); restored after."""
    orc.set_math_mode(True)
    yield orc
    orc.set_math_mode(False)


@pytest.fixture(scope="session")
def scene_images(pt):
    cache = {}

    def get(scene_id):
        if scene_id not in cache:
            cache[scene_id] = {n: pt.decode_image_rgb8(os.path.join(pt.ASSET_DIR, n)) for n in pt.SCENE_IMAGE_FILES.get(scene_id, [])}
        return cache[scene_id]

    return get
