import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The sharded-rollout tests run up to 8 ranks as contexts of ONE process on ONE GPU with device-side flag waits between
# them (peer exchange): every context's stream needs a hardware queue of its own, or a waiting kernel blocks the queue
# that carries the kernel it waits for.  HIP multiplexes streams over 4 hardware queues by default.  (One process per GPU,
# the production layout, has one stream per device and is not affected.)  Must be set before the HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
