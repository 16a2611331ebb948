import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # libtorch's intra-op pool is sized by the machine's cores, not by the container's CPU quota (gps_slam_amd/dist_util.py,
    # cap_host_threads: a process over its quota is frozen whole for the rest of a 100 ms accounting period)
    from gps_slam_amd.dist_util import cap_host_threads
    cap_host_threads()
