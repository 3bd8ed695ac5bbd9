import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: imports /root/reference (only present in the build container)")
    # the reference's own LegController converts 1-element arrays to scalars (numpy >= 1.25 deprecation): its noise, not a finding of these tests
    config.addinivalue_line("filterwarnings", "ignore:Conversion of an array with ndim:DeprecationWarning")
