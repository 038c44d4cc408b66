import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


# the soak module is opt-in (IMP_SOAK=n): without the switch it is not collected at all
collect_ignore = [] if os.environ.get('IMP_SOAK') else ['test_gpu_soak.py', 'diag_soak.py']
