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


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests are skipped (not failed) on a host without a GPU or without the built HIP library"""
    import pytest
    reason = None
    try:
        import torch
        if not torch.cuda.is_available():
            reason = 'no GPU on this host'
    except Exception as e:          # pragma: no cover
        reason = f'torch unavailable: {e}'
    if reason is None and not os.path.exists(os.path.join(ROOT, 'imp-release_amd', 'csrc', 'libimp_hip.so')):
        reason = 'libimp_hip.so is not built (python __graft_entry__.py)'
    if reason:
        skip = pytest.mark.skip(reason=reason)
        for it in items:
            if 'gpu' in it.keywords:
                it.add_marker(skip)


def pytest_terminal_summary(terminalreporter):
    import helpers
    n = sum(c for _, c in helpers.EXCUSED)
    terminalreporter.write_line(f'parity: {n} index mismatches excused as threshold ties in non-strict comparisons (kernel A/B runs only: since round 4 the '
                                f'N = 2048 / 4096 sizes are compared with reference-captured fixtures too); golden-fixture comparisons are strict on the match '
                                f'INDICES (0 mismatches allowed, scores within 1e-4); mutual-nearest-neighbour flips of UNMATCHED low-score keypoints are '
                                f'tolerated only in *trained* fixtures and listed below')
    for what, c in helpers.EXCUSED:
        terminalreporter.write_line(f'  excused: {what}: {c}')
    if helpers.LOW_FLIPS:
        terminalreporter.write_line(f'parity: {sum(c for _, c in helpers.LOW_FLIPS)} mutual-nearest-neighbour flips on keypoints unmatched on both sides '
                                    f'(score < p, indices identical) tolerated in trained-weight fixtures / soak:')
        for what, c in helpers.LOW_FLIPS:
            terminalreporter.write_line(f'  low-score flip: {what}: {c}')
    if helpers.SP_MOVED:
        moved, cut = sum(m for _, m, _ in helpers.SP_MOVED), sum(b for _, _, b in helpers.SP_MOVED)
        terminalreporter.write_line(f'superpoint (sorted top-k outputs): {moved} list positions differ inside runs of reference scores closer than the '
                                    f'score tolerance, {cut} keypoints swapped with an equally scored one at the cut; everything else identical')
        for what, _, _ in helpers.SP_MOVED:
            terminalreporter.write_line(f'  {what}')
    if getattr(helpers, 'HARD_SET', None):
        terminalreporter.write_line('harder two-view set (tests/golden/hard_loops.npz, loops vs the imported reference):')
        for line in helpers.HARD_SET:
            terminalreporter.write_line('  ' + line)
