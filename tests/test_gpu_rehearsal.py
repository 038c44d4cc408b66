"""GPU: the whole-chip waiting kernels beside work they do not control (round 6, VERDICT r5 #2c, #4, #5).

The chip-resident Sinkhorn and the fused layer launch need every workgroup of the launch co-resident; the spin gate (context.hip) orders only this
library's own waiting launches.  A collective's kernel that waits for a slower peer holds CUs outside the gate - on one GPU that is rehearsed with a
dummy kernel that parks 1 / 8 / 32 workgroups of 96 KB LDS each for 0.1 ... 5 ms (imp_debug_hold_cus): a waiting launch's workgroup cannot land beside
one, so the launch stalls until the holder leaves; its bounded waits (2^21 polls, seconds) must never run out, and results must not move."""
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np
import pytest
import torch

from helpers import eval_config, make_hip_model
from imp_release_amd import _lib, eval_loop, pipeline, synthetic

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = 'cuda'


def _last_json(out):
    for ln in reversed(out.strip().splitlines()):
        if ln.startswith('{'):
            return json.loads(ln)
    raise AssertionError('no JSON line in:\n' + out[-2000:])


def test_bounded_soak_three_streams_a_second_process_group_and_cu_holders():
    """~30 s of bench-like replica churn (1 / 2 / 3 steps in flight on the RCCL lane at exchange_every 1 and 8) beside a second communicator's
    all-gathers and CU-holding kernels: every step identical, no call raises (a voided waiting launch is repaired inside its call), and what was
    voided - nothing, on the boxes this was developed on - comes with its post-mortem record"""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29655', HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'soak_worker.py'), os.environ.get('IMP_SOAK_SECONDS', '30'), '1'], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _last_json(r.stdout)
    print('soak:', json.dumps(j))
    assert j['results_identical'] and j['calls_raised'] == 0 and j['steps'] > 100, j
    assert j['second_group_ok'] and j['second_group_collectives'] > 0 and j['cu_holds'] > 0, j
    assert j['voided_launches'] == j['repaired_in_call'], f'a voided waiting launch was not repaired inside its call: {j}'
    assert j['voided_launches'] == 0, f'waiting launches timed out - post-mortems: {j["postmortems"]}'


@pytest.mark.parametrize('wgs', [1, 8, 32])
def test_three_replicas_through_the_lane_beside_a_kernel_that_holds_compute_units(wgs):
    """what an RCCL kernel waiting for a 9 %-slower peer looks like: `wgs` parked workgroups for 0.1 - 5 ms, again and again, while three replicas run
    steps (fused layer launches + resident Sinkhorn at the bench geometry).  No IMP_E_RESIDENT, identical results; the stall is printed"""
    cfg = eval_config(n_layers=3, sinkhorn_iterations=50)
    sd = synthetic.make_state_dict(cfg, 'GM', seed=2)
    m = make_hip_model('GM', cfg, sd)
    pair = synthetic.make_correlated_pair(2048, 2048, seed=7, batch=4)
    data = {k: torch.from_numpy(v).to(DEV) for k, v in pair.items() if k != 'image_shape'}
    data['image0'] = data['image1'] = torch.zeros(pair['image_shape'], device=DEV)
    reps = eval_loop.replicate(m, 3)

    def make_step(mm):
        def fn():
            out = mm.produce_matches(data, p=0.2, only_last=True)
            return out['indices0'][-1], out['mscores0'][-1]
        return fn

    def run(n):
        pp = pipeline.StepPipeline([make_step(r) for r in reps], 4, device=torch.device(DEV, 0))
        t0 = time.perf_counter()
        out = pp.run(n)
        torch.cuda.synchronize()
        return out, (time.perf_counter() - t0) / n * 1e3

    with torch.no_grad():
        run(6)
        ref, quiet_ms = run(30)
        stop, held = threading.Event(), [0, 0.0]

        def holder():
            torch.cuda.set_device(0)
            st = torch.cuda.Stream(device=DEV)
            L = _lib.lib()
            rng = np.random.default_rng(wgs)
            while not stop.is_set():
                us = int(rng.integers(100, 5000))
                assert L.imp_debug_hold_cus(0, wgs, us, st.cuda_stream) == 0
                st.synchronize()
                held[0] += 1; held[1] += us * 1e-3
                time.sleep(0.001)

        th = threading.Thread(target=holder)
        th.start()
        try:
            got, noisy_ms = run(30)
        finally:
            stop.set()
            th.join()
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    for r in reps:
        assert r._ensure_ctx().resident_health() == (0, 0), 'a waiting launch timed out beside the CU holder'
    print(f'CU holder with {wgs} workgroup(s): {quiet_ms:.3f} ms per step quiet, {noisy_ms:.3f} ms beside {held[0]} holds ({held[1]:.0f} ms held in total): '
          f'stall {100 * (noisy_ms / quiet_ms - 1):+.1f} %')
