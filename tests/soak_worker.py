"""Helper process of tests/test_gpu_rehearsal.py: what bench.py does around its pipelines, for SECONDS seconds, at the bench geometry (GM, L = 9, T = 100,
4 pairs of 2048 keypoints) - build replicas, run a few steps with 1 / 2 / 3 in flight through the ordered exchange lane (RCCL, one rank,
IMP_FORCE_COLLECTIVES=1), drop them - while a SECOND process group with its own stream and host thread issues all-gathers and a third thread parks
CU-holding kernels on the chip (imp_debug_hold_cus).  Counts the waiting launches (chip-resident Sinkhorn, fused layer) that timed out, prints their
post-mortem records and whether every step still produced the same result.  One JSON line.   usage: python tests/soak_worker.py [seconds] [hold 0|1]"""
import json
import os
import sys
import threading
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    from helpers import eval_config, make_hip_model
    from imp_release_amd import _lib, eval_loop, pipeline, synthetic
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 25.0
    hold = len(sys.argv) > 2 and sys.argv[2] == '1'
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    cfg = eval_config(n_layers=9, sinkhorn_iterations=100)
    sd = synthetic.make_state_dict(cfg, 'GM', seed=0)
    model = make_hip_model('GM', cfg, sd)
    B, N = 4, 2048
    pair = synthetic.make_correlated_pair(N, N, seed=100, batch=B)
    data = {k: torch.from_numpy(v).to(dev) for k, v in pair.items() if k != 'image_shape'}
    data['image0'] = data['image1'] = torch.zeros(pair['image_shape'], device=dev)

    def make_step(m):
        def step_fn():
            out = m.produce_matches(data, p=0.2, only_last=True)
            return out['indices0'][-1], out['mscores0'][-1]
        return step_fn

    os.environ['IMP_FORCE_COLLECTIVES'] = '1'
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    g2 = dist.new_group(ranks=[0], backend='nccl')
    stop = threading.Event()
    noise = {'collectives': 0, 'holds': 0, 'ok': True}

    def second_group():
        torch.cuda.set_device(dev)
        st = torch.cuda.Stream(device=dev)
        x = torch.arange(4096, device=dev, dtype=torch.float32)
        with torch.cuda.stream(st):
            while not stop.is_set():
                out = torch.empty_like(x)
                dist.all_gather_into_tensor(out, x, group=g2)
                st.synchronize()
                noise['ok'] = noise['ok'] and bool(torch.equal(out, x))
                noise['collectives'] += 1
                time.sleep(0.002)

    def holder():
        torch.cuda.set_device(dev)
        st = torch.cuda.Stream(device=dev)
        L = _lib.lib()
        rng = np.random.default_rng(1)
        while not stop.is_set():
            L.imp_debug_hold_cus(0, int(rng.choice([1, 8, 32])), int(rng.integers(100, 5000)), st.cuda_stream)
            st.synchronize()
            noise['holds'] += 1
            time.sleep(0.003)

    threads = [threading.Thread(target=second_group)] + ([threading.Thread(target=holder)] if hold else [])
    for t in threads:
        t.start()
    t_end = time.time() + secs
    cycles = steps = raised = 0
    voided, repaired, pms, same, ref = 0, 0, [], True, None
    seen, pm_seen = {}, set()
    try:
        while time.time() < t_end:
            for k in (1, 2, 3):
                reps = [model] if k == 1 else eval_loop.replicate(model, k)
                pp = pipeline.StepPipeline([make_step(m) for m in reps], B, device=dev, exchange_every=1 + 7 * (cycles % 2))
                try:
                    n = 10 + cycles % 5
                    r = pp.run(n)
                    torch.cuda.synchronize()
                    steps += n
                    if ref is None:
                        ref = (r[0].clone(), r[1].clone())
                    elif not (torch.equal(r[0], ref[0]) and torch.equal(r[1], ref[1])):
                        same = False
                except _lib.ResidentSinkhornTimeout:
                    raised += 1
                    torch.cuda.synchronize()
                for m in reps:                     # what these contexts saw (replicas are dropped below; the base model's counters run on)
                    ctx = m._ensure_ctx()
                    h = ctx.resident_health(False)
                    if h is False:                 # a fresh event: the look recovered the context
                        h = ctx.resident_health(False)
                    seen[id(ctx)] = (int(h[0]) if h else 0, ctx.resident_repaired())
                    pm = ctx.resident_postmortem()
                    if pm is not None and (id(ctx), pm['voided_so_far']) not in pm_seen:
                        pm_seen.add((id(ctx), pm['voided_so_far']))
                        pms.append(pm)
                if k > 1:                          # (the copies' counters end here; reps[0] is the base model, counted at the end)
                    for m in reps[1:]:
                        v, rp = seen.pop(id(m._ensure_ctx()))
                        voided += v; repaired += rp
                del pp, reps
            cycles += 1
        for v, rp in seen.values():
            voided += v; repaired += rp
    finally:
        stop.set()
        for t in threads:
            t.join()
    print(json.dumps({'seconds': secs, 'cycles': cycles, 'steps': steps, 'calls_raised': raised, 'voided_launches': voided, 'repaired_in_call': repaired,
                      'results_identical': same, 'second_group_collectives': noise['collectives'], 'second_group_ok': noise['ok'], 'cu_holds': noise['holds'],
                      'postmortems': pms[:8]}))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
