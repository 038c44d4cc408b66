"""Helper process of test_gpu_dist.py: ONE rank, backend nccl (= RCCL), IMP_FORCE_COLLECTIVES=1 so that the per-step result
exchange really goes through all_gather_into_tensor on the ordered lane of StepPipeline, with 3 model replicas in flight.
Prints one JSON line: whether every step's exchanged result equals the sequential, collective-free result."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    from helpers import eval_config, make_hip_model
    from imp_release_amd import eval_loop, pipeline, synthetic
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    cfg = eval_config(n_layers=3, sinkhorn_iterations=20)
    sd = synthetic.make_state_dict(cfg, 'GM', seed=1)
    m = make_hip_model('GM', cfg, sd)
    datas = []
    for k in range(4):
        pair = synthetic.make_correlated_pair(700, 640, seed=160 + k, batch=2)
        d = {kk: torch.from_numpy(v).to(dev) for kk, v in pair.items() if kk != 'image_shape'}
        d['image0'] = d['image1'] = torch.zeros(pair['image_shape'], device=dev)
        datas.append(d)

    def make_fn(model, start, stride):
        st = {'s': start}

        def fn():
            d = datas[st['s'] % len(datas)]
            st['s'] += stride
            out = model.produce_matches(d, p=0.2, only_last=True)
            return out['indices0'][-1], out['mscores0'][-1]
        return fn

    os.environ.pop('IMP_FORCE_COLLECTIVES', None)
    seq = pipeline.StepPipeline([make_fn(m, 0, 1)], 2, device=dev).run(7, keep=True)          # no process group yet
    os.environ['IMP_FORCE_COLLECTIVES'] = '1'
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    reps = eval_loop.replicate(m, 3)
    par = pipeline.StepPipeline([make_fn(r, i, 3) for i, r in enumerate(reps)], 2, device=dev).run(7, keep=True)
    same = all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(seq, par))
    distinct = not torch.equal(seq[0][0], seq[1][0])
    # round 5: ONE collective per 3 steps (exchange_every: dist.all_gather_matches_steps on the same lane)
    par3 = pipeline.StepPipeline([make_fn(r, i, 3) for i, r in enumerate(reps)], 2, device=dev, exchange_every=3).run(7, keep=True)
    same3 = len(par3) == 7 and all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(seq, par3))
    # round 4: a SECOND communicator alive and busy in the same process (another process group, its own stream and host
    # thread) beside the ordered exchange lane and the spin gate of the waiting kernels: the steps must still come out identical
    import threading
    g2 = dist.new_group(ranks=[0], backend='nccl')
    stop, counts = threading.Event(), []

    def noise():
        torch.cuda.set_device(dev)
        s2 = torch.cuda.Stream(dev)
        x, out = torch.ones(4096, device=dev), torch.empty(4096, device=dev)
        n = 0
        with torch.cuda.stream(s2):
            while not stop.is_set():
                dist.all_gather_into_tensor(out, x, group=g2)
                n += 1
                if n % 16 == 0:
                    s2.synchronize()
            s2.synchronize()
        counts.append((n, bool((out == 1).all())))

    th = threading.Thread(target=noise, daemon=True)
    th.start()
    par2 = pipeline.StepPipeline([make_fn(r, i, 3) for i, r in enumerate(reps)], 2, device=dev).run(7, keep=True)
    stop.set()
    th.join(timeout=60)
    same2 = all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(seq, par2))
    healthy = all(r._ensure_ctx().resident_health() == (0, 0) for r in reps)
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps({'steps': len(par), 'same': bool(same), 'distinct_batches': bool(distinct),
                      'backend': 'nccl', 'matched': int((par[-1][0] >= 0).sum()), 'same_beside_a_second_process_group': bool(same2), 'same_with_one_exchange_per_3_steps': bool(same3),
                      'second_group_collectives': counts[0][0] if counts else -1, 'second_group_ok': bool(counts and counts[0][1]),
                      'no_waiting_kernel_timed_out': bool(healthy)}), flush=True)


if __name__ == '__main__':
    main()
