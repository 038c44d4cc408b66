#!/usr/bin/env python3
"""Synthetic analogue of `python -m eval.eval_imp --matching_method IMP|EIMP --use_iterative` (BASELINE config 5):
N independent synthetic pairs through the iterative loop, sharded over the ranks of one node.

    python tools/eval_synthetic.py --pairs 16 --model EIMP --kpts 2048
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/eval_synthetic.py --pairs 4000 --model EIMP

Pairs are two-view consistent (synthetic.make_two_view_pair: a known relative pose behind the re-observed keypoints), so the report
of eval/eval_imp.py:213-227 - pose AUC@5/10/20/50, precision, matching score - is computed and printed like the reference's
(on seeded random weights the numbers say how the pipeline behaves, not how well a trained matcher does).
--pose none: no pose is ever found (the loops run all 15 iterations, pose errors are infinite); --pose gpu: the GPU pose step
(imp_release_amd.pose, csrc/pose.hip: NOT OpenCV's MAGSAC) sits in the loop's estimate_pose slot, up to 7 calls per pair + 1."""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imp_release_amd as P
from imp_release_amd import synthetic, eval_loop


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pairs', type=int, default=16)
    ap.add_argument('--model', choices=['IMP', 'EIMP'], default='EIMP')
    ap.add_argument('--kpts', type=int, default=2048)
    ap.add_argument('--weights', choices=['matching', 'uniform', 'trained'], default='matching',
                    help="synthetic.make_state_dict style: 'matching' = a matcher that works on these pairs (the report means something)")
    ap.add_argument('--bin-score', type=float, default=None, help='default: 30 with --weights matching, else 5')
    ap.add_argument('--workers', type=int, default=1, help='pairs in flight per GPU (model replicas + streams)')
    ap.add_argument('--pose', choices=['none', 'gpu'], default='gpu')
    ap.add_argument('--overlap', type=float, default=0.6)
    ap.add_argument('--noise-px', type=float, default=0.5)
    ap.add_argument('--hard', action='store_true', help='the harder evaluation set of round 4 (synthetic.make_hard_two_view_pair: N ~ U(1000, 2048), '
                                                        'overlap 0.2-0.8, noise 0.5-2 px, 30-70 %% look-alike outliers) instead of fixed-size easy pairs')
    ap.add_argument('--lockstep', type=int, default=1, help='that many pairs advance together as one ragged batch (matching_iterative_lockstep / matching_iterative_uncertainty_lockstep)')
    ap.add_argument('--schedule', choices=['block', 'lpt'], default='block', help="pairs -> ranks: contiguous blocks or longest-first by n0 * n1")
    a = ap.parse_args()
    rank, world, lr = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(lr)
    dev = torch.device('cuda', lr)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    cfg = {'descriptor_dim': 256, 'sinkhorn_iterations': 20, 'match_threshold': 0.2, 'with_sinkhorn': True, 'n_layers': 15,
           'GNN_layers': ['self', 'cross'] * 15, 'ac_fn': 'relu', 'norm_fn': 'in', 'n_min_tokens': 256}
    name = 'AdaGMN' if a.model == 'EIMP' else 'DGNNS'
    bs = a.bin_score if a.bin_score is not None else (synthetic.MATCHING_BIN_SCORE if a.weights == 'matching' else 5.0)
    sd = synthetic.make_state_dict(cfg, name, seed=0, bin_score=bs, style=a.weights)
    m = getattr(P, name)(cfg).eval()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.to(dev)

    cache, pinned = {}, {}                       # synthetic pairs are generated (and staged in pinned memory, like a loader would) once, outside the timed loop
    UP = ('keypoints0', 'keypoints1', 'scores0', 'scores1', 'descriptors0', 'descriptors1')

    def host_pair(pid):
        if pid not in cache:
            cache[pid] = (synthetic.make_hard_two_view_pair(seed=1000 + pid) if a.hard else
                          synthetic.make_two_view_pair(a.kpts, a.kpts - 37, seed=1000 + pid, overlap=a.overlap, noise_px=a.noise_px))
            pinned[pid] = {k: torch.from_numpy(cache[pid][k]).pin_memory() for k in UP}
        return cache[pid]

    def provider(pid):
        pair = host_pair(pid)
        d = {k: pinned[pid][k].to(dev, non_blocking=True) for k in UP}          # six async uploads on the worker's stream
        d['image0'] = d['image1'] = torch.empty(pair['image_shape'], device='meta')      # only .shape is read
        d['pts0_cpu'] = pair['keypoints0'][0]; d['pts1_cpu'] = pair['keypoints1'][0]
        d.update({k: pair[k] for k in ('K0', 'K1', 'T_0to1', 'E')})
        return d

    for pid in range(a.pairs):                   # (every rank generates all pairs: the longest-first schedule needs every pair's size)
        host_pair(pid)
    reps = eval_loop.replicate(m, a.workers)
    kw = dict(eimp=a.model == 'EIMP', workers=a.workers, replicas=reps, lockstep=a.lockstep, schedule=a.schedule,
              pair_cost=lambda pid: cache[pid]['keypoints0'].shape[1] * cache[pid]['keypoints1'].shape[1])
    if a.pose == 'gpu':
        from imp_release_amd import pose as gpose
        kw['estimate_pose'] = gpose.estimate_pose
    eval_loop.run_pairs_sharded(m, provider, min(a.pairs, 2 * world * a.workers), **kw)      # warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    table = eval_loop.run_pairs_sharded(m, provider, a.pairs, **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    health = []
    for r_ in reps:
        try:
            h_ = r_._ensure_ctx().resident_health(raise_on_timeout=False)
            health.append(list(h_) if h_ is not False else 'voided')
        except Exception as ex:                      # noqa: BLE001
            health.append(str(ex)[:60])
    if rank == 0:
        print(json.dumps({'resident_health': health, 'model': a.model, 'pairs': a.pairs, 'n_gpus': world, 'kpts': a.kpts, 'workers_per_gpu': a.workers, 'pose': a.pose, 'weights': a.weights, 'pairs_per_s': a.pairs / dt,
                          'includes': 'H2D upload of every pair on the host path of each rank (pairs pre-generated)',
                          'report': eval_loop.aggregate(table)}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
