#!/usr/bin/env python3
"""Counterpart of `python -m eval.eval_imp --matching_method IMP|EIMP --use_iterative` on a dumped pair dataset
(reference layout: components/readers.py:8-33 - the HDF5 dump itself when h5py is present, or its npz mirror written by
imp_release_amd.data.convert_h5_to_npz / write_npz_store).  Pairs are sharded over the ranks of one node, K pairs in flight
per GPU.

    python tools/eval_dataset.py --dataset /data/yfcc_sp_2000_npz --weights imp.pth --model IMP
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/eval_dataset.py --dataset ... --workers 3

The pose step of the reference (cv2 USAC_MAGSAC, eval/pose_estimation.py) is used when cv2 and the reference's
`eval.pose_estimation.estimate_pose` are importable (pass --reference-root); otherwise the loops run without early exit and
only the matching statistics are reported."""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imp_release_amd as P
from imp_release_amd import data as pdata, eval_loop, synthetic


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dataset', required=True, help='*.hdf5 dump of the reference or a directory of pair_<i>.npz files')
    ap.add_argument('--num-kpt', type=int, default=2000)
    ap.add_argument('--model', choices=['IMP', 'EIMP'], default='IMP')
    ap.add_argument('--height', type=int, required=False, default=None, help='image height of an *.hdf5 dump (it stores paths, not sizes)')
    ap.add_argument('--width', type=int, required=False, default=None)
    ap.add_argument('--weights', default=None, help="checkpoint with a 'model' state dict (eval/eval_imp.py:333); "
                                                    'seeded random weights when omitted')
    ap.add_argument('--workers', type=int, default=3)
    ap.add_argument('--pairs', type=int, default=0, help='0 = all')
    ap.add_argument('--reference-root', default=None, help='path of the reference repo (for its cv2 pose step)')
    a = ap.parse_args()
    rank, world, lr = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(lr)
    dev = torch.device('cuda', lr)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    if a.dataset.endswith(('.hdf5', '.h5')) and (a.height is None or a.width is None):
        ap.error('--height and --width are required for an hdf5 dump (keypoint normalisation depends on them)')
    store = pdata.H5PairStore(a.dataset, a.num_kpt, image_sizes=((a.height, a.width), (a.height, a.width))) if a.dataset.endswith(('.hdf5', '.h5')) else pdata.NpzPairStore(a.dataset, a.num_kpt)
    n = len(store) if a.pairs <= 0 else min(a.pairs, len(store))
    cfg = {'descriptor_dim': 256, 'sinkhorn_iterations': 20, 'match_threshold': 0.2, 'with_sinkhorn': True, 'n_layers': 15,
           'GNN_layers': ['self', 'cross'] * 15, 'ac_fn': 'relu', 'norm_fn': 'in', 'n_min_tokens': 256}
    name = 'AdaGMN' if a.model == 'EIMP' else 'DGNNS'
    m = getattr(P, name)(cfg).eval()
    if a.weights:
        m.load_state_dict(torch.load(a.weights, map_location='cpu')['model'], strict=True)
    else:
        sd = synthetic.make_state_dict(cfg, name, seed=0, bin_score=5.0)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.to(dev)
    estimate_pose = None
    if a.reference_root:
        try:
            sys.path.insert(0, a.reference_root)
            from eval.pose_estimation import estimate_pose     # cv2 USAC_MAGSAC step of the reference, unchanged
        except Exception as ex:
            print(f'[rank {rank}] pose step unavailable ({ex}); running without early exit', file=sys.stderr)
            estimate_pose = None
    reps = eval_loop.replicate(m, a.workers)
    provider = lambda pid: pdata.feed_data(store.record(pid), dev)
    kw = dict(eimp=a.model == 'EIMP', workers=a.workers, replicas=reps, estimate_pose=estimate_pose)
    eval_loop.run_pairs_sharded(m, provider, min(n, 2 * world * a.workers), **kw)            # warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    table = eval_loop.run_pairs_sharded(m, provider, n, **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({'model': a.model, 'pairs': n, 'n_gpus': world, 'workers_per_gpu': a.workers, 'pairs_per_s': n / dt,
                          'pose_step': 'reference cv2' if estimate_pose else 'none (no early exit)',
                          'report': eval_loop.aggregate(table)}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
