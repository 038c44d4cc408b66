#!/usr/bin/env python3
"""Benchmark of the IMP matching hot path on MI355X (contract: one JSON line on rank 0).

Workload (BASELINE.json metric / configs[2], SURVEY.md §8d "C3"): GM one-shot matcher, 9 self+cross
iterations, 100 Sinkhorn iterations, only_last, N = M = 2048 synthetic SuperPoint-like keypoints per
image, 4 pairs per GPU (batch 32 over 8 GPUs); weak scaling: every rank processes its own 4 pairs per
step, no data-path collective, one RCCL all-gather of the per-pair results at the end of the step.
A "step" = produce_matches over the rank's batch with inputs already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--kpts 2048] [--pairs-per-gpu 4]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Extra objects on the JSON line:
  roofline      dominant kernel = the flash-attention kernel (60 % of the pair's FLOPs; attn_f16x3_pp_kernel<64, false> in the default
                split-half f16x3 arithmetic, attn_f32_kernel<64, 4> with --precision f32): algorithmic FLOPs per launch
                (4*N*M*D per image side) / average launch duration measured with HIP events on the launch
                stream; peak = 2500 / 3 TFLOP/s (dense f16 MFMA, three executed products per algorithmic one) or the
                fp32-input MFMA 157.3 TFLOP/s (MI355X_MICROARCH.md); traffic = HBM bytes per launch from the newest
                committed PMC pass under profiles/ that holds this kernel at this launch geometry
  cpu_baseline  the oracle (torch-CPU fp32 restatement of the reference, validated against it) timed on this
                host's cores on a bounded sample of the same workload (rank 0, N=1 only)
  c2_* / eimp_* / superpoint_* / c5_*   the other BASELINE configurations on the same GPU (rank 0, N=1 only; not the metric)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: f16/bf16 MFMA, dense (the 5 PF headline is 2:1 sparse)
PEAK_HBM_GBS = 8000.0


def eval_config(n_layers, sinkhorn_iterations):
    return {'descriptor_dim': 256, 'sinkhorn_iterations': sinkhorn_iterations, 'match_threshold': 0.2,
            'with_sinkhorn': True, 'n_layers': n_layers, 'GNN_layers': ['self', 'cross'] * n_layers,
            'ac_fn': 'relu', 'norm_fn': 'in', 'n_min_tokens': 256}


def _cpu_worker(kpts, iters, sinkhorn, budget_s):
    """runs in a FRESH process whose OMP_NUM_THREADS was fixed before torch was imported"""
    from imp_release_amd import synthetic
    from oracle import imp_oracle as orc          # oracle/ = checker + CPU baseline only, never the product path
    cfg = eval_config(iters, sinkhorn)
    sd = synthetic.make_state_dict(cfg, 'GM', seed=0)
    o = orc.MatcherOracle(cfg, sd, 'GM')
    pair = synthetic.make_correlated_pair(kpts, kpts, seed=1000)
    data = {k: torch.from_numpy(v) for k, v in pair.items() if k != 'image_shape'}
    data['image0'] = data['image1'] = torch.zeros(pair['image_shape'])
    with torch.no_grad():
        t0 = time.perf_counter()
        o.produce_matches(data, p=0.2, only_last=True)          # warm-up pair (also bounds the sample)
        first = time.perf_counter() - t0
        n, t0 = 0, time.perf_counter()
        while True:
            o.produce_matches(data, p=0.2, only_last=True)
            n += 1
            el = time.perf_counter() - t0
            if el + first > budget_s or n >= 16:
                break
    print(json.dumps({'pairs': n, 'seconds': el, 'threads': torch.get_num_threads()}), flush=True)


def cpu_baseline(args, quick=False):
    """Times the oracle (torch-CPU fp32 restatement of the reference) on this host's cores, at the benchmark's own size.
    torch-CPU does not scale to every core on this op mix, so a few thread counts (16 ... 128) are probed on ONE pair of
    the real size (N = --kpts) each, in a fresh subprocess with a hard timeout, and the fastest is used for the bounded
    sample (~15 s) that is reported."""
    import subprocess
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    cpu_model = 'unknown'
    try:
        with open('/proc/cpuinfo') as f:
            for ln in f:
                if ln.startswith('model name'):
                    cpu_model = ln.split(':', 1)[1].strip()
                    break
    except OSError:
        pass

    def run(threads, kpts, budget, timeout):
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES='')
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-worker', str(kpts), str(args.iters),
                                str(args.sinkhorn), str(budget)], env=env, capture_output=True, text=True, timeout=timeout)
            return json.loads(r.stdout.strip().splitlines()[-1])
        except Exception:
            return None

    # (quick: multi-GPU runs - rank 0 only, the other ranks wait at the final barrier - probe one thread count and sample for 8 s)
    probes = {}
    for t in sorted({min(ncpu, c) for c in ((32,) if quick else (16, 32, 64, 128))}):
        r = run(t, args.kpts, 0.5, 40)          # 1 warm-up pair + 1 timed pair of the real size
        if r:
            probes[t] = r['pairs'] / r['seconds']
    if not probes:
        return None
    best = max(probes, key=probes.get)
    r = run(best, args.kpts, 8.0 if quick else 15.0, 90)
    if not r:
        return None
    return {'value': r['pairs'] / r['seconds'], 'unit': 'image-pairs/s', 'cores': r['threads'], 'kind': 'port',
            'cpu_model': cpu_model, 'host_cpus': ncpu,
            'sample': f"{r['pairs']} pair(s) after 1 warm-up, N={args.kpts}, L={args.iters}, T={args.sinkhorn}, "
                      f"oracle/imp_oracle.py under torch {torch.__version__} CPU fp32, {r['threads']} threads (fastest of "
                      f"{ {k: round(v, 3) for k, v in probes.items()} } pairs/s probed at N={args.kpts}), host: {ncpu} x {cpu_model}"}


def postmortems(models):
    """the post-mortem record of the last voided waiting launch of every context that has one (include/imp_hip.h imp_resident_postmortem: which wait of which
    workgroup gave up, on which CU, what it had seen, what the host knew) - the evidence for a time-out whose cause is not known"""
    out = []
    for m in models:
        try:
            pm = m._ensure_ctx().resident_postmortem()
            if pm:
                out.append(pm)
        except Exception:                                # noqa: BLE001
            pass
    return out


def voided_launches(models):
    """launches of the waiting kernels (chip-resident Sinkhorn, fused layer MLP) that timed out on these models' contexts so far (the library voids such a
    call and the loops re-run it: include/imp_hip.h IMP_E_RESIDENT)"""
    total = 0
    for m in models:
        try:
            ctx = m._ensure_ctx()
            r = ctx.resident_health(False)
            if r is False:                               # a fresh one: the look recovered the context
                r = ctx.resident_health(False)
            total += int(r[0]) if r else 0
        except Exception:                                # noqa: BLE001
            pass
    return total


def batch1_latencies(dev, args, out=None):
    """fills (and returns) `out`: the keys measured before a failure survive it"""
    import imp_release_amd as P
    from imp_release_amd import matching, synthetic

    def model_of(name, cfg, **kw):
        sd = synthetic.make_state_dict(cfg, name, seed=0, **kw)
        m = getattr(P, name)(dict(cfg, precision=args.precision)).eval()
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        return m.to(dev)

    def data_of(n0, n1, seed):
        pair = synthetic.make_correlated_pair(n0, n1, seed=seed)
        d = {k: torch.from_numpy(v).to(dev) for k, v in pair.items() if k != 'image_shape'}
        d['image0'] = d['image1'] = torch.zeros(pair['image_shape'], device=dev)
        return d

    def timeit(fn, reps, warm):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    out = {} if out is None else out
    with torch.no_grad():
        m = model_of('GM', eval_config(9, 100))
        d = data_of(1024, 1024, 5)
        out['c2_latency_ms'] = timeit(lambda: m.produce_matches(d, p=0.2, only_last=True), 20, 3)
        # the same pair with the fused call captured once in a hipGraph and replayed (the capture keeps the chip-resident Sinkhorn)
        try:
            ctx = m._ensure_ctx()
            margs = (d['keypoints0'], d['scores0'], d['descriptors0'], d['keypoints1'], d['scores1'], d['descriptors1'],
                     float(d['image0'].shape[-1]), float(d['image0'].shape[-2]), float(m._bin(None)), 100, True, 0.2)
            gout = ctx.match_pair(*margs)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                ctx.match_pair(*margs, out=gout)
            out['c2_latency_graph_ms'] = timeit(graph.replay, 20, 3)
            del graph
        except Exception as ex:                       # noqa: BLE001 - the key is optional
            out['c2_latency_graph_ms'] = None
            out['c2_latency_graph_error'] = repr(ex)[:200]
        del m
        cfg = eval_config(15, 20)
        m = model_of('AdaGMN', cfg, bin_score=5.0)
        d = data_of(4096, 4000, 41)
        d['pts0_cpu'] = d['keypoints0'][0].cpu().numpy(); d['pts1_cpu'] = d['keypoints1'][0].cpu().numpy()
        d['K0'] = d['K1'] = np.eye(3); d['T_0to1'] = np.eye(4)
        tr = []
        out['eimp_n4096_ms_per_pair'] = timeit(
            lambda: matching.matching_iterative_uncertainty(d, m, 15, 0.1, 25, 1.0, {'pose': 1.5}, trace=tr), 4, 1)
        out['eimp_n4096_trajectory'] = [(t['n0'], t['n1']) for t in tr[-7:]]
        # AdaGMN.produce_matches, the MASKED adaptive pooling of nets/adgm.py:327-526, at N = 1024, batch 4 (never timed before):
        # since round 5 the four pairs' kept keypoints are scored as one ragged batch per iteration
        try:
            mm_ = model_of('AdaGMN', eval_config(15, 20), bin_score=5.0)
            pr_ = synthetic.make_correlated_pair(1024, 1000, seed=61, batch=4)
            dm_ = {k: torch.from_numpy(v).to(dev) for k, v in pr_.items() if k != 'image_shape'}
            dm_['image0'] = dm_['image1'] = torch.zeros(pr_['image_shape'], device=dev)
            out['adagmn_masked_n1024_b4_ms_per_call'] = timeit(lambda: mm_.produce_matches(dm_, p=0.2), 5, 2)
            del mm_
        except Exception as ex:                       # noqa: BLE001 - the key is optional
            out['adagmn_masked_n1024_b4_ms_per_call'] = None
            out['adagmn_masked_error'] = repr(ex)[:200]
        # the SuperPoint front-end (image -> keypoints + descriptors) that feeds the matcher, and the chain image
        # pair -> 2 x SuperPoint -> GM (configs[1] shape: top-1024 keypoints per image), everything on the GPU
        from imp_release_amd.superpoint import SuperPoint
        sp = SuperPoint({'state_dict': synthetic.make_superpoint_state_dict(seed=1), 'max_keypoints': 1024}, device=dev)
        imgs = [torch.from_numpy(synthetic.make_image(480, 640, seed=s)).to(dev) for s in (5, 6)]
        out['superpoint_480x640_ms_per_image'] = timeit(lambda: sp({'image': imgs[0]}), 30, 3)
        m = model_of('GM', eval_config(9, 100))

        both = torch.cat(imgs)

        def chain():
            dd = {}
            o = sp({'image': both})                      # the two images of the pair in one front-end call
            for i in (0, 1):
                dd[f'keypoints{i}'] = o['keypoints'][i][None]
                dd[f'scores{i}'] = o['scores'][i][None]
                dd[f'descriptors{i}'] = o['descriptors'][i].t()[None].contiguous()
                dd[f'image{i}'] = both[i:i + 1]
            return m.produce_matches(dd, p=0.2, only_last=True)
        out['image_pair_to_matches_ms'] = timeit(chain, 20, 3)
        # BASELINE configs[4]: the full iterative pose loop (eval/eval_imp.py --use_iterative) with the pose step ON THE GPU in
        # its estimate_pose slot and the metrics tail (AUC@5/10/20, precision) - IMP and EIMP, >= 1000 pair evaluations each,
        # 3 pairs in flight.  Pairs are two-view consistent synthetic scenes (64 distinct ones, cycled; uploaded per evaluation)
        from imp_release_amd import eval_loop, pose as gpose
        try:      # one call of the pose step (host arrays in and out) on the planted correspondences of a two-view pair, 1024 five-point samples
            from imp_release_amd.synthetic import make_two_view_pair as _tv
            _p = _tv(1800, 1800, seed=77)
            _tm = np.asarray(_p['true_matches'])                                                   # [M, 2]: the planted correspondences
            _k0, _k1 = _p['keypoints0'][0][_tm[:, 0]], _p['keypoints1'][0][_tm[:, 1]]
            out['pose_step_matches'] = int(_k0.shape[0])
            out['pose_step_ms_per_call'] = timeit(lambda: gpose.estimate_pose(kpts0=_k0, kpts1=_k1, K0=_p['K0'], K1=_p['K1'], norm_thresh=1.0), 30, 3)
        except Exception as ex:                       # noqa: BLE001 - the key is optional
            out['pose_step_ms_per_call'] = None
            out['pose_step_error'] = repr(ex)[:200]
        # round 4: a WORKLOAD, not a best case - 4000 evaluations (the size of YFCC-4000, configs/yfcc_eval_gm.yaml:20) over 128
        # distinct scenes with N ~ U(1000, 2048) keypoints per image, overlap ~ U(0.2, 0.8), pixel noise ~ U(0.5, 2), 30-70 % look-alike
        # outliers among the planted correspondences (synthetic.make_hard_two_view_pair); every timed section runs twice (spread)
        n_eval, n_distinct = (4000, 128) if not args.quick_c5 else (96, 24)
        host_pairs = [synthetic.make_hard_two_view_pair(seed=7000 + i) for i in range(n_distinct)]
        UP = ('keypoints0', 'keypoints1', 'scores0', 'scores1', 'descriptors0', 'descriptors1')
        pinned = [{k: torch.from_numpy(pr[k]).pin_memory() for k in UP} for pr in host_pairs]      # a loader's pinned staging buffers

        def provider(pid):
            pr = host_pairs[pid % n_distinct]
            dd = {k: pinned[pid % n_distinct][k].to(dev, non_blocking=True) for k in UP}           # six async uploads on the worker's stream
            dd['image0'] = dd['image1'] = torch.empty(pr['image_shape'], device='meta')
            dd['pts0_cpu'], dd['pts1_cpu'] = pr['keypoints0'][0], pr['keypoints1'][0]
            dd.update({k: pr[k] for k in ('K0', 'K1', 'T_0to1', 'E')})
            return dd

        del m, sp
        for tag, name in (('imp', 'DGNNS'), ('eimp', 'AdaGMN')):
            mm = model_of(name, eval_config(15, 20), bin_score=synthetic.MATCHING_BIN_SCORE, style='matching')
            # 4 pairs advance in lock step as one ragged batch (the native drivers imp_loop_lockstep / imp_loop_lockstep_uncertainty: one launch
            # per layer for all of them, per-pair early exit; EIMP: per-pair pooling inside the batch), 3 such groups in flight
            # 4 groups in flight; the pairs of every window of n_distinct consecutive evaluations are grouped by size (a loader's look-ahead;
            # within a window no scene repeats)
            kw = dict(eimp=name == 'AdaGMN', estimate_pose=gpose.estimate_pose, workers=4, lockstep=4, group_similar=n_distinct,
                      pair_cost=lambda pid: host_pairs[pid % n_distinct]['keypoints0'].shape[1] * host_pairs[pid % n_distinct]['keypoints1'].shape[1])
            reps = eval_loop.replicate(mm, kw['workers'])
            kw['replicas'] = reps
            # warm-up: every distinct scene once, so that each replica's workspaces, LDS grants and pinned buffers have seen the largest sizes before the
            # clock starts (rounds 4-5 warmed up on 24 pairs and the first timed run came out 8 % under the second: VERDICT r5 #7)
            eval_loop.run_pairs_sharded(mm, provider, n_distinct, **kw)
            rates = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                table = eval_loop.run_pairs_sharded(mm, provider, n_eval, **kw)
                torch.cuda.synchronize()
                rates.append(n_eval / (time.perf_counter() - t0))
            out[f'c5_{tag}_pairs_per_s'] = sorted(rates)[1]                                         # median of three timed runs
            out[f'c5_{tag}_runs_pairs_per_s'] = rates
            out[f'c5_{tag}_spread'] = (max(rates) - min(rates)) / sorted(rates)[1]
            rep = eval_loop.aggregate(table)
            nit = table[:, eval_loop.SUMMARY_COLUMNS.index('n_iterations')].astype(int)
            rep['n_iterations_histogram'] = {int(k_): int(v_) for k_, v_ in zip(*np.unique(nit, return_counts=True))}
            out[f'c5_{tag}_report'] = rep
            # the round-3 schedule (single pairs, 3 in flight) on the same set, for the record
            kw3 = dict(eimp=name == 'AdaGMN', estimate_pose=gpose.estimate_pose, workers=3, replicas=reps[:3])
            eval_loop.run_pairs_sharded(mm, provider, 12, **kw3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            t3 = eval_loop.run_pairs_sharded(mm, provider, n_eval // 4, **kw3)
            torch.cuda.synchronize()
            out[f'c5_{tag}_single_pairs_3_in_flight_pairs_per_s'] = (n_eval // 4) / (time.perf_counter() - t0)
            out[f'c5_{tag}_single_pairs_auc5'] = eval_loop.aggregate(t3)['auc@5']
            out[f'c5_{tag}_voided_launches'] = voided_launches(reps)        # time-outs of waiting kernels that the loops met and re-ran (0 = none)
            if out[f'c5_{tag}_voided_launches']:
                out[f'c5_{tag}_postmortems'] = postmortems(reps)
            del mm, reps
        out['c5_note'] = (f'BASELINE configs[4] on ONE GPU: {n_eval} evaluations of matching_iterative (imp: DGNNS) / matching_iterative_uncertainty (eimp: AdaGMN, adaptive '
                          f'pooling, with_uncertainty as eval/eval_imp.py:95-105), 4 pairs in lock step as one ragged batch (imp_loop_lockstep / imp_loop_lockstep_uncertainty), '
                          f'4 groups in flight, groups formed by size inside windows of {n_distinct} consecutive evaluations; *_single_pairs_3_in_flight_* = one pair per call, 3 in flight ({n_eval // 4} evaluations); over {n_distinct} '
                          'distinct two-view synthetic scenes (N ~ U(1000, 2048) keypoints per image, overlap 0.2-0.8, pixel noise 0.5-2, 30-70 % look-alike '
                          'outliers, known relative pose), 15 iterations, early exit on pose convergence, pose step = csrc/pose.hip in the estimate_pose slot (NOT '
                          "OpenCV MAGSAC), H2D upload of every pair included; report = eval/eval_imp.py:213-227's numbers with the "
                          "synthetic 'matching' weights (synthetic.make_state_dict style='matching': a hand-built matcher that works on these pairs, "
                          'NOT a trained model - the numbers describe the pipeline)')
    out['batch1_note'] = ('c2 = BASELINE configs[1] (GM, N=1024, 9 iterations, 100 Sinkhorn, batch 1, one call after the other; c2_latency_graph_ms: the fused call '
                          'imp_match_pair captured in a hipGraph and replayed); '
                          'eimp = configs[3] (AdaGMN sliced loop from N=4096/4000, 15 iterations, 7 score+pool steps, bin_score 5, pose stubbed); '
                          'superpoint = nets/superpoint.py forward on one 480x640 image, top-1024, seeded random weights; image_pair_to_matches = '
                          'SuperPoint on both images (one call, batch 2) + GM (L=9, T=100) on the 1024 + 1024 keypoints it returns, batch 1')
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--cpu-worker':
        _cpu_worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # defaults: 40 timed steps after 10 untimed ones (0.2 s of GPU time).  A 3-step warm-up (12 ms) ends while the clocks are still settling and
    # a 20-step region (80 ms) spreads +-4 % from run to run on one box (profiles/r03/README.md)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--kpts', type=int, default=2048)
    ap.add_argument('--pairs-per-gpu', type=int, default=4)
    ap.add_argument('--iters', type=int, default=9)
    ap.add_argument('--sinkhorn', type=int, default=100)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-f32-mode', action='store_true', help="skip the value_f32_mode leg (the same steps with precision='f32')")
    ap.add_argument('--quick-c5', action='store_true', help='configs[4] keys on 48 instead of 1000 pair evaluations (smoke runs)')
    ap.add_argument('--no-batch1', action='store_true', help='skip the extra batch-1 latency keys (profiling runs: keeps their kernels out of the trace)')
    ap.add_argument('--in-flight', type=int, default=0,
                    help='batch-steps in flight per GPU (model replicas, one stream + host thread each; the result '
                         'exchange stays one ordered lane). 1 = strictly one step after the other; 0 (default) = the fastest of 1, 2 and 3, '
                         'decided by a short untimed calibration (round 4: with one step in flight every layer is ONE fused launch, with '
                         'several the two-launch layers interleave - which wins depends on the box)')
    ap.add_argument('--exchange-every', type=int, default=0,
                    help='steps per result exchange (one RCCL all-gather of that many steps: pipeline.StepPipeline exchange_every). 0 (default) = '
                         '1 on one GPU, 8 on several: a collective couples the ranks - and its kernel holds CUs beside launches that need the whole '
                         'chip while it waits for a late peer - so the multi-GPU job pays that once per 8 steps, not at every 3.7-ms step')
    ap.add_argument('--h2d', action='store_true',
                    help='additionally time the same steps with every batch uploaded from pinned host memory inside the '
                         'step (PCIe-inclusive rate, reported as value_with_h2d; never the headline value)')
    ap.add_argument('--sinkhorn-storage', type=int, choices=[3, 4], default=4,
                    help='bytes per matrix element the Sinkhorn iterations stream (4 = fp32, default; 3 = opt-in 3-byte copy)')
    ap.add_argument('--precision', choices=['f16x3', 'f32'], default='f16x3',
                    help='matrix arithmetic: split-half f16 x3 MFMA (fp32-level results, default) or native fp32 MFMA')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the matching hot path has no CPU fallback')
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N` (how the driver's scaling sweep may call it): launch ourselves as one process
        # per GPU under torch.distributed.run and hand the ranks the same command line
        if torch.cuda.device_count() < args.gpus:
            raise SystemExit(f'--gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) are visible')
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(('127.0.0.1', 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        raise SystemExit(subprocess.run(cmd, env=env).returncode)
    if world != args.gpus:
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    numa_cpus = None
    if world > 1:
        # one process per GPU: this rank's host threads (step workers, exchange lane, pose workers, prefetcher) next to its GPU's NUMA node
        from imp_release_amd import dist as _pd
        numa_cpus = _pd.pin_to_gpu_numa(local_rank)
    import torch.distributed as dist
    use_pg = world > 1 or bool(os.environ.get('IMP_FORCE_COLLECTIVES'))     # the latter: one-rank check of the RCCL path
    if use_pg:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    import imp_release_amd as P
    from imp_release_amd import dist as pdist, eval_loop, pipeline, synthetic

    cfg = eval_config(args.iters, args.sinkhorn)
    sd = synthetic.make_state_dict(cfg, 'GM', seed=0)
    model = P.GM(dict(cfg, precision=args.precision, sinkhorn_storage=args.sinkhorn_storage)).eval()
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    model = model.to(dev)

    B, N = args.pairs_per_gpu, args.kpts
    n_total = B * world
    s, e = pdist.shard_range(n_total, rank, world)           # this rank's pairs (seed = base + pair id)
    pairs = [synthetic.make_correlated_pair(N, N, seed=100 + pid) for pid in range(s, e)]
    data = {k: torch.from_numpy(np.concatenate([p[k] for p in pairs], 0)).to(dev)
            for k in pairs[0] if k != 'image_shape'}
    data['image0'] = data['image1'] = torch.zeros(pairs[0]['image_shape'], device=dev)

    # a step = one pass of the matcher over this rank's batch + the exchange of its results.  --steps of them are
    # timed; up to --in-flight overlap on this GPU (independent batch-steps on replicas of the model, one stream and
    # host thread each, ONE ordered exchange lane: pipeline.StepPipeline)
    def make_step(m):
        def step_fn():
            out = m.produce_matches(data, p=0.2, only_last=True)
            return out['indices0'][-1], out['mscores0'][-1]
        return step_fn

    xk = args.exchange_every if args.exchange_every > 0 else (1 if world == 1 else 8)
    calibration = None
    calibration_error = None
    if args.in_flight <= 0:
        # untimed calibration: the same steps with 1, 2 and 3 in flight, the fastest setting is the one that gets timed (round 5: 2 added - with the
        # in-call range recovery every call waits for its own work, and two streams hide that wait with less interference than three: 1108-1121 vs 1101)
        # (round 5: the arms used to be timed once each, 12 steps, in the order 1, 2, 3 right after the model was built - the first arm ran on an idle chip at boost clock and
        # was picked in about one run of ten, which then timed 1 020 instead of 1 110 pairs/s (profiles/r05/envab_hostknobs.log).  Now: the chip is brought to its sustained
        # state first, and every arm is timed in two interleaved rounds)
        cal_steps = 16
        try:
            pipes_, reps_by_k = {}, {}
            for k_ in (1, 2, 3):
                reps_ = [model] if k_ == 1 else eval_loop.replicate(model, k_)
                reps_by_k[k_] = reps_
                pipes_[k_] = pipeline.StepPipeline([make_step(m) for m in reps_], n_total, device=dev, exchange_every=xk)
                pipes_[k_].run(max(4, k_))                # every replica sizes its workspace
            pipes_[3].run(40)                             # ~0.15 s of the real load: clocks and power settle
            torch.cuda.synchronize()
            spent = {1: 0.0, 2: 0.0, 3: 0.0}
            for _round in range(2):
                for k_ in (1, 2, 3):
                    pipes_[k_].run(2)
                    torch.cuda.synchronize()
                    t0_ = time.perf_counter()
                    pipes_[k_].run(cal_steps)
                    torch.cuda.synchronize()
                    spent[k_] += time.perf_counter() - t0_
            rates = {k_: 2 * cal_steps / spent[k_] for k_ in spent}
        except Exception as e_:                           # noqa: BLE001 - (one GPU) a voided waiting launch during the untimed calibration: take the usual winner on a fresh model
            if world > 1:
                raise
            pipes_, reps_by_k = None, None
            torch.cuda.synchronize()
            model = eval_loop.replicate(model, 2)[1]
            rates = {1: 0.0, 2: 0.0, 3: 1.0}
            calibration_error = repr(e_)[:200]
        best_ = torch.tensor([rates[1], rates[2], rates[3]], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(best_, op=dist.ReduceOp.MIN)        # every rank must take the same setting
        args.in_flight = 1 + int(torch.argmax(best_).item())
        calibration = {'steps_per_s_1_in_flight': rates[1], 'steps_per_s_2_in_flight': rates[2], 'steps_per_s_3_in_flight': rates[3], 'chosen': args.in_flight}
        if calibration_error:
            calibration['error'] = calibration_error
    inflight = max(1, args.in_flight)
    if calibration is not None and not calibration_error and pipes_ is not None:
        # the timed steps run on the calibrated pipeline itself (its replicas have their workspaces, streams and kernel-choice history from ~70 untimed steps: the driver's
        # `--warmup 5` on FRESH replicas timed ~3 % below the rate the calibration had just measured); the other arms are dropped
        replicas, pipe = reps_by_k[inflight], pipes_[inflight]
        pipes_, reps_by_k = None, None
    else:
        replicas = [model] if inflight == 1 else eval_loop.replicate(model, inflight)
        pipe = pipeline.StepPipeline([make_step(m) for m in replicas], n_total, device=dev, exchange_every=xk)

    def fence():
        torch.cuda.synchronize()
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(p_, steps):
        fence()
        t0_ = time.perf_counter()
        r_ = p_.run(steps)
        fence()
        return r_, time.perf_counter() - t0_

    n_warm = max(args.warmup, inflight)              # every replica sizes its workspace before the clock starts
    # A chip-resident launch that times out voids its call, the library says so at the context's next entry point (IMP_E_RESIDENT) and the documented answer
    # is "re-run the batch".  Seen once in a full run of this file (profiles/r05/MEASURED.md): the measurement then starts over on fresh replicas (one GPU only:
    # a rank that repeats its steps alone would leave the others' collectives unmatched) and the line says how often
    from imp_release_amd import _lib as _plib
    headline_retries = 0
    while True:
        try:
            pipe.run(n_warm)
            res, dt = timed(pipe, args.steps)
            break
        except _plib.ResidentSinkhornTimeout:
            if world > 1 or headline_retries >= 2:
                raise
            headline_retries += 1
            torch.cuda.synchronize()
            model = eval_loop.replicate(model, 2)[1]                 # a fresh context with the same weights
            replicas = [model] if inflight == 1 else eval_loop.replicate(model, inflight)
            pipe = pipeline.StepPipeline([make_step(m) for m in replicas], n_total, device=dev, exchange_every=xk)
    # the same number of steps strictly one after the other (one replica), reported next to the headline
    serial_s = None
    serial_error = None
    try:
        pipe1 = pipeline.StepPipeline([make_step(model)], n_total, device=dev, exchange_every=xk)
        pipe1.run(1)
        _, dt1 = timed(pipe1, args.steps)
        serial_s = torch.tensor([dt1], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(serial_s, op=dist.ReduceOp.MAX)
        serial_s = float(serial_s.item())
    except _plib.ResidentSinkhornTimeout as e_:      # (one GPU: reported, never fatal for the headline; several: the ranks' collectives no longer match - give up)
        if world > 1:
            raise
        serial_s, serial_error = None, repr(e_)[:200]
        model = eval_loop.replicate(model, 2)[1]
    # several ranks: the headline exchanges results once per xk steps (default 8) - the rate with ONE exchange per step beside it (the two
    # are not the same measurement; round 4's multi-GPU lines were taken at 1)
    value_x1 = None
    if xk != 1 and (world > 1 or os.environ.get('IMP_FORCE_COLLECTIVES')):
        pipe_x1 = pipeline.StepPipeline([make_step(m) for m in replicas], n_total, device=dev, exchange_every=1)
        pipe_x1.run(max(2, n_warm // 2))
        _, dtx = timed(pipe_x1, args.steps)
        tx = torch.tensor([dtx], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tx, op=dist.ReduceOp.MAX)
        value_x1 = n_total * args.steps / float(tx.item())
        del pipe_x1
    h2d_s = None
    if args.h2d:
        host = {k: v.cpu().pin_memory() for k, v in data.items() if k not in ('image0', 'image1')}

        def make_h2d_step(m):
            dbuf = {k: torch.empty_like(v, device=dev) for k, v in host.items()}
            dbuf['image0'] = dbuf['image1'] = data['image0']

            def step_fn():
                for k, v in host.items():
                    dbuf[k].copy_(v, non_blocking=True)         # on the worker's stream, ahead of the kernels that read it
                out = m.produce_matches(dbuf, p=0.2, only_last=True)
                return out['indices0'][-1], out['mscores0'][-1]
            return step_fn

        pipe_h = pipeline.StepPipeline([make_h2d_step(m) for m in replicas], n_total, device=dev, exchange_every=xk)
        pipe_h.run(inflight)
        _, h2d_s = timed(pipe_h, args.steps)
    elapsed = torch.tensor([dt], dtype=torch.float64, device=dev)
    per_rank_s = [dt]
    if use_pg:               # every rank's own clock around the same timed region (the headline takes the MAX): a slow GPU / rank shows here
        allt = torch.empty(world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allt, elapsed)
        per_rank_s = allt.tolist()
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    ranks_seen = [0]
    if use_pg:               # which ranks really took part (from an all-gather, not from the environment)
        mine = torch.tensor([rank], dtype=torch.int64, device=dev)
        seen = torch.empty(world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(seen, mine)
        ranks_seen = seen.tolist()
    assert res[0].shape[0] == n_total
    n_matched = int((res[0] >= 0).sum())

    # the precision-strict number: the same steps, one in flight, with EVERY matrix product on the native fp32-input MFMA
    # (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains, the reference's arithmetic) instead of the split-half f16x3 scheme
    f32_mode = None
    if args.precision != 'f32' and not args.no_f32_mode:
        try:
            m32 = P.GM(dict(cfg, precision='f32', sinkhorn_storage=args.sinkhorn_storage)).eval()
            m32.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
            m32 = m32.to(dev)
            p32 = pipeline.StepPipeline([make_step(m32)], n_total, device=dev, exchange_every=xk)
            p32.run(2)
            k32 = max(4, args.steps // 2)
            _, dt32 = timed(p32, k32)
            t32 = torch.tensor([dt32], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(t32, op=dist.ReduceOp.MAX)
            f32_mode = {'value': n_total * k32 / float(t32.item()), 'ms_per_step': float(t32.item()) / k32 * 1e3, 'steps': k32,
                        'note': "precision='f32': native fp32-input MFMA for every product (157.3 TFLOP/s roof), one step in flight; "
                                'the same fixtures are green in this mode (tests/test_gpu_parity.py, both precisions)'}
            del m32, p32
        except Exception as e_:       # noqa: BLE001 - reported, never fatal for the headline
            f32_mode = {'error': repr(e_)[:200]}

    # roofline leg of the dominant kernel, measured live with HIP events on the launch stream
    ctx = model._ensure_ctx()
    try:
        attn_ms, sclk_mhz = ctx.time_attention_clock(B, N, 10)
    except Exception:                 # noqa: BLE001
        attn_ms, sclk_mhz = ctx.time_attention(B, N, 10), 0.0
    attn_flops = 4.0 * N * N * 256 * 2 * B            # 4*N*M*D per image side (QK^T + PV), 2 sides, B pairs
    achieved = attn_flops / (attn_ms * 1e-3) / 1e12
    f16x3 = ctx.precision == 'f16x3'
    # f16x3 executes 3 f16 MFMA flops per algorithmic (fp32-equivalent) flop: the roof for ALGORITHMIC flops is the
    # dense f16 peak / 3; the native fp32-MFMA roof (157.3) is what the same math costs without the split
    peak = PEAK_F16_MFMA_TFLOPS / 3.0 if f16x3 else PEAK_F32_MFMA_TFLOPS
    try:
        sk_ms = ctx.time_sinkhorn(B, N, 50)         # per Sinkhorn ITERATION, on the path the product takes for this shape
        sk_resident = ctx.resident_status()[1]
    except Exception:                               # noqa: BLE001 - a secondary leg never costs the line
        ctx = eval_loop.replicate(model, 2)[1]._ensure_ctx()
        sk_ms = ctx.time_sinkhorn(B, N, 50)
        sk_resident = ctx.resident_status()[1]
    ld = (N + 1 + 3) // 4 * 4
    # bytes ONE pass over the matrix moves (what an iteration costs when P is streamed: the streaming path reads P once per
    # iteration + the column partial vectors; the chip-resident kernel keeps P in registers and moves only vectors)
    sk_bytes = B * (N + 1) * ld * 4.0 + 2.0 * B * ((N + 1 + 15) // 16) * ld * 4.0
    layer_sides = 4 * args.iters
    pair_flops = (4 * args.iters * (20 * 256 ** 2 * N + 4 * N * N * 256) + 2 * 2 * N * 108640 + 4 * 256 ** 2 * N
                  + 2 * N * N * 256 + args.sinkhorn * 4 * (N + 1) ** 2)

    # HBM traffic of the dominant kernel: measured offline with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in their own
    # passes (tools/gpu_pmc.sh; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction) and committed under
    # profiles/; reported only when the profiled kernel and launch geometry are the ones timed here
    traffic = None
    # the f16x3 build takes the phase-staggered 8-wave kernel (256 queries per workgroup) at this size
    kname = 'attn_f16x3_pp_kernel<64, false>' if f16x3 else 'attn_f32_kernel<64, 4>'
    kgrid = -(-N // 256) * 4 * 2 * B * 512 if f16x3 else -(-N // 128) * 4 * 2 * B * 256
    import glob
    tfile = None
    for tpath in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', 'traffic*.json')), reverse=True):      # newest round first
        for k, v in json.load(open(tpath)).items():
            if traffic is None and k.startswith(kname) and k.endswith(f'grid={kgrid}'):
                traffic = v.get('fetch_bytes', v.get('fetch_bytes_corrected')) + v['write_bytes']
                tfile = os.path.relpath(tpath, ROOT)
        if traffic is not None:
            break
    if rank == 0:
        line = {
            'metric': 'image-pairs/s (N=2048 kpts, 9 self+cross iters, 100 Sinkhorn)',
            'value': n_total * args.steps / elapsed, 'unit': 'image-pairs/s', 'n_gpus': world, 'ranks_seen': ranks_seen, 'steps': args.steps,
            'warmup': n_warm, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'per_rank_ms_per_step': [t_ / args.steps * 1e3 for t_ in per_rank_s], 'value_f32_mode': f32_mode,
            'value_result_exchange_every_step': value_x1,
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32 (products as split-half f16x3 MFMA, fp32 accumulate)' if f16x3 else 'f32', 'data': 'synthetic',
            'config': {'workload': f'GM one-shot matcher (nets/gm.py produce_matches only_last): N=M={N} keypoints, '
                                   f'{args.iters} self+cross iterations, {args.sinkhorn} Sinkhorn iterations, '
                                   f'{B} pairs per GPU (BASELINE configs[2]: batch 32 over 8 GPUs), norm_fn=in, '
                                   f'seeded random weights',
                       'pairs_per_gpu': B, 'keypoints': N, 'parallelism': f'pair-sharded x{world}',
                       'steps_in_flight_per_gpu': inflight, 'steps_in_flight_calibration': calibration, 'steps_per_result_exchange': xk,
                       'rank0_numa_cpus': None if numa_cpus is None else len(numa_cpus),
                       'sinkhorn_storage_bytes': args.sinkhorn_storage,
                       'matched_keypoints': n_matched},
            'roofline': {'bound': 'mfma', 'kernel': kname,
                         'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak,
                         'traffic': traffic, 'traffic_unit': f'bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE, {tfile})',
                         'algorithmic_bytes_per_launch': (3 * 256 + 256) * 4.0 * N * 2 * B,
                         'launch_ms': attn_ms, 'flops_per_launch': attn_flops,
                         # the clock the timed launches really ran at (workgroup 0: s_memtime cycles / s_memrealtime ticks x 100 MHz, include/imp_hip.h
                         # imp_time_attention_clock): the chip holds 2400 MHz only below its power limit, and this kernel is above it
                         'sclk_mhz_observed': sclk_mhz or None, 'sclk_mhz_nominal': 2400.0,
                         'frac_vs_clock_limited_roof': (achieved / (peak * sclk_mhz / 2400.0)) if sclk_mhz else None,
                         'peak_note': ('algorithmic fp32-equivalent flops; peak = 2500 TF dense f16 MFMA / 3 products per '
                                       'flop (executed MFMA rate = 3 x achieved); native fp32-MFMA roof would be 157.3')
                         if f16x3 else 'native fp32-input MFMA, dense',
                         'vs_native_f32_mfma_roof': achieved / PEAK_F32_MFMA_TFLOPS,
                         # the same measurement under the three readings of "fraction of the MFMA roof" (VERDICT r1 #7):
                         'frac_executed_vs_f16_peak': (3.0 if f16x3 else 1.0) * achieved / (PEAK_F16_MFMA_TFLOPS if f16x3 else PEAK_F32_MFMA_TFLOPS),
                         'frac_algorithmic_vs_f16_peak': achieved / PEAK_F16_MFMA_TFLOPS,
                         'launches_per_step': layer_sides // 2,
                         'whole_path_tflops': pair_flops * n_total * args.steps / elapsed / 1e12 / world,
                         'sinkhorn_iteration': {
                             'path': 'chip-resident (P in VGPRs for all T iterations, tagged vector exchanges without barriers; a pair lives on two XCDs: column '
                                     'sums reduced inside each XCD L2, half sums swapped once across the fabric per iteration)'
                             if sk_resident else 'streaming (P read once per iteration, 2 launches)',
                             'bound': 'latency (one fabric crossing + two L2 hand-offs per iteration) and the per-iteration arithmetic' if sk_resident else 'hbm',
                             'iteration_ms': sk_ms, 'matrix_bytes': sk_bytes,
                             'matrix_bytes_per_iteration_time_GBs': sk_bytes / (sk_ms * 1e-3) / 1e9,
                             'peak_GBs': PEAK_HBM_GBS,
                             'note': 'resident: no HBM traffic per iteration; the GB/s figure is what a streaming '
                                     'implementation would need to match it' if sk_resident else ''}},
        }
        line['headline_retries_after_a_voided_resident_launch'] = headline_retries
        line['voided_launches_on_the_headline_replicas'] = voided_launches(replicas)
        if line['voided_launches_on_the_headline_replicas']:
            line['headline_postmortems'] = postmortems(replicas)
        line['one_step_in_flight'] = ({'error': serial_error} if serial_error else None) if serial_s is None else {
            'value': n_total * args.steps / serial_s, 'ms_per_step': serial_s / args.steps * 1e3,
            'note': 'same K steps strictly sequential on one model instance (no overlap between batch-steps); since round 4 every layer of '
                    'such a step is ONE fused launch (MLP0 -> InstanceNorm statistics exchange -> MLP3 -> next projection)'}
        # the layer GEMMs live (HIP events, 20 launches each): the round-3 pair of launches and the fused launch that replaced it
        try:
            tl = {w: ctx.time_layer_gemm(B, N, w, -2) * 1e3 for w in (1, 3, 4)}
            line['layer_gemm_us'] = {'mlp0': tl[1], 'mlp3_plus_projection': tl[3], 'two_launches': tl[1] + tl[3], 'fused_launch': tl[4],
                                     'note': 'gemm_wf.hip, B x N of this run, back-to-back launches of one kernel (the in-pipeline durations are in profiles/)'}
        except Exception as e_:        # (shapes the weight-fragment kernels do not take)
            line['layer_gemm_us'] = {'error': str(e_)[:200]}
        if h2d_s is not None:
            line['value_with_h2d'] = {'value': n_total * args.steps / h2d_s,
                                      'ms_per_step': h2d_s / args.steps * 1e3,
                                      'note': 'every step uploads its batch (keypoints, scores, descriptors of '
                                              'both images) from pinned host memory on the step stream'}
        if world == 1 and not args.no_batch1:
            # ragged batch (round 4): 4 pairs with N ~ U(1200, 2048) keypoints per image in ONE padded batch under per-pair counts
            # (include/imp_hip.h imp_set_counts) - what real SuperPoint output looks like; the reference runs such pairs one at a time
            try:
                rg = np.random.default_rng(5)
                sizes = [(int(rg.integers(1200, 2049)), int(rg.integers(1200, 2049))) for _ in range(B)]
                N0, N1 = max(s_[0] for s_ in sizes), max(s_[1] for s_ in sizes)
                rd = {}
                singles = [synthetic.make_correlated_pair(a_, b_, seed=300 + i_) for i_, (a_, b_) in enumerate(sizes)]
                for key, n_ in (('keypoints0', N0), ('keypoints1', N1), ('scores0', N0), ('scores1', N1), ('descriptors0', N0), ('descriptors1', N1)):
                    arr = np.zeros((B, n_) + singles[0][key].shape[2:], np.float32)
                    for i_, sg in enumerate(singles):
                        arr[i_, :sg[key].shape[1]] = sg[key][0]
                    rd[key] = torch.from_numpy(arr).to(dev)
                rd['image0'] = rd['image1'] = data['image0']
                rd['num_keypoints0'] = [s_[0] for s_ in sizes]; rd['num_keypoints1'] = [s_[1] for s_ in sizes]
                for _ in range(4):
                    model.produce_matches(rd, p=0.2, only_last=True)
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                for _ in range(30):
                    ro = model.produce_matches(rd, p=0.2, only_last=True)
                torch.cuda.synchronize()
                dtr = time.perf_counter() - t0_
                # the same pairs one call each (what a batch-1 user pays)
                sdata = []
                for sg in singles:
                    d1 = {k: torch.from_numpy(v).to(dev) for k, v in sg.items() if k != 'image_shape'}
                    d1['image0'] = d1['image1'] = data['image0']
                    sdata.append(d1)
                for d1 in sdata:
                    model.produce_matches(d1, p=0.2, only_last=True)
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                for _ in range(10):
                    for d1 in sdata:
                        model.produce_matches(d1, p=0.2, only_last=True)
                torch.cuda.synchronize()
                dts = time.perf_counter() - t0_
                line['ragged_b4_pairs_per_s'] = B * 30 / dtr
                line['ragged_b4_note'] = {'sizes': sizes, 'matched_keypoints': int((ro['indices0'][-1] >= 0).sum()),
                                          'same_pairs_one_call_each_pairs_per_s': B * 10 / dts,
                                          'note': 'GM L=9 T=100, one step in flight; pairs padded to the largest, per-pair counts by imp_set_counts'}
            except Exception as e_:
                line['ragged_b4_pairs_per_s'] = None
                line['ragged_b4_note'] = {'error': str(e_)[:300]}
            # the batch-1 configurations of BASELINE.json on the same GPU (not the metric; recorded so that every round
            # shows them): configs[1] GM N=1024 L=9 T=100 batch 1, and configs[3] the EIMP sliced loop from N=4096
            b1_ = {}
            try:
                batch1_latencies(dev, args, b1_)
            except Exception as e_:                 # noqa: BLE001 - the extra keys never cost the headline line (those measured before the failure are kept)
                b1_['batch1_error'] = repr(e_)[:300]
            line.update(b1_)
        if not args.no_cpu_baseline:       # rank 0 only (this branch), at every N: the other ranks wait at the final barrier meanwhile
            try:
                line['cpu_baseline'] = cpu_baseline(args, quick=world > 1)
            except Exception as e_:                 # noqa: BLE001
                line['cpu_baseline'] = {'error': repr(e_)[:300]}
        else:
            line['cpu_baseline'] = None
    # the line must be the LAST thing on stdout: RCCL's version banner sits in the C library's stdio buffer of the ranks (stdout is a pipe) and would otherwise
    # be flushed behind it at exit - every rank flushes, the ranks meet, then rank 0 prints
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:                                    # noqa: BLE001
        pass
    sys.stdout.flush()
    if use_pg:
        dist.barrier()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if use_pg:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
