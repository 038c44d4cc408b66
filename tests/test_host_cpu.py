"""CPU: host logic, C-ABI surface (load + symbols only, no compute without a GPU), multi-rank sharding."""
import ctypes
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import imp_release_amd as P
from imp_release_amd import _lib, dist as pdist, synthetic
from helpers import ROOT, eval_config


def test_library_loads_and_exports_every_declared_symbol():
    L = _lib.lib()
    header = open(os.path.join(ROOT, 'include', 'imp_hip.h')).read()
    declared = set(re.findall(r'\b(imp_[a-z_0-9]+)\s*\(', header)) - {'imp_hip'}
    assert declared, 'no declarations parsed'
    for sym in sorted(declared):
        assert hasattr(L, sym), f'libimp_hip.so does not export {sym}'
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    assert b'gfx950' in L.imp_version()


def test_library_is_built_for_gfx950_only():
    out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf', '-S', _lib.LIB_PATH], capture_output=True, text=True)
    if out.returncode != 0:
        pytest.skip('llvm-readelf unavailable')
    assert '.hip_fatbin' in out.stdout
    raw = open(_lib.LIB_PATH, 'rb').read()
    assert b'gfx950' in raw and b'gfx942' not in raw and b'sm_' not in raw


@pytest.mark.parametrize('model,cfg', [('GM', dict(n_layers=9, norm_fn='in')), ('GM', dict(n_layers=2, norm_fn='bn')),
                                       ('DGNNS', dict()), ('AdaGMN', dict()),
                                       ('GM', dict(n_layers=3, descriptor_dim=128))])
def test_state_dict_schema_is_the_reference_schema(model, cfg):
    cfg = eval_config(**cfg)
    m = getattr(P, model)(cfg)
    want = synthetic.make_state_dict(cfg, model)
    got = m.state_dict()
    assert list(got.keys()) == list(want.keys()) or set(got.keys()) == set(want.keys())
    for k, v in want.items():
        assert tuple(got[k].shape) == tuple(np.asarray(v).shape), k
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in want.items()}, strict=True)
    if model != 'GM':
        assert sum(p.numel() for p in m.parameters()) == 19231809     # SURVEY.md §8b [probed]


def test_sharing_pattern_and_defaults():
    assert P.GM.default_config['norm_fn'] == 'bn' and P.GM.default_config['sinkhorn_iterations'] == 20
    d = P.DGNNS(eval_config())
    assert d.sharing_layers[:8] == [False] * 6 + [True, True] and d.sharing_layers[8:12] == [False, False, True, True]
    assert P.GM(eval_config(n_layers=2)).sharing_layers == [False] * 4
    assert P.AdaGMN(eval_config()).pool_sizes[:4] == [0, 0, 0, 0]


def test_no_cpu_fallback_and_error_behaviour():
    m = P.GM(eval_config(n_layers=1)).eval()
    d = {'descriptors0': torch.zeros(1, 4, 256), 'descriptors1': torch.zeros(1, 4, 256),
         'keypoints0': torch.zeros(1, 4, 2), 'keypoints1': torch.zeros(1, 4, 2),
         'scores0': torch.zeros(1, 4), 'scores1': torch.zeros(1, 4),
         'image0': torch.zeros(1, 3, 8, 8), 'image1': torch.zeros(1, 3, 8, 8)}
    with pytest.raises(RuntimeError, match='no CPU path'):
        m.produce_matches(d)
    e = dict(d, keypoints0=torch.zeros(1, 0, 2))
    out = m.produce_matches(e)                       # nets/gm.py:154-162: empty keypoints short-circuit
    assert out['skip_train'] and out['matches0'].shape == (0,) and out['matches1'].dtype == torch.int32
    m.train()
    with pytest.raises(NotImplementedError):
        m(d)
    with pytest.raises(RuntimeError):
        m.kenc(torch.zeros(1))                       # parameter containers never compute


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libimp_hip.so')
    with pytest.raises(_lib.HipLibraryMissing):
        _lib.lib()


def test_shard_range_partitions_everything():
    for n in (0, 1, 7, 32, 4000):
        for w in (1, 2, 3, 8):
            blocks = [pdist.shard_range(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [e - s for s, e in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_pack_roundtrip():
    i = torch.randint(-1, 2048, (3, 17), dtype=torch.int64)
    m = torch.rand(3, 17)
    a, b = pdist.unpack_matches(pdist.pack_matches(i, m), 17)
    assert torch.equal(a, i) and torch.equal(b, m)
    # one-row slices and odd keypoint counts (a [1, 12n] row keeps the 12n pitch, which is not 8-byte aligned for odd n)
    for n in (1, 33, 775):
        i = torch.randint(-1, 4096, (3, n), dtype=torch.int64)
        m = torch.rand(3, n)
        p = pdist.pack_matches(i, m)
        for r in range(3):
            a, b = pdist.unpack_matches(p[r:r + 1], n)
            assert torch.equal(a, i[r:r + 1]) and torch.equal(b, m[r:r + 1])


def test_metrics_tail_vs_reference_golden():
    from imp_release_amd import metrics
    from helpers import load_golden
    _, z = load_golden('metrics')
    assert np.allclose(metrics.pose_auc(z['errs'], z['ths'].tolist()), z['auc'], atol=1e-12)
    et, eR = metrics.compute_pose_error(z['T'], z['R'], z['t'])
    assert abs(et - float(z['err_t'])) < 1e-10 and abs(eR - float(z['err_R'])) < 1e-10
    mask, dis = metrics.compute_epi_inlier(z['x1'], z['x2'], z['E'], 0.3, return_error=True)
    assert np.array_equal(mask, z['mask']) and np.allclose(dis, z['dis'], atol=1e-12)
    assert metrics.pose_auc([1., 2., 30.], [5])[0] > 0.4


def test_eval_loop_summary_rows():
    from imp_release_amd import eval_loop
    C = eval_loop.SUMMARY_COLUMNS
    i0 = np.array([3, -1, 0, -1]); ms = np.array([0.5, 0.0, 0.3, 0.0], dtype=np.float32)
    r = dict(zip(C, eval_loop.summarize((i0, ms, None, None, 15), eimp=False)))
    assert (r['n_iterations'], r['n_matches'], r['n_kept0']) == (15, 2, -1) and abs(r['mean_mscore'] - 0.4) < 1e-6
    assert np.isnan(r['err_R']) and np.isnan(r['precision'])          # no ground truth given
    r = dict(zip(C, eval_loop.summarize((np.zeros((7, 2)), np.zeros((9, 2)), None, None, i0, ms, None, None, 12), eimp=True)))
    assert (r['n_iterations'], r['n_matches'], r['n_kept0'], r['n_kept1']) == (12, 2, 7, 9)


def test_eval_loop_metrics_tail_on_a_two_view_pair():
    """eval/eval_imp.py:112-141 through eval_loop.summarize / aggregate: precision and matching score against the ground-truth E,
    pose error of the loop's pose or - when it found none - of estimate_pose on the final matches, AUC of max(err_R, err_t)"""
    from imp_release_amd import eval_loop, metrics
    from oracle import pose_oracle
    p = synthetic.make_two_view_pair(500, 470, seed=4)
    tm = p['true_matches']
    data = {'pts0_cpu': p['keypoints0'][0], 'pts1_cpu': p['keypoints1'][0], **{k: p[k] for k in ('K0', 'K1', 'T_0to1', 'E')}}
    i0 = -np.ones(500, dtype=np.int64)
    i0[tm[:, 0]] = tm[:, 1]
    wrong = tm[:40, 0]
    i0[wrong] = (i0[wrong] + 17) % 470                           # 40 wrong matches
    ms = (i0 > -1).astype(np.float32) * 0.5
    C = eval_loop.SUMMARY_COLUMNS
    pose = lambda **k: pose_oracle.estimate_pose(iterations=512, **k)
    r = dict(zip(C, eval_loop.summarize((i0, ms, None, None, 15), False, data, pose, 1.0)))
    n = int((i0 > -1).sum())
    assert abs(r['precision'] - (n - 40) / n) < 0.02 and abs(r['matching_score'] - (n - 40) / 500) < 0.02
    assert r['err_R'] < 1.0 and r['err_t'] < 3.0                  # the pose step recovers the pose from the final matches
    # the loop's own pose wins over a fresh estimate (eval/eval_imp.py:139-140)
    R_gt, t_gt = p['T_0to1'][:, :3], p['T_0to1'][:, 3]
    r2 = dict(zip(C, eval_loop.summarize((i0, ms, R_gt, t_gt, 9), False, data, None, 1.0)))
    assert r2['err_R'] < 1e-6 and r2['err_t'] < 1e-4 and r2['n_iterations'] == 9
    # no pose at all: infinite error, counted as a miss by the AUC
    r3 = eval_loop.summarize((i0, ms, None, None, 15), False, data, None, 1.0)
    assert np.isinf(r3[0]) and np.isinf(r3[1])
    rep = eval_loop.aggregate(np.stack([np.array(list(r.values())), r3]))
    assert rep['pairs'] == 2 and rep['pose_found'] == 0.5 and 40.0 < rep['auc@20'] <= 50.0
    assert abs(rep['auc@5'] - 100 * metrics.pose_auc([max(r['err_R'], r['err_t']), np.inf], [5])[0]) < 0.01


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from imp_release_amd import dist as pdist
dist.init_process_group('gloo', rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']))
rank, world = dist.get_rank(), dist.get_world_size()
n_total, N = 5, 33
g = torch.Generator().manual_seed(0)
full_i = torch.randint(-1, N, (n_total, N), generator=g, dtype=torch.int64)
full_m = torch.rand(n_total, N, generator=g)
s, e = pdist.shard_range(n_total, rank, world)
gi, gm = pdist.all_gather_matches(full_i[s:e], full_m[s:e], n_total)
assert torch.equal(gi, full_i) and torch.equal(gm, full_m), rank
# per-pair summary rows of the sharded evaluation loop
import numpy as np
from imp_release_amd import eval_loop
table = np.arange(7 * 9, dtype=np.float64).reshape(7, 9)
s2, e2 = pdist.shard_range(7, rank, world)
got = eval_loop.gather_rows_across_ranks(table[s2:e2], 7)
assert np.array_equal(got, table), rank
# several batch-steps in flight (pipeline.StepPipeline): 3 workers with skewed speeds, ONE ordered exchange lane
import time
from imp_release_amd import pipeline
n_tot, Nk = 4, 17
def make_fn(i):
    state = {'s': i}
    def fn():
        s_ = state['s']; state['s'] += 3
        time.sleep(0.002 * ((i + rank) % 3))           # workers finish out of order, differently on each rank
        gen = torch.Generator().manual_seed(1000 + s_)
        fi = torch.randint(-1, Nk, (n_tot, Nk), generator=gen, dtype=torch.int64)
        fm = torch.rand(n_tot, Nk, generator=gen)
        a, b = pdist.shard_range(n_tot, rank, world)
        return fi[a:b], fm[a:b]
    return fn
pipe = pipeline.StepPipeline([make_fn(i) for i in range(3)], n_tot)
outs = pipe.run(10, keep=True)
assert len(outs) == 10
for s_, (gi_, gm_) in enumerate(outs):
    gen = torch.Generator().manual_seed(1000 + s_)
    fi = torch.randint(-1, Nk, (n_tot, Nk), generator=gen, dtype=torch.int64)
    fm = torch.rand(n_tot, Nk, generator=gen)
    assert torch.equal(gi_, fi) and torch.equal(gm_, fm), (rank, s_)
# ... and with ONE collective per 4 steps (exchange_every: a fast rank then waits for a slow one once per 4 steps)
pipe = pipeline.StepPipeline([make_fn(i) for i in range(3)], n_tot, exchange_every=4)
outs = pipe.run(10, keep=True)
assert len(outs) == 10
for s_, (gi_, gm_) in enumerate(outs):
    gen = torch.Generator().manual_seed(1000 + s_)
    fi = torch.randint(-1, Nk, (n_tot, Nk), generator=gen, dtype=torch.int64)
    fm = torch.rand(n_tot, Nk, generator=gen)
    assert torch.equal(gi_, fi) and torch.equal(gm_, fm), (rank, s_)
# the sharded evaluation loop under DELIBERATE imbalance (round 4, SURVEY 8e): longest-first assignment of pairs to ranks, lock-step
# groups, and the id-based gather - the table must equal the sequential one whatever the schedule
from imp_release_amd import matching
costs = [1, 50, 2, 3, 40, 1, 1, 30, 2, 1, 1]
def provider(pid):
    return {'pid': pid, 'n': 10 + pid, 'cost': costs[pid], 'keypoints0': torch.zeros(1, 10 + pid, 2), 'keypoints1': torch.zeros(1, 12, 2)}
def fake_one(data):
    n, pid = data['n'], data['pid']
    idx = np.full(n, -1, dtype=np.int64); idx[:pid % n] = np.arange(pid % n) % 12
    ms = (np.arange(n) % 5).astype(np.float32) / 5
    return idx, ms, None, None, pid % 7 + 1
def fake_loop(data, m, *a, **k):
    time.sleep(0.001 * data['cost'])
    return fake_one(data)
def fake_lockstep(datas, m, *a, **k):
    time.sleep(0.001 * max(d['cost'] for d in datas))
    return [fake_one(d) for d in datas]
matching.matching_iterative = fake_loop
matching.matching_iterative_lockstep = fake_lockstep
class FakeModel:
    def _device(self):
        return torch.device('cpu')
expect = np.stack([eval_loop.summarize(fake_one(provider(i)), False, provider(i)) for i in range(len(costs))])
groups_seen = []
def fake_lockstep_rec(datas, m, *a, **k):
    groups_seen.append([d['pid'] for d in datas])
    return fake_lockstep(datas, m, *a, **k)
for kw in (dict(), dict(schedule='lpt', pair_cost=lambda i: costs[i]), dict(schedule='lpt', pair_cost=lambda i: costs[i], lockstep=3), dict(lockstep=2),
           dict(lockstep=2, group_similar=4, pair_cost=lambda i: costs[i]), dict(lockstep=3, group_similar=100, pair_cost=lambda i: costs[i], schedule='lpt'),
           dict(schedule='dynamic'), dict(schedule='dynamic', lockstep=3), dict(schedule='dynamic', lockstep=2, group_similar=4, pair_cost=lambda i: costs[i])):
    tab = eval_loop.run_pairs_sharded(FakeModel(), provider, len(costs), **kw)
    assert tab.shape == expect.shape and np.array_equal(np.nan_to_num(tab, nan=-7.0), np.nan_to_num(expect, nan=-7.0)), (rank, kw.keys())
# group_similar: inside every window of W pairs of a rank the groups are formed in descending cost order (world 1 here: checked on rank 0's own call)
matching.matching_iterative_lockstep = fake_lockstep_rec
if rank == 0:
    import torch.distributed as _d
    _saved = (_d.is_initialized,)
    _d.is_initialized = lambda: False            # a single-rank call inside the 2-rank job
    try:
        tab = eval_loop.run_pairs_sharded(FakeModel(), provider, 8, lockstep=2, group_similar=4, pair_cost=lambda i: costs[i])
    finally:
        _d.is_initialized = _saved[0]
    assert np.array_equal(np.nan_to_num(tab, nan=-7.0), np.nan_to_num(expect[:8], nan=-7.0))
    assert groups_seen == [[1, 3], [2, 0], [4, 7], [5, 6]], groups_seen       # windows [0..3] and [4..7], each by descending cost (ties: lower id first)
matching.matching_iterative_lockstep = fake_lockstep
# the EIMP loop through the same schedules (round 4: groups via matching_iterative_uncertainty_lockstep, with_uncertainty passed like eval_imp.py does)
seen_unc = []
def fake_unc_one(data):
    idx, ms, R, t, nit = fake_one(data)
    n = data['n']
    keep = max(1, n - data['pid'] % 3)
    return (np.zeros((keep, 2), np.float32), np.zeros((12, 2), np.float32), np.zeros((keep, 2), np.float32), np.zeros((12, 2), np.float32), idx[:keep], ms[:keep], R, t, nit)
def fake_unc_loop(data, m, *a, **k):
    seen_unc.append(k.get('with_uncertainty'))
    return fake_unc_one(data)
def fake_unc_lockstep(datas, m, *a, **k):
    seen_unc.append(k.get('with_uncertainty'))
    return [fake_unc_one(d) for d in datas]
matching.matching_iterative_uncertainty = fake_unc_loop
matching.matching_iterative_uncertainty_lockstep = fake_unc_lockstep
expect_u = np.stack([eval_loop.summarize(fake_unc_one(provider(i)), True, provider(i)) for i in range(len(costs))])
for kw in (dict(), dict(lockstep=3), dict(lockstep=2, schedule='lpt', pair_cost=lambda i: costs[i], group_similar=4)):
    tab = eval_loop.run_pairs_sharded(FakeModel(), provider, len(costs), eimp=True, **kw)
    assert tab.shape == expect_u.shape and np.array_equal(np.nan_to_num(tab, nan=-7.0), np.nan_to_num(expect_u, nan=-7.0)), (rank, 'eimp', kw.keys())
assert seen_unc and all(v is True for v in seen_unc), seen_unc
parts = pdist.lpt_assignment(costs, world)
loads = [sum(costs[i] for i in p_) for p_ in parts]
assert sorted(sum(parts, [])) == list(range(len(costs))) and max(loads) - min(loads) <= max(costs), loads
blocks = [sum(costs[i] for i in range(*pdist.shard_range(len(costs), r_, world))) for r_ in range(world)]
assert max(loads) < max(blocks), (loads, blocks)          # the contiguous blocks are the worse split of this list
dist.barrier()
dist.destroy_process_group()
print('rank', rank, 'ok')
'''


def test_world_size_2_all_gather_over_gloo(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0, out


_WORKER4 = r'''
import os, sys, time, torch, torch.distributed as dist
import numpy as np
sys.path.insert(0, sys.argv[1])
from imp_release_amd import dist as pdist, eval_loop, matching
dist.init_process_group('gloo', rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']))
rank, world = dist.get_rank(), dist.get_world_size()
# 48 pairs whose cost (the loop's exit iteration: 6 ... 15) is SKEWED along the list: the first quarter - one rank's contiguous block - holds
# all the long ones.  No size-based static split sees that; the shared counter does
nit = [15] * 12 + [6] * 36
def provider(pid):
    return {'pid': pid, 'n': 20, 'keypoints0': torch.zeros(1, 20, 2), 'keypoints1': torch.zeros(1, 12, 2)}
def fake_one(data):
    pid = data['pid']
    idx = np.full(20, -1, dtype=np.int64); idx[:pid % 20] = np.arange(pid % 20) % 12
    return idx, (np.arange(20) % 5).astype(np.float32) / 5, None, None, nit[pid]
spent = [0.0]
def fake_loop(data, m, *a, **k):
    t = 0.002 * nit[data['pid']]
    time.sleep(t); spent[0] += t
    return fake_one(data)
def fake_lockstep(datas, m, *a, **k):
    t = 0.002 * max(nit[d['pid']] for d in datas)
    time.sleep(t); spent[0] += t
    return [fake_one(d) for d in datas]
matching.matching_iterative = fake_loop
matching.matching_iterative_lockstep = fake_lockstep
class FakeModel:
    def _device(self):
        return torch.device('cpu')
expect = np.stack([eval_loop.summarize(fake_one(provider(i)), False, provider(i)) for i in range(len(nit))])
loads = {}
for name, kw in (('block', dict()), ('dynamic', dict(schedule='dynamic')), ('dynamic_groups', dict(schedule='dynamic', lockstep=2))):
    spent[0] = 0.0
    dist.barrier()
    tab = eval_loop.run_pairs_sharded(FakeModel(), provider, len(nit), **kw)
    assert np.array_equal(np.nan_to_num(tab, nan=-7.0), np.nan_to_num(expect, nan=-7.0)), (rank, name)
    t = torch.tensor([spent[0]], dtype=torch.float64)
    allt = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(allt, t)
    loads[name] = [float(x) for x in allt]
total = 0.002 * sum(nit)
assert max(loads['block']) >= 0.002 * 15 * 12 - 1e-9, loads            # rank 0's contiguous block holds every long pair
for name in ('dynamic', 'dynamic_groups'):
    assert max(loads[name]) <= total / world + 0.002 * 15 * 2 + 1e-9, (name, loads)      # within two of the longest items of the even split
    assert max(loads[name]) < 0.75 * max(loads['block']), (name, loads)
if rank == 0:
    print('loads (s of loop time per rank):', {k: [round(v, 3) for v in x] for k, x in loads.items()})
dist.barrier()
dist.destroy_process_group()
print('rank', rank, 'ok')
'''


def test_world_size_4_dynamic_pair_queue_over_gloo(tmp_path):
    """VERDICT r4 #8b: configs[4] on several ranks with pairs whose exit iteration is skewed - the dynamic schedule (one shared counter in
    the job's store, dist.DynamicPairQueue) evens the ranks' loop time out where contiguous blocks leave one rank with all the long pairs;
    the summary table is the same under every schedule"""
    script = tmp_path / 'worker4.py'
    script.write_text(_WORKER4)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(4):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='4', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0, out


def test_lpt_assignment_and_numa_lookup(tmp_path):
    """round 4: rank-level schedule for pairs of unequal cost, and rank -> NUMA-node CPU affinity from sysfs"""
    from imp_release_amd import dist as pdist
    costs = [5, 1, 1, 1, 9, 2, 2, 7]
    parts = pdist.lpt_assignment(costs, 3)
    assert sorted(sum(parts, [])) == list(range(8)) and all(p == sorted(p) for p in parts)
    loads = [sum(costs[i] for i in p) for p in parts]
    assert max(loads) <= 10 and pdist.lpt_assignment(costs, 3) == parts                 # 28 / 3 = 9.33: within one small item of the optimum; deterministic
    assert pdist.lpt_assignment([3, 3, 3], 1) == [[0, 1, 2]] and pdist.lpt_assignment([], 2) == [[], []]
    # a fake sysfs: GPU 0 on node 1 (CPUs 4-7, 12), GPU 1 on node -1 (no affinity), GPU 2 missing
    for i, node in ((0, '1'), (1, '-1')):
        d = tmp_path / 'class' / 'drm' / f'card{i}' / 'device'
        d.mkdir(parents=True)
        (d / 'numa_node').write_text(node + '\n')
    nd = tmp_path / 'devices' / 'system' / 'node' / 'node1'
    nd.mkdir(parents=True)
    (nd / 'cpulist').write_text('4-7,12\n')
    assert pdist.gpu_numa_cpus(0, str(tmp_path)) == [4, 5, 6, 7, 12]
    assert pdist.gpu_numa_cpus(1, str(tmp_path)) is None and pdist.gpu_numa_cpus(2, str(tmp_path)) is None
    before = sorted(os.sched_getaffinity(0))
    try:
        got = pdist.pin_to_gpu_numa(0, str(tmp_path))
        want = sorted(set(before) & {4, 5, 6, 7, 12})
        assert got == (want or None) and sorted(os.sched_getaffinity(0)) == (want or before)
        os.environ['IMP_NUMA_AFFINITY'] = '0'
        assert pdist.pin_to_gpu_numa(0, str(tmp_path)) is None
    finally:
        os.environ.pop('IMP_NUMA_AFFINITY', None)
        os.sched_setaffinity(0, before)


def test_step_pipeline_orders_results_and_surfaces_errors():
    from imp_release_amd import pipeline
    import time

    def make_fn(i):
        state = {'s': i}

        def fn():
            s_ = state['s']; state['s'] += 2
            time.sleep(0.003 * (1 - i))
            return torch.full((2, 3), s_, dtype=torch.int64), torch.full((2, 3), float(s_))
        return fn

    pipe = pipeline.StepPipeline([make_fn(0), make_fn(1)], 2)
    outs = pipe.run(7, keep=True)
    assert [int(o[0][0, 0]) for o in outs] == list(range(7))
    assert int(pipeline.StepPipeline([make_fn(0)], 2).run(4)[0][0, 0]) == 6        # one worker: its 4th call

    def boom():
        raise RuntimeError('step failed')
    with pytest.raises(RuntimeError, match='step failed'):
        pipeline.StepPipeline([boom, boom], 2).run(4)


def _toy_records(n, seed=0):
    from imp_release_amd import synthetic
    recs = []
    for i in range(n):
        p = synthetic.make_correlated_pair(60 + 7 * i, 50 + 3 * i, seed=seed + i)
        recs.append({'K1': np.eye(3) * (1 + i), 'K2': np.eye(3) * 2, 'R': np.eye(3), 'T': np.array([3., 0., 4.]) * (i + 1),
                     'e': np.full((3, 3), float(i)), 'f': np.zeros((3, 3)),
                     'kpt1': np.concatenate([p['keypoints0'][0], p['scores0'][0][:, None]], 1),
                     'kpt2': np.concatenate([p['keypoints1'][0], p['scores1'][0][:, None]], 1),
                     'desc1': p['descriptors0'][0], 'desc2': p['descriptors1'][0], 'size1': (480, 640), 'size2': (360, 500)})
    return recs


def test_pair_store_roundtrip_and_feed_dict(tmp_path):
    """reference dump layout (components/readers.py:8-33) through the npz mirror: fields, num_kpt cut, t normalised, and
    the feed dict of eval/eval_imp.py:50-80 (HWC image shape kept: the loops read shape[2:4] = (W, 3))"""
    from imp_release_amd import data
    recs = _toy_records(4)
    assert data.write_npz_store(recs, str(tmp_path)) == 4
    store = data.NpzPairStore(str(tmp_path), num_kpt=55)
    assert len(store) == 4
    r = store.record(2)
    assert r['x1'].shape == (55, 3) and r['desc1'].shape == (55, 256) and r['x2'].shape == (55, 3)
    assert np.allclose(r['t'], [0.6, 0.0, 0.8]) and np.array_equal(r['K1'], np.eye(3) * 3) and r['index'] == 2
    assert np.array_equal(r['x1'], recs[2]['kpt1'][:55]) and np.array_equal(r['desc2'], recs[2]['desc2'][:55])
    d = data.feed_data(r, 'cpu')
    assert tuple(d['keypoints0'].shape) == (1, 55, 2) and tuple(d['scores1'].shape) == (1, 55)
    assert torch.equal(d['scores0'][0], torch.from_numpy(recs[2]['kpt1'][:55, 2].astype(np.float32)))
    assert tuple(d['image0'].shape) == (1, 480, 640, 3) and tuple(d['image1'].shape) == (1, 360, 500, 3)
    assert d['T_0to1'].shape == (3, 4) and np.array_equal(d['pts1_cpu'], recs[2]['kpt2'][:55, :2])
    got = [x['index'] for x in data.PinnedPrefetcher(store, [3, 0, 2], 'cpu', depth=2)]
    assert got == [3, 0, 2]
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match='h5py'):
            data.H5PairStore(str(tmp_path / 'x.hdf5'), image_sizes=((480, 640), (480, 640)), backend='h5py')
        (tmp_path / 'x.hdf5').write_bytes(b'not an hdf5 file' * 64)
        with pytest.raises(ValueError, match='signature'):
            data.H5PairStore(str(tmp_path / 'x.hdf5'), image_sizes=((480, 640), (480, 640)))


def test_pair_stores_vs_the_reference_reader(tmp_path):
    """`H5PairStore.record` / `NpzPairStore.record` against the outputs of the reference's own
    `components/readers.py: standard_reader.run` (tests/golden/reader_standard.npz, captured by tools/make_golden.py
    case_reader).  Both sides read the same seeded synthetic dump through the in-memory stand-in for the HDF5 FILE
    (helpers.MemH5; h5py is not installed) - the reader logic is pinned, real HDF5 decoding is not."""
    import sys
    from helpers import MemH5, load_golden, make_reader_records
    from imp_release_amd import data
    spec, z = load_golden('reader_standard')
    recs = make_reader_records(spec['seed'])
    MemH5.put('mem://test', recs)
    saved = sys.modules.get('h5py')
    sys.modules['h5py'] = MemH5.module()
    try:
        sizes = lambda i: (recs[i]['size1'], recs[i]['size2'])             # noqa: E731
        h5 = data.H5PairStore('mem://test', spec['num_kpt'], image_sizes=sizes)
        assert data.convert_h5_to_npz('mem://test', str(tmp_path), image_sizes=sizes) == int(z['n_pairs'])
    finally:
        if saved is None:
            del sys.modules['h5py']
        else:
            sys.modules['h5py'] = saved
    npz = data.NpzPairStore(str(tmp_path), spec['num_kpt'])
    assert len(h5) == len(npz) == int(z['n_pairs'])
    for i in range(len(h5)):
        for store in (h5, npz):
            r = store.record(i)
            for k in ('K1', 'K2', 'R', 't', 'x1', 'x2', 'desc1', 'desc2', 'e', 'f', 'r_gt', 't_gt'):
                want = z[f'{k}_{i}']
                got = np.asarray(r[k])
                assert got.dtype == want.dtype and got.shape == want.shape and np.array_equal(got, want), (k, i, got.dtype, want.dtype)
            assert r['index'] == i
            d = data.feed_data(r, 'cpu')
            # the reference uploads the decoded HWC image and the loops read only its shape (eval/eval_imp.py:56-57)
            assert tuple(d['image0'].shape) == (1,) + tuple(int(v) for v in z[f'img1_shape_{i}'])
            assert tuple(d['image1'].shape) == (1,) + tuple(int(v) for v in z[f'img2_shape_{i}'])
            assert np.array_equal(d['T_0to1'], np.hstack([z[f'R_{i}'], z[f't_{i}'].reshape(3, 1)]))
        assert h5.record(i)['img_path1'] == recs[i]['img_path1'] and h5.record(i)['img_path2'] == recs[i]['img_path2']


def test_real_hdf5_dump_through_the_builtin_decoder(tmp_path):
    """SURVEY §8 f-2, VERDICT r3 'real HDF5 decoding': tests/golden/reader_dump.hdf5 was written by h5py ITSELF (3.3.0 / HDF5 1.10.6, found
    under an Anaconda interpreter of the build image: tools/make_h5_fixture_h5py.py) with the statements of dump/dumper/base_dumper.py:85-111
    (superblock 0, symbol-table groups, contiguous float32 / float64 datasets, variable-length ASCII strings in the global heap) from the
    records the reference's own reader was run on.  imp_release_amd.h5lite decodes it to exactly those arrays, and H5PairStore on
    the FILE returns what components/readers.py standard_reader.run returned (tests/golden/reader_standard.npz)."""
    import os
    from helpers import GOLD, load_golden, make_reader_records
    from imp_release_amd import data, h5lite
    spec, z = load_golden('reader_standard')
    recs = make_reader_records(spec['seed'])
    path = os.path.join(GOLD, 'reader_dump.hdf5')
    with h5lite.File(path) as f:
        assert sorted(f.keys()) == sorted(data.FIELDS + ('img_path1', 'img_path2')) and len(f['K1']) == len(recs) and 'desc1' in f and 'nope' not in f
        for i, r in enumerate(recs):
            for k in data.FIELDS:
                a = f[k][str(i)][()]
                assert a.dtype == np.asarray(r[k]).dtype and a.shape == np.asarray(r[k]).shape and np.array_equal(a, r[k]), (k, i)
                assert np.array_equal(np.asarray(f[k][str(i)]), r[k]) and np.array_equal(f[f'/{k}/{i}'][:5], np.asarray(r[k])[:5])
            assert f['img_path1'][str(i)][()][0].decode() == r['img_path1'] and f['img_path2'][str(i)].shape == (1,)
        with pytest.raises(KeyError):
            f['K1']['17']
    sizes = lambda i: (recs[i]['size1'], recs[i]['size2'])             # noqa: E731
    store = data.H5PairStore(path, spec['num_kpt'], image_sizes=sizes, backend='h5lite')
    assert store.backend == 'h5lite' and len(store) == int(z['n_pairs'])
    for i in range(len(store)):
        r = store.record(i)
        for k in ('K1', 'K2', 'R', 't', 'x1', 'x2', 'desc1', 'desc2', 'e', 'f'):
            ref = z[f'{k}_{i}']
            assert r[k].dtype == ref.dtype and np.array_equal(r[k], ref), (i, k)
        assert r['img_path1'] == recs[i]['img_path1'] and r['img_path2'] == recs[i]['img_path2']
    store.close()
    assert data.convert_h5_to_npz(path, str(tmp_path), image_sizes=sizes, backend='h5lite') == len(recs)
    npz = data.NpzPairStore(str(tmp_path), spec['num_kpt'])
    assert np.array_equal(npz.record(1)['desc2'], z['desc2_1'])


def test_builtin_hdf5_decoder_other_corners_of_the_format():
    """files of the HDF5 C library with libver='latest' (superblock 3, version-2 object headers with checksummed chunks, compact link
    messages, version-4 layout messages) and with chunked storage (version-1 B-tree chunk index, shuffle + deflate, edge chunks, an
    unfiltered chunked dataset, a group spread over several symbol-table nodes); what the decoder does not read it names"""
    import os
    from helpers import GOLD, load_golden, make_reader_records
    from imp_release_amd import h5lite
    recs = make_reader_records(load_golden('reader_standard')[0]['seed'])
    ramp = (np.arange(100 * 37, dtype=np.float32).reshape(100, 37) % 17).astype(np.float32)
    with h5lite.File(os.path.join(GOLD, 'reader_dump_latest.hdf5')) as f:
        g = f['pair']
        assert sorted(f.keys()) == ['dense', 'more', 'pair'] and len(g) == 8
        assert np.array_equal(g['K1'][()], recs[0]['K1']) and np.array_equal(np.asarray(g['kpt1']), recs[0]['kpt1'])
        assert np.array_equal(g['ramp_single_chunk_deflate'][()], ramp)
        assert np.array_equal(g['ids_compact_i32'][()], np.arange(-5, 20, dtype=np.int32)) and g['ids_compact_i32'].dtype == np.int32
        assert np.array_equal(g['counts_i64'][()], np.array([[1, -2, 3], [2 ** 40, 5, -6]]))
        assert g['path_fixed'][()][0].decode() == recs[0]['img_path1']
        assert [v.decode() for v in g['paths_vlen'][()]] == [recs[0]['img_path1'], recs[0]['img_path2'], '']
        assert np.array_equal(f['more/bytes_u8'][()], np.arange(200, dtype=np.uint8)) and np.array_equal(f['/more/shorts_i16'][()], [-300, 7, 300])
        assert np.array_equal(g['ramp_chunked_shuffle_deflate'][()], ramp)                                       # fixed-array chunk index, filtered
        assert np.array_equal(f['more/ids_1500_chunks'][()], np.arange(3000, dtype=np.int32) * 3)                # paged fixed array
        assert np.array_equal(f['more/grid_implicit'][()], np.arange(240, dtype=np.float64).reshape(20, 12))    # implicit index, edge chunks
        assert np.array_equal(f['more/ids_1100_chunks_deflate_fletcher'][()], np.arange(4400, dtype=np.int16))  # paged, filtered, checksummed
        with pytest.raises(NotImplementedError, match='densely'):
            len(f['dense'])
    with h5lite.File(os.path.join(GOLD, 'reader_dump_chunked.hdf5')) as f:
        assert np.array_equal(f['ramp'][()], ramp) and f['ramp'].shape == (100, 37)
        assert np.array_equal(f['ids_chunked_unfiltered'][()], np.arange(50))
        assert len(f['many']) == 40 and all(np.array_equal(f['many'][str(i)][()], [i, i * i]) for i in range(40))


def test_superpoint_align_corners_rule():
    """nets/superpoint.py:89 `int(torch.__version__[2]) > 2`: the third CHARACTER of the version string"""
    from imp_release_amd.superpoint import reference_align_corners as rule
    assert [rule(v) for v in ('1.2.0', '1.3.1', '1.7.1', '1.9.0', '1.10.2', '1.12.1', '2.0.1', '2.2.0', '2.3.0', '2.9.1', '2.10.0+rocm7.0')] == \
        [False, True, True, True, False, False, False, False, True, True, False]


def test_error_classes_of_the_binding_are_distinct():
    """"does not fit" (IMP_E_NOFIT) and "timed out" (IMP_E_RESIDENT) are told apart by error code / exception class; the lock-step
    wrappers split a group on the first and re-run it on the second"""
    from imp_release_amd import _lib
    assert _lib.IMP_E_NOFIT == -8 and _lib.IMP_E_RESIDENT == -6 and _lib.IMP_E_RANGE == -7
    assert issubclass(_lib.ResidentDoesNotFit, _lib.ImpError) and issubclass(_lib.ResidentSinkhornTimeout, _lib.ImpError)
    assert not issubclass(_lib.ResidentDoesNotFit, _lib.ResidentSinkhornTimeout)
    assert not issubclass(_lib.ResidentSinkhornTimeout, _lib.ResidentDoesNotFit)
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'imp_hip.h')).read()
    assert '#define IMP_E_NOFIT (-8)' in hdr


def test_weights_version_sees_a_replaced_parameter_object():
    """the cached parameter list of GM._weights_version is keyed on the identity of the modules' current Parameter / buffer
    objects - replacing one (not an in-place update, not load_state_dict on the top module) must change the version the HIP context is keyed on"""
    cfg = eval_config(n_layers=2)
    m = P.GM(cfg).eval()
    v0 = m._weights_version()
    assert m._weights_version() == v0
    conv = m.final_proj[0]
    conv.weight = torch.nn.Parameter(conv.weight.detach().clone() * 2.0)          # a NEW Parameter object on a submodule
    v1 = m._weights_version()
    assert v1 != v0
    m.final_proj[1].load_state_dict({k: v.clone() + 1 for k, v in m.final_proj[1].state_dict().items()}, assign=True)      # submodule-level load, objects replaced
    assert m._weights_version() != v1
    with torch.no_grad():
        m.bin_score.add_(1.0)                                                       # in-place: _version
    assert m._weights_version()[1] != v1[1]


def test_attention_handle_behaves_like_the_probability_tensor():
    """VERDICT r4 missing #5: model.self_prob* / cross_prob* are handles (the kernels keep Q, K, lse) - a caller that does tensor arithmetic on
    them (nets/gms.py:236-248 returns them as lists) gets the tensor's behaviour: torch functions, operators, indexing, methods, attributes"""
    from imp_release_amd.modules import AttentionHandle

    class FakeCtx:
        calls = 0

        def attention_prob(self, which, B, nq, nk, dev):
            FakeCtx.calls += 1
            return torch.softmax(torch.arange(B * 4 * nq * nk, dtype=torch.float32).view(B, 4, nq, nk) / 7.0, dim=-1)

    class FakeModel:
        _attn_generation = [3, 0, 0, 0]
        _ctx = FakeCtx()

        def _device(self):
            return 'cpu'

    h = AttentionHandle(FakeModel(), 0, 3, (1, 4, 3, 5))
    t = FakeCtx().attention_prob(0, 1, 3, 5, 'cpu')
    assert torch.equal(torch.sum(h, dim=1), t.sum(1)) and torch.equal(h * 2, t * 2) and torch.equal(2 - h, 2 - t) and torch.equal(h[0, 1], t[0, 1])
    assert h.dim() == 4 and h.size(2) == 3 and h.shape == (1, 4, 3, 5) and len(h) == 1 and h.dtype == torch.float32
    assert torch.equal(h.sum((1, 2)), t.sum((1, 2))) and torch.equal(torch.cat([h, h], 0), torch.cat([t, t], 0)) and bool((h >= 0).all())
    assert torch.equal(h @ t.transpose(-1, -2), t @ t.transpose(-1, -2)) and torch.allclose(h.sum(-1), torch.ones(1, 4, 3))
    FakeModel._attn_generation[0] = 4                       # a later layer overwrote the cache: the materialised tensor stays, a fresh handle refuses
    assert torch.equal(h.cpu(), t)
    with pytest.raises(RuntimeError):
        AttentionHandle(FakeModel(), 0, 3, (1, 4, 3, 5)).materialize()


def test_bench_cli_parses_the_drivers_command_line_and_refuses_to_run_without_a_gpu():
    """bench.py is what the driver runs (`python bench.py --gpus N --steps K --warmup W`): the file must compile, know those flags, and - on a host
    without a GPU - stop with a clear message instead of measuring anything else (the matching hot path has no CPU fallback)."""
    import ast
    import sys
    path = os.path.join(ROOT, 'bench.py')
    ast.parse(open(path).read())
    out = subprocess.run([sys.executable, path, '--help'], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ('--gpus', '--steps', '--warmup', '--in-flight', '--exchange-every', '--no-cpu-baseline'):
        assert flag in out.stdout, flag
    if not torch.cuda.is_available():
        run = subprocess.run([sys.executable, path, '--gpus', '1', '--steps', '20', '--warmup', '5'], capture_output=True, text=True, timeout=300)
        assert run.returncode != 0 and 'needs a GPU' in (run.stderr + run.stdout)
