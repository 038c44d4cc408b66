"""GPU: the whole hot path through the reference-shaped boundary (GM / DGNNS / AdaGMN over the C-ABI)
against (a) the golden vectors captured from the imported reference, (b) the oracle on the same seeded
inputs, and (c) size-independent properties at the BASELINE sizes (N = 2048) where the oracle is slow.

Parity bar (BASELINE.json north_star): match indices identical, scores within 1e-4."""
import numpy as np
import pytest
import torch

from helpers import build_case, compare_matches, eval_config, golden_names, load_golden, make_hip_model
from imp_release_amd import eval_loop, matching as hip_matching, synthetic
from oracle import imp_oracle as orc

pytestmark = pytest.mark.gpu
DEV = 'cuda'
TOL = 1e-4


def _cpu(x):
    return x.detach().cpu()


@pytest.mark.parametrize('precision', ['f16x3', 'f32'])
@pytest.mark.parametrize('name', golden_names(['gm_l', 'dgnns_l', 'adagmn_masked']))
def test_produce_matches_vs_golden(name, precision):
    spec, z = load_golden(name)
    cfg, sd, data = build_case(spec, DEV)
    m = make_hip_model(spec, cfg, sd, precision=precision)
    assert m._ensure_ctx().precision == precision
    call = spec.get('call', {})
    with torch.no_grad():
        out = m.produce_matches(data, **call)
    n = int(z['n_emitted'])
    assert len(out['indices0']) == n
    msgs = []
    for i in range(n):
        assert out['indices0'][i].dtype == torch.int64
        msgs.append(compare_matches(_cpu(out['indices0'][i]), _cpu(out['mscores0'][i]), z[f'indices0_{i}'],
                                    z[f'mscores0_{i}'], call.get('p', 0.2), TOL, f'{name}[{i}]'))
    if 'score_rowsum' in z.files and out.get('scores'):
        s = _cpu(out['scores'][-1])[0].double()
        assert np.abs(s.sum(-1).numpy() - z['score_rowsum']).max() < 5e-4
        assert np.abs(s[:8, :8].numpy() - z['score_corner']).max() < TOL
    print('\n'.join(msgs))


@pytest.mark.parametrize('name', golden_names(['gm_run', 'adagmn_run']))
def test_run_vs_golden(name):
    spec, z = load_golden(name)
    cfg, sd, data = build_case(spec, DEV)
    m = make_hip_model(spec, cfg, sd)
    ctx = m._ensure_ctx()
    nk0 = ctx.normalize_keypoints(data['keypoints0'], 640, 480)
    nk1 = ctx.normalize_keypoints(data['keypoints1'], 640, 480)
    rd = {'desc1': data['descriptors0'], 'desc2': data['descriptors1'],
          'x1': torch.cat([nk0, data['scores0'][..., None]], -1), 'x2': torch.cat([nk1, data['scores1'][..., None]], -1)}
    with torch.no_grad():
        out = m(rd, mode=1)
    if 'p' in out:
        s = _cpu(out['p'])[0].double()
        assert np.abs(s.sum(-1).numpy() - z['score_rowsum']).max() < 5e-4
        assert np.abs(s[:8, :8].numpy() - z['score_corner']).max() < TOL
    else:
        assert np.array_equal(_cpu(out['index0']).numpy(), z['index0'])
        assert np.array_equal(_cpu(out['index1']).numpy(), z['index1'])


def _loop_data(data):
    d = dict(data)
    d['pts0_cpu'] = data['keypoints0'][0].cpu().numpy()
    d['pts1_cpu'] = data['keypoints1'][0].cpu().numpy()
    d['K0'] = d['K1'] = np.eye(3)
    d['T_0to1'] = np.eye(4)
    return d


def test_imp_iterative_loop_vs_golden():
    spec, z = load_golden('imp_loop_n400')
    cfg, sd, data = build_case(spec, DEV)
    m = make_hip_model(spec, cfg, sd)
    trace = []
    with torch.no_grad():
        i0, ms0, R, t, nit = hip_matching.matching_iterative(_loop_data(data), m, 15, 0.1, 25, 1.0, {'pose': 1.5},
                                                             trace=trace)
    assert nit == int(z['n_iter']) and R is None
    assert [t['it'] for t in trace] == [3, 5, 7, 9, 11, 13, 14]
    for k, t in enumerate(trace):
        compare_matches(t['indices0'], t['mscores0'], z[f'it{k}_indices0'], z[f'it{k}_mscores0'], 0.1, TOL, f'imp it{t["it"]}')
    compare_matches(i0, ms0, z['indices0'], z['mscores0'], 0.2, TOL, 'imp loop final')


@pytest.mark.parametrize('precision', ['f16x3', 'f32'])
def test_eimp_sliced_loop_vs_golden(precision):
    """BASELINE config 4 analogue: real ragged slicing (pool -> compaction -> gather), pinned to the
    reference's pruning trajectory 1024/1000 -> 751/725 -> ... (tests/golden/eimp_loop_sliced_n1024)."""
    spec, z = load_golden('eimp_loop_sliced_n1024')
    cfg, sd, data = build_case(spec, DEV)
    m = make_hip_model(spec, cfg, sd, precision=precision)
    trace = []
    with torch.no_grad():
        p0, p1, nk0, nk1, i0, ms0, R, t, nit = hip_matching.matching_iterative_uncertainty(
            _loop_data(data), m, 15, 0.1, 25, 1.0, {'pose': 1.5}, with_uncertainty=False, trace=trace)
    assert nit == int(z['n_iter'])
    traj = [(t['n0'], t['n1']) for t in trace]
    assert traj == [tuple(r) for r in z['trajectory'].tolist()], f'pruning trajectory {traj}'
    k0, k1 = data['keypoints0'][0].cpu().numpy(), data['keypoints1'][0].cpu().numpy()
    for k, t in enumerate(trace):
        assert np.array_equal(t['pts0'], k0[z[f'it{k}_keep0']]) and np.array_equal(t['pts1'], k1[z[f'it{k}_keep1']]), f'keep set it{k}'
        compare_matches(t['indices0'], t['mscores0'], z[f'it{k}_indices0'], z[f'it{k}_mscores0'], 0.1, TOL, f'eimp it{t["it"]}')
    assert p0.shape == z['pts0_final'].shape and p1.shape == z['pts1_final'].shape, \
        f'pruned sizes {p0.shape[0]}/{p1.shape[0]} vs reference {z["pts0_final"].shape[0]}/{z["pts1_final"].shape[0]}'
    assert np.array_equal(p0, z['pts0_final']) and np.array_equal(p1, z['pts1_final'])
    compare_matches(i0, ms0, z['indices0'], z['mscores0'], 0.2, TOL, 'eimp loop final')


def test_reference_style_step_api_loop_matches_fused_path():
    """eval/matching.py drives the model through [B, D, N] tensors (encode_keypoint, forward_one_layer,
    compute_distance, compute_score, compute_matches): same result as the fused produce_matches."""
    cfg = eval_config(n_layers=3)
    sd = synthetic.make_state_dict(cfg, 'DGNNS', seed=2)
    m = make_hip_model('DGNNS', cfg, sd)
    pair = synthetic.make_correlated_pair(210, 190, seed=4)
    data = {k: torch.from_numpy(v).to(DEV) for k, v in pair.items() if k != 'image_shape'}
    data['image0'] = data['image1'] = torch.zeros(pair['image_shape'], device=DEV)
    with torch.no_grad():
        fused = m.produce_matches(data, p=0.2, only_last=True)
        ctx = m._ensure_ctx()
        nk0 = ctx.normalize_keypoints(data['keypoints0'], 640, 480)
        nk1 = ctx.normalize_keypoints(data['keypoints1'], 640, 480)
        desc0, desc1 = data['descriptors0'].transpose(1, 2), data['descriptors1'].transpose(1, 2)
        enc0, enc1 = m.encode_keypoint(norm_kpts0=nk0, norm_kpts1=nk1, scores0=data['scores0'], scores1=data['scores1'])
        assert enc0.shape == (1, 256, 210)
        desc0, desc1 = desc0 + enc0, desc1 + enc1
        for it in range(3):
            desc0, desc1 = m.forward_one_layer(desc0=desc0, desc1=desc1, M0=None, M1=None, layer_i=it * 2)
            desc0, desc1 = m.forward_one_layer(desc0=desc0, desc1=desc1, M0=None, M1=None, layer_i=it * 2 + 1)
        dist = m.compute_distance(desc0=desc0, desc1=desc1, layer_id=2)
        score = m.compute_score(dist=dist, dustbin=m.bin_score, iteration=m.sinkhorn_iterations)
        i0, i1, ms0, ms1 = m.compute_matches(scores=score, p=0.2)
    assert m.self_prob0.materialize().shape == (1, 4, 210, 210) and m.cross_prob0.shape == (1, 4, 190, 210)
    assert torch.equal(i0, fused['indices0'][-1]) and torch.equal(ms0, fused['mscores0'][-1])


# ---------------------------------------------------------------------------------------------------
# BASELINE sizes (N = 2048, L = 9, T = 100): oracle comparison once + size-independent properties
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope='module', params=['f16x3', 'f32'])
def big(request):
    cfg = eval_config(n_layers=9, sinkhorn_iterations=100)
    sd = synthetic.make_state_dict(cfg, 'GM', seed=1)
    m = make_hip_model('GM', cfg, sd, precision=request.param)
    pair = synthetic.make_correlated_pair(2048, 2048, seed=31, batch=2)
    data = {k: torch.from_numpy(v).to(DEV) for k, v in pair.items() if k != 'image_shape'}
    data['image0'] = data['image1'] = torch.zeros(pair['image_shape'], device=DEV)
    with torch.no_grad():
        out = m.produce_matches(data, p=0.2, only_last=True)
    return cfg, sd, m, pair, data, out


def test_full_size_vs_oracle(big):
    cfg, sd, m, pair, data, out = big
    o = orc.MatcherOracle(cfg, sd, 'GM')
    cdata = {k: v[:1].cpu() for k, v in data.items()}
    torch.set_num_threads(max(1, torch.get_num_threads()))
    with torch.no_grad():
        ref = o.produce_matches(cdata, p=0.2, only_last=True)
    print(compare_matches(_cpu(out['indices0'][-1][:1]), _cpu(out['mscores0'][-1][:1]), ref['indices0'][-1].numpy(),
                          ref['mscores0'][-1].numpy(), 0.2, TOL, 'N=2048 L=9 T=100'))


def test_full_size_properties(big):
    cfg, sd, m, pair, data, out = big
    i0, ms0, score = out['indices0'][-1], out['mscores0'][-1], out['scores'][-1]
    B, N = i0.shape
    assert int((i0 >= 0).sum()) > 0.2 * N * B
    # Sinkhorn: columns meet their marginals exactly after the last step; everything finite and non-negative
    cs = score.double().sum(1)
    assert (cs[:, :-1] - 1).abs().max().item() < 1e-4 and (cs[:, -1] - (N + 1)).abs().max().item() < 2e-3
    assert torch.isfinite(score).all() and (score >= 0).all()
    # mutual consistency recomputed from the score tensor (integer work: exact)
    r0, r1, rm0, rm1 = m.compute_matches(score, 0.2)
    assert torch.equal(r0, i0) and torch.equal(rm0, ms0)
    for b in range(B):
        v = torch.where(r0[b] >= 0)[0]
        assert torch.equal(r1[b][r0[b][v]], v)
    # batch invariance: pair 1 alone == pair 1 inside the batch (per-sample ops only)
    solo = {k: (v[1:2] if v.shape[0] == B else v) for k, v in data.items()}
    with torch.no_grad():
        o1 = m.produce_matches(solo, p=0.2, only_last=True)
    print(compare_matches(_cpu(o1['indices0'][-1]), _cpu(o1['mscores0'][-1]), _cpu(i0[1:2]).numpy(), _cpu(ms0[1:2]).numpy(),
                          0.2, TOL, 'batch invariance'))
    # permutation equivariance: shuffling image 1's keypoints permutes the matches accordingly
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(1)).to(DEV)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(N, device=DEV)
    shuf = dict(solo)
    for k in ('keypoints1', 'scores1', 'descriptors1'):
        shuf[k] = solo[k][:, perm]
    with torch.no_grad():
        o2 = m.produce_matches(shuf, p=0.2, only_last=True)
    j = o2['indices0'][-1][0]
    mapped = torch.where(j >= 0, perm[j.clamp(min=0)], j)
    print(compare_matches(_cpu(mapped)[None], _cpu(o2['mscores0'][-1]), _cpu(o1['indices0'][-1]).numpy(),
                          _cpu(o1['mscores0'][-1]).numpy(), 0.2, TOL, 'permutation equivariance'))
    # determinism: identical bits run to run (fixed reduction orders, no float atomics)
    with torch.no_grad():
        o3 = m.produce_matches(solo, p=0.2, only_last=True)
    assert torch.equal(o3['indices0'][-1], o1['indices0'][-1]) and torch.equal(o3['mscores0'][-1], o1['mscores0'][-1])
    assert torch.equal(o3['scores'][-1], o1['scores'][-1])


def test_bench_batch_of_four_takes_the_pingpong_attention_path():
    """bench.py's launch geometry (4 pairs x 2048 keypoints = one 256-query workgroup per CU): repeatable bit for bit,
    batch-invariant (a pair inside the batch == the pair alone, where the same kernels run on a quarter of the chip),
    and pair 0 checked against the oracle at the full BASELINE size."""
    cfg = eval_config(n_layers=9, sinkhorn_iterations=100)
    sd = synthetic.make_state_dict(cfg, 'GM', seed=1)
    m = make_hip_model('GM', cfg, sd)
    pair = synthetic.make_correlated_pair(2048, 2048, seed=77, batch=4)
    data = {k: torch.from_numpy(v).to(DEV) for k, v in pair.items() if k != 'image_shape'}
    data['image0'] = data['image1'] = torch.zeros(pair['image_shape'], device=DEV)
    with torch.no_grad():
        out4 = m.produce_matches(data, p=0.2, only_last=True)
        again = m.produce_matches(data, p=0.2, only_last=True)
    assert torch.equal(out4['indices0'][-1], again['indices0'][-1]) and torch.equal(out4['mscores0'][-1], again['mscores0'][-1])
    for b in (1, 3):
        solo = {k: (v[b:b + 1] if v.shape[0] == 4 else v) for k, v in data.items()}
        with torch.no_grad():
            o1 = m.produce_matches(solo, p=0.2, only_last=True)
        print(compare_matches(_cpu(out4['indices0'][-1][b:b + 1]), _cpu(out4['mscores0'][-1][b:b + 1]),
                              _cpu(o1['indices0'][-1]).numpy(), _cpu(o1['mscores0'][-1]).numpy(), 0.2, TOL,
                              f'batch of 4 vs solo, pair {b}'))
    o = orc.MatcherOracle(cfg, sd, 'GM')
    cdata = {k: v[:1].cpu() for k, v in data.items()}
    with torch.no_grad():
        ref = o.produce_matches(cdata, p=0.2, only_last=True)
    print(compare_matches(_cpu(out4['indices0'][-1][:1]), _cpu(out4['mscores0'][-1][:1]), ref['indices0'][-1].numpy(),
                          ref['mscores0'][-1].numpy(), 0.2, TOL, 'N=2048 L=9 T=100 B=4 pair 0 vs oracle'))


@pytest.mark.parametrize('seed', [101, 102, 103, 104])
def test_full_size_more_pairs_vs_oracle(seed):
    """the BASELINE workload (N=2048, L=9, T=100) on further seeded pairs, ragged second image: indices identical,
    scores within 1e-4 of the oracle"""
    cfg = eval_config(n_layers=9, sinkhorn_iterations=100)
    sd = synthetic.make_state_dict(cfg, 'GM', seed=seed)
    m = make_hip_model('GM', cfg, sd)
    pair = synthetic.make_correlated_pair(2048, 2048 - 3 * (seed % 7), seed=seed)
    data = {k: torch.from_numpy(v).to(DEV) for k, v in pair.items() if k != 'image_shape'}
    data['image0'] = data['image1'] = torch.zeros(pair['image_shape'], device=DEV)
    with torch.no_grad():
        out = m.produce_matches(data, p=0.2, only_last=True)
        ref = orc.MatcherOracle(cfg, sd, 'GM').produce_matches({k: v.cpu() for k, v in data.items()}, p=0.2, only_last=True)
    print(compare_matches(_cpu(out['indices0'][-1]), _cpu(out['mscores0'][-1]), ref['indices0'][-1].numpy(),
                          ref['mscores0'][-1].numpy(), 0.2, TOL, f'N=2048 seed {seed}'))


def test_eimp_pruning_path_at_4096():
    """BASELINE config 4 size: N = 4096 start, sliced EIMP loop, pruning must happen and stay consistent."""
    cfg = eval_config()
    sd = synthetic.make_state_dict(cfg, 'AdaGMN', seed=9, bin_score=5.0)
    m = make_hip_model('AdaGMN', cfg, sd)
    pair = synthetic.make_correlated_pair(4096, 4000, seed=41)
    data = {k: torch.from_numpy(v).to(DEV) for k, v in pair.items() if k != 'image_shape'}
    data['image0'] = data['image1'] = torch.zeros(pair['image_shape'], device=DEV)
    with torch.no_grad():
        p0, p1, nk0, nk1, i0, ms0, R, t, nit = hip_matching.matching_iterative_uncertainty(
            _loop_data(data), m, 15, 0.1, 25, 1.0, {'pose': 1.5})
    assert nit == 15 and p0.shape[0] < 4096 and p1.shape[0] < 4000 and p0.shape[0] == i0.shape[0]
    # survivors are a subsequence of the original keypoints (compaction keeps ascending order)
    k0 = pair['keypoints0'][0]
    pos = [np.nonzero((k0 == r).all(1))[0][0] for r in p0[:50]]
    assert pos == sorted(pos)
    assert ((i0 >= -1) & (i0 < p1.shape[0])).all()
    print(f'N trajectory end: {p0.shape[0]}/{p1.shape[0]}, matches {(i0 >= 0).sum()}')


def test_fused_call_is_hip_graph_capturable():
    """imp_match_pair enqueues ~300 kernels and never synchronises or allocates once the workspace is sized: the
    whole pair can be captured in a HIP graph and replayed (bitwise the same result as the eager call)."""
    cfg = eval_config(n_layers=3)
    sd = synthetic.make_state_dict(cfg, 'DGNNS', seed=2)
    m = make_hip_model('DGNNS', cfg, sd)
    pair = synthetic.make_correlated_pair(300, 280, seed=9, batch=2)
    d = {k: torch.from_numpy(v).to(DEV) for k, v in pair.items() if k != 'image_shape'}
    ctx = m._ensure_ctx()
    args = (d['keypoints0'], d['scores0'], d['descriptors0'], d['keypoints1'], d['scores1'], d['descriptors1'],
            640., 480., 1.0, 20, True, 0.2)
    eager = ctx.match_pair(*args, want_side1=True)
    out = {k: torch.zeros_like(v) for k, v in eager.items()}
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ctx.match_pair(*args, out=out)
    for _ in range(2):
        for v in out.values():
            v.zero_()
        g.replay()
        torch.cuda.synchronize()
        for k in eager:
            assert torch.equal(out[k], eager[k]), k


def test_sharded_eval_loop_single_rank():
    """BASELINE config 5 shape on one rank: several independent pairs through the EIMP loop, summary table out"""
    from imp_release_amd import eval_loop
    cfg = eval_config()
    sd = synthetic.make_state_dict(cfg, 'AdaGMN', seed=9, bin_score=5.0)
    m = make_hip_model('AdaGMN', cfg, sd)

    def provider(pid):
        pair = synthetic.make_correlated_pair(600 + 10 * pid, 580, seed=50 + pid)
        d = {k: torch.from_numpy(v).to(DEV) for k, v in pair.items() if k != 'image_shape'}
        d['image0'] = d['image1'] = torch.zeros(pair['image_shape'], device=DEV)
        return _loop_data(d)

    table = eval_loop.run_pairs_sharded(m, provider, 3, eimp=True)
    assert table.shape == (3, 5) and (table[:, 0] == 15).all()
    assert (table[:, 1] > 25).all() and (table[:, 3] < 640).all() and (table[:, 3] > 0).all()
    again = eval_loop.run_pairs_sharded(m, provider, 3, eimp=True)
    assert np.array_equal(table, again)          # deterministic


def test_batch_steps_in_flight_reproduce_the_sequential_results():
    """bench.py's default mode: 3 batch-steps in flight (3 replicas, 3 streams) - every step's result must equal the
    one-after-the-other result bit for bit"""
    from imp_release_amd import pipeline
    cfg = eval_config(n_layers=3, sinkhorn_iterations=20)
    sd = synthetic.make_state_dict(cfg, 'GM', seed=1)
    m = make_hip_model('GM', cfg, sd)
    datas = []
    for k in range(4):
        pair = synthetic.make_correlated_pair(1024, 1000, seed=60 + k, batch=2)
        d = {kk: torch.from_numpy(v).to(DEV) for kk, v in pair.items() if kk != 'image_shape'}
        d['image0'] = d['image1'] = torch.zeros(pair['image_shape'], device=DEV)
        datas.append(d)

    def make_fn(model, start, stride):
        state = {'s': start}

        def fn():
            d = datas[state['s'] % len(datas)]; state['s'] += stride
            out = model.produce_matches(d, p=0.2, only_last=True)
            return out['indices0'][-1], out['mscores0'][-1]
        return fn

    seq = pipeline.StepPipeline([make_fn(m, 0, 1)], 2, device=DEV).run(8, keep=True)
    reps = eval_loop.replicate(m, 3)
    par = pipeline.StepPipeline([make_fn(r, i, 3) for i, r in enumerate(reps)], 2, device=DEV).run(8, keep=True)
    for s_, (a, b) in enumerate(zip(seq, par)):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), s_
    assert not torch.equal(seq[0][0], seq[1][0])          # the steps really see different batches


def test_prefetched_dataset_feeds_the_loop_identically(tmp_path):
    """pairs read from the npz mirror of the reference's dump layout, staged in pinned memory and uploaded on the
    prefetcher's stream, must give the loop the same results as tensors placed on the device directly"""
    from imp_release_amd import data as pdata
    recs = []
    for i in range(4):
        p = synthetic.make_correlated_pair(640 + 16 * i, 600, seed=90 + i)
        recs.append({'K1': np.eye(3), 'K2': np.eye(3), 'R': np.eye(3), 'T': np.array([1., 2., 2.]), 'e': np.zeros((3, 3)),
                     'f': np.zeros((3, 3)), 'kpt1': np.concatenate([p['keypoints0'][0], p['scores0'][0][:, None]], 1),
                     'kpt2': np.concatenate([p['keypoints1'][0], p['scores1'][0][:, None]], 1),
                     'desc1': p['descriptors0'][0], 'desc2': p['descriptors1'][0], 'size1': (480, 640), 'size2': (480, 640)})
    pdata.write_npz_store(recs, str(tmp_path))
    store = pdata.NpzPairStore(str(tmp_path))
    cfg = eval_config()
    sd = synthetic.make_state_dict(cfg, 'AdaGMN', seed=9, bin_score=5.0)
    m = make_hip_model('AdaGMN', cfg, sd)
    direct = []
    with torch.no_grad():
        for i in range(4):
            d = pdata.feed_data(store.record(i), DEV)
            direct.append(hip_matching.matching_iterative_uncertainty(d, m, 15, 0.1, 25, 1.0, {'pose': 1.5}))
        for i, d in enumerate(pdata.PinnedPrefetcher(store, range(4), DEV, depth=2)):
            out = hip_matching.matching_iterative_uncertainty(d, m, 15, 0.1, 25, 1.0, {'pose': 1.5})
            assert np.array_equal(out[4], direct[i][4]) and np.array_equal(out[5], direct[i][5]) and out[0].shape == direct[i][0].shape


def test_eval_loop_pairs_in_flight_give_identical_rows():
    """workers=K (K model replicas, K streams, K host threads) must reproduce the sequential table bit for bit; the
    injected pose step sleeps like a host-side solver so that the overlap is observable"""
    import time
    cfg = eval_config()
    sd = synthetic.make_state_dict(cfg, 'AdaGMN', seed=9, bin_score=5.0)
    m = make_hip_model('AdaGMN', cfg, sd)

    def provider(pid):
        pair = synthetic.make_correlated_pair(700, 650, seed=500 + pid)
        d = {k: torch.from_numpy(v).to(DEV) for k, v in pair.items() if k != 'image_shape'}
        d['image0'] = d['image1'] = torch.zeros(pair['image_shape'], device=DEV)
        return _loop_data(d)

    def slow_pose(*a, **k):
        time.sleep(0.01)
        return None

    reps = eval_loop.replicate(m, 3)
    eval_loop.run_pairs_sharded(m, provider, 3, eimp=True, workers=3, replicas=reps)             # warm-up (workspaces)
    t0 = time.perf_counter()
    seq = eval_loop.run_pairs_sharded(m, provider, 6, eimp=True, estimate_pose=slow_pose)
    t1 = time.perf_counter()
    par = eval_loop.run_pairs_sharded(m, provider, 6, eimp=True, estimate_pose=slow_pose, workers=3, replicas=reps)
    t2 = time.perf_counter()
    assert np.array_equal(seq, par), (seq, par)
    print(f'6 pairs: sequential {t1 - t0:.3f} s, 3 in flight {t2 - t1:.3f} s')


@pytest.mark.parametrize('precision', ['f16x3', 'f32'])
@pytest.mark.parametrize('model', ['GM', 'DGNNS'])
def test_odd_and_tiny_shapes_vs_oracle(model, precision):
    """ragged / tiny keypoint counts straddling every tile boundary (1, 2, 31..33, 63..65, 127..129, 255..257)"""
    nl = 4 if model == 'DGNNS' else 2
    cfg = eval_config(n_layers=nl, sinkhorn_iterations=10)
    sd = synthetic.make_state_dict(cfg, model, seed=3)
    m = make_hip_model(model, cfg, sd, precision=precision)
    o = orc.MatcherOracle(cfg, sd, model)
    shapes = [(1, 1), (2, 3), (5, 1), (31, 33), (32, 64), (63, 65), (64, 127), (129, 128), (255, 257), (300, 17)]
    for k, (n0, n1) in enumerate(shapes):
        pair = synthetic.make_correlated_pair(n0, n1, seed=70 + k, batch=2 if k % 3 == 0 else 1)
        data = {kk: torch.from_numpy(v).to(DEV) for kk, v in pair.items() if kk != 'image_shape'}
        data['image0'] = data['image1'] = torch.zeros(pair['image_shape'], device=DEV)
        with torch.no_grad():
            got = m.produce_matches(data, p=0.2, only_last=True)
            ref = o.produce_matches({kk: v.cpu() for kk, v in data.items()}, p=0.2, only_last=True)
        compare_matches(_cpu(got['indices0'][-1]), _cpu(got['mscores0'][-1]), ref['indices0'][-1].numpy(),
                        ref['mscores0'][-1].numpy(), 0.2, TOL, f'{model} {precision} n0={n0} n1={n1}')
        assert torch.isfinite(got['mscores0'][-1]).all()
