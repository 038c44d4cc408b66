"""GPU: the whole hot path through the reference-shaped boundary (GM / DGNNS / AdaGMN over the C-ABI)
against (a) the golden vectors captured from the imported reference, (b) the oracle on the same seeded
inputs, and (c) size-independent properties at the BASELINE sizes (N = 2048) where the oracle is slow.

Parity bar (BASELINE.json north_star): match indices identical, scores within 1e-4."""
import numpy as np
import pytest
import torch

from helpers import build_case, compare_matches, eval_config, golden_names, lib_options, load_golden, make_hip_model
from imp_release_amd import eval_loop, matching as hip_matching, synthetic
from oracle import imp_oracle as orc

pytestmark = pytest.mark.gpu
DEV = 'cuda'
TOL = 1e-4


def _cpu(x):
    return x.detach().cpu()


@pytest.mark.parametrize('precision', ['f16x3', 'f32'])
@pytest.mark.parametrize('name', golden_names(['gm_l', 'dgnns_l', 'adagmn_masked', 'gm_trained', 'dgnns_trained']))
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
        # (relative for the dustbin row, whose N + 1 entries sum to ~N)
        assert (np.abs(s.sum(-1).numpy() - z['score_rowsum']) <= 5e-4 * np.maximum(1.0, np.abs(z['score_rowsum']))).all()
        assert np.abs(s[:8, :8].numpy() - z['score_corner']).max() < TOL
    print('\n'.join(msgs))


COND_REPORT = []


@pytest.mark.parametrize('fixture', ['conditioning_n1024', 'conditioning_n2048'])
@pytest.mark.parametrize('precision', ['f16x3', 'f32'])
def test_parity_under_conditioning(precision, fixture):
    """VERDICT r4 #5a: parity as the network gets worse conditioned, measured against where the REFERENCE stops defining the answer.
    tests/golden/conditioning_n1024.npz (tools/parity_vs_conditioning.py, which imports the reference) holds, for q / k gains 1 ... 5 of
    the trained-style weights at N = 1024 (GM, L = 9, T = 100): the fp64 run of the reference and the deviation of the reference's OWN fp32
    evaluations (1 thread, 8 threads) from it.  At every gain the HIP path must disagree with the fp64 yardstick no more than the reference's
    fp32 does: match indices - no more disagreements than the larger of the reference's two counts + the keypoints whose mscore an fp32
    evaluation of the reference itself moves by more than 1e-3 (0 + 0 up to gain 3: strict there);
    scores - within max(1e-4, 2 x the reference's larger fp32 deviation) (up to gain 3 the reference's noise is < 1e-4 and the bar is the
    north star's 1e-4; at gain 4 two fp32 evaluations of the reference differ by 1e-3, at gain 5 by 0.1 - no implementation can be held to
    1e-4 there).  This measurement replaces the prose justification of the `low_score_flips` tolerance.
    Round 6: the same sweep at the HEADLINE size, N = 2048, gains 1 / 2 / 3 (conditioning_n2048.npz) - the 1e-4 margin there is thin
    (two fp32 CPU evaluations, reference and oracle, differ by 2.4e-4 in mscores0 on gm_l9_t100_n2048_b4) and is now measured, not met by luck."""
    spec, z = load_golden(fixture)
    cfg = eval_config(**spec['config'])
    pair = synthetic.make_correlated_pair(spec['n'], spec['n'], seed=spec['dseed'])
    data = {k: torch.from_numpy(v).to(DEV) for k, v in pair.items() if k != 'image_shape'}
    data['image0'] = data['image1'] = torch.zeros(pair['image_shape'], device=DEV)
    rows = []
    for g in spec['gains']:
        tag = f'g{int(round(g * 10)):02d}'
        sd = synthetic.make_state_dict(cfg, 'GM', seed=spec['wseed'], style=spec['style'], qk_gain=g)
        m = make_hip_model('GM', cfg, sd, precision=precision)
        with torch.no_grad():
            out = m.produce_matches(data, **spec['call'])
        i_hip, ms_hip = _cpu(out['indices0'][-1])[0].numpy(), _cpu(out['mscores0'][-1])[0].double().numpy()
        i64, ms64 = z[f'{tag}_indices0'], z[f'{tag}_mscores0']
        noise = z[f'{tag}_ref_noise']
        bad = int((i_hip != i64).sum())
        agree = i_hip == i64
        dms = float(np.abs(ms_hip - ms64)[agree].max(initial=0.0))
        ref_bad, ref_dms = int(max(noise[0], noise[2])) + int(noise[6]), float(max(noise[1], noise[3]))
        rows.append((g, bad, dms, ref_bad, ref_dms))
        del m
    COND_REPORT.append((precision, rows))
    print(f'conditioning sweep at N = {spec["n"]} ({precision}): q/k gain | HIP vs fp64: idx, max|dms| | reference fp32 vs fp64 (worse of 1 / 8 threads): idx, max|dms|')
    for g, bad, dms, rb, rd in rows:
        print(f'  {g:4.1f} | {bad:4d} {dms:9.2e} | {rb:4d} {rd:9.2e}')
    for g, bad, dms, rb, rd in rows:
        assert bad <= rb, f'gain {g}: {bad} index disagreements with the fp64 yardstick, the reference itself has {rb}'
        assert dms <= max(TOL, 2.0 * rd), f'gain {g}: max|dmscore| {dms:.2e} against the fp64 yardstick; the reference fp32 deviates by {rd:.2e}'


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


LOOPS = [('imp_loop_n400', False), ('imp_loop_exit_n400', False), ('eimp_loop_sliced_n1024', True),
         ('eimp_loop_uncert_exit_n1024', True), ('eimp_loop_uncert_full_n700', True),
         ('eimp_loop_trained_n1024', True),       # trained-like weights (synthetic style='trained'): peaky attention
         ('eimp_loop_sliced_n4096', True)]        # BASELINE configs[3]: N = 4096 / 4000 -> 2436 / 2403, captured from the reference (round 4)


def _check_loop_against_golden(name, z, data, ret, trace, stub):
    """ret = (pts0, pts1, indices0, mscores0, R, t, n_iter) of a loop run; everything strict (bit-exact indices)"""
    p0, p1, i0, ms0, R, t, nit = ret
    assert nit == int(z['n_iter']), f'{name}: n_iter {nit} vs reference {int(z["n_iter"])}'
    traj = [(t_['n0'], t_['n1']) for t_ in trace]
    assert traj == [tuple(r) for r in z['trajectory'].tolist()], f'{name}: pruning trajectory {traj}'
    k0, k1 = data['keypoints0'][0].cpu().numpy(), data['keypoints1'][0].cpu().numpy()
    for k, t_ in enumerate(trace):
        assert np.array_equal(t_['pts0'], k0[z[f'it{k}_keep0']]) and np.array_equal(t_['pts1'], k1[z[f'it{k}_keep1']]), \
            f'{name}: keep set it{k}'
        # trained-like weights (peaky attention): one or two UNMATCHED keypoints with scores ~0.01 < p sit on exact mutual-nearest-
        # neighbour ties that flip between any two fp32 evaluations (reference fp32 vs fp64 shows the same, synthetic.make_state_dict);
        # indices stay identical.  Tolerated there, counted in the terminal summary; every other fixture: none
        compare_matches(t_['indices0'], t_['mscores0'], z[f'it{k}_indices0'], z[f'it{k}_mscores0'], 0.1, TOL,
                        f'{name} it{t_["it"]}', low_score_flips=2 if 'trained' in name else 0)
    assert np.array_equal(p0, z['pts0_final']) and np.array_equal(p1, z['pts1_final']), f'{name}: final keypoint sets'
    # the returned indices: after an early exit they are the inlier-filtered matches (eval/matching.py:112-113)
    assert np.array_equal(i0, z['indices0']), f'{name}: returned indices ({(i0 != z["indices0"]).sum()} differ)'
    assert np.abs(ms0.astype(np.float64) - z['mscores0']).max() <= TOL
    if 'R' in z.files:
        assert R is not None and np.allclose(R, z['R']) and np.allclose(t, z['t'])
        assert [c[0] for c in stub.calls] == z['pose_calls'].tolist(), f'{name}: matches handed to the pose step'
    else:
        assert R is None and t is None


@pytest.mark.parametrize('precision', ['f16x3', 'f32'])
@pytest.mark.parametrize('name,unc', LOOPS)
def test_iterative_loops_vs_golden(name, unc, precision):
    """eval/matching.py:16-123 (IMP) and :126-276 (EIMP, real ragged slicing = BASELINE config 4 analogue), pinned to
    fixtures captured from the reference: the no-pose trajectory (all 15 iterations), and - driven by the same
    deterministic PoseStub the reference was driven by - the pose-change early exit with inlier-filtered indices
    (:84-117) and the with_uncertainty pool thresholds (:243-252)."""
    spec, z = load_golden(name)
    cfg, sd, data = build_case(spec, DEV)
    m = make_hip_model(spec, cfg, sd, precision=precision)
    sched = spec.get('pose_schedule')
    stub = synthetic.PoseStub(sched) if sched is not None else None
    trace = []
    with torch.no_grad():
        if unc:
            p0, p1, nk0, nk1, i0, ms0, R, t, nit = hip_matching.matching_iterative_uncertainty(
                _loop_data(data), m, 15, 0.1, 25, 1.0, {'pose': 1.5}, method=38,
                with_uncertainty=bool(spec.get('with_uncertainty', False)), estimate_pose=stub, trace=trace)
            assert nk0.shape == (p0.shape[0], 2) and nk1.shape == (p1.shape[0], 2)
        else:
            ld = _loop_data(data)
            i0, ms0, R, t, nit = hip_matching.matching_iterative(ld, m, 15, 0.1, 25, 1.0, {'pose': 1.5}, method=38,
                                                                 estimate_pose=stub, trace=trace)
            p0, p1 = ld['pts0_cpu'], ld['pts1_cpu']
    _check_loop_against_golden(name, z, data, (p0, p1, i0, ms0, R, t, nit), trace, stub)


def _reference_call_sequence_eimp(data, model, stub, with_uncertainty, trace):
    """The call sequence a drop-in user's eval/matching.py:126-276 performs, through the PUBLIC module API only and on
    the reference's [B, D, N] layout: encode_keypoint -> per iteration (slice desc[:, :, sel_ids] / norm_kpts) ->
    forward_one_layer(M0=None, M1=None) x2 -> self_prob*/cross_prob* attributes -> compute_distance ->
    compute_score(dustbin=model.bin_score) -> compute_matches -> pose -> model.pool(pred_score=..., prob00=...).
    Written for this test (own bookkeeping); what matters is that nothing below touches ctx.* or pool_host."""
    from imp_release_amd.modules import normalize_keypoints           # = `from nets.gm import normalize_keypoints`
    nk0 = normalize_keypoints(kpts=data['keypoints0'], image_shape=data['image0'].shape)
    nk1 = normalize_keypoints(kpts=data['keypoints1'], image_shape=data['image1'].shape)
    d0, d1 = data['descriptors0'].transpose(1, 2), data['descriptors1'].transpose(1, 2)
    pts0, pts1 = data['keypoints0'][0].cpu().numpy(), data['keypoints1'][0].cpu().numpy()
    e0, e1 = model.encode_keypoint(norm_kpts0=nk0, norm_kpts1=nk1, scores0=data['scores0'], scores1=data['scores1'])
    d0, d1 = d0 + e0, d1 + e1
    sel0 = sel1 = None
    last = None
    score = None
    for it in range(15):
        if sel0 is not None:
            d0, pts0, nk0 = d0[:, :, sel0], pts0[sel0.cpu().numpy()], nk0[:, sel0, :]
        if sel1 is not None:
            d1, pts1, nk1 = d1[:, :, sel1], pts1[sel1.cpu().numpy()], nk1[:, sel1.cpu(), :]
        sel0 = sel1 = None
        d0, d1 = model.forward_one_layer(desc0=d0, desc1=d1, M0=None, M1=None, layer_i=it * 2)
        d0, d1 = model.forward_one_layer(desc0=d0, desc1=d1, M0=None, M1=None, layer_i=it * 2 + 1)
        if it not in (3, 5, 7, 9, 11, 13, 14):
            continue
        prob00, prob11, prob01, prob10 = model.self_prob0, model.self_prob1, model.cross_prob0, model.cross_prob1
        dist = model.compute_distance(desc0=d0, desc1=d1, layer_id=it)
        score = model.compute_score(dist=dist, dustbin=model.bin_score, iteration=model.sinkhorn_iterations)
        i0, i1, m0, m1 = model.compute_matches(scores=score, p=0.1)
        i0c, m0c = i0[0].cpu().numpy(), m0[0].cpu().numpy()
        trace.append({'it': it, 'n0': d0.shape[2], 'n1': d1.shape[2], 'indices0': i0c, 'mscores0': m0c,
                      'pts0': pts0, 'pts1': pts1})
        if torch.sum(i0 > -1) < 25:
            last = None
            continue
        a = np.nonzero(i0c > -1)[0]
        pm = np.stack([a, i0c[a]], 1)
        ret = stub(kpts0=pts0[pm[:, 0]], kpts1=pts1[pm[:, 1]], K0=np.eye(3), K1=np.eye(3), norm_thresh=1.0, method=38) \
            if stub is not None else None
        if ret is None:
            R = t = None
            inl, ratio = np.zeros(len(pm), bool), 0
        else:
            _, R, t, inl = ret
            ratio = inl.sum() / len(pm)
        diff = np.inf
        if last is not None and R is not None:
            diff = max(hip_matching.angle_error_mat(last[0], R), hip_matching.angle_error_vec(last[1], t))
        last = None if R is None else (R, t)
        th = 0.2 * ratio if (with_uncertainty and ratio != 0) else 0.2
        sel0, sel1 = model.pool(pred_score=score, prob00=prob00, prob01=prob01, prob11=prob11, prob10=prob10,
                                mscore_th=th, uncertainty_ratio=1.0)
        if diff <= 1.5:
            out = np.zeros_like(i0c) - 1
            out[pm[inl, 0]] = pm[inl, 1]
            return pts0, pts1, out, m0c, R, t, it + 1
    i0, i1, m0, m1 = model.compute_matches(scores=score, p=0.2)
    return pts0, pts1, i0[0].cpu().numpy(), m0[0].cpu().numpy(), None, None, 15


@pytest.mark.parametrize('name', ['eimp_loop_sliced_n1024', 'eimp_loop_uncert_exit_n1024'])
def test_public_module_api_replay_of_the_eimp_loop(name):
    """"eval/matching.py runs unchanged" shown, not asserted: the reference's call sequence through the public module
    API only ([B, D, N] views, desc[:, :, sel_ids] slices that are NOT token-major contiguous, AttentionHandles read
    BEFORE compute_distance and handed to model.pool) reproduces the reference-captured fixture, and therefore
    the library's own fused loop (imp_release_amd.matching), bit for bit."""
    spec, z = load_golden(name)
    cfg, sd, data = build_case(spec, DEV)
    m = make_hip_model(spec, cfg, sd)
    sched = spec.get('pose_schedule')
    stub = synthetic.PoseStub(sched) if sched is not None else None
    trace = []
    with torch.no_grad():
        ret = _reference_call_sequence_eimp(data, m, stub, bool(spec.get('with_uncertainty', False)), trace)
    _check_loop_against_golden(name, z, data, ret, trace, stub)
    stub2 = synthetic.PoseStub(sched) if sched is not None else None
    with torch.no_grad():
        own = hip_matching.matching_iterative_uncertainty(_loop_data(data), m, 15, 0.1, 25, 1.0, {'pose': 1.5}, method=38,
                                                          with_uncertainty=bool(spec.get('with_uncertainty', False)),
                                                          estimate_pose=stub2)
    assert np.array_equal(own[4], ret[2]) and np.array_equal(own[5], ret[3]) and own[8] == ret[6]


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
    # round 5: the handles behave like the [B, 4, N, M] tensors of nets/gm.py:272-283 under tensor arithmetic (rows of a softmax sum to 1)
    rows = torch.sum(m.cross_prob0, dim=-1)
    assert rows.shape == (1, 4, 190) and torch.allclose(rows, torch.ones_like(rows), atol=1e-5) and float((m.self_prob1 * 2).max()) <= 2.0 + 1e-5
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


def test_full_size_vs_the_reference(big):
    """BASELINE.json's metric configuration (GM, N = M = 2048, 9 iterations, 100 Sinkhorn) against a fixture captured from the
    imported reference on exactly these seeds (round 4; rounds 1-3 compared with the oracle, non-strict): STRICT - indices
    bit-identical on both pairs of the batch, scores within 1e-4"""
    cfg, sd, m, pair, data, out = big
    spec, z = load_golden('gm_l9_t100_n2048_b2')
    assert (spec['wseed'], spec['dseed'], spec['batch'], spec['n0']) == (1, 31, 2, 2048)
    print(compare_matches(_cpu(out['indices0'][-1]), _cpu(out['mscores0'][-1]), z['indices0_0'], z['mscores0_0'], 0.2, TOL,
                          'N=2048 L=9 T=100 B=2 vs reference fixture'))


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
    # all four pairs against the fixture captured from the imported reference on these seeds (round 4): strict
    spec, z = load_golden('gm_l9_t100_n2048_b4')
    assert (spec['wseed'], spec['dseed'], spec['batch']) == (1, 77, 4)
    print(compare_matches(_cpu(out4['indices0'][-1]), _cpu(out4['mscores0'][-1]), z['indices0_0'], z['mscores0_0'], 0.2, TOL,
                          'N=2048 L=9 T=100 B=4 vs reference fixture'))


@pytest.mark.parametrize('seed', [101, 102, 103, 104])
def test_full_size_more_pairs_vs_the_reference(seed):
    """the BASELINE workload (N=2048, L=9, T=100) on further seeded pairs with their own weights, ragged second image, against
    fixtures captured from the imported reference: strict"""
    spec, z = load_golden(f'gm_l9_t100_n2048_s{seed}')
    cfg, sd, data = build_case(spec, DEV)
    assert data['keypoints1'].shape[1] == 2048 - 3 * (seed % 7)
    m = make_hip_model('GM', cfg, sd)
    with torch.no_grad():
        out = m.produce_matches(data, p=0.2, only_last=True)
    print(compare_matches(_cpu(out['indices0'][-1]), _cpu(out['mscores0'][-1]), z['indices0_0'], z['mscores0_0'], 0.2, TOL,
                          f'N=2048 seed {seed} vs reference fixture'))


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
    """imp_match_pair never synchronises or allocates once the workspace is sized: the whole pair can be captured in a HIP
    graph and replayed.  Since round 3 a capture records the chip-resident Sinkhorn too - on the capturing stream itself, with its
    exchange tags and XCC tickets taken from device memory and advanced by the launch's last workgroup, so that every replay
    exchanges under fresh tags: five replays, each bitwise equal to the eager resident call, then an eager call,
    then a replay again (both kinds of launch share the exchange buffers).  option ot_graph = 0 keeps the round-2 behaviour (the capture
    records the streaming kernels): bitwise equal to the eager streaming path."""
    import os
    cfg = eval_config(n_layers=3)
    sd = synthetic.make_state_dict(cfg, 'DGNNS', seed=2)
    m = make_hip_model('DGNNS', cfg, sd)
    os.environ['IMP_OT_RESIDENT'] = '0'
    try:
        ms = make_hip_model('DGNNS', cfg, sd)
        ms._ensure_ctx()
    finally:
        del os.environ['IMP_OT_RESIDENT']
    with lib_options(ot_graph=0):
        mg0 = make_hip_model('DGNNS', cfg, sd)
        mg0._ensure_ctx()
    pair = synthetic.make_correlated_pair(300, 280, seed=9, batch=2)
    d = {k: torch.from_numpy(v).to(DEV) for k, v in pair.items() if k != 'image_shape'}
    args = (d['keypoints0'], d['scores0'], d['descriptors0'], d['keypoints1'], d['scores1'], d['descriptors1'],
            640., 480., 1.0, 20, True, 0.2)
    eager_stream = ms._ensure_ctx().match_pair(*args, want_side1=True)
    for model, want, what in ((m, None, 'resident'), (mg0, eager_stream, 'streaming')):
        ctx = model._ensure_ctx()
        eager = ctx.match_pair(*args, want_side1=True)
        want = eager if want is None else want
        out = {k: torch.zeros_like(v) for k, v in eager.items()}
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            ctx.match_pair(*args, out=out)

        def replay_and_check():
            for v in out.values():
                v.zero_()
            g.replay()
            torch.cuda.synchronize()
            for k in want:
                assert torch.equal(out[k], want[k]), (what, k)
        for _ in range(5):
            replay_and_check()
        again = ctx.match_pair(*args, want_side1=True)              # an ordinary launch between replays
        torch.cuda.synchronize()
        for k in eager:
            assert torch.equal(again[k], eager[k]), (what, k)
        replay_and_check()
        assert ctx.resident_health(raise_on_timeout=False) is not False
        compare_matches(_cpu(out['indices0']), _cpu(out['mscores0']), _cpu(eager_stream['indices0']).numpy(),
                        _cpu(eager_stream['mscores0']).numpy(), 0.2, 1e-5, f'graph replay ({what}) vs eager streaming')


def test_graph_replays_survive_the_tag_wrap():
    """the replayed resident launches take their exchange tags from a device-side counter in the upper half of the 32-bit
    space; after ~7 million replays it starts over.  The launch that wraps it tells the library (word 2 of the mapped health page), and
    the next entry point clears the exchange buffers so that no old tag can be met again.  option ot_graph_tag0 (test hook) starts the
    counter two launches short of the wrap: replays before, across and after it - and eager calls in between - stay bitwise equal."""
    import os
    cfg = eval_config(n_layers=3)
    sd = synthetic.make_state_dict(cfg, 'DGNNS', seed=2)
    with lib_options(ot_graph_tag0=hex(0xFFFFF000 - 2 * (3 * 20 + 4) - 8)):
        m = make_hip_model('DGNNS', cfg, sd)
        ctx = m._ensure_ctx()
    pair = synthetic.make_correlated_pair(300, 280, seed=9, batch=2)
    d = {k: torch.from_numpy(v).to(DEV) for k, v in pair.items() if k != 'image_shape'}
    args = (d['keypoints0'], d['scores0'], d['descriptors0'], d['keypoints1'], d['scores1'], d['descriptors1'],
            640., 480., 1.0, 20, True, 0.2)
    eager = ctx.match_pair(*args, want_side1=True)
    out = {k: torch.zeros_like(v) for k, v in eager.items()}
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ctx.match_pair(*args, out=out)
    assert ctx.tag_wraps() == 0
    for i in range(8):
        for v in out.values():
            v.zero_()
        g.replay()
        torch.cuda.synchronize()
        for k in eager:
            assert torch.equal(out[k], eager[k]), (i, k)
        if i % 3 == 2:
            again = ctx.match_pair(*args, want_side1=True)              # an entry point: notices the wrap, clears the buffers
            torch.cuda.synchronize()
            for k in eager:
                assert torch.equal(again[k], eager[k]), (i, k)
    assert ctx.resident_health() == (0, 0)
    assert ctx.tag_wraps() == 1, ctx.tag_wraps()


WF_FIXTURES = ['gm_l3_alliters_b2', 'gm_l3_bigmean', 'gm_l9_t100_ragged', 'dgnns_l5_alliters', 'adagmn_masked_l9']


@pytest.mark.parametrize('name', WF_FIXTURES)
def test_weight_fragment_gemms_forced_vs_golden(name):
    """csrc/gemm_wf.hip takes over the layer convolutions only for launches that cover the chip (option gemm_wf = 1, default); forced
    on everywhere (=2) it has to reproduce the reference fixtures like the default kernels do: ragged row counts, batch 2,
    |mean| >> std channels in front of the InstanceNorm (its per-block (sum, M2) statistics), attention-sharing layers"""
    import os
    spec, z = load_golden(name)
    cfg, sd, data = build_case(spec, device=DEV)
    with lib_options(gemm_wf=2, wf_chain_min=1):        # ... and the chained launch (conv 3 + the next layer's projection) at every size
        m = make_hip_model(spec, cfg, sd)
        m._ensure_ctx()
    with torch.no_grad():
        out = m.produce_matches(data, **spec.get('call', {}))
    for i in range(int(z['n_emitted'])):
        print(compare_matches(_cpu(out['indices0'][i]), _cpu(out['mscores0'][i]), z[f'indices0_{i}'], z[f'mscores0_{i}'],
                              spec.get('call', {}).get('p', 0.2), TOL, f'{name}[{i}] wf forced'))


@pytest.mark.parametrize('model,n0,n1,B', [('GM', 2048, 2048, 4), ('DGNNS', 1000, 1100, 3), ('GM', 130, 97, 2), ('DGNNS', 64, 64, 1), ('GM', 1, 5, 1)])
def test_chained_projection_is_bit_identical_to_the_separate_launches(model, n0, n1, B):
    """gemm_wf.hip CHAIN computes layer i + 1's q|k|v (value only for a sharing layer) from the tile layer i's last convolution
    just produced: the same arithmetic on the same operands as the separate projection launch, so the matches must be IDENTICAL
    bit for bit (both with the weight-fragment kernels forced on, so that the projection kernel is the same one)"""
    import os
    cfg = eval_config(n_layers=5 if model == 'DGNNS' else 3, sinkhorn_iterations=20)
    sd = synthetic.make_state_dict(cfg, model, seed=8)
    models = []
    for chain in (1, 0):
        with lib_options(gemm_wf=2, wf_chain=chain, wf_chain_min=1):
            m = make_hip_model(model, cfg, sd)
            m._ensure_ctx()
            models.append(m)
    pair = synthetic.make_correlated_pair(n0, n1, seed=n0 + B, batch=B)
    data = {k: torch.from_numpy(v).to(DEV) for k, v in pair.items() if k != 'image_shape'}
    data['image0'] = data['image1'] = torch.zeros(pair['image_shape'], device=DEV)
    with torch.no_grad():
        a = models[0].produce_matches(data, p=0.2, only_last=True)
        b = models[1].produce_matches(data, p=0.2, only_last=True)
    assert torch.equal(a['indices0'][-1], b['indices0'][-1]) and torch.equal(a['mscores0'][-1], b['mscores0'][-1])


def _two_models(model, cfg, sd, opts_a, opts_b):
    out = []
    for opts in (opts_a, opts_b):
        with lib_options(**opts):
            m = make_hip_model(model, cfg, sd)
            m._ensure_ctx()
            out.append(m)
    return out


@pytest.mark.parametrize('model,n0,n1,B', [('GM', 2048, 2048, 4), ('GM', 2048, 1990, 4), ('DGNNS', 1000, 1100, 3), ('GM', 130, 97, 2), ('AdaGMN', 420, 400, 1),
                                           ('DGNNS', 1500, 1400, 2)])
def test_fused_layer_launch_is_bit_identical_to_the_two_launch_path(model, n0, n1, B):
    """round 4: a layer's MLP0 -> InstanceNorm -> MLP3 (-> next projection) as ONE launch with an in-kernel statistics exchange
    (gemm_wf.hip gemm_wf_fused_kernel) against the round-3 path (MLP0 with last-arriver statistics, then MLP3 + chain): the same
    arithmetic on the same operands in the same order - matches, match scores and the whole score tensor must agree bit for bit.
    Forced on for the small shapes (option wf_fused_min = 1; attention-sharing layers, the masked AdaGMN loop, ragged images)."""
    cfg = eval_config(n_layers=5 if model != 'GM' else 3, sinkhorn_iterations=20)
    sd = synthetic.make_state_dict(cfg, model, seed=8, bin_score=5.0 if model == 'AdaGMN' else 1.0)
    base = {'gemm_wf': 2, 'wf_chain_min': 1}
    fused, plain = _two_models(model, cfg, sd, dict(base, wf_fused=1, wf_fused_min=1), dict(base, wf_fused=0))
    pair = synthetic.make_correlated_pair(n0, n1, seed=n0 + B, batch=B)
    data = {k: torch.from_numpy(v).to(DEV) for k, v in pair.items() if k != 'image_shape'}
    data['image0'] = data['image1'] = torch.zeros(pair['image_shape'], device=DEV)
    kw = dict(p=0.2) if model == 'AdaGMN' else dict(p=0.2, only_last=True)
    with torch.no_grad():
        a = fused.produce_matches(data, **kw)
        b = plain.produce_matches(data, **kw)
    assert int((a['indices0'][-1] >= 0).sum()) > 0 and torch.isfinite(a['mscores0'][-1]).all()
    assert torch.equal(a['indices0'][-1], b['indices0'][-1]) and torch.equal(a['mscores0'][-1], b['mscores0'][-1])
    if a.get('scores'):
        assert torch.equal(a['scores'][-1], b['scores'][-1])
    assert fused._ensure_ctx().resident_health() == (0, 0)


def test_fused_layer_time_out_voids_the_call_and_steps_the_context_down():
    """option wf_fused_fake = 1 (test hook): one workgroup of every fused launch withholds its statistics, so every wait on them times
    out.  The call still ends (bounded polls), the pair whose exchange failed comes back VOID (NaN descriptors -> no matches - never plausible numbers), the next
    entry point on the context reports it (IMP_E_RESIDENT -> ResidentSinkhornTimeout), and from then on the context runs the
    two-launch path: identical to a context that never used the fused kernel."""
    from imp_release_amd import _lib
    cfg = eval_config(n_layers=3, sinkhorn_iterations=20)
    sd = synthetic.make_state_dict(cfg, 'GM', seed=8)
    fake, plain = _two_models('GM', dict(cfg, range_recovery=False), sd, {'wf_fused_fake': 1}, {'wf_fused': 0})      # (range_recovery=False: the never-waiting library; the default repairs the call - next test)
    pair = synthetic.make_correlated_pair(2048, 2048, seed=5, batch=4)
    data = {k: torch.from_numpy(v).to(DEV) for k, v in pair.items() if k != 'image_shape'}
    data['image0'] = data['image1'] = torch.zeros(pair['image_shape'], device=DEV)
    with torch.no_grad():
        void = fake.produce_matches(data, p=0.2, only_last=True)
        torch.cuda.synchronize()
        # the withheld record belongs to pair 0, image 0: that pair's descriptors are NaN from the first layer on -> no match can pass
        # (the other pairs of the batch never waited on it)
        assert bool((void['indices0'][-1][0] == -1).all()) and not bool((void['mscores0'][-1][0] > 0).any())
        with pytest.raises(_lib.ResidentSinkhornTimeout):
            fake.produce_matches(data, p=0.2, only_last=True)
        good = fake.produce_matches(data, p=0.2, only_last=True)
        want = plain.produce_matches(data, p=0.2, only_last=True)
    assert torch.equal(good['indices0'][-1], want['indices0'][-1]) and torch.equal(good['mscores0'][-1], want['mscores0'][-1])
    assert fake._ensure_ctx().resident_health()[0] == 1


def test_fused_layer_time_out_is_repaired_inside_the_same_call():
    """round 6 (VERDICT r5 #2a): default configuration (in-call recovery on).  The call whose fused layer launch timed out waits for its own
    work, sees the word, lets the context step down to the two-launch layers and enqueues its work again: the SAME call returns what a
    context without the fused kernel returns, bit for bit - one-shot call, composed pass (all iterations) and the step API alike."""
    cfg = eval_config(n_layers=3, sinkhorn_iterations=20)
    sd = synthetic.make_state_dict(cfg, 'GM', seed=8)
    pair = synthetic.make_correlated_pair(2048, 2048, seed=5, batch=4)
    data = {k: torch.from_numpy(v).to(DEV) for k, v in pair.items() if k != 'image_shape'}
    data['image0'] = data['image1'] = torch.zeros(pair['image_shape'], device=DEV)
    for only_last in (True, False):
        fake, plain = _two_models('GM', cfg, sd, {'wf_fused_fake': 1}, {'wf_fused': 0})
        with torch.no_grad():
            got = fake.produce_matches(data, p=0.2, only_last=only_last)
            want = plain.produce_matches(data, p=0.2, only_last=only_last)
        assert len(got['indices0']) == (1 if only_last else 3)
        for a, b in zip(got['indices0'] + got['mscores0'] + got['scores'], want['indices0'] + want['mscores0'] + want['scores']):
            assert torch.equal(a, b)
        ctx = fake._ensure_ctx()
        assert ctx.resident_health()[0] == 1
        pm = ctx.resident_postmortem()
        assert pm is not None and pm['kind'] == 3 and pm['phase'] in (1, 2) and pm['fused_layers_on'] == 1, pm
        if only_last:
            assert ctx.resident_repaired() == 1
            print('post-mortem of the withheld statistics record:', pm)
    # step API: the layer call itself is repaired
    fake, plain = _two_models('GM', cfg, sd, {'wf_fused_fake': 1}, {'wf_fused': 0})
    outs = []
    for m in (fake, plain):
        with torch.no_grad():
            ctx = m._ensure_ctx(check=True)
            e0, e1 = m.encode_keypoint(ctx.normalize_keypoints(data['keypoints0'], 640.0, 480.0), ctx.normalize_keypoints(data['keypoints1'], 640.0, 480.0),
                                       data['scores0'], data['scores1'])
            d0, d1 = data['descriptors0'].transpose(1, 2) + e0, data['descriptors1'].transpose(1, 2) + e1
            for li in range(2):
                d0, d1 = m.forward_one_layer(d0, d1, None, None, li)
            outs.append((d0.clone(), d1.clone()))
    assert torch.isfinite(outs[0][0]).all() and torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert fake._ensure_ctx().resident_health()[0] == 1


def test_weight_fragment_gemms_default_rule_agrees_with_gemm_f32():
    """B = 4, N = 2048 (the bench shape) takes the weight-fragment MLP kernels by default: same matches as with them switched off"""
    import os
    cfg = eval_config(n_layers=3, sinkhorn_iterations=20)
    sd = synthetic.make_state_dict(cfg, 'GM', seed=3)
    on = make_hip_model('GM', cfg, sd)
    with lib_options(gemm_wf=0):
        off = make_hip_model('GM', cfg, sd)
        off._ensure_ctx()
    pair = synthetic.make_correlated_pair(2048, 2048, seed=77, batch=4)
    data = {k: torch.from_numpy(v).to(DEV) for k, v in pair.items() if k != 'image_shape'}
    data['image0'] = data['image1'] = torch.zeros(pair['image_shape'], device=DEV)
    with torch.no_grad():
        a = off.produce_matches(data, p=0.2, only_last=True)
        b = on.produce_matches(data, p=0.2, only_last=True)
    print(compare_matches(_cpu(b['indices0'][-1]), _cpu(b['mscores0'][-1]), _cpu(a['indices0'][-1]).numpy(),
                          _cpu(a['mscores0'][-1]).numpy(), 0.2, TOL, 'weight-fragment MLP kernels vs gemm_f32', strict=False))


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
    col = {n: table[:, i] for i, n in enumerate(eval_loop.SUMMARY_COLUMNS)}
    assert table.shape == (3, len(eval_loop.SUMMARY_COLUMNS)) and (col['n_iterations'] == 15).all()
    assert (col['n_matches'] > 25).all() and (col['n_kept0'] < 640).all() and (col['n_kept0'] > 0).all()
    again = eval_loop.run_pairs_sharded(m, provider, 3, eimp=True)
    assert np.array_equal(table, again, equal_nan=True)          # deterministic


@pytest.mark.parametrize('eimp', [False, True])
def test_evaluation_tail_on_two_view_pairs_with_the_gpu_pose_step(eimp):
    """BASELINE config 5 end to end on one rank: two-view-consistent synthetic pairs -> iterative loop with the GPU pose step in its
    estimate_pose slot -> per-pair (err_R, err_t, precision, ...) rows -> the report of eval/eval_imp.py:213-227; pairs in flight
    give the same table as the sequential loop"""
    from imp_release_amd import pose as gpose
    name = 'AdaGMN' if eimp else 'DGNNS'
    cfg = eval_config()
    sd = synthetic.make_state_dict(cfg, name, seed=9, bin_score=synthetic.MATCHING_BIN_SCORE, style='matching')
    m = make_hip_model(name, cfg, sd)

    def provider(pid):
        pair = synthetic.make_two_view_pair(700, 660, seed=300 + pid)
        d = {k: torch.from_numpy(pair[k]).to(DEV) for k in ('keypoints0', 'keypoints1', 'scores0', 'scores1', 'descriptors0', 'descriptors1')}
        d['image0'] = d['image1'] = torch.zeros(pair['image_shape'], device=DEV)
        d['pts0_cpu'], d['pts1_cpu'] = pair['keypoints0'][0], pair['keypoints1'][0]
        d.update({k: pair[k] for k in ('K0', 'K1', 'T_0to1', 'E')})
        return d

    seq = eval_loop.run_pairs_sharded(m, provider, 6, eimp=eimp, estimate_pose=gpose.estimate_pose)
    reps = eval_loop.replicate(m, 3)
    par = eval_loop.run_pairs_sharded(m, provider, 6, eimp=eimp, estimate_pose=gpose.estimate_pose, workers=3, replicas=reps)
    assert np.array_equal(seq, par, equal_nan=True)
    rep = eval_loop.aggregate(seq)
    print(name, rep)
    assert rep['pairs'] == 6 and not np.isnan(seq[:, :4]).any()
    # style='matching' weights pair up the re-observed keypoints (near-identical descriptors), so the final matches are mostly
    # epipolar-consistent and the pose is recovered
    assert rep['precision'] > 60.0 and rep['pose_found'] == 1.0 and rep['auc@20'] > 50.0


@pytest.mark.parametrize('n0,n1,B,fused', [(1024, 1000, 2, '1'), (2048, 2048, 4, '1'), (2048, 2048, 4, '2')])
def test_batch_steps_in_flight_reproduce_the_sequential_results(n0, n1, B, fused, monkeypatch):
    """bench.py's default mode: 3 batch-steps in flight (3 replicas, 3 streams) - every step's result must equal the
    one-after-the-other result bit for bit.  At the bench geometry (4 x 2048) every layer is a fused launch whose workgroups wait for
    each other and the Sinkhorn is the chip-resident kernel: the three streams' waiting kernels are serialised by the spin gate
    (context.hip) - a collision would end in time-outs and NaN results.  By default a context leaves the fused layer launch alone while
    other streams share the gate (the two-launch layers interleave better); IMP_WF_FUSED=2 keeps it: then EVERY layer of the three
    replicas goes through the gate"""
    from imp_release_amd import pipeline
    monkeypatch.setenv('IMP_WF_FUSED', fused)
    cfg = eval_config(n_layers=3, sinkhorn_iterations=20)
    sd = synthetic.make_state_dict(cfg, 'GM', seed=1)
    m = make_hip_model('GM', cfg, sd)
    datas = []
    for k in range(4):
        pair = synthetic.make_correlated_pair(n0, n1, seed=60 + k, batch=B)
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
        assert torch.isfinite(a[1]).all() and int((a[0] >= 0).sum()) > 0
    assert not torch.equal(seq[0][0], seq[1][0])          # the steps really see different batches
    for r in [m] + reps:
        assert r._ensure_ctx().resident_health() == (0, 0)  # no waiting kernel of any replica ever timed out


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
    assert np.array_equal(seq, par, equal_nan=True), (seq, par)
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


@pytest.mark.parametrize('model,n0,n1,B', [('GM', 1024, 1000, 2), ('DGNNS', 300, 280, 1), ('GM', 150, 97, 3)])
def test_kv_split_half_images_do_not_change_the_matches(model, n0, n1, B):
    """round 3: the projections write K / V as the split-half image [hi | lo] the attention kernel used to build while staging.  The
    matrix pipe sees bit-identical operands either way, so a one-shot match is IDENTICAL with the round-2 format (option kv_image = 0);
    (sizes <= 192 queries additionally move from the lock-step kernels to the ping-pong kernel: another summation order)"""
    import os
    cfg = eval_config(n_layers=5 if model == 'DGNNS' else 3, sinkhorn_iterations=20)
    sd = synthetic.make_state_dict(cfg, model, seed=12)
    on = make_hip_model(model, cfg, sd)
    on._ensure_ctx()
    with lib_options(kv_image=0):
        off = make_hip_model(model, cfg, sd)
        off._ensure_ctx()
    pair = synthetic.make_correlated_pair(n0, n1, seed=n0 + B, batch=B)
    data = {k: torch.from_numpy(v).to(DEV) for k, v in pair.items() if k != 'image_shape'}
    data['image0'] = data['image1'] = torch.zeros(pair['image_shape'], device=DEV)
    with torch.no_grad():
        a = on.produce_matches(data, p=0.2, only_last=True)
        b = off.produce_matches(data, p=0.2, only_last=True)
    if min(n0, n1) > 192:
        assert torch.equal(a['indices0'][-1], b['indices0'][-1]) and torch.equal(a['mscores0'][-1], b['mscores0'][-1])
    else:
        print(compare_matches(_cpu(a['indices0'][-1]), _cpu(a['mscores0'][-1]), _cpu(b['indices0'][-1]).numpy(), _cpu(b['mscores0'][-1]).numpy(),
                              0.2, 1e-5, f'kv image vs fp32 format {n0}x{n1}', strict=False))
