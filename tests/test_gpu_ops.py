"""GPU: every kernel behind the C-ABI against the oracle / an fp64 torch reference on seeded inputs.
Tolerances are written per test; indices are compared exactly."""
import numpy as np
import pytest
import torch

from helpers import eval_config, make_hip_model
from imp_release_amd import synthetic
from oracle import imp_oracle as orc

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _model(model='GM', precision=None, **over):
    cfg = eval_config(**{'n_layers': 2, **over})
    sd = synthetic.make_state_dict(cfg, model=model, seed=5)
    return cfg, sd, make_hip_model(model, cfg, sd, precision=precision), orc.MatcherOracle(cfg, sd, model=model)


@pytest.fixture(scope='module', params=['f16x3', 'f32'])
def gm(request):
    return _model(precision=request.param)


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize('M,N,K', [(300, 256, 256), (2048, 768, 256), (64, 32, 32), (1, 512, 512), (257, 129, 64),
                                   (130, 256, 512), (1000, 1000, 256)])
def test_linear_fp32_mfma(gm, M, N, K):
    ctx = gm[2]._ensure_ctx()
    x, W, b = _rand(M, K, seed=1), _rand(N, K, seed=2) / K ** .5, _rand(N, seed=3)
    y = ctx.op_linear(x.to(DEV), W.to(DEV), b.to(DEV)).cpu()
    ref = (x.double() @ W.double().t() + b.double())
    err = (y.double() - ref).abs().max().item()
    # fp32 MFMA: fp32 summation-order noise; f16x3: 2^-22 relative per product on top (still fp32-class)
    assert err < 3e-6 * max(1.0, ref.abs().max().item()) * (K / 32) ** .5, f'linear {M}x{N}x{K}: max err {err:.3e}'
    # transposition check (asymmetric operands): compare against the transposed product explicitly
    if M == N:
        assert (y.double() - ref.t()).abs().max().item() > 1e-2


# (B, M, K, ksplit, N, variant, pass_split): every instantiation of csrc/gemm_wf.hip, ragged row counts, single rows, pass splits
WF_CASES = [(2, 300, 256, 256, 768, 'plain', 1), (1, 64, 256, 256, 256, 'plain', 2), (1, 1, 256, 256, 768, 'plain', 6), (3, 1000, 256, 256, 768, 'plain', 3),
            (2, 300, 512, 256, 512, 'stats', 1), (1, 129, 512, 256, 512, 'stats', 4), (4, 2048, 512, 256, 512, 'stats', 1), (1, 5, 512, 256, 512, 'stats', 2),
            (1, 4100, 512, 256, 512, 'stats', 1), (2, 300, 512, 512, 256, 'norm', 1), (1, 70, 512, 512, 256, 'norm', 2), (2, 2048, 512, 512, 256, 'norm', 1),
            (2, 300, 512, 512, 256, 'chain768', 1), (1, 1, 512, 512, 256, 'chain256', 1), (4, 2048, 512, 512, 256, 'chain768', 1),
            (1, 130, 512, 512, 256, 'chain256', 1), (2, 200, 256, 256, 512, 'stats', 2)]


@pytest.mark.parametrize('B,M,K,ks,N,variant,psplit', WF_CASES)
def test_weight_fragment_layer_gemm(gm, B, M, K, ks, N, variant, psplit):
    """csrc/gemm_wf.hip alone against fp64: plain projection, MLP conv 0 (K-split concat + InstanceNorm statistics merged by the last
    workgroup to arrive), MLP conv 3 (norm + ReLU prologue, residual) and the chained conv 3 + next-layer projection"""
    ctx = gm[2]._ensure_ctx()
    x = _rand(B, M, ks, seed=1)
    x2 = _rand(B, M, K - ks, seed=2) if ks < K else None
    W, b = _rand(N, K, seed=3) / K ** .5, _rand(N, seed=4)
    kw = {}
    xin = x.double()
    if variant in ('norm', 'chain768', 'chain256'):
        # |mean| >> std on some channels, like the fixture gm_l3_bigmean
        x = x + torch.linspace(-20, 20, ks)[None, None, :]
        mean = x.double().mean(1)
        rstd = 1.0 / (x.double().var(1, unbiased=False) + 1e-3).sqrt()
        kw['stats_in'] = torch.stack([mean, rstd], -1).float().to(DEV)
        kw['residual'] = _rand(B, M, N, seed=5).to(DEV)
        xin = torch.relu((x.double() - kw['stats_in'][..., 0].cpu().double()[:, None]) * kw['stats_in'][..., 1].cpu().double()[:, None])
    if variant.startswith('chain'):
        N2 = int(variant[5:])
        W2, b2 = _rand(N2, 256, seed=6) / 16, _rand(N2, seed=7)
        kw['W2'], kw['bias2'] = W2.to(DEV), b2.to(DEV)
    y, so, y2 = ctx.op_layer_gemm(x.to(DEV), W.to(DEV), b.to(DEV), x2=None if x2 is None else x2.to(DEV), want_stats=variant == 'stats',
                                  pass_split=psplit, **kw)
    a = xin if x2 is None else torch.cat([xin, x2.double()], -1)
    ref = a @ W.double().t() + b.double()
    if 'residual' in kw:
        ref = ref + kw['residual'].cpu().double()
    tol = 3e-6 * max(1.0, ref.abs().max().item()) * (K / 32) ** .5
    err = (y.cpu().double() - ref).abs().max().item()
    assert err < tol, f'{variant} {B}x{M}x{N}x{K}: max err {err:.3e} (tol {tol:.1e})'
    if variant == 'stats':
        yd = y.cpu().double()
        mean, rstd = yd.mean(1), 1.0 / (yd.var(1, unbiased=False) + 1e-3).sqrt()
        assert (so[..., 0].cpu().double() - mean).abs().max().item() < 2e-6 * max(1.0, mean.abs().max().item())
        assert ((so[..., 1].cpu().double() - rstd) / rstd).abs().max().item() < 2e-6
    if y2 is not None:
        ref2 = y.cpu().double() @ W2.double().t() + b2.double()      # from the stored y: the chained GEMM reads exactly those values
        err2 = (y2.cpu().double() - ref2).abs().max().item()
        assert err2 < 3e-6 * max(1.0, ref2.abs().max().item()) * (256 / 32) ** .5, f'chained projection: max err {err2:.3e}'


# (B, M, N2): one 64-row tile per workgroup, B * ceil(M / 64) <= 256; ragged last tiles, tile counts that do not divide 512 (uneven
# channel slices), a single tile, the bench geometry per image (4 x 2048) and the full chip (8 x 2048 = 256 tiles)
FUSED_CASES = [(4, 2048, 768), (8, 2048, 768), (1, 1000, 768), (3, 77, 256), (1, 64, 768), (2, 1500, 768), (1, 5, 256), (2, 2048, 0), (1, 4100, 768),
               (5, 333, 256)]


@pytest.mark.parametrize('B,M,N2', FUSED_CASES)
def test_fused_layer_mlp_is_bit_identical_to_the_two_launches(gm, B, M, N2):
    """csrc/gemm_wf.hip gemm_wf_fused_kernel (round 4): mlp.0 -> InstanceNorm statistics exchanged between the workgroups of the launch
    -> ReLU -> mlp.3 + residual (-> chained projection) in ONE launch.  Same operands, same per-value instruction sequences, same merge
    order as MLP0 (statistics by the last-arriving workgroup) + MLP3 (norm prologue, chain): every output bit must agree.  Also against
    fp64.  Inputs with |mean| >> std on some hidden channels (bias ramp), like the fixture gm_l3_bigmean."""
    ctx = gm[2]._ensure_ctx()
    x, a = _rand(B, M, 256, seed=11), _rand(B, M, 256, seed=12)
    W0, b0 = _rand(512, 512, seed=13) / 512 ** .5, _rand(512, seed=14) + torch.linspace(-20, 20, 512)
    W3, b3 = _rand(256, 512, seed=15) / 512 ** .5, _rand(256, seed=16)
    W2 = b2 = None
    if N2:
        W2, b2 = (_rand(N2, 256, seed=17) / 16).to(DEV), _rand(N2, seed=18).to(DEV)
    xd, ad, W0d, b0d, W3d, b3d = (t.to(DEV) for t in (x, a, W0, b0, W3, b3))
    h, st, _ = ctx.op_layer_gemm(xd, W0d, b0d, x2=ad, want_stats=True)
    y_ref, _, y2_ref = ctx.op_layer_gemm(h, W3d, b3d, residual=xd, stats_in=st, W2=W2, bias2=b2)
    y, y2 = ctx.op_fused_mlp(xd, ad, W0d, b0d, W3d, b3d, W2, b2)
    assert torch.isfinite(y).all()
    nbad = int((y != y_ref).sum())
    assert nbad == 0, f'fused vs two launches: {nbad} of {y.numel()} descriptor values differ, max {float((y - y_ref).abs().max()):.3e}'
    if N2:
        assert torch.equal(y2, y2_ref), f'chained projection: {int((y2 != y2_ref).sum())} values differ'
    # fp64
    hd = torch.cat([x, a], -1).double() @ W0.double().t() + b0.double()
    hn = torch.relu((hd - hd.mean(1, keepdim=True)) / (hd.var(1, unbiased=False, keepdim=True) + 1e-3).sqrt())
    ref = x.double() + hn @ W3.double().t() + b3.double()
    err = (y.cpu().double() - ref).abs().max().item()
    # (a hidden value is (h - mean) * rstd with |h| up to ~25 and fp32-level h: 25 * 2^-22 * rstd(~1) per value, 512 terms)
    assert err < 2e-4, f'fused MLP vs fp64: max err {err:.3e}'


def test_fused_layer_mlp_time_out_is_reported_and_poisons_the_outputs(gm):
    """one workgroup withholds its statistics (test hook): every wait on them is bounded, the launch ends (seconds), the outputs are
    NaN - never plausible numbers from an unfinished exchange - and the call reports IMP_E_RESIDENT"""
    from imp_release_amd import _lib
    ctx = gm[2]._ensure_ctx()
    B, M = 2, 700
    x, a = _rand(B, M, 256, seed=1).to(DEV), _rand(B, M, 256, seed=2).to(DEV)
    W0, b0 = (_rand(512, 512, seed=3) / 512 ** .5).to(DEV), _rand(512, seed=4).to(DEV)
    W3, b3 = (_rand(256, 512, seed=5) / 512 ** .5).to(DEV), _rand(256, seed=6).to(DEV)
    W2, b2 = (_rand(768, 256, seed=7) / 16).to(DEV), _rand(768, seed=8).to(DEV)
    y = y2 = None
    with pytest.raises(_lib.ResidentSinkhornTimeout):
        y, y2 = ctx.op_fused_mlp(x, a, W0, b0, W3, b3, W2, b2, fake=True)
    # the same call without the hook works afterwards (fresh exchange buffers per call)
    y, y2 = ctx.op_fused_mlp(x, a, W0, b0, W3, b3, W2, b2)
    assert torch.isfinite(y).all() and torch.isfinite(y2).all()


def _ref_attention(qkv_q, qkv_kv, D, mask=None):
    """fp64 reference of nets/layers.py:121-131 on packed head-major projections"""
    B, nq, _ = qkv_q.shape
    nk = qkv_kv.shape[1]
    dh = D // 4
    q = qkv_q[..., :D].double().view(B, nq, 4, dh).transpose(1, 2)
    k = qkv_kv[..., D:2 * D].double().view(B, nk, 4, dh).transpose(1, 2)
    v = qkv_kv[..., 2 * D:].double().view(B, nk, 4, dh).transpose(1, 2)
    s = q @ k.transpose(-1, -2) / dh ** .5
    if mask is not None:
        s = s.masked_fill(~mask[:, None, None, :].bool(), -torch.finfo(torch.float32).max)
    prob = torch.softmax(s, -1)
    out = (prob @ v).transpose(1, 2).reshape(B, nq, D)
    return out, torch.logsumexp(s, -1)


@pytest.mark.parametrize('B,nq,nk,D,masked', [(1, 128, 128, 256, False), (2, 300, 307, 256, False),
                                              (1, 64, 2048, 256, False), (4, 1024, 1024, 256, False),
                                              (1, 200, 190, 128, False), (2, 257, 131, 256, True),
                                              (1, 33, 65, 128, True), (4, 1500, 1300, 128, True),
                                              (2, 2048, 2048, 128, False),       # 32-channel heads, ping-pong kernel
                                              # launches too small to fill the chip: the keys are split over 2-4 workgroups
                                              # per query tile and the last one merges the partial (O, m, l)
                                              (1, 1024, 1024, 256, False), (1, 600, 2100, 256, True), (2, 1000, 3000, 256, False),
                                              (1, 257, 1030, 128, True), (1, 4096, 4000, 256, False)])
def test_attention_fp32_mfma(gm, B, nq, nk, D, masked):
    ctx = gm[2]._ensure_ctx()
    qq, kv = _rand(B, nq, 3 * D, seed=4), _rand(B, nk, 3 * D, seed=5)
    mask = None
    if masked:
        mask = (torch.rand(B, nk, generator=torch.Generator().manual_seed(6)) > 0.4).to(torch.uint8)
        mask[:, :70] = 0                      # a whole first key tile masked out
        mask[:, -1] = 1
    out, lse = ctx.op_attention(qq.to(DEV), kv.to(DEV), None if mask is None else mask.to(DEV))
    ref, ref_lse = _ref_attention(qq, kv, D, mask)
    e_out = (out.cpu().double() - ref).abs().max().item()
    e_lse = (lse.cpu().double() - ref_lse).abs().max().item()
    assert e_out < 2e-5, f'attention out err {e_out:.3e} (lse err {e_lse:.3e})'
    assert e_lse < 2e-5, f'attention lse err {e_lse:.3e}'


@pytest.mark.parametrize('B,nq,nk,masked', [(8, 2048, 2048, False), (8, 2048, 64, False), (8, 2048, 100, True),
                                           (8, 2048, 130, False), (8, 2001, 1999, True), (16, 1000, 333, False),
                                           (8, 1793, 257, False)])
def test_attention_large_grid_pingpong_path(gm, B, nq, nk, masked):
    """>= 256 workgroups of 256 queries: the f16x3 build takes the phase-staggered (ping-pong) kernel with its
    4-slot LDS ring; key counts cover 1, 2, 3, 5 and 32 tiles, ragged tails, masks and a spike that forces the
    running-max rescale late in the stream"""
    ctx = gm[2]._ensure_ctx()
    D = 256
    qq, kv = _rand(B, nq, 3 * D, seed=14), _rand(B, nk, 3 * D, seed=15)
    if nk > 200:
        kv[0, nk - 60, D:2 * D] = qq[0, 5, :D] * 4.0
    mask = None
    if masked:
        mask = (torch.rand(B, nk, generator=torch.Generator().manual_seed(16)) > 0.4).to(torch.uint8)
        mask[:, :70] = 0
        mask[:, -1] = 1
    out, lse = ctx.op_attention(qq.to(DEV), kv.to(DEV), None if mask is None else mask.to(DEV))
    ref, ref_lse = _ref_attention(qq, kv, D, mask)
    e_out = (out.cpu().double() - ref).abs().max().item()
    e_lse = (lse.cpu().double() - ref_lse).abs().max().item()
    assert e_out < 5e-5, f'attention out err {e_out:.3e} (lse err {e_lse:.3e})'
    assert e_lse < 5e-4, f'attention lse err {e_lse:.3e}'


def test_attention_online_softmax_rescale_branch(gm):
    """a key that dominates late in the stream forces the running-max rescale (rare data-dependent branch)"""
    ctx = gm[2]._ensure_ctx()
    B, n, D = 1, 256, 256
    qq, kv = _rand(B, n, 3 * D, seed=7), _rand(B, n, 3 * D, seed=8)
    kv[0, 200, D:2 * D] = qq[0, 5, :D] * 4.0          # spike: key 200 (4th tile) against query 5, all heads
    out, lse = ctx.op_attention(qq.to(DEV), kv.to(DEV))
    ref, ref_lse = _ref_attention(qq, kv, D)
    assert (out.cpu().double() - ref).abs().max().item() < 5e-5
    assert (lse.cpu().double() - ref_lse).abs().max().item() < 5e-4


@pytest.mark.parametrize('n0,n1,T,sink', [(64, 64, 20, True), (300, 307, 100, True), (1, 5, 3, True),
                                          (255, 256, 20, True), (130, 97, 0, False), (1024, 1000, 100, True),
                                          (2400, 2600, 20, True), (3400, 3500, 6, True)])   # 13-chunk fused rows / two-pass fallback
def test_compute_score_and_matches(gm, n0, n1, T, sink):
    cfg, sd, m, o = gm
    ctx = m._ensure_ctx()
    B = 2
    dist = _rand(B, n0, n1, seed=9, scale=2.0)
    k = min(n0, n1) // 2
    for b in range(B):
        ids = torch.randperm(min(n0, n1), generator=torch.Generator().manual_seed(b))[:k]
        dist[b, ids, ids] += 6.0
    bin_score = 1.3
    got = ctx.compute_score(dist.to(DEV), bin_score, T, sink)
    aug = orc.dustbin_augment(dist, torch.tensor(bin_score))
    ref = orc.sinkhorn(aug, T) if sink else orc.dual_softmax(aug)
    # the dustbin corner holds O(N) mass: tolerance = 2e-5 absolute + 4 ulp relative
    err = ((got.cpu() - ref).abs() - 5e-7 * ref.abs()).max().item()
    assert err < 2e-5, f'compute_score err {err:.3e}'
    if sink and T > 0:       # property: after the last step every column sums to its marginal
        cs = got.cpu().double().sum(1)
        assert (cs[:, :-1] - 1).abs().max().item() < 1e-4 and (cs[:, -1] - (n1 + 1)).abs().max().item() < 1e-3
    # matches on the SAME score tensor must be identical (integer work)
    i0, i1, m0, m1 = ctx.compute_matches(got, 0.2)
    r0, r1, rm0, rm1 = orc.compute_matches(got.cpu(), 0.2)
    assert torch.equal(i0.cpu(), r0) and torch.equal(i1.cpu(), r1)
    assert torch.equal(m0.cpu(), rm0) and torch.equal(m1.cpu(), rm1)


def test_compute_matches_tie_break_first_index(gm):
    ctx = gm[2]._ensure_ctx()
    s = torch.rand(1, 41, 38, generator=torch.Generator().manual_seed(3)) * 0.1
    s[0, 3, 7] = s[0, 3, 20] = 0.9          # row tie -> first column
    s[0, 10, 5] = s[0, 30, 5] = 0.8         # column tie -> first row
    s[0, :, -1] = 5.0                        # dustbin column / row must be ignored
    s[0, -1, :] = 5.0
    got = ctx.compute_matches(s.to(DEV), 0.2)
    ref = orc.compute_matches(s, 0.2)
    for g, r in zip(got, ref):
        assert torch.equal(g.cpu(), r)


@pytest.mark.parametrize('norm,act,D', [('in', 'relu', 256), ('bn', 'lrelu', 256), ('in', 'gelu', 128)])
def test_encode_keypoints(norm, act, D):
    cfg, sd, m, o = _model(norm_fn=norm, ac_fn=act, descriptor_dim=D)
    ctx = m._ensure_ctx()
    pair = synthetic.make_pair(300, 129, desc_dim=D, seed=3, batch=2)
    t = {k: torch.from_numpy(v) for k, v in pair.items() if k != 'image_shape'}
    nk0 = orc.normalize_keypoints(t['keypoints0'], pair['image_shape'])
    nk1 = orc.normalize_keypoints(t['keypoints1'], pair['image_shape'])
    g0 = ctx.normalize_keypoints(t['keypoints0'].to(DEV), 640, 480)
    assert (g0.cpu() - nk0).abs().max().item() < 1e-6
    e0, e1 = ctx.encode_keypoints(nk0.to(DEV), t['scores0'].to(DEV), nk1.to(DEV), t['scores1'].to(DEV))
    r0, r1 = o.encode_keypoint(nk0, nk1, t['scores0'], t['scores1'])
    err = max((e0.cpu() - r0).abs().max().item(), (e1.cpu() - r1).abs().max().item())
    assert err < 5e-5, f'kenc err {err:.3e}'
    # fused residual
    f0, f1 = ctx.encode_keypoints(nk0.to(DEV), t['scores0'].to(DEV), nk1.to(DEV), t['scores1'].to(DEV),
                                  t['descriptors0'].to(DEV), t['descriptors1'].to(DEV))
    assert (f0.cpu() - (t['descriptors0'] + r0)).abs().max().item() < 5e-5


@pytest.mark.parametrize('precision', ['f16x3', 'f32'])
@pytest.mark.parametrize('model,norm,D,n0,n1', [('GM', 'in', 256, 300, 307), ('GM', 'bn', 256, 128, 64),
                                                ('DGNNS', 'in', 256, 200, 260), ('GM', 'in', 128, 150, 140)])
def test_forward_layers_and_cached_attention(model, norm, D, n0, n1, precision):
    nl = 4 if model == 'DGNNS' else 2
    cfg, sd, m, o = _model(model, precision=precision, norm_fn=norm, descriptor_dim=D, n_layers=nl)
    ctx = m._ensure_ctx()
    B = 2
    x0, x1 = _rand(B, n0, D, seed=11, scale=0.5), _rand(B, n1, D, seed=12, scale=0.5)
    g0, g1 = x0.to(DEV), x1.to(DEV)
    r0, r1 = x0, x1
    for li in range(2 * nl):
        g0, g1 = ctx.forward_layer(li, g0, g1)
        r0, r1 = o.forward_one_layer(r0, r1, li)
        err = max((g0.cpu() - r0).abs().max().item(), (g1.cpu() - r1).abs().max().item())
        assert err < 1e-4 * (li + 1), f'{model} layer {li} (shared={o.shared[li]}): err {err:.3e}'
    # re-materialised probabilities == what the reference caches (nets/layers.py:132)
    for which, ref in ((0, o.self_prob0), (1, o.self_prob1), (2, o.cross_prob1), (3, o.cross_prob0)):
        got = ctx.attention_prob(which, B, ref.shape[2], ref.shape[3], DEV)
        assert (got.cpu() - ref).abs().max().item() < 2e-5, f'prob {which}'
        recv = ctx.attention_received(which, B, ref.shape[3], DEV)
        assert (recv.cpu() - orc.attention_received(ref)).abs().max().item() < 1e-6, f'received {which}'
    # compute_distance
    d = ctx.compute_distance(nl - 1, g0, g1)
    rd = o.compute_distance(r0, r1, nl - 1)
    assert (d.cpu() - rd).abs().max().item() < 2e-4


def test_pool_and_gather():
    cfg, sd, m, o = _model('AdaGMN', n_layers=2)
    ctx = m._ensure_ctx()
    n0, n1, D = 333, 300, 256
    x0, x1 = _rand(1, n0, D, seed=21, scale=0.5), _rand(1, n1, D, seed=22, scale=0.5)
    g0, g1 = x0.to(DEV), x1.to(DEV)
    r0, r1 = x0, x1
    for li in range(2):
        g0, g1 = ctx.forward_layer(li, g0, g1)
        r0, r1 = o.forward_one_layer(r0, r1, li)
    gen = torch.Generator().manual_seed(5)
    score = torch.rand(1, n0 + 1, n1 + 1, generator=gen) * (0.5 / n1)
    ids = torch.randperm(min(n0, n1), generator=gen)[:120]
    score[0, ids, ids] += 0.4
    for th, nmin in ((0.2, 256), (0.05, 0), (50.0, 256), (0.2, 400)):
        got = ctx.pool(score.to(DEV), th, 1.0, nmin)
        ref = orc.pool(score, o.self_prob0, o.cross_prob0, o.self_prob1, o.cross_prob1, th, 1.0, nmin)
        for side in range(2):
            if ref[side] is None:
                assert got[side] is None, f'th={th} nmin={nmin} side {side}'
            else:
                assert got[side] is not None and torch.equal(got[side].cpu(), ref[side]), f'th={th} side {side}'
    keep = torch.tensor([5, 0, 7, 300, 332])
    assert torch.equal(ctx.gather_rows(g0, keep.to(DEV)).cpu(), g0.cpu()[:, keep])
    # selection on caller vectors incl. an even-count lower median
    mass = torch.tensor([0.5, 0.0, 0.9, 0.3, 0.0, 0.7])
    a_s = torch.tensor([0.10, 0.30, 0.20, 0.40, 0.25, 0.05])
    a_c = torch.tensor([0.60, 0.10, 0.20, 0.30, 0.90, 0.40])
    got = ctx.pool_select(mass.to(DEV), a_s.to(DEV), a_c.to(DEV), 0.25)
    assert torch.equal(got.cpu(), orc._pool_side(mass, a_s, a_c, 0.25))
    m0, m1 = ctx.score_mass(score[0].to(DEV))
    assert (m0.cpu() - score[0, :-1, :-1].sum(-1)).abs().max().item() < 1e-5
    assert (m1.cpu() - score[0, :-1, :-1].sum(0)).abs().max().item() < 1e-5


def test_pool_select_fuzz_with_ties_and_duplicates():
    """integer work must be exact: 200 random selection problems with heavy ties (quantised values), exact zeros,
    subnormals, sizes 1..5000 and thresholds that keep none / few / all candidates - kept id lists must equal the oracle's
    (torch.median = LOWER median; union; ascending)"""
    ctx = _model('AdaGMN', n_layers=2)[2]._ensure_ctx()
    g = torch.Generator().manual_seed(123)
    for trial in range(200):
        n = int(torch.randint(1, 5001, (1,), generator=g)) if trial % 4 else int(torch.randint(1, 70, (1,), generator=g))
        levels = int(torch.randint(1, 40, (1,), generator=g))
        def vec():
            v = torch.randint(0, levels + 1, (n,), generator=g).float() / levels
            kind = int(torch.randint(0, 4, (1,), generator=g))
            if kind == 1:
                v = v * 1e-41                      # subnormal range
            elif kind == 2:
                v = v + torch.rand(n, generator=g) * 1e-7
            return v
        mass, a_s, a_c = vec(), vec(), vec()
        thr = [0.0, 0.5, 2.0, float(mass.median())][trial % 4]
        ref = orc._pool_side(mass, a_s, a_c, thr)
        got = ctx.pool_select(mass.to(DEV), a_s.to(DEV), a_c.to(DEV), thr)
        if ref is None:
            assert got is None, (trial, n, thr)
        else:
            assert got is not None and torch.equal(got.cpu(), ref), (trial, n, thr, levels)


def test_masked_commit_equals_the_reference_index_arithmetic():
    """imp_masked_commit (masked AdaGMN bookkeeping of one pair, one launch) against nets/adgm.py:447-453,498-504 written with torch
    indexing: matches scattered to full-size rows, id lists composed with the pool's selection, key masks - exact"""
    ctx = _model('AdaGMN', n_layers=2)[2]._ensure_ctx()
    g = torch.Generator().manual_seed(77)
    for trial in range(40):
        nK0, nK1 = int(torch.randint(1, 3000, (1,), generator=g)), int(torch.randint(1, 3000, (1,), generator=g))
        n0 = int(torch.randint(1, nK0 + 1, (1,), generator=g)); n1 = int(torch.randint(1, nK1 + 1, (1,), generator=g))
        g0 = torch.randperm(nK0, generator=g)[:n0].sort().values
        g1 = torch.randperm(nK1, generator=g)[:n1].sort().values
        i0 = torch.randint(-1, n1, (n0,), generator=g)
        if trial % 5 == 0:
            i0[:] = -1
        m0 = torch.rand(n0, generator=g)
        update = trial % 3 != 0
        keep0 = None if trial % 4 == 1 else torch.randperm(n0, generator=g)[:int(torch.randint(1, n0 + 1, (1,), generator=g))].sort().values
        keep1 = None if trial % 4 == 2 else torch.randperm(n1, generator=g)[:int(torch.randint(1, n1 + 1, (1,), generator=g))].sort().values
        ref_i = torch.full((nK0,), -1, dtype=torch.long); ref_m = torch.zeros(nK0)
        v = i0 >= 0
        ref_i[g0[v]] = g1[i0[v]]
        ref_m[g0] = m0
        out_i = torch.full((2, nK0), -1, dtype=torch.long, device=DEV); out_m = torch.zeros(2, nK0, device=DEV)
        mk0 = torch.zeros(2, nK0, dtype=torch.uint8, device=DEV); mk1 = torch.zeros(2, nK1, dtype=torch.uint8, device=DEV)
        D = lambda t: None if t is None else t.to(DEV)      # noqa: E731
        ng0, ng1 = ctx.masked_commit(D(g0), D(g1), D(i0), D(m0), out_i[1], out_m[1], D(keep0), D(keep1), mk0[1], mk1[1], update=update)
        torch.cuda.synchronize()
        assert torch.equal(out_i[1].cpu(), ref_i) and torch.equal(out_m[1].cpu(), ref_m), trial
        assert (out_i[0] == -1).all() and (out_m[0] == 0).all() and (mk0[0] == 0).all() and (mk1[0] == 0).all()     # only this pair's rows
        if update:
            r0 = g0 if keep0 is None else g0[keep0]
            r1 = g1 if keep1 is None else g1[keep1]
            assert torch.equal(ng0.cpu(), r0) and torch.equal(ng1.cpu(), r1), trial
            e0 = torch.zeros(nK0, dtype=torch.uint8); e0[r0] = 1
            e1 = torch.zeros(nK1, dtype=torch.uint8); e1[r1] = 1
            assert torch.equal(mk0[1].cpu(), e0) and torch.equal(mk1[1].cpu(), e1), trial
        else:
            assert ng0 is None and ng1 is None and (mk0 == 0).all() and (mk1 == 0).all()


@pytest.mark.parametrize('n0,n1,T', [(64, 64, 20), (300, 307, 100), (1, 5, 3), (1024, 1000, 100), (2048, 2048, 100)])
def test_sinkhorn_on_the_three_byte_copy(n0, n1, T):
    """opt-in storage mode (imp_set_sinkhorn_storage(3)): the iterations stream a 3-byte copy of P, the scores are still
    formed from the fp32 matrix.  u, v move by ~1e-5 relative: inner scores (<= 1) stay within 2e-5 of the oracle, the
    O(N) dustbin entries within 3e-5 relative; column marginals hold to the same relative accuracy; matches taken from the
    score tensor are exact integer work as always"""
    cfg = eval_config(n_layers=2, sinkhorn_storage=3)
    sd = synthetic.make_state_dict(cfg, model='GM', seed=5)
    ctx = make_hip_model('GM', cfg, sd)._ensure_ctx()
    B = 2
    dist = _rand(B, n0, n1, seed=9, scale=2.0)
    k = min(n0, n1) // 2
    for b in range(B):
        ids = torch.randperm(min(n0, n1), generator=torch.Generator().manual_seed(b))[:k]
        dist[b, ids, ids] += 6.0
    got = ctx.compute_score(dist.to(DEV), 1.3, T, True)
    ref = orc.sinkhorn(orc.dustbin_augment(dist, torch.tensor(1.3)), T)
    d = (got.cpu() - ref).abs()
    assert (d - 3e-5 * ref.abs()).max().item() < 2e-5, f'max abs {d.max().item():.3e}'
    assert d[:, :-1, :-1].max().item() < 2e-5                      # everything a match score can be
    cs = got.cpu().double().sum(1)
    assert (cs[:, :-1] - 1).abs().max().item() < 1e-4 and ((cs[:, -1] - (n1 + 1)).abs() / (n1 + 1)).max().item() < 3e-5
    i0, i1, m0, m1 = ctx.compute_matches(got, 0.2)
    r0, r1, rm0, rm1 = orc.compute_matches(got.cpu(), 0.2)
    assert torch.equal(i0.cpu(), r0) and torch.equal(i1.cpu(), r1) and torch.equal(m0.cpu(), rm0)
    # and against the fp32-streaming mode: identical match indices
    ctx4 = make_hip_model('GM', eval_config(n_layers=2), sd)._ensure_ctx()
    got4 = ctx4.compute_score(dist.to(DEV), 1.3, T, True)
    j0 = ctx4.compute_matches(got4, 0.2)[0]
    assert torch.equal(j0, i0)


@pytest.mark.parametrize('scale,B,nq,nk', [(4.0, 8, 2048, 1024), (6.0, 2, 700, 900), (0.01, 2, 512, 512)])
def test_attention_extreme_logits(gm, scale, B, nq, nk):
    """logits with a standard deviation of ~16 / ~36 (every tile moves the softmax reference by far more than the 2^14 lazy
    window, so the exact-maximum path runs constantly) and nearly flat logits (the reference never moves after tile 0)"""
    ctx = gm[2]._ensure_ctx()
    D = 256
    qq, kv = _rand(B, nq, 3 * D, seed=24), _rand(B, nk, 3 * D, seed=25)
    qq[..., :D] *= scale
    kv[..., D:2 * D] *= scale
    out, lse = ctx.op_attention(qq.to(DEV), kv.to(DEV))
    ref, ref_lse = _ref_attention(qq, kv, D)
    e_out = (out.cpu().double() - ref).abs().max().item()
    e_lse = ((lse.cpu().double() - ref_lse).abs() / ref_lse.abs().clamp(min=1.0)).max().item()
    assert torch.isfinite(out).all() and torch.isfinite(lse).all()
    # the logits themselves carry an absolute fp32 error ~ |S| * 1e-6, which the exponential turns into a relative one
    assert e_out < 3e-5 * max(1.0, scale * scale), f'attention out err {e_out:.3e}'
    assert e_lse < 2e-5, f'attention lse rel err {e_lse:.3e}'


def _range_case():
    cfg = eval_config(n_layers=2, sinkhorn_iterations=10)
    sd = synthetic.make_state_dict(cfg, 'GM', seed=3)
    pair = synthetic.make_correlated_pair(300, 280, seed=5)
    data = {k: torch.from_numpy(v).to(DEV) for k, v in pair.items() if k != 'image_shape'}
    data['image0'] = data['image1'] = torch.zeros(pair['image_shape'], device=DEV)
    big = dict(data)
    big['descriptors0'] = data['descriptors0'] * 1.0
    big['descriptors0'][0, 7, 3] = 7.0e4                           # one operand beyond the fp16 range
    return cfg, sd, data, big


@pytest.mark.parametrize('resident', ['1', '0'])
def test_operand_beyond_the_fp16_range_is_reported_not_silent(resident):
    """f16x3 arithmetic WITHOUT in-call recovery (range_recovery=False: the library never waits for the GPU): |x| >= 65504 in a matrix
    operand makes the scores NaN.  The match kernel notices, the matches of that call are void (-1) and the next entry point raises
    IMP_E_RANGE; precision f32 handles the same data"""
    from imp_release_amd._lib import OperandRangeError
    import os
    os.environ['IMP_OT_RESIDENT'] = resident            # chip-resident and streaming Sinkhorn kernels: both end in the same match kernel
    cfg, sd, data, big = _range_case()
    try:
        m = make_hip_model('GM', dict(cfg, range_recovery=False), sd)
        m._ensure_ctx()
    finally:
        del os.environ['IMP_OT_RESIDENT']
    with torch.no_grad():
        out = m.produce_matches(big, p=0.2, only_last=True)
        torch.cuda.synchronize()
        assert (out['indices0'][-1] == -1).all()                   # void, never plausible garbage
        with pytest.raises(OperandRangeError):
            m.produce_matches(data, p=0.2, only_last=True)
        good = m.produce_matches(data, p=0.2, only_last=True)      # the context keeps working
        assert (good['indices0'][-1] >= 0).any()
        m32 = make_hip_model('GM', cfg, sd, precision='f32')
        o32 = m32.produce_matches(big, p=0.2, only_last=True)      # native fp32 MFMA: no such limit
        m32.produce_matches(data, p=0.2, only_last=True)           # ... and nothing to report
    torch.cuda.synchronize()
    assert torch.isfinite(o32['mscores0'][-1]).all() and m._ensure_ctx().range_counts() == (1, 0)


@pytest.mark.parametrize('resident', ['1', '0'])
def test_operand_beyond_the_fp16_range_is_recovered_inside_the_call(resident):
    """the default (config key range_recovery, include/imp_hip.h imp_set_range_recovery; VERDICT r4 #5b): the call that meets the operand
    waits for its own work, sees the range word and runs itself again on the native fp32 MFMA path - the caller gets exactly what a
    precision='f32' model returns for these inputs (the reference returns finite matches for such data, nets/gm.py:305-320), nothing is
    raised at the next call, and ordinary data afterwards runs in f16x3 again"""
    import os
    os.environ['IMP_OT_RESIDENT'] = resident
    cfg, sd, data, big = _range_case()
    try:
        m = make_hip_model('GM', dict(cfg, range_recovery=True), sd)         # (explicit: the default, whatever IMP_RANGE_RECOVERY says)
        m32 = make_hip_model('GM', cfg, sd, precision='f32')
        mref = make_hip_model('GM', dict(cfg, range_recovery=False), sd)
        m._ensure_ctx(); m32._ensure_ctx(); mref._ensure_ctx()      # (contexts read the environment when they are created)
    finally:
        del os.environ['IMP_OT_RESIDENT']
    with torch.no_grad():
        out = m.produce_matches(big, p=0.2, only_last=True)
        want = m32.produce_matches(big, p=0.2, only_last=True)
        torch.cuda.synchronize()
        assert (want['indices0'][-1] >= 0).any() and torch.isfinite(want['mscores0'][-1]).all()
        assert torch.equal(out['indices0'][-1], want['indices0'][-1]) and torch.equal(out['mscores0'][-1], want['mscores0'][-1])
        assert m._ensure_ctx().range_counts() == (1, 1) and m._ensure_ctx().precision == 'f16x3'
        good = m.produce_matches(data, p=0.2, only_last=True)      # no IMP_E_RANGE left behind; back on the split-half arithmetic
        ref = mref.produce_matches(data, p=0.2, only_last=True)
        torch.cuda.synchronize()
        assert torch.equal(good['indices0'][-1], ref['indices0'][-1]) and torch.equal(good['mscores0'][-1], ref['mscores0'][-1])
        # the all-iterations path (layer calls + tail per iteration): the NaN descriptors reach the tail from the caller's earlier calls, the tail
        # cannot repair them - the module re-runs the whole pass on the fp32 path instead
        alli = m.produce_matches(big, p=0.2, only_last=False)
        want_all = m32.produce_matches(big, p=0.2, only_last=False)
        torch.cuda.synchronize()
        for a_, w_ in zip(alli['indices0'], want_all['indices0']):
            assert torch.equal(a_, w_)
    assert m._ensure_ctx().precision == 'f16x3'
