"""SuperPoint front-end on the GPU against the golden fixtures captured from the imported reference
(tools/make_golden.py case_superpoint) and, stage by stage, against oracle/superpoint_oracle.py.

Bar: keypoints (integer pixel coordinates, order included) bit-exact; scores and descriptors within 1e-5 / 1e-4 absolute
(descriptors are unit vectors).  The matrix products run as split-half f16x3 MFMAs (fp32-level), so the keypoint SET only
moves on exact ties of the fp32 score map; the fixtures were chosen where fp32 and fp64 oracles agree (tools/make_golden.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import golden_names, load_golden, match_keypoint_lists
from imp_release_amd import synthetic

pytestmark = pytest.mark.gpu

SP_FIXTURES = golden_names(['superpoint_'])


def _module(spec, **over):
    from imp_release_amd.superpoint import SuperPoint
    sd = synthetic.make_superpoint_state_dict(seed=spec['wseed'], descriptor_dim=spec.get('descriptor_dim', 256))
    cfg = {**spec['config'], 'state_dict': sd, **over}
    return SuperPoint(cfg, device=torch.device('cuda:0')), sd


def _image(spec):
    return torch.from_numpy(synthetic.make_image(spec['height'], spec['width'], seed=spec['iseed'], batch=spec.get('batch', 1)))


def test_fixtures_exist():
    assert len(SP_FIXTURES) >= 7


@pytest.mark.parametrize('name', SP_FIXTURES)
def test_forward_vs_golden(name):
    spec, z = load_golden(name)
    sp, _ = _module(spec, align_corners=bool(int(z['align_corners'])))
    img = _image(spec).cuda()
    out = sp({'image': img})
    topk = spec['config'].get('max_keypoints', -1) >= 0
    for b in range(img.shape[0]):
        kp = out['keypoints'][b].cpu().numpy()
        sc = out['scores'][b].cpu().numpy()
        assert kp.dtype == np.float32
        perm, moved, boundary = match_keypoint_lists(kp, sc, z[f'keypoints_{b}'].astype(np.float32), z[f'scores_{b}'], topk)
        # positions can only change inside runs of reference scores closer than the score tolerance
        near = int((-np.diff(z[f'scores_{b}']) < 2e-5).sum()) if topk else 0
        assert moved <= 2 * near and boundary <= 1, (moved, boundary, near)
        de = out['descriptors'][b].cpu()
        n = kp.shape[0]
        assert tuple(de.shape) == (spec.get('descriptor_dim', 256), n) and de.is_contiguous()
        ok = perm >= 0
        ref_pos = np.maximum(perm, 0)
        head = ok & (ref_pos < 48)                                    # keypoints whose full descriptor the fixture holds
        assert np.abs(de[:, head].numpy() - z[f'desc_head_{b}'][:, ref_pos[head]]).max(initial=0.0) < 1e-4
        assert np.abs(de[::32][:, ok].numpy() - z[f'desc_rows_{b}'][:, ref_pos[ok]]).max(initial=0.0) < 1e-4
        assert np.abs(de.double().sum(0).numpy()[ok] - z[f'desc_sum_{b}'][ref_pos[ok]]).max(initial=0.0) < 2e-3
        assert np.abs(np.linalg.norm(de.numpy(), axis=0) - 1).max(initial=0.0) < 1e-5


@pytest.mark.parametrize('name', SP_FIXTURES)
def test_extract_vs_golden(name):
    """extract() (nets/superpoint.py:140-168): dense score map and dense descriptors"""
    spec, z = load_golden(name)
    sp, _ = _module(spec)
    img = _image(spec).cuda()
    scores, desc = sp.extract({'image': img})
    assert abs(float(scores.double().sum()) - float(z['dense_scores_sum'])) < 1e-3 * max(1.0, abs(float(z['dense_scores_sum'])))
    for b in range(img.shape[0]):
        assert np.abs(scores[b, 3::8, 5::8].cpu().numpy() - z[f'dense_scores_probe_{b}']).max() < 1e-5
        assert np.abs(desc[b, ::16, ::3, ::3].cpu().numpy() - z[f'dense_desc_probe_{b}']).max() < 1e-4


CONVS = [(0, 'conv1a', 1, 37, 53, True, False), (1, 'conv1b', 64, 24, 40, True, True), (1, 'conv1b', 64, 17, 23, True, False),
         (1, 'conv1b', 64, 17, 23, True, True), (2, 'conv2a', 64, 16, 16, False, False), (4, 'conv3a', 64, 20, 28, True, False),
         (5, 'conv3b', 128, 22, 30, True, True), (7, 'conv4b', 128, 9, 11, True, False), (8, 'heads', 128, 15, 20, True, False),
         (9, 'convDb', 256, 15, 20, False, False), (9, 'convDb', 256, 1, 1, False, False)]


@pytest.mark.parametrize('layer,name,cin,H,W,relu,pool', CONVS)
def test_single_convolution_vs_torch(layer, name, cin, H, W, relu, pool):
    """every kernel configuration of the stack on its own (ragged tiles, odd sizes under pooling, 1x1) against conv2d in fp64"""
    spec = dict(wseed=5, config={})
    sp, sd = _module(spec)
    t = {k: torch.from_numpy(v) for k, v in sd.items()}
    g = torch.Generator().manual_seed(layer * 100 + H)
    x = torch.randn(2, cin, H, W, generator=g)
    if layer == 8:
        wt, bs = torch.cat([t['convPa.weight'], t['convDa.weight']]), torch.cat([t['convPa.bias'], t['convDa.bias']])
    else:
        wt, bs = t[name + '.weight'], t[name + '.bias']
    ref = F.conv2d(x.double(), wt.double(), bs.double(), padding=wt.shape[-1] // 2)
    ref = torch.relu(ref) if relu else ref
    ref = F.max_pool2d(ref, 2, 2) if pool else ref
    out = sp.op_conv(layer, x.cuda(), relu=relu, pool=pool).cpu().double()
    assert out.shape == ref.shape
    assert float((out - ref).abs().max()) < 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize('H,W,radius', [(96, 128, 4), (104, 72, 0), (64, 200, 1), (136, 136, 7)])
def test_nms_kernel_is_exact_on_its_own_input(H, W, radius):
    """simple_nms (nets/superpoint.py:49-63) involves only comparisons: the kernel must reproduce the oracle bit for bit when
    both start from the SAME dense score map (the GPU's)"""
    from oracle import superpoint_oracle as spo
    spec = dict(wseed=6, config=dict(nms_radius=radius))
    sp, _ = _module(spec)
    img = torch.from_numpy(synthetic.make_image(H, W, seed=H + radius)).cuda()
    dense, _ = sp.extract({'image': img})
    nms, _ = sp.extract({'image': img}, nms=True)
    want = spo.simple_nms(dense.cpu(), radius)
    assert torch.equal(nms.cpu(), want)


def test_nms_with_exact_ties_and_plateaus():
    """quantised score maps are full of equal neighbours (every member of a plateau is a maximum): same rule as the reference"""
    from oracle import superpoint_oracle as spo
    spec = dict(wseed=6, config={})
    sp, sd = _module(spec)
    # a detector head with zero weights and zero bias gives a constant map (all ties): softmax = 1 / 65 everywhere
    sd2 = {k: v.copy() for k, v in sd.items()}
    sd2['convPb.weight'][:] = 0
    sd2['convPb.bias'][:] = 0
    from imp_release_amd.superpoint import SuperPoint
    sp2 = SuperPoint({'state_dict': sd2, 'max_keypoints': -1, 'remove_borders': 0}, device=torch.device('cuda:0'))
    img = torch.from_numpy(synthetic.make_image(40, 48, seed=1)).cuda()
    out = sp2({'image': img})
    assert out['keypoints'][0].shape[0] == 40 * 48                       # every pixel is a maximum of the constant map
    want = spo.forward(sd2, img.cpu(), remove_borders=0, align_corners=False)
    assert torch.equal(out['keypoints'][0].cpu(), want['keypoints'][0])


@pytest.mark.parametrize('k', [1, 7, 250])
def test_top_k_is_sorted_and_matches_the_oracle(k):
    from oracle import superpoint_oracle as spo
    spec = dict(wseed=7, config=dict(max_keypoints=k))
    sp, sd = _module(spec, align_corners=True)
    img = torch.from_numpy(synthetic.make_image(160, 200, seed=11))
    out = sp({'image': img.cuda()})
    want = spo.forward(sd, img, max_keypoints=k, align_corners=True)
    sc = out['scores'][0].cpu()
    assert sc.shape[0] == k and bool((sc[:-1] >= sc[1:]).all())
    perm, moved, boundary = match_keypoint_lists(out['keypoints'][0].cpu().numpy(), sc.numpy(), want['keypoints'][0].numpy(),
                                                 want['scores'][0].numpy(), True)
    assert moved <= 4 and boundary == 0
    assert float((out['descriptors'][0].cpu() - want['descriptors'][0][:, perm]).abs().max()) < 1e-4


def test_top_k_radix_select_path_with_48000_candidates():
    """nms_radius 0 on a 200 x 240 map: every pixel above the threshold is a candidate"""
    from oracle import superpoint_oracle as spo
    spec = dict(wseed=8, config=dict(max_keypoints=500, nms_radius=0, keypoint_threshold=0.0, remove_borders=0))
    sp, sd = _module(spec, align_corners=False)
    img = torch.from_numpy(synthetic.make_image(200, 240, seed=12))
    out = sp({'image': img.cuda()})
    want = spo.forward(sd, img, nms_radius=0, keypoint_threshold=0.0, max_keypoints=500, remove_borders=0, align_corners=False)
    match_keypoint_lists(out['keypoints'][0].cpu().numpy(), out['scores'][0].cpu().numpy(), want['keypoints'][0].numpy(),
                         want['scores'][0].numpy(), True)


def test_top_k_of_5000_out_of_48000_candidates():
    from oracle import superpoint_oracle as spo
    spec = dict(wseed=8, config=dict(max_keypoints=5000, nms_radius=0, keypoint_threshold=0.0, remove_borders=0))
    sp, sd = _module(spec, align_corners=False)
    img = torch.from_numpy(synthetic.make_image(200, 240, seed=13))
    out = sp({'image': img.cuda()})
    want = spo.forward(sd, img, nms_radius=0, keypoint_threshold=0.0, max_keypoints=5000, remove_borders=0, align_corners=False)
    sc = out['scores'][0].cpu()
    assert sc.shape[0] == 5000 and bool((sc[:-1] >= sc[1:]).all())
    match_keypoint_lists(out['keypoints'][0].cpu().numpy(), sc.numpy(), want['keypoints'][0].numpy(), want['scores'][0].numpy(), True)


def test_top_k_with_exact_ties_at_the_cut_is_deterministic():
    """all scores equal (zero detector head): torch.topk's choice among equal values is unspecified; this implementation takes
    the lowest indices, i.e. the first k keypoints in nonzero (row-major) order, on every run"""
    from imp_release_amd.superpoint import SuperPoint
    sd = synthetic.make_superpoint_state_dict(seed=6)
    sd['convPb.weight'][:] = 0
    sd['convPb.bias'][:] = 0
    img = torch.from_numpy(synthetic.make_image(64, 72, seed=1)).cuda()
    full = SuperPoint({'state_dict': sd, 'max_keypoints': -1}, device=torch.device('cuda:0'))({'image': img})
    n = full['keypoints'][0].shape[0]
    assert n == 56 * 64                                                   # every pixel inside the border is a maximum
    for k in (100, 3000):
        sp = SuperPoint({'state_dict': sd, 'max_keypoints': k}, device=torch.device('cuda:0'))
        for _ in range(2):
            out = sp({'image': img})
            assert torch.equal(out['keypoints'][0], full['keypoints'][0][:k])
            assert torch.equal(out['descriptors'][0], full['descriptors'][0][:, :k])


def test_large_image_768x1024_top2048():
    """a size the evaluation scripts of the SuperGlue lineage use for outdoor images; 6000+ tiles in the first layer, more candidates
    than the direct-sort limit, k = 2048"""
    from oracle import superpoint_oracle as spo
    spec = dict(wseed=11, config=dict(max_keypoints=2048))
    sp, sd = _module(spec, align_corners=False)
    img = torch.from_numpy(synthetic.make_image(768, 1024, seed=31))
    out = sp({'image': img.cuda()})
    with torch.no_grad():
        want = spo.forward(sd, img, max_keypoints=2048, align_corners=False)
    perm, moved, boundary = match_keypoint_lists(out['keypoints'][0].cpu().numpy(), out['scores'][0].cpu().numpy(),
                                                 want['keypoints'][0].numpy(), want['scores'][0].numpy(), True)
    assert boundary <= 1
    ok = perm >= 0
    assert float((out['descriptors'][0].cpu()[:, ok] - want['descriptors'][0][:, perm[ok]]).abs().max()) < 1e-4


@pytest.mark.parametrize('H,W', [(8, 8), (9, 17), (31, 33), (16, 200), (200, 24)])
def test_tiny_and_odd_image_sizes(H, W):
    """the smallest legal image (one 8 x 8 cell), sizes that are not multiples of 8 or of the tile, extreme aspect ratios"""
    from oracle import superpoint_oracle as spo
    spec = dict(wseed=12, config=dict(max_keypoints=-1, remove_borders=1, nms_radius=2))
    sp, sd = _module(spec, align_corners=True)
    img = torch.from_numpy(synthetic.make_image(H, W, seed=H * 31 + W))
    out = sp({'image': img.cuda()})
    with torch.no_grad():
        want = spo.forward(sd, img, nms_radius=2, remove_borders=1, align_corners=True)
    assert torch.equal(out['keypoints'][0].cpu(), want['keypoints'][0])
    if want['keypoints'][0].shape[0]:
        assert float((out['scores'][0].cpu() - want['scores'][0]).abs().max()) < 1e-5
        assert float((out['descriptors'][0].cpu() - want['descriptors'][0]).abs().max()) < 1e-4
    dense, desc = sp.extract({'image': img.cuda()})
    assert tuple(dense.shape) == (1, H // 8 * 8, W // 8 * 8) and tuple(desc.shape) == (1, 256, H // 8, W // 8)


@pytest.mark.parametrize('case', range(12))
def test_randomised_configurations_vs_oracle(case):
    """every fast NMS radius (1..6) and the generic ones (0, 7), borders, thresholds, top-k on / off, odd sizes, D = 128 / 256"""
    from oracle import superpoint_oracle as spo
    g = np.random.default_rng(1000 + case)
    H, W = int(g.integers(40, 200)), int(g.integers(40, 260))
    radius = [1, 2, 3, 4, 5, 6, 0, 7, 4, 3, 2, 5][case]
    cfg = dict(nms_radius=radius, remove_borders=int(g.integers(0, 9)), keypoint_threshold=float(g.choice([0.0, 0.001, 0.0025, 0.01, 0.05])),
               max_keypoints=int(g.choice([-1, -1, 50, 400])), descriptor_dim=int(g.choice([128, 256])))
    spec = dict(wseed=100 + case, descriptor_dim=cfg['descriptor_dim'], config=cfg)
    ac = bool(case & 1)
    sp, sd = _module(spec, align_corners=ac)
    img = torch.from_numpy(synthetic.make_image(H, W, seed=500 + case))
    out = sp({'image': img.cuda()})
    with torch.no_grad():
        want = spo.forward(sd, img, nms_radius=radius, keypoint_threshold=cfg['keypoint_threshold'], max_keypoints=cfg['max_keypoints'],
                           remove_borders=cfg['remove_borders'], align_corners=ac)
    topk = cfg['max_keypoints'] >= 0 and want['keypoints'][0].shape[0] == cfg['max_keypoints']
    perm, moved, boundary = match_keypoint_lists(out['keypoints'][0].cpu().numpy(), out['scores'][0].cpu().numpy(),
                                                 want['keypoints'][0].numpy(), want['scores'][0].numpy(), topk)
    assert boundary <= 1, (cfg, H, W)
    ok = perm >= 0
    if ok.any():
        assert float((out['descriptors'][0].cpu()[:, ok] - want['descriptors'][0][:, perm[ok]]).abs().max()) < 1e-4


def test_no_keypoints_above_the_threshold():
    spec = dict(wseed=9, config=dict(keypoint_threshold=2.0))
    sp, _ = _module(spec)
    out = sp({'image': torch.from_numpy(synthetic.make_image(64, 64, seed=2)).cuda()})
    assert out['keypoints'][0].shape == (0, 2) and out['scores'][0].shape == (0,) and out['descriptors'][0].shape == (256, 0)


def test_argument_errors():
    from imp_release_amd import _lib
    from imp_release_amd.superpoint import SuperPoint
    sd = synthetic.make_superpoint_state_dict(seed=0)
    with pytest.raises(ValueError):
        SuperPoint({'state_dict': sd, 'max_keypoints': 0})                 # nets/superpoint.py:161-163
    sp = SuperPoint({'state_dict': sd}, device=torch.device('cuda:0'))
    with pytest.raises(_lib.ImpError):
        sp({'image': torch.zeros(1, 1, 4, 4).cuda()})
    with pytest.raises(ValueError):
        sp({'image': torch.zeros(1, 3, 32, 32).cuda()})
    bad = {k: v for k, v in sd.items() if k != 'conv3a.bias'}
    with pytest.raises(_lib.ImpError):
        SuperPoint({'state_dict': bad}, device=torch.device('cuda:0'))


def test_front_end_feeds_the_matcher():
    """image pair -> SuperPoint -> GM.produce_matches, everything on the GPU, against the oracle chain on the same inputs"""
    from helpers import eval_config, make_hip_model
    from oracle import imp_oracle as orc
    from oracle import superpoint_oracle as spo
    H, W, k = 240, 320, 256
    spec = dict(wseed=0, config=dict(max_keypoints=k))
    sp, ssd = _module(spec, align_corners=False)
    cfg = eval_config(n_layers=3, sinkhorn_iterations=20)
    msd = synthetic.make_state_dict(cfg, 'GM', seed=1)
    gm = make_hip_model('GM', cfg, msd)
    imgs = [torch.from_numpy(synthetic.make_image(H, W, seed=s)) for s in (21, 22)]
    data_g, data_o = {}, {}
    for i, im in enumerate(imgs):
        og = sp({'image': im.cuda()})
        oo = spo.forward(ssd, im, max_keypoints=k, align_corners=False)
        perm, _, boundary = match_keypoint_lists(og['keypoints'][0].cpu().numpy(), og['scores'][0].cpu().numpy(),
                                                 oo['keypoints'][0].numpy(), oo['scores'][0].numpy(), True)
        assert boundary == 0
        og = {key: [og[key][0][..., torch.from_numpy(np.argsort(perm)).cuda()] if key == 'descriptors'
                    else og[key][0][torch.from_numpy(np.argsort(perm)).cuda()]] for key in og}      # reference order
        for d, o in ((data_g, og), (data_o, oo)):
            d[f'keypoints{i}'] = o['keypoints'][0][None]
            d[f'scores{i}'] = o['scores'][0][None]
            d[f'descriptors{i}'] = o['descriptors'][0].t()[None].contiguous()
            d[f'image{i}'] = im.to(o['keypoints'][0].device)
    got = gm.produce_matches(data_g, p=0.2, only_last=True)
    want = orc.MatcherOracle(cfg, msd, 'GM').produce_matches(data_o, p=0.2, only_last=True)
    gi, wi = got['indices0'][-1].cpu().numpy(), want['indices0'][-1].numpy()
    assert (gi != wi).sum() <= 2
