"""GPU: opt-in soak (IMP_SOAK=1) - many random model / shape / seed combinations against the oracle.  Not part of the
default run (takes minutes of CPU oracle time); every case must meet the same bar: indices identical, scores within 1e-4."""
import os

import numpy as np
import pytest
import torch

from helpers import compare_matches, eval_config, make_hip_model
from imp_release_amd import synthetic
from oracle import imp_oracle as orc

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.environ.get('IMP_SOAK'), reason='set IMP_SOAK=1 to run the soak')]
DEV = 'cuda'


def _cases():
    rng = np.random.default_rng(2026)
    out = []
    for i in range(int(os.environ.get('IMP_SOAK', '0') or 0) * 12):
        model = ['GM', 'DGNNS', 'AdaGMN'][i % 3]
        out.append(dict(model=model, n0=int(rng.integers(1, 1400)), n1=int(rng.integers(1, 1400)), B=int(rng.integers(1, 4)),
                        L=int(rng.integers(1, 6)) if model == 'GM' else 15, T=int(rng.choice([0, 5, 20, 100])),
                        D=int(rng.choice([256, 256, 128])), seed=1000 + i, sink=bool(rng.integers(0, 5) > 0)))
    return out


@pytest.mark.parametrize('case', _cases(), ids=lambda c: f"{c['model']}-{c['n0']}x{c['n1']}-B{c['B']}-L{c['L']}-T{c['T']}-D{c['D']}")
def test_random_case_vs_oracle(case):
    cfg = eval_config(n_layers=case['L'], sinkhorn_iterations=max(case['T'], 1), descriptor_dim=case['D'],
                      with_sinkhorn=case['sink'] and case['T'] > 0)
    sd = synthetic.make_state_dict(cfg, case['model'], seed=case['seed'])
    m = make_hip_model(case['model'], cfg, sd)
    pair = synthetic.make_correlated_pair(case['n0'], case['n1'], desc_dim=case['D'], seed=case['seed'], batch=case['B'])
    data = {k: torch.from_numpy(v).to(DEV) for k, v in pair.items() if k != 'image_shape'}
    data['image0'] = data['image1'] = torch.zeros(pair['image_shape'], device=DEV)
    kw = {} if case['model'] == 'AdaGMN' else {'only_last': True}
    with torch.no_grad():
        out = m.produce_matches(data, p=0.2, **kw)
        ref = orc.MatcherOracle(cfg, sd, case['model']).produce_matches({k: v.cpu() for k, v in data.items()}, p=0.2, **kw)
    assert len(out['indices0']) == len(ref['indices0'])
    for i in range(len(ref['indices0'])):
        msg = compare_matches(out['indices0'][i].cpu().numpy(), out['mscores0'][i].cpu().numpy(), ref['indices0'][i].numpy(),
                        ref['mscores0'][i].numpy(), 0.2, 1e-4, f'{case} [{i}]', low_score_flips=4)
        if 'flips 0)' not in msg:
            print('SOAK-NOTE', msg)
