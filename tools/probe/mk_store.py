import sys, numpy as np
sys.path.insert(0, '.')
from imp_release_amd import data, synthetic
recs = []
for i in range(12):
    p = synthetic.make_correlated_pair(2000, 1900, seed=300 + i)
    recs.append({'K1': np.eye(3), 'K2': np.eye(3), 'R': np.eye(3), 'T': np.array([1., 2., 2.]), 'e': np.zeros((3, 3)), 'f': np.zeros((3, 3)),
                 'kpt1': np.concatenate([p['keypoints0'][0], p['scores0'][0][:, None]], 1), 'kpt2': np.concatenate([p['keypoints1'][0], p['scores1'][0][:, None]], 1),
                 'desc1': p['descriptors0'][0], 'desc2': p['descriptors1'][0], 'size1': (480, 640), 'size2': (480, 640)})
print(data.write_npz_store(recs, '/tmp/store_npz'))
