import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from imp_release_amd import pose
from oracle import pose_oracle as po
torch.zeros(1).cuda()
for n in (200, 1000, 3000):
    k0, k1, K, R, t, truth = po.synthetic_scene(n, outliers=0.3, noise=0.3, seed=1)
    pose.estimate_pose(k0, k1, K, K, 1.0)
    t0 = time.perf_counter()
    for _ in range(20): r = pose.estimate_pose(k0, k1, K, K, 1.0)
    dt = (time.perf_counter() - t0) / 20
    t1 = time.perf_counter(); c = po.estimate_pose(k0, k1, K, K, 1.0, iterations=64); tc = (time.perf_counter() - t1) * 16
    print(f'n={n}: GPU pose call (default: 1024 five-point samples x <= 10 models, MAGSAC++ quality; host in/out) {dt * 1e3:.2f} ms ; numpy twin extrapolated to 1024 samples {tc * 1e3:.0f} ms ; inliers {r[3].sum()}')
