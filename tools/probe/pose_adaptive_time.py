"""pose step: fixed budget vs adaptive termination - wall time per call and samples drawn (round 5)"""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from imp_release_amd import pose as gpose
from oracle import pose_oracle as po
for o in (0.3, 0.5, 0.65):
    k0, k1, K, R, t, tr = po.synthetic_scene(1000, outliers=o, noise=0.4, seed=3)
    for ad in (False, True):
        gpose.pose_stats(reset=True)
        for _ in range(5):
            gpose.estimate_pose(k0, k1, K, K, 1.0, adaptive=ad)
        gpose.pose_stats(reset=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50):
            gpose.estimate_pose(k0, k1, K, K, 1.0, adaptive=ad)
        dt = (time.perf_counter() - t0) / 50 * 1e3
        c, s = gpose.pose_stats()
        print("outliers %.2f adaptive %s: %.3f ms per call, %.0f samples per call" % (o, ad, dt, s / c), flush=True)
