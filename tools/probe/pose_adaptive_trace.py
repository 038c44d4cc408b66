import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from imp_release_amd import pose as gpose
from oracle import pose_oracle as po
o = float(sys.argv[1]); ad = sys.argv[2] == '1'
k0,k1,K,R,t,tr = po.synthetic_scene(1000, outliers=o, noise=0.4, seed=3)
for _ in range(30): gpose.estimate_pose(k0,k1,K,K,1.0,adaptive=ad)
