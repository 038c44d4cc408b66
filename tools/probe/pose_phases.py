"""phase times of the five-point solver inside the hypotheses kernel (build with -DFP_PROFILE: imp-release_amd/csrc/variants/libimp_hip_fpprof.so)
    IMP_HIP_LIB=$PWD/imp-release_amd/csrc/variants/libimp_hip_fpprof.so python tools/probe/pose_phases.py"""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from imp_release_amd import pose
from oracle import pose_oracle as po
torch.zeros(1).cuda()
k0, k1, K, R, t, truth = po.synthetic_scene(1000, outliers=0.3, noise=0.3, seed=1)
pose.estimate_pose(k0, k1, K, K, 1.0)
torch.cuda.synchronize()
