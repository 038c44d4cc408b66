"""sclk / power while the pose step runs back to back (is the small-kernel chain of a pose call clocked down?), alone and with a
second stream keeping the chip busy.   python tools/probe/pose_clock.py"""
import subprocess, sys, threading, time
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from imp_release_amd import pose
from oracle import pose_oracle as po
torch.zeros(1).cuda()
k0, k1, K, R, t, truth = po.synthetic_scene(1000, outliers=0.3, noise=0.3, seed=1)

def sample(stop, out):
    while not stop.is_set():
        r = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--csv'], capture_output=True, text=True, timeout=5)
        ln = [l for l in r.stdout.splitlines() if l.startswith('card0')]
        out.append(ln[0] if ln else '?')
        time.sleep(0.3)

def run(tag, busy):
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out)); th.start()
    bstop = threading.Event()
    def burn():
        s = torch.cuda.Stream()
        a = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16)
        with torch.cuda.stream(s):
            while not bstop.is_set():
                for _ in range(20): a @ a
                s.synchronize()
    bt = threading.Thread(target=burn) if busy else None
    if bt: bt.start()
    time.sleep(0.5)
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 2.5:
        pose.estimate_pose(k0, k1, K, K, 1.0); n += 1
    dt = (time.perf_counter() - t0) / n
    bstop.set(); stop.set(); th.join()
    if bt: bt.join()
    print(f'== {tag}: {dt * 1e3:.3f} ms per pose call')
    for l in out: print('   ', l)

run('pose alone', False)
run('pose + a bf16 GEMM stream keeping the chip busy', True)
