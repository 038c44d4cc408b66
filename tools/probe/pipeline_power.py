"""Clock / power trace of the WHOLE benchmark pipeline (not one kernel back to back): is the step power-limited, and at which
sclk do its kernels run?  rocm-smi is sampled every ~0.25 s while `bench.py --steps S` runs in a child process.
    python tools/probe/pipeline_power.py [in_flight=3] [steps=1500] [extra bench flags ...]"""
import os, subprocess, sys, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
k = sys.argv[1] if len(sys.argv) > 1 else '3'
steps = sys.argv[2] if len(sys.argv) > 2 else '1500'
cmd = [sys.executable, os.path.join(root, 'bench.py'), '--steps', steps, '--warmup', '10', '--no-cpu-baseline', '--no-batch1', '--quick-c5', '--in-flight', k] + sys.argv[3:]
t0 = time.time()
child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
rows = []
while child.poll() is None:
    try:
        r = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--csv'], capture_output=True, text=True, timeout=5)
        ln = [l for l in r.stdout.strip().splitlines() if l.startswith('card0')]
        rows.append((time.time() - t0, ln[0] if ln else r.stdout.strip()[:200]))
    except Exception as e:      # noqa: BLE001
        rows.append((time.time() - t0, f'rocm-smi failed: {e}'))
    time.sleep(0.25)
out = child.stdout.read().strip().splitlines()
print('fields: fclk, level, mclk, level, sclk, level, socclk, level, package W')
for t, l in rows:
    print(f'  t={t:6.2f}s  {l}')
print(out[-1][:300] if out else 'no bench output')
