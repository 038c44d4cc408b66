"""attention launch time at the bench geometry with the clock it ran at (library = IMP_HIP_LIB or the default):  python tools/probe/attn_time.py [B N reps rounds]"""
import statistics
import sys

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import eval_config, make_hip_model  # noqa: E402
from imp_release_amd import synthetic  # noqa: E402

B, N, reps, rounds = (int(a) for a in (sys.argv[1:5] + ['4', '2048', '40', '6'][len(sys.argv) - 1:]))
cfg = eval_config(n_layers=1)
m = make_hip_model('GM', cfg, synthetic.make_state_dict(cfg, 'GM', seed=0))
ctx = m._ensure_ctx()
ctx.time_attention_clock(B, N, 10)
r = [ctx.time_attention_clock(B, N, reps) for _ in range(rounds)]
print(f'attention B={B} N={N}: median {statistics.median(x[0] for x in r) * 1e3:.2f} us (min {min(x[0] for x in r) * 1e3:.2f}), clock median {statistics.median(x[1] for x in r):.0f} MHz')
