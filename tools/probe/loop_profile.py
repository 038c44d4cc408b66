"""cProfile of the iterative loop with ONE worker (host-side cost of an evaluation: where does a pair's wall time go?)
    python tools/probe/loop_profile.py [IMP|EIMP] [pairs]"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = ['eval_synthetic.py', '--pairs', sys.argv[2] if len(sys.argv) > 2 else '80', '--model', sys.argv[1] if len(sys.argv) > 1 else 'IMP', '--kpts', '2048', '--workers', '1', '--pose', 'gpu']
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'eval_synthetic.py'), run_name='__main__')
finally:
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
    print(s.getvalue()[:6000])
