"""Where the host time of an EAGER airplane train step goes (cProfile over tools/bench_train.py's step function).
    python tools/diag/eager_cpu_profile.py [--api fused|swap] [--rows 45]"""
import cProfile
import pstats
import runpy
import sys

api = sys.argv[sys.argv.index('--api') + 1] if '--api' in sys.argv else 'fused'
rows = int(sys.argv[sys.argv.index('--rows') + 1]) if '--rows' in sys.argv else 45
sys.argv = ['bench_train.py', '--steps', '3', '--api', api]
ns = runpy.run_path('tools/bench_train.py', run_name='bench_train_profiled')
import torch

step = ns['step']
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(rows)
st.sort_stats('tottime').print_stats(25)
