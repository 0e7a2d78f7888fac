import sys, cProfile, pstats, torch, runpy
sys.argv = ['bench_train.py', '--steps', '1']
sys.path.insert(0, '.')
ns = runpy.run_path('tools/bench_train.py', run_name='notmain')
step = ns['step']
for _ in range(2): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(28)
