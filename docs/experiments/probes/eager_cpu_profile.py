"""Where does the HOST time of one eager airplane training step go?  (cProfile of bench_train's step, GPU synchronised)"""
import cProfile, pstats, sys, torch, importlib.util
sys.argv = ['bench_train.py', '--steps', '3']
sys.path.insert(0, '.')
spec = importlib.util.spec_from_file_location('bt', 'tools/bench_train.py')
bt = importlib.util.module_from_spec(spec); spec.loader.exec_module(bt)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    bt.step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr).sort_stats('cumulative')
st.print_stats(45)
