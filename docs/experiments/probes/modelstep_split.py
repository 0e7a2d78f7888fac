import sys, time, torch, runpy
sys.argv = ['bench_train.py', '--steps', '1']
sys.path.insert(0, '.')
ns = runpy.run_path('tools/bench_train.py', run_name='notmain')
model, crit, opt, g_in, p_in = ns['model'], ns['crit'], ns['opt'], ns['g_in'], ns['p_in']
def sync(): torch.cuda.synchronize(); return time.perf_counter()
acc = {'fwd_host': 0, 'fwd_total': 0, 'bwd_host': 0, 'bwd_total': 0, 'opt_host': 0, 'opt_total': 0, 'zero': 0}
n = 5
for i in range(n + 2):
    t0 = sync()
    opt.zero_grad(set_to_none=True)
    t1 = time.perf_counter()
    enc, dec = model.forward_fused(g_in, p_in)
    loss = crit.fused(enc, dec)[0]
    t2 = time.perf_counter(); t3 = sync()
    loss.backward()
    t4 = time.perf_counter(); t5 = sync()
    opt.step()
    t6 = time.perf_counter(); t7 = sync()
    if i >= 2:
        for k, v in (('zero', t1 - t0), ('fwd_host', t2 - t1), ('fwd_total', t3 - t1), ('bwd_host', t4 - t3), ('bwd_total', t5 - t3),
                     ('opt_host', t6 - t5), ('opt_total', t7 - t5)):
            acc[k] += v / n * 1e3
    del loss, enc, dec
print({k: round(v, 2) for k, v in acc.items()})
