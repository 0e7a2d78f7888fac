"""bench.py host logic that needs no GPU: the self-spawn of `--gpus N`."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_spawns_its_own_ranks_when_no_launcher_is_present():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--steps', '3', '--warmup', '1',
                          '--dry-run-spawn'], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    cmd = json.loads(out.stdout.strip().splitlines()[-1])['spawn']
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node=4' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    i = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[i + 1:i + 3] == ['--gpus', '4'] and '--steps' in cmd[i:]


def test_roofline_peak_is_the_executing_unit():
    sys.path.insert(0, ROOT)
    import bench
    m = dict(cfg=bench.WORKLOADS['airplane'], achieved=200.0, kern_ms=0.5, pts_per_launch=4 * 64 * 2048)
    r = bench.roofline_record('airplane', m)
    assert abs(r['peak'] - 2500.0 / 3) < 0.1 and r['frac'] < 1 and abs(r['frac'] - 200.0 / (2500.0 / 3)) < 1e-3
    assert r['ratio_vs_fp32_mfma'] > 1            # the secondary ratio may exceed 1: that unit is not the one executing
