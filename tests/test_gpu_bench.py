"""bench.py end to end on the GPU box: the single-GPU line and the self-spawned 2-rank rehearsal (two processes on the one
card, gloo for the barrier / max-over-ranks because RCCL refuses two ranks on one device)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '5', '--warmup', '3', '--no-cpu-baseline'] + extra,
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_single_gpu_line_has_roofline_and_m1():
    line = _run([])
    assert line['n_gpus'] == 1 and line['value'] > 0 and line['scaling'] == 'weak'
    r = line['roofline']
    assert r['bound'] == 'mfma' and 0 < r['frac'] < 1 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    assert line['also']['m1']['value'] > 0 and 0 < line['also']['m1']['roofline']['frac'] < 1
    # roofline.traffic of the headline kernel is measured by the run itself (two rocprofv3 --pmc child passes): HBM-side bytes per
    # launch between the algorithmic 18.9 MB (36 B x 524 288 point passes) and a few times that
    assert 'measured by this run' in r['traffic_source'], r['traffic_source']
    assert 18.0e6 < r['traffic'] < 80.0e6, r['traffic']
    m1r = line['also']['m1']['roofline']                                          # the metric's own shape: measured live as well
    assert 'measured by this run' in m1r['traffic_source'] and 2.3e6 < m1r['traffic'] < 30.0e6, m1r
    for w in ('ae', 'svr', 'k16', 'k16_b1'):                                      # every shape's traffic is measured by the run itself
        rw = line['also'][w]['roofline']
        assert 'measured by this run' in rw['traffic_source'] and 1.0e6 < rw['traffic'] < 60.0e6, (w, rw)
    x = line['also']['exact_fp32']                                                # the fp32-MFMA comparison point and the re-run launch's cost
    assert 0.2 < x['frac_of_fp32_mfma_peak'] < 1.0 and 1.2 < x['split_over_exact'] < 6.0 and -2.0 < x['rerun_launch_us'] < 15.0, x
    ts = line['also']['train_step']                      # the whole training step, timed in a child process
    assert 'error' not in ts and 5 < ts['ms_per_step'] < 60, ts
    # ... and the data-parallel code path on a 1-rank RCCL group: every collective inside the graph, at most 15 % slower
    dp = ts['data_parallel_path_1rank']
    assert 'error' not in dp and dp['statistic_all_reduces_in_graph'] == 132 and dp['ms_per_step'] < 1.15 * ts['ms_per_step'], ts
    sw = ts['import_swap_only']                          # the literal decoder-import swap: K sequential calls -> one K-batched pass
    assert sw['with_this_loss']['ms_per_step'] < 0.6 * sw['with_this_loss_no_sibling_batching']['ms_per_step'], sw
    assert sw['reference_loss_loop']['ms_per_step'] > sw['with_this_loss']['ms_per_step'], sw
    ae = ts['ae_shard']                                  # configs[2]'s per-rank shard (16 shapes, G = 512) through the same path
    assert 'error' not in ae and 2 < ae['plain_ms_per_step'] <= ae['data_parallel_path_1rank_ms_per_step'] < 40, ae


def test_gpus_2_spawns_itself_and_reports_the_aggregate():
    line = _run(['--gpus', '2', '--backend', 'gloo', '--share-device', '--no-also'])
    assert line['n_gpus'] == 2 and line['value'] > 0
    # value = points of BOTH ranks / max-over-ranks time
    pts = 2 * line['config']['per_gpu_batch'] * line['config']['points_per_shape'] * line['config']['components']
    assert abs(line['value'] - pts / (line['ms_per_step'] * 1e-3) / 1e6) / line['value'] < 1e-2


def test_gpus_2_training_step_record_rehearsal():
    """bench.py --gpus 2 also times the data-parallel training step (one child per rank, the children form their own group):
    rehearsed on the one card over gloo (eager step: gloo's collectives are host-side, nothing to capture)."""
    line = _run(['--gpus', '2', '--backend', 'gloo', '--share-device', '--also-select', 'train_step', '--train-step-steps', '2'])
    ts = line['also']['train_step']
    for label, b in (('global_batch_64', 32), ('per_rank_batch_64', 64)):
        assert 'error' not in ts[label] and ts[label]['per_rank_batch'] == b and ts[label]['ms_per_step'] > 0, ts
