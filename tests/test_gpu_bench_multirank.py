"""`bench.py --gpus 8` end to end on ONE device: eight rank processes (gloo, sharing the GPU) run the timed loop of the multi-GPU path -
flat gradient bucket, all-reduce on the side stream, 1/world folded into Adam, per-rank Philox streams - and the record says what ran:
n_gpus 8, backend gloo, rccl_world null (a gloo world is not an RCCL world), replicas bit-identical after the loop.  The launcher had
only ever run at N = 2 (VERDICT r3 next #7a); the first real 8-GPU run should not be the first run at N = 8."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_bench_gpus_8_gloo_on_one_device():
    e = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--backend', 'gloo', '--steps', '2', '--warmup', '1',
                        '--no-roofline', '--no-cpu-baseline', '--feed', 'device', '--no-ab-legs'], env=e, capture_output=True, text=True, timeout=580)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 8 and rec['scaling'] == 'weak'
    cfg = rec['config']
    assert cfg['backend'] == 'gloo' and cfg['rccl_world'] is None
    assert cfg['replicas_identical'] is True
    assert cfg['loss_sane'] is True
    assert cfg['collective']['all_reduces_per_step'] == 6
    assert rec['value'] > 0 and rec['steps'] == 2
    assert cfg['collective']['in_graph'] is False            # the record leg: eager side-stream collective between per-step graphs
    assert 0.0 < cfg['collective']['scaling_efficiency_bound'] <= 1.5


@pytest.mark.timeout(600)
def test_bench_gpus_2_gloo_record_first_then_the_split_flush_leg():
    """N = 2 over gloo on one device, with the A/B leg the gloo backend has (the eager engine with the split flush): the record line is the
    only JSON line on stdout, the leg - a fresh pair of rank processes started after it - reports on stderr with replicas still
    bit-identical (the legs cannot cost the record; on RCCL they are the captured collectives)."""
    e = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--steps', '3', '--warmup', '1',
                        '--no-roofline', '--no-cpu-baseline', '--feed', 'device'], env=e, capture_output=True, text=True, timeout=580)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['config']['replicas_identical'] is True and rec['config']['collective']['in_graph'] is False
    legs = [ln for ln in r.stderr.splitlines() if ln.startswith('bench: leg eager_split_flush: ')]
    assert len(legs) == 1, r.stderr[-3000:]
    leg = json.loads(legs[0][len('bench: leg eager_split_flush: '):])
    assert leg.get('error') is None and leg['split_flush'] is True and leg['hipgraph'] is False
    assert leg['replicas_identical'] is True and leg['loss_sane'] is True and leg['ms_per_step'] > 0
