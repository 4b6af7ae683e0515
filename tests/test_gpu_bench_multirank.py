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
                        '--no-roofline', '--no-cpu-baseline', '--feed', 'device'], env=e, capture_output=True, text=True, timeout=580)
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
