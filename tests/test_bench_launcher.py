"""`python bench.py --gpus N` without a launcher around it: the script starts its own N rank processes (before anything touches the
GPU), relays rank 0's record and reports the world it really ran.  CPU-only: `--spawn-check` makes the ranks join
the group and all-reduce their rank numbers instead of training."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(argv), env=e, capture_output=True, text=True, timeout=300)


def test_gpus_2_spawns_two_ranks_and_prints_one_json_line():
    r = _run('--gpus', '2', '--backend', 'gloo', '--spawn-check')
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                     # library chatter goes to stderr
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['rank_sum'] == 3.0          # both ranks were in the all-reduce
    assert rec['backend'] == 'gloo' and rec['rccl_world'] is None  # a gloo world is not reported as an RCCL world


def test_launcher_always_runs_one_rank_through_the_same_path():
    r = _run('--gpus', '1', '--launcher', 'always', '--spawn-check')
    assert r.returncode == 0, r.stderr
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec['n_gpus'] == 1 and rec['backend'] is None and rec['rccl_world'] is None


def test_more_rccl_ranks_than_devices_is_refused_before_any_rank_starts():
    import torch
    if torch.cuda.device_count() >= 3:
        import pytest
        pytest.skip('needs a box with fewer than 3 devices')
    r = _run('--gpus', '3')
    assert r.returncode == 2 and 'device(s) visible' in r.stderr and r.stdout.strip() == ''


def test_world_size_that_contradicts_gpus_is_an_error_not_a_warning():
    r = _run('--gpus', '2', '--spawn-check', env={'WORLD_SIZE': '1', 'RANK': '0'})
    assert r.returncode != 0 and 'WORLD_SIZE=1' in r.stderr


def test_a_failing_rank_fails_the_launcher():
    # rank processes that cannot rendezvous with the requested backend must surface as a non-zero exit, not as a hang or a 0
    r = _run('--gpus', '2', '--backend', 'no_such_backend', '--spawn-check')
    assert r.returncode != 0


def test_one_dying_rank_ends_the_run_instead_of_hanging_the_others():
    """Rank 1 exits with status 7 before joining the group: rank 0 would wait in the rendezvous; the launcher must end it and return 7
    within seconds (CTGAN_TEST_DIE_RANK is read by bench.py's rank side in --spawn-check mode only)."""
    import time
    t0 = time.time()
    r = _run('--gpus', '2', '--backend', 'gloo', '--spawn-check', env={'CTGAN_TEST_DIE_RANK': '1'})
    assert r.returncode == 7, (r.returncode, r.stderr[-500:])
    assert time.time() - t0 < 60


def test_sigterm_to_the_launcher_ends_its_ranks(tmp_path):
    """A driver timeout sends SIGTERM to the launcher: the ranks it started - here rank 1 hangs before the rendezvous and rank 0 waits for
    it - must be gone when the launcher exits (they would otherwise hold the GPUs until the RCCL timeout)."""
    import signal
    import time
    e = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    pidfile = str(tmp_path / 'rank_pid')
    e.update(CTGAN_TEST_HANG_RANK='1', CTGAN_TEST_PID_FILE=pidfile)
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--spawn-check'], env=e,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    try:
        t0 = time.time()
        while not os.path.exists(pidfile + '.1') and time.time() - t0 < 120:
            time.sleep(0.2)
        assert os.path.exists(pidfile + '.1'), 'the hanging rank never started'
        time.sleep(0.5)
        rank_pid = int(open(pidfile + '.1').read())
        p.send_signal(signal.SIGTERM)
        p.wait(timeout=30)
        assert p.returncode == 128 + signal.SIGTERM, p.returncode
        time.sleep(0.5)
        alive = True
        try:
            os.kill(rank_pid, 0)
        except ProcessLookupError:
            alive = False
        if alive:       # a zombie of another parent would still answer kill(0): check its state
            try:
                state = open('/proc/%d/stat' % rank_pid).read().split()[2]
                alive = state != 'Z'
            except FileNotFoundError:
                alive = False
        assert not alive, 'rank process %d survived the launcher' % rank_pid
    finally:
        if p.poll() is None:
            p.kill()
            p.wait()


def test_ab_legs_start_after_the_record_and_a_hanging_leg_cannot_cost_it():
    """N > 1: the record is printed - and relayed by the launcher - BEFORE any A/B leg starts; a leg is a fresh group of rank
    processes started by the ranks of the finished main run, and one that hangs is killed after --leg-timeout while the run still exits 0
    with its one record line.  (--spawn-check: the same run_ab_legs / pick_ports code as the training bench, without a GPU.)"""
    import time
    r = _run('--gpus', '2', '--backend', 'gloo', '--spawn-check', env={'CTGAN_TEST_SPAWN_LEG': '1'})
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1 and json.loads(lines[0])['rank_sum'] == 3.0, r.stdout
    legs = [ln for ln in r.stderr.splitlines() if ln.startswith('bench: leg spawn: ')]
    assert len(legs) == 1, r.stderr[-2000:]
    leg = json.loads(legs[0][len('bench: leg spawn: '):])
    assert leg['leg'] == 'spawn' and leg['rank_sum'] == 3.0 and leg['master_port'] != json.loads(lines[0])['master_port']     # its own rendezvous
    # rank 1 of the leg never joins: both leg processes are killed at the timeout, the main record is untouched, exit status 0
    e = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    e.update(CTGAN_TEST_SPAWN_LEG='1', CTGAN_TEST_HANG_LEG='1')
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--spawn-check', '--leg-timeout', '8'],
                         env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    t0 = time.time()
    first = p.stdout.readline()                       # arrives while the leg is still hanging
    t_rec = time.time() - t0
    out, err = p.communicate(timeout=120)
    t_all = time.time() - t0
    assert json.loads(first)['rank_sum'] == 3.0
    assert p.returncode == 0, err[-2000:]
    assert t_all - t_rec > 6.0, (t_rec, t_all)        # the record was out (at least) the leg's timeout before the run ended
    legs = [ln for ln in err.splitlines() if ln.startswith('bench: leg spawn: ')]
    assert len(legs) == 1 and 'killed after --leg-timeout' in legs[0], err[-2000:]
    assert [ln for ln in out.splitlines() if ln.startswith('{')] == []
