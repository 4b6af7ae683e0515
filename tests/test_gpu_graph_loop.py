"""The BENCHMARKED loop - hipGraph replay of the fused critic / generator steps in the reference's order
([G] + N_CRITIC x D, TF/CT_gan_cifar_resnet.py:393-404) - checked value by value over several iterations.

Round 1 timed a loop whose critic cost ran to -6e18: the three graphs shared one memory pool and were replayed in an order
other than the capture order, so the fake batches of critic steps 2..5 were overwritten by the first critic replay
(engine.GraphedTrainer._capture).  No test looked at a VALUE of the graphed loop.  These do:
  * every loss term of every critic step of the replayed loop equals the eager loop on the same Philox streams (same kernels
    in the same order - the comparison is tight), and stays in a sane band;
  * the fake batches the critic replays read are tanh outputs (|x| <= 1) - the direct symptom of the round-1 aliasing;
  * the all-switches-off op-by-op eager loop agrees with the default fused loop (free-running drift tolerance);
  * a GraphedTrainer resumed from a checkpoint continues bit-exactly (capture leaves no trace in optimizer / RNG state).
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

TERMS = ('cost', 'wgan', 'ct', 'gp', 'acgan')


def _batches(B, n=16, seed=1234):
    import numpy as np
    nrng = np.random.default_rng(seed)
    return [(torch.from_numpy(nrng.integers(0, 256, (B, 3072), dtype=np.int32)).cuda(),
             torch.from_numpy(nrng.integers(0, 10, (B,), dtype=np.int32)).cuda()) for _ in range(n)]


def _run_loop(dim, B, iters, graphs, batches, seed=0, start=1):
    """bench.py's loop, recording every critic step.  -> (records, final critic theta, final generator theta)"""
    import ctgan_amd.gan_cifar_resnet as R
    import ctgan_amd.tflib as lib
    from ctgan_amd.engine import GraphedTrainer
    lib.delete_all_params(); lib.set_device(None); lib.set_seed(seed)
    R.configure(DIM_G=dim, DIM_D=dim, BATCH_SIZE=B)
    R.build_params()
    tr = R.Trainer(seed=2024)
    eng = GraphedTrainer(tr, use_graphs=graphs)
    assert eng.graphed == graphs, eng.graph_error
    cur = [0]

    def nb():
        cur[0] = (cur[0] + 1) % len(batches)
        return batches[cur[0]]
    recs = []
    for it in range(start, start + iters):
        g_cost = float(eng.g_step(it)['cost'].item())
        bs = [nb() for _ in range(R.cfg.N_CRITIC)]
        if eng.graphed:
            for i, (_, lab) in enumerate(bs):
                eng.labels_all[i * B:(i + 1) * B].copy_(lab)
            eng.f_graph.replay()
            fakes = eng.fake_all
        else:
            fakes = tr.generate_fakes(torch.cat([lab for _, lab in bs], 0))
        for i, (x, lab) in enumerate(bs):
            out = eng.d_step(x, lab, it, fake=fakes[i])
            rec = {k: float(out[k].item()) for k in TERMS}
            rec['fake_max'] = float(fakes[i].abs().max().item())      # read AFTER the replay that could clobber it
            rec['g_cost'] = g_cost
            recs.append(rec)
    th = (tr.d_opt.theta.clone(), tr.g_opt.theta.clone())
    lib.delete_all_params(); R.configure()
    return recs, th[0], th[1]


@pytest.mark.parametrize('dim,B,iters', [(32, 8, 12), (128, 64, 8)])
def test_graph_replay_loop_equals_eager_loop(dim, B, iters):
    batches = _batches(B)
    g_recs, g_d, g_g = _run_loop(dim, B, iters, True, batches)
    e_recs, e_d, e_g = _run_loop(dim, B, iters, False, batches)
    assert len(g_recs) == len(e_recs) == 5 * iters
    for n, (a, b) in enumerate(zip(g_recs, e_recs)):
        assert a['fake_max'] <= 1.0 and b['fake_max'] <= 1.0, 'critic step %d read a fake batch that is not a tanh output: %r' % (n, a)
        for k in TERMS:
            assert math.isfinite(a[k]) and abs(a[k]) < 1e3, 'step %d %s = %r' % (n, k, a[k])
            # identical kernels, launch order and Philox streams: anything beyond round-off of the printed scalars is a bug
            assert abs(a[k] - b[k]) <= 1e-5 * max(1.0, abs(b[k])), 'step %d %s: graph %r eager %r' % (n, k, a[k], b[k])
    assert torch.equal(g_d, e_d) and torch.equal(g_g, e_g)              # the weights after the loop are the same bits


def test_fused_loop_tracks_op_by_op_loop(monkeypatch):
    """Default switches (tape, shared tail, fused heads, grouped weight gradients) vs every switch off, both eager, free
    running over 6 iterations at reduced width: same Philox streams, different kernels/summation orders."""
    import ctgan_amd.functional as F
    import ctgan_amd.gan_cifar_resnet as R
    B, dim, iters = 8, 32, 6
    batches = _batches(B)
    fused, _, _ = _run_loop(dim, B, iters, False, batches)
    for mod, names in ((R, ('HEAD_FUSION', 'TRUNK_SHARE', 'TAIL_SHARE', 'PREP_FUSION', 'DROP_FUSION')),
                       (F, ('WGRAD_GROUPED', 'FEWCH_DEFER', 'DEFER_WGRADS'))):
        for n in names:
            monkeypatch.setattr(mod, n, False)
    plain, _, _ = _run_loop(dim, B, iters, False, batches)
    for n, (a, b) in enumerate(zip(fused, plain)):
        for k in TERMS:
            # free-running drift over 30 critic updates (Adam's early steps are sign-like, so round-off differences move
            # weights by whole steps): the WGAN / ACGAN terms stay within 5e-3; the penalty terms are means of squared small
            # differences ((|grad| - 1)^2, (D - D')^2) and amplify it - measured 6e-3 on gp at step 15
            tol = 3e-2 if k in ('gp', 'ct') else 5e-3
            assert abs(a[k] - b[k]) <= tol * max(1.0, abs(b[k])), 'step %d %s: fused %r op-by-op %r' % (n, k, a[k], b[k])


def test_graphed_trainer_resumes_bit_exactly(tmp_path):
    """checkpoint.load ran before GraphedTrainer._capture, whose warm-up advanced the Philox counter and zeroed the
    optimizers' step counts.  Capture now restores all of it: 2 iterations + save + 1 iteration == load + 1 iteration."""
    import ctgan_amd.gan_cifar_resnet as R
    import ctgan_amd.tflib as lib
    from ctgan_amd import checkpoint
    from ctgan_amd.engine import GraphedTrainer
    B, dim = 8, 16
    batches = _batches(B, n=4, seed=7)

    def run(n_iters, resume_from=None, save_at=None):
        lib.delete_all_params(); lib.set_device(None); lib.set_seed(4)
        R.configure(DIM_G=dim, DIM_D=dim, BATCH_SIZE=B)
        R.build_params()
        tr = R.Trainer(seed=9)
        start = checkpoint.load(resume_from, tr) if resume_from else 0
        eng = GraphedTrainer(tr)
        assert eng.graphed, eng.graph_error
        k = [start * 5]

        def nxt():
            k[0] += 1
            return batches[k[0] % 4]
        for it in range(start, n_iters):
            eng.train_iteration(it, nxt)
            if save_at is not None and it + 1 == save_at:
                checkpoint.save(str(tmp_path / 'ck.pt'), tr, iteration=it + 1)
        return (tr.d_opt.theta.clone(), tr.g_opt.theta.clone(), tr.d_opt.m.clone(), tr.d_opt.v.clone(), tr.rng.ctr.clone(),
                tr.d_opt.t, tr.g_opt.t)
    try:
        a = run(3, save_at=2)
        b = run(3, resume_from=str(tmp_path / 'ck.pt'))
        for x, y in zip(a[:5], b[:5]):
            assert torch.equal(x, y)
        assert a[5:] == b[5:] == (15, 2)
    finally:
        lib.delete_all_params(); R.configure()


def _oracle_loop(dim, B, iters, batches_cpu, seed=0, dtype=torch.float64):
    from oracle import loop, nets as onets, tflib_ref as oref
    reg = oref.Registry(dtype=dtype, seed=seed)
    cfg = onets.ResnetCfg(DIM_G=dim, DIM_D=dim)
    lab0 = torch.zeros(2, dtype=torch.int32)
    onets.resnet_discriminator(reg, cfg, onets.resnet_generator(reg, cfg, 2, lab0, torch.zeros(2, 128, dtype=dtype)), lab0, 1., 1., 1.)
    cur = [0]

    def nb():
        cur[0] = (cur[0] + 1) % len(batches_cpu)
        return batches_cpu[cur[0]]
    return loop.resnet_train_loop(reg, cfg, nb, iters, B, 2024, start_iteration=1, dtype=dtype)[:2]


def _fixture_loop(path, dim, B, iters):
    import numpy as np
    z = np.load(path)
    cfg = [int(v) for v in z['cfg']]
    assert cfg[:2] == [dim, B] and cfg[2] >= iters and cfg[3:] == [2024, 0], cfg       # (dim, B, iters, Philox seed, init seed)
    keys = [str(k) for k in z['keys']]
    d = [dict(zip(keys, (float(v) for v in row))) for row in z['d'][:5 * iters]]
    return d, [float(v) for v in z['g'][:iters]]


@pytest.mark.parametrize('dim,B,iters', [(32, 8, 2), (128, 64, 2)])
def test_graph_replay_loop_matches_oracle_loop(dim, B, iters):
    """The benchmarked loop (hipGraph replay, all fusions, in-kernel Philox) against the oracle's restatement of the
    reference loop as written, FREE RUNNING from the same initial weights, batches and Philox streams, in the reference's
    order [G] + 5 x D with LR decay (TF/CT_gan_cifar_resnet.py:393-404).

    Free running means round-off differences are fed back through Adam, whose early steps are sign-like (a weight whose
    gradient is within fp32 noise of zero moves a full +-lr either way), so ANY fp32 evaluation of the graph leaves the fp64
    trajectory step by step.  The yardstick is therefore the oracle's own fp32 twin run on the same streams: the device
    must stay within the north star's 1e-3 of the fp64 truth, or within 3x the drift the fp32 twin has accumulated by that
    step, whichever is larger (terms scaled by max(1, |wgan term|): `cost` is a small difference of O(1..10) terms)."""
    import json
    import os
    batches = _batches(B)
    cpu = [(x.cpu(), y.cpu()) for x, y in batches]
    got, _, _ = _run_loop(dim, B, iters, True, batches)
    fix = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'resnet_loop_%d_%d.npz' % (dim, B))
    if os.path.exists(fix):
        # the oracle loop at this size takes minutes of host time per run (fp64 + its fp32 twin): its trace is a committed fixture
        # (tests/golden/make_golden.py loopfull - seeds, not tensors: same initial weights, batches and Philox streams as _run_loop)
        d_ref, g_ref = _fixture_loop(fix, dim, B, iters)
        d_twin, g_twin = _fixture_loop(fix.replace('.npz', '_f32twin.npz'), dim, B, iters)
    else:
        d_ref, g_ref = _oracle_loop(dim, B, iters, cpu)
        d_twin, g_twin = _oracle_loop(dim, B, iters, cpu, dtype=torch.float32)
    assert len(got) == len(d_ref) == 5 * iters and len(g_ref) == iters
    rows, twin_max = [], 0.0
    for n, (a, b, t) in enumerate(zip(got, d_ref, d_twin)):
        scale = max(1.0, abs(b['wgan_only']), abs(b['gp']))
        e_dev = max(abs(a[k] - b[k]) for k in TERMS) / scale
        twin_max = max(twin_max, max(abs(t[k] - b[k]) for k in TERMS) / scale)
        rows.append({'step': n, 'dev_rel_err': e_dev, 'twin_rel_err_running_max': twin_max, 'cost_dev': a['cost'], 'cost_ref': b['cost']})
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/loop_vs_oracle_%d_%d.json' % (dim, B), 'w') as f:
        json.dump(rows, f, indent=1)
    for r in rows:
        # From critic step 7 on the loop is in its chaotic regime and the twin is ONE sample of what fp32 evaluations do there - on another
        # host CPU the same twin lands elsewhere (step 9: 3.2e-3 on the GPU box's CPU in round 2, 4.8e-4 in the container that wrote the
        # committed fixture; the device: 2.3e-3 - 3.9e-3).  The envelope doubles per step like the measured amplification.
        envelope = 1.5e-3 * 2.0 ** (r['step'] - 7) if r['step'] >= 7 else 0.0
        assert r['dev_rel_err'] <= max(1e-3, 3.0 * r['twin_rel_err_running_max'], envelope), rows
    assert rows[0]['dev_rel_err'] <= 2e-4                      # the first critic step is a pure single-step comparison
    for i in range(iters):
        a, b, t = got[5 * i]['g_cost'], g_ref[i], g_twin[i]
        # + the critic's output bias: a null direction of the critic loss that random-walks by O(lr) per critic step in fp32
        assert abs(a - b) <= max(1e-3 * max(1.0, abs(b)), 3.0 * abs(t - b)) + 3e-4 * 5 * (i + 1), 'generator step %d: device %r oracle %r' % (i, a, b)


def test_thousand_iteration_curves_track_the_fp64_oracle_as_closely_as_its_own_fp32_twin():
    """What is asserted: (i) every loss term of the first 8 critic steps within 1e-3 relative of the fp64 oracle (pointwise); (ii) over
    1,000 iterations the 50-iteration window MEANS of every term within max(20 %, 2 x the oracle's own fp32 twin's deviation) of the
    fp64 curve.  That is NOT the north star's "within 1e-3 relative over 1k steps" read pointwise - no fp32 evaluation of this
    chaotic loop can meet that (the fp32 twin of the oracle itself leaves the fp64 trajectory at step 11-29, see below) - it is the
    strongest statement the arithmetic admits, and the test is named after it.
    north_star: "D/G loss curves within 1e-3 relative of the reference over 1k steps".  tests/golden/resnet_loop_trace.npz
    is the oracle's fp64 free-running trace of 1,000 iterations (6,000 optimizer steps) at DIM 32 / B 8 (seeds, not tensors:
    make_golden.py loop_trace_fixture); the device replays the same loop through GraphedTrainer in fp32.  A GAN's training
    trajectory is chaotic - two fp32 evaluations of the same graph (a different summation order is enough) separate
    exponentially - so a pointwise 1e-3 bound can only hold until round-off differences have been amplified to that level;
    the curves are then compared as CURVES (windowed means).  The measured drift is written to gpurun_out/loop_drift.json
    and summarised in DESIGN.md section 2."""
    import json
    import os

    import numpy as np
    fx = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'resnet_loop_trace.npz'))
    dim, B, iters = (int(v) for v in fx['cfg'][:3])
    keys = [str(k) for k in fx['keys']]
    ref = fx['d']                                            # [5 * iters, len(keys)]
    got, _, _ = _run_loop(dim, B, iters, True, _batches(B))
    dev = np.array([[r[k] for k in keys if k != 'wgan_only'] for r in got])
    cols = [i for i, k in enumerate(keys) if k != 'wgan_only']
    ref = ref[:, cols]
    names = [keys[i] for i in cols]
    assert np.isfinite(dev).all() and np.abs(dev).max() < 1e3
    rel = np.abs(dev - ref) / np.maximum(1.0, np.abs(ref))
    g_dev = np.array([got[5 * i]['g_cost'] for i in range(iters)])
    g_rel = np.abs(g_dev - fx['g']) / np.maximum(1.0, np.abs(fx['g']))
    first_over = {n: int(np.argmax(rel[:, j] > 1e-3)) if (rel[:, j] > 1e-3).any() else len(rel) for j, n in enumerate(names)}
    win = 250                                                # critic steps per window (50 iterations)
    wm = lambda a: a[:len(a) // win * win].reshape(-1, win, a.shape[1]).mean(1)
    wrel = np.abs(wm(dev) - wm(ref)) / np.maximum(1.0, np.abs(wm(ref)))
    out = {'dim': dim, 'B': B, 'iters': iters, 'terms': names,
           'first_critic_step_over_1e-3': first_over,
           'max_rel_by_step_decade': {str(hi): {n: float(rel[:hi, j].max()) for j, n in enumerate(names)} for hi in (10, 100, 1000, len(rel))},
           'windowed_mean_rel_err_max': {n: float(wrel[:, j].max()) for j, n in enumerate(names)},
           'g_cost_max_rel_first_10_100_all': [float(g_rel[:10].max()), float(g_rel[:100].max()), float(g_rel.max())],
           'cost_curve_dev_window_means': [float(v) for v in wm(dev)[:, names.index('cost')]],
           'cost_curve_ref_window_means': [float(v) for v in wm(ref)[:, names.index('cost')]]}
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/loop_drift.json', 'w') as f:
        json.dump(out, f, indent=1)
    # the oracle's own fp32 twin on the same streams (make_golden.py loop32): the drift ANY fp32 evaluation shows
    twin_path = os.path.join(os.path.dirname(__file__), 'golden', 'resnet_loop_trace_f32twin.npz')
    twin_w = None
    if os.path.exists(twin_path):
        tw = np.load(twin_path)['d'][:, cols]
        trel = np.abs(tw - ref) / np.maximum(1.0, np.abs(ref))
        twin_w = np.abs(wm(tw) - wm(ref)) / np.maximum(1.0, np.abs(wm(ref)))
        out['fp32_twin'] = {'first_critic_step_over_1e-3': {n: int(np.argmax(trel[:, j] > 1e-3)) if (trel[:, j] > 1e-3).any() else len(trel)
                                                            for j, n in enumerate(names)},
                            'max_rel_by_step_decade': {str(hi): {n: float(trel[:hi, j].max()) for j, n in enumerate(names)} for hi in (10, 100, 1000, len(trel))},
                            'windowed_mean_rel_err_max': {n: float(twin_w[:, j].max()) for j, n in enumerate(names)}}
        with open('gpurun_out/loop_drift.json', 'w') as f:
            json.dump(out, f, indent=1)
    # Measured (profiles/history/r02_loop_drift.json): every term within 1e-3 for the first 10 critic steps (2 iterations), 1e-1 ..
    # 3e-1 after 100 steps, O(1) pointwise after 1,000 iterations (different trajectory of the same chaotic system); the
    # CURVES agree: 50-iteration window means of cost / wgan within 7 % / 11 %, acgan 0.3 %, ct 3 %, gp 2 %.
    assert rel[:8].max() <= 1e-3, out['max_rel_by_step_decade']
    assert g_rel[:2].max() <= 1e-3 + 3e-3
    for j, n in enumerate(names):
        bound = 0.2 if twin_w is None else max(0.2, 2.0 * float(twin_w[:, j].max()))
        assert wrel[:, j].max() <= bound, (n, float(wrel[:, j].max()), bound)


@pytest.mark.parametrize('dim,B,iters', [(32, 8, 6), (128, 64, 4)])
def test_whole_iteration_graph_equals_eager_loop(dim, B, iters):
    """engine.GraphedTrainer.train_iteration replays ONE hipGraph per iteration (generator step + the fake batches + N_CRITIC
    critic steps; world == 1) - what bench.py times.  Against the eager Trainer.train_iteration on the same batches and Philox
    streams: the last critic step's loss terms of every iteration and the final weights are the same bits."""
    import ctgan_amd.gan_cifar_resnet as R
    import ctgan_amd.tflib as lib
    from ctgan_amd.engine import GraphedTrainer
    batches = _batches(B)

    def run(graphs):
        lib.delete_all_params(); lib.set_device(None); lib.set_seed(0)
        R.configure(DIM_G=dim, DIM_D=dim, BATCH_SIZE=B)
        R.build_params()
        tr = R.Trainer(seed=2024)
        eng = GraphedTrainer(tr, use_graphs=graphs)
        assert eng.graphed == graphs, eng.graph_error
        if graphs:
            assert eng.it_graph is not None
        cur = [0]

        def nb():
            cur[0] = (cur[0] + 1) % len(batches)
            return batches[cur[0]]
        recs = []
        for it in range(0, iters):                   # iteration 0 has no generator step: per-step graphs; then the iteration graph
            out = eng.train_iteration(it, nb)
            recs.append({k: float(out[k].item()) for k in TERMS})
        res = (recs, tr.d_opt.theta.clone(), tr.g_opt.theta.clone(), tr.d_opt.t, tr.g_opt.t, int(tr.rng.ctr.item()))
        lib.delete_all_params(); R.configure()
        return res
    g, e = run(True), run(False)
    assert g[3:] == e[3:] == (5 * iters, iters - 1, 7 * iters - 1)
    for n, (a, b) in enumerate(zip(g[0], e[0])):
        for k in TERMS:
            assert math.isfinite(a[k]) and abs(a[k]) < 1e3
            assert abs(a[k] - b[k]) <= 1e-5 * max(1.0, abs(b[k])), 'iteration %d %s: graph %r eager %r' % (n, k, a[k], b[k])
    assert torch.equal(g[1], e[1]) and torch.equal(g[2], e[2])


def test_all_reduce_captured_in_the_step_graphs():
    """CTGAN_AR_IN_GRAPH (engine.GraphedTrainer(ar_in_graph=True)): the gradient all-reduce and the Adam step captured INSIDE the step /
    iteration graphs, so that the multi-GPU loop is the single-GPU one-graph-per-iteration replay.  On one GPU: a 1-rank RCCL group
    (tests/ar_in_graph_check.py in a subprocess) - the loop with the captured collective must end with bit-identical weights to the
    loop without any collective, and a finite cost."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(here, 'ar_in_graph_check.py'), '32', '8', '3', '64', '8'], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert rec['backend'] == 'nccl' and rec['d_equal'] and rec['g_equal'], rec
    assert math.isfinite(rec['cost_in_graph']) and rec['cost_plain'] == rec['cost_in_graph'], rec
    # grad_scale = 1 / world != 1 through the in-graph path (gather, all-reduce, Adam with the average folded in)
    assert rec['scaled_d_equal'] and rec['scaled_g_equal'] and rec['scaled_differs_from_unscaled'], rec
    # Trainer.split_flush under capture (round 5): the critic bucket's prefix on its own collective, forked onto a side stream inside the
    # graph and joined before Adam.  Two iterations against the one-bucket form: the two grouped weight-gradient launches plan their
    # split-K chunks separately (another fp32 summation order), and Adam's first steps are sign-like, so single near-zero-gradient
    # entries may move by a step in the other direction - the weights agree to 2e-5 but for a small fraction of entries, never by more
    # than a few learning rates; the generator (whose gradients do not pass through the split) agrees to rounding through the critic.
    sp = rec['split_flush']
    assert sp is not None and math.isfinite(sp['cost_split']) and abs(sp['cost_split'] - sp['cost_one_bucket']) <= 1e-3 * max(1.0, abs(sp['cost_one_bucket'])), sp
    assert sp['theta_max_abs_diff'] <= 2e-3 and sp['theta_frac_above_2e-5'] <= 0.02, sp
