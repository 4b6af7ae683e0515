"""Fuzz of every host-side planner behind a `*_workspace_bytes` / routing query of the C-ABI (no launch, no GPU): random geometries and
job tables - 1-3 segments per filter, tiny and huge row counts, odd image sizes, channel counts on and off the tile grid - each call
under a hard timeout in a child process.  A planner that spins (round 3: `multi_plan` in csrc/igemm.hip never returned for more segments
than planned splits) shows up here as a timeout instead of a hung GPU suite.  Also checks the invariants callers rely on: the grouped
query is deterministic (also through the per-thread memo of the last eight tables), monotone in what it covers (slabs for every split
fit), and agrees between a group of one and the same problem inside a larger group's total."""
import ctypes
import multiprocessing as mp
import random

import pytest


def _geom_cases(rng, n):
    out = []
    for _ in range(n):
        C = rng.choice([3, 32, 64, 96, 128, 160, 256, 512, 1024])
        Ko = rng.choice([1, 3, 10, 32, 64, 128, 256, 512, 1024])
        H = rng.choice([1, 2, 4, 7, 8, 14, 16, 28, 32, 64, 128])
        W = H if rng.random() < 0.8 else rng.choice([4, 8, 12, 16, 32])
        k = rng.choice([1, 2, 3, 4, 5])
        st = rng.choice([1, 1, 2])
        if H < k and st == 2:
            st = 1
        out.append((C, H, W, Ko, k, st))
    return out


def _rows(rng):
    return rng.choice([1, 1, 2, 3, 4, 5, 7, 8, 16, 33, 64, 100, 128, 192, 384, 1000, 4096])


def _worker(seed, n_iter, q):
    import ctgan_amd.kernels as K
    from ctgan_amd._lib import WgradGroup, lib
    rng = random.Random(seed)
    done = 0
    seen = []
    try:
        for it in range(n_iter):
            ng = rng.randint(1, 12)
            arr = (WgradGroup * ng)()
            geoms = _geom_cases(rng, ng)
            for G, (C, H, W, Ko, k, st) in zip(arr, geoms):
                g = K.ConvGeom(C, H, W, Ko, k, k, st, False)
                nseg = rng.randint(1, 3)
                Ns = [_rows(rng) for _ in range(nseg)]
                G.d = g.desc(Ns[0], (C * H * W, 1, W * C, C), (Ko * g.P * g.Q, 1, g.Q * Ko, Ko))
                G.nseg = nseg
                for i, n in enumerate(Ns):
                    G.Ns[i] = n
                    G.seg_flags[i] = rng.choice([0, 2, 4, 6])
                    G.xs[i] = 0x10000
                    G.dys[i] = 0x20000
                G.dw = 0x30000
                G.db = 0x40000 if any(G.seg_flags[i] & 4 for i in range(nseg)) and Ko % 4 == 0 else None
            q.put(('start', seed, it, [(tuple(gm), [arr[i].Ns[j] for j in range(arr[i].nseg)]) for i, gm in enumerate(geoms)]))
            # fp32 family: grouped query, per-problem tile, multi-segment query
            a = lib.ctgan_conv2d_wgrad_group_workspace_bytes(arr, ng)
            b = lib.ctgan_conv2d_wgrad_group_workspace_bytes(arr, ng)
            assert a == b, 'non-deterministic plan'
            for i in range(ng):
                lib.ctgan_conv2d_wgrad_group_tile(ctypes.byref(arr[i]))
                ns = (ctypes.c_int32 * arr[i].nseg)(*[arr[i].Ns[j] for j in range(arr[i].nseg)])
                lib.ctgan_conv2d_wgrad_multi_workspace_bytes(ctypes.byref(arr[i].d), arr[i].nseg, ns)
                for op in (0, 1, 2):
                    lib.ctgan_conv2d_workspace_bytes(ctypes.byref(arr[i].d), op)
                    for mma in (1, 2, 3):
                        lib.ctgan_conv2d16_supported(ctypes.byref(arr[i].d), op, mma)
                    lib.ctgan_conv2d16_x3_prefers(ctypes.byref(arr[i].d), op)
                    lib.ctgan_conv2d16_workspace_bytes(ctypes.byref(arr[i].d), op)
                for mma in (1, 2, 3):
                    lib.ctgan_conv2d16_wgrad_workspace_bytes(ctypes.byref(arr[i].d), mma)
            # 16-bit family: members only (the query returns 0 when any problem is outside the family)
            for mma in (1, 2, 3):
                members = [i for i in range(ng) if lib.ctgan_conv2d16_wgrad_group_workspace_bytes(ctypes.byref(arr[i]), 1, mma) > 0]
                if not members:
                    continue
                sub = (WgradGroup * len(members))()
                for k2, i in enumerate(members):
                    sub[k2] = arr[i]
                tot = lib.ctgan_conv2d16_wgrad_group_workspace_bytes(sub, len(members), mma)
                assert tot > 0 and tot == lib.ctgan_conv2d16_wgrad_group_workspace_bytes(sub, len(members), mma)
                # every problem needs at least one slab per segment
                need = 0
                for i in members:
                    d = arr[i].d
                    need += arr[i].nseg * 4 * (d.R * d.S * d.C + (1 if arr[i].db else 0)) * d.K
                assert tot >= need, (tot, need)
                # plans are memoised per thread (eight tables): an earlier table asked again, between other tables, must get its own answer
                seen.append((sub, len(members), mma, tot))
                for sub0, n0, mma0, tot0 in rng.sample(seen, min(2, len(seen))):
                    assert lib.ctgan_conv2d16_wgrad_group_workspace_bytes(sub0, n0, mma0) == tot0, 'stale or colliding memoised plan'
                del seen[:-24]
            done += 1
        q.put(('ok', seed, done, None))
    except Exception as e:      # noqa: BLE001
        q.put(('error', seed, done, repr(e)))


@pytest.mark.timeout(300)
@pytest.mark.parametrize('seed', [1, 2, 3, 4])
def test_planners_terminate_and_are_consistent_on_random_job_tables(seed):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(seed, 150, q))
    p.start()
    # drain while waiting (a child cannot exit before its queued messages are flushed into the pipe)
    import queue as _q
    import time
    deadline = time.time() + 120
    last, status = None, None
    while status is None and time.time() < deadline:
        try:
            m = q.get(timeout=1.0)
        except _q.Empty:
            if not p.is_alive() and q.empty():
                break
            continue
        if m[0] == 'start':
            last = m
        else:
            status = m
    hung = status is None and p.is_alive()
    if p.is_alive():
        p.terminate()
    p.join()
    assert not hung, 'a planner did not return within the time limit; last job table: %r' % (last,)
    assert status is not None and status[0] == 'ok', (status, last)
