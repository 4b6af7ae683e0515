"""Host logic of the deferred weight-gradient flush (functional._flush_groups), on the CPU stand-ins: a grouped call that raises
NotImplementedError must not make the fallback re-run groups an EARLIER grouped call already finished - with an in-place addend
(`inplace=True`: dw already holds the result of a use launched at request time) that would add the queued segments twice."""
import torch

import ctgan_amd.functional as F
import ctgan_amd.kernels as K


def _group(g, gen, n_rows, inplace):
    grp = F._new_group(g, 'cpu')
    x = torch.randn(n_rows, g.C, g.H, g.W, generator=gen)
    gy = torch.randn(n_rows, g.K, g.P, g.Q, generator=gen)
    grp.segs.append((x, gy, False, False))
    want = K.conv_wgrad(x, gy, g)
    if inplace:
        add = torch.randn(grp.dw.shape, generator=gen)
        grp.dw.copy_(add)
        grp.inplace = True
        want = want + add
    return grp, want


def test_a_failing_grouped_call_does_not_rerun_finished_groups(cpu_kernels, monkeypatch):
    gen = torch.Generator().manual_seed(3)
    g = K.ConvGeom(32, 8, 8, 32, 3, 3, 1, False)
    made = [_group(g, gen, 2, inplace=(i % 2 == 0)) for i in range(5)]
    grps = [m[0] for m in made]
    real = K.conv_wgrad_group
    calls = []

    def flaky(groups):
        calls.append(len(groups))
        if len(calls) == 2:
            raise NotImplementedError('second part: unsupported')      # (validated before any launch: nothing was written)
        return real(groups)
    monkeypatch.setattr(K, 'conv_wgrad_group', flaky)
    monkeypatch.setattr(K, 'WGRAD_GROUP_LIMIT', 2)
    F._flush_groups(grps)
    assert calls == [2, 2]                        # the third part is not attempted after the failure; parts 2 and 3 fall back per group
    for grp, want in made:
        assert torch.allclose(grp.dw, want, rtol=1e-5, atol=1e-5), (grp.inplace, (grp.dw - want).abs().max().item())


def test_grouped_flush_equals_per_group_flush(cpu_kernels, monkeypatch):
    gen = torch.Generator().manual_seed(4)
    g = K.ConvGeom(32, 8, 8, 32, 3, 3, 1, False)
    made = [_group(g, gen, 3, inplace=(i == 1)) for i in range(3)]
    F._flush_groups([m[0] for m in made])
    for grp, want in made:
        assert torch.allclose(grp.dw, want, rtol=1e-5, atol=1e-5)
