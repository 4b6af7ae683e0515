"""TF-form Adam over one flat fp32 buffer per network (csrc/optim_rng.hip: adam_kernel).

Mirrors `tf.train.AdamOptimizer(learning_rate=LR*decay, beta1, beta2)` + compute_gradients /
apply_gradients (TF/CT_gan_cifar_resnet.py:333-338; TF/CT_gan_cifar.py:153-154).  The flat
parameter / gradient buffers double as the all-reduce buckets of the batch-sharded step (ddp.py).
"""
import torch

from . import kernels as K
from . import tflib as lib


class FlatAdam:
    def __init__(self, named_params, beta1, beta2, eps=1e-8, state=None):
        """named_params: [(name, Parameter)] - the trainable variable list, in registry order.  state: optional float[4] device
        slice to keep {lr, beta1^t, beta2^t, -} in (two optimizers sharing one allocation take a common learning rate in one fill)."""
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        if not self.params:
            raise ValueError('empty parameter list')
        heads = {n.split('.')[0] for n in self.names}
        self.group = heads.pop() if len(heads) == 1 else None     # one network per optimizer: only its caches go stale on step()
        dev = self.params[0].device
        self.beta1, self.beta2, self.eps = float(beta1), float(beta2), float(eps)
        self.sizes = [p.numel() for p in self.params]
        total = sum(self.sizes)
        self.theta = torch.empty(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.v = torch.zeros(total, dtype=torch.float32, device=dev)
        # re-home every parameter inside the flat buffer (same values, same shapes)
        off = 0
        with torch.no_grad():
            for p, n in zip(self.params, self.sizes):
                self.theta[off:off + n].copy_(p.reshape(-1))
                p.data = self.theta[off:off + n].view(p.shape)
                off += n
        # {lr, beta1^t, beta2^t, skipped}; TF initialises the power accumulators to beta (t = 1).  skipped: elements the update
        # kernels left untouched because their gradient was not finite, accumulated over the run (skipped())
        init = torch.tensor([0.0, self.beta1, self.beta2, 0.0], dtype=torch.float32, device=dev)
        if state is None:
            self.state = init
        else:
            assert state.shape == (4,) and state.dtype == torch.float32 and state.device.type == dev.type and state.is_contiguous()
            self.state = state
            self.state.copy_(init)
        self._lr_last = None      # value currently in state[0] if written by set_lr (None = unknown)
        self.t = 0
        self._keepalive = []
        self.offsets = [sum(self.sizes[:i]) for i in range(len(self.sizes))]

    def set_lr(self, lr):
        """Write the scalar learning rate into the device-resident state (outside any captured graph).  The value rides in
        the fill kernel's ARGUMENTS - no host staging buffer a later call could overwrite before an asynchronous copy of it
        has run - and the launch is skipped while the rate does not change (the N_CRITIC critic steps of an iteration)."""
        lr = float(lr)
        if self._lr_last == lr:
            return
        self.state[0:1].fill_(lr)
        self._lr_last = lr

    def skipped(self):
        """Number of parameter elements the update kernels have left untouched so far because their (scaled) gradient was NaN / inf
        (csrc/optim_rng.hip adam_elem; the reference's tf.train.AdamOptimizer would propagate the NaN instead).  Non-zero means the
        run overflowed (fp16 mode under its fixed loss scale) or diverged: the trainers log it, bench.py fails its loss guard on it.
        Reads device memory: call it outside captured regions, at logging cadence."""
        return int(self.state[3].item())

    def gather_grads(self, grads, lo=0, hi=None):
        """Pack per-parameter gradients (None = zero) into the flat bucket with ONE kernel (the pointer table rides
        in the kernel arguments: no per-variable copy, hipGraph-capture safe).  lo / hi: only parameters [lo, hi) - `grads` still
        lists all of them - for a step that hands its bucket to the all-reduce in two parts; returns that part of the bucket."""
        hi = len(self.sizes) if hi is None else hi
        end = self.offsets[hi] if hi < len(self.sizes) else self.grad.numel()
        part = self.grad[self.offsets[lo]:end] if lo < hi else self.grad[0:0]
        if lo >= hi:
            return part
        if self.grad.device.type != 'cuda':
            for i in range(lo, hi):
                g, off, n = grads[i], self.offsets[i], self.sizes[i]
                if g is None:
                    self.grad[off:off + n].zero_()
                else:
                    self.grad[off:off + n].copy_(g.reshape(-1))
            return part
        srcs = [g.contiguous() if g is not None else None for g in grads[lo:hi]]
        K.pack(srcs, self.offsets[lo:hi], self.sizes[lo:hi], self.grad)
        self._keepalive = (self._keepalive if lo else []) + srcs          # sources stay allocated until the next step's first gather
        return part

    def step(self, grad_scale=1.0, rng=None):
        """theta <- Adam(theta, grad); then ONE launch advances the beta-power accumulators and, with `rng`, that stream's step
        counter (DeviceRNG.end_step)."""
        K.adam_step(self.theta, self.grad, self.m, self.v, self.state, self.beta1, self.beta2, self.eps, grad_scale)
        self._end(rng)

    def _end(self, rng):
        K.step_advance(self.state, self.beta1, self.beta2, rng.ctr if rng is not None else None, 1)
        self.t += 1
        lib.bump_epoch(self.group)    # this network's weights changed: its derived-filter caches are stale

    def update(self, grads, grad_scale=1.0, rng=None):
        """gather_grads + step with the bucket and the update in ONE launch: the single-rank form of a step (no collective between
        gather and update)."""
        if self.grad.device.type != 'cuda' or len(grads) > K.ADAM_PACKED_MAX:
            self.gather_grads(grads)
            return self.step(grad_scale, rng)
        srcs = [g.contiguous() if g is not None else None for g in grads]
        K.adam_step_packed(srcs, self.offsets, self.sizes, self.grad, self.theta, self.m, self.v, self.state, self.beta1, self.beta2,
                           self.eps, grad_scale)
        self._keepalive = srcs
        self._end(rng)

    def load_named_slots(self, m_by_name, v_by_name, t):
        """Overwrite the Adam slots from per-parameter tensors (teacher-forced parity tests, resume)."""
        off = 0
        for name, n in zip(self.names, self.sizes):
            self.m[off:off + n].copy_(m_by_name[name].reshape(-1).to(self.m.device, torch.float32))
            self.v[off:off + n].copy_(v_by_name[name].reshape(-1).to(self.v.device, torch.float32))
            off += n
        self.t = int(t)
        self.state[1] = self.beta1 ** (self.t + 1)
        self.state[2] = self.beta2 ** (self.t + 1)

    def state_dict(self):
        return {'m': self.m.cpu().clone(), 'v': self.v.cpu().clone(), 'state': self.state.cpu().clone(), 't': self.t}

    def load_state_dict(self, sd):
        self.m.copy_(sd['m']); self.v.copy_(sd['v']); self.state.copy_(sd['state']); self.t = int(sd['t'])
        self.state[3:4].zero_()            # skipped() counts THIS session's skips: a resumed run does not inherit the saved run's
        self._lr_last = None
