"""bottle + the flat-buffer Adam that replaces torch.optim.Adam / clip_grad_norm_.

bottle: /root/reference/algorithms/repo/models/utils.py:9-16.
FlatAdam: the reference builds Adam(list_of_params, lr) (dreamer.py:96,106,114; repo.py:23) and
calls zero_grad / clip_grad_norm_ / step (repo.py:87-90).  Here all parameters of a group live
in ONE contiguous device buffer (each nn.Parameter is a view into it, as is its .grad), so the
global-norm clip and the Adam update are two streaming kernels over the flat buffers, and a
data-parallel all-reduce is one collective.  state_dict()/load_state_dict() keep
torch.optim.Adam's layout so reference checkpoints round-trip.
"""
import torch

from .... import ops


def bottle(f, xs):
    """Apply f to (time*batch, ...) views of (time, batch, ...) tensors and fold back."""
    horizon, batch_size = xs[0].shape[:2]
    ys = f(*(x.reshape(horizon * batch_size, *x.shape[2:]) for x in xs))
    if isinstance(ys, tuple):
        return tuple(y.reshape(horizon, batch_size, *y.shape[1:]) for y in ys)
    return ys.reshape(horizon, batch_size, *ys.shape[1:])


def adam_param_group(lr, betas, eps, n_params):
    """The `param_groups[0]` entry torch.optim.Adam of THIS torch build would write for these
    hyper-parameters: key set and key order come from a throw-away torch.optim.Adam instance, so a
    checkpoint written here has the layout the reference's `Adam.state_dict()` has on the same install
    (tests/golden/checkpoint_manifest.json pins it for the build container's torch)."""
    probe = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=float(lr), betas=tuple(betas), eps=float(eps))
    group = dict(probe.state_dict()["param_groups"][0])
    group["params"] = list(range(n_params))
    return group


class FlatAdam:
    """Adam over a flat parameter buffer with fused global-norm clipping."""

    @staticmethod
    def padded_numel(params):
        """Length of the flat buffer FlatAdam lays these parameters out in (16-byte aligned sub-views)."""
        return sum((p.numel() + 3) // 4 * 4 for p in params)

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8, grad=None):
        """grad: optional caller-owned storage for the flat gradient (padded_numel floats) -- the agents
        lay the actor's and the critic's gradients out in ONE buffer so that a data-parallel job exchanges
        them with one collective."""
        self.params = list(params)
        assert self.params, "FlatAdam needs at least one parameter"
        self.lr, self.betas, self.eps = float(lr), tuple(betas), float(eps)
        self.step_count = 0
        dev = self.params[0].device
        sizes = [p.numel() for p in self.params]
        # 16-byte aligned offsets so every view can be read with 128-bit loads
        self.offsets, off = [], 0
        for n in sizes:
            self.offsets.append(off)
            off += (n + 3) // 4 * 4
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        if grad is None:
            grad = torch.zeros(off, dtype=torch.float32, device=dev)
        assert grad.shape == (off,) and grad.dtype == torch.float32 and grad.is_contiguous(), (grad.shape, off)
        self.grad = grad
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        self.sqnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.skip = None   # the current update's status word (ops.take_scan_status): non-zero = the step is skipped
        self.gviews = []
        with torch.no_grad():
            for p, o, n in zip(self.params, self.offsets, sizes):
                self.flat[o : o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat[o : o + n].view(p.shape)
                gv = self.grad[o : o + n].view(p.shape)
                p.grad = gv
                self.gviews.append(gv)

    @classmethod
    def view(cls, parent, n_params, lr):
        """A second optimiser over the FIRST n_params parameters of `parent` (same storage for values and gradients,
        its own moments and step count): the reference's `Adam(self.encoder.parameters())` beside the model optimiser
        (repo_adapt.py:29) -- two torch optimisers over shared parameters."""
        self = cls.__new__(cls)
        self.params = parent.params[:n_params]
        self.lr, self.betas, self.eps = float(lr), parent.betas, parent.eps
        self.step_count = 0
        self.offsets = parent.offsets[:n_params]
        self.numel = parent.offsets[n_params] if n_params < len(parent.params) else parent.numel
        self.flat, self.grad = parent.flat[: self.numel], parent.grad[: self.numel]
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.sqnorm = torch.zeros(1, dtype=torch.float32, device=self.flat.device)
        self.skip = None
        self.gviews = parent.gviews[:n_params]
        return self

    # -- reference-style surface -----------------------------------------------------
    def zero_grad(self, set_to_none=False):
        self.grad.zero_()

    def clip_and_step(self, max_norm):
        """clip_grad_norm_(params, max_norm) + step() (repo.py:89-90).  The squared global
        norm stays on the device in self.sqnorm (read it after the update's single sync)."""
        self.step_count += 1
        ops.grad_sqnorm(self.grad, out=self.sqnorm)
        ops.clip_adam(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.sqnorm, max_norm, self.lr,
                      self.step_count, self.betas, self.eps, skip=self.skip)

    def step(self):
        self.step_count += 1
        ops.clip_adam(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, None, 0.0, self.lr, self.step_count,
                      self.betas, self.eps, skip=self.skip)

    # -- torch.optim.Adam-compatible checkpoint layout -------------------------------------
    def state_dict(self):
        state = {}
        if self.step_count > 0:
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                n = p.numel()
                state[i] = {
                    "step": torch.tensor(float(self.step_count)),
                    "exp_avg": self.exp_avg[o : o + n].view(p.shape).clone(),
                    "exp_avg_sq": self.exp_avg_sq[o : o + n].view(p.shape).clone(),
                }
        return {"state": state, "param_groups": [adam_param_group(self.lr, self.betas, self.eps, len(self.params))]}

    def load_state_dict(self, sd):
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps = float(g["lr"]), tuple(g["betas"]), float(g["eps"])
        steps = set()
        with torch.no_grad():
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                st = sd["state"].get(i)
                if st is None:
                    continue
                n = p.numel()
                self.exp_avg[o : o + n].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[o : o + n].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(float(st["step"])))
        assert len(steps) <= 1, "per-parameter step counts differ; not an Adam state this optimiser can hold"
        self.step_count = steps.pop() if steps else 0
