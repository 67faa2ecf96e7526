"""Optimizer wrappers with the reference's names (transvae/opt.py) over one fused Adam launch."""
import torch

from . import ops


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (no weight decay, no amsgrad) in ONE multi-tensor HIP launch.

    Keeps the torch Optimizer bookkeeping (``param_groups``, ``state`` with ``step`` /
    ``exp_avg`` / ``exp_avg_sq`` per parameter) so ``state_dict()`` has the layout the
    reference stores in its checkpoints (trans_models.py:339, opt.py:17-19).
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._fused = None
        self._step = 0
        self.fuse_zero_grad = False
        self.grad_scale = 1.0          # folded into the launch (1/world_size under data parallelism)

    def _engine(self):
        if self._fused is None:
            if len(self.param_groups) != 1:
                raise ops._lib.TvaeError("fused Adam supports a single param group")
            g = self.param_groups[0]
            self._fused = ops.FusedAdam(g["params"], betas=g["betas"], eps=g["eps"])
            for p, m, v in zip(g["params"], self._fused.exp_avg, self._fused.exp_avg_sq):
                st = self.state[p]
                if "exp_avg" in st:                       # state restored by load_state_dict
                    m.copy_(st["exp_avg"])
                    v.copy_(st["exp_avg_sq"])
                    self._step = int(st.get("step", self._step))
                st["exp_avg"], st["exp_avg_sq"] = m, v
        return self._fused

    @torch.no_grad()
    def step(self, closure=None):
        eng = self._engine()
        self._step += 1
        g = self.param_groups[0]
        eng.set_hyper(g["lr"], self._step, self.grad_scale)
        eng.launch(zero_grad=self.fuse_zero_grad)
        for p in g["params"]:
            self.state[p]["step"] = self._step


class NoamOpt:
    """lr = factor * d_model^-0.5 * min(step^-0.5, step * warmup^-1.5) around an optimizer
    (reference transvae/opt.py:8-39, same attributes: ``state_dict`` is a dict ATTRIBUTE carrying
    the wrapped optimizer's state plus 'step' and 'rate')."""

    def __init__(self, model_size, factor, warmup, optimizer):
        self.optimizer = optimizer
        self.warmup = warmup
        self.factor = factor
        self.model_size = model_size
        self.state_dict = self.optimizer.state_dict()
        self.state_dict['step'] = 0
        self.state_dict['rate'] = 0

    def step(self):
        self.state_dict['step'] += 1
        rate = self.rate()
        for group in self.optimizer.param_groups:
            group['lr'] = rate
        self.state_dict['rate'] = rate
        self.optimizer.step()

    def sync_state_dict(self):
        """Mirror the wrapped optimizer's state into ``state_dict`` (the reference does this on
        every step, opt.py:29-30; here it is deferred to checkpoint time -- it copies nothing on
        the device, but building the dict per step is host work the hot loop does not need)."""
        for k, v in self.optimizer.state_dict().items():
            self.state_dict[k] = v

    def rate(self, step=None):
        if step is None:
            step = self.state_dict['step']
        return self.factor * (self.model_size ** (-0.5) * min(step ** (-0.5), step * self.warmup ** (-1.5)))

    def load_state_dict(self, state_dict):
        # reference quirk kept (opt.py:38-39): only the dict is replaced, so 'step' (hence the
        # learning rate) resumes while Adam's moments restart from zero.
        self.state_dict = state_dict


class AdamOpt:
    "Fixed-lr Adam wrapper (reference transvae/opt.py:41-52)"

    def __init__(self, params, lr, optimizer):
        self.optimizer = optimizer(params, lr)
        self.state_dict = self.optimizer.state_dict()

    def step(self):
        self.optimizer.step()
        self.state_dict = self.optimizer.state_dict()

    def sync_state_dict(self):
        "same call as NoamOpt's (VAEShell.train makes it before every checkpoint): refresh the dict attribute"
        self.state_dict = self.optimizer.state_dict()

    def load_state_dict(self, state_dict):
        self.state_dict = state_dict


def get_std_opt(model):
    return NoamOpt(model.src_embed[0].d_model, 2, 4000, Adam(model.parameters(), lr=0, betas=(0.9, 0.98), eps=1e-9))
