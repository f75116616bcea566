"""Adam for the CVAE trainers (train_s1.py:229 / train_s2.py:295-296: ``optim.Adam(model_h.parameters(), lr=...)``, one step per batch) as ONE
hand-written multi-tensor pass (csrc/adam.hip: ``psi_adam_step``).

``Adam`` IS a ``torch.optim.Adam``: same constructor defaults, same ``state`` / ``state_dict()`` layout (``step`` / ``exp_avg`` / ``exp_avg_sq`` per
parameter — checkpoints of either load into the other), same update (the operation order of PyTorch's fused implementation).  Only
``step()`` differs: instead of ``_foreach_add`` on the step counters plus four ``multi_tensor_apply`` launches (HumanCVAES2: 122 tensors,
15.7 M parameters, 440 MB of traffic) it is one or two launches that take the tensors' addresses in their kernel arguments.  The parameters
of a group are always stepped together, so their ``step`` entries are ONE shared device scalar (every entry of ``state`` refers to it).

The HIP kernel is the only implementation of the GPU path; parameter groups it does not cover (CPU tensors in the CPU tests of the training
loop, amsgrad / maximize / tensor learning rates, parameters that are neither contiguous nor channels_last) go through torch.optim.Adam's
own update on per-parameter step counters.

Checkpoints travel both ways: ``state_dict()`` hands out one ``step`` tensor PER parameter (clones of the shared counter: torch.optim.Adam
increments every entry it is given, so aliased entries would be counted once per parameter), ``load_state_dict()`` re-unifies the counters
of a loaded state eagerly (a later graph capture must not meet separate tensors), and moments saved in another memory layout than the
parameter has now (contiguous moments of a convolution weight the model has since switched to channels_last) are re-laid out at the next step."""
from __future__ import annotations

import ctypes

import torch

from . import hip


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, fused=True):
        params = list(params)
        on_gpu = all(p.is_cuda for g in (params if params and isinstance(params[0], dict) else [{'params': params}]) for p in g['params'])
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, fused=bool(fused) and on_gpu)
        self._ticket = {}
        self.hip_steps = 0                      # steps taken on the hand-written kernel (the tests assert the path that ran)

    # ------------------------------------------------------------------------------------------
    def _hip_ok(self, group):
        if not group.get('fused'):
            return False
        if group.get('amsgrad') or group.get('maximize') or group.get('differentiable'):
            return False
        if torch.is_tensor(group['lr']):
            return False
        return all(p.is_cuda and p.dtype == torch.float32 and not p.is_sparse for p in group['params'])

    @staticmethod
    def _dense_like(p, t):
        """t laid out exactly like p (same strides): the kernel walks the four tensors' memory side by side."""
        if t.stride() == p.stride():
            return t
        out = torch.empty_like(p)
        out.copy_(t)
        return out

    @staticmethod
    def _dense(params):
        return all(p.is_contiguous() or p.is_contiguous(memory_format=torch.channels_last) for p in params)

    def unify_steps(self):
        """Eagerly make the step counters of every group the HIP kernel covers ONE device scalar (what ``step()`` does lazily): call it
        before capturing ``step()`` into a graph when the state came from elsewhere (``load_state_dict`` does)."""
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            return
        for group in self.param_groups:
            params = [p for p in group['params'] if p in self.state and torch.is_tensor(self.state[p].get('step'))]
            if params and self._hip_ok(group) and self._dense(params):
                self._shared_step(group, params, [self.state[p]['step'] for p in params])

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self.unify_steps()

    def state_dict(self):
        sd = super().state_dict()
        # one step tensor per parameter (the entries of `state` alias ONE counter; a consumer that increments each entry must not see that)
        sd['state'] = {k: (dict(v, step=v['step'].clone()) if torch.is_tensor(v.get('step')) else v) for k, v in sd['state'].items()}
        return sd

    def _shared_step(self, group, params, steps):
        """One device scalar for the whole group; parameters that were stepped a different number of times cannot share one -> None."""
        first = steps[0]
        if all(s.data_ptr() == first.data_ptr() for s in steps):
            return first
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('psi Adam: the step counters must be unified before a graph capture (take one eager step first)')
        vals = torch.stack([s.detach().reshape(()).float() for s in steps])
        if float(vals.min()) != float(vals.max()):
            return None
        shared = first.detach().reshape(()).float().clone()
        for p in params:
            self.state[p]['step'] = shared
        return shared

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps = [], [], [], [], [], []
            self._init_group(group, params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps)
            if not params:
                continue
            # (layout checks BEFORE the counters are unified: a group that takes torch's update keeps per-parameter counters)
            shared = self._shared_step(group, params, steps) if (self._hip_ok(group) and self._dense(params)) else None
            if shared is None:
                self._torch_group_step(group, params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps)
                continue
            n = len(params)
            gs = [self._dense_like(p, g) for p, g in zip(params, grads)]
            for i, p in enumerate(params):
                # moments saved in another layout than the parameter has now (a checkpoint written before the model's convolution weights went
                # channels_last, a torch.optim.Adam state): re-laid out once, in place of the loaded tensors
                for key, lst in (('exp_avg', exp_avgs), ('exp_avg_sq', exp_avg_sqs)):
                    if lst[i].stride() != p.stride():
                        lst[i] = self.state[p][key] = self._dense_like(p, lst[i])
            arr = ctypes.c_void_p * n
            dev = params[0].device
            tk = self._ticket.get(dev)
            if tk is None:
                tk = self._ticket[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
            b1, b2 = group['betas']
            with torch.cuda.device(dev):
                hip.check(hip.lib().psi_adam_step(arr(*[p.data_ptr() for p in params]), arr(*[g.data_ptr() for g in gs]),
                                                  arr(*[m.data_ptr() for m in exp_avgs]), arr(*[v.data_ptr() for v in exp_avg_sqs]),
                                                  (ctypes.c_long * n)(*[p.numel() for p in params]), n, shared.data_ptr(), tk.data_ptr(),
                                                  float(group['lr']), float(b1), float(b2), float(group['eps']), float(group['weight_decay']),
                                                  hip.stream()), 'psi_adam_step')
            torch.autograd.graph.increment_version(params)      # the kernel wrote the parameters behind autograd's back: saved-tensor checks stay valid
            self.hip_steps += 1
        return loss

    def _torch_group_step(self, group, params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps):
        from torch.optim.adam import adam
        if len({s.data_ptr() for s in steps if torch.is_tensor(s)}) < len(steps):
            # counters that alias one tensor (a state this class stepped before): torch adds 1 to every entry it is given
            steps = [s.clone() for s in steps]
            for p, s in zip(params, steps):
                self.state[p]['step'] = s
        b1, b2 = group['betas']
        adam(params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps, amsgrad=group['amsgrad'], has_complex=False, beta1=b1, beta2=b2,
             lr=group['lr'], weight_decay=group['weight_decay'], eps=group['eps'], maximize=group['maximize'], foreach=group['foreach'],
             capturable=group['capturable'], differentiable=group['differentiable'], fused=group['fused'],
             grad_scale=getattr(self, 'grad_scale', None), found_inf=getattr(self, 'found_inf', None),
             decoupled_weight_decay=group.get('decoupled_weight_decay', False))
