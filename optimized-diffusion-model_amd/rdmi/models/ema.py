"""Exponential moving average of the trainable parameters.

Host mirror of Reflected-Diffusion/models/ema.py:10-99 (same constructor, update/copy_to/store/restore,
state_dict layout: decay, num_updates, shadow_params as an ORDERED LIST over parameters with
requires_grad, i.e. all 260 except time_embed.W).  The per-tensor loops are fused with
torch._foreach ops (one launch per call instead of 260) -- plain device-memory plumbing.
"""
import torch


class ExponentialMovingAverage:
    def __init__(self, parameters, decay, use_num_updates=True):
        if decay < 0.0 or decay > 1.0:
            raise ValueError('Decay must be between 0 and 1')
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        self.shadow_params = [p.clone().detach() for p in parameters if p.requires_grad]
        self.collected_params = []

    def next_decay(self):
        """Advance the update counter and return this update's decay min(decay, (1 + n) / (10 + n))   (RD/models/ema.py:43-46).
        Used by update() and by the fused optimizer step (rdmi.losses), which applies the shadow update in its own kernel."""
        decay = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            decay = min(decay, (1 + self.num_updates) / (10 + self.num_updates))
        return decay

    def update(self, parameters):
        """s -= (1 - d) * (s - p),  d = min(decay, (1 + n) / (10 + n))   (RD/models/ema.py:32-52)."""
        decay = self.next_decay()
        with torch.no_grad():
            params = [p.detach() for p in parameters if p.requires_grad]
            diff = torch._foreach_sub(self.shadow_params, params)
            torch._foreach_mul_(diff, 1.0 - decay)
            torch._foreach_sub_(self.shadow_params, diff)

    def copy_to(self, parameters):
        params = [p for p in parameters if p.requires_grad]
        with torch.no_grad():
            torch._foreach_copy_([p.data for p in params], [s.data for s in self.shadow_params])

    def store(self, parameters):
        self.collected_params = [p.clone() for p in parameters]

    def restore(self, parameters):
        with torch.no_grad():
            for c, p in zip(self.collected_params, parameters):
                p.data.copy_(c.data)

    def state_dict(self):
        return dict(decay=self.decay, num_updates=self.num_updates, shadow_params=self.shadow_params)

    def load_state_dict(self, state_dict):
        self.decay = state_dict['decay']
        self.num_updates = state_dict['num_updates']
        self.shadow_params = state_dict['shadow_params']
