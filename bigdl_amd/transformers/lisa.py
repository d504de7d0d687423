"""LISA (layer-wise importance sampled AdamW): all decoder layers are frozen except ``n_layers`` drawn at random
every ``interval_steps`` optimizer steps (embeddings and the head stay trainable).

Reference: python/llm/src/ipex_llm/transformers/lisa.py:23-73 ``DynamicLayerActivationCallback``.
"""
from __future__ import annotations

import numpy as np
from transformers import TrainerCallback


class DynamicLayerActivationCallback(TrainerCallback):
    def __init__(self, n_layers, interval_steps, model, seed=None):
        super().__init__()
        self.n_layers = n_layers
        self.interval_steps = interval_steps
        self.model = model
        self.rng = np.random.default_rng(seed)
        base = model
        for attr in ("base_model", "model"):
            while hasattr(base, attr) and getattr(base, attr) is not base and not hasattr(base, "layers"):
                base = getattr(base, attr)
        self.layers = base.layers if hasattr(base, "layers") else base.model.layers
        self.total_layers = len(self.layers)
        self.active_layers_indices = []

    def freeze_all_layers(self):
        for layer in self.layers:
            for p in layer.parameters():
                if p.dtype.is_floating_point:
                    p.requires_grad = False

    def on_step_begin(self, args, state, control, **kwargs):
        if state.global_step % self.interval_steps == 0:
            self.switch_active_layers()

    def switch_active_layers(self):
        self.freeze_all_layers()
        self.active_layers_indices = sorted(self.rng.choice(range(self.total_layers), self.n_layers, replace=False).tolist())
        for idx in self.active_layers_indices:
            for p in self.layers[idx].parameters():
                if p.dtype.is_floating_point:       # quantized byte buffers cannot take gradients
                    p.requires_grad = True
