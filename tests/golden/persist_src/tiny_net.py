"""A persistent module written against the torch_utils API only (ops + persistence), the way the
reference's model files are. tests/golden/make_golden_persistence.py pickles an instance of it with
the REFERENCE torch_utils; the pickle embeds this source (persistence protocol v6)."""

import numpy as np
import torch

from torch_utils import persistence
from torch_utils.ops import bias_act, upfirdn2d


@persistence.persistent_class
class TinyUpsampler(torch.nn.Module):
    def __init__(self, channels, up=2, slope=0.3):
        super().__init__()
        self.up, self.slope = up, slope
        self.weight = torch.nn.Parameter(torch.linspace(-1, 1, channels * channels).reshape(channels, channels, 1, 1))
        self.bias = torch.nn.Parameter(torch.linspace(0.5, -0.5, channels))
        self.register_buffer('taps', upfirdn2d.setup_filter([1, 3, 3, 1]))

    def forward(self, x):
        x = torch.nn.functional.conv2d(x, self.weight)
        x = upfirdn2d.upsample2d(x, self.taps, up=self.up)
        return bias_act.bias_act(x, self.bias, act='lrelu', alpha=self.slope, gain=float(np.sqrt(2)), clamp=4.0)
