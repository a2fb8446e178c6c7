"""Positional code of the network inputs (the module the reference keeps in src/model/code.py).

What is fixed by the reference is the checkpoint contract -- two buffers, `_freqs` and `_phases`, shaped
(1, 2*num_freqs, 1), every octave twice with phases 0 and pi/2 (src/model/code.py:17-28: cos(t) is taken as
sin(t + pi/2)) -- and the output order `[x, sin(f0 x), cos(f0 x), sin(f1 x), ...]` with the input dimension
innermost (code.py:30-42).  Everything else here is this package's own:

* inside the renderer the code of a point never exists in memory: the fused kernels evaluate it in registers
  (csrc/pnr_device.h, `geometry`) from the SAME constants;
* a stand-alone call of the module runs `pnr_positional_encoding` (csrc/pnr_encode.hip) on the HIP device, with
  `pnr_positional_encoding_backward` behind it for autograd.  There is no CPU path.
"""
import math

import torch

from .. import ops


class _Encode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, freqs2, phases2, include_input):
        ctx.save_for_backward(x, freqs2, phases2)
        ctx.include_input = include_input
        return ops.positional_encoding(x, freqs2, phases2, include_input)

    @staticmethod
    def backward(ctx, g):
        x, freqs2, phases2 = ctx.saved_tensors
        return ops.positional_encoding_backward(x, g.contiguous(), freqs2, phases2, ctx.include_input), None, None, None


class PositionalEncoding(torch.nn.Module):
    def __init__(self, num_freqs=6, d_in=3, freq_factor=math.pi, include_input=True):
        super().__init__()
        self.num_freqs = int(num_freqs)
        self.d_in = int(d_in)
        self.freq_factor = float(freq_factor)
        self.include_input = bool(include_input)
        self.d_out = self.d_in * (2 * self.num_freqs + (1 if self.include_input else 0))
        octaves = self.freq_factor * torch.pow(2.0, torch.arange(self.num_freqs, dtype=torch.float32))
        self.freqs = octaves
        quarter_turn = torch.tensor([0.0, 0.5 * math.pi], dtype=torch.float32)
        self.register_buffer("_freqs", octaves.repeat_interleave(2).reshape(1, 2 * self.num_freqs, 1))
        self.register_buffer("_phases", quarter_turn.repeat(self.num_freqs).reshape(1, 2 * self.num_freqs, 1))

    def forward(self, x):
        """x (batch, d_in) float32 on the HIP device -> (batch, d_out)."""
        if x.dim() != 2 or x.shape[1] != self.d_in:
            raise ValueError(f"PositionalEncoding: expected (batch, {self.d_in}), got {tuple(x.shape)}")
        return _Encode.apply(x, self._freqs, self._phases, self.include_input)

    @classmethod
    def from_conf(cls, conf, d_in=3):
        return cls(num_freqs=conf.get_int("num_freqs", 6), d_in=d_in, freq_factor=conf.get_float("freq_factor", math.pi),
                   include_input=conf.get_bool("include_input", True))
