"""PositionalEncoding: parameter/buffer-compatible with src/model/code.py:6-52.

The fused kernel evaluates the encoding in-register (pnr_mlp.hip `geometry`); this module keeps
the `_freqs` / `_phases` buffers for state_dict compatibility and a torch forward for callers
that use the module on its own."""
import numpy as np
import torch


class PositionalEncoding(torch.nn.Module):
    def __init__(self, num_freqs=6, d_in=3, freq_factor=np.pi, include_input=True):
        super().__init__()
        self.num_freqs = num_freqs
        self.d_in = d_in
        self.freq_factor = float(freq_factor)
        self.freqs = freq_factor * 2.0 ** torch.arange(0, num_freqs)
        self.d_out = self.num_freqs * 2 * d_in
        self.include_input = include_input
        if include_input:
            self.d_out += d_in
        self.register_buffer("_freqs", torch.repeat_interleave(self.freqs, 2).view(1, -1, 1))
        _phases = torch.zeros(2 * self.num_freqs)
        _phases[1::2] = np.pi * 0.5
        self.register_buffer("_phases", _phases.view(1, -1, 1))

    def forward(self, x):
        embed = x.unsqueeze(1).repeat(1, self.num_freqs * 2, 1)
        embed = torch.sin(torch.addcmul(self._phases, embed, self._freqs))
        embed = embed.view(x.shape[0], -1)
        if self.include_input:
            embed = torch.cat((x, embed), dim=-1)
        return embed

    @classmethod
    def from_conf(cls, conf, d_in=3):
        return cls(conf.get_int("num_freqs", 6), d_in, conf.get_float("freq_factor", np.pi),
                   conf.get_bool("include_input", True))
