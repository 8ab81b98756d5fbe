"""NeRF positional encoding (reference code/model/embedder.py:5-37,71-88)."""
import torch


class Embedder:
    """[x, sin(2^k x), cos(2^k x)] for k < num_freqs (log-sampled bands, input included)."""

    def __init__(self, input_dims, num_freqs):
        self.input_dims = input_dims
        self.num_freqs = num_freqs
        self.out_dim = input_dims * (1 + 2 * num_freqs)
        self.freq_bands = 2.0 ** torch.linspace(0.0, num_freqs - 1, num_freqs)

    def embed(self, x):
        parts = [x]
        for f in self.freq_bands.tolist():
            parts.append(torch.sin(x * f))
            parts.append(torch.cos(x * f))
        return torch.cat(parts, -1)


def get_embedder(multires, input_dims=3, embed_type="nerf"):
    if embed_type != "nerf":
        raise NotImplementedError("only the 'nerf' embedding used by the shipped configs is implemented")
    eo = Embedder(input_dims, multires)
    return eo.embed, eo.out_dim
