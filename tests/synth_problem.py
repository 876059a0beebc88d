"""A LEARNABLE synthetic classification problem at the benchmark's class count (no dataset ships): class = smooth random
spatio-temporal template, clip = template + noise.  Generated on the CPU from seeded generators, so the fixture generator
(tools/gen_golden.py g16, which runs the REFERENCE's evaluate_synset on it) and the GPU tests build the same tensors on any
machine; the fixture stores checksums of them."""
import torch


def template_problem(C: int = 50, T: int = 8, S: int = 64, n_test: int = 4, noise_train: float = 1.0, noise_test: float = 3.0,
                     seed: int = 1606):
    """-> (train clips (C,T,3,S,S) -- one per class, the 'synthetic set' of an IPC=1 evaluation --, train labels,
    test clips (C*n_test,T,3,S,S), test labels)."""
    g = torch.Generator().manual_seed(int(seed))
    t = torch.randn(C, T, 3, S, S, generator=g)
    tmpl = torch.nn.functional.avg_pool2d(t.view(-1, 1, S, S), 9, 1, 4).view(C, T, 3, S, S) * 6         # smooth patterns, rms ~0.67
    train = tmpl + noise_train * torch.randn(tmpl.shape, generator=g)
    test = tmpl.repeat_interleave(n_test, 0) + noise_test * torch.randn((C * n_test, T, 3, S, S), generator=g)
    return train, torch.arange(C), test, torch.arange(C).repeat_interleave(n_test)


def checksum(x: torch.Tensor):
    x = x.double()
    return [float(x.sum()), float(x.abs().sum()), float((x * torch.arange(x.numel(), dtype=torch.float64).view(x.shape) % 7.0).sum())]
