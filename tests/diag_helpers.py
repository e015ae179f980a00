"""Numpy stand-in for the GPU moment kernels, used only to drive the host-side / distributed
combination logic of fugue_amd.diagnostics in CPU tests."""
import numpy as np


class NumpyMoments:
    def __init__(self, draws: np.ndarray):          # [n][d][C]
        self.x = np.asarray(draws, dtype=np.float64)
        self.n, self.d, self.C = self.x.shape

    def moments(self) -> np.ndarray:
        n, half = self.n, self.n // 2
        out = np.zeros((self.d, 6, self.C))
        for k, (a, b) in enumerate(((0, n), (0, half), (half, 2 * half))):
            seg = self.x[a:b]
            mean = seg.sum(axis=0) / max(1, b - a)
            out[:, 2 * k] = mean
            out[:, 2 * k + 1] = ((seg - mean) ** 2).sum(axis=0)
        return out

    def autocov_sums(self, lag0: int, n_lags: int) -> np.ndarray:
        c = self.x - self.x.mean(axis=0, keepdims=True)
        out = np.zeros((self.d, n_lags))
        for k in range(n_lags):
            lag = lag0 + k
            if lag < self.n:
                out[:, k] = ((c[:self.n - lag] * c[lag:]).sum(axis=0) / self.n).sum(axis=1)
        return out


def ar1(rng, n, m, phi):
    x = np.zeros((n, m))
    x[0] = rng.standard_normal(m) / np.sqrt(1 - phi ** 2)
    for t in range(1, n):
        x[t] = phi * x[t - 1] + rng.standard_normal(m)
    return x
