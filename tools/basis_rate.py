"""Basis-network forward (models/dngo.lua:155-171) by grid size and activation: what a tile costs, what the launch costs.
usage: basis_rate.py   (GPU box)"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bot7_amd  # noqa: E402
c = bot7_amd.Context(0)
d = 5
rng = np.random.default_rng(0)
dims = [d, 50, 50, 50]
W = [rng.normal(scale=1.0 / np.sqrt(dims[i]), size=(dims[i + 1], dims[i])) for i in range(3)]
b = [rng.normal(scale=0.1, size=dims[i + 1]) for i in range(3)]
for act in (None, "ReLU", "Tanh"):
    for M in (256, 4096, 32768, 65536, 131072, 262144, 1048576):
        c.grid_sobol(M, d, 2, download=False)
        c.blr_basis(W, b, act)
        c.profile_enable(True)
        ts = []
        for _ in range(7):
            c.profile_reset()
            c.blr_basis(W, b, act)
            ts.append(c.profile_get("basis")[0])
        c.profile_enable(False)
        ms = float(np.median(ts))
        print("act %-5s M %8d: %.1f us  (%.2f us per 16-row tile per wave-slot of 2048)" % (act, M, ms * 1e3, ms * 1e3 / max(1.0, M / 16 / 2048)), flush=True)
