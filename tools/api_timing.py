"""Wall time of the smol-shaped API (Sampler.run) on the headline model next to the kernel time:
where the host side spends it (uploads, device ring download, container bookkeeping).

  python tools/api_timing.py
"""
import functools
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smol_amd import moca, synth  # noqa: E402
from smol_amd.engine import Engine  # noqa: E402

spent = {}


def timed(name):
    f = getattr(Engine, name)

    @functools.wraps(f)
    def g(self, *a, **k):
        t = time.perf_counter()
        out = f(self, *a, **k)
        spent[name] = spent.get(name, 0.0) + time.perf_counter() - t
        return out

    setattr(Engine, name, g)


for n in ("set_state", "get_state", "run_sampled", "set_temperature"):
    timed(n)

model = synth.build_cluster_model(synth.fcc_prim(), {2: 6.0, 3: 5.0})
sc = synth.build_supercell(model, [16, 16, 16])
ens = moca.Ensemble.from_cluster_expansion(sc, synth.random_coefs(model))
sampler = moca.Sampler.from_ensemble(ens, temperature=2500, nwalkers=4096, step_type="swap")
occ = (np.random.default_rng(0).random((4096, sc.num_sites)) < 0.5).astype(np.int32)
for n, thin in ((98304, 4096), (100_000, 10_000), (200_000, 200_000), (200_000, 200_000)):
    spent.clear()
    t = time.perf_counter()
    sampler.run(n, occ if len(sampler.samples) == 0 else None, thin_by=thin)
    dt = time.perf_counter() - t
    kern = sampler._get_engine().last_kernel_ms() * 1e-3
    print(f"steps {n} thin_by {thin}: wall {dt:.3f} s, kernel {kern:.3f} s, "
          f"{2 * n * 4096 / dt:.3e} flips/s through the API; engine calls: "
          + ", ".join(f"{k} {v:.3f}" for k, v in spent.items()), flush=True)
t = time.perf_counter()
print("mean enthalpy", sampler.samples.mean_enthalpy(), "efficiency", sampler.efficiency(),
      f"({time.perf_counter() - t:.3f} s)")
t = time.perf_counter()
o = sampler.samples.get_occupancies(flat=False)
print("get_occupancies", o.shape, o.dtype, f"{time.perf_counter() - t:.3f} s")
