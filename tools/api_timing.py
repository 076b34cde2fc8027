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


for n in ("set_state", "get_state", "run_sampled_async", "fetch_samples", "set_temperature"):
    timed(n)

model = synth.build_cluster_model(synth.fcc_prim(), {2: 6.0, 3: 5.0})
sc = synth.build_supercell(model, [16, 16, 16])
ens = moca.Ensemble.from_cluster_expansion(sc, synth.random_coefs(model))
sampler = moca.Sampler.from_ensemble(ens, temperature=2500, nwalkers=4096, step_type="swap")
occ = (np.random.default_rng(0).random((4096, sc.num_sites)) < 0.5).astype(np.int32)
for n, thin in ((98304, 4096), (98304, 4096), (98304, 4096), (100_000, 10_000), (200_000, 200_000), (200_000, 200_000)):
    spent.clear()
    t = time.perf_counter()
    sampler.run(n, occ if len(sampler.samples) == 0 else None, thin_by=thin)
    dt = time.perf_counter() - t
    kern = sampler._get_engine().last_kernel_ms() * 1e-3  # (of the LAST launch: a run is several blocks)
    print(f"steps {n} thin_by {thin}: wall {dt:.3f} s, last launch {kern:.3f} s, "
          f"{2 * n * 4096 / dt:.3e} flips/s through the API; engine calls: "
          + ", ".join(f"{k} {v:.3f}" for k, v in spent.items()), flush=True)
t = time.perf_counter()
print("mean enthalpy", sampler.samples.mean_enthalpy(), "efficiency", sampler.efficiency(),
      f"({time.perf_counter() - t:.3f} s)")
t = time.perf_counter()
o = sampler.samples.get_occupancies(flat=False)
print("get_occupancies", o.shape, o.dtype, f"{time.perf_counter() - t:.3f} s")


# ---- biased and Wang-Landau kernels through the same API (ABI 7: their samples are recorded on the device too) ----
def kernel_rate(eng, nsteps, flips_per_step, launches=3):
    """steps/s of plain launches on the sampler's own handle (the kernel-only rate of this state)"""
    ms = []
    for _ in range(launches):
        eng.run(nsteps)
        ms.append(eng.last_kernel_ms())
    return eng.R * nsteps / (np.mean(ms) * 1e-3)


from smol_amd import workloads  # noqa: E402

# config 9: ternary rocksalt 12^3, triplet CE + Ewald + mu, semigrand flips under a SquareChargeBias
model9, sc9, ew9 = workloads._rocksalt(12)
ens9 = moca.Ensemble.from_cluster_expansion(sc9, synth.random_coefs(model9), ewald_term=ew9, ewald_coefficient=0.1)
mu9 = np.random.default_rng(7).uniform(-workloads.CONFIG3_MU, workloads.CONFIG3_MU, 3)
ens9.chemical_potentials = {sp: float(mu9[i]) for i, sp in enumerate(ens9.active_sublattices[0].species)}
R9 = 2048
s9 = moca.Sampler.from_ensemble(ens9, temperature=workloads.CONFIG9_T, nwalkers=R9, step_type="flip", seeds=list(range(R9)),
                                bias_type="square-charge", bias_kwargs={"penalty": workloads.CONFIG9_PENALTY})
occ9 = workloads.neutral_rocksalt_occupancy(sc9, 0, R9)
s9.run(20 * 3456, occ9, thin_by=3456)  # (warm-up: slots grow, the chain leaves its start)
for n, thin in ((40 * 3456, 3456), (40 * 3456, 3456)):
    spent.clear()
    t = time.perf_counter()
    s9.run(n, None, thin_by=thin)
    dt = time.perf_counter() - t
    print(f"config 9 through Sampler.run: steps {n} thin_by {thin}: wall {dt:.3f} s, {n * R9 / dt:.3e} steps/s; engine calls: "
          + ", ".join(f"{k} {v:.3f}" for k, v in spent.items()), flush=True)
k9 = kernel_rate(s9.engine, 3456 * 4, 1)
print(f"config 9 kernel-only on the same handle ({s9.engine.kernel_info()}): {k9:.3e} steps/s", flush=True)

# config 4: Wang-Landau (1024 walkers, 512 bins): every sample carries the walkers' entropy / histogram /
# occurrences [512] and mean features [512 x 8] -- 46 MB per sample
model4 = synth.build_cluster_model(synth.fcc_prim(), {2: 6.0, 3: 5.0})
sc4 = synth.build_supercell(model4, [16, 16, 16])
ens4 = moca.Ensemble.from_cluster_expansion(sc4, synth.random_coefs(model4, seed=20260928))
R4 = 1024
occ4 = workloads.balanced_binary(sc4, 0, R4, seed=4)
h0 = float(ens4.natural_parameters @ ens4.compute_feature_vector(occ4[0]))
s4 = moca.Sampler.from_ensemble(ens4, h0 - 160.37, h0 + 95.63, 0.5, kernel_type="Wang-Landau", step_type="swap", nwalkers=R4,
                                seeds=list(range(R4)), check_period=1000, flatness=0.8)
s4.run(4 * 20000, occ4, thin_by=20000)
s4.clear_samples()
for n, thin in ((8 * 20000, 20000), (8 * 20000, 20000)):
    spent.clear()
    t = time.perf_counter()
    s4.run(n, None if len(s4.samples) else s4.engine.get_state()["occupancy"], thin_by=thin)
    dt = time.perf_counter() - t
    print(f"config 4 through Sampler.run: steps {n} thin_by {thin}: wall {dt:.3f} s, {n * R4 / dt:.3e} steps/s; engine calls: "
          + ", ".join(f"{k} {v:.3f}" for k, v in spent.items()), flush=True)
    s4.clear_samples()
k4 = kernel_rate(s4.engine, 20000, 2)
print(f"config 4 kernel-only on the same handle ({s4.engine.kernel_info()}): {k4:.3e} steps/s", flush=True)
