"""Synthetic cluster-expansion model generator (host-side setup, NumPy only).

This module produces, without pymatgen, the read-only tables that the hot
path consumes, with the same *semantics* as the reference's cluster-subspace
machinery.  It is the build's own restatement of SURVEY.md §8(a) rows T1-T3:

* site bases (sinusoid / indicator, optional QR orthonormalisation)
      -> smol/cofe/space/basis.py:207-258, :448-459, :573-586
* orbits from diameter cutoffs, their ordering and ids
      -> smol/cofe/space/clusterspace.py:1297-1310, :1369-1565
* bit combos and correlation tensors, flat tensors and stride vectors
      -> smol/cofe/space/orbit.py:137-155, :217-275
* supercell cluster-site index tables (equivalent-cluster major,
  translation minor; within-row order = the cluster's own site order)
      -> smol/cofe/space/clusterspace.py:1329-1366
* per-site reduced ("local") tables and the cluster ratio
      -> smol/moca/processor/expansion.py:120-138
* ECI and cluster-interaction tensors
      -> smol/cofe/expansion.py:172-201

Nothing here runs on the hot path: it is the table *producer*.  The consumer is
the HIP engine (smol_amd/csrc) through the C-ABI in include/smolmc.h.

Conventions: a crystal site is ``(b, n)`` = basis index ``b`` of the primitive
cell plus an integer lattice vector ``n``.  Supercell site index is
``b * P + t`` with ``t`` the linear index of the lattice point (sublattice
major, like the reference's supercell construction).
"""

from __future__ import annotations

import itertools
from dataclasses import dataclass, field
from functools import reduce

import numpy as np

EPS_MULT = 10  # smol/cofe/space/basis.py: zeroing threshold multiplier
SITE_TOL = 1e-6


# --------------------------------------------------------------------------- #
# site bases
# --------------------------------------------------------------------------- #
def site_function_array(nspecies, flavor="sinusoid", orthonormal=True, measure=None):
    """Rows = non-constant site functions phi_1..phi_{S-1}, columns = species code.

    Follows smol/cofe/space/basis.py: sinusoid (:573-586) phi_n(s) =
    -cos(2 pi ceil(n/2) s / S) for odd n, -sin(...) for even n; indicator
    (:448-459,:568-570) phi_n(s) = [s == n-1]; orthonormalisation (:234-258) is
    a QR of sqrt(measure) * F with the first row rescaled to the constant 1.
    """
    S = int(nspecies)
    if S < 2:
        return np.zeros((0, max(S, 1)))
    s = np.arange(S)
    rows = []
    for n in range(1, S):
        if flavor == "sinusoid":
            a = -(-n // 2)
            if n % 2 == 0:
                rows.append(-np.sin(2 * np.pi * a * s / S))
            else:
                rows.append(-np.cos(2 * np.pi * a * s / S))
        elif flavor == "indicator":
            rows.append((s == (n - 1)).astype(np.float64))
        else:
            raise ValueError(f"unknown basis flavor {flavor}")
    f_array = np.vstack((np.ones(S), np.array(rows)))
    if orthonormal:
        m = np.full(S, 1.0 / S) if measure is None else np.asarray(measure, float)
        q_mat, r_mat = np.linalg.qr((np.sqrt(m) * f_array).T, mode="complete")
        r_mat[abs(r_mat) < EPS_MULT * np.finfo(np.float64).eps] = 0.0
        q_mat[abs(q_mat) < EPS_MULT * np.finfo(np.float64).eps] = 0.0
        f_array = (q_mat.T / q_mat[:, 0]).astype(np.float64)
    return np.ascontiguousarray(f_array[1:], dtype=np.float64)


# --------------------------------------------------------------------------- #
# primitive cell + symmetry
# --------------------------------------------------------------------------- #
@dataclass
class PrimCell:
    """Primitive cell: lattice rows are the lattice vectors (Angstrom)."""

    lattice: np.ndarray  # (3,3)
    frac_coords: np.ndarray  # (nb,3)
    nspecies: list  # allowed species per basis site (1 => inactive)
    charges: list = None  # per basis site: list of charges per species code
    labels: list = None  # symmetry label per basis site (defaults to nspecies/charges)
    species: list = None  # species names per basis site, in code order (SiteSpace order)

    def __post_init__(self):
        self.lattice = np.asarray(self.lattice, dtype=np.float64)
        self.frac_coords = np.asarray(self.frac_coords, dtype=np.float64)
        self.nspecies = [int(x) for x in self.nspecies]
        if self.charges is None:
            self.charges = [[0.0] * s for s in self.nspecies]
        if self.labels is None:
            self.labels = [
                (self.nspecies[b], tuple(-1e9 if q is None else q for q in self.charges[b]))
                for b in range(self.nb)
            ]

    @property
    def nb(self):
        return len(self.frac_coords)

    @property
    def active(self):
        return [b for b in range(self.nb) if self.nspecies[b] > 1]

    def cart(self, b, n):
        return (self.frac_coords[b] + np.asarray(n)) @ self.lattice


def find_symops(prim: PrimCell):
    """Brute-force space-group operations (W, t) in fractional coordinates.

    x' = W x + t with W an integer matrix (entries in {-1,0,1}, sufficient for a
    reduced cell) preserving the metric tensor, t chosen so that the decorated
    basis maps onto itself.  Replaces the pymatgen SpacegroupAnalyzer call at
    smol/cofe/space/clusterspace.py:295.
    """
    A = prim.lattice  # rows = vectors
    G = A @ A.T
    ops = []
    cands = np.array(list(itertools.product((-1, 0, 1), repeat=9))).reshape(-1, 3, 3)
    dets = np.round(np.linalg.det(cands)).astype(int)
    cands = cands[np.abs(dets) == 1]
    # frac row-vector convention: x_frac' = x_frac @ Wt ; metric G' = Wt^T ... use col form
    # cart = frac @ A ; rotation acts on column frac: f' = W f ; metric: W^T Gc W = Gc, Gc = A A^T
    # metric preserved to 1e-5 relative: lattices read from files carry ~1e-7 noise (pymatgen's
    # symmetry finder, which the reference uses, works with a 0.1 A tolerance)
    tol = 1e-5 * np.abs(G).max()
    ok = np.all(np.abs(np.einsum("nji,jk,nkl->nil", cands, G, cands) - G) < tol, axis=(1, 2))
    cands = cands[ok]
    f0 = prim.frac_coords
    for W in cands:
        imgs = f0 @ W.T
        seen = set()
        for j in range(prim.nb):
            if prim.labels[j] != prim.labels[0]:
                continue
            t = f0[j] - imgs[0]
            t = t - np.floor(t + 1e-9)
            key = tuple(np.round(t, 6) % 1.0)
            if key in seen:
                continue
            seen.add(key)
            good = True
            for b in range(prim.nb):
                p = imgs[b] + t
                d = p[None, :] - f0
                d = d - np.round(d)
                hit = np.where(np.all(np.abs(d) < 1e-6, axis=1))[0]
                if len(hit) != 1 or prim.labels[hit[0]] != prim.labels[b]:
                    good = False
                    break
            if good:
                ops.append((W.copy(), t.copy()))
    return ops


def _apply_op(prim, op, members):
    """Image of a list of (b, n) sites under (W, t): returns list of (b', n')."""
    W, t = op
    out = []
    for b, n in members:
        f = prim.frac_coords[b] + np.asarray(n, dtype=np.float64)
        p = W @ f + t
        d = p[None, :] - prim.frac_coords
        r = np.round(d)
        hit = np.where(np.all(np.abs(d - r) < 1e-6, axis=1))[0]
        bb = int(hit[0])
        out.append((bb, tuple(int(x) for x in r[bb])))
    return out


def _canon_set(members):
    """Translation-invariant canonical key of a cluster as a *set* of sites."""
    best = None
    for _, n0 in members:
        shifted = sorted((b, tuple(np.subtract(n, n0))) for b, n in members)
        key = tuple(shifted)
        if best is None or key < best:
            best = key
    return best


def _diameter(prim, members):
    pts = np.array([prim.cart(b, n) for b, n in members])
    if len(pts) == 1:
        return 0.0
    d = np.linalg.norm(pts[:, None, :] - pts[None, :, :], axis=-1)
    return float(d.max())


# --------------------------------------------------------------------------- #
# orbits
# --------------------------------------------------------------------------- #
@dataclass
class Orbit:
    """One orbit of symmetry-equivalent clusters with its function tables."""

    base: list  # [(b, n)] base cluster, site order defines bit order
    clusters: list  # list of member lists (images, same site order as base)
    permutations: np.ndarray  # (nperm, I) int
    diameter: float
    nbits: list  # S_i - 1 per cluster site
    bases: list  # function arrays per cluster site
    id: int = -1
    bit_id: int = -1
    bit_combos: list = field(default_factory=list)
    corr_tensors: np.ndarray = None  # (K, S_1..S_I)

    @property
    def size(self):
        return len(self.base)

    @property
    def multiplicity(self):
        return len(self.clusters)

    @property
    def bit_combo_multiplicities(self):
        return [len(c) for c in self.bit_combos]

    def __len__(self):
        return len(self.bit_combos)

    @property
    def flat_correlation_tensors(self):
        ct = self.corr_tensors
        return np.ascontiguousarray(ct.reshape(ct.shape[0], -1), dtype=np.float64)

    @property
    def flat_tensor_indices(self):
        shape = self.corr_tensors.shape
        ind = np.cumprod(np.append(shape[2:], 1)[::-1])[::-1]
        return np.ascontiguousarray(ind, dtype=np.int32)

    def _finish(self):
        # bit combos: smol/cofe/space/orbit.py:137-155
        all_combos = []
        for bc in itertools.product(*[range(nb) for nb in self.nbits]):
            if not any(
                np.array_equal(bc, row) for combo in all_combos for row in combo
            ):
                bc = np.array(bc, dtype=np.int32)
                all_combos.append(np.unique(bc[self.permutations], axis=0))
        self.bit_combos = all_combos
        # correlation tensors: smol/cofe/space/orbit.py:217-249
        ct = np.zeros((len(all_combos), *(b.shape[1] for b in self.bases)))
        for k, combos in enumerate(all_combos):
            for bits in combos:
                ct[k] += reduce(
                    lambda a, b: np.tensordot(a, b, axes=0),
                    (self.bases[i][b] for i, b in enumerate(bits)),
                )
            ct[k] /= len(combos)
        self.corr_tensors = ct.astype(np.float64)


def _make_orbit(prim, symops, base, bases_by_b):
    base = [(b, tuple(n)) for b, n in base]
    key0 = _canon_set(base)
    clusters, keys, perms = [], [], []
    for op in symops:
        img = _apply_op(prim, op, base)
        k = _canon_set(img)
        if k == key0:
            # permutation: recentre and map (smol/cofe/space/orbit.py:446-470)
            # find translation aligning img with base as sets
            for _, n0 in img:
                for _, m0 in base:
                    sh = np.subtract(m0, n0)
                    shifted = [(b, tuple(np.add(n, sh))) for b, n in img]
                    if sorted(shifted) == sorted(base):
                        mapping = [shifted.index(s) for s in base]
                        perms.append(mapping)
                        break
                else:
                    continue
                break
        if k not in keys:
            keys.append(k)
            clusters.append(img)
    nstab = len(perms)
    perms = np.unique(np.array(perms, dtype=np.int64), axis=0)
    orb = Orbit(
        base=base,
        clusters=clusters,
        permutations=perms,
        diameter=_diameter(prim, base),
        nbits=[prim.nspecies[b] - 1 for b, _ in base],
        bases=[bases_by_b[b] for b, _ in base],
    )
    if nstab * len(clusters) != len(symops):
        raise RuntimeError("symmetry bookkeeping mismatch while building an orbit")
    orb._key = min(keys)
    return orb


@dataclass
class ClusterModel:
    """Orbit tables of a cluster subspace (our stand-in for ClusterSubspace)."""

    prim: PrimCell
    orbits: list
    symops: list
    basis_flavor: str
    num_orbits: int = 0
    num_corr_functions: int = 0

    # -- multiplicity vectors: smol/cofe/space/clusterspace.py:384-450 ------
    @property
    def orbit_multiplicities(self):
        return np.array([1] + [o.multiplicity for o in self.orbits])

    @property
    def function_orbit_ids(self):
        ids = [0]
        for o in self.orbits:
            ids += len(o) * [o.id]
        return np.array(ids)

    @property
    def function_ordering_multiplicities(self):
        return np.array(
            [1] + [m for o in self.orbits for m in o.bit_combo_multiplicities]
        )

    @property
    def function_total_multiplicities(self):
        return (
            self.orbit_multiplicities[self.function_orbit_ids]
            * self.function_ordering_multiplicities
        )

    def orbit_data(self):
        """Tuple of (id, bit_id, flat corr tensors, strides): smol/utils/cluster/__init__.py:4-15."""
        return tuple(
            (o.id, o.bit_id, o.flat_correlation_tensors, o.flat_tensor_indices)
            for o in self.orbits
        )

    # -- ECI / interaction tensors: smol/cofe/expansion.py:172-201 ----------
    def eci(self, coefs):
        return np.asarray(coefs, float) / self.function_total_multiplicities

    def cluster_interaction_tensors(self, coefs):
        eci = self.eci(coefs)
        out = [float(coefs[0])]
        for o in self.orbits:
            t = sum(
                m * eci[o.bit_id + i] * tensor
                for i, (m, tensor) in enumerate(
                    zip(o.bit_combo_multiplicities, o.corr_tensors)
                )
            )
            out.append(np.ascontiguousarray(t, dtype=np.float64))
        return out


def build_cluster_model(prim, cutoffs, basis="sinusoid", orthonormal=True):
    """Generate orbits from diameter cutoffs {size: cutoff}.

    Restates smol/cofe/space/clusterspace.py:1369-1565: point orbits for each
    symmetry-distinct active site; size-k orbits grown from size-(k-1) base
    clusters by adding one neighbour, kept when the diameter is within the
    cutoff, deduplicated by symmetry; sorted by (diameter, -multiplicity,
    #functions) within a size; ids and bit ids start at 1.
    """
    symops = find_symops(prim)
    bases_by_b = {
        b: site_function_array(prim.nspecies[b], basis, orthonormal)
        for b in prim.active
    }
    zero = (0, 0, 0)
    orbits = {1: []}
    seen = set()
    for b in prim.active:
        orb = _make_orbit(prim, symops, [(b, zero)], bases_by_b)
        if orb._key not in seen:
            seen.add(orb._key)
            orbits[1].append(orb)
    orbits[1].sort(key=lambda o: (-o.multiplicity, 0))
    for size in sorted(cutoffs):
        if size < 2:
            continue
        cutoff = cutoffs[size]
        rng = int(np.ceil(cutoff / np.min(np.linalg.norm(prim.lattice, axis=1)))) + 1
        neigh = [
            (b, n)
            for b in prim.active
            for n in itertools.product(range(-rng, rng + 1), repeat=3)
        ]
        new, seen = [], set()
        for orb in orbits.get(size - 1, []):
            if orb.diameter > cutoff:
                continue
            base_pts = np.array([prim.cart(b, n) for b, n in orb.base])
            for b, n in neigh:
                if (b, n) in orb.base:
                    continue
                p = prim.cart(b, n)
                if np.max(np.linalg.norm(base_pts - p, axis=1)) > cutoff + 1e-8:
                    continue
                members = orb.base + [(b, n)]
                key = None
                # cheap orbit key: min canonical set over all symops
                keys = [_canon_set(_apply_op(prim, op, members)) for op in symops]
                key = min(keys)
                if key in seen:
                    continue
                seen.add(key)
                new.append(_make_orbit(prim, symops, members, bases_by_b))
        for o in new:
            o._finish()
        if new:
            new.sort(key=lambda o: (np.round(o.diameter, 6), -o.multiplicity, len(o)))
            orbits[size] = new
    for o in orbits[1]:
        o._finish()
    ordered = [o for size in sorted(orbits) for o in orbits[size]]
    oid, bid = 1, 1
    for o in ordered:
        o.id, o.bit_id = oid, bid
        oid += 1
        bid += len(o)
    model = ClusterModel(prim, ordered, symops, basis)
    model.num_orbits, model.num_corr_functions = oid, bid
    return model


# --------------------------------------------------------------------------- #
# supercell tables
# --------------------------------------------------------------------------- #
@dataclass
class SupercellTables:
    """All read-only tables for one (model, supercell) pair."""

    model: ClusterModel
    scmatrix: np.ndarray
    size: int  # P
    num_sites: int
    lattice_points: np.ndarray  # (P,3) integer vectors, index = t
    site_b: np.ndarray  # (N,) basis index of each site
    site_t: np.ndarray  # (N,) lattice point index of each site
    full_indices: list  # per orbit int32 [mult*P, I]
    local_sites: dict = None  # built lazily

    def site_index(self, b, n):
        return b * self.size + self._point_index(np.asarray(n))

    def _point_index(self, n):
        inv = self._inv
        f = np.asarray(n, dtype=np.float64) @ inv  # supercell frac coords
        f = f - np.floor(f + 1e-9)
        key = np.round(f * self._det).astype(np.int64) % self._det
        if key.ndim == 1:
            return self._lut[tuple(key)]
        return self._lut[key[..., 0], key[..., 1], key[..., 2]]

    # -- local tables: smol/moca/processor/expansion.py:120-138 -------------
    def local_tables(self):
        """dict site -> list of (orbit_pos, rows int32[J,I], ratio).

        ``orbit_pos`` indexes ``model.orbits``.  Row order is ascending row
        index of the full table, as ``cluster_indices[in_inds]`` gives.
        """
        if self.local_sites is not None:
            return self.local_sites
        out = {s: [] for s in range(self.num_sites)}
        for pos, rows in enumerate(self.full_indices):
            J, I = rows.shape
            rid = np.arange(J)
            pairs_s, pairs_r = [], []
            for i in range(I):
                fresh = np.ones(J, dtype=bool)
                for i2 in range(i):
                    fresh &= rows[:, i2] != rows[:, i]
                pairs_s.append(rows[fresh, i])
                pairs_r.append(rid[fresh])
            ps = np.concatenate(pairs_s)
            pr = np.concatenate(pairs_r)
            order = np.lexsort((pr, ps))
            ps, pr = ps[order], pr[order]
            bounds = np.flatnonzero(np.diff(ps)) + 1
            starts = np.concatenate(([0], bounds))
            ends = np.concatenate((bounds, [len(ps)]))
            for a, e in zip(starts, ends):
                s = int(ps[a])
                loc = np.ascontiguousarray(rows[pr[a:e]], dtype=np.int32)
                out[s].append((pos, loc, J / (e - a)))
        self.local_sites = {s: v for s, v in out.items() if v}
        return self.local_sites


def build_supercell(model: ClusterModel, scmatrix):
    """Cluster-site index tables for a supercell (rows of scmatrix = supercell vectors
    in units of the primitive vectors).  smol/cofe/space/clusterspace.py:1329-1366."""
    scm = np.asarray(scmatrix, dtype=np.int64)
    if scm.ndim == 1:
        scm = np.diag(scm)
    det = int(round(abs(np.linalg.det(scm))))
    inv = np.linalg.inv(scm.astype(np.float64))
    # enumerate lattice points inside the supercell
    if np.array_equal(scm, np.diag(np.diag(scm))):
        n1, n2, n3 = (int(x) for x in np.diag(scm))
        pts = np.array(list(itertools.product(range(n1), range(n2), range(n3))))
    else:
        corners = np.array(list(itertools.product((0, 1), repeat=3))) @ scm
        lo, hi = corners.min(axis=0), corners.max(axis=0)
        grid = np.array(
            list(itertools.product(*[range(int(a), int(b) + 1) for a, b in zip(lo, hi)]))
        )
        f = grid @ inv
        inside = np.all((f > -1e-9) & (f < 1 - 1e-9), axis=1)
        pts = grid[inside]
        fk = np.round(pts @ inv * det).astype(np.int64)
        pts = pts[np.lexsort((fk[:, 2], fk[:, 1], fk[:, 0]))]
    assert len(pts) == det, (len(pts), det)
    prim = model.prim
    P = det
    N = prim.nb * P
    sc = SupercellTables(
        model=model,
        scmatrix=scm,
        size=P,
        num_sites=N,
        lattice_points=pts,
        site_b=np.repeat(np.arange(prim.nb), P),
        site_t=np.tile(np.arange(P), prim.nb),
        full_indices=[],
    )
    sc._inv, sc._det = inv, det
    lut = np.full((det, det, det), -1, dtype=np.int64) if det <= 64 else None
    keys = np.round((pts @ inv) * det).astype(np.int64) % det
    if lut is None:
        sc._lut = _DictLut({tuple(k): i for i, k in enumerate(keys)})
    else:
        lut[keys[:, 0], keys[:, 1], keys[:, 2]] = np.arange(det)
        sc._lut = lut
    for orb in model.orbits:
        rows = np.empty((orb.multiplicity * P, orb.size), dtype=np.int64)
        for e, members in enumerate(orb.clusters):
            for i, (b, n) in enumerate(members):
                t = sc._point_index(pts + np.asarray(n)[None, :])
                rows[e * P : (e + 1) * P, i] = b * P + t
        sc.full_indices.append(np.ascontiguousarray(rows, dtype=np.int32))
    return sc


class _DictLut:
    """(k0,k1,k2) -> lattice point index for large supercells (vectorised lookup)."""

    def __init__(self, d):
        ks = np.array(list(d.keys()), dtype=np.int64)
        vs = np.array(list(d.values()), dtype=np.int64)
        self._m = int(ks.max()) + 1
        flat = (ks[:, 0] * self._m + ks[:, 1]) * self._m + ks[:, 2]
        order = np.argsort(flat)
        self._flat, self._vals = flat[order], vs[order]

    def __getitem__(self, key):
        k0, k1, k2 = key
        flat = (np.asarray(k0) * self._m + np.asarray(k1)) * self._m + np.asarray(k2)
        pos = np.searchsorted(self._flat, flat)
        return self._vals[pos]


# --------------------------------------------------------------------------- #
# canned crystals
# --------------------------------------------------------------------------- #
def fcc_prim(a=4.09, nspecies=2):
    """FCC primitive cell, one active site (AuPd-like, tests/data/AuPd_prim.json lattice)."""
    lat = 0.5 * a * np.array([[0, 1, 1], [1, 0, 1], [1, 1, 0]], dtype=float)
    return PrimCell(lat, [[0, 0, 0]], [nspecies])


def fcc_conventional_prim(a=4.09, nspecies=2):
    """FCC as a simple-cubic cell with a 4-site basis (config 1's 4x4x4 conventional cell)."""
    lat = a * np.eye(3)
    fc = [[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]]
    return PrimCell(lat, fc, [nspecies] * 4)


def rocksalt_prim(a=4.2, cation_charges=(1.0, 3.0, 4.0), anion_charge=-2.0, anion_charges=None):
    """Rocksalt primitive cell: cation site (len(cation_charges) species) + anion site.

    A charge of ``None`` marks a Vacancy (last in the site space like the reference,
    smol/cofe/space/domain.py:157-161; it gets no Ewald index).  ``anion_charges`` with more
    than one entry makes the anion sublattice active too (e.g. (-2, -1) for O2-/F-)."""
    lat = 0.5 * a * np.array([[0, 1, 1], [1, 0, 1], [1, 1, 0]], dtype=float)
    names = ["Li+", "Mn3+", "Ti4+", "Nb5+", "Zr4+"]
    an = list(anion_charges) if anion_charges is not None else [anion_charge]
    cat_names = [("Vacancy" if q is None else names[i]) for i, q in enumerate(cation_charges)]
    return PrimCell(
        lat,
        [[0, 0, 0], [0.5, 0.5, 0.5]],
        [len(cation_charges), len(an)],
        charges=[list(cation_charges), an],
        species=[cat_names, ["O2-", "F-", "S2-"][: len(an)]],
    )


def random_coefs(model, seed=20260928, scale=0.02, empty=0.0):
    """coefs[0]=empty, others U(-scale, scale) * total multiplicity."""
    rng = np.random.default_rng(seed)
    eci = rng.uniform(-scale, scale, size=model.num_corr_functions)
    eci[0] = empty
    coefs = eci * model.function_total_multiplicities
    coefs[0] = empty
    return coefs
