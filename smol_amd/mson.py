"""Import of smol's serialized models (MSON / JSON ``as_dict`` output) without pymatgen or smol.

SURVEY.md §8(f) rank 2: the input side of the hot path.  Reads the dictionaries written by

    ClusterExpansion.as_dict      smol/cofe/expansion.py:486-535
    ClusterSubspace.as_dict       smol/cofe/space/clusterspace.py:1703-1727
    Orbit.as_dict                 smol/cofe/space/orbit.py:581-600
    StructureWrangler entries     smol/cofe/wrangling/wrangler.py (ComputedStructureEntry.data)

and produces the flattened tables of include/smolmc.h for any supercell matrix:

    correlation tensors   mean over the stored bit combos of the outer products of the stored
                          site-basis rows (orbit.py:217-249), C-order flattening and strides
                          (:251-275); ids in orbit order (clusterspace.py:1297-1310)
    cluster index tables  the model's own cached ``_supercell_orb_inds`` when it holds the
                          supercell, else regenerated from the stored base clusters and symmetry
                          operations (orbit.py:173-190, clusterspace.py:1329-1366)
    local tables          per-site row subsets + ratios (moca/processor/expansion.py:120-138)
    interaction tensors   cofe/expansion.py:172-201
    Ewald tables          index layout of cofe/extern/ewald.py:64-100 over the supercell; matrix
                          from smol_amd.ewald (own Ewald sum, pymatgen's screening parameter
                          and cutoffs) -- checked against the Ewald energies the reference
                          stored with its training structures (tests/test_mson_golden.py)
    sublattices           sites grouped by identical site space (moca/processor/base.py:245-268)

Two conventions of pymatgen (pinned pymatgen==2025.01.09, not vendored in the reference) have
to be restated because they fix the *ordering* the stored tables rely on:

  * supercell site order: for each site of the primitive cell, all lattice translations in the
    order of ``lattice_points_in_supercell`` (integer points of the primitive lattice inside the
    supercell, enumerated a-major in primitive coordinates);
  * species order inside a site space: ``sorted()`` of pymatgen ``Species`` = by Pauling
    electronegativity, then symbol, then oxidation state; a vacancy is always last
    (cofe/space/domain.py:157-161).

Both are verified on the reference's own data: the regenerated cluster index tables equal the 17
cached ones of docs/src/notebooks/data/basic_ce*.mson entry for entry, and occupancies built with
these conventions reproduce the stored correlation vectors (tests/test_mson_golden.py).
"""

from __future__ import annotations

import functools
import itertools
import json

import numpy as np

from . import capi
from . import ewald as ewald_mod

SITE_TOL = 1e-6  # smol/utils/cluster/... SITE_TOL used by coord_list_mapping_pbc / is_coord_subset

# Pauling electronegativities (the ``X`` pymatgen sorts species by); NaN entries of pymatgen's
# table (noble gases without a value) sort last, as there.
PAULING_X = {
    "H": 2.20, "Li": 0.98, "Be": 1.57, "B": 2.04, "C": 2.55, "N": 3.04, "O": 3.44, "F": 3.98,
    "Na": 0.93, "Mg": 1.31, "Al": 1.61, "Si": 1.90, "P": 2.19, "S": 2.58, "Cl": 3.16,
    "K": 0.82, "Ca": 1.00, "Sc": 1.36, "Ti": 1.54, "V": 1.63, "Cr": 1.66, "Mn": 1.55, "Fe": 1.83,
    "Co": 1.88, "Ni": 1.91, "Cu": 1.90, "Zn": 1.65, "Ga": 1.81, "Ge": 2.01, "As": 2.18, "Se": 2.55,
    "Br": 2.96, "Kr": 3.00, "Rb": 0.82, "Sr": 0.95, "Y": 1.22, "Zr": 1.33, "Nb": 1.6, "Mo": 2.16,
    "Tc": 1.9, "Ru": 2.2, "Rh": 2.28, "Pd": 2.20, "Ag": 1.93, "Cd": 1.69, "In": 1.78, "Sn": 1.96,
    "Sb": 2.05, "Te": 2.1, "I": 2.66, "Xe": 2.6, "Cs": 0.79, "Ba": 0.89, "La": 1.10, "Ce": 1.12,
    "Pr": 1.13, "Nd": 1.14, "Sm": 1.17, "Gd": 1.20, "Dy": 1.22, "Ho": 1.23, "Er": 1.24, "Tm": 1.25,
    "Lu": 1.27, "Hf": 1.3, "Ta": 1.5, "W": 2.36, "Re": 1.9, "Os": 2.2, "Ir": 2.20, "Pt": 2.28,
    "Au": 2.54, "Hg": 2.00, "Tl": 1.62, "Pb": 2.33, "Bi": 2.02, "Th": 1.3, "U": 1.38,
}


class Species:
    """Element + oxidation state, ordered like pymatgen's ``Species`` / ``Element``."""

    def __init__(self, element, oxidation_state=None):
        self.element = str(element)
        self.oxidation_state = None if oxidation_state is None else float(oxidation_state)

    @property
    def charge(self):
        return 0.0 if self.oxidation_state is None else self.oxidation_state

    @property
    def name(self):
        q = self.oxidation_state
        if q is None:
            return self.element
        mag = abs(q)
        txt = "" if mag == 1 else (str(int(mag)) if float(mag).is_integer() else f"{mag:g}")
        return f"{self.element}{txt}{'+' if q >= 0 else '-'}"

    def sort_key(self):
        if self.element not in PAULING_X:
            raise ValueError(
                f"no electronegativity on record for {self.element!r}: cannot reproduce pymatgen's "
                "species order for this site space (extend smol_amd.mson.PAULING_X)")
        return (PAULING_X[self.element], self.element, self.charge)

    def __eq__(self, other):
        return isinstance(other, Species) and self.name == other.name

    def __hash__(self):
        return hash(self.name)

    def __repr__(self):
        return self.name


VACANCY = "Vacancy"


def site_space_of(species_dicts):
    """Ordered allowed species of one site from a pymatgen ``PeriodicSite.species`` list:
    sorted species, implicit vacancy last when the occupations sum to < 0.99
    (cofe/space/domain.py:70-82,157-161).  Returns (names, charges) -- vacancy = (VACANCY, None)."""
    sp = sorted((Species(s["element"], s.get("oxidation_state")) for s in species_dicts),
                key=Species.sort_key)
    names = [s.name for s in sp]
    charges = [s.charge for s in sp]
    if sum(float(s.get("occu", 1.0)) for s in species_dicts) < 0.99:
        names.append(VACANCY)
        charges.append(None)
    return tuple(names), tuple(charges)


# --------------------------------------------------------------------------------------
# pymatgen conventions restated (ordering only)
# --------------------------------------------------------------------------------------
def lattice_points_in_supercell(scmatrix):
    """Fractional coordinates (supercell basis) of the primitive-lattice points inside the
    supercell, in pymatgen's enumeration order: the integer points of the bounding box of the
    supercell's corners, first primitive coordinate slowest, kept when inside [0, 1)^3."""
    scm = np.asarray(scmatrix, dtype=np.int64)
    corners = np.array(list(itertools.product((0, 1), repeat=3)), dtype=np.int64) @ scm
    lo, hi = corners.min(axis=0), corners.max(axis=0) + 1
    grid = np.array(list(itertools.product(*[range(int(a), int(b)) for a, b in zip(lo, hi)])),
                    dtype=np.float64)
    frac = grid @ np.linalg.inv(scm.astype(np.float64))
    keep = np.all(frac < 1 - 1e-10, axis=1) & np.all(frac >= -1e-10, axis=1)
    pts = frac[keep]
    if len(pts) != int(round(abs(np.linalg.det(scm)))):
        raise RuntimeError("lattice point enumeration failed for this supercell matrix")
    return pts


def _wrap01(x):
    w = x - np.floor(x)
    w[w >= 1.0] = 0.0  # (x = -1e-17 wraps to 1.0 in float64; the tree wants [0, 1))
    return w


class _PbcMatcher:
    """Matches points to ``targets`` modulo lattice translations (what pymatgen's coord_list_mapping_pbc does:
    every fractional coordinate within ``atol`` of the target's, modulo 1).  A periodic k-d tree over the wrapped
    targets (maximum norm on the unit torus -- the same criterion) instead of all point x target differences: the
    cluster tables of an 8^3 LiNiO2 supercell 34 s -> 0.3 s, entry for entry the same
    (tests/test_mson_golden.py regenerates the reference's cached tables with it)."""

    def __init__(self, targets, atol=SITE_TOL):
        self.targets, self.atol = np.asarray(targets, dtype=np.float64), atol
        try:
            from scipy.spatial import cKDTree

            self.tree = cKDTree(_wrap01(self.targets), boxsize=1.0)
        except ImportError:  # (no scipy: the all-pairs comparison)
            self.tree = None

    def match(self, points):
        """Index into the targets of every row of ``points``; raises if a point has no image."""
        points = np.asarray(points, dtype=np.float64)
        if self.tree is None:
            return _pbc_match_all_pairs(points, self.targets, self.atol)
        # (cKDTree's bound is a strict `<`; the all-pairs definition and pymatgen's coord_list_mapping_pbc accept a
        # difference of exactly atol: one ulp above it makes the two agree at the tolerance boundary)
        dist, idx = self.tree.query(_wrap01(points), k=1, p=np.inf, distance_upper_bound=np.nextafter(self.atol, np.inf),
                                    workers=-1 if len(points) > 100000 else 1)
        if not np.all(np.isfinite(dist)):
            raise ValueError("a cluster site has no image in the supercell")
        return idx.astype(np.int64)


def _pbc_match(points, targets, atol=SITE_TOL):
    return _PbcMatcher(targets, atol).match(points)


def _pbc_match_all_pairs(points, targets, atol=SITE_TOL):
    """_pbc_match by comparing every point with every target (the definition; O(points x targets))."""
    t = targets - np.floor(targets + 1e-12)
    out = np.empty(len(points), dtype=np.int64)
    for start in range(0, len(points), 4096):
        p = points[start:start + 4096]
        d = p[:, None, :] - t[None, :, :]
        d -= np.round(d)
        hit = np.all(np.abs(d) <= atol, axis=-1)
        if not np.all(hit.any(axis=1)):
            raise ValueError("a cluster site has no image in the supercell")
        out[start:start + 4096] = hit.argmax(axis=1)
    return out


# --------------------------------------------------------------------------------------
class MsonOrbit:
    """One orbit of the serialized ClusterSubspace (orbit.py:564-600)."""

    def __init__(self, d, oid, bit_id):
        self.id, self.bit_id = oid, bit_id
        self.frac_coords = np.asarray(d["sites"], dtype=np.float64)  # base cluster, prim basis
        self.bits = [list(b) for b in d["bits"]]
        self.symops = [np.asarray(s["matrix"], dtype=np.float64) for s in d["structure_symops"]]
        # site bases: rows phi_0 == 1, phi_1 .. phi_{S-1} (basis.py:207-222); the evaluator uses
        # rows 1.. ("function_array")
        self.basis_arrays = [np.asarray(sb["func_array"], dtype=np.float64)[1:] for sb in d["site_bases"]]
        self.flavor = d["site_bases"][0].get("flavor")
        combos = d.get("_bit_combos")
        if combos is None:
            raise ValueError("orbit without stored _bit_combos (file written by a very old smol)")
        self.bit_combos = [np.asarray(c, dtype=np.int32).reshape(-1, len(self.bits)) for c in combos]
        self._clusters = None

    num_sites = property(lambda self: len(self.frac_coords))
    bit_combo_multiplicities = property(lambda self: [len(c) for c in self.bit_combos])

    def __len__(self):
        return len(self.bit_combos)

    @property
    def correlation_tensors(self):
        """[K, S_1, .., S_I]: mean over each combo's bit orderings of the outer product of the
        sites' basis rows (orbit.py:234-249)."""
        shape = tuple(a.shape[1] for a in self.basis_arrays)
        out = np.zeros((len(self.bit_combos),) + shape)
        for k, combo in enumerate(self.bit_combos):
            for bits in combo:
                term = np.ones(())
                for i, b in enumerate(bits):
                    term = np.tensordot(term, self.basis_arrays[i][b], axes=0)
                out[k] += term
            out[k] /= len(combo)
        return out

    @property
    def flat_correlation_tensors(self):
        ct = self.correlation_tensors
        return np.ascontiguousarray(ct.reshape(ct.shape[0], -1))

    @property
    def flat_tensor_indices(self):
        shape = self.correlation_tensors.shape
        return np.ascontiguousarray(np.cumprod(np.append(shape[2:], 1)[::-1])[::-1], dtype=np.int32)

    @property
    def clusters(self):
        """Symmetry-equivalent clusters in the reference's generation order: the base cluster,
        then each structure symmetry operation in stored order applied to it, kept when it is not
        a lattice translate of one already found (orbit.py:173-190, cluster.py:44-59,214-219)."""
        if self._clusters is None:
            def normalise(fc):
                c = fc.mean(axis=0)
                shift = np.floor(c)
                return fc - shift, c - shift

            found = [normalise(self.frac_coords)]
            for op in self.symops:
                new = self.frac_coords @ op[:3, :3].T + op[:3, 3]
                fc, c = normalise(new)
                dup = False
                for gc, cc in found:
                    other = fc + np.round(cc - c)
                    if all(np.any(np.all(np.abs(other - row) < SITE_TOL, axis=1)) for row in gc):
                        dup = True
                        break
                if not dup:
                    found.append((fc, c))
            self._clusters = [f[0] for f in found]
        return self._clusters

    multiplicity = property(lambda self: len(self.clusters))


class MsonSubspace:
    """ClusterSubspace dictionary -> everything the engine tables need."""

    def __init__(self, d):
        st = d["structure"]
        self.lattice = np.asarray(st["lattice"]["matrix"], dtype=np.float64)
        self.frac_coords = np.array([s["abc"] for s in st["sites"]], dtype=np.float64)
        spaces = [site_space_of(s["species"]) for s in st["sites"]]
        self.site_species = [sp[0] for sp in spaces]  # names per prim site, code order
        self.site_charges = [sp[1] for sp in spaces]
        self.orbits = []
        oid = bit = 1
        for size in sorted(d["orbits"], key=int):  # clusterspace.py:1303-1306
            for od in d["orbits"][size]:
                orb = MsonOrbit(od, oid, bit)
                self.orbits.append(orb)
                oid += 1
                bit += len(orb)
        self.num_orbits, self.num_corr_functions = oid, bit
        self.external_terms = [t.get("@class") for t in d.get("external_terms", [])]
        self.ewald_term = next((t for t in d.get("external_terms", []) if t.get("@class") == "EwaldTerm"), None)
        # cached index tables, keyed like the reference: sorted rows (clusterspace.py:1319-1321)
        self.cached_indices = {}
        for scm, arrays in d.get("_supercell_orb_inds", []):
            key = tuple(sorted(tuple(int(x) for x in row) for row in scm))
            if arrays and isinstance(arrays[0][0], int):  # pre-release layout [(orbit id, rows)]
                arrays = [a[1] for a in arrays]
            self.cached_indices[key] = tuple(np.asarray(a, dtype=np.int32) for a in arrays)

    # ---- per-function bookkeeping (clusterspace.py:384-450) -----------------------------
    @property
    def orbit_multiplicities(self):
        return np.array([1] + [o.multiplicity for o in self.orbits])

    @property
    def function_orbit_ids(self):
        return np.array([0] + [o.id for o in self.orbits for _ in range(len(o))])

    @property
    def function_total_multiplicities(self):
        ordering = np.array([1] + [m for o in self.orbits for m in o.bit_combo_multiplicities])
        return self.orbit_multiplicities[self.function_orbit_ids] * ordering

    def orbit_data(self):
        """smol/utils/cluster/__init__.py:4-15."""
        return tuple((o.id, o.bit_id, o.flat_correlation_tensors, o.flat_tensor_indices)
                     for o in self.orbits)

    def eci(self, coefs):
        """cofe/expansion.py:172-184 (coefficients of the correlation functions only)."""
        return np.asarray(coefs, dtype=np.float64)[: self.num_corr_functions] / self.function_total_multiplicities

    def cluster_interaction_tensors(self, coefs):
        """cofe/expansion.py:186-201: [coefs[0], per orbit sum_k m_k eci[bit_id + k] ct_k]."""
        eci = self.eci(coefs)
        out = [float(coefs[0])]
        for o in self.orbits:
            ct = o.correlation_tensors
            out.append(sum(m * eci[o.bit_id + k] * ct[k] for k, m in enumerate(o.bit_combo_multiplicities)))
        return out

    @functools.cached_property
    def prim(self):
        """The primitive cell in the shape smol_amd.synth / smol_amd.moca work with."""
        from .synth import PrimCell

        return PrimCell(self.lattice, self.frac_coords, [len(sp) for sp in self.site_species],
                        charges=[list(q) for q in self.site_charges], labels=list(self.site_species),
                        species=[list(sp) for sp in self.site_species])

    # ---- supercells ------------------------------------------------------------------------
    def supercell(self, scmatrix):
        return MsonSupercell(self, scmatrix)

    def generate_orbit_indices(self, scmatrix, supercell_frac=None):
        """clusterspace.py:1329-1366 on the stored clusters: rows = (equivalent cluster major,
        lattice translation minor), columns = the cluster's own site order."""
        scm = np.asarray(scmatrix, dtype=np.int64)
        inv = np.linalg.inv(scm.astype(np.float64))
        pts = lattice_points_in_supercell(scm)
        if supercell_frac is None:
            supercell_frac = self.supercell(scm).frac_coords
        out = []
        matcher = _PbcMatcher(supercell_frac)
        for orb in self.orbits:
            prim = np.array(orb.clusters)  # (mult, I, 3)
            fc = prim @ inv
            t = fc[:, None, :, :] + pts[None, :, None, :]
            rows = matcher.match(t.reshape(-1, 3)).reshape(-1, orb.num_sites)
            out.append(np.ascontiguousarray(rows, dtype=np.int32))
        return tuple(out)

    def orbit_indices(self, scmatrix):
        """The model's cached table when it has one for this supercell (the reference would use
        exactly that), else regenerated."""
        key = tuple(sorted(tuple(int(x) for x in row) for row in np.asarray(scmatrix)))
        return self.cached_indices.get(key) or self.generate_orbit_indices(scmatrix)


class MsonSupercell:
    """The primitive structure times a supercell matrix, in pymatgen's site order."""

    def __init__(self, subspace, scmatrix):
        self.subspace = subspace
        self.scmatrix = np.asarray(scmatrix, dtype=np.int64)
        self.size = int(round(abs(np.linalg.det(self.scmatrix))))
        self.lattice = self.scmatrix.astype(np.float64) @ subspace.lattice
        pts = lattice_points_in_supercell(self.scmatrix)
        inv = np.linalg.inv(self.scmatrix.astype(np.float64))
        nb = len(subspace.frac_coords)
        self.site_prim = np.repeat(np.arange(nb), self.size)  # prim site of every supercell site
        self.frac_coords = (subspace.frac_coords @ inv)[self.site_prim] + np.tile(pts, (nb, 1))
        self.num_sites = nb * self.size
        self.lattice_points = pts
        # the attribute names smol_amd.synth.SupercellTables carries, so that capi.TableSet.from_synth
        # and the smol_amd.moca processors take an imported model like a generated one
        self.model, self.site_b = subspace, self.site_prim
        self._full = None
        self.nspecies = np.array([len(subspace.site_species[b]) for b in self.site_prim], dtype=np.int32)

    @property
    def full_indices(self):
        if self._full is None:
            self._full = self.subspace.orbit_indices(self.scmatrix)
        return self._full

    def local_tables(self):
        return local_tables(self.full_indices, self.num_sites)

    def occupancy_from_sites(self, species_names, site_mapping):
        """Encoded occupancy of a structure whose site j sits on supercell site
        ``site_mapping[j]`` (clusterspace.py:834-856): unmapped sites hold the vacancy."""
        sub = self.subspace
        occ = np.empty(self.num_sites, dtype=np.int32)
        where = {int(s): j for j, s in enumerate(site_mapping)}
        for i, b in enumerate(self.site_prim):
            allowed = sub.site_species[b]
            name = species_names[where[i]] if i in where else VACANCY
            if name not in allowed:
                raise ValueError(f"A site in given structure has an  unrecognized species {name}.")
            occ[i] = allowed.index(name)
        return occ

    def occupancy_from_coords(self, species_names, frac_coords):
        """Encoded occupancy of a perfect-lattice structure given in this supercell's basis (the
        wrangler's ``refined_structure``): every site is matched to the supercell site at its
        position; unmatched supercell sites hold the vacancy."""
        return self.occupancy_from_sites(species_names, _pbc_match(np.asarray(frac_coords), self.frac_coords,
                                                                   atol=1e-4))

    def sublattices(self):
        """Sites grouped by identical site space, in order of first appearance
        (processor/base.py:245-268); single-species sublattices are inactive."""
        groups = {}
        for i, b in enumerate(self.site_prim):
            groups.setdefault(self.subspace.site_species[b], []).append(i)
        out = []
        for names, sites in groups.items():
            b0 = self.site_prim[sites[0]]
            out.append(dict(species=names, charges=self.subspace.site_charges[b0],
                            sites=np.array(sites), codes=np.arange(len(names)),
                            active_sites=np.array(sites if len(names) > 1 else [], dtype=np.int64)))
        return out

    def ewald_tables(self, eta=None, real_space_cut=None, recip_space_cut=None, use_term="total"):
        """(ewald_inds int32[N, max_species], matrix f64[M, M], charges f64[M]).

        Index table: running counter over (site, non-vacancy species) in site order, -1 elsewhere
        (cofe/extern/ewald.py:84-97).  Matrix: total Ewald matrix of the structure that carries
        every allowed species on its site (point terms on the diagonal), pymatgen's conventions
        for the screening parameter and cutoffs (smol_amd.ewald.ewald_matrix_pmg)."""
        sub = self.subspace
        width = int(self.nspecies.max())
        inds = -np.ones((self.num_sites, width), dtype=np.int32)
        site_of, q = [], []
        for i, b in enumerate(self.site_prim):
            for code, charge in enumerate(sub.site_charges[b]):
                if charge is None:
                    continue
                inds[i, code] = len(site_of)
                site_of.append(i)
                q.append(charge)
        site_of, q = np.array(site_of), np.array(q, dtype=np.float64)
        trans = None
        if self.size > 1:
            # index of the lattice point t2 - t1 (the first translation is the origin, so the row of
            # site (b, 0) is the representative row of prim site b)
            pts = self.lattice_points
            trans = np.empty((self.size, self.size), dtype=np.int64)
            matcher = _PbcMatcher(pts)
            for t1 in range(self.size):
                trans[t1] = matcher.match(pts - pts[t1])
            if not np.allclose(pts[0], 0.0):
                trans = None
        mat = ewald_mod.ewald_matrix_pmg(self.lattice, self.frac_coords, site_of, q, eta=eta,
                                         real_space_cut=real_space_cut, recip_space_cut=recip_space_cut,
                                         translation_index=trans, use_term=use_term)
        return inds, mat, q


def local_tables(full_indices, num_sites):
    """Per-site reduction of the full cluster tables (moca/processor/expansion.py:120-138):
    site -> [(orbit position, rows containing the site, ratio = rows_full / rows_local)]."""
    out = {}
    for pos, rows in enumerate(full_indices):
        if rows.size == 0:
            continue
        # the rows that contain a site, for every site at once: (site, row) pairs sorted by site then row, a row
        # that names a site twice (aliased tiny supercells) counted once -- the same lists, in the same order, as
        # `rows[np.any(rows == site, axis=-1)]` site by site
        nr, width = rows.shape
        flat, rid = rows.ravel(), np.repeat(np.arange(nr), width)
        order = np.lexsort((rid, flat))
        fs, fr = flat[order], rid[order]
        keep = np.ones(len(fs), dtype=bool)
        keep[1:] = (fs[1:] != fs[:-1]) | (fr[1:] != fr[:-1])
        fs, fr = fs[keep], fr[keep]
        cut = np.flatnonzero(np.concatenate(([True], fs[1:] != fs[:-1], [True])))
        for a, b in zip(cut[:-1], cut[1:]):
            out.setdefault(int(fs[a]), []).append((pos, np.ascontiguousarray(rows[fr[a:b]]), nr / float(b - a)))
    return out


class MsonClusterExpansion:
    """ClusterExpansion dictionary (cofe/expansion.py:486-535): subspace + coefficients."""

    def __init__(self, d):
        if "ClusterExpansion" in d and "cluster_subspace" not in d:
            d = d["ClusterExpansion"]  # the notebooks' save_work files hold several objects
        self.subspace = MsonSubspace(d["cluster_subspace"])
        self.coefs = np.asarray(d["coefs"], dtype=np.float64)
        fm = d.get("feature_matrix")
        self.feature_matrix = None if fm is None else np.asarray(fm, dtype=np.float64)
        self.n_external = len(self.subspace.external_terms)
        want = self.subspace.num_corr_functions + self.n_external
        if len(self.coefs) != want:
            raise AttributeError(
                f"Feature matrix shape does not match the number of coefficients: {len(self.coefs)} "
                f"coefficients for {want} features.")

    @property
    def ce_coefs(self):
        return self.coefs[: self.subspace.num_corr_functions]

    @property
    def eci(self):
        return self.subspace.eci(self.coefs)

    @property
    def cluster_interaction_tensors(self):
        return self.subspace.cluster_interaction_tensors(self.coefs)

    def tables(self, scmatrix, feature_mode=capi.FEATURES_INTERACTIONS, mu_table=None,
               with_ewald=None, **usher):
        """capi.TableSet of this expansion on a supercell: what ``Ensemble.from_cluster_expansion``
        assembles in the reference (moca/ensemble.py:135-217): a ClusterDecompositionProcessor
        (default) or ClusterExpansionProcessor, composed with an EwaldProcessor whose coefficient is
        the last fitted coefficient when the subspace carries an EwaldTerm (:191-199)."""
        sub = self.subspace
        cell = sub.supercell(scmatrix)
        ew = coef = q = None
        if with_ewald is None:
            with_ewald = sub.ewald_term is not None
        if with_ewald:
            inds, mat, q = self.ewald_tables(cell)
            ew, coef = (inds, mat), (float(self.coefs[-1]) if self.n_external else 1.0)
        tab = capi.TableSet.from_synth(cell, self.ce_coefs, feature_mode=feature_mode, ewald=ew,
                                       ewald_coef=1.0 if coef is None else coef, mu_table=mu_table,
                                       ewald_charges=q, **usher)
        tab.supercell = cell
        return tab

    def ewald_tables(self, cell):
        """Ewald index table, matrix and charges of a supercell with the parameters of the
        model's EwaldTerm (cofe/extern/ewald.py:30-58; defaults when the model has none)."""
        t = self.subspace.ewald_term or {}
        return cell.ewald_tables(t.get("eta"), t.get("real_space_cut"), t.get("recip_space_cut"),
                                 use_term=t.get("use_term", "total"))  # EwaldTerm.use_term, ewald.py:44-58


def _read_json(path_or_dict):
    if isinstance(path_or_dict, dict):
        return path_or_dict
    import gzip

    opener = gzip.open if str(path_or_dict).endswith(".gz") else open
    with opener(path_or_dict, "rt") as fh:
        return json.load(fh)


def load_mson(path_or_dict):
    """Load a ClusterExpansion (or a notebook ``save_work`` file that contains one)."""
    return MsonClusterExpansion(_read_json(path_or_dict))


def wrangler_entries(path_or_dict):
    """The training structures a StructureWrangler stored (wrangler.py:744-785): per entry the
    supercell matrix, the species name of every site of the entry's structure with the site
    mapping of those sites onto the supercell (the arguments ``occupancy_from_structure`` was
    called with), the refined structure (perfect-lattice copy: species + fractional coordinates in
    the supercell basis, vacancies dropped), the stored correlation vector (incl. external terms)
    and the size in prims."""
    d = _read_json(path_or_dict)
    w = d.get("StructureWrangler", d)

    def arr(x):
        return np.asarray(x["data"] if isinstance(x, dict) else x)

    def names_of(sites):
        return [Species(s["species"][0]["element"], s["species"][0].get("oxidation_state")).name
                for s in sites]

    out = []
    for e in w["_entries"]:
        data = e["data"]
        ref = data["refined_structure"]
        out.append(dict(supercell_matrix=arr(data["supercell_matrix"]).astype(np.int64),
                        species=names_of(e["structure"]["sites"]),
                        site_mapping=[int(x) for x in data["site_mapping"]],
                        refined_species=names_of(ref["sites"]),
                        refined_frac_coords=np.array([s["abc"] for s in ref["sites"]], dtype=np.float64),
                        refined_lattice=np.asarray(ref["lattice"]["matrix"], dtype=np.float64),
                        correlations=arr(data["correlations"]).astype(np.float64),
                        size=int(data["size"]), energy=float(e["energy"])))
    return out
