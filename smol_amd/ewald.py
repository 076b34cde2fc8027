"""Own Ewald-summation matrix + index table for the electrostatic term (host setup).

The reference takes the matrix *values* from a third-party package
(pymatgen.analysis.ewald.EwaldSummation, pinned pymatgen==2025.01.09; call sites
smol/moca/processor/ewald.py:83-99 and smol/cofe/extern/ewald.py:152-177) which is
not available here, so absolute values are "parity unpinned" (SURVEY.md §8c).  What
*is* pinned is everything the hot path does with the matrix: the index table layout
(smol/cofe/extern/ewald.py:84-97) and the single-flip delta
(smol/utils/cluster/ewald.pyx:9-59), because oracle, reference core and HIP engine
are all fed the identical matrix produced here.

Matrix convention (same as the reference's use of it): E = sum_{a,b occupied} M[a,b]
(smol/moca/processor/ewald.py:141-145), M symmetric, self/point terms on the
diagonal.  Known-answer check: rocksalt Madelung constant 1.747565.
"""

from __future__ import annotations

import itertools

import numpy as np
from scipy.special import erfc

CONV_FACT = 14.399645478425668  # e^2 / (4 pi eps0) in eV * Angstrom


def ewald_indices(nspecies_per_site, vacancy_codes=None):
    """Index table int32[N, max_species]: running counter over (site, species),
    -1 where a site has no such species or the species is a vacancy.

    Restates smol/cofe/extern/ewald.py:84-97.  ``vacancy_codes`` maps site ->
    set of species codes that are vacancies (skipped, -1).
    """
    nsp = np.asarray(nspecies_per_site, dtype=np.int64)
    width = int(nsp.max())
    inds = -np.ones((len(nsp), width), dtype=np.int32)
    c = 0
    for s, S in enumerate(nsp):
        vac = () if vacancy_codes is None else vacancy_codes.get(s, ())
        for code in range(S):
            if code in vac:
                continue
            inds[s, code] = c
            c += 1
    return np.ascontiguousarray(inds), c


def _geometric_kernel(lattice, frac, eta=None, acc=12.0):
    """g[i,j] such that E = sum_ij q_i q_j g[i,j] (incl. i==j self-image terms,
    excluding the -sqrt(eta/pi) point term which is returned separately)."""
    lattice = np.asarray(lattice, float)
    frac = np.asarray(frac, float)
    n = len(frac)
    vol = abs(np.linalg.det(lattice))
    if eta is None:
        eta = (n * 0.01 / vol) ** (1 / 3) * np.pi  # same heuristic family as common codes
        eta = max(eta, 0.05)
    sq = np.sqrt(eta)
    rcut = np.sqrt(acc / eta) if acc else 6.0 / sq
    gcut = 2 * sq * np.sqrt(acc) if acc else 12.0 * sq
    cart = frac @ lattice
    recip = 2 * np.pi * np.linalg.inv(lattice).T  # rows = reciprocal vectors
    # real space
    lens = np.linalg.norm(lattice, axis=1)
    heights = vol / np.array(
        [
            np.linalg.norm(np.cross(lattice[1], lattice[2])),
            np.linalg.norm(np.cross(lattice[2], lattice[0])),
            np.linalg.norm(np.cross(lattice[0], lattice[1])),
        ]
    )
    nmax = np.ceil(rcut / heights).astype(int) + 1
    shifts = np.array(
        list(itertools.product(*[range(-m, m + 1) for m in nmax])), dtype=float
    ) @ lattice
    d0 = cart[None, :, :] - cart[:, None, :]  # (n,n,3) r_j - r_i
    g_real = np.zeros((n, n))
    for chunk in np.array_split(shifts, max(1, len(shifts) // 64)):
        d = d0[:, :, None, :] + chunk[None, None, :, :]
        r = np.linalg.norm(d, axis=-1)
        mask = (r > 1e-8) & (r <= rcut)
        with np.errstate(divide="ignore", invalid="ignore"):
            term = np.where(mask, erfc(sq * r) / r, 0.0)
        g_real += term.sum(axis=-1)
    g_real *= 0.5
    # reciprocal space
    rheights = 2 * np.pi / lens  # conservative bound on reciprocal spacing
    rl = np.linalg.norm(recip, axis=1)
    rvol = abs(np.linalg.det(recip))
    rh = rvol / np.array(
        [
            np.linalg.norm(np.cross(recip[1], recip[2])),
            np.linalg.norm(np.cross(recip[2], recip[0])),
            np.linalg.norm(np.cross(recip[0], recip[1])),
        ]
    )
    gmax = np.ceil(gcut / rh).astype(int) + 1
    hkl = np.array(list(itertools.product(*[range(-m, m + 1) for m in gmax])), dtype=float)
    hkl = hkl[np.any(hkl != 0, axis=1)]
    gv = hkl @ recip
    g2 = np.sum(gv * gv, axis=1)
    keep = g2 <= gcut * gcut
    gv, g2 = gv[keep], g2[keep]
    w = np.exp(-g2 / (4 * eta)) / g2
    phase = cart @ gv.T  # (n, ng)
    cs, sn = np.cos(phase), np.sin(phase)
    g_recip = (cs * w) @ cs.T + (sn * w) @ sn.T
    g_recip *= 2 * np.pi / vol
    point = -np.sqrt(eta / np.pi)
    del rheights, rl
    return g_real, g_recip, point, eta


def ewald_matrix(lattice, frac, charges, eta=None, acc=12.0):
    """Dense Ewald matrix for point charges ``charges`` at ``frac`` (eV)."""
    g_real, g_recip, point, _ = _geometric_kernel(lattice, frac, eta, acc)
    q = np.asarray(charges, float)
    m = (g_real + g_recip) * np.outer(q, q)
    m[np.diag_indices_from(m)] += point * q * q
    m *= CONV_FACT
    return np.ascontiguousarray(0.5 * (m + m.T))


def supercell_ewald(sc, eta=None, acc=12.0, use_term="total"):
    """(ewald_inds int32[N,Smax], matrix float64[M,M]) for a SupercellTables; ``use_term`` as in
    EwaldTerm (cofe/extern/ewald.py:159-177): the total matrix or its real-space / reciprocal-space /
    point part.

    Exploits translation invariance: the geometric kernel is evaluated for the
    first lattice point of each basis site against all sites (nb x N block), then
    expanded to the dense (M, M) matrix the reference layout requires.
    """
    prim = sc.model.prim
    P, N = sc.size, sc.num_sites
    nsp = np.array([prim.nspecies[b] for b in sc.site_b])
    vac = {}
    for s in range(N):
        v = {c for c, q in enumerate(prim.charges[sc.site_b[s]]) if q is None}
        if v:
            vac[s] = v
    inds, M = ewald_indices(nsp, vac)
    # supercell lattice and site fractional coords (in supercell basis)
    sc_lat = sc.scmatrix.astype(float) @ prim.lattice
    inv = np.linalg.inv(sc.scmatrix.astype(float))
    frac_prim = prim.frac_coords[sc.site_b] + sc.lattice_points[sc.site_t]
    frac_sc = frac_prim @ inv
    # kernel rows for the representative site of every basis index
    g_rows = np.zeros((prim.nb, N))
    point = 0.0
    for b in range(prim.nb):
        # place representative first; kernel of (rep, all sites)
        rep = b * P
        g_real, g_recip, point, eta = _pair_rows(sc_lat, frac_sc, rep, eta, acc)
        g_rows[b], with_point = _select_term(g_real, g_recip, use_term)
    # expand by translation: g[s1, s2] = g_rows[b1][ b2*P + idx(t2 - t1) ]
    pts = sc.lattice_points
    g_full = np.empty((N, N))
    for t1 in range(P):
        rel = sc._point_index(pts - pts[t1][None, :])  # index of (t2 - t1)
        for b1 in range(prim.nb):
            row = g_rows[b1].reshape(prim.nb, P)[:, rel].reshape(-1)
            g_full[b1 * P + t1] = row
    g_full = 0.5 * (g_full + g_full.T)
    # species-resolved matrix
    qs = np.zeros(M)
    site_of = np.zeros(M, dtype=np.int64)
    for s in range(N):
        for code in range(nsp[s]):
            k = inds[s, code]
            if k >= 0:
                qs[k] = prim.charges[sc.site_b[s]][code]
                site_of[k] = s
    mat = g_full[np.ix_(site_of, site_of)]
    del g_full
    for a0 in range(0, M, 512):  # mat[a, b] *= q_a q_b in row blocks: no second M x M array (ewald_matrix_pmg)
        mat[a0:a0 + 512] *= np.outer(qs[a0:a0 + 512], qs)
    # two species on the same site never coexist; keep kernel value (as the
    # reference's overlapping-site structure would) but the self term only on diag
    if with_point:
        mat[np.diag_indices_from(mat)] += point * qs * qs
    mat *= CONV_FACT
    return inds, np.ascontiguousarray(mat)


def _pair_rows(lattice, frac, rep, eta, acc, acc_g=None):
    """Kernel row of site ``rep`` against all sites (real, recip, point, eta); ``acc_g`` is the
    accuracy exponent of the reciprocal-space cutoff when it differs from the real-space one."""
    lattice = np.asarray(lattice, float)
    n = len(frac)
    vol = abs(np.linalg.det(lattice))
    if eta is None:
        eta = max((n * 0.01 / vol) ** (1 / 3) * np.pi, 0.05)
    sq = np.sqrt(eta)
    rcut = np.sqrt(acc / eta)
    gcut = 2 * sq * np.sqrt(acc if acc_g is None else acc_g)
    cart = frac @ lattice
    heights = vol / np.array(
        [
            np.linalg.norm(np.cross(lattice[1], lattice[2])),
            np.linalg.norm(np.cross(lattice[2], lattice[0])),
            np.linalg.norm(np.cross(lattice[0], lattice[1])),
        ]
    )
    nmax = np.ceil(rcut / heights).astype(int) + 1
    shifts = np.array(
        list(itertools.product(*[range(-m, m + 1) for m in nmax])), dtype=float
    ) @ lattice
    d0 = cart - cart[rep]
    g_real = np.zeros(n)
    for chunk in np.array_split(shifts, max(1, len(shifts) * n // 2_000_000 + 1)):
        r = np.linalg.norm(d0[:, None, :] + chunk[None, :, :], axis=-1)
        mask = (r > 1e-8) & (r <= rcut)
        with np.errstate(divide="ignore", invalid="ignore"):
            g_real += np.where(mask, erfc(sq * r) / r, 0.0).sum(axis=-1)
    g_real *= 0.5
    recip = 2 * np.pi * np.linalg.inv(lattice).T
    rvol = abs(np.linalg.det(recip))
    rh = rvol / np.array(
        [
            np.linalg.norm(np.cross(recip[1], recip[2])),
            np.linalg.norm(np.cross(recip[2], recip[0])),
            np.linalg.norm(np.cross(recip[0], recip[1])),
        ]
    )
    gmax = np.ceil(gcut / rh).astype(int) + 1
    hkl = np.array(list(itertools.product(*[range(-m, m + 1) for m in gmax])), dtype=float)
    hkl = hkl[np.any(hkl != 0, axis=1)]
    gv = hkl @ recip
    g2 = np.sum(gv * gv, axis=1)
    keep = g2 <= gcut * gcut
    gv, g2 = gv[keep], g2[keep]
    w = np.exp(-g2 / (4 * eta)) / g2
    g_recip = np.zeros(n)
    for gc, wc in zip(
        np.array_split(gv, max(1, len(gv) * n // 4_000_000 + 1)),
        np.array_split(w, max(1, len(gv) * n // 4_000_000 + 1)),
    ):
        phase = d0 @ gc.T
        g_recip += np.cos(phase) @ wc
    g_recip *= 2 * np.pi / vol
    return g_real, g_recip, -np.sqrt(eta / np.pi), eta


# --------------------------------------------------------------------------------------
# pymatgen-convention matrix for imported models (smol_amd.mson)
# --------------------------------------------------------------------------------------
PMG_ACC = float(np.log(10.0 ** 12))  # EwaldSummation(acc_factor=12): terms below 1e-12 dropped
PMG_W = 1.0 / np.sqrt(2.0)  # EwaldSummation(w=1/sqrt(2)): real / reciprocal work balance


def pmg_eta(n_sites, volume):
    """Screening parameter pymatgen's EwaldSummation picks when none is given:
    (n w / V^2)^(1/3) pi with n the number of sites of the structure it is handed -- for smol
    that is the "Ewald structure" carrying every allowed species (cofe/extern/ewald.py:64-100)."""
    return (n_sites * PMG_W / volume ** 2) ** (1.0 / 3.0) * np.pi


USE_TERMS = ("total", "real", "reciprocal", "point")  # EwaldTerm.ewald_term_options (cofe/extern/ewald.py:28)


def _select_term(g_real, g_recip, use_term):
    """Geometric kernel of the chosen term and whether the point (self) term joins the diagonal:
    EwaldTerm.get_ewald_matrix (cofe/extern/ewald.py:159-177) picks pymatgen's total / real-space /
    reciprocal-space matrix or the diagonal matrix of the point energies."""
    if use_term not in USE_TERMS:
        raise AttributeError(f"Provided use_term {use_term} is not a valid option. Please choose one of "
                             f"{USE_TERMS}.")  # the reference's message, ewald.py:52-56
    if use_term == "total":
        return g_real + g_recip, True
    if use_term == "real":
        return g_real, False
    if use_term == "reciprocal":
        return g_recip, False
    return np.zeros_like(g_real), True


def ewald_matrix_pmg(lattice, frac, site_of, charges, eta=None, real_space_cut=None,
                     recip_space_cut=None, translation_index=None, use_term="total"):
    """Total Ewald matrix (real + reciprocal, point terms on the diagonal) over M point charges
    ``charges[k]`` sitting on site ``site_of[k]`` of a periodic cell (several charges may share a
    site: the allowed species of a disordered site; their mutual entry is the site's own image
    sum, they never coexist).  Conventions of pymatgen.analysis.ewald.EwaldSummation as smol calls
    it (moca/processor/ewald.py:83-99): eta from pmg_eta(M, V), cutoffs accf / sqrt(eta) and
    2 sqrt(eta) accf with accf^2 = ln 1e12, E = sum_ab M[a, b] in eV.  The formulas are the
    textbook Ewald sum (this module's own implementation); what is taken from pymatgen is only
    the choice of eta and cutoffs, so that individual matrix entries -- which depend on eta
    through the neutralising background -- are comparable, not just charge-neutral totals.

    ``translation_index`` (optional, int[P, P]): for supercells whose sites are ordered (prim site
    major, lattice translation minor), translation_index[t1][t2] = index of the lattice point
    t2 - t1; then only one kernel row per prim site is summed and the rest follows by translation
    invariance (O(nb N) instead of O(N^2) lattice sums)."""
    lattice = np.asarray(lattice, float)
    frac = np.asarray(frac, float)
    site_of = np.asarray(site_of, dtype=np.int64)
    q = np.asarray(charges, float)
    vol = abs(np.linalg.det(lattice))
    if eta is None:
        eta = pmg_eta(len(q), vol)
    acc_r = PMG_ACC if real_space_cut is None else eta * real_space_cut ** 2
    acc_g = PMG_ACC if recip_space_cut is None else recip_space_cut ** 2 / (4.0 * eta)
    n = len(frac)
    g = np.empty((n, n))
    point = -np.sqrt(eta / np.pi)
    if translation_index is None:
        for i in range(n):
            g_real, g_recip, point, _ = _pair_rows(lattice, frac, i, eta, acc_r, acc_g)
            g[i], with_point = _select_term(g_real, g_recip, use_term)
    else:
        P = len(translation_index)
        nb = n // P
        for b1 in range(nb):
            g_real, g_recip, point, _ = _pair_rows(lattice, frac, b1 * P, eta, acc_r, acc_g)
            row, with_point = _select_term(g_real, g_recip, use_term)
            row = row.reshape(nb, P)
            for t1 in range(P):  # site (b1, t1) sees (b2, t2) like (b1, 0) sees (b2, t2 - t1)
                g[b1 * P + t1] = row[:, translation_index[t1]].reshape(-1)
    g = 0.5 * (g + g.T)
    mat = g[np.ix_(site_of, site_of)]
    del g
    # mat[a, b] *= q_a q_b in row blocks: the same products as `* np.outer(q, q)` without a second M x M array
    # (M = 10 368 for the reference's LiNiO2 model in a 12^3 cell: 860 MB each)
    for a0 in range(0, len(q), 512):
        mat[a0:a0 + 512] *= np.outer(q[a0:a0 + 512], q)
    if with_point:
        mat[np.diag_indices_from(mat)] += point * q * q
    mat *= CONV_FACT
    return np.ascontiguousarray(mat)
