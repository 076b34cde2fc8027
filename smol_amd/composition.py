"""Composition space of a multi-sublattice lattice model and the flip tables built from it
(host setup for the TableFlip step type).

What it mirrors (behaviour, attribute names and error cases; the code is this repo's own):

  * ``CompositionSpace``      smol/moca/composition/space.py:73-429 -- the integer lattice
    of species counts ``n`` obeying ``A n = b * supercell_size`` (charge neutrality, one
    site-count row per sublattice, extra equalities), its vertices / minimal supercell,
    a base solution, lattice basis, grid enumeration and the flip table;
  * integer linear algebra    smol/utils/math.py:149-294 (Smith-form based solver),
    :396-564 (natural-number solutions), :567-829 (flip size, connectivity, greedy basis
    optimisation, ergodic completion), :832-867 (feasibility mask of table directions);
  * TableFlip a-priori factor smol/moca/kernel/mcusher.py:660-711 for any number of
    sublattices (the device kernel implements the single-active-sublattice case; this
    host version is what its table and weights are prepared and cross-checked with).

The reference delegates the linear programs to cvxpy and the vertex enumeration to the
``polytope`` package; neither exists here, so vertices are enumerated as basic feasible
solutions in exact rational arithmetic and the LP bounds come from scipy's HiGHS.

Species are plain ``(name, oxidation_state)`` pairs (``Species`` below); a vacancy has
charge 0.  Pinned by the reference's own known answers (tests/test_composition.py):
tests/test_moca/test_comp_space.py:248-271 (bases / flip tables of four systems) and
tests/test_moca/test_mcushers.py:199-234 (a-priori factors).
"""

from __future__ import annotations

import itertools
import math
from collections import namedtuple
from fractions import Fraction

import numpy as np

NUM_TOL = 1e-6  # smol/utils/math.py:28

Species = namedtuple("Species", ["name", "oxi_state"])


def as_species(sp):
    """('Li+', 1) / Species / 'Li+' (charge parsed from the trailing sign) -> Species."""
    if isinstance(sp, Species):
        return sp
    if isinstance(sp, (tuple, list)) and len(sp) == 2:
        return Species(str(sp[0]), int(round(float(sp[1] or 0))))
    s = str(sp)
    if s.lower() in ("vacancy", "vac", "va"):
        return Species("Vacancy", 0)
    body = s.rstrip("+-")
    tail = s[len(body):]
    if not tail:
        return Species(s, 0)
    digits = "".join(ch for ch in body[::-1] if ch.isdigit())  # digits just before the sign
    k = 0
    while k < len(body) and body[len(body) - 1 - k].isdigit():
        k += 1
    mag = int(body[len(body) - k:]) if k else len(tail)
    del digits
    return Species(s, mag if tail[0] == "+" else -mag)


def get_dim_ids_by_sublattice(site_spaces):
    """Index of every (sublattice, species) pair in the "counts" vector
    (smol/moca/occu_utils.py:8-24)."""
    out, k = [], 0
    for sp in site_spaces:
        out.append(list(range(k, k + len(sp))))
        k += len(sp)
    return out


# --------------------------------------------------------------------------------------
# integer linear algebra
# --------------------------------------------------------------------------------------
def _column_echelon(A):
    """Unimodular V (python ints) with A V = [H | 0], H lower-trapezoidal with one pivot per
    independent row: integer column operations only (Euclid on the pivot row)."""
    A = [[int(x) for x in row] for row in np.asarray(A)]
    m, d = len(A), len(A[0])
    H = [row[:] for row in A]
    V = [[int(i == j) for j in range(d)] for i in range(d)]

    def colop(j0, j1, a, b, c, e):  # (col j0, col j1) <- (a j0 + b j1, c j0 + e j1)
        for M in (H, V):
            for row in M:
                x, y = row[j0], row[j1]
                row[j0], row[j1] = a * x + b * y, c * x + e * y

    piv_rows, col = [], 0
    for i in range(m):
        if col >= d:
            break
        # gcd of row i over the free columns col.., accumulated into column `col`
        for j in range(col + 1, d):
            while H[i][j] != 0:
                if H[i][col] == 0 or abs(H[i][j]) < abs(H[i][col]):
                    colop(col, j, 0, 1, 1, 0)  # swap
                    if H[i][j] == 0:
                        break
                q = H[i][j] // H[i][col]
                colop(col, j, 1, 0, -q, 1)  # col j -= q * col
        if H[i][col] != 0:
            if H[i][col] < 0:
                for M in (H, V):
                    for row in M:
                        row[col] = -row[col]
            piv_rows.append(i)
            col += 1
    return H, V, piv_rows


def solve_diophantines(A, b=None):
    """Integer solutions of ``A n = b``: returns ``(n0, basis)`` with the rows of ``basis``
    spanning the integer null lattice of A (same contract as smol/utils/math.py:243-294;
    the particular n0 / basis may differ from the reference's Smith-form ones -- any
    lattice basis is admissible, and the optimised tables are compared as sets).
    Raises ValueError when no integer solution exists."""
    A = np.asarray(A).astype(object)
    m, d = A.shape
    b = [0] * m if b is None else [int(round(float(x))) for x in b]
    H, V, piv_rows = _column_echelon(A)
    r = len(piv_rows)
    y = [0] * r
    # forward substitution on the pivot rows, then consistency of the dependent rows
    for k, i in enumerate(piv_rows):
        acc = b[i] - sum(H[i][c] * y[c] for c in range(k))
        if acc % H[i][k] != 0:
            raise ValueError("Diophantine equations A n = b are not feasible!")
        y[k] = acc // H[i][k]
    for i in range(m):
        if sum(H[i][c] * y[c] for c in range(r)) != b[i]:
            raise ValueError("Diophantine equations A n = b are not feasible!")
    n0 = np.array([sum(V[i][c] * y[c] for c in range(r)) for i in range(d)], dtype=np.int64)
    basis = np.array([[V[i][c] for i in range(d)] for c in range(r, d)], dtype=np.int64).reshape(d - r, d)
    return n0, basis


def _rationalize(x, max_denominator=1000, dtol=NUM_TOL):
    f = Fraction(float(x)).limit_denominator(max_denominator)
    if abs(float(f) - float(x)) > dtol:
        raise ValueError(f"Can't find a rational number near {x} within tolerance!")
    return f


def integerize_vector(v, max_denominator=1000, dtol=NUM_TOL):
    """(integer vector, multiplier): smallest common multiple clearing the denominators
    (smol/utils/math.py:89-116)."""
    fr = [_rationalize(x, max_denominator, dtol) for x in np.asarray(v, dtype=float).ravel()]
    mul = 1
    for f in fr:
        mul = mul * f.denominator // math.gcd(mul, f.denominator)
    return np.array([int(f * mul) for f in fr], dtype=np.int64).reshape(np.shape(v)), int(mul)


def integerize_multiple(vs, max_denominator=1000, dtol=NUM_TOL):
    return integerize_vector(vs, max_denominator, dtol)


def _solve_exact(M, rhs):
    """Solve a square rational system by Gauss-Jordan; None if singular."""
    n = len(M)
    a = [[Fraction(x) for x in row] + [Fraction(r)] for row, r in zip(M, rhs)]
    for c in range(n):
        p = next((i for i in range(c, n) if a[i][c] != 0), None)
        if p is None:
            return None
        a[c], a[p] = a[p], a[c]
        inv = 1 / a[c][c]
        a[c] = [x * inv for x in a[c]]
        for i in range(n):
            if i != c and a[i][c] != 0:
                f = a[i][c]
                a[i] = [x - f * y for x, y in zip(a[i], a[c])]
    return [a[i][n] for i in range(n)]


def get_nonneg_float_vertices(A, b):
    """Vertices of ``{n >= 0 : A n = b}`` (smol/utils/math.py:297-336): the basic feasible
    solutions, enumerated exactly -- choose rank(A) independent columns, solve, keep the
    non-negative ones."""
    A = np.asarray(A)
    m, d = A.shape
    Af = [[Fraction(int(x)) for x in row] for row in A]
    bf = [_rationalize(x) for x in b]
    H, _, piv_rows = _column_echelon(A)
    r = len(piv_rows)
    rows = piv_rows  # an independent subset of the equations
    verts = set()
    for cols in itertools.combinations(range(d), r):
        sol = _solve_exact([[Af[i][c] for c in cols] for i in rows], [bf[i] for i in rows])
        if sol is None or any(x < 0 for x in sol):
            continue
        full = [Fraction(0)] * d
        for c, x in zip(cols, sol):
            full[c] = x
        if all(sum(Af[i][c] * full[c] for c in range(d)) == bf[i] for i in range(m)):
            verts.add(tuple(full))
    if not verts:
        raise ValueError("Provided equation An=b is not feasible under n>=0.")
    return np.array(sorted(verts), dtype=object)


def _lp_extremes_first_dim(G, h):
    """min / max of x0 over {x : G x <= h} (smol/utils/math.py:451-481; HiGHS instead of
    cvxpy)."""
    from scipy.optimize import linprog

    d = G.shape[1]
    c = np.zeros(d)
    c[0] = 1.0
    out = []
    for sign in (1.0, -1.0):
        res = linprog(sign * c, A_ub=G, b_ub=h, bounds=[(None, None)] * d, method="highs")
        if res.status == 3:
            raise ValueError("Inequalities are not bounded!")
        if res.status != 0:
            return None
        out.append(sign * res.fun)
    return out[0], out[1]


def _snap(x, tol):
    return round(x) if abs(x - round(x)) <= tol else x


def get_natural_solutions(n0, vs, integer_tol=NUM_TOL, step=1):
    """All integer x with ``n0 + x @ vs >= 0`` (smol/utils/math.py:484-564): branch on the
    first coordinate over its LP range (every ``step``-th value), recurse on the rest."""
    n0 = np.asarray(n0, dtype=np.int64)
    vs = np.asarray(vs, dtype=np.int64)
    n = vs.shape[0]
    if n == 1:
        v = vs[0]
        lo, hi = -np.inf, np.inf
        for a, c in zip(n0, v):
            if c > 0:
                lo = max(lo, -a / c)
            elif c < 0:
                hi = min(hi, -a / c)
            elif a < 0:
                return np.zeros((0, 1), dtype=np.int64)
        if not (np.isfinite(lo) and np.isfinite(hi)):
            raise ValueError("Inequalities are not bounded!")
        lo, hi = math.ceil(_snap(lo, integer_tol)), math.floor(_snap(hi, integer_tol))
        return np.arange(lo, hi + 1, step, dtype=np.int64).reshape(-1, 1)
    ext = _lp_extremes_first_dim(-vs.T.astype(float), n0.astype(float))
    if ext is None:
        return np.zeros((0, n), dtype=np.int64)
    lo, hi = math.ceil(_snap(ext[0], integer_tol)), math.floor(_snap(ext[1], integer_tol))
    blocks = []
    for m in range(lo, hi + 1, step):
        sub = get_natural_solutions(n0 + m * vs[0], vs[1:], integer_tol, step)
        if len(sub):
            blocks.append(np.hstack([np.full((len(sub), 1), m, dtype=np.int64), sub]))
    return np.vstack(blocks) if blocks else np.zeros((0, n), dtype=np.int64)


def flip_size(u):
    """Number of sites a direction changes (smol/utils/math.py:567-589)."""
    u = np.asarray(u, dtype=np.int64)
    if u.sum() != 0:
        raise ValueError(f"Flip vector {u} does not conserve number of sites!")
    return int(u[u > 0].sum())


def connectivity(u, ns):
    """Number of grid-point pairs joined by +-u (smol/utils/math.py:609-636)."""
    pts = {tuple(r) for r in np.asarray(ns, dtype=np.int64).tolist()}
    u = np.asarray(u, dtype=np.int64)
    return sum(1 for p in pts if tuple((np.array(p) + u).tolist()) in pts)


def _rank(M):
    return int(np.linalg.matrix_rank(np.asarray(M, dtype=float))) if len(M) else 0


def get_optimal_basis(n0, vs, xs, max_loops=100):
    """Greedy basis optimisation (smol/utils/math.py:659-747): among the current vectors
    and all pairwise sums / differences keep, in order of (flip size ascending,
    connectivity descending), the first n independent ones; repeat until stable."""
    vs = np.asarray(vs, dtype=np.int64)
    ns = np.asarray(xs, dtype=np.int64) @ vs + np.asarray(n0, dtype=np.int64)
    n = len(vs)

    def canon(V):  # first column non-negative, as a set of rows
        return {tuple((r if r[0] >= 0 else -r).tolist()) for r in V}

    cur = vs.copy()
    for _ in range(max_loops):
        cand = [r for r in cur]
        for i, j in itertools.combinations(range(n), 2):
            cand += [cur[i] + cur[j], cur[i] - cur[j]]
        cand.sort(key=lambda u: (flip_size(u), -connectivity(u, ns)))  # stable
        new = []
        for u in cand:
            if len(new) == n:
                break
            if _rank(new + [u]) == len(new) + 1:
                new.append(u)
        new = np.array(new, dtype=np.int64)
        if new.shape == cur.shape and canon(new) == canon(cur):
            break
        cur = new
    return cur


def get_ergodic_vectors(n0, vs, xs, k=3):
    """Add directions until every grid point has a neighbour (smol/utils/math.py:750-829):
    candidates are the offsets to the k nearest grid points of each isolated point, tried in
    order of flip size."""
    from scipy.spatial import KDTree

    vs = np.asarray(vs, dtype=np.int64)
    ns = np.asarray(xs, dtype=np.int64) @ vs + np.asarray(n0, dtype=np.int64)
    pts = {tuple(r) for r in ns.tolist()}

    def isolated(table, points):
        steps = np.vstack([table, -table])
        return np.array([not any(tuple((p + s).tolist()) in pts for s in steps) for p in points], dtype=bool)

    lonely = ns[isolated(vs, ns)] if len(ns) else ns
    if len(lonely) == 0:
        return vs
    tree = KDTree(ns)
    cands = []
    for p in lonely:
        dist, ids = tree.query(p, k=min(k, len(ns)))
        dist, ids = np.atleast_1d(dist), np.atleast_1d(ids)
        if dist[0] == 0:
            ids = ids[1:]
        for q in ns[ids]:
            u = tuple((q - p).tolist())
            if u in cands or tuple((-np.array(u)).tolist()) in cands:
                continue
            cands.append(u)
    cands.sort(key=flip_size)
    table, rem = vs.copy(), lonely.copy()
    for u in cands:
        table = np.vstack([table, np.array(u, dtype=np.int64)])
        rem = rem[isolated(table, rem)]
        if len(rem) == 0:
            break
    return table


def flip_weights_mask(flip_vectors, n, max_n=None):
    """Feasibility of every table direction and its inverse at counts ``n``
    (smol/utils/math.py:832-867): interleaved (u0, -u0, u1, -u1, ...)."""
    fv = np.asarray(flip_vectors, dtype=np.int64)
    dirs = np.stack([fv, -fv], axis=1).reshape(-1, fv.shape[1])
    n = np.asarray(n, dtype=np.int64)
    hi = np.full(len(n), np.inf) if max_n is None else np.broadcast_to(np.asarray(max_n), n.shape)
    return ~(np.any(dirs + n < 0, axis=-1) | np.any(dirs + n > hi, axis=-1))


def table_log_priori_factor(flip_table, flip_weights, swap_weight, n_now, u, max_n=None):
    """log of the a-priori ratio of TableFlip for the table direction ``u`` taken at counts
    ``n_now`` (smol/moca/kernel/mcusher.py:660-711): log(p_next / p_now) plus
    sum_dims ln n_now! - ln n_next!.  A zero ``u`` (canonical swap) gives 0; a ``u`` that is
    no table row raises ValueError."""
    from scipy.special import gammaln

    table = np.asarray(flip_table, dtype=np.int64)
    u = np.asarray(u, dtype=np.int64)
    if not u.any():
        return 0.0
    fid = direction = None
    for i, v in enumerate(table):
        if np.array_equal(v, u):
            fid, direction = i, 0
            break
        if np.array_equal(-v, u):
            fid, direction = i, 1
            break
    if fid is None:
        raise ValueError(f"Step with count change {u.tolist()} is not in flip table.")
    w = np.asarray(flip_weights, dtype=float)
    if len(w) == len(table):
        w = np.repeat(w, 2)
    n_now = np.asarray(n_now, dtype=np.int64)
    w_now = w * flip_weights_mask(table, n_now, max_n)
    p_now = (1 - swap_weight) * w_now[2 * fid + direction] / w_now.sum()
    n_next = n_now + u
    w_next = w * flip_weights_mask(table, n_next, max_n)
    p_next = (1 - swap_weight) * w_next[2 * fid + (1 - direction)] / w_next.sum()
    out = math.log(p_next / p_now)
    for d in np.flatnonzero(u):
        out += gammaln(n_now[d] + 1) - gammaln(n_next[d] + 1)
    return float(out)


# --------------------------------------------------------------------------------------
class CompositionSpace:
    """Integer composition lattice under charge neutrality / extra constraints
    (smol/moca/composition/space.py:73-429).

    ``site_spaces``: per sublattice, the allowed species as ``(name, oxidation_state)``.
    ``other_constraints``: tuples ``(a, b, rel)`` with ``rel`` in ``"eq" "==" "=" "<=" "leq"
    ">=" "geq"``, read per primitive cell exactly as in the reference (string equations are
    not parsed here)."""

    def __init__(self, site_spaces, sublattice_sizes=None, charge_neutral=True, other_constraints=None,
                 optimize_basis=False, table_ergodic=False):
        self.site_spaces = [[as_species(sp) for sp in sl] for sl in site_spaces]
        self.num_dims = sum(len(sl) for sl in self.site_spaces)
        self.dim_ids = get_dim_ids_by_sublattice(self.site_spaces)
        uniq = []
        for sp in itertools.chain(*self.site_spaces):
            if sp not in uniq:
                uniq.append(sp)
        self.species = sorted(uniq)
        self.species_ids = [[self.species.index(sp) for sp in sl] for sl in self.site_spaces]
        if sublattice_sizes is None:
            self.sublattice_sizes = [1] * len(self.site_spaces)
        elif len(sublattice_sizes) == len(self.site_spaces):
            self.sublattice_sizes = [int(x) for x in sublattice_sizes]
        else:
            raise ValueError("Sub-lattice number is not the same in parameters bits and sublattice_sizes.")
        self.charge_neutral = charge_neutral
        self.optimize_basis = optimize_basis
        self.table_ergodic = table_ergodic
        self.other_constraints = other_constraints
        eqs, leqs = [], []
        for a, bb, rel in (other_constraints or []):
            a = np.asarray(a, dtype=float)
            if len(a) != self.num_dims:
                raise ValueError(f"Constraint length: {len(a)} does not match dimensions: {self.num_dims}!")
            if rel in ("eq", "==", "="):
                eqs.append((a, float(bb)))
            elif rel in ("<=", "leq"):
                leqs.append((a, float(bb)))
            elif rel in (">=", "geq"):
                leqs.append((-a, -float(bb)))
            else:
                raise ValueError(f"Unknown relation {rel!r} in composition constraint.")
        self._other_eq_constraints, self._other_leq_constraints = eqs, leqs
        A, b = [], []
        if charge_neutral:
            A.append([sp.oxi_state for sl in self.site_spaces for sp in sl])
            b.append(0.0)
        for ids, size in zip(self.dim_ids, self.sublattice_sizes):
            row = [0] * self.num_dims
            for i in ids:
                row[i] = 1
            A.append(row)
            b.append(float(size))
        for a, bb in eqs:
            _, scale = integerize_vector(a)
            A.append(np.round(a * scale).astype(np.int64).tolist())
            b.append(bb * scale)
        self._A = np.array(A, dtype=np.int64)
        self._b = np.array(b, dtype=float)
        if _rank(self._A) >= self.num_dims:
            raise ValueError("Valid constraints more than number of dimensions!")
        self._A_leq = np.array([a for a, _ in leqs]) if leqs else None
        self._b_leq = np.array([bb for _, bb in leqs]) if leqs else None
        self._prim_vertices = self._min_supercell_size = self._flip_table = None
        self._n0 = self._vs = None
        self._comp_grids = {}

    # ------------------------------------------------------------------------------
    @property
    def prim_vertices(self):
        """Vertices of the per-primitive-cell composition polytope (leq constraints ignored),
        as floats; rows in "counts" format."""
        if self._prim_vertices is None:
            self._prim_vertices_exact = get_nonneg_float_vertices(self._A, self._b)
            self._prim_vertices = self._prim_vertices_exact.astype(float)
        return self._prim_vertices

    @property
    def min_supercell_size(self):
        """Smallest number of primitive cells that makes every vertex integral."""
        if self._min_supercell_size is None:
            _ = self.prim_vertices
            mul = 1
            for f in self._prim_vertices_exact.ravel():
                mul = mul * f.denominator // math.gcd(mul, f.denominator)
            self._min_supercell_size = int(mul)
        return self._min_supercell_size

    @property
    def num_unconstrained_compositions(self):
        out = 1
        for sl, size in zip(self.site_spaces, self.sublattice_sizes):
            out *= (size * self.min_supercell_size) ** len(sl)
        return int(out)

    def _min_feasible_size(self):
        return integerize_vector(self._b)[1]

    def get_supercell_base_solution(self, supercell_size=None):
        """One integer (not necessarily non-negative) solution at ``supercell_size``; scales
        linearly with the size so that grids at multiples coincide."""
        if supercell_size is None:
            supercell_size = self.min_supercell_size
        mf = self._min_feasible_size()
        if supercell_size % mf != 0:
            raise ValueError("Composition constraints can not have any integral solution in a "
                             f"super-cell of {supercell_size} prims!")
        if self._n0 is None:
            self._n0, _ = solve_diophantines(self._A, np.round(self._b * mf))
        return self._n0 * (supercell_size // mf)

    @property
    def basis(self):
        """Basis of the integer composition lattice (rows, "counts" format); optimised for
        small flips / high connectivity on the minimal supercell when ``optimize_basis``."""
        if self._vs is None:
            n0, vs = solve_diophantines(self._A, np.round(self._b * self.min_supercell_size))
            if self.optimize_basis and len(vs):
                xs = get_natural_solutions(n0, vs)
                vs = get_optimal_basis(n0, vs, xs)
                self._n0_min = n0
            self._vs = vs
        return self._vs

    @property
    def min_supercell_grid(self):
        return self.get_composition_grid(self.min_supercell_size)

    def get_composition_grid(self, supercell_size=1, step=1):
        """Integer compositions in lattice coordinates x (``n = n0 + x @ basis``) at the given
        supercell size, filtered by the leq constraints."""
        key = (int(supercell_size), int(step))
        if key not in self._comp_grids:
            n0 = self.get_supercell_base_solution(supercell_size)
            xs = get_natural_solutions(n0, self.basis, step=step)
            if self._A_leq is not None and len(xs):
                ns = xs @ self.basis + n0
                ok = np.all(self._A_leq @ ns.T <= self._b_leq[:, None] * supercell_size + NUM_TOL, axis=0)
                xs = xs[ok]
            self._comp_grids[key] = xs
        return self._comp_grids[key]

    @property
    def flip_table(self):
        """Flip vectors of TableFlip: the basis, completed to ergodicity on the minimal
        supercell when ``table_ergodic``."""
        if self._flip_table is None:
            if not self.table_ergodic or len(self.basis) == 0:
                self._flip_table = self.basis.copy()
            else:
                n0 = self.get_supercell_base_solution(self.min_supercell_size)
                self._flip_table = get_ergodic_vectors(n0, self.basis, self.min_supercell_grid)
        return self._flip_table

    @property
    def flip_reactions(self):
        return [flip_vec_to_reaction(u, self.site_spaces) for u in self.flip_table]

    # ------------------------------------------------------------------------------
    def sublattice_flip_table(self, sublattice_index):
        """Rows of the flip table restricted to one sublattice's species, for the device
        TableFlip kernel (single active sublattice).  Raises ValueError if a direction
        touches another sublattice."""
        ids = self.dim_ids[sublattice_index]
        other = [i for i in range(self.num_dims) if i not in ids]
        table = np.asarray(self.flip_table, dtype=np.int64)
        if table[:, other].any():
            raise ValueError("flip table couples several sublattices; the device TableFlip step "
                             "handles one active sublattice")
        return np.ascontiguousarray(table[:, ids].astype(np.int32))


def flip_vec_to_reaction(u, site_spaces):
    """'2 Mn3+(0) -> 1 Li+(0) + 1 Ti4+(0)' (smol/moca/composition/space.py:43-70)."""
    u = np.asarray(u, dtype=np.int64)
    lhs, rhs = [], []
    for sl_id, (sl, ids) in enumerate(zip(site_spaces, get_dim_ids_by_sublattice(site_spaces))):
        for sp, d in zip(sl, ids):
            name = as_species(sp).name
            if u[d] < 0:
                lhs.append(f"{-u[d]} {name}({sl_id})")
            elif u[d] > 0:
                rhs.append(f"{u[d]} {name}({sl_id})")
    return " + ".join(lhs) + " -> " + " + ".join(rhs)
