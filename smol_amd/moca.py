"""Host-side mirror of smol.moca's interface for the accelerated path.

Same names, argument meaning and error behaviour as the reference objects they stand
in for, so user scripts and parity tests read like smol's own:

    Sublattice                      smol/moca/sublattice.py:23
    ClusterExpansionProcessor       smol/moca/processor/expansion.py:39
    ClusterDecompositionProcessor   smol/moca/processor/expansion.py:243
    EwaldProcessor                  smol/moca/processor/ewald.py:26
    CompositeProcessor              smol/moca/processor/composite.py:18
    Ensemble                        smol/moca/ensemble.py:102
    Metropolis / WangLandau         smol/moca/kernel/metropolis.py:52, wanglandau.py:17
    Trace                           smol/moca/trace.py:8
    SampleContainer                 smol/moca/sampler/container.py:25
    Sampler                         smol/moca/sampler/sampler.py:22

What differs, by design: the boundary sits at the *Sampler* level.  A
Sampler owns ONE engine handle holding all its walkers on one GPU; ``run`` advances
every walker ``thin_by`` steps per kernel launch instead of looping over Python kernel
objects one flip at a time.  The per-flip ``Processor`` methods are kept (GPU-backed,
one launch per call) for parity tests and user-side checks, not for sampling.

Everything here drives the HIP engine through smol_amd.engine; there is no CPU path.
"""

from __future__ import annotations

import json
import os
import warnings
from datetime import datetime
from types import SimpleNamespace

import numpy as np

from . import capi
from .engine import Engine

kB = 8.617333262145e-5  # smol/constants.py:4


# --------------------------------------------------------------------------- #
class Trace(SimpleNamespace):
    """Namespace of ndarrays recorded during sampling (smol/moca/trace.py:8-46)."""

    def __init__(self, /, **kwargs):
        if not all(isinstance(v, np.ndarray) for v in kwargs.values()):
            raise TypeError("Trace only supports attributes of type ndarray.")
        super().__init__(**kwargs)

    @property
    def names(self):
        return tuple(self.__dict__.keys())

    def items(self):
        yield from self.__dict__.items()

    def __setattr__(self, name, value):
        if isinstance(value, float):
            value = np.array([value], dtype=np.float64)
        if isinstance(value, int):
            value = np.array([value], dtype=np.int32)
        if not isinstance(value, np.ndarray):
            raise TypeError("Trace only supports attributes of type ndarray.")
        self.__dict__[name] = value

    def as_dict(self):
        return self.__dict__.copy()


class StepTrace(Trace):
    """Trace of one attempted step: the walker's values plus ``delta_trace``, the changes the step
    made to them (smol/moca/trace.py:46-90); ``names`` / ``items`` skip the inner trace."""

    def __init__(self, /, **kwargs):
        super().__init__(**kwargs)
        self.__dict__["delta_trace"] = Trace()

    @property
    def names(self):
        return tuple(n for n in self.__dict__ if n != "delta_trace")

    def items(self):
        for name, value in self.__dict__.items():
            if name != "delta_trace":
                yield name, value

    def __setattr__(self, name, value):
        if name == "delta_trace":
            raise ValueError("Attribute name 'delta_trace' is reserved.")
        super().__setattr__(name, value)

    def as_dict(self):
        d = self.__dict__.copy()
        d["delta_trace"] = d["delta_trace"].as_dict()
        return d


class Sublattice:
    """Sites sharing one site space (smol/moca/sublattice.py:23-107)."""

    def __init__(self, species, sites, charges=None):
        self.species = tuple(species)
        # oxidation state of every species (0 for a vacancy); used by CompositionSpace
        given = [0] * len(self.species) if charges is None else list(charges)
        self.charges = tuple(0 if q is None else q for q in given)
        self.sites = np.unique(np.asarray(sites, dtype=np.int64))
        self.active_sites = self.sites.copy()
        if len(self.species) <= 1:
            self.restrict_sites(self.sites)
        self.encoding = np.arange(len(self.species), dtype=np.int32)

    @property
    def is_active(self):
        return len(self.active_sites) > 0

    @property
    def restricted_sites(self):
        return np.setdiff1d(self.sites, self.active_sites)

    def restrict_sites(self, sites):
        sites = set(int(s) for s in sites)
        self.active_sites = np.array([i for i in self.active_sites if i not in sites], dtype=np.int64)

    def reset_restricted_sites(self):
        if len(self.species) > 1:
            self.active_sites = self.sites.copy()

    def _code_of(self, sp):
        if isinstance(sp, (int, np.integer)):
            code = int(sp)
            if code not in self.encoding:
                raise ValueError(f"code {code} is not in this sublattice's encoding {self.encoding}")
            return code
        return int(self.encoding[self.species.index(sp)])

    def split_by_species(self, occu, species_in_partitions):
        """Split into one sublattice per partition of this one's species, by what occupies each
        site in ``occu`` (smol/moca/sublattice.py:109-186): partition p gets the sites holding
        one of its species (given as species or as codes), keeps those codes as its encoding, and a
        single-species partition is restricted (inactive).  The ACTIVE sites are listed code by code in
        ascending code order, as the reference lists them; ``sites`` is sorted (the constructor's np.unique)."""
        occu = np.asarray(occu)
        out = []
        for partition in species_in_partitions:
            codes = sorted(self._code_of(sp) for sp in partition)
            where = [int(np.where(self.encoding == c)[0][0]) for c in codes]
            sites = np.concatenate([self.sites[occu[self.sites] == c] for c in codes] + [np.zeros(0, np.int64)])
            actives = np.concatenate([self.active_sites[occu[self.active_sites] == c] for c in codes]
                                     + [np.zeros(0, np.int64)])
            # (`sites` stays what the constructor makes of it -- sorted, np.unique, sublattice.py:57 -- only the
            # ACTIVE sites are listed code by code, :176-178: reset_restricted_sites then restores sorted sites)
            part = Sublattice([self.species[i] for i in where], sites, [self.charges[i] for i in where])
            part.active_sites = actives.astype(np.int64)
            part.encoding = np.array(codes, dtype=np.int32)
            if len(codes) == 1:
                part.restrict_sites(part.sites)
            out.append(part)
        return out

    def __eq__(self, other):
        """Same species, encoding and sites (restrictions are not compared, sublattice.py:188-199)."""
        return (isinstance(other, Sublattice) and self.species == other.species
                and np.array_equal(self.encoding, other.encoding) and np.array_equal(self.sites, other.sites))

    __hash__ = None

    def as_dict(self):
        """JSON-able record (sublattice.py:201-217)."""
        return dict(species=list(self.species), charges=list(self.charges), sites=self.sites.tolist(),
                    encoding=self.encoding.tolist(), active_sites=self.active_sites.tolist())

    @classmethod
    def from_dict(cls, d):
        sub = cls(d["species"], d["sites"], d.get("charges"))
        sub.sites = np.array(d["sites"], dtype=np.int64)
        sub.encoding = np.array(d["encoding"], dtype=np.int32)
        sub.active_sites = np.array(d["active_sites"], dtype=np.int64)
        return sub


# --------------------------------------------------------------------------- #
# processors
# --------------------------------------------------------------------------- #
def default_species(prim):
    names = []
    for b, S in enumerate(prim.nspecies):
        names.append([f"{chr(65 + b)}{c}" for c in range(S)])
    return names


def _eval_device():
    """HIP device of the one-walker evaluation handles: this rank's GPU under a one-process-per-GPU
    launcher (LOCAL_RANK), else device 0."""
    if "LOCAL_RANK" not in os.environ:
        return 0
    from . import parallel

    return parallel.local_device(int(os.environ["LOCAL_RANK"]))


class Processor:
    """Base of the GPU-backed processors (smol/moca/processor/base.py:26)."""

    def __init__(self, supercell, coefficients):
        self.supercell = supercell
        self.cluster_subspace = supercell.model
        self.coefs = np.asarray(coefficients, dtype=np.float64)
        self.size = supercell.size
        self.num_sites = supercell.num_sites
        self.supercell_matrix = supercell.scmatrix
        self._eval_engine = None
        self._eval_tables = None

    # -- pieces of the flattened tables this processor contributes --------------
    def _table_kwargs(self):
        raise NotImplementedError

    def _make_tables(self, mu_table=None, sublattices=None):
        kw = self._table_kwargs()
        tab = capi.TableSet.from_synth(
            self.supercell, kw["ce_coefs_expansion"], feature_mode=kw["feature_mode"],
            ewald=kw.get("ewald"), ewald_coef=kw.get("ewald_coef", 1.0), mu_table=mu_table,
            ewald_charges=kw.get("ewald_charges", "auto"),
        )
        if kw.get("ce_natural") is not None:
            tab._keep["ce_coefs"][:] = kw["ce_natural"]
        if kw.get("interaction_tensors") is not None:
            its = kw["interaction_tensors"]
            flat = np.concatenate([np.ravel(np.asarray(x, dtype=np.float64)) for x in its[1:]])
            if flat.size != tab._keep["interaction_tensors"].size:
                raise ValueError(
                    "The number of cluster interaction tensors must match the number of orbits"
                )
            tab._keep["interaction_tensors"][:] = flat
            tab.struct.offset = float(its[0])
        if sublattices is not None:
            _override_sublattices(tab, sublattices)
        return tab

    def _engine(self):
        if self._eval_engine is None:
            self._eval_tables = self._make_tables()
            self._eval_engine = Engine(self._eval_tables, capi.make_config(1, device=_eval_device()))
        return self._eval_engine

    _feature_slice = slice(None)

    # -- reference API -----------------------------------------------------------
    def compute_feature_vector(self, occupancy):
        try:
            occupancy = np.array(occupancy, dtype=np.int32)
        except ValueError:
            types = {type(n) for n in occupancy}
            raise ValueError(f"occupancy contains {types}, but should be integers!")
        out = self._engine().eval_full(occupancy[None, :])[0][self._feature_slice]
        return float(out[0]) if self._scalar_feature else out

    _scalar_feature = False

    def compute_feature_vector_change(self, occupancy, flips):
        try:
            occupancy = np.array(occupancy, dtype=np.int32)
        except ValueError:
            types = {type(n) for n in occupancy}
            raise ValueError(f"occupancy contains {types}, but should be integers!")
        flips = list(flips)
        if len(flips) > capi.MAX_STEP_FLIPS:
            # sequential semantics (expansion.py:217-229): chain in records of eight flips
            total = 0.0
            occ = occupancy.copy()
            for i in range(0, len(flips), capi.MAX_STEP_FLIPS):
                chunk = flips[i:i + capi.MAX_STEP_FLIPS]
                total = total + self.compute_feature_vector_change(occ, chunk)
                for s, c in chunk:
                    occ[s] = c
            return total
        out = self._engine().eval_delta(occupancy, capi.step_rows([flips]))[0][self._feature_slice]
        return float(out[0]) if self._scalar_feature else out

    def compute_property(self, occupancy):
        return np.dot(self.coefs, self.compute_feature_vector(occupancy))

    def compute_property_change(self, occupancy, flips):
        return np.dot(self.coefs, self.compute_feature_vector_change(occupancy, flips))

    def compute_average_drift(self, iterations=1000, rng=None):
        """Average forward / reverse drift of local updates (processor/base.py:270-312)."""
        rng = np.random.default_rng(rng)
        prim = self.supercell.model.prim
        nsp = np.array([prim.nspecies[b] for b in self.supercell.site_b])
        active = np.flatnonzero(nsp > 1)
        occ = (rng.random(self.num_sites) * nsp).astype(np.int32)
        fwd = rev = 0.0
        f_prev = self.compute_feature_vector(occ)
        for _ in range(iterations):
            s = int(rng.choice(active))
            c = int((occ[s] + 1 + rng.integers(0, nsp[s] - 1)) % nsp[s])
            d = self.compute_feature_vector_change(occ, [(s, c)])
            new = occ.copy()
            new[s] = c
            dr = self.compute_feature_vector_change(new, [(s, int(occ[s]))])
            f_new = self.compute_feature_vector(new)
            fwd += np.sum(np.abs(f_new - f_prev - d))
            rev += np.sum(np.abs(f_prev - f_new - dr))
            occ, f_prev = new, f_new
        return fwd / iterations, rev / iterations

    def get_sublattices(self):
        """One Sublattice per distinct site space (processor/base.py get_sublattices)."""
        prim = self.supercell.model.prim
        names = getattr(prim, "species", None) or default_species(prim)
        groups = {}
        for b in range(prim.nb):
            key = prim.labels[b]
            sites = np.flatnonzero(self.supercell.site_b == b)
            if key in groups:
                groups[key][1].append(sites)
            else:
                groups[key] = (names[b], [sites], prim.charges[b])
        return [Sublattice(sp, np.concatenate(st), q) for sp, st, q in groups.values()]

    @property
    def allowed_species(self):
        """Species names of every site's site space, in code order (processor/base.py:104-107)."""
        prim = self.supercell.model.prim
        names = getattr(prim, "species", None) or default_species(prim)
        return [list(names[b]) for b in self.supercell.site_b]

    def encode_occupancy(self, occupancy):
        """Species names (or codes already) -> int32 codes (processor/base.py:228-237)."""
        allowed = self.allowed_species
        if len(occupancy) != len(allowed):
            raise ValueError(f"occupancy has {len(occupancy)} entries, the supercell {len(allowed)} sites")
        return np.array([sp if isinstance(sp, (int, np.integer)) else species.index(sp)
                         for species, sp in zip(allowed, occupancy)], dtype=np.int32)

    def decode_occupancy(self, encoded_occupancy):
        """int codes -> species names (processor/base.py:239-243)."""
        return [species[int(i)] for i, species in zip(encoded_occupancy, self.allowed_species)]


class ClusterExpansionProcessor(Processor):
    """Correlation-vector features (smol/moca/processor/expansion.py:39-241)."""

    def __init__(self, supercell, coefficients):
        if len(coefficients) != supercell.model.num_corr_functions:
            raise ValueError(
                f"The provided coefficients are not the right length. Got {len(coefficients)} "
                f"coefficients, the length must be {supercell.model.num_corr_functions} based on "
                "the provided cluster subspace."
            )
        super().__init__(supercell, coefficients)

    def _table_kwargs(self):
        return dict(feature_mode=capi.FEATURES_CORRELATIONS, ce_coefs_expansion=self.coefs)


class ClusterDecompositionProcessor(Processor):
    """Cluster-interaction features (smol/moca/processor/expansion.py:243-489)."""

    def __init__(self, supercell, interaction_tensors, coefficients=None):
        model = supercell.model
        if len(interaction_tensors) != model.num_orbits:
            raise ValueError(
                f"The number of cluster interaction tensors must match the number  of orbits in "
                f"the subspace. Got {len(interaction_tensors)} interaction tensors, but need "
                f"{model.num_orbits}  for the given cluster_subspace."
            )
        coefficients = model.orbit_multiplicities if coefficients is None else coefficients
        super().__init__(supercell, np.asarray(coefficients, dtype=np.float64))
        self._interaction_tensors = interaction_tensors

    def _table_kwargs(self):
        return dict(
            feature_mode=capi.FEATURES_INTERACTIONS,
            ce_coefs_expansion=np.zeros(self.supercell.model.num_corr_functions),
            ce_natural=self.coefs,
            interaction_tensors=self._interaction_tensors,
        )


class EwaldProcessor(Processor):
    """Electrostatic feature (smol/moca/processor/ewald.py:26-203).

    ``ewald_term`` is ``(ewald_inds, ewald_matrix[, charges])``; when None it is computed with the
    build's own Ewald sum (smol_amd.ewald; values unpinned vs pymatgen).  The optional
    per-entry ``charges`` (length M) let the engine factorise the matrix (compact Ewald / potential
    field); without them they are taken from the prim cell's oxidation states."""

    _scalar_feature = True

    def __init__(self, supercell, ewald_term=None, coefficient=1.0, use_term="total"):
        super().__init__(supercell, np.asarray(coefficient, dtype=np.float64))
        if ewald_term is None:  # (use_term: EwaldTerm.ewald_term_options, cofe/extern/ewald.py:28,159-177)
            from . import ewald as _ew

            ewald_term = _ew.supercell_ewald(supercell, use_term=use_term)
        self._ewald_inds = np.ascontiguousarray(ewald_term[0], dtype=np.int32)
        self.ewald_matrix = np.ascontiguousarray(ewald_term[1], dtype=np.float64)
        self._ewald_charges = (np.ascontiguousarray(ewald_term[2], dtype=np.float64)
                               if len(ewald_term) > 2 and ewald_term[2] is not None else "auto")
        self._feature_slice = slice(supercell.model.num_orbits, supercell.model.num_orbits + 1)

    def _table_kwargs(self):
        m = self.supercell.model
        return dict(
            feature_mode=capi.FEATURES_INTERACTIONS,
            ce_coefs_expansion=np.zeros(m.num_corr_functions),
            ce_natural=np.zeros(m.num_orbits),
            ewald=(self._ewald_inds, self.ewald_matrix),
            ewald_coef=float(self.coefs),
            ewald_charges=self._ewald_charges,
        )

    def compute_property(self, occupancy):
        return self.coefs * self.compute_feature_vector(occupancy)

    def compute_property_change(self, occupancy, flips):
        return self.coefs * self.compute_feature_vector_change(occupancy, flips)


class CompositeProcessor(Processor):
    """CE + external term (smol/moca/processor/composite.py:18-190)."""

    def __init__(self, supercell):
        super().__init__(supercell, np.zeros(0))
        self._processors = []

    @property
    def processors(self):
        return self._processors

    def add_processor(self, processor):
        if isinstance(processor, CompositeProcessor):
            raise AttributeError("A CompositeProcessor can not be added into another CompositeProcessor")
        if processor.supercell is not self.supercell:
            raise ValueError("processors must share the supercell")
        if len(self._processors) >= 2 or (self._processors and not isinstance(processor, EwaldProcessor)):
            raise ValueError("supported composition: one cluster processor followed by one EwaldProcessor")
        self._processors.append(processor)
        self.coefs = np.append(self.coefs, processor.coefs)
        self._eval_engine = None

    def _table_kwargs(self):
        if not self._processors:
            raise ValueError("empty CompositeProcessor")
        kw = dict(self._processors[0]._table_kwargs())
        if len(self._processors) == 2:
            ew = self._processors[1]
            kw["ewald"] = (ew._ewald_inds, ew.ewald_matrix)
            kw["ewald_coef"] = float(ew.coefs)
            kw["ewald_charges"] = ew._ewald_charges
        return kw


def _override_sublattices(tab, sublattices):
    """Re-point a TableSet's sublattice arrays at user-given Sublattice objects."""
    import ctypes as C

    subs = [s for s in sublattices if s.is_active]
    k = tab._keep
    k["sub_site_ptr"] = np.concatenate(([0], np.cumsum([len(s.active_sites) for s in subs]))).astype(np.int64)
    k["sub_active_sites"] = np.concatenate([s.active_sites for s in subs]).astype(np.int32)
    k["sub_code_ptr"] = np.concatenate(([0], np.cumsum([len(s.encoding) for s in subs]))).astype(np.int64)
    k["sub_codes"] = np.concatenate([s.encoding for s in subs]).astype(np.int32)
    probs = getattr(tab, "_user_probs", None)
    k["sub_probs"] = np.full(len(subs), 1.0 / len(subs)) if probs is None else np.asarray(probs, float)
    t = tab.struct
    t.n_sublattices = len(subs)
    for name, ct in (("sub_site_ptr", C.c_int64), ("sub_active_sites", C.c_int32),
                     ("sub_code_ptr", C.c_int64), ("sub_codes", C.c_int32), ("sub_probs", C.c_double)):
        setattr(t, name, k[name].ctypes.data_as(C.POINTER(ct)))


# --------------------------------------------------------------------------- #
class Ensemble:
    """Thermodynamic ensemble (smol/moca/ensemble.py:102-430)."""

    def __init__(self, processor, sublattices=None, chemical_potentials=None):
        self._processor = processor
        self._sublattices = processor.get_sublattices() if sublattices is None else sublattices
        self.natural_parameters = np.array(processor.coefs, dtype=np.float64)  # ensemble.py:126
        self.thermo_boundaries = {}
        self._chemical_potentials = None
        self._mu_table = None
        self.chemical_potentials = chemical_potentials

    @classmethod
    def from_cluster_expansion(cls, supercell, coefficients, processor_type="decomposition",
                               ewald_term=None, ewald_coefficient=None, **kwargs):
        """Counterpart of Ensemble.from_cluster_expansion (ensemble.py:133-217) for the
        build's own tables: ``supercell`` is a smol_amd.synth.SupercellTables and
        ``coefficients`` the expansion coefficients (num_corr_functions long).  When
        ``ewald_coefficient`` is given an EwaldProcessor is composed in."""
        if hasattr(supercell, "subspace") and hasattr(supercell, "ce_coefs"):
            # the reference's own call shape: (cluster_expansion, supercell_matrix) with an expansion
            # loaded from smol's serialization (smol_amd.mson.load_mson)
            return cls.from_mson(supercell, coefficients, processor_type=processor_type, **kwargs)
        model = supercell.model
        if processor_type == "decomposition":
            ce = ClusterDecompositionProcessor(supercell, model.cluster_interaction_tensors(coefficients))
        elif processor_type == "expansion":
            ce = ClusterExpansionProcessor(supercell, coefficients)
        else:
            raise ValueError(f"Processor type {processor_type} not supported!")
        if ewald_coefficient is None:
            return cls(ce, **kwargs)
        comp = CompositeProcessor(supercell)
        comp.add_processor(ce)
        comp.add_processor(EwaldProcessor(supercell, ewald_term, ewald_coefficient))
        return cls(comp, **kwargs)

    @classmethod
    def from_mson(cls, model, supercell_matrix, processor_type="decomposition", **kwargs):
        """Ensemble of a model serialized by smol (``ClusterExpansion.as_dict`` JSON / a notebook
        ``save_work`` file, or an already loaded smol_amd.mson.MsonClusterExpansion) on a
        supercell: what ``Ensemble.from_cluster_expansion(expansion, supercell_matrix)`` builds in
        the reference (ensemble.py:133-217) -- a cluster processor, composed with an
        EwaldProcessor carrying the last fitted coefficient when the subspace has an EwaldTerm."""
        from . import mson

        ce = model if isinstance(model, mson.MsonClusterExpansion) else mson.load_mson(model)
        cell = ce.subspace.supercell(supercell_matrix)
        ewald_term = coef = None
        if ce.subspace.ewald_term is not None:
            ewald_term, coef = ce.ewald_tables(cell), float(ce.coefs[-1])  # (inds, matrix, charges)
        return cls.from_cluster_expansion(cell, ce.ce_coefs, processor_type=processor_type,
                                          ewald_term=ewald_term, ewald_coefficient=coef, **kwargs)

    # -- properties ----------------------------------------------------------------
    @property
    def processor(self):
        return self._processor

    @property
    def num_sites(self):
        return self._processor.num_sites

    @property
    def num_energy_coefs(self):
        return len(self._processor.coefs)

    @property
    def system_size(self):
        return self._processor.size

    @property
    def sublattices(self):
        return self._sublattices

    @property
    def active_sublattices(self):
        return [s for s in self._sublattices if s.is_active]

    @property
    def restricted_sites(self):
        return np.concatenate([s.restricted_sites for s in self._sublattices] + [np.zeros(0, int)])

    @property
    def species(self):
        return [sp for s in self.active_sublattices for sp in s.species]

    @property
    def chemical_potentials(self):
        return self._chemical_potentials

    @chemical_potentials.setter
    def chemical_potentials(self, value):
        """ChemicalPotentialManager.__set__ (ensemble.py:35-73)."""
        had = self._chemical_potentials is not None
        if value is None:
            if had:
                self.natural_parameters = self.natural_parameters[:-1]
                self.thermo_boundaries.pop("chemical_potentials", None)
            self._chemical_potentials, self._mu_table = None, None
            return
        value = {k: float(v) for k, v in value.items() if k in self.species}
        if set(value) != set(self.species):
            raise ValueError(
                "Chemical potentials given are missing species. Values must be given for each of "
                f"the following: {self.species}"
            )
        if not had:
            self.natural_parameters = np.append(self.natural_parameters, -1.0)  # ensemble.py:25,61-65
        self._chemical_potentials = value
        num_cols = max(max(s.encoding) for s in self._sublattices) + 1  # _build_table :90-99
        table = np.zeros((self.num_sites, num_cols))
        for s in self.active_sublattices:
            table[s.sites[:, None], s.encoding] = [value[sp] for sp in s.species]
        self._mu_table = table
        self.thermo_boundaries["chemical_potentials"] = value

    def restrict_sites(self, sites):
        for s in self._sublattices:
            s.restrict_sites(sites)

    def reset_restricted_sites(self):
        for s in self._sublattices:
            s.reset_restricted_sites()

    def split_sublattice_by_species(self, sublattice_id, occu, species_in_partitions):
        """Replace sublattice ``sublattice_id`` by its split by occupying species
        (ensemble.py:288-321), e.g. Li/vacancy, transition metals and O of one site space as
        sublattices that swaps cannot mix.  The chemical potentials are rebuilt for the species that
        remain on active sublattices."""
        splits = self._sublattices[sublattice_id].split_by_species(occu, species_in_partitions)
        self._sublattices = (list(self._sublattices[:sublattice_id]) + splits
                             + list(self._sublattices[sublattice_id + 1:]))
        if self._chemical_potentials is not None:
            self.chemical_potentials = {sp: self._chemical_potentials[sp] for sp in self.species}

    def composition_space(self, charge_balanced=True, other_constraints=None, optimize_basis=False,
                          table_ergodic=False):
        """CompositionSpace over ALL sublattices of this ensemble, sizes reduced by their gcd,
        as TableFlip builds it (mcusher.py:489-518)."""
        from .composition import CompositionSpace

        sizes = np.array([len(s.sites) for s in self._sublattices], dtype=np.int64)
        sizes = sizes // np.gcd.reduce(sizes)
        spaces = [list(zip(s.species, s.charges)) for s in self._sublattices]
        return CompositionSpace(spaces, sizes, charge_neutral=charge_balanced,
                                other_constraints=other_constraints, optimize_basis=optimize_basis,
                                table_ergodic=table_ergodic)

    # -- tables for the engine -----------------------------------------------------
    def make_tables(self, flip_table=None, flip_weights=None, swap_weight=0.1):
        """Flattened tables of this ensemble, in its own site numbering (restricted sites and sublattices split by
        species leave active sites scattered: smolmc_create renumbers them behind the C-ABI, ``kernel_info`` then
        says "relabelled=1", and callers never see it)."""
        tab = self._processor._make_tables(mu_table=self._mu_table, sublattices=self._sublattices)
        if flip_table is not None:
            tab = self._with_flip_table(tab, flip_table, flip_weights, swap_weight)
        return tab

    def _with_flip_table(self, tab, flip_table, flip_weights, swap_weight):
        """Attach a TableFlip table.  The reference's flip vectors span the species of ALL
        sublattices in order (mcusher.py:489-503); the engine wants the active ones only."""
        import ctypes as C

        ft = np.atleast_2d(np.asarray(flip_table, dtype=np.int64))
        widths = [len(s.species) for s in self._sublattices]
        act = [s.is_active for s in self._sublattices]
        if ft.shape[1] == sum(widths):
            cols, off = [], 0
            for w, a in zip(widths, act):
                if a:
                    cols += list(range(off, off + w))
                elif np.any(ft[:, off:off + w] != 0):
                    raise ValueError("flip table changes species on an inactive sublattice")
                off += w
            ft = ft[:, cols]
        elif ft.shape[1] != sum(w for w, a in zip(widths, act) if a):
            raise ValueError("flip_table width does not match the sublattice species")
        ft = np.ascontiguousarray(ft, dtype=np.int32)
        fw = np.ones(2 * len(ft)) if flip_weights is None else np.asarray(flip_weights, dtype=np.float64)
        if len(fw) == len(ft):
            fw = np.repeat(fw, 2)
        if len(fw) != 2 * len(ft):
            raise ValueError(
                f"{len(fw)} weights provided. You must provide either 1* or 2* weights given "
                f"{len(ft)} flip vectors!"
            )
        tab._keep["flip_table"], tab._keep["flip_weights"] = ft, np.ascontiguousarray(fw)
        t = tab.struct
        t.n_flip_vectors = len(ft)
        t.swap_weight = float(swap_weight)
        t.flip_table = ft.ctypes.data_as(C.POINTER(C.c_int32))
        t.flip_weights = tab._keep["flip_weights"].ctypes.data_as(C.POINTER(C.c_double))
        return tab

    def _eval(self):
        # content key: a different restriction of the same size, or a new mu table that reuses
        # the id() of a freed one, must rebuild the handle
        key = (None if self._mu_table is None else self._mu_table.tobytes(),
               tuple(np.asarray(s.active_sites).tobytes() + b"|" + np.asarray(s.encoding).tobytes()
                     for s in self._sublattices))
        if getattr(self, "_eval_key", None) != key:
            self._eval_tables = self.make_tables()
            self._eval_engine = Engine(self._eval_tables, capi.make_config(1, device=_eval_device()))
            self._eval_key = key
        return self._eval_engine

    def compute_feature_vector(self, occupancy):
        """ensemble.py:323-351."""
        return self._eval().eval_full(np.asarray(occupancy, dtype=np.int32)[None, :])[0]

    def compute_feature_vector_change(self, occupancy, step):
        """ensemble.py:353-376 (steps of up to SMOLMC_MAX_STEP_FLIPS flips, e.g. TableFlip's)."""
        return self._eval().eval_delta(np.asarray(occupancy, dtype=np.int32), capi.step_rows([list(step)]))[0]


# --------------------------------------------------------------------------- #
# kernels (specifications of what the engine runs for each walker)
# --------------------------------------------------------------------------- #
STEP_TYPES = {"flip": capi.STEP_FLIP, "swap": capi.STEP_SWAP, "table-flip": capi.STEP_TABLE_FLIP,
              "tableflip": capi.STEP_TABLE_FLIP}


# --------------------------------------------------------------------------- #
# bias terms (smol/moca/kernel/bias.py)
# --------------------------------------------------------------------------- #
class MCBias:
    """Host record of a bias term: builds the per-(site, species code) table the engine
    consumes (smol/moca/kernel/bias.py:24-93)."""

    bias_type = capi.BIAS_NONE
    penalty = 0.0

    def __init__(self, sublattices):
        self.sublattices = list(sublattices)
        self.active_sublattices = [s for s in self.sublattices if s.is_active]
        self.spec = dict(type=self.__class__.__name__)

    def _shape(self):
        return (sum(len(s.sites) for s in self.sublattices),
                max(int(max(s.encoding)) for s in self.sublattices) + 1)


class FugacityBias(MCBias):
    """Fugacity-fraction bias (bias.py:96-226): bias = sum_sites log f(site, species)."""

    bias_type = capi.BIAS_FUGACITY

    def __init__(self, sublattices, fugacity_fractions=None):
        super().__init__(sublattices)
        self._species = [set(s.species) for s in self.active_sublattices]
        if fugacity_fractions is None:
            # the reference takes the prim's site compositions (bias.py:139-141); the synthetic
            # prims carry none, so the default is the uniform composition
            fugacity_fractions = [{sp: 1.0 / len(s.species) for sp in s.species}
                                  for s in self.active_sublattices]
        self.fugacity_fractions = fugacity_fractions

    @property
    def fugacity_fractions(self):
        return self._fus

    @fugacity_fractions.setter
    def fugacity_fractions(self, value):
        value = [dict(v) for v in value]
        if not all(sum(fus.values()) == 1 for fus in value):
            raise ValueError("Fugacity ratios must add to one.")  # bias.py:161-162
        if len(value) != len(self._species):
            raise ValueError("one fugacity-fraction dictionary per active sublattice is required")
        for spec, vals in zip(self._species, value):
            if spec != set(vals.keys()):
                raise ValueError(
                    "Fugacity fractions given are missing or not valid species.\n"
                    f"Values must be given for each  of the following: {self._species}"
                )  # bias.py:163-170
        self._fus = value
        table = np.ones(self._shape())  # _build_fu_table (bias.py:208-226)
        for fus, s in zip(value, self.active_sublattices):
            table[s.sites[:, None], s.encoding] = np.array([fus[sp] for sp in s.species])[None, :]
        self._table = table
        self.spec["fugacity_fractions"] = value

    def compute_bias(self, occupancy):
        return float(sum(np.log(self._table[i, c]) for i, c in enumerate(occupancy)))


class SquareChargeBias(MCBias):
    """Square-charge bias (bias.py:229-277): bias = -penalty * (net charge)^2."""

    bias_type = capi.BIAS_SQUARE_CHARGE

    def __init__(self, sublattices, penalty=0.5):
        super().__init__(sublattices)
        if penalty <= 0:
            raise ValueError("Penalty factor should be > 0!")
        self.penalty = float(penalty)
        table = np.zeros(self._shape())
        for s in self.sublattices:
            table[s.sites[:, None], s.encoding] = np.array(s.charges, dtype=float)[None, :]
        self._table = table
        self.spec["penalty"] = self.penalty

    def compute_bias(self, occupancy):
        c = self._table[np.arange(len(occupancy)), occupancy].sum()
        return float(-self.penalty * c**2)


class SquareHyperplaneBias(MCBias):
    """Square hyperplane bias (bias.py:290-366): bias = -penalty * ||A n - b||^2 with n the
    species counts of the occupancy in "counts" format -- the species of every sublattice in
    order, concatenated over ALL sublattices (occu_utils.py:8-23,62-100) -- i.e. a penalty on the
    distance from composition constraints A n = b (per supercell).  Up to four hyperplanes."""

    bias_type = capi.BIAS_SQUARE_HYPERPLANE

    def __init__(self, sublattices, hyperplane_normals, hyperplane_intercepts, penalty=0.5):
        super().__init__(sublattices)
        if penalty <= 0:
            raise ValueError("Penalty factor should be > 0!")
        self.penalty = float(penalty)
        self._A = np.atleast_2d(np.array(hyperplane_normals, dtype=int))
        self._b = np.atleast_1d(np.array(hyperplane_intercepts, dtype=int))
        self.d = sum(len(s.species) for s in self.sublattices)
        if self._A.shape != (len(self._b), self.d):
            raise ValueError(f"hyperplane_normals must be [len(intercepts) x {self.d}] (one column per "
                             "species of every sublattice)")
        if not 1 <= len(self._b) <= 4:
            raise NotImplementedError("the engine supports 1 to 4 hyperplanes")
        # get_dim_ids_table (occu_utils.py:8-23): dimension of (site, code) in the counts vector
        n_sites, width = self._shape()
        self._dim_ids_table = -np.ones((n_sites, width), dtype=int)
        off = 0
        for s in self.sublattices:
            self._dim_ids_table[s.sites[:, None], s.encoding] = off + np.arange(len(s.species))[None, :]
            off += len(s.species)
        table = np.zeros((len(self._b), n_sites, width))
        valid = self._dim_ids_table >= 0
        for r in range(len(self._b)):
            table[r][valid] = self._A[r][self._dim_ids_table[valid]]
        self._table = table
        self.intercepts = self._b.astype(float)
        self.spec.update(penalty=self.penalty, hyperplane_normals=self._A.tolist(),
                         hyperplane_intercepts=self._b.tolist())

    def counts(self, occupancy):
        """occu_to_counts (occu_utils.py:62-100)."""
        ids = self._dim_ids_table[np.arange(len(occupancy)), occupancy]
        return np.bincount(ids[ids >= 0], minlength=self.d)

    def compute_bias(self, occupancy):
        n = self.counts(np.asarray(occupancy))
        return float(-self.penalty * np.sum((self._A @ n - self._b) ** 2))


BIAS_TYPES = {"fugacity": FugacityBias, "fugacity-bias": FugacityBias, "fugacitybias": FugacityBias,
              "square-charge": SquareChargeBias, "square-charge-bias": SquareChargeBias,
              "squarechargebias": SquareChargeBias,
              "square-hyperplane": SquareHyperplaneBias, "square-hyperplane-bias": SquareHyperplaneBias,
              "squarehyperplanebias": SquareHyperplaneBias}


def mcbias_factory(bias_type, sublattices, **kwargs):
    """bias.py mcbias_factory: class from its (hyphenated / camel-case) name."""
    key = str(bias_type).lower().replace("_", "-")
    if key not in BIAS_TYPES:
        raise NotImplementedError(f"{bias_type} is not implemented on the MI355X engine "
                                  f"(available: FugacityBias, SquareChargeBias, SquareHyperplaneBias).")
    return BIAS_TYPES[key](sublattices, **kwargs)


class MCKernel:
    """Per-walker kernel record (smol/moca/kernel/base.py:169-343).  Holds the parameters
    the reference keeps per kernel object; the stepping itself happens on the GPU."""

    kernel_type = None
    valid_bias = ("FugacityBias", "SquareChargeBias", "SquareHyperplaneBias")

    def __init__(self, ensemble, step_type, *args, seed=None, bias_type=None, bias_kwargs=None, **kwargs):
        # any spelling the reference's class_name_from_str accepts (class_utils.py:10-34):
        # "TableFlip", "Table-Flip", "table-flip", "Flip", "swap", ...
        key = str(step_type).lower().replace("-", "").replace("_", "")
        key = "table-flip" if key == "tableflip" else key
        if key not in STEP_TYPES:
            raise ValueError(
                f"{step_type} is not a valid MCUsher for this kernel (supported: Flip, Swap, TableFlip)."
            )
        step_type = key
        self._ensemble = ensemble
        self.natural_params = ensemble.natural_parameters
        self.step_type = step_type
        # MCUsher arguments (TableFlip: flip_table, flip_weights, swap_weight; mcusher.py:414-426)
        self.usher_kwargs = {k: kwargs[k] for k in ("flip_table", "flip_weights", "swap_weight")
                             if k in kwargs}
        if STEP_TYPES[step_type] == capi.STEP_TABLE_FLIP and self.usher_kwargs.get("flip_table") is None:
            # TableFlip.__init__ (mcusher.py:489-518): no table given -> build it from the
            # sublattices' species / charges with a CompositionSpace
            self.usher_kwargs["flip_table"] = ensemble.composition_space(
                charge_balanced=kwargs.get("charge_balanced", True),
                other_constraints=kwargs.get("other_constraints"),
                optimize_basis=kwargs.get("optimize_basis", False),
                table_ergodic=kwargs.get("table_ergodic", False),
            ).flip_table
        self._seed = seed if seed is not None else np.random.SeedSequence().entropy
        self.spec = dict(kernel=self.__class__.__name__, seed=self._seed, step=step_type)
        self._bias = None
        if bias_type is not None:  # kernel/base.py:229-235
            self.bias = mcbias_factory(bias_type, ensemble.sublattices, **(bias_kwargs or {}))

    @property
    def bias(self):
        return self._bias

    @bias.setter
    def bias(self, bias):
        if self.valid_bias is None:
            raise ValueError("Cannot apply bias to Wang-Landau simulation!")  # wanglandau.py:127-128
        if bias.__class__.__name__ not in self.valid_bias:
            raise ValueError(f"{type(bias)} is not a valid MCBias for this kernel.")  # base.py:281-282
        self._bias = bias
        self.spec["bias"] = bias.spec

    @property
    def ensemble(self):
        return self._ensemble

    @property
    def seed(self):
        return self._seed

    @property
    def seed64(self):
        return np.uint64(int(self._seed) & 0xFFFFFFFFFFFFFFFF)

    # -- single-kernel stepping (kernel/base.py:100-166,287-289,345-366) -----------------
    # The reference steps ONE chain on the host through these; here they drive a one-walker
    # engine handle owned by the kernel (same kernels, same random stream as walker 0 of a
    # Sampler seeded with this kernel's seed).  An interface for tests, notebooks and custom
    # drivers -- a launch per step -- not the sampling path (Sampler.run).
    _owner = None  # (sampler, local walker) once a Sampler took this kernel

    def _solo(self):
        if getattr(self, "_solo_sampler", None) is None:
            container = Sampler._container_for(self._ensemble, self, 1)
            container.metadata["walker_range"] = (0, 1, 1)
            self._solo_sampler = Sampler([self], container, device=_eval_device(), _bind=False)
        return self._solo_sampler

    def _walker_trace(self, tr, w=0):
        return StepTrace(**{k: (np.asarray(v[w]).astype(np.int32) if k == "occupancy" else np.asarray(v[w]))
                        for k, v in tr.items()})

    def set_aux_state(self, occupancy, *args, **kwargs):
        """Load ``occupancy`` as this kernel's chain state: features, enthalpy, bias, Ewald field,
        species counts and (Wang-Landau) the current bin are recomputed from it on the device; the
        random stream, counters and Wang-Landau arrays carry on (kernel/base.py:287-289,
        wanglandau.py:290-300)."""
        self._solo()._load_state(np.asarray(occupancy))

    def compute_initial_trace(self, occupancy):
        """Trace of ``occupancy`` before any step (kernel/base.py:345-366)."""
        solo = self._solo()
        solo._load_state(np.asarray(occupancy))
        tr = self._walker_trace(solo._current_trace(solo._get_engine()))
        tr.accepted = np.array([True], dtype=bool)
        return tr

    def single_step(self, occupancy):
        """Attempt one MC step from ``occupancy`` (kernel/base.py:145-166): the step is proposed,
        priced and accepted / rejected on the device; ``occupancy`` is updated in place when the step
        is accepted and the trace carries ``delta_trace`` (features, enthalpy, bias changes of the
        PROPOSED step when accepted, zero otherwise -- the rejected proposal never leaves the GPU)."""
        solo = self._solo()
        eng = solo._get_engine()
        occupancy_in = occupancy
        solo._load_state(np.asarray(occupancy))
        before = solo._current_trace(eng)
        eng.run(1)
        tr = self._walker_trace(solo._current_trace(eng))
        delta = tr.delta_trace
        delta.features = tr.features - before.features[0]
        delta.enthalpy = np.array(tr.enthalpy[0] - before.enthalpy[0, 0], dtype=np.float64)
        if self._bias is not None:
            delta.bias = np.array(tr.bias[0] - before.bias[0, 0], dtype=np.float64)
        if isinstance(occupancy_in, np.ndarray):
            occupancy_in[:] = tr.occupancy
            tr.occupancy = occupancy_in
        self._last_trace = tr
        return tr

    @property
    def trace(self):
        """Trace of the last ``single_step`` / of this kernel's walker in its Sampler."""
        if getattr(self, "_last_trace", None) is not None:
            return self._last_trace
        if self._owner is not None:
            sampler, w = self._owner
            return self._walker_trace(sampler._current_trace(sampler._get_engine()), w)
        raise AttributeError("no step has been taken with this kernel yet")


class Metropolis(MCKernel):
    """Metropolis-Hastings kernel (smol/moca/kernel/metropolis.py:52-100)."""

    kernel_type = capi.KERNEL_METROPOLIS

    def __init__(self, ensemble, step_type, temperature, *args, seed=None, **kwargs):
        super().__init__(ensemble, step_type, *args, seed=seed, **kwargs)
        self.kB = kB
        self.temperature = temperature

    @property
    def temperature(self):
        return self._temperature

    @temperature.setter
    def temperature(self, temperature):
        self._temperature = float(temperature)
        self.beta = 1.0 / (self.kB * self._temperature)  # kernel/base.py:418-422


class UniformlyRandom(MCKernel):
    """A kernel that accepts every proposed step (smol/moca/kernel/random.py:16-38): the infinite-
    temperature limit of Metropolis -- exponent = log a-priori factor (+ delta bias) -- which is
    exactly what the engine's Metropolis kernels compute at beta = 0.  No ``temperature`` in its
    trace, as in the reference (it is not a thermal kernel)."""

    kernel_type = capi.KERNEL_METROPOLIS
    engine_temperature = np.inf  # beta = 1 / (kB T) = 0 on the device


class WangLandau(MCKernel):
    """Wang-Landau kernel (smol/moca/kernel/wanglandau.py:17-300)."""

    kernel_type = capi.KERNEL_WANGLANDAU
    valid_bias = None  # "Wang-Landau does not need bias" (wanglandau.py:24)

    def __init__(self, ensemble, step_type, min_enthalpy, max_enthalpy, bin_size, *args,
                 flatness=0.8, mod_factor=1.0, check_period=1000, update_period=1,
                 mod_update=None, seed=None, **kwargs):
        if min_enthalpy > max_enthalpy:
            raise ValueError("min_enthalpy can not be larger than max_enthalpy.")
        if (max_enthalpy - min_enthalpy) / bin_size <= 1:
            raise ValueError(
                "The values provided for min and max enthalpy and bin sizer result in a single bin!"
            )
        if mod_factor <= 0:
            raise ValueError("mod_factor must be greater than 0.")
        super().__init__(ensemble, step_type, *args, seed=seed, **kwargs)
        self.flatness, self.check_period, self.update_period = flatness, check_period, update_period
        self._m0 = mod_factor
        self._window = (min_enthalpy, max_enthalpy, bin_size)
        # mod_update (wanglandau.py:100-105): a number divides the modification factor on the device;
        # a callable is applied by the host at every flatness check (Sampler._wl_host_checks: the
        # launches are then split at the check period and the device never checks by itself)
        self._mod_callable = mod_update if callable(mod_update) else None
        self._mod_divisor = 2.0 if (mod_update is None or callable(mod_update)) else float(mod_update)
        self._levels = np.arange(min_enthalpy, max_enthalpy, bin_size)
        self.spec.update(min_enthalpy=min_enthalpy, max_enthalpy=max_enthalpy, bin_size=bin_size,
                         flatness=flatness, check_period=check_period, update_period=update_period,
                         levels=self._levels.tolist())

    @property
    def bin_size(self):
        return self._window[2]

    # -- state of this kernel's walker (wanglandau.py:150-173), read from the device -------
    def _wl_row(self):
        if self._owner is not None:
            sampler, w = self._owner
        else:
            sampler, w = self._solo(), 0
        wl = sampler._get_engine().get_wl()
        return {k: v[w] for k, v in wl.items()}

    @property
    def levels(self):
        """Visited enthalpy levels."""
        return self._levels[self._wl_row()["entropy"] > 0]

    @property
    def entropy(self):
        """log(dos) on visited levels."""
        S = self._wl_row()["entropy"]
        return S[S > 0]

    @property
    def dos(self):
        """Density of states on visited levels."""
        S = self.entropy
        return np.exp(S - S.min())

    @property
    def histogram(self):
        """Histogram on visited levels."""
        row = self._wl_row()
        return row["histogram"][row["entropy"] > 0]

    @property
    def mod_factor(self):
        return float(self._wl_row()["mod_factor"])


KERNELS = {"metropolis": Metropolis, "wanglandau": WangLandau, "wang-landau": WangLandau,
           "uniformlyrandom": UniformlyRandom, "uniformly-random": UniformlyRandom, "random": UniformlyRandom}


def mckernel_factory(kernel_type, ensemble, step_type, *args, **kwargs):
    """smol/moca/kernel/__init__.py:35-56."""
    key = str(kernel_type).lower().replace("_", "")
    if key not in KERNELS:
        raise NotImplementedError(f"{kernel_type} is not implemented on the MI355X engine.")
    return KERNELS[key](ensemble, step_type, *args, **kwargs)


# --------------------------------------------------------------------------- #
# sample storage
# --------------------------------------------------------------------------- #
def _merge_walkers(a):
    """(samples, walkers, ...) -> (samples * walkers, ...), sample-major, singleton axes dropped
    (what the reference calls "flat", container.py:514-519)."""
    return np.squeeze(a.reshape((a.shape[0] * a.shape[1],) + a.shape[2:]))


class SampleContainer:
    """Thinned samples of one Sampler (interface of smol/moca/sampler/container.py:25-660).

    Storage follows the engine, not the reference's pre-allocated append-and-slice arrays: the
    device ring of ``smolmc_run_sampled`` hands over whole *blocks* of samples
    ``(n, nwalkers, ...)``; the container keeps the blocks as delivered and joins them once,
    lazily, when a getter first needs them.  ``allocate`` / ``vacuum`` therefore have nothing
    to do and exist only because callers written for smol invoke them.  The HDF5 backend is
    replaced by ``to_npz`` / ``from_npz`` (h5py is not available)."""

    def __init__(self, ensemble, sample_trace, sampling_metadata=None):
        self._ensemble = ensemble
        self.natural_parameters = ensemble.natural_parameters
        self._num_energy_coefs = ensemble.num_energy_coefs
        self.metadata = dict(sampling_metadata or {})
        # name -> (dtype, per-sample shape): the schema of a block
        self._schema = {k: (v.dtype, tuple(v.shape[1:])) for k, v in sample_trace.items()}
        self._blocks = [{k: v for k, v in sample_trace.items()}] if len(sample_trace.occupancy) else []
        self._joined = {}
        self._total_steps = 0

    # ---- block store ---------------------------------------------------------------
    def append_block(self, block, thinned_by):
        """Append ``n`` consecutive samples: ``block[name]`` has shape (n, nwalkers, ...)."""
        n = len(block["occupancy"])
        entry = {}
        for name, (dtype, shape) in self._schema.items():
            arr = np.asarray(block[name], dtype=dtype)
            entry[name] = arr.reshape((n,) + shape)
        self._blocks.append(entry)
        self._joined = {}
        self._total_steps += n * int(thinned_by)

    def save_sampled_trace(self, trace, thinned_by):
        """One sample (container.py:384-397): a block of length one."""
        self.append_block({k: np.asarray(v)[None] for k, v in trace.items()}, thinned_by)

    def _col(self, name):
        """All samples of one traced value, (nsamples, nwalkers, ...).  Joined per name and on
        demand: a mean enthalpy does not concatenate gigabytes of occupancies."""
        if name not in self._joined:
            dt, shape = self._schema[name]
            parts = [b[name] for b in self._blocks]
            self._joined[name] = (parts[0] if len(parts) == 1 else np.concatenate(parts) if parts
                                  else np.empty((0,) + shape, dtype=dt))
            if len(parts) > 1:  # keep one copy: the blocks now alias the joined array
                at = 0
                for b in self._blocks:
                    n = len(b[name])
                    b[name] = self._joined[name][at:at + n]
                    at += n
        return self._joined[name]

    def _all(self):
        return {k: self._col(k) for k in self._schema}

    @property
    def _trace(self):
        return Trace(**self._all())

    def allocate(self, nsamples):
        """No-op: blocks arrive sized by the device ring (container.py:409-413 pre-allocates)."""

    def vacuum(self):
        """No-op: there is no slack to give back (container.py:415-418)."""

    def clear(self):
        self._blocks, self._joined, self._total_steps = [], {}, 0

    def last_occupancy(self):
        """Occupancies (nwalkers, N) of the most recent sample (no join of the blocks)."""
        return self._blocks[-1]["occupancy"][-1].astype(np.int32)

    # ---- bookkeeping -----------------------------------------------------------------
    ensemble = property(lambda self: self._ensemble)
    sublattices = property(lambda self: self._ensemble.sublattices)
    num_samples = property(lambda self: sum(len(b["occupancy"]) for b in self._blocks))
    total_mc_steps = property(lambda self: self._total_steps)
    shape = property(lambda self: self._schema["occupancy"][1])
    traced_values = property(lambda self: tuple(self._schema))

    def __len__(self):
        return self.num_samples

    def sampling_efficiency(self, discard=0, flat=True):
        """Fraction of recorded samples whose last step was accepted (container.py:131-142)."""
        acc = self._col("accepted")[discard:]
        eff = acc.mean(axis=0)
        return eff.mean() if flat else eff

    # ---- selection + reductions --------------------------------------------------------
    def _select(self, name, discard, thin_by):
        """Samples discard + thin_by - 1, discard + 2 thin_by - 1, ... of one traced value in
        its storage dtype (the reference's selection rule, container.py:181-199)."""
        return self._col(name)[discard + thin_by - 1:: thin_by]

    def get_trace_value(self, name, discard=0, thin_by=1, flat=True):
        picked = self._select(name, discard, thin_by)
        if name == "occupancy":  # stored as bytes, the reference's trace.occupancy is int32
            picked = picked.astype(np.int32)
        return _merge_walkers(picked) if flat else picked

    def mean_trace_value(self, name, discard=0, thin_by=1, flat=True):
        return self.get_trace_value(name, discard, thin_by, flat).mean(axis=0)

    def trace_value_variance(self, name, discard=0, thin_by=1, flat=True):
        return self.get_trace_value(name, discard, thin_by, flat).var(axis=0)

    def get_temperatures(self, discard=0, thin_by=1):
        return self.get_trace_value("temperature", discard, thin_by)

    def get_energies(self, discard=0, thin_by=1, flat=True):
        """Energy = natural parameters . features over the energy coefficients only
        (the chemical work / bias terms are not part of it, container.py:217-233)."""
        n = self._num_energy_coefs
        if len(self.natural_parameters) == n:
            return self.get_trace_value("enthalpy", discard, thin_by, flat)
        feats = self.get_trace_value("features", discard, thin_by, flat=False)
        energies = (feats[..., :n] @ self.natural_parameters[:n])[..., None]
        return _merge_walkers(energies) if flat else energies

    def _argmin_occupancy(self, values, discard, thin_by, flat):
        where = values.argmin(axis=0)
        occ = self._select("occupancy", discard, thin_by)
        if flat:
            occ = _merge_walkers(occ)
            return occ[where].astype(np.int32)
        return occ[where, np.arange(self.shape[0])][0].astype(np.int32)

    def get_minimum_enthalpy_occupancy(self, discard=0, thin_by=1, flat=True):
        return self._argmin_occupancy(self.get_enthalpies(discard, thin_by, flat), discard, thin_by, flat)

    def get_minimum_energy_occupancy(self, discard=0, thin_by=1, flat=True):
        return self._argmin_occupancy(self.get_energies(discard, thin_by, flat), discard, thin_by, flat)

    # ---- compositions --------------------------------------------------------------------
    def get_sublattice_species_counts(self, sublattice, discard=0, thin_by=1, flat=True):
        """Counts of each species of one sublattice, last axis ordered like its site space
        (container.py:349-382)."""
        if not any(sublattice is s for s in self.sublattices):
            raise ValueError(
                "Sublattice provided is not recognized.\n Provide one included in the sublattices "
                "attribute of this SampleContainer."
            )
        occ = self._select("occupancy", discard, thin_by)[..., sublattice.sites]
        counts = np.stack([(occ == code).sum(axis=-1) for code in np.asarray(sublattice.encoding)],
                          axis=-1).astype(float)
        return _merge_walkers(counts) if flat else counts

    def get_species_counts(self, discard=0, thin_by=1, flat=True):
        """Counts per species name summed over the sublattices that host it (container.py:336-347)."""
        out = {}
        for sub in self.sublattices:
            counts = self.get_sublattice_species_counts(sub, discard, thin_by, flat)
            for j, sp in enumerate(sub.species):
                out[sp] = out.get(sp, 0) + counts[..., j].astype(np.int64)
        return out

    def get_sublattice_compositions(self, sublattice, discard=0, thin_by=1, flat=True):
        return self.get_sublattice_species_counts(sublattice, discard, thin_by, flat) / len(sublattice.sites)

    def get_compositions(self, discard=0, thin_by=1, flat=True):
        """Fraction of ALL sites held by each species (container.py:240-243)."""
        return {sp: c / self.shape[1] for sp, c in self.get_species_counts(discard, thin_by, flat).items()}

    def mean_composition(self, discard=0, thin_by=1, flat=True):
        return {sp: c.mean(axis=0) for sp, c in self.get_compositions(discard, thin_by, flat).items()}

    def composition_variance(self, discard=0, thin_by=1, flat=True):
        return {sp: c.var(axis=0) for sp, c in self.get_compositions(discard, thin_by, flat).items()}

    def mean_sublattice_composition(self, sublattice, discard=0, thin_by=1, flat=True):
        return self.get_sublattice_compositions(sublattice, discard, thin_by, flat).mean(axis=0)

    def sublattice_composition_variance(self, sublattice, discard=0, thin_by=1, flat=True):
        return self.get_sublattice_compositions(sublattice, discard, thin_by, flat).var(axis=0)

    def get_orbit_factors(self, function_orbit_ids, discard=0, thin_by=1, flat=True):
        """Sum of natural_parameter * feature over the functions of each orbit id
        (container.py:269-280)."""
        vals = self.natural_parameters * self.get_feature_vectors(discard=discard, thin_by=thin_by, flat=flat)
        ids = np.asarray(function_orbit_ids)
        return np.array([np.sum(vals[..., ids == i]) for i in range(len(self.natural_parameters))])

    # ---- persistence -----------------------------------------------------------------------
    def to_npz(self, path):
        """Checkpoint (.npz stand-in for to_hdf5, container.py:615): one array per trace name +
        nsamples / total_mc_steps, like the HDF5 'trace' group."""
        np.savez_compressed(path, nsamples=self.num_samples, total_mc_steps=self._total_steps,
                            **{f"trace/{k}": v for k, v in self._all().items()})

    def get_sampled_species(self, indices, flat=True):
        """Species of every site for the samples ``indices`` -- what ``get_sampled_structures``
        (container.py:144-181) returns as pymatgen Structures, here as lists of species names in site
        order (``Processor.decode_occupancy``; lattice and coordinates are the supercell's own)."""
        indices = [indices] if isinstance(indices, (int, np.integer)) else list(indices)
        occupancies = self.get_occupancies(flat=flat)[indices]
        decode = self._ensemble.processor.decode_occupancy
        if flat:
            return [decode(occu) for occu in occupancies]
        return [[decode(occu) for occu in walkers] for walkers in occupancies]

    def get_sampled_structures(self, indices, flat=True):
        raise NotImplementedError("pymatgen Structures are not built here; get_sampled_species(indices) gives the "
                                  "species of every site, in the site order of the supercell")

    def to_hdf5(self, file_path):
        """container.py:615: HDF5 needs h5py, which this build does not depend on -- the same content
        goes to ``to_npz`` (one file) or to the streaming directory of ``get_backend``."""
        raise NotImplementedError("HDF5 output needs h5py; use to_npz(path) / from_npz, or Sampler.run(..., "
                                  "stream_chunk=n, stream_file=directory) and SampleContainer.from_stream")

    @classmethod
    def from_hdf5(cls, file_path, swmr_mode=False, ensemble=None):
        raise NotImplementedError("HDF5 input needs h5py; use from_npz(path, ensemble) or from_stream(directory, ensemble)")

    def as_dict(self):
        """JSON-able record of the samples (container.py:525-547): every traced value as nested
        lists with its dtype, the counters and the sampling metadata.  The ensemble itself is a set
        of flattened tables here, not a serialisable object: ``from_dict`` takes it as an argument."""
        return {
            "@class": self.__class__.__name__,
            "nsamples": self.num_samples,
            "total_mc_steps": self._total_steps,
            "metadata": _jsonable(self.metadata),
            "sublattices": [sub.as_dict() for sub in self.sublattices],
            "natural_parameters": np.asarray(self.natural_parameters).tolist(),
            "trace": {k: {"dtype": str(v.dtype), "data": v.tolist()} for k, v in self._all().items()},
        }

    @classmethod
    def from_dict(cls, d, ensemble):
        """Container of ``as_dict``'s record on ``ensemble`` (container.py:549-575)."""
        arrays = {k: np.array(v["data"], dtype=np.dtype(v["dtype"])) for k, v in d["trace"].items()}
        n = int(d["nsamples"])
        c = cls(ensemble, Trace(**{k: v[:0] for k, v in arrays.items()}), d.get("metadata"))
        if n:
            c.append_block({k: v[:n] for k, v in arrays.items()}, 0)
        c._total_steps = int(d["total_mc_steps"])
        return c

    # Streaming backend: a directory of part-NNNNNN.npz files + manifest.json.  It stands in for the
    # HDF5 file of the reference (container.py:420-437 flush_to_backend, :439-504 get_backend; h5py is
    # not available here): every flush writes the samples held in memory as the next part and drops
    # them, so a long run keeps at most one chunk on the host.  Appending to an existing directory
    # continues its numbering, as the reference appends to an existing file.
    def get_backend(self, file_path, alloc_nsamples=0, swmr_mode=False):
        return NpzStream(file_path, self)

    def flush_to_backend(self, backend):
        backend.write(self)
        self.clear()

    @classmethod
    def from_stream(cls, path, ensemble):
        """All parts of a streaming directory as one container (the reference's from_hdf5)."""
        man = NpzStream.read_manifest(path)
        c = None
        for i in range(man["nparts"]):
            d = np.load(os.path.join(path, f"part-{i:06d}.npz"))
            arrays = {k[6:]: d[k] for k in d.files if k.startswith("trace/")}
            if c is None:
                c = cls(ensemble, Trace(**{k: v[:0] for k, v in arrays.items()}), man.get("metadata"))
            c.append_block(arrays, 0)
        if c is None:
            raise ValueError(f"{path} holds no samples")
        c._total_steps = int(man["total_mc_steps"])
        return c

    @classmethod
    def from_npz(cls, path, ensemble):
        d = np.load(path)
        n = int(d["nsamples"])
        arrays = {k[6:]: d[k][:n] for k in d.files if k.startswith("trace/")}
        c = cls(ensemble, Trace(**{k: v[:0] for k, v in arrays.items()}))
        c.append_block(arrays, 0)
        c._total_steps = int(d["total_mc_steps"])
        return c


class NpzStream:
    """Writer side of the streaming directory (see SampleContainer.get_backend)."""

    def __init__(self, path, container):
        self.path = str(path)
        os.makedirs(self.path, exist_ok=True)
        if os.path.exists(os.path.join(self.path, "manifest.json")):
            self.manifest = self.read_manifest(self.path)
            if self.manifest["traced_values"] != list(container.traced_values) or \
                    self.manifest["shape"] != list(container.shape):
                raise RuntimeError(
                    f"Backend file {self.path} holds samples of another shape / set of traced values."
                )
        else:
            self.manifest = dict(nparts=0, nsamples=0, total_mc_steps=0, shape=list(container.shape),
                                 traced_values=list(container.traced_values),
                                 metadata=_jsonable(container.metadata))
            self._save_manifest()

    @staticmethod
    def read_manifest(path):
        with open(os.path.join(path, "manifest.json")) as f:
            return json.load(f)

    def _save_manifest(self):
        tmp = os.path.join(self.path, "manifest.json.tmp")
        with open(tmp, "w") as f:
            json.dump(self.manifest, f)
        os.replace(tmp, os.path.join(self.path, "manifest.json"))  # readers never see a torn manifest

    def write(self, container):
        n = container.num_samples
        if n == 0:
            return
        np.savez(os.path.join(self.path, f"part-{self.manifest['nparts']:06d}.npz"),
                 **{f"trace/{k}": v for k, v in container._all().items()})
        self.manifest["nparts"] += 1
        self.manifest["nsamples"] += n
        self.manifest["total_mc_steps"] += container.total_mc_steps
        self._save_manifest()

    def close(self):
        pass


def _jsonable(x):
    if isinstance(x, dict):
        return {str(k): _jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_jsonable(v) for v in x]
    if isinstance(x, np.ndarray):
        return x.tolist()
    if isinstance(x, (np.integer, np.floating, np.bool_)):
        return x.item()
    return x if isinstance(x, (str, int, float, bool, type(None))) else repr(x)


def _install_reductions():
    """get_X / mean_X / X_variance / get_minimum_X for the scalar-like traced values: the
    reference writes these ~20 methods out one by one (container.py:201-333); they all are one
    selection followed by one NumPy reduction."""
    plural = {"occupancy": "occupancies", "enthalpy": "enthalpies", "features": "feature_vectors"}
    singular = {"enthalpy": "enthalpy", "features": "feature_vector", "energy": "energy"}

    def selector(field):
        def get(self, discard=0, thin_by=1, flat=True):
            return self.get_trace_value(field, discard, thin_by, flat)
        return get

    for field, name in plural.items():
        fn = selector(field)
        fn.__name__ = f"get_{name}"
        setattr(SampleContainer, fn.__name__, fn)

    def reducer(getter_name, how):
        def red(self, discard=0, thin_by=1, flat=True):
            return getattr(getattr(self, getter_name)(discard, thin_by, flat), how)(axis=0)
        return red

    for field, name in singular.items():
        getter = "get_energies" if field == "energy" else f"get_{plural[field]}"
        for pattern, how in ((f"mean_{name}", "mean"), (f"{name}_variance", "var")):
            fn = reducer(getter, how)
            fn.__name__ = pattern
            setattr(SampleContainer, pattern, fn)
        if field != "features":
            fn = reducer(getter, "min")
            fn.__name__ = f"get_minimum_{name}"
            setattr(SampleContainer, fn.__name__, fn)


_install_reductions()


# --------------------------------------------------------------------------- #
class Sampler:
    """MCMC driver (smol/moca/sampler/sampler.py:22-445) on the batched GPU engine.

    Multi-GPU: walkers are independent (sampler.py:111-116,436-440), so a Sampler built with
    ``rank`` / ``world_size`` (default: taken from an initialised torch.distributed process
    group, else a single rank) owns only its contiguous block ``walker_range`` of the
    ``nwalkers`` global walkers on the device of its rank.  Seeds and initial occupancies are
    indexed by GLOBAL walker, so walker g runs the same chain whatever the world size;
    ``global_statistics`` all-reduces the running sums over ranks (RCCL), the only collective."""

    def __init__(self, kernels, container, engine=None, walker_range=None, device=0, world_size=1,
                 _bind=True):
        self._kernels = kernels
        if _bind:  # kernel-level accessors (WangLandau.dos, MCKernel.trace) read this sampler's engine
            for w, k in enumerate(kernels):
                k._owner = (self, w)
        self._container = container
        self._container.metadata["kernels"] = [k.spec for k in kernels]
        self._engine = engine
        self._engine_key = None
        self._walker_range = walker_range or (0, len(kernels))
        self._device, self._world = int(device), int(world_size)
        self._state_loaded = False  # walker states uploaded to the engine at least once
        self._resume_at = None      # (container, its sample count) the device state equals, see _load_state
        self._kept_last = False     # the container holds the one sample a streamed run kept (keep_last_chunk)

    @classmethod
    def from_ensemble(cls, ensemble, *args, step_type=None, kernel_type=None, seeds=None,
                      nwalkers=1, rank=None, world_size=None, device=None, **kwargs):
        """sampler.py:52-139: default step 'flip' when chemical potentials are set else
        'swap'; default kernel Metropolis; one kernel (seed) per walker."""
        from . import parallel

        if step_type is None:
            step_type = "flip" if ensemble.chemical_potentials is not None else "swap"
        if kernel_type is None:
            kernel_type = "Metropolis"
        if seeds is not None and len(seeds) != nwalkers:
            raise ValueError("Number of seeds does not match number of kernels!")
        if rank is None or world_size is None:
            rank, world_size = parallel.rank_and_world()
        first, count = parallel.shard(nwalkers, rank, world_size)
        if count == 0:
            raise ValueError(f"rank {rank} of {world_size} would own no walker: nwalkers={nwalkers}")
        local_seeds = [None] * count if seeds is None else list(seeds[first:first + count])
        kernels = [mckernel_factory(kernel_type, ensemble, step_type, *args, seed=s, **kwargs)
                   for s in local_seeds]
        container = cls._container_for(ensemble, kernels[0], count)
        container.metadata["walker_range"] = (first, count, nwalkers)
        if device is None:
            device = parallel.local_device(rank)
        return cls(kernels, container, walker_range=(first, count), device=device, world_size=world_size)

    @staticmethod
    def _container_for(ensemble, k0, count):
        """Empty SampleContainer with the trace schema of kernel ``k0`` for ``count`` walkers."""
        F = len(ensemble.natural_parameters)
        # occupancies are STORED as the device ring's bytes (a quarter of the int32 the reference
        # keeps) and handed out as int32 by the container's getters
        per_walker = dict(occupancy=((ensemble.num_sites,), np.uint8), features=((F,), np.float64),
                          enthalpy=((1,), np.float64))
        if isinstance(k0, Metropolis):
            per_walker["temperature"] = ((1,), np.float64)
        per_walker["accepted"] = ((1,), bool)
        if k0.bias is not None:  # trace.bias (kernel/base.py:362-363)
            per_walker["bias"] = ((1,), np.float64)
        if isinstance(k0, WangLandau):  # wanglandau.py:268-288
            L = len(k0._levels)
            per_walker.update(histogram=((L,), np.int64), occurrences=((L,), np.int64),
                              entropy=((L,), np.float64), cumulative_mean_features=((L, F), np.float64),
                              mod_factor=((1,), np.float64))
        schema = Trace(**{k: np.empty((0, count) + shp, dtype=dt) for k, (shp, dt) in per_walker.items()})
        return SampleContainer(ensemble, schema, ensemble.thermo_boundaries)

    # -- accessors -----------------------------------------------------------------
    mckernels = property(lambda self: self._kernels)
    seeds = property(lambda self: [k.seed for k in self._kernels])
    samples = property(lambda self: self._container)
    walker_range = property(lambda self: self._walker_range)

    @property
    def engine(self):
        return self._get_engine()

    def efficiency(self, discard=0, flat=True):
        return self.samples.sampling_efficiency(discard=discard, flat=flat)

    def clear_samples(self):
        self.samples.clear()

    def global_statistics(self, discard=0):
        """Sums over ALL walkers of ALL ranks of the recorded samples: walkers, samples, mean
        enthalpy, enthalpy variance, acceptance.  One all-reduce of five float64 (identity
        without a process group)."""
        import torch

        from . import parallel

        H = self.samples.get_enthalpies(discard=discard, flat=False).astype(np.float64)
        acc = self.samples.get_trace_value("accepted", discard=discard, flat=False)
        local = torch.tensor([float(H.shape[1]), float(H.size), float(H.sum()), float((H * H).sum()),
                              float(acc.sum())], dtype=torch.float64)
        if parallel.collective_device() == "cuda":
            local = local.cuda(self._device)
        w, n, s1, s2, a = parallel.global_sums(local).cpu().numpy()
        mean = s1 / max(n, 1.0)
        return dict(walkers=int(w), samples=int(n), mean_enthalpy=mean,
                    enthalpy_variance=s2 / max(n, 1.0) - mean * mean, acceptance=a / max(n, 1.0))

    # -- engine plumbing -------------------------------------------------------------
    def _model_key(self):
        """Content key of everything baked into the engine handle: the chemical-potential table,
        the active sites of every sublattice and the bias table (an id()/count key would miss a
        different restriction of the same size or a recycled id)."""
        import hashlib

        k0 = self._kernels[0]
        ens = k0.ensemble
        h = hashlib.blake2b(digest_size=16)
        mu = ens._mu_table
        h.update(b"mu" if mu is None else np.ascontiguousarray(mu).tobytes())
        for sub in ens.sublattices:
            h.update(np.ascontiguousarray(sub.active_sites, dtype=np.int64).tobytes() + b"|")
            h.update(np.ascontiguousarray(sub.encoding, dtype=np.int64).tobytes() + b"/")
        if k0.bias is not None:
            h.update(np.ascontiguousarray(k0.bias._table).tobytes())
            h.update(repr((k0.bias.bias_type, k0.bias.penalty)).encode())
            icpt = getattr(k0.bias, "intercepts", None)
            if icpt is not None:
                h.update(b"icpt" + np.ascontiguousarray(icpt, dtype=np.float64).tobytes())
        for name in sorted(k0.usher_kwargs):  # flip table / weights / swap_weight are baked in too
            v = k0.usher_kwargs[name]
            h.update(name.encode() + (b"none" if v is None else np.ascontiguousarray(v, dtype=np.float64).tobytes()))
        return h.hexdigest(), self._device

    def _get_engine(self, device=None):
        if device is not None:
            self._device = int(device)
        k0 = self._kernels[0]
        ens = k0.ensemble
        key = self._model_key()
        if self._engine is None or self._engine_key != key:
            tables = ens.make_tables(**k0.usher_kwargs)
            if k0.bias is not None:
                tables.set_bias(k0.bias.bias_type, k0.bias._table, k0.bias.penalty,
                                intercepts=getattr(k0.bias, "intercepts", None))
            if isinstance(k0, WangLandau):
                # (a callable mod_update: flatness checks on the host, check period 0 switches the
                # device's own check off)
                cfg = capi.make_config(
                    len(self._kernels), capi.KERNEL_WANGLANDAU, STEP_TYPES[k0.step_type], self._device,
                    min_enthalpy=k0._window[0], max_enthalpy=k0._window[1], bin_size=k0._window[2],
                    flatness=k0.flatness, mod_factor=k0._m0, mod_update=k0._mod_divisor,
                    check_period=0 if k0._mod_callable is not None else k0.check_period,
                    update_period=k0.update_period,
                )
            else:
                cfg = capi.make_config(len(self._kernels), capi.KERNEL_METROPOLIS,
                                       STEP_TYPES[k0.step_type], self._device)
            self._engine = Engine(tables, cfg)
            self._engine_key = key
            self._state_loaded = False
        return self._engine

    def _temperatures(self):
        """Temperature of every walker as the engine wants it (Wang-Landau: unused; UniformlyRandom:
        infinite, i.e. beta = 0)."""
        return np.array([getattr(k, "temperature", getattr(k, "engine_temperature", 0.0)) for k in self._kernels],
                        dtype=np.float64)

    def _local_occupancies(self, occupancies):
        """Accepts this rank's walkers (count, N), all walkers (nwalkers, N) -- the rank takes
        its block -- or a 1-D occupancy for a single walker (sampler.py:442-end)."""
        occ = np.asarray(occupancies)
        first, count = self._walker_range
        nglobal = self.samples.metadata.get("walker_range", (0, count, count))[2]
        N = self.samples.shape[1]
        if occ.shape == (count, N):
            pass
        elif occ.shape == (nglobal, N):
            occ = occ[first:first + count]
        elif occ.ndim == 1 and count == 1 and occ.shape[0] == N:
            occ = occ.reshape(1, N)
        else:
            raise AttributeError(
                "The given initial occupancies have incompompatible dimensions. Shape should be "
                f"{self.samples.shape}."
            )
        return occ.astype(np.int32, copy=False)  # (the engine copies it to the device)

    def _load_state(self, initial_occupancies):
        """Walker states on the device.  ``None`` = continue from the last recorded sample: that
        IS the device state when nothing else touched the engine or the container since the
        last run (the steps after the last sample of a run are never taken), so nothing is
        uploaded -- only the temperatures are refreshed (anneal changes them between runs)."""
        eng = self._get_engine()
        if initial_occupancies is None:
            if self._state_loaded and self._resume_at == (id(self.samples), self.samples.num_samples):
                eng.set_temperature(self._temperatures())
                return None
            initial_occupancies = self.samples.last_occupancy()
        occupancies = self._local_occupancies(initial_occupancies)
        seeds = np.array([k.seed64 for k in self._kernels], dtype=np.uint64)
        # a kernel's Generator, accept counters and WL aux arrays persist across run()
        # calls of one sampler (sampler.py:254-262): only the first call starts fresh
        eng.set_state(occupancies, seeds, self._temperatures(), reset_aux=not self._state_loaded)
        self._state_loaded = True
        return occupancies

    def setup_sample(self, initial_occupancies):
        """sampler.py:386-434: copy / reshape occupancies, set aux states, initial trace."""
        occupancies = self._load_state(initial_occupancies)
        return occupancies, self._current_trace(self._get_engine())

    def _current_trace(self, eng):
        st = eng.get_state()
        nw = len(self._kernels)
        tr = Trace(occupancy=st["occupancy"], features=st["features"],
                   enthalpy=st["enthalpy"].reshape(nw, 1))
        k0 = self._kernels[0]
        if isinstance(k0, Metropolis):
            tr.temperature = self._temperatures().reshape(nw, 1)
        tr.accepted = st["accepted"].reshape(nw, 1)
        if k0.bias is not None:
            tr.bias = eng.get_bias().reshape(nw, 1)
        if isinstance(k0, WangLandau):
            wl = eng.get_wl()
            tr.histogram, tr.occurrences, tr.entropy = wl["histogram"], wl["occurrences"], wl["entropy"]
            tr.cumulative_mean_features = wl["mean_features"]
            tr.mod_factor = wl["mod_factor"].reshape(nw, 1)
        return tr

    def _sample_blocks(self, nsteps, initial_occupancies, thin_by, max_block=0, state_loaded=False):
        """Generator over blocks of thinned samples, dict name -> (n, nwalkers, ...): the unit
        the device ring delivers (``smolmc_run_sampled``): the samples of a block are recorded on
        the device -- inside one launch for Metropolis kernels, as launch + snapshot pairs queued without
        a host round trip for biased and Wang-Landau kernels (trace.bias; the per-walker L and L x F
        arrays of the Wang-Landau trace) -- and the block's download overlaps the next block's kernel."""
        if nsteps % thin_by != 0:
            warnings.warn(
                f"The number of steps {nsteps} is not a multiple of thin_by  {thin_by}. The last "
                f"{nsteps % thin_by} will be ignored.",
                category=RuntimeWarning,
            )
        if not state_loaded:
            self._load_state(initial_occupancies)
        self._resume_at = None
        eng = self._get_engine()
        nsamples = nsteps // thin_by
        k0 = self._kernels[0]
        is_wl, has_bias = isinstance(k0, WangLandau), k0.bias is not None
        if is_wl and k0._mod_callable is not None:
            # a callable mod_update (wanglandau.py:100-105): the flatness checks run on the host between
            # launches, so these samples come back one launch at a time
            for _ in range(nsamples):
                self._wl_run_with_host_checks(eng, thin_by)
                yield {k: v[None] for k, v in self._current_trace(eng).items()}
            return
        nw, N = len(self._kernels), k0.ensemble.num_sites
        # bytes of one sample of all walkers in the ring: occupancy bytes + features (+ the Wang-Landau trace:
        # entropy / histogram / occurrences [L] and the mean features [L x F] of every walker, wanglandau.py:247-251)
        F = len(k0.ensemble.natural_parameters)
        per_sample = nw * (N + 8 * F + 17)
        if is_wl:
            L = len(k0._levels)
            per_sample += nw * L * (24 + 8 * F)
        per_block = max(1, min(nsamples, (256 << 20) // max(1, per_sample)))  # <= ~256 MiB per ring slot
        # ... and several blocks per run, so that the download of one overlaps the kernel of the next (the ring
        # has two slots: block k + 1 is queued before block k is fetched)
        if nsamples * per_sample > (32 << 20):
            per_block = min(per_block, max(1, -(-nsamples // 8)))
        if max_block > 0:
            per_block = min(per_block, int(max_block))
        temps = self._temperatures().reshape(1, nw, 1)
        sizes = [min(per_block, nsamples - start) for start in range(0, nsamples, per_block)]

        def queue(n):
            eng.run_sampled_async(n, thin_by, occupancy=True, bias=has_bias, wl=is_wl)

        # Block i + 1 is queued before block i is handed out, so a consumer that stops early -- a `break` out of
        # Sampler.sample(), an exception in run() between two blocks -- leaves a block on the (cached) engine's ring.
        # The ring is emptied when this generator starts and when it ends, however it ends: the next run's first
        # fetch is its own first block.
        eng.discard_samples()
        try:
            yield from self._ring_blocks(eng, sizes, queue, temps, nw, k0, has_bias, is_wl)
        finally:
            eng.discard_samples()

    @staticmethod
    def _ring_blocks(eng, sizes, queue, temps, nw, k0, has_bias, is_wl):
        if sizes:
            queue(sizes[0])
        for i, n in enumerate(sizes):
            if i + 1 < len(sizes):
                queue(sizes[i + 1])
            ring = eng.fetch_samples(packed=True)
            block = dict(occupancy=ring["occupancy"], features=ring["features"],
                         enthalpy=ring["enthalpy"][..., None], temperature=np.broadcast_to(temps, (n, nw, 1)),
                         accepted=ring["accepted"][..., None])
            if not isinstance(k0, Metropolis):  # (UniformlyRandom, Wang-Landau: no temperature in the trace)
                del block["temperature"]
            if has_bias:
                block["bias"] = ring["bias"][..., None]
            if is_wl:
                block.update(histogram=ring["histogram"], occurrences=ring["occurrences"], entropy=ring["entropy"],
                             cumulative_mean_features=ring["mean_features"], mod_factor=ring["mod_factor"][..., None])
            yield block

    def _wl_run_with_host_checks(self, eng, nsteps):
        """``nsteps`` Wang-Landau steps with the flatness check of wanglandau.py:253-264 done on the
        host, so that ``mod_update`` can be any callable (:100-105): the launches end where the step
        counter reaches a multiple of ``check_period``; there every walker whose histogram is flat
        (all visited bins above ``flatness`` times their mean, at least two visited) gets its
        histogram cleared and its modification factor mapped through the callable."""
        k0 = self._kernels[0]
        period = int(k0.check_period)
        done = int(eng.get_state(occupancy=False)["n_steps"][0])  # (all walkers of a sampler step together)
        left = int(nsteps)
        while left > 0:
            n = min(left, period - done % period)
            eng.run(n)
            done += n
            left -= n
            if done % period == 0:
                wl = eng.get_wl()
                S, hist, m = wl["entropy"], wl["histogram"], wl["mod_factor"]
                changed = False
                for r in range(len(m)):
                    seen = S[r] > 0
                    if seen.sum() >= 2 and (hist[r][seen] > k0.flatness * hist[r][seen].mean()).all():
                        hist[r] = 0
                        m[r] = self._kernels[r]._mod_callable(m[r])
                        changed = True
                if changed:
                    eng.set_wl(histogram=hist, mod_factor=m)

    def aux_checkpoint(self):
        """Everything beyond the occupancies that a run needs to continue bit-for-bit in another
        process -- the reference reserves ``aux_checkpoint`` for this and never fills it
        (sampler/container.py:89,539-541): step / accept counters (the walkers' positions in their
        random streams) and, for Wang-Landau, the entropy / histogram / occurrences / mean-feature
        arrays and the modification factors.  Restore with ``restore_aux`` after ``setup_sample`` (or
        a first ``run``) with the checkpoint's ``occupancy``."""
        eng = self._get_engine()
        st = eng.get_state()
        ck = dict(occupancy=st["occupancy"], n_steps=st["n_steps"], n_accepted=st["n_accepted"])
        if isinstance(self._kernels[0], WangLandau):
            ck.update({"wl_" + k: v for k, v in eng.get_wl().items()})
        return ck

    def restore_aux(self, checkpoint):
        """Put a checkpoint of ``aux_checkpoint`` back in a (possibly fresh) engine handle: occupancies
        and seeds with ``reset_aux=True`` -- the seeds only travel with a reset, smolmc.h -- then the
        checkpoint's counters and Wang-Landau arrays on top of the reset state."""
        eng = self._get_engine()
        seeds = np.array([k.seed64 for k in self._kernels], dtype=np.uint64)
        eng.set_state(self._local_occupancies(checkpoint["occupancy"]), seeds, self._temperatures(), reset_aux=True)
        eng.set_counters(checkpoint["n_steps"], checkpoint["n_accepted"])
        if isinstance(self._kernels[0], WangLandau):
            eng.set_wl(checkpoint["wl_entropy"], checkpoint["wl_histogram"], checkpoint["wl_occurrences"],
                       checkpoint["wl_mean_features"], checkpoint["wl_mod_factor"])
        self._state_loaded = True
        self._resume_at = (id(self.samples), self.samples.num_samples)

    def sample(self, nsteps, initial_occupancies, thin_by=1, progress=False):
        """Generator over thinned traces, one Trace per sample (sampler.py:164-210)."""
        for block in self._sample_blocks(nsteps, initial_occupancies, thin_by):
            for i in range(len(block["occupancy"])):
                tr = Trace(**{k: np.asarray(v[i]) for k, v in block.items()})
                tr.occupancy = tr.occupancy.astype(np.int32)  # (trace.occupancy is int32 in the reference)
                yield tr

    def run(self, nsteps, initial_occupancies=None, thin_by=1, progress=False, stream_chunk=0,
            stream_file=None, keep_last_chunk=False, swmr_mode=False):
        """sampler.py:212-301.  ``stream_chunk`` > 0 writes every chunk of samples to the streaming
        directory ``stream_file`` (see SampleContainer.get_backend) and keeps none in memory."""
        if initial_occupancies is None:
            if self.samples.num_samples == 0:
                raise RuntimeError(
                    "There are no saved samples to obtain the initial occupancies."
                    "These must be provided."
                )
        elif self.samples.num_samples > 0:
            warnings.warn(
                "Initial occupancies where provided with a pre-existing set of samples.\n Make "
                "real sure that is what you want. If not, reset the samples in the sampler.",
                RuntimeWarning,
            )
        backend, last = None, None
        if stream_chunk > 0:  # sampler.py:264-301; the backend is a directory of .npz parts here
            if stream_file is None:
                stream_file = os.path.join(os.getcwd(), "moca-samples-" + datetime.now().strftime("%Y-%m-%d-%H%M%S%f"))
            backend = self.samples.get_backend(stream_file, nsteps // thin_by, swmr_mode=swmr_mode)
        self._load_state(initial_occupancies)
        if backend is not None and self._kept_last:
            self.samples.clear()  # the sample kept by the previous streamed run is already in its stream
        self._kept_last = False
        for block in self._sample_blocks(nsteps, None, thin_by, max_block=stream_chunk, state_loaded=True):
            self.samples.append_block(block, thinned_by=thin_by)
            if backend is not None:
                # the most recent sample, whether or not this block fills a chunk: the tail flush
                # below must leave the FINAL recorded sample behind, not the end of the last full chunk
                last = {k: np.array(v[-1:]) for k, v in block.items()}
                if self.samples.num_samples >= stream_chunk:
                    self.samples.flush_to_backend(backend)
        if backend is not None:
            self.samples.flush_to_backend(backend)  # (the tail shorter than a chunk)
            backend.close()
            if keep_last_chunk and last is not None:  # the last sample stays in memory, e.g. to start the next run from it
                self.samples.append_block(last, thinned_by=0)
                self._kept_last = True
        # the device now holds the last recorded sample (see _load_state)
        self._resume_at = (id(self.samples), self.samples.num_samples)

    def anneal(self, temperatures, mcmc_steps, initial_occupancies=None, thin_by=1, progress=False,
               **kwargs):
        """Simulated annealing (sampler.py:303-384): one ``run`` per temperature, each continuing
        from the last sample of the previous one."""
        if not isinstance(self._kernels[0], Metropolis):
            raise AttributeError("anneal is only available for samplers with a thermal kernel")
        if temperatures[0] < temperatures[-1]:
            raise ValueError(
                "End temperature is greater than start temperature "
                f"{temperatures[-1]:.2f} > {temperatures[0]:.2f}."
            )
        start = initial_occupancies
        for temperature in temperatures:
            for kernel in self._kernels:
                kernel.temperature = temperature
            self.run(mcmc_steps, initial_occupancies=start, thin_by=thin_by, progress=progress)
            start = None
