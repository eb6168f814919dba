"""Atoms, basis-set lookup and the ordered Cartesian AO list for z-axis diatomics.

This is the *input contract* of the hot path: it reproduces the AO ordering and the
geometry conventions of the reference so that the tensors produced by the HIP engine
line up index-for-index with the reference's.

Reference behaviour mirrored (h-brough/TUNA v0.12.0, all paths under /root/reference/TUNA):
  * basis-name mangling and lookup       tuna_basis.py:186-236  (generate_basis)
  * AO list order                        tuna_molecule.py:532-587 (form_basis)
  * Cartesian component order in a shell tuna_molecule.py:597-624 (convert_angular_momentum_to_subshell)
  * geometry: atom A at the origin, atom B at (0,0,R) in bohr
                                         tuna_util.py:845-878 (clean_coordinates), :38-57 (constants)
  * ghost atoms "X<sym>": basis functions but no charge   tuna_molecule.py:76-79
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from functools import lru_cache

import numpy as np

# CODATA-2022 derived, exactly as the reference builds it (tuna_util.py:38-57):
_h = 6.62607015e-34
_e = 1.602176634e-19
_me = 9.1093837139e-31
_eps0 = 8.8541878188e-12
_hbar = _h / (2 * np.pi)
BOHR_IN_METRES = 4 * np.pi * _eps0 * _hbar ** 2 / (_me * _e ** 2)
BOHR_RADIUS_IN_ANGSTROM = BOHR_IN_METRES * 10 ** 10  # 0.5291772105443463


def angstrom_to_bohr(r: float) -> float:
    return r / BOHR_RADIUS_IN_ANGSTROM


SYMBOLS = ["H", "HE", "LI", "BE", "B", "C", "N", "O", "F", "NE", "NA", "MG", "AL", "SI", "P", "S", "CL", "AR"]
ATOMIC_NUMBER = {s: i + 1 for i, s in enumerate(SYMBOLS)}
SHELL_LETTERS = "SPDFGH"

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "basis_sets.json")


def mangle_basis_name(basis_set: str) -> str:
    """tuna_basis.py:201-215 -- 'cc-pVTZ' -> 'CC_PVTZ', '6-31G*' -> '_6_31GSTAR' ..."""
    bas = (basis_set.upper().replace("-", "_").replace("*", "STAR").replace("+", "PLUS").replace("[", "BRA")
           .replace("(", "BRA").replace(",", "COMMA").replace("]", "KET").replace(")", "KET"))
    if bas[0].isdigit():
        bas = "_" + bas
    return bas


@lru_cache(maxsize=1)
def _basis_tables() -> dict:
    with open(_DATA) as f:
        return json.load(f)


def available_basis_sets() -> list[str]:
    return sorted(_basis_tables())


def atomic_basis(basis_set: str, Z: int) -> list:
    """[(letter, [(exp, coef), ...]), ...] in the order the reference lists them."""
    tables = _basis_tables()
    key = mangle_basis_name(basis_set)
    if key not in tables:
        raise KeyError(f"basis set {basis_set!r} is not shipped with tuna_amd (have: {', '.join(sorted(tables))})")
    entry = tables[key].get(str(Z))
    if entry is None:
        raise KeyError(f"The chosen basis set, {basis_set}, is not parameterised for Z={Z}!")
    return [(L, [(float(e), float(c)) for e, c in prims]) for L, prims in entry]


def cartesian_components(L: int) -> list[tuple[int, int, int]]:
    """x^L ... y^L ... z^L order of tuna_molecule.py:622."""
    return [(i, j, L - i - j) for i in range(L, -1, -1) for j in range(L - i, -1, -1)]


@dataclass
class Atom:
    symbol: str
    Z: int            # basis charge (element whose basis functions sit here)
    charge: int       # nuclear charge (0 for ghost atoms)
    origin: np.ndarray


@dataclass
class Shell:
    atom: int
    origin: np.ndarray
    L: int
    exps: np.ndarray
    coefs: np.ndarray   # raw contraction coefficients (un-normalised)

    @property
    def n_cart(self) -> int:
        return (self.L + 1) * (self.L + 2) // 2

    @property
    def n_sph(self) -> int:
        return 2 * self.L + 1


@dataclass
class AOList:
    """Flat description of the Cartesian AO list -- exactly what the C ABI takes (include/tunafock.h)."""
    origin: np.ndarray      # f64 [n,3]
    lmn: np.ndarray         # i32 [n,3]
    nprim: np.ndarray       # i32 [n]
    prim_off: np.ndarray    # i32 [n+1]
    exps: np.ndarray        # f64 [sum nprim]
    coefs: np.ndarray       # f64 [sum nprim]   raw coefficients
    shell_of_ao: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))

    @property
    def n(self) -> int:
        return int(self.origin.shape[0])


def make_atoms(symbols: list[str], R_bohr: float | None) -> list[Atom]:
    atoms = []
    for k, s in enumerate(symbols):
        s = s.upper()
        ghost = s.startswith("X") and s[1:] in ATOMIC_NUMBER
        el = s[1:] if ghost else s
        if el not in ATOMIC_NUMBER:
            raise KeyError(f"unknown atom {s!r} (H..Ar supported, tuna_util.py:1678-1912)")
        z = 0.0 if k == 0 else float(R_bohr)
        atoms.append(Atom(s, ATOMIC_NUMBER[el], 0 if ghost else ATOMIC_NUMBER[el], np.array([0.0, 0.0, z])))
    if len(atoms) not in (1, 2):
        raise ValueError("atoms and diatomics only")
    return atoms


def build_shells(atoms: list[Atom], basis_set: str | dict, decontract: bool = False) -> list[Shell]:
    """Shell list in AO order.  `basis_set` may be a name or {Z: [(letter, [(e,c)..]), ...]}.

    DECONTRACT (tuna_molecule.py:564-570) in the reference emits, for each Cartesian component, one
    AO per primitive; we emit one single-primitive shell per primitive instead, which is the shell-ordered
    equivalent used by the synthetic scaling series.
    """
    shells: list[Shell] = []
    for ia, atom in enumerate(atoms):
        table = basis_set[atom.Z] if isinstance(basis_set, dict) else atomic_basis(basis_set, atom.Z)
        for letter, prims in table:
            L = SHELL_LETTERS.find(letter.upper())
            if L < 0:
                raise ValueError('Only up to "H" type basis functions are implemented!')
            e = np.array([p[0] for p in prims], dtype=np.float64)
            c = np.array([p[1] for p in prims], dtype=np.float64)
            if decontract:
                for ek in e:
                    shells.append(Shell(ia, atom.origin.copy(), L, np.array([ek]), np.array([1.0])))
            else:
                shells.append(Shell(ia, atom.origin.copy(), L, e, c))
    return shells


def expand_cartesian_aos(shells: list[Shell]) -> AOList:
    """One entry per Cartesian AO, every AO carrying its own copy of the primitives (tuna_molecule.py:574)."""
    origin, lmn, nprim, exps, coefs, shell_of = [], [], [], [], [], []
    for si, sh in enumerate(shells):
        for comp in cartesian_components(sh.L):
            origin.append(sh.origin)
            lmn.append(comp)
            nprim.append(len(sh.exps))
            exps.append(sh.exps)
            coefs.append(sh.coefs)
            shell_of.append(si)
    nprim = np.asarray(nprim, dtype=np.int32)
    prim_off = np.zeros(len(nprim) + 1, dtype=np.int32)
    np.cumsum(nprim, out=prim_off[1:])
    return AOList(np.ascontiguousarray(origin, dtype=np.float64).reshape(-1, 3),
                  np.ascontiguousarray(lmn, dtype=np.int32).reshape(-1, 3), nprim, prim_off,
                  np.concatenate(exps).astype(np.float64), np.concatenate(coefs).astype(np.float64),
                  np.asarray(shell_of, dtype=np.int32))


def even_tempered_basis(n_s: int, n_p: int, n_d: int, n_f: int, ratio: float = 2.5,
                        alpha0=(0.05, 0.08, 0.15, 0.30)) -> list:
    """Synthetic uncontracted even-tempered basis of SURVEY.md section 8d: alpha_k = alpha0_l * ratio^k."""
    out = []
    for L, n in enumerate((n_s, n_p, n_d, n_f)):
        for k in range(n):
            out.append((SHELL_LETTERS[L], [(alpha0[L] * ratio ** k, 1.0)]))
    return out


def synthetic_counts(n_sph_total: int) -> tuple[int, int, int, int]:
    """(n_s,n_p,n_d,n_f) per atom for the synthetic series, scaled from the 20s15p13d10f (=400) worked example."""
    per_atom = n_sph_total / 2.0
    f = per_atom / 200.0
    n_s, n_p, n_d, n_f = (max(1, round(20 * f)), max(1, round(15 * f)), max(0, round(13 * f)), max(0, round(10 * f)))
    # nudge s count so that the total matches as closely as possible
    def tot(a, b, c, d):
        return a + 3 * b + 5 * c + 7 * d
    n_s = max(1, n_s + int(round(per_atom - tot(n_s, n_p, n_d, n_f))))
    return n_s, n_p, n_d, n_f


def electron_count(atoms: list[Atom], charge: int = 0) -> int:
    return int(sum(a.charge for a in atoms)) - charge


def nuclear_repulsion(atoms: list[Atom]) -> float:
    """tuna_kernel.py:741"""
    if len(atoms) < 2:
        return 0.0
    return float(atoms[0].charge * atoms[1].charge / np.linalg.norm(atoms[1].origin - atoms[0].origin))
