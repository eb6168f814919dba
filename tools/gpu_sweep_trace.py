"""GPU: SCF trajectory of one system of tests/golden/sweep_systems.json, native cycle beside the NumPy oracle (usage: python tools/gpu_sweep_trace.py TAG)."""
import sys, json, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from conftest import atom_arrays
from oracle import oracle as orc, scf_oracle as so
from tuna_amd import molecule as mol, spherical
from tuna_amd.engine import Engine
g=json.load(open('/root/repo/tests/golden/sweep_systems.json'))[sys.argv[1] if len(sys.argv) > 1 else 'lif_631g']
atoms=mol.make_atoms(g['symbols'], mol.angstrom_to_bohr(g['R_angstrom'])); shells=mol.build_shells(atoms,g['basis']); aos=mol.expand_cartesian_aos(shells)
U=spherical.transformation_matrix([s.L for s in shells])
xyz,chg,org=atom_arrays(atoms)
So,To,Vo,_,_=orc.one_electron(aos,xyz,chg,org)
So,To,Vo=[U@M@U.T for M in (So,To,Vo)]
E=so.eri_to_spherical(U,orc.eri(aos))
ranges=[sum(s.n_sph for s in shells if s.atom==a) for a in range(len(atoms))]
with Engine(0) as eng:
    eng.set_basis(aos).build_eri(True)
    S,T,V,_,_=eng.one_electron(xyz,chg,org,spherical=True)
    print("max diff S,T,V", abs(S-So).max(), abs(T-To).max(), abs(V-Vo).max())
    X,_,_=eng.orthogonaliser(S)
    Xo,_,_=so.orthogonaliser(So)
    print("X diff", abs(X-Xo).max())
    P0,E0=so.core_guess(T,V,X,g['n_occ'])
    r=eng.scf_rhf(S,T,V,P0,E0,g['n_occ'],mol.nuclear_repulsion(atoms),X=X,conv="extreme",damping="dynamic",n_atom_ao=ranges)
    ro=so.run_rhf(So,To,Vo,E,Xo,P0,E0,g['n_occ'],mol.nuclear_repulsion(atoms),ranges,conv="extreme",damping="dynamic")
    print(r['n_iter'], ro['n_iter'])
    n=max(r['n_iter'],ro['n_iter'])
    for k in range(n):
        a=r['table'][k] if k<r['n_iter'] else None; b=ro['table'][k] if k<ro['n_iter'] else None
        print(k+1, "%.12f"%a[1] if a is not None else "-", "%.12f"%b[1] if b is not None else "-", "%.2e"%(a[1]-b[1]) if a is not None and b is not None else "", "damp %.3f %.3f"%(a[6] if a is not None else -1, b[6] if b is not None else -1))
