"""FragPart -- the index lists a fragmentation produces (molbe/autofrag.py:38-130), as plain data.

Fragmentation itself (autogen / chemgen / graphgen) is upstream of the hot path and stays in QuEmb; this class
only carries its OUTPUT.  Any object with the same attribute names (e.g. the reference's own FragPart) is
accepted by `quemb_amd.mbe.BE`.
"""

from __future__ import annotations

import json
from dataclasses import dataclass, field
from pathlib import Path


@dataclass
class FragPart:
    AO_per_frag: list
    AO_per_edge_per_frag: list
    ref_frag_idx_per_edge_per_frag: list
    relAO_per_origin_per_frag: list
    weight_and_relAO_per_center_per_frag: list
    relAO_per_edge_per_frag: list = field(default_factory=list)
    relAO_in_ref_per_edge_per_frag: list = field(default_factory=list)
    n_BE: int = 2
    frozen_core: bool = False
    iao_valence_basis: object = None
    iao_valence_only: bool = False
    ncore: int | None = None              # molbe/autofrag.py:127-145 -- filled by `set_core(mol)` when frozen_core
    no_core_idx: list | None = None
    core_list: list | None = None

    def __post_init__(self):
        AO = self.AO_per_frag
        if not self.relAO_per_edge_per_frag:
            # position of every edge AO inside its own fragment (autofrag.py:554-620)
            self.relAO_per_edge_per_frag = [[[AO[i].index(a) for a in e] for e in self.AO_per_edge_per_frag[i]] for i in range(len(AO))]
        if not self.relAO_in_ref_per_edge_per_frag:
            # position of the same AOs inside the fragment where they are centre sites (autofrag.py:652-698)
            self.relAO_in_ref_per_edge_per_frag = [
                [[AO[r].index(a) for a in e] for e, r in zip(self.AO_per_edge_per_frag[i], self.ref_frag_idx_per_edge_per_frag[i])]
                for i in range(len(AO))]
        self.weight_and_relAO_per_center_per_frag = [(float(w), list(c)) for w, c in self.weight_and_relAO_per_center_per_frag]

    @property
    def n_frag(self):
        return len(self.AO_per_frag)

    def all_centers_are_origins(self):
        return all(sorted(c) == sorted(o) for (_, c), o in zip(self.weight_and_relAO_per_center_per_frag, self.relAO_per_origin_per_frag))

    def replicate_sites(self, k: int) -> "FragPart":
        """The same fragmentation with k AOs per site (site a -> AOs k*a .. k*a+k-1): what the atom-based lists of a linear
        H chain become in a basis with k functions per atom."""
        ex = lambda lst: [k * a + j for a in lst for j in range(k)]
        return FragPart(AO_per_frag=[ex(f) for f in self.AO_per_frag],
                        AO_per_edge_per_frag=[[ex(e) for e in edges] for edges in self.AO_per_edge_per_frag],
                        ref_frag_idx_per_edge_per_frag=[list(r) for r in self.ref_frag_idx_per_edge_per_frag],
                        relAO_per_origin_per_frag=[ex(o) for o in self.relAO_per_origin_per_frag],
                        weight_and_relAO_per_center_per_frag=[(w, ex(c)) for w, c in self.weight_and_relAO_per_center_per_frag],
                        n_BE=self.n_BE)

    def set_core(self, mol) -> "FragPart":
        """ncore / no_core_idx / core_list of a frozen-core fragmentation (molbe/autofrag.py:135-145 -> helper.get_core)."""
        self.ncore, self.no_core_idx, self.core_list = get_core(mol)
        return self

    def freeze_core(self, mol) -> "FragPart":
        """The frozen-core counterpart of an all-electron fragmentation: the core AOs (the first `ncore_(Z)` functions of every
        atom) leave every list and the remaining AOs are renumbered in order -- what autogen does with `coreshift`
        (molbe/autofrag.py:519-548); the valence-only lists index the columns of the core-projected Loewdin W."""
        ncore, no_core_idx, core_list = get_core(mol)
        new = {old: k for k, old in enumerate(no_core_idx)}
        AO = [[new[a] for a in f if a in new] for f in self.AO_per_frag]
        pos = [{a: k for k, a in enumerate(f)} for f in AO]
        rel = lambda I, rel_old: [pos[I][new[self.AO_per_frag[I][r]]] for r in rel_old if self.AO_per_frag[I][r] in new]
        out = FragPart(AO_per_frag=AO,
                       AO_per_edge_per_frag=[[[new[a] for a in e if a in new] for e in edges] for edges in self.AO_per_edge_per_frag],
                       ref_frag_idx_per_edge_per_frag=[list(r) for r in self.ref_frag_idx_per_edge_per_frag],
                       relAO_per_origin_per_frag=[rel(I, o) for I, o in enumerate(self.relAO_per_origin_per_frag)],
                       weight_and_relAO_per_center_per_frag=[(w, rel(I, c)) for I, (w, c) in enumerate(self.weight_and_relAO_per_center_per_frag)],
                       n_BE=self.n_BE, frozen_core=True)
        out.ncore, out.no_core_idx, out.core_list = ncore, no_core_idx, core_list
        return out

    @classmethod
    def from_json(cls, path, key, n_BE=2):
        d = json.loads(Path(path).read_text())[key]
        return cls(AO_per_frag=d["AO_per_frag"], AO_per_edge_per_frag=d["AO_per_edge_per_frag"],
                   ref_frag_idx_per_edge_per_frag=d["ref_frag_idx_per_edge_per_frag"],
                   relAO_per_origin_per_frag=d["relAO_per_origin_per_frag"],
                   weight_and_relAO_per_center_per_frag=d["weight_and_relAO_per_center_per_frag"],
                   relAO_per_edge_per_frag=d.get("relAO_per_edge_per_frag", []),
                   relAO_in_ref_per_edge_per_frag=d.get("relAO_in_ref_per_edge_per_frag", []), n_BE=n_BE,
                   frozen_core=bool(d.get("frozen_core", False)))


def ncore_(z: int) -> int:
    """Number of frozen core orbitals of an atom (shared/helper.py:104-121)."""
    if 1 <= z <= 2:
        return 0
    if z <= 12:
        return 1
    if z <= 30:
        return 5
    if z <= 38:
        return 9
    if z <= 48:
        return 14
    if z <= 56:
        return 18
    raise ValueError("Ncore not computed in helper.ncore(), add it yourself!")


def get_core(mol):
    """(Ncore, AOs that are not core, cores per atom) -- molbe/helper.py:194-217."""
    Ncore, idx, corelist = 0, [], []
    for ix, bas in enumerate(mol.aoslice_by_atom()):
        nc = ncore_(int(mol.atom_charge(ix)))
        corelist.append(nc)
        Ncore += nc
        idx.extend(range(int(bas[2]) + nc, int(bas[3])))
    return Ncore, idx, corelist
