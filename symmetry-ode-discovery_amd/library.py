"""Host-side description of the feature library Theta (mirrors csrc/library.hpp).

Column order: [1 | z_i | z_i z_j (i<=j) | z_i z_j z_k (i<=j<=k) | ... | sin z_i | exp z_i]
(reference sindy.py:68-77); the nested non-decreasing index tuples are exactly
``itertools.combinations_with_replacement`` in lexicographic order, continued to order 5.
"""
from __future__ import annotations

import itertools
import math

MAX_ORDER = 5


def poly_tuples(d: int, n: int):
    return list(itertools.combinations_with_replacement(range(d), n))


def poly_term_count(d: int, order: int) -> int:
    return 1 + sum(math.comb(d + n - 1, n) for n in range(1, order + 1))


def term_count(d: int, order: int, include_sine: bool = False, include_exp: bool = False) -> int:
    """Number of library columns (reference sindy.py:179-189, generalised past cubic)."""
    return poly_term_count(d, order) + (d if include_sine else 0) + (d if include_exp else 0)


def exponents(d: int, order: int):
    """Exponent vector of every polynomial column, constant first."""
    out = [tuple([0] * d)]
    for n in range(1, order + 1):
        for tup in poly_tuples(d, n):
            e = [0] * d
            for i in tup:
                e[i] += 1
            out.append(tuple(e))
    return out


def term_names(d: int, order: int, include_sine: bool = False, include_exp: bool = False, var: str = "z"):
    """Printable factor of each column ('' for the constant), in the style of sindy.py:206-247."""
    names = [""]
    for n in range(1, order + 1):
        for tup in poly_tuples(d, n):
            names.append("*".join(f"{var}{i}" for i in tup))
    if include_sine:
        names += [f"sin({var}{i})" for i in range(d)]
    if include_exp:
        names += [f"exp({var}{i})" for i in range(d)]
    return names
