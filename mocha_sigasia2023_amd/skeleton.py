"""Skeleton graph constants for the MOCHA Generator path.

Own restatement (closed form) of what the reference builds in
``net/graph.py``:

* hop-distance partitioned, column-normalised adjacency stacks
  (``Graph_Joint`` net/graph.py:6-153, ``Graph_Bodypart`` :156-287,
  ``get_hop_distance`` :290-301, ``normalize_digraph`` :304-312, strategy
  ``'distance'`` :127-131);
* joint -> body-part mean-pool matrix (``PoolJointToBodypart`` :326-465) and the
  body-part -> joint copy matrix (``UnpoolBodypartToJoint`` :468-608).

Hop distances come from a breadth-first search over the undirected skeleton tree instead
of the reference's dense matrix powers; the result is the same integer table.  Everything
is computed in float64 and cast to float32 last, as the reference does (model.py:116,144).

Only the two layouts the hot path can reach are tabulated: ``'mocha'`` (24 joints, the
shipped model, configs/config.yaml:8-10,42) and ``'mixamo'`` (22 joints, the synthetic
22-joint shape of BASELINE.json; net/graph.py:18-31,329-345).
"""
from __future__ import annotations

from collections import deque
from dataclasses import dataclass

import numpy as np

# parents (root = -1) and body-part groups, part order 0..5 as the pool matrices use it.
LAYOUTS = {
    # net/graph.py:65-79 (edges), :401-417 (pool groups)
    "mocha": {
        "parents": [-1, 0, 1, 2, 3, 0, 5, 6, 7, 8, 9, 10, 11, 8, 13, 14, 8, 16, 17, 18, 0, 20, 21, 22],
        "parts": [
            [0, 5, 6, 7, 8],      # Spine
            [1, 2, 3, 4],         # LeftLeg
            [9, 10, 11, 12],      # LeftArm
            [13, 14, 15],         # Neck
            [16, 17, 18, 19],     # RightArm
            [20, 21, 22, 23],     # RightLeg
        ],
    },
    # net/graph.py:18-31 (edges), :329-345 (pool groups)
    "mixamo": {
        "parents": [-1, 0, 1, 2, 3, 4, 3, 6, 7, 8, 3, 10, 11, 12, 0, 14, 15, 16, 0, 18, 19, 20],
        "parts": [
            [0, 1, 2, 3],         # Spine
            [4, 5],               # Neck
            [6, 7, 8, 9],         # LeftArm
            [10, 11, 12, 13],     # RightArm
            [14, 15, 16, 17],     # RightLeg
            [18, 19, 20, 21],     # LeftLeg
        ],
    },
}
LAYOUT_ID = {"mocha": 0, "mixamo": 1}

# body-part graph: star, part 0 linked to parts 1..5 (net/graph.py:207-218), same for all layouts
BODY_PARENTS = [-1, 0, 0, 0, 0, 0]
NBODY = 6


def hop_distance(parents, max_hop: int) -> np.ndarray:
    """hop[v, w] = tree distance between v and w, ``inf`` beyond ``max_hop``."""
    n = len(parents)
    nbr = [[] for _ in range(n)]
    for i, p in enumerate(parents):
        if p >= 0:
            nbr[i].append(p)
            nbr[p].append(i)
    hop = np.full((n, n), np.inf)
    for s in range(n):
        hop[s, s] = 0
        q = deque([(s, 0)])
        seen = {s}
        while q:
            v, d = q.popleft()
            if d == max_hop:
                continue
            for u in nbr[v]:
                if u not in seen:
                    seen.add(u)
                    hop[s, u] = d + 1
                    q.append((u, d + 1))
    return hop


def distance_adjacency(parents, max_hop: int) -> np.ndarray:
    """A[k, v, w] = 1/|N(w)| if hop(v, w) == k else 0, N(w) = {v : hop(v, w) <= max_hop}.

    float64; columns of sum_k A[k] sum to 1 (net/graph.py:116-131 with dilation 1).
    """
    hop = hop_distance(parents, max_hop)
    n = len(parents)
    reach = (hop <= max_hop)
    col = reach.sum(axis=0).astype(np.float64)          # |N(w)|
    A = np.zeros((max_hop + 1, n, n), dtype=np.float64)
    for k in range(max_hop + 1):
        m = hop == k
        A[k][m] = (1.0 / col)[None, :].repeat(n, 0)[m]
    return A


def pool_matrix(layout: str) -> np.ndarray:
    """(V, 6) float32: column p = 1/|part p| on the joints of part p (net/graph.py:459-461)."""
    tab = LAYOUTS[layout]
    v = len(tab["parents"])
    w = np.zeros((v, NBODY), dtype=np.float32)
    for p, joints in enumerate(tab["parts"]):
        w[joints, p] = 1.0
    return (w / w.sum(axis=0, keepdims=True)).astype(np.float32)


def unpool_matrix(layout: str) -> np.ndarray:
    """(6, V) float32 copy matrix: every joint takes its part's value (net/graph.py:602-604)."""
    tab = LAYOUTS[layout]
    v = len(tab["parents"])
    w = np.zeros((NBODY, v), dtype=np.float32)
    for p, joints in enumerate(tab["parts"]):
        w[p, joints] = 1.0
    return (w / w.sum(axis=0, keepdims=True)).astype(np.float32)


@dataclass(frozen=True)
class SkeletonConstants:
    layout: str
    V: int
    parents: tuple
    part_of: tuple            # joint -> body part
    A_j: np.ndarray           # (3, V, V) float32
    A_b: np.ndarray           # (2, 6, 6) float32
    pool: np.ndarray          # (V, 6) float32
    unpool: np.ndarray        # (6, V) float32


def skeleton_constants(layout: str = "mocha", joint_max_hop: int = 2, body_max_hop: int = 1) -> SkeletonConstants:
    tab = LAYOUTS[layout]
    parents = tab["parents"]
    part_of = [0] * len(parents)
    for p, joints in enumerate(tab["parts"]):
        for j in joints:
            part_of[j] = p
    return SkeletonConstants(
        layout=layout,
        V=len(parents),
        parents=tuple(parents),
        part_of=tuple(part_of),
        A_j=distance_adjacency(parents, joint_max_hop).astype(np.float32),
        A_b=distance_adjacency(BODY_PARENTS, body_max_hop).astype(np.float32),
        pool=pool_matrix(layout),
        unpool=unpool_matrix(layout),
    )
