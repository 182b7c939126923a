"""Oracle: heap-layout reduction trees (TEST INFRASTRUCTURE ONLY).

Restates rltime/history/data_structures/segment_tree.py (OpenAI-baselines
segment tree): ``2*capacity`` slots, root at 1, leaves at [capacity, 2*capacity).

Slots hold whatever scalar objects the caller stores (Python float / int,
np.float32, np.float64) and parents are produced by the plain binary operator,
so NumPy-2 scalar promotion decides each node's precision exactly as in the
reference.
"""
import operator


class HeapTree:
    """segment_tree.py:10-101 (SegmentTree)."""

    def __init__(self, capacity, combine, neutral):
        if capacity <= 0 or capacity & (capacity - 1):
            raise ValueError("capacity must be a positive power of two")
        self.capacity = capacity
        self.combine = combine
        self.nodes = [neutral] * (2 * capacity)

    def set_leaf(self, leaf, value):
        """segment_tree.py:87-97 (__setitem__): write the leaf, then re-derive
        every ancestor from its two children up to the root."""
        pos = leaf + self.capacity
        self.nodes[pos] = value
        pos >>= 1
        while pos:
            self.nodes[pos] = self.combine(
                self.nodes[2 * pos], self.nodes[2 * pos + 1])
            pos >>= 1

    def leaf(self, leaf):
        """segment_tree.py:99-101 (__getitem__)."""
        assert 0 <= leaf < self.capacity
        return self.nodes[leaf + self.capacity]

    def total(self):
        """segment_tree.py:60-85 with the default full range: the recursion
        terminates immediately at the root (start==node_start, end==node_end),
        so the full-range reduce IS the root slot."""
        return self.nodes[1]


class SumTree(HeapTree):
    """segment_tree.py:104-142 (SumSegmentTree)."""

    def __init__(self, capacity):
        super().__init__(capacity, operator.add, 0.0)

    def descend(self, mass):
        """segment_tree.py:116-142 (find_prefixsum_idx): go left when the left
        child is STRICTLY greater than the remaining mass, else subtract it and
        go right."""
        assert 0 <= mass <= self.total() + 1e-5
        pos = 1
        while pos < self.capacity:
            left = self.nodes[2 * pos]
            if left > mass:
                pos = 2 * pos
            else:
                mass -= left
                pos = 2 * pos + 1
        return pos - self.capacity


class MinTree(HeapTree):
    """segment_tree.py:145-156 (MinSegmentTree)."""

    def __init__(self, capacity):
        super().__init__(capacity, min, float("inf"))
