"""Minimal observation / action space descriptors (the reference uses
gym.spaces.{Box,Discrete,Tuple}; gym is not a dependency here).  Any object
with the same attributes (``shape`` / ``n`` / ``spaces``) works, including real
gym spaces."""
import numpy as np


class Box:
    def __init__(self, low, high, shape, dtype=np.float32):
        self.low, self.high = low, high
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)

    def __repr__(self):
        return "Box%s" % (self.shape,)


class Discrete:
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()

    def __repr__(self):
        return "Discrete(%d)" % self.n


class Tuple:
    def __init__(self, spaces):
        self.spaces = tuple(spaces)

    def __repr__(self):
        return "Tuple%s" % (self.spaces,)


def is_tuple_space(space):
    return hasattr(space, "spaces")
