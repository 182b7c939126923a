import numpy as np


class Space:
    shape = None
    dtype = None


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.asarray(low).shape
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), self.shape)
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self.shape)

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)

    def __repr__(self):
        return "Box%s" % (self.shape,)


class Discrete(Space):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)

    def sample(self):
        return np.random.randint(self.n)

    def __repr__(self):
        return "Discrete(%d)" % self.n


class Tuple(Space):
    def __init__(self, spaces):
        self.spaces = tuple(spaces)

    def __repr__(self):
        return "Tuple%s" % (self.spaces,)
