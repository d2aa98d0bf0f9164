"""Covariance, RelativisticVector, RelativisticObject: the reference's 4-vector value types (src/vectors.rs:13-173,
re-exported by src/lib.rs).  Host-side value types only: on the GPU the tags are compile-time facts (positions
contravariant, momenta covariant throughout).  Reference panics surface as CovarianceError."""
import enum

import numpy as np


class CovarianceError(RuntimeError):
    """the reference panics (mixed covariance, division by zero)"""


class Covariance(enum.Enum):
    Covariant = "Covariant"
    Contravariant = "Contravariant"

    def __str__(self):  # Display, src/vectors.rs:18-25
        return self.value


class RelativisticVector:
    """four components (time first) + covariance tag; arithmetic as src/vectors.rs:63-128"""

    __slots__ = ("vector", "covariance")

    def __init__(self, vector, covariance):
        v = np.array(vector, dtype=np.float64).reshape(-1)
        if v.size != 4:
            raise ValueError("a RelativisticVector has four components")
        if not isinstance(covariance, Covariance):
            raise TypeError("covariance must be a Covariance")
        self.vector, self.covariance = v, covariance

    def v(self, i):
        return float(self.vector[i])

    def copy(self):
        return RelativisticVector(self.vector.copy(), self.covariance)

    def __str__(self):
        return "%s (%s)" % (self.covariance, ", ".join(repr(float(c)) for c in self.vector))

    __repr__ = __str__

    def _scalar(self, other, op):
        return RelativisticVector(op(self.vector, np.float64(other)), self.covariance)

    def __add__(self, other):
        if isinstance(other, RelativisticVector):
            if self.covariance != other.covariance:
                raise CovarianceError("Cannot add vectors with different covariance")
            return RelativisticVector(self.vector + other.vector, self.covariance)
        return self._scalar(other, np.add)

    def __sub__(self, other):
        if isinstance(other, RelativisticVector):
            if self.covariance != other.covariance:
                raise CovarianceError("Cannot subtract vectors with different covariance")
            return RelativisticVector(self.vector - other.vector, self.covariance)
        return self._scalar(other, np.subtract)

    def __mul__(self, other):
        return self._scalar(other, np.multiply)

    def __truediv__(self, other):
        if float(other) == 0.0:
            raise CovarianceError("Division by zero")
        return self._scalar(other, np.divide)


class RelativisticObject:
    """position + momentum (src/vectors.rs:135-173)"""

    __slots__ = ("position", "momentum")

    def __init__(self, position, momentum):
        if not isinstance(position, RelativisticVector) or not isinstance(momentum, RelativisticVector):
            raise TypeError("position and momentum must be RelativisticVectors")
        self.position, self.momentum = position, momentum

    def x(self, i):
        return self.position.v(i)

    def p(self, i):
        return self.momentum.v(i)

    def covariance_x(self):
        return self.position.covariance

    def covariance_p(self):
        return self.momentum.covariance

    def copy(self):
        return RelativisticObject(self.position.copy(), self.momentum.copy())

    def __repr__(self):
        return "RelativisticObject(position=%s, momentum=%s)" % (self.position, self.momentum)
