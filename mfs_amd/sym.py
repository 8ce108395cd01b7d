"""Symbolic tracing of model callables.

The reference's filters take arbitrary JAX callables (mfs/one_dim/filtering.py:32-36,92-98) and JAX traces them into
one XLA program.  Callables cannot cross a C ABI, so this module plays the tracer's role: model functions are CALLED
with the placeholder objects below and reduce to small coefficient tables that the HIP kernels consume
(include/mfs_hip.h, `mfs_model_1d`).  Supported function class (covers every model the reference ships):

  drift / dispersion / conditional mean / variance : polynomials in u, u = x or u = tanh(x)
  measurement models : Bernoulli(logistic(poly_3(x))), Poisson(softplus(l x)), Normal(l0 x + l1, var)

Coefficients may carry a leading batch axis (per-replicate parameters), e.g. `drift(x, p)` with `p.shape == (B,)`.
Anything outside this class raises `NotDeviceDescribable` -- there is no CPU fallback.
"""
import numbers

import numpy as np

__all__ = ['Poly', 'X', 'Y', 'ORDER', 'MEAN', 'SCALE', 'tanh', 'exp', 'log', 'sqrt', 'NotDeviceDescribable',
           'LikelihoodSpec', 'as_poly', 'is_symbolic']


class NotDeviceDescribable(TypeError):
    """The callable cannot be reduced to the coefficient tables the HIP kernels understand."""


class _Placeholder:
    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return f'<mfs_amd placeholder {self.name}>'


class _Measurement(_Placeholder):
    """The measurement y.  Scalar in the 1-D filters; in the N-D filters it may be a vector (ys of shape (T, ny),
    reference tests/test_filtering.py:36): `y[k]` is column k, and a likelihood evaluated elementwise on (y, x) pairs
    column k with state component k."""

    def __init__(self, name='y', ycol=None):
        super().__init__(name)
        self.ycol = ycol

    def __getitem__(self, k):
        if self.ycol is not None:
            raise NotDeviceDescribable('y[k] is already a scalar column')
        if isinstance(k, (int, np.integer)):
            return _Measurement(f'y[{int(k)}]', int(k))
        raise NotDeviceDescribable('measurements can only be indexed by an integer column')


Y = _Measurement()             # the measurement
ORDER = _Placeholder('order')  # the moment order n
MEAN = _Placeholder('mean')    # the centre of central / scaled moments
SCALE = _Placeholder('scale')


# The dtype of every coefficient array the tracer builds.  float64, except while `mfs_amd.estimation` traces a model at
# theta + i h e_p: the tables are analytic in the parameters, so Im(table) / h is their EXACT parameter derivative (the
# complex-step derivative: no subtraction, hence no step-size compromise) -- what differentiating the traced polynomials
# symbolically would give, for any dependence on theta the model's own arithmetic can express (exp(theta), softplus, ...).
_DTYPE = [np.float64]


class coefficient_dtype:
    """Context manager: trace with coefficient arrays of `dtype` (np.complex128 for complex-step tangents)."""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        self.prev = _DTYPE[0]
        _DTYPE[0] = self.dtype
        return self

    def __exit__(self, *exc):
        _DTYPE[0] = self.prev
        return False


def _zeros(shape):
    return np.zeros(shape, dtype=_DTYPE[0])


def _as_coef(c):
    """scalar -> shape (1,); array (B,) -> shape (B, 1) (a batched constant)."""
    c = np.asarray(c, dtype=_DTYPE[0])
    return c.reshape(c.shape + (1,))


def _pad(a, deg):
    if a.shape[-1] - 1 >= deg:
        return a
    pad = np.zeros(a.shape[:-1] + (deg + 1 - a.shape[-1],), dtype=a.dtype)
    return np.concatenate([a, pad], axis=-1)


class Poly:
    """sum_j coef[..., j] u^j with u = x ('x'), u = tanh x ('tanh') or a constant (umap None)."""
    __array_priority__ = 1000
    __array_ufunc__ = None  # ndarray (op) Poly defers to Poly.__r(op)__; np.sin(Poly) etc. raise TypeError

    def __init__(self, coef, umap=None, comp=None):
        self.coef = np.asarray(coef, dtype=_DTYPE[0])
        if self.coef.ndim == 0:
            self.coef = self.coef.reshape(1)
        self.umap = umap
        self.comp = comp   # N-D likelihood tracing: which state component this expression is a function of (None: any / 1-D)

    # -- helpers
    @property
    def degree(self):
        return self.coef.shape[-1] - 1

    def is_const(self):
        return self.degree == 0 or not np.any(self.coef[..., 1:])

    def trimmed(self, tol=0.0):
        c = self.coef
        while c.shape[-1] > 1 and np.all(np.abs(c[..., -1]) <= tol):
            c = c[..., :-1]
        return Poly(c, self.umap if c.shape[-1] > 1 else None, self.comp)

    @staticmethod
    def _merge_comp(a, b):
        if a.comp is not None and b.comp is not None and a.comp != b.comp:
            raise NotDeviceDescribable('a likelihood factor may depend on one state component only '
                                       '(write the likelihood as a product of per-component factors)')
        return a.comp if a.comp is not None else b.comp

    @staticmethod
    def _merge_umap(a, b):
        ua = None if a.is_const() else a.umap
        ub = None if b.is_const() else b.umap
        if ua and ub and ua != ub:
            raise NotDeviceDescribable('expressions mixing x and tanh(x) polynomially are not supported on the device')
        return ua or ub

    @staticmethod
    def lift(v):
        if isinstance(v, Poly):
            return v
        if isinstance(v, _Placeholder) or isinstance(v, _Expr):
            raise NotDeviceDescribable(f'{v!r} cannot be used in a polynomial expression')
        try:
            return Poly(_as_coef(v))
        except (TypeError, ValueError) as e:
            raise NotDeviceDescribable(f'{type(v).__name__} cannot be used in a polynomial expression') from e

    # -- ring operations
    def __add__(self, o):
        o = Poly.lift(o)
        deg = max(self.degree, o.degree)
        return Poly(_pad(self.coef, deg) + _pad(o.coef, deg), Poly._merge_umap(self, o), Poly._merge_comp(self, o))

    __radd__ = __add__

    def __neg__(self):
        return Poly(-self.coef, self.umap, self.comp)

    def __sub__(self, o):
        return self + (-Poly.lift(o))

    def __rsub__(self, o):
        return Poly.lift(o) + (-self)

    def __mul__(self, o):
        o = Poly.lift(o)
        umap = Poly._merge_umap(self, o)
        a, b = self.coef, o.coef
        shape = np.broadcast_shapes(a.shape[:-1], b.shape[:-1])
        out = _zeros(shape + (a.shape[-1] + b.shape[-1] - 1,))
        for i in range(a.shape[-1]):
            out[..., i:i + b.shape[-1]] += a[..., i:i + 1] * b
        return Poly(out, umap, Poly._merge_comp(self, o))

    __rmul__ = __mul__

    def __truediv__(self, o):
        o = Poly.lift(o)
        if not o.is_const():
            raise NotDeviceDescribable('division by a non-constant expression')
        return Poly(self.coef / o.coef[..., :1], self.umap, self.comp)

    def __rtruediv__(self, o):
        raise NotDeviceDescribable('division by a non-constant expression')

    def __pow__(self, k):
        if not (isinstance(k, numbers.Integral) or (isinstance(k, float) and float(k).is_integer())) or k < 0:
            raise NotDeviceDescribable(f'power {k!r}: only non-negative integer powers are polynomial')
        out = Poly(np.ones(1))
        for _ in range(int(k)):
            out = out * self
        return out

    # -- calculus and evaluation
    def du(self):
        """d/du."""
        if self.degree == 0:
            return Poly(_zeros(self.coef.shape[:-1] + (1,)))
        j = np.arange(1, self.degree + 1, dtype=np.float64)
        return Poly(self.coef[..., 1:] * j, self.umap)

    def dx(self):
        """d/dx, with du/dx = 1 (u = x) or 1 - u^2 (u = tanh x)."""
        d = self.du()
        if self.umap == 'tanh':
            return d * Poly(np.array([1., 0., -1.]), 'tanh')
        return d

    def __call__(self, x):
        x = np.asarray(x, dtype=np.float64)
        u = np.tanh(x) if self.umap == 'tanh' else x
        acc = np.zeros(np.broadcast_shapes(self.coef.shape[:-1], u.shape))
        for j in range(self.degree, -1, -1):
            acc = acc * u + self.coef[..., j]
        return acc

    def __repr__(self):
        return f'Poly(deg={self.degree}, u={self.umap}, batch={self.coef.shape[:-1]})'


X = Poly(np.array([0., 1.]), 'x')  # the state variable


def state_vector(d: int) -> np.ndarray:
    """Object array of the d state components, each tagged with its index (N-D likelihood tracing)."""
    xs = np.empty((d,), dtype=object)
    for k in range(d):
        xs[k] = Poly(np.array([0., 1.]), 'x', comp=k)
    return xs


# ---------------------------------------------------------------------------------------------------------------------
# the few non-polynomial nodes measurement models need
# ---------------------------------------------------------------------------------------------------------------------
class _Expr:
    pass


class _Exp(_Expr):  # exp(q)
    def __init__(self, q):
        self.q = q

    def __add__(self, o):
        if isinstance(o, numbers.Real) and o == 1:
            return _OnePlusExp(self.q)
        raise NotDeviceDescribable('exp(.) may only appear as 1 + exp(.)')

    __radd__ = __add__


class _OnePlusExp(_Expr):  # 1 + exp(q)
    def __init__(self, q):
        self.q = q

    def __rtruediv__(self, o):
        if isinstance(o, numbers.Real) and o == 1:
            return _Logistic(-self.q)  # 1 / (1 + exp(q)) = logistic(-q)
        raise NotDeviceDescribable('only 1 / (1 + exp(.)) is supported')


class _Logistic(_Expr):  # 1 / (1 + exp(-z))
    def __init__(self, z):
        self.z = z


class _Softplus(_Expr):  # log(1 + exp(q))
    def __init__(self, q):
        self.q = q


class _Bearing(_Expr):  # arctan2(x[1], x[0]) of the two tagged state components
    pass


def arctan2(x1, x0):
    """numpy.arctan2 on numbers; on the traced state components x[1], x[0] (in that order, as
    /root/reference/examples/2d_bearing_only.ipynb cell 7 writes it) the bearing of the state."""
    if isinstance(x1, Poly) or isinstance(x0, Poly):
        ok = all(isinstance(v, Poly) and v.umap in (None, 'x') and v.degree == 1 and np.all(_pad(v.coef, 1) == [0., 1.])
                 for v in (x1, x0))
        if not ok or x1.comp != 1 or x0.comp != 0:
            raise NotDeviceDescribable('arctan2: the device takes arctan2(x[1], x[0]) of the plain state components')
        return _Bearing()
    return np.arctan2(x1, x0)


def is_symbolic(v):
    return isinstance(v, (Poly, _Expr, _Placeholder))


def tanh(v):
    if isinstance(v, Poly):
        c = v.trimmed().coef
        if v.umap == 'x' and c.shape == (2,) and c[0] == 0. and c[1] == 1.:
            return Poly(np.array([0., 1.]), 'tanh')
        raise NotDeviceDescribable('tanh is only supported of the bare state variable x')
    return np.tanh(v)


def exp(v):
    if isinstance(v, Poly):
        return _Exp(v)
    return np.exp(v)


def log(v):
    if isinstance(v, _OnePlusExp):
        return _Softplus(v.q)
    if is_symbolic(v):
        raise NotDeviceDescribable('log is only supported as log(1 + exp(.))')
    return np.log(v)


def sqrt(v):
    if is_symbolic(v):
        raise NotDeviceDescribable('sqrt of a symbolic expression')
    return np.sqrt(v)


def as_poly(v, what='expression'):
    """Coerce the result of tracing a drift / dispersion / mean / variance callable to a Poly."""
    if isinstance(v, Poly):
        return v
    if is_symbolic(v):
        raise NotDeviceDescribable(f'{what} is not a polynomial in x or tanh(x)')
    try:
        return Poly(_as_coef(v))
    except (TypeError, ValueError) as e:
        raise NotDeviceDescribable(f'{what} returned {type(v).__name__}, not a polynomial') from e


class LikelihoodSpec:
    """Device description of p(y | x): kind in {'bernoulli_logistic', 'poisson_softplus', 'gaussian'}, params (..., P)."""
    KINDS = {'bernoulli_logistic': 0, 'poisson_softplus': 1, 'gaussian': 2, 'bearing_gaussian': 3}



    def __init__(self, kind, params, component=0, ycol=0):
        assert kind in self.KINDS
        self.kind = kind
        self.params = np.asarray(params, dtype=_DTYPE[0])
        self.component = component  # which state component the factor reads (N-D models)
        self.ycol = ycol            # which measurement column it reads (N-D models with vector measurements)

    @property
    def factors(self):
        return [self]

    def __mul__(self, o):
        if isinstance(o, numbers.Real) and o == 1:
            return self
        if isinstance(o, (LikelihoodSpec, LikelihoodProduct)):
            return LikelihoodProduct(self.factors + o.factors)
        raise NotDeviceDescribable('a likelihood can only be multiplied by another likelihood factor')

    __rmul__ = __mul__

    def __repr__(self):
        return f'LikelihoodSpec({self.kind}, params shape {self.params.shape}, x[{self.component}], y[{self.ycol}])'


class LikelihoodProduct:
    """p(y | x) = prod_f factor_f: what `math.prod(norm.pdf(y, x, sd))` on vector y, x traces to
    (reference tests/test_filtering.py:44-46)."""

    def __init__(self, factors):
        self.factors = list(factors)

    def __mul__(self, o):
        if isinstance(o, numbers.Real) and o == 1:
            return self
        if isinstance(o, (LikelihoodSpec, LikelihoodProduct)):
            return LikelihoodProduct(self.factors + o.factors)
        raise NotDeviceDescribable('a likelihood can only be multiplied by another likelihood factor')

    __rmul__ = __mul__

    def __repr__(self):
        return f'LikelihoodProduct({self.factors})'


class LikelihoodVector:
    """Elementwise likelihood of vector measurements / states (one factor per entry); iterable, so that
    `math.prod(...)` / `np.prod(...)` reduce it to a LikelihoodProduct."""

    def __init__(self, factors):
        self.factors = list(factors)

    def __iter__(self):
        return iter(self.factors)

    def __len__(self):
        return len(self.factors)

    def prod(self, *_a, **_k):
        return LikelihoodProduct(self.factors)
