"""Import shim used ONLY by oracle/gen_golden.py in the build container.

Makes the (absent) `numba` package a no-op so the reference can be imported in
the pure-Python mode its own CI pins (NUMBA_DISABLE_JIT=1,
/root/reference/.github/workflows/ci.yml:36-39).  Test infrastructure; never
imported by the product package.
"""
__version__ = "0.0-shim"


def _passthrough(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def deco(fn):
        return fn
    return deco


njit = jit = vectorize = guvectorize = _passthrough
prange = range


class _Types:
    def __getattr__(self, name):
        return name


types = _Types()
float64 = "float64"
int64 = "int64"
