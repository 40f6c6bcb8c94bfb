"""numba.typed shim: List is a plain Python list (see numba/__init__.py)."""


class List(list):
    @classmethod
    def empty_list(cls, *a, **k):
        return cls()


class Dict(dict):
    @classmethod
    def empty(cls, *a, **k):
        return cls()
