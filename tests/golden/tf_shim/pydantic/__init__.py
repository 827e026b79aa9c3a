"""Test-only stand-in: the reference's dataset registry classes are pydantic-v1
style; the container has pydantic 2.x.  Only needed so `import emgraph` succeeds
inside tests/golden/make_golden.py."""


class BaseModel:
    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    def __init_subclass__(cls, **kw):
        super().__init_subclass__()


def Field(default=None, **kw):  # noqa: N802
    return default


def validator(*a, **k):
    def deco(f):
        return f
    return deco
