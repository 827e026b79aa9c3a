class _P:
    def __getattr__(self, n):
        return self

    def __call__(self, *a, **k):
        return self


projector = _P()
