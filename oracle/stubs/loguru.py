"""Container-only stub of `loguru` (logging is a no-op). Test infrastructure only."""


class _Logger:
    def __getattr__(self, name):
        def _noop(*a, **k):
            return None

        return _noop


logger = _Logger()
