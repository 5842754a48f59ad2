from ._leiden import leiden

__all__ = ["leiden"]
