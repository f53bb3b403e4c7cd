from .csr import CSRGraph  # noqa: F401
