"""
Kernel backend selection.  The ONLY backend shipped is ``graphrole_amd.kernels`` (hand-written
HIP behind libgrx.so); it raises when no GPU or no library is present -- there is no CPU fallback.
``use()`` exists so the CPU-side tests can inject a test double (tests/fake_kernels.py, built on
oracle/) to exercise the host logic and the gloo sharding path without a GPU.
"""
_active = None


def get():
    global _active
    if _active is None:
        from graphrole_amd import kernels
        _active = kernels
    return _active


def use(module) -> None:
    """Tests only: replace the kernel module (None restores the HIP backend)."""
    global _active
    _active = module
