"""-m gpu: every libtavhip kernel against a plain PyTorch reference of the same op (tests/kernel_checks.py)."""
import pytest

pytestmark = pytest.mark.gpu


def _cases():
    import kernel_checks
    return kernel_checks.all_checks()


@pytest.mark.parametrize("idx", range(len(_cases())))
def test_kernel_check(gpu, idx):
    cases = _cases()
    for name, err, tol, ok in cases[idx]():
        assert ok, f"{name}: rel err {err:.3e} > tol {tol:.1e}"
