"""The driver's round-end entry point, run as a test: ``__graft_entry__.smoke()`` must pass on the GPU box with the defaults the
package ships (it asserts the mode it ran and checks loss / gradient against the CPU oracle itself)."""
import pytest

pytestmark = pytest.mark.gpu


def test_graft_entry_smoke_passes_with_the_shipped_defaults():
    import __graft_entry__ as g
    g.smoke()
