"""The committed independent-model fixtures (tests/golden/bulk_vectors.json) against the C
oracle (CPU tier) and against the HIP engine through the C ABI (GPU tier)."""
import pytest

from tests import golden_checks as G


def test_oracle_matches_independent_model_fixtures(oracle):
    G.check_backend(oracle, G.load())


@pytest.mark.gpu
def test_hip_matches_independent_model_fixtures():
    import dusk_zerocaf_amd as z
    eng = z.Engine()
    try:
        G.check_backend(eng, G.load())
    finally:
        eng.close()
