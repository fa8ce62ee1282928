"""Differential fuzzing of BasicCounter / pearson against the oracle (tests/fuzz_differential.py):
fixed seeds, a bounded number of cases per seed."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fuzz_cases(seed):
    from fuzz_differential import fuzz
    assert fuzz(seed, budget_s=25.0, max_cases=60) >= 10


@pytest.mark.parametrize("seed", [11, 12])
def test_fuzz_pearson_api(seed):
    from fuzz_pearson import fuzz
    assert fuzz(seed, budget_s=8.0, max_cases=4000) >= 200


@pytest.mark.parametrize("seed", [21, 22])
def test_fuzz_fasta_reader(seed):
    from fuzz_fasta import fuzz
    assert fuzz(seed, budget_s=6.0, max_cases=3000) >= 200


@pytest.mark.parametrize("seed", [31, 32])
def test_fuzz_consumers(seed):
    from fuzz_consumers import fuzz
    assert fuzz(seed, budget_s=6.0, max_cases=1500) >= 100
