"""The native CSV reader against pandas on random files (host code: runs without a GPU)."""
import pytest


@pytest.mark.parametrize("seed", [101, 102])
def test_fuzz_csv_reader_against_pandas(seed):
    from fuzz_csv_reader import fuzz
    files, native = fuzz(seed, budget_s=8.0, max_cases=4000)
    assert files >= 300 and native >= 100


@pytest.mark.parametrize("seed", [201])
def test_fuzz_writers_against_numpy_and_pandas(seed):
    from fuzz_csv_writer import fuzz
    assert fuzz(seed, budget_s=8.0, max_cases=3000) >= 300
