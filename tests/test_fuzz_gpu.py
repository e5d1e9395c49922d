"""A short randomised parity sweep (tests/fuzz_parity.py): all five CAVI models through the model classes, f64 and
f32, on small random problems with odd shapes, against the CPU oracle."""
import pytest

pytestmark = pytest.mark.gpu


def test_random_small_problems_match_the_oracle(capsys):
    import fuzz_parity
    failures, worst = fuzz_parity.sweep(150, seed=20251226, quiet=True)
    out = capsys.readouterr().out
    assert failures == 0, out
    assert len(worst) == 10                      # five kinds x two dtypes were all drawn
    for (kind, dtype), err in worst.items():
        assert err <= (1e-12 if dtype == "f64" else 1e-4), (kind, dtype, err)


def test_random_problems_through_the_three_stage_path(capsys):
    """accumulate -> ncclAllReduce (one-rank RCCL) per item row chunk -> finalize against the fused sweeps."""
    import fuzz_parity
    failures, worst = fuzz_parity.sweep_three_stage(80, seed=3, quiet=True)
    assert failures == 0, capsys.readouterr().out
    assert worst <= 2e-4
