"""A short randomised parity sweep (tests/fuzz_parity.py): all five CAVI models through the model classes, f64 and
f32, on small random problems with odd shapes, against the CPU oracle."""
import os

import pytest

pytestmark = pytest.mark.gpu


def test_random_small_problems_match_the_oracle(capsys):
    import fuzz_parity
    failures, worst = fuzz_parity.sweep(180, seed=20251226, quiet=True)
    out = capsys.readouterr().out
    assert failures == 0, out
    assert len(worst) == 12                      # six kinds x two dtypes were all drawn
    for (kind, dtype), err in worst.items():
        assert err <= (1e-12 if dtype == "f64" else 1e-4), (kind, dtype, err)


def test_random_problems_through_the_three_stage_path(capsys):
    """accumulate -> ncclAllReduce (one-rank RCCL) per item row chunk -> finalize against the fused sweeps."""
    import fuzz_parity
    failures, worst = fuzz_parity.sweep_three_stage(80, seed=3, quiet=True)
    assert failures == 0, capsys.readouterr().out
    assert worst <= 2e-4


@pytest.mark.parametrize("world", [2, 3])
def test_random_problems_sharded_over_ranks_match_one_context(world, tmp_path):
    """Sharded fits (hostshm ranks on the one GPU; full-frame and presharded; random item row chunks) of random
    small problems against the single-context fit of the same model: tests/fuzz_sharded.py."""
    import multiprocessing as mp
    import socket
    import fuzz_sharded
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "sweep")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=fuzz_sharded.worker, args=(r, world, port, 40, 100 + world, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(400)
    hung = [p for p in procs if p.is_alive()]
    for p in hung:
        p.kill()
        p.join()
    logs = "\n".join(open(f"{out}.rank{r}").read() for r in range(world) if os.path.exists(f"{out}.rank{r}"))
    assert not hung, "a rank hung\n" + logs
    assert [p.exitcode for p in procs] == [0] * world, logs
    assert logs.count("done 40 0 ") == world, logs


def test_random_rating_lists_device_index_equals_host_index(capsys):
    import fuzz_parity
    assert fuzz_parity.sweep_index(25, seed=6) == 0, capsys.readouterr().out


def test_medium_problems_with_64_to_256_rating_tasks_match_the_oracle(capsys):
    """2.2M-9M ratings: the task lengths between the small tests' 32 and the full-size tests' 512."""
    import fuzz_parity
    failures, worst = fuzz_parity.sweep_medium(3, seed=8)
    assert failures == 0, capsys.readouterr().out
    assert worst <= 1e-11


def test_random_problems_from_concurrent_host_threads(capsys):
    """Six host threads run parity trials at once, each on its own contexts."""
    import fuzz_parity
    assert fuzz_parity.sweep_threads(6, 25, seed=50) == 0, capsys.readouterr().out
