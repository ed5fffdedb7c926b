"""A bounded, seeded slice of the builder-side soak runs (tools/soak_grid.py, soak_align.py, soak_batch.py) inside
`-m gpu`, so that the driver's own run exercises them: random cloud shapes (lines, lattices, duplicates, tiny / huge
scales), ragged sizes, first (unseeded, expanding) and seeded grid sweeps, whole alignments with threshold exits and
fall-backs, random lock-step groups -- each against an independent path (exact kernel + host loop, or the pairs one by
one), bit for bit.  The seeded chain (sweeps 2 and 3 of a case; icp.cpp:566-593 with the previous match as the bound)
is where a slip in the cube / ball trimming of the grid scan would show.  Each slice is time-boxed (the cases are the
same on every run up to the point where the budget ends)."""
import pytest

from icp_slam_prototype_amd import binding

import soak_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from icp_slam_prototype_amd import build

    build.build()
    c = binding.Context(0)
    yield c
    c.close()


def test_soak_slice_grid_sweeps_vs_exact_kernel(ctx):
    done, bad = soak_cases.soak_grid(ctx, 6000, seed0=5000, budget_s=20)
    assert done >= 20 and bad == 0, (done, bad)


def test_soak_slice_grid_sweeps_vs_cpu_oracle(ctx, oracle):
    done, bad = soak_cases.soak_grid(ctx, 60, seed0=7000, budget_s=15, oracle=oracle)
    assert done >= 5 and bad == 0, (done, bad)


def test_soak_slice_alignments_grid_device_loop_vs_exact_host_loop(ctx):
    done, bad = soak_cases.soak_align(ctx, 4000, seed0=9000, budget_s=20)
    assert done >= 20 and bad == 0, (done, bad)


def test_soak_slice_lockstep_batches_vs_single_pairs():
    done, bad = soak_cases.soak_batch(1000, seed0=12000, budget_s=20)
    assert done >= 20 and bad == 0, (done, bad)
