"""The program generator of tests/tape_fuzz.py, run on the CPU backend alone: every drawn program must run, and float32 must stay
close to a float64 run of the same program (a generator that draws exploding programs would make the GPU comparison meaningless)."""
import numpy as np
import pytest
from lightgrad_amd import CpuTensor
from common import float64_tape
from tape_fuzz import draw_program, run_program, compare


@pytest.mark.parametrize("block", range(4))
def test_programs_run_and_are_well_conditioned(block):
    for seed in range(block * 10, block * 10 + 10):
        prog = draw_program(seed)
        got = run_program(CpuTensor, prog)
        with float64_tape():
            ref = run_program(CpuTensor, prog, dtype=np.float64)
        assert len(got) >= 3
        compare(ref, got, rtol=5e-3, atol=5e-4, what="seed %d %r" % (seed, prog))
