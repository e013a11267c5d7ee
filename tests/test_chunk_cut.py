"""How a mini-batch is cut into chunks for the pipeline (mapper.c: cut_plan_*): every read exactly once, no chunk above MGA_CHUNK, ramps at the
job's two ends, equal chunks in between (round 5: full chunks + a remainder left chunks of a few hundred reads that pay a pass's whole chain of
launches for no work).  The GAF never depends on the cut (tests/test_gpu_e2e.py::test_pipeline_knobs_do_not_change_the_output)."""
import ctypes as C

import pytest

import minigraph_amd as mga

FIRST, LAST = 1, 2


def cut(n, chunk, flags, even=1, tail=1):
    L = mga.load()
    L.mga_debug_cut.restype = C.c_int
    a = (C.c_int * 8192)()
    m = L.mga_debug_cut(n, chunk, flags, even, tail, a, 8192)
    assert 0 < m <= 8192
    return list(a[:m])


@pytest.mark.parametrize("even,tail", [(0, 1), (1, 1), (1, 2), (1, 3), (1, 8)])
def test_every_read_once_and_no_chunk_above_the_cap(even, tail):
    for chunk in (1, 7, 64, 1000, 16384):
        for flags in (0, FIRST, LAST, FIRST | LAST):
            for n in list(range(1, 40)) + [chunk - 1, chunk, chunk + 1, 2 * chunk + 3, 7 * chunk + 5, 50000, 125000, 1000003]:
                if n < 1 or n > 300 * chunk:
                    continue
                c = cut(n, chunk, flags, even, tail)
                assert sum(c) == n and min(c) >= 1 and max(c) <= chunk, (n, chunk, flags, c)


def test_equal_chunks_and_ramps_at_bench_sizes():
    # the bench's three -K batches of one step (50 000 + 50 000 + 25 000 reads of 10 kb)
    assert cut(50000, 16384, 0, even=0) == [16384, 16384, 16384, 848]                 # rounds 1-4
    assert cut(50000, 16384, 0) == [12500] * 4
    assert cut(50000, 16384, FIRST)[:2] == [4096, 8192] and min(cut(50000, 16384, FIRST)) == 4096
    assert cut(25000, 16384, LAST, even=0) == [16384, 8616]
    assert cut(25000, 16384, LAST) == [8404, 8404, 8192]
    assert cut(25000, 16384, LAST, tail=3) == [10664, 8192, 4096, 2048]                # tapered drain
    c = cut(1000000, 16384, FIRST | LAST, tail=3)
    assert c[:2] == [4096, 8192] and c[-3:] == [8192, 4096, 2048] and max(c[2:-3]) - min(c[2:-3]) <= 8
    # a job that is one small batch is not ramped
    assert cut(40000, 16384, FIRST | LAST) == [13334, 13334, 13332]
    assert cut(257, 16384, FIRST | LAST) == [257]


def cut2(n, chunk, flags, job_pos, levels, even=1, tail=1):
    L = mga.load()
    L.mga_debug_cut2.restype = C.c_int
    a = (C.c_int * 8192)()
    m = L.mga_debug_cut2(n, chunk, flags, even, tail, job_pos, levels, a, 8192)
    return list(a[:m])


def test_the_ramp_belongs_to_the_job_not_to_its_first_batch():
    """round 5: the reader's first batch is a short one (6 400 reads: the GPU starts while the second batch is parsed) -- the ramp of small first chunks continues in the next
    batch instead of ending with the first"""
    assert cut2(6400, 16384, FIRST, 0, 3) == [2048, 4352]                 # chunk/8, then what is left (chunk/4 would leave too little behind it)
    assert cut2(50000, 16384, 0, 2, 3) == [8192, 13936, 13936, 13936]      # two chunks of the job are out: chunk/2 is still due
    assert cut2(50000, 16384, 0, 3, 3) == [12500] * 4                      # the ramp is over
    assert cut2(125000, 16384, FIRST | LAST, 0, 3)[:3] == [2048, 4096, 8192]
    for n in (1, 100, 5000, 6400, 20000, 50000, 1000003):
        for pos in range(0, 5):
            for flags in (0, FIRST, LAST, FIRST | LAST):
                c = cut2(n, 16384, flags, pos, 3, tail=3)
                assert sum(c) == n and min(c) >= 1 and max(c) <= 16384, (n, pos, flags, c)
