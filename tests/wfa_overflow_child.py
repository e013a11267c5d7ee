"""child process of test_wfa_ladder_list_overflow: MGA_WFA_ARRIVALS_PCT is read once per process"""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import minigraph_amd as mga
import refbind as rb

rng = np.random.default_rng(77)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
T, Q = [], []
for it in range(12000):  # short and unrelated: every pair starts in the narrowest window and climbs several rungs
    L = int(rng.integers(30, 70))
    T.append(bytes(rng.choice(ACGT, size=L).tobytes()))
    Q.append(bytes(rng.choice(ACGT, size=int(rng.integers(30, 70))).tobytes()))
ora = rb.Oracle()
sc, cg = mga.wfa_batch(T, Q)
for i in range(len(T)):
    es, ec = ora.wfa(T[i], Q[i])
    assert es == sc[i] and np.array_equal(ec, cg[i]), (i, es, sc[i])
print("OVERFLOW-OK")
