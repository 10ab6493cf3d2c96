"""Debug aid: fc2-shaped product (K = 3072, N = 768) at the full C3 row count, repeated."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prosody_control_french_tts_amd as pkg
rng = np.random.default_rng(0)
M2 = int(sys.argv[1]) if len(sys.argv) > 1 else 384000
with pkg.ProsodyEngine(0) as eng:
    for (N, K, epi) in [(768, 3072, 0), (3072, 768, 1)]:
        M1 = 6000
        A = rng.standard_normal((M2, K), dtype=np.float32); B = (rng.standard_normal((N, K)) * 0.05).astype(np.float32); bias = rng.standard_normal(N).astype(np.float32)
        small = eng.selftest_gemm(A[:M1], B, bias, epi, 1500, 1536)
        for rep in range(4):
            big = eng.selftest_gemm(A, B, bias, epi, 1500, 1536)
            d = np.argwhere(big[:M1] != small)
            print((N, K, epi), "rep", rep, "diffs in the first rows", len(d), "rows", np.unique(d[:, 0])[[0, -1]] if len(d) else "", "cols", np.unique(d[:, 1])[[0, -1]] if len(d) else "")
            if len(d):
                r = d[0, 0]; print("   row", r, "got", big[r, :6], "want", small[r, :6], "bias", bias[:6])
            tail = eng.selftest_gemm(A[-M1:], B, bias, epi, 1500, 1536)
            d = np.argwhere(big[-M1:] != tail)
            print("      diffs in the last rows", len(d))
