"""Debug aid: is the persistent 256 x 256 GEMM deterministic, and independent of how many rows follow?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prosody_control_french_tts_amd as pkg
rng = np.random.default_rng(0)
with pkg.ProsodyEngine(0) as eng:
    for (N, K, epi) in [(1536, 768, 0), (768, 768, 0), (3072, 768, 1), (768, 3072, 0), (768, 768, 2)]:
        M1, M2 = 6000, 96000
        A = rng.standard_normal((M2, K)).astype(np.float32); B = (rng.standard_normal((N, K)) * 0.05).astype(np.float32); bias = rng.standard_normal(N).astype(np.float32)
        small = [eng.selftest_gemm(A[:M1], B, bias, epi, 1500, 1536) for _ in range(3)]
        big = [eng.selftest_gemm(A, B, bias, epi, 1500, 1536) for _ in range(2)]
        def head(x): return x[:4] if epi == 2 else x[:M1]
        def diff(x, y):
            d = np.argwhere(head(x) != head(y))
            return len(d), (np.unique(d[:, 0])[:6], np.unique(d[:, -1] % 256)[:16]) if len(d) else ""
        print((N, K, epi), "small reruns", diff(small[0], small[1])[0], diff(small[0], small[2])[0], "big reruns", diff(big[0], big[1])[0], "small vs big", diff(small[0], big[0]))
