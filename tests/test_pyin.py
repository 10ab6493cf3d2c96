"""pYIN restatement (oracle/pyin_oracle.py, third party behaviour restated: parity unpinned) on analytic cases, and
the host plan tables of the product against the restatement's dense matrices."""
import numpy as np

from oracle import pyin_oracle as PO
from prosody_control_french_tts_amd.visualisation import acoustic_analysis as AA


def test_known_answers():
    sr = 16000
    t = np.arange(int(0.8 * sr)) / sr
    for f in (110.0, 220.0, 523.25):
        y = (0.4 * np.sin(2 * np.pi * f * t) + 0.2 * np.sin(2 * np.pi * 2 * f * t)).astype(np.float32)
        f0, voiced, vp = PO.pyin(y, sr)
        mid = slice(8, len(f0) - 8)
        assert voiced[mid].all() and np.all(vp[mid] > 0.9)
        assert np.max(np.abs(12 * np.log2(f0[mid] / f))) <= 0.06            # within half a 0.1-semitone bin
    f0, voiced, vp = PO.pyin(np.zeros(sr // 2, np.float32), sr)
    assert not voiced.any() and np.all(np.isnan(f0)) and np.all(vp == 0)
    rng = np.random.default_rng(0)
    f0, voiced, vp = PO.pyin((0.1 * rng.standard_normal(sr // 2)).astype(np.float32), sr)
    assert voiced.mean() < 0.2
    assert len(f0) == 1 + (sr // 2) // 256


def test_fft_route_and_exact_difference_function_agree():
    sr = 16000
    rng = np.random.default_rng(1)
    t = np.arange(int(0.6 * sr)) / sr
    y = (0.3 * np.sin(2 * np.pi * (180 + 40 * t) * t) + 0.02 * rng.standard_normal(len(t))).astype(np.float32)
    y = (np.round(y * 32768) / 32768).astype(np.float32)                    # int16-valued samples, as librosa.load gives for 16-bit WAVs
    a = PO.pyin(y, sr, want_intermediate=True)
    b = PO.pyin(y, sr, want_intermediate=True, exact=True)
    assert np.max(np.abs(a[3]["yin"] - b[3]["yin"])) < 1e-3                 # float32 FFT noise around the exact values
    same = (a[0] == b[0]) | (np.isnan(a[0]) & np.isnan(b[0]))
    assert same.mean() >= 0.97 and np.max(np.abs(a[2] - b[2])) < 0.1


def test_host_plan_tables_equal_the_dense_matrices():
    for sr in (16000, 44100):
        plan, tables, freqs = AA.pyin_plan(sr)
        nb, W = plan.n_pitch_bins, plan.trans_width
        half = W // 2
        assert nb == 608 and plan.min_period == int(np.floor(sr / 2000.0)) and W == (71 if sr == 16000 else 31)
        T = PO.transition_local_triangle(nb, W)
        full = np.log(np.kron(np.array([[0.99, 0.01], [0.01, 0.99]]), T) + PO.TINY64)
        o = 100 + 100 + 101 + 2 * (AA.MAX_TROUGHS + 1)
        o += o % 2
        pairs = tables[o:].reshape(W, nb, 2); lt = pairs[:, :, 0]; ls = pairs[:, :, 1]
        for e in range(-half, half + 1):
            ks = np.arange(max(0, -e), min(nb, nb - e))
            assert np.array_equal(lt[e + half, ks], full[ks, ks + e]) and np.array_equal(ls[e + half, ks], full[ks, nb + ks + e])
        th = np.linspace(0, 1, 101)
        assert np.array_equal(tables[:100], th[1:])
        assert np.allclose(tables[100:200], np.diff(PO.beta_cdf_2_18(th)), rtol=0, atol=0)
        assert np.allclose(freqs, 60.0 * 2 ** (np.arange(nb) / 120.0))
