import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def engine():
    """One libpce context on cuda:0 for the whole GPU session (fails loudly without a GPU)."""
    import prosody_control_french_tts_amd as P
    eng = P.ProsodyEngine(0)
    yield eng
    eng.close()


@pytest.fixture(scope="session")
def synth16k():
    """Four deterministic 3-second 16 kHz clips + edge cases used by several parity tests."""
    import numpy as np
    from prosody_control_french_tts_amd import synth
    clips = [synth.synth_clip(i, seconds=3.0) for i in range(4)]
    clips.append(np.zeros(16000, dtype=np.int16))                                   # pure silence
    rng = np.random.default_rng(7)
    clips.append((rng.standard_normal(12345) * 3000).astype(np.int16))              # noise, odd length
    clips.append(np.full(8000, -32768, dtype=np.int16))                             # int16 minimum, DC
    t = np.arange(24000) / 16000.0
    clips.append(np.round(0.6 * 32767 * np.sin(2 * np.pi * 220.0 * t)).astype(np.int16))   # pure tone
    return clips
