import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """tests/test_gpu_world2.py runs FIRST: its tests start several rank processes on the one GPU, and this (parent) process must not hold an
    engine context of its own beside them -- the session ``engine`` below is created by the first test that asks for it, i.e. after them."""
    first = [it for it in items if it.nodeid.startswith("tests/test_gpu_world2.py") or "/test_gpu_world2.py" in it.nodeid]
    if first:
        rest = [it for it in items if it not in first]
        items[:] = first + rest


@pytest.fixture(scope="session")
def engine():
    """One libpce context on cuda:0 for the whole GPU session (fails loudly without a GPU)."""
    import prosody_control_french_tts_amd as P
    eng = P.ProsodyEngine(0)
    yield eng
    eng.close()


DEFAULT_OPERANDS = "fp16-resid16"


@pytest.fixture(autouse=True)
def _default_operands(request):
    """GPU tests start on the product's default operand mode (fp16 operands + fp16 residual stream) whatever the previous test selected."""
    if "engine" in request.fixturenames:
        request.getfixturevalue("engine").whisper_set_operands(os.environ.get("PCE_TEST_OPERANDS", DEFAULT_OPERANDS))
    yield


# what the rounding of a 16-bit OUTPUT allows per operand type: unit-test bounds of the GEMM / attention kernels
# enc_l2 / enc_max: a whole encoder stack against the fp32 restatement (relative L2; worst element in units of the output's sigma)
OPERANDS = {"bf16": dict(torch="bfloat16", l2=4e-3, rel=2.0 ** -7, abs=1e-2, attn_abs=1e-2, attn_l2=6e-3, enc_l2=2e-2, enc_max=6e-2),
            "fp16": dict(torch="float16", l2=6e-4, rel=2.0 ** -10, abs=2e-3, attn_abs=2e-3, attn_l2=1e-3, enc_l2=1.5e-3, enc_max=1.2e-2)}
# the product default: fp16 operands + the encoder's residual stream in fp16 (24 more roundings at Whisper-small depth: measured 1.04e-3 / 0.009 sigma)
OPERANDS["fp16-resid16"] = dict(OPERANDS["fp16"], enc_l2=2.5e-3, enc_max=2.5e-2)


@pytest.fixture(params=["fp16-resid16", "fp16", "bf16"])
def ops(request, engine):
    """Runs a test once per operand type of the Whisper / BERT kernels; yields that type's name, torch dtype name and bounds."""
    engine.whisper_set_operands(request.param)
    yield dict(OPERANDS[request.param], name=request.param)
    engine.whisper_set_operands(DEFAULT_OPERANDS)


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
