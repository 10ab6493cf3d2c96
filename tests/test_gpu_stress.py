"""Race screen at the bench's batch size (tools/stress_repeat.py): in a fresh process -- cold caches, first launches -- the Whisper-small
encoder over 256 clips and a 32-step device-resident decoding loop are repeated, and every repetition must reproduce the first bit for
bit.  The persistent GEMM's ping-pong K-step, the attention kernels and the decoding loop all rest on counted waits and barriers placed
by hand; a read before its LDS-DMA has landed shows up here as a repetition that differs (it did, once, for a variant that was dropped)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("operands", ["fp16", "bf16"])
def test_encoder_and_decoding_loop_repeat_bit_for_bit(operands):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_repeat.py"), "3", operands], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 differ from the first" in r.stdout
