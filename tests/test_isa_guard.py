"""The shipped libpce.so holds no packed fp32 instruction of the form that goes wrong beside MFMA waves (tools/isa_guard.py,
profiles/r06/multiprocess_glitch.txt).  Runs on the CPU: the library is disassembled, nothing is launched."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_no_packed_fp32_instruction_with_the_unsafe_operand_selection():
    import isa_guard
    lib = os.path.join(ROOT, "prosody-control-french-tts_amd", "libpce.so")
    rows, unsafe = isa_guard.census(lib)
    assert len(rows) > 50 and sum(r[1] for r in rows) > 1000          # (the disassembly really saw the library's kernels)
    assert not unsafe, "\n".join(f"{k}: {t}" for k, t in unsafe[:20])
    by = {r[0]: r for r in rows}
    # the three kernels that held the form keep their packed arithmetic where it is safe (STFT: hand-written selections) or carry none (PCE_NO_PK_F32)
    assert any("k_stft_raw" in k and r[1] > 100 for k, r in by.items())
    assert not any("k_logmel_frames" in k or "k_xq_fused" in k for k in by)


def test_the_guard_sees_an_unsafe_instruction_when_there_is_one(tmp_path):
    """The detector itself: a two-line kernel with the hand-written unsafe form, compiled for gfx950, must be reported."""
    import subprocess
    import isa_guard
    src = tmp_path / "bad.hip"
    src.write_text('#include <hip/hip_runtime.h>\ntypedef float v2f __attribute__((ext_vector_type(2)));\n'
                   '__global__ void k_bad(v2f *p) { v2f a = p[threadIdx.x], b = p[threadIdx.x + 64], r;\n'
                   '  asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b)); p[threadIdx.x] = r; }\n')
    so = tmp_path / "libbad.so"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", str(so), str(src)], check=True, capture_output=True)
    rows, unsafe = isa_guard.census(str(so))
    assert len(unsafe) == 1 and "k_bad" in unsafe[0][0] and "op_sel:[0,1]" in unsafe[0][1]
