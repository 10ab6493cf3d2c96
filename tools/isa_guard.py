"""ISA guard of libpce.so: no packed fp32 instruction with the operand selection that goes wrong beside MFMA waves.

Finding of round 6 (profiles/r06/multiprocess_glitch.txt; reduced case tools/lab/pk_victim.hip): on the MI355X boxes of this pool
``v_pk_mul_f32`` / ``v_pk_add_f32`` / ``v_pk_fma_f32`` with ``op_sel:[0,1,...]`` -- the LOW result lane reads src0's low half and src1's
high half, whatever ``op_sel_hi`` says -- return wrong values in lanes 48..63 while ANOTHER wave on the same SIMD executes MFMA
instructions: another stream of the process, or another process on the device (the reference's default is five worker processes on
one GPU, /root/reference config.yaml:57-58).  Every other selection ([0,0], [1,0], [1,1]) measured clean.  The compiler forms the bad
selection on its own for complex products and pair sums, so the library is checked as BUILT: this script pulls the gfx950 code objects
out of the shared library's ``.hip_fatbin`` section, disassembles them and lists the offending instructions per kernel.

usage: python tools/isa_guard.py [path/to/libpce.so]        exit code 1 when an instruction of the unsafe form is present
"""
from __future__ import annotations

import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM_BIN = os.environ.get("PCE_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
PK = re.compile(r"\b(v_pk_(?:mul|add|fma)_f32)\s+([^\n/;]*)")
UNSAFE = re.compile(r"op_sel:\[0,1")


def _section(blob: bytes, name: bytes) -> bytes:
    """Bytes of one section of a 64-bit little-endian ELF (no third-party ELF reader in the image)."""
    if blob[:4] != b"\x7fELF" or blob[4] != 2:
        raise ValueError("not a 64-bit ELF file")
    shoff, = struct.unpack_from("<Q", blob, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", blob, 0x3A)
    def sh(i):
        return struct.unpack_from("<IIQQQQIIQQ", blob, shoff + i * shentsize)
    stroff, strsize = sh(shstrndx)[4], sh(shstrndx)[5]
    strtab = blob[stroff:stroff + strsize]
    for i in range(shnum):
        h = sh(i)
        if strtab[h[0]:strtab.index(b"\0", h[0])] == name:
            return blob[h[4]:h[4] + h[5]]
    raise ValueError(f"no section {name.decode()}")


def code_objects(lib_path: str, arch: str = "gfx950"):
    """Device ELF images for ``arch`` inside a HIP shared library: the fat binary is a run of clang offload bundles (one per translation unit)."""
    fat = _section(open(lib_path, "rb").read(), b".hip_fatbin")
    out, pos = [], fat.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", fat, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", fat, p)
            triple = fat[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if arch in triple and size:
                out.append(fat[pos + off:pos + off + size])
        pos = fat.find(MAGIC, pos + len(MAGIC))
    return out


def census(lib_path: str):
    """-> (rows, unsafe): rows = [(kernel, packed fp32 instructions, with any op_sel, unsafe form)], unsafe = [(kernel, instruction text)]."""
    objdump = os.path.join(LLVM_BIN, "llvm-objdump")
    rows, unsafe = [], []
    with tempfile.TemporaryDirectory() as td:
        for k, img in enumerate(code_objects(lib_path)):
            path = os.path.join(td, f"co{k}.elf")
            with open(path, "wb") as f:
                f.write(img)
            txt = subprocess.run([objdump, "-d", path], capture_output=True, text=True, check=True).stdout
            parts = re.split(r"\n[0-9a-f]+ <([^>]+)>:\n", txt)
            for i in range(1, len(parts), 2):
                name, body = parts[i], parts[i + 1]
                pk = PK.findall(body)
                if not pk:
                    continue
                bad = [(op, a.strip()) for op, a in pk if UNSAFE.search(a)]
                rows.append((name, len(pk), sum(1 for _, a in pk if "op_sel:" in a), len(bad)))
                unsafe += [(name, f"{op} {a}") for op, a in bad]
    return rows, unsafe


def main(argv):
    lib = argv[1] if len(argv) > 1 else os.path.join(ROOT, "prosody-control-french-tts_amd", "libpce.so")
    rows, unsafe = census(lib)
    n_pk = sum(r[1] for r in rows)
    print(f"{lib}: {len(rows)} kernels hold {n_pk} packed fp32 instructions, {sum(r[2] for r in rows)} of them with an op_sel modifier, "
          f"{len(unsafe)} of the unsafe form op_sel:[0,1,..]")
    for name, text in unsafe[:40]:
        print(f"  UNSAFE  {name[:80]}: {text}")
    return 1 if unsafe else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
