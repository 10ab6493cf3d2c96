"""MI355X-native prosody-extraction + alignment engine behind the ``Code/Pipeline`` +
``Code/Aligners`` API of hi-paris/Prosody-Control-French-TTS.

The arithmetic runs in ``libpce.so`` (hand-written HIP for gfx950, C ABI in
``include/pce.h``); this package is the thin host side: ctypes binding
(:mod:`.engine`), the reference's slicing/decoding rules (:mod:`.hostrules`) and drop-in
mirrors of the reference modules (``Pipeline``, ``Aligners``, ``Preprocessing``).
There is no CPU fallback: constructing :class:`ProsodyEngine` without the built library
or without a GPU raises.
"""
from .engine import ProsodyEngine, PceError, build_native, native_library_path, PitchParams  # noqa: F401

__all__ = ["ProsodyEngine", "PceError", "build_native", "native_library_path", "PitchParams"]
