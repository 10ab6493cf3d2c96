// pce_whisper_f16.hip -- the same kernels and entry points computing on fp16 operands (the reference's own arithmetic:
// openai-whisper's fp16=True default, Code/Aligners/use_whisper_timestamped.py:163).
#define PCE_OP_T _Float16
#define PCE_OP_INDEX 1
#define PCE_WFN(name) name##_f16
#include "pce_whisper_impl.inc"
