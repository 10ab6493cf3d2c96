// pce_whisper_bf16.hip -- the Whisper / BERT kernels and entry points (pce_whisper_impl.inc) computing on bf16 operands.
#define PCE_OP_T __bf16
#define PCE_OP_INDEX 0
#define PCE_WFN(name) name##_bf16
#include "pce_whisper_impl.inc"
