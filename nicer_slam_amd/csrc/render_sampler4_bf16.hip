// render_sampler4.hip compiled with plain bf16 GEMM operands (fp32 accumulate): the optional "bf16 MLP" precision mode.
// Everything in namespace nsa becomes nsa_bf16, every entry point nsa_xxx becomes nsa_xxx_bf16.
#define NSA_PIECES 1
#define nsa nsa_bf16
#define NSA_ENTRY(x) x##_bf16
#include "render_sampler4.hip"
