// render_colour.hip compiled with plain bf16 GEMM operands (fp32 accumulate): the optional "bf16 MLP" precision mode.
// Everything in namespace nsa becomes nsa_bf16, every entry point nsa_xxx becomes nsa_xxx_bf16 (internal: reached through
// the fp32 entry points when nsa_grid_t.precision == 1).
#define NSA_PIECES 1
#define nsa nsa_bf16
#define NSA_ENTRY(x) x##_bf16
#include "render_colour.hip"
