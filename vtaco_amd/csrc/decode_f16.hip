// Split-f16 ("f16x3") instantiations of the decode kernels (decode.hip): their own translation unit because the
// relu + hi/lo split relies on v_fma_mixlo_f16 / v_fma_mixhi_f16, which LLVM selects only when the SLP vectoriser has
// not already paired the two f32 fmas of a register pair into a v_pk_fma_f32 (Makefile: -fno-slp-vectorize here).
#define VT_DECODE_F16_TU 1
#include "decode.hip"
