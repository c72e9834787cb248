// apply_int_u16 (the reference's `extended_numeric_types` feature) — src/kernels/arithmetic/dispatch.rs:380-387.
#include "ma_binary.hpp"

MA_DEFINE_APPLY(int, u16, uint16_t)
