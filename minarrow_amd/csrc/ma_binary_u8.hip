// apply_int_u8 (the reference's `extended_numeric_types` feature) — src/kernels/arithmetic/dispatch.rs:380-387.
#include "ma_binary.hpp"

MA_DEFINE_APPLY(int, u8, uint8_t)
