// apply_int_i16 (the reference's `extended_numeric_types` feature) — src/kernels/arithmetic/dispatch.rs:380-387.
#include "ma_binary.hpp"

MA_DEFINE_APPLY(int, i16, int16_t)
