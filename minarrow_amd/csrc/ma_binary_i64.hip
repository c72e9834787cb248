// apply_int_i64 and its fused scalar-broadcast forms — src/kernels/arithmetic/dispatch.rs:65-133, :376-379.
#include "ma_binary.hpp"

MA_DEFINE_APPLY(int, i64, int64_t)
MA_DEFINE_APPLY_TWO_MASKS(i64, int64_t)
