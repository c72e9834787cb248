// apply_int_i32 and its fused scalar-broadcast forms — src/kernels/arithmetic/dispatch.rs:65-133, :376-379.
#include "ma_binary.hpp"

MA_DEFINE_APPLY(int, i32, int32_t)
MA_DEFINE_APPLY_TWO_MASKS(i32, int32_t)
