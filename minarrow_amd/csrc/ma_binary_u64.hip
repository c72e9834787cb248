// apply_int_u64 and its fused scalar-broadcast forms — src/kernels/arithmetic/dispatch.rs:65-133, :376-379.
#include "ma_binary.hpp"

MA_DEFINE_APPLY(int, u64, uint64_t)
MA_DEFINE_APPLY_TWO_MASKS(u64, uint64_t)
