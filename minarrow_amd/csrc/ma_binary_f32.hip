// apply_float_f32, apply_fma_f32 and the fused scalar-broadcast forms —
// src/kernels/arithmetic/dispatch.rs:138-290, :389-418.
#include "ma_binary.hpp"

MA_DEFINE_APPLY(float, f32, float)
MA_DEFINE_APPLY_FMA(f32, float)
