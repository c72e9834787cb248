// apply_float_f64, apply_fma_f64 and the fused scalar-broadcast forms —
// src/kernels/arithmetic/dispatch.rs:138-290, :389-418.
#include "ma_binary.hpp"

MA_DEFINE_APPLY(float, f64, double)
MA_DEFINE_APPLY_FMA(f64, double)
