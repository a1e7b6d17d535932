// 2-D Q1 fused Poisson kernels for ngp_1d = 3 (see poisson2d_q1.inl).
#define DN_NGP 3
#include "poisson2d_q1.inl"
