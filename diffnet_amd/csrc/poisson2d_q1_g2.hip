// 2-D Q1 fused Poisson kernels for ngp_1d = 2 (see poisson2d_q1.inl).
#define DN_NGP 2
#include "poisson2d_q1.inl"
