// 3-D Q1 fused Poisson kernels for ngp_1d = 2 (see poisson3d_q1.inl).
#define DN_NGP 2
#include "poisson3d_q1.inl"
