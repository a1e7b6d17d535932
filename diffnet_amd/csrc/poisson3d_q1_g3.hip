// 3-D Q1 fused Poisson kernels for ngp_1d = 3 (see poisson3d_q1.inl).
#define DN_NGP 3
#include "poisson3d_q1.inl"
