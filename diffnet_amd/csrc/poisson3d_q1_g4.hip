// 3-D Q1 fused Poisson kernels for ngp_1d = 4 (see poisson3d_q1.inl).
#define DN_NGP 4
#include "poisson3d_q1.inl"
