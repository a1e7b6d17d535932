// Per-element arithmetic of the fused Poisson operator (energy density, its gradient / the weak-form
// residual) for one structured Q_P element, evaluated entirely in registers.
//
// What the reference does per element and Gauss point with dense nd tables
// (DiffNet/DiffNetFEM.py:196-227, 405-453 tables; IBN_2D.py:123-131, 12_klsum.py:96-122 use) is done
// here by sum factorisation: the 1-D basis tables are applied axis by axis.  For Q1 the 1-D
// basis is a lerp, phi(xi) = (1-b, b) with b = (1+xi)/2 and phi' = (-1/2, +1/2), so every 1-D
// contraction is one subtraction plus one FMA per Gauss point and the derivative is the
// difference itself; that specialisation is what keeps the kernel near the HBM roofline
// instead of the VALU roofline.
//
// nu and f are always interpolated from nodal values (an absent nu is the constant field 1, an
// absent f the constant 0: exact under interpolation), so the element code has no runtime flags;
// FGP selects forcing given directly at the Gauss points (e8_2d_poisson_mms.py:46).
//
// Conventions: W_g = w[jg] * wx[ig] (* w[kg]); the returned energy parts are
//   e1 = sum_g W_g nu_g |grad u|^2_g      (caller multiplies by c)
//   e2 = sum_g W_g f_g u_g
// and the nodal contributions are g_a = sum_g W_g ( alpha nu_g gradN_a.grad u_g - beta N_a f_g ).
#pragma once
#include "dn_common.h"

namespace dn {

// Value types of the element arithmetic: float, or v2f = two independent elements in the halves of a 64-bit register pair.  On gfx950 a
// packed fp32 instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) costs a SIMD ~3.4 cycles per wave whatever its operands, a plain
// one 2.9 (registers only) to 4.0 (SGPR operand) -- per element-operation 1.7 against ~3.2 at this code's operand mix
// (tools/micro/valu_pk.hip, profiles/r3_valu_pk.txt).  SGPR table entries broadcast into both halves through op_sel, for free.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float vfma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ v2f vfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f vfma(float a, v2f b, v2f c) { return __builtin_elementwise_fma((v2f)(a), b, c); }

// Kernel-argument resident (SGPR) tables: everything wave-uniform the element code needs.
struct ElemTab {
    float b[4][4];       // b[ig][ib]  = phi_ib(xi_ig)
    float dx[4][4];      // dbasis[ig][ib] * 2/hx   (generic-degree path)
    float dy[4][4];
    float dz[4][4];
    float w[4];          // 1-D Gauss weights
    float wx[4];         // w[ig] * wscale (the x axis carries the Jacobian / user scale)
    float w2[4][4];      // w2[jg][ig] = w[jg] * wx[ig]
    float hs[3];         // Q1 path: 0.5 * 2/h_d = 1/h_d
    float ahs[3];        // alpha * hs[d]
    float alpha, beta, c;
    // Q1 marching kernels: moments of the 1-D rule against the lerp weight b = phi_1(xi):
    //   m[r] = sum_g w[g] * b[g]^r  (r = 0,1,2);  kx[r][ig] = wx[ig] * m[r]
    float m[4];          // m[r] = sum_g w[g] * b[g]^r, r = 0..3
    float mxs[4];        // m[r] * wscale (the x axis carries the Jacobian / user scale): closed-form 2-D Q1 kernel
    float kx[3][4];
    float q1c[4];        // alpha*hs0^2, alpha*hs1^2, hs0^2, hs1^2  (2-D Q1 layer: derivative scales applied once per element)
    // 3-D Q1 marching kernel, second form (q1_layer_3d_w): the in-plane weights live in the staged coefficient planes, the
    // user scale (wscale) in kap[] and in the final energy scale, the 1/h factors are applied once per element.
    float wb[4];         // w[g] * b[g]           (pure 1-D weights: no wscale)
    float kap[3];        // alpha * wscale * hs[d]^2   cotangent scale of direction d
    float hs2[3];        // hs[d]^2                    energy scale of direction d (wscale is applied in finish_sums)
    float m01, m12;      // m[0] - m[1], m[1] - m[2]
    float esc;           // wscale: factor of both energy sums of the second-form kernel (1 for the other kernels)
    float nbw;           // -beta * wscale
    // closed-form 2-D Q1 kernel: the 1-D element mass matrix [[c00, c01], [c01, c11]] = sum_g w_g [(1-b)^2, b(1-b); b(1-b), b^2] of the
    // rule (from its moments, in double): the forcing term is (mass_x (x) mass_y) f, applied axis by axis (x carries wscale)
    float q1mx[3], q1my[3];   // c00, c01, c11
};

// ---------------------------------------------------------------------------------------------
// 2-D element.  u/nu/f: nodal values [jb][ib]; fg: forcing at Gauss points [jg*NGP+ig] (FGP).
// ---------------------------------------------------------------------------------------------
template <int P, int NGP, bool FGP>
__device__ __forceinline__ void elem2d(const ElemTab& T, const float (&u)[P + 1][P + 1], const float (&nu)[P + 1][P + 1],
                                       const float (&f)[P + 1][P + 1], const float* fg, float (&g)[P + 1][P + 1],
                                       float& e1, float& e2) {
    constexpr int NB = P + 1;
    float a1 = 0.f, a2 = 0.f;
    static_assert(P >= 2, "Q1 elements run through the marching kernels (q1_layer_2d / q1_layer_3d)");
    // Table-driven sum factorisation, one x-Gauss point at a time: x-stage of u / nu / f for that point (per node row),
    // its NGP y-points (energy density, cotangents, y-transpose), then the x-transpose of that point straight into g.
    // The live set is one point's stage values and cotangents instead of all points'.
#pragma unroll
    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
        for (int ib = 0; ib < NB; ++ib) g[jb][ib] = 0.f;
#pragma unroll
    for (int ig = 0; ig < NGP; ++ig) {
        float tv[NB], td[NB], tn[NB], tf[NB], rv[NB], rd[NB];
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
            float a = 0.f, d = 0.f, n = 0.f, q = 0.f;
#pragma unroll
            for (int ib = 0; ib < NB; ++ib) {
                a = fmaf(T.b[ig][ib], u[jb][ib], a);
                d = fmaf(T.dx[ig][ib], u[jb][ib], d);
                n = fmaf(T.b[ig][ib], nu[jb][ib], n);
                if constexpr (!FGP) q = fmaf(T.b[ig][ib], f[jb][ib], q);
            }
            tv[jb] = a; td[jb] = d; tn[jb] = n; tf[jb] = q; rv[jb] = 0.f; rd[jb] = 0.f;
        }
#pragma unroll
        for (int jg = 0; jg < NGP; ++jg) {
            float val = 0.f, ux = 0.f, uy = 0.f, nuv = 0.f, fv = 0.f;
#pragma unroll
            for (int jb = 0; jb < NB; ++jb) {
                val = fmaf(T.b[jg][jb], tv[jb], val);
                ux = fmaf(T.b[jg][jb], td[jb], ux);
                uy = fmaf(T.dy[jg][jb], tv[jb], uy);
                nuv = fmaf(T.b[jg][jb], tn[jb], nuv);
                if constexpr (!FGP) fv = fmaf(T.b[jg][jb], tf[jb], fv);
            }
            if constexpr (FGP) fv = fg[jg * NGP + ig];
            const float Wn = T.w2[jg][ig] * nuv, Wf = T.w2[jg][ig] * fv;
            a1 = fmaf(Wn, ux * ux + uy * uy, a1);
            a2 = fmaf(Wf, val, a2);
            const float qx = T.alpha * Wn * ux, qy = T.alpha * Wn * uy, qv = -T.beta * Wf;
#pragma unroll
            for (int jb = 0; jb < NB; ++jb) {
                rv[jb] = fmaf(T.b[jg][jb], qv, rv[jb]);
                rv[jb] = fmaf(T.dy[jg][jb], qy, rv[jb]);
                rd[jb] = fmaf(T.b[jg][jb], qx, rd[jb]);
            }
        }
#pragma unroll
        for (int jb = 0; jb < NB; ++jb)
#pragma unroll
            for (int ib = 0; ib < NB; ++ib) {
                g[jb][ib] = fmaf(T.b[ig][ib], rv[jb], g[jb][ib]);
                g[jb][ib] = fmaf(T.dx[ig][ib], rd[jb], g[jb][ib]);
            }
    }
    e1 = a1;
    e2 = a2;
}

// ---------------------------------------------------------------------------------------------
// 2-D Q1, marching form.  One element between node rows j (suffix 0) and j+1 (suffix 1), given the
// x-stage values of both rows at the element's x-Gauss points:
//   TU[ig] = lerp_x(u)(xi_ig), DX = u[e+1]-u[e];  TN, TF likewise for nu and f.
// Because nu, f and u are bilinear and the rule is a tensor product, every sum over the y-Gauss points
// collapses onto the three moments m[0..2] of the 1-D rule (ElemTab), so the work is O(NGP), not O(NGP^2).
// Outputs: cotangents of the x-stage values of both rows (ct0/ct1 for TU, cdx0/cdx1 for DX) -- the caller
// adds the carried cotangent of the shared row and applies the x-stage transpose once per row -- and the
// two energy parts (see the conventions at the top of this file).
// ---------------------------------------------------------------------------------------------
template <int NGP, bool FGP>
__device__ __forceinline__ void q1_layer_2d(const ElemTab& T, const float (&TU0)[NGP], const float (&TU1)[NGP], const float DX0,
                                            const float DX1, const float (&TN0)[NGP], const float (&TN1)[NGP],
                                            const float (&TF0)[NGP], const float (&TF1)[NGP], const float* fg,
                                            float (&ct0)[NGP], float (&ct1)[NGP], float& cdx0, float& cdx1, float& e1,
                                            float& e2) {
    // The 1/h factors of the derivatives are applied once per element (q1c[]) instead of once per Gauss point:
    //   e1 = hs0^2 * sum_jg w Qx' r_jg^2 + hs1^2 * sum_ig Qy d_ig^2,   r = raw x-difference lerp, d = raw y-difference
    float a1x = 0.f, a1y = 0.f, a2 = 0.f;
    // nu moments along x
    float S0 = 0.f, S1 = 0.f, dyn[NGP];
#pragma unroll
    for (int i = 0; i < NGP; ++i) {
        dyn[i] = TN1[i] - TN0[i];
        S0 = fmaf(T.wx[i], TN0[i], S0);
        S1 = fmaf(T.wx[i], dyn[i], S1);
    }
    // x-derivative depends on the y-Gauss index only
    const float ddx = DX1 - DX0;
    float c1s = 0.f, css = 0.f;
#pragma unroll
    for (int jg = 0; jg < NGP; ++jg) {
        const float r = fmaf(T.b[jg][1], ddx, DX0);
        const float t = fmaf(T.b[jg][1], S1, S0) * r;          // (nu-moment) * raw derivative
        a1x = fmaf(T.w[jg] * t, r, a1x);
        const float cx = T.w[jg] * t;
        css += cx;
        c1s = fmaf(T.b[jg][1], cx, c1s);
    }
    const float kx = T.q1c[0];                                   // alpha * hs0^2
    cdx1 = kx * c1s;
    cdx0 = kx * css - cdx1;
    const float nb = -T.beta, ky = T.q1c[1];                     // alpha * hs1^2
#pragma unroll
    for (int ig = 0; ig < NGP; ++ig) {
        const float dyv = TU1[ig] - TU0[ig];
        const float t = fmaf(T.kx[1][ig], dyn[ig], T.kx[0][ig] * TN0[ig]) * dyv;
        a1y = fmaf(t, dyv, a1y);
        float cs, c1;
        if constexpr (FGP) {
            cs = 0.f; c1 = 0.f;
#pragma unroll
            for (int jg = 0; jg < NGP; ++jg) {
                const float wf = T.w2[jg][ig] * fg[jg * NGP + ig];
                cs += wf;
                c1 = fmaf(T.b[jg][1], wf, c1);
            }
        } else {
            const float dyf = TF1[ig] - TF0[ig];
            cs = fmaf(T.kx[1][ig], dyf, T.kx[0][ig] * TF0[ig]);
            c1 = fmaf(T.kx[2][ig], dyf, T.kx[1][ig] * TF0[ig]);
        }
        a2 = fmaf(cs, TU0[ig], a2);
        a2 = fmaf(c1, dyv, a2);
        ct1[ig] = fmaf(nb, c1, ky * t);
        ct0[ig] = fmaf(nb, cs, -ct1[ig]);
    }
    e1 = fmaf(T.q1c[2], a1x, T.q1c[3] * a1y);                    // hs0^2 * a1x + hs1^2 * a1y
    e2 = a2;
}

// ---------------------------------------------------------------------------------------------
// 3-D Q1, marching form.  One element between node planes k (L) and k+1 (U), given each plane's in-plane stage at
// the element's in-plane Gauss points m = (jg, ig):
//   VU[jg][ig] value of u, VX[jg] = lerp_y of the x-differences, VY[ig] = y-difference of the x-lerped rows,
//   VN / VF values of nu / f.
// As in 2-D every sum over the z-Gauss points collapses onto the moments m[0..2] of the 1-D rule, and the sums over
// one in-plane index onto the small vectors A0/A1 (per jg) and B0/B1 (per ig): O(NGP^2) work per element.
// Outputs: cotangents of both planes' stage values (…L / …U) and the two energy parts.
// ---------------------------------------------------------------------------------------------
template <int NGP, bool FGP>
__device__ __forceinline__ void q1_layer_3d(const ElemTab& T, const float (&LU)[NGP][NGP], const float (&UU)[NGP][NGP],
                                            const float (&LX)[NGP], const float (&UX)[NGP], const float (&LY)[NGP],
                                            const float (&UY)[NGP], const float (&LN)[NGP][NGP], const float (&UN)[NGP][NGP],
                                            const float (&LF)[NGP][NGP], const float (&UF)[NGP][NGP], const float* fg,
                                            float (&cUL)[NGP][NGP], float (&cUU)[NGP][NGP], float (&cXL)[NGP], float (&cXU)[NGP],
                                            float (&cYL)[NGP], float (&cYU)[NGP], float& e1, float& e2) {
    float a1 = 0.f, a2 = 0.f;
    float A0[NGP], A1[NGP], B0[NGP], B1[NGP];
#pragma unroll
    for (int i = 0; i < NGP; ++i) A0[i] = A1[i] = B0[i] = B1[i] = 0.f;
    const float nb = -T.beta;
#pragma unroll
    for (int jg = 0; jg < NGP; ++jg) {
#pragma unroll
        for (int ig = 0; ig < NGP; ++ig) {
            const float w2 = T.w2[jg][ig];
            const float dzu = UU[jg][ig] - LU[jg][ig];
            const float uz = T.hs[2] * dzu;
            const float dzn = UN[jg][ig] - LN[jg][ig];
            const float wn0 = w2 * LN[jg][ig], wn1 = w2 * dzn;
            A0[jg] += wn0; A1[jg] += wn1; B0[ig] += wn0; B1[ig] += wn1;
            const float Qz = fmaf(T.m[1], wn1, T.m[0] * wn0);
            const float qzu = Qz * uz;
            a1 = fmaf(qzu, uz, a1);
            const float cz = T.ahs[2] * qzu;
            float cs, c1;
            if constexpr (FGP) {
                cs = 0.f; c1 = 0.f;
#pragma unroll
                for (int kg = 0; kg < NGP; ++kg) {
                    const float wf = T.w[kg] * w2 * fg[(kg * NGP + jg) * NGP + ig];
                    cs += wf;
                    c1 = fmaf(T.b[kg][1], wf, c1);
                }
            } else {
                const float dzf = UF[jg][ig] - LF[jg][ig];
                const float wf0 = w2 * LF[jg][ig], wf1 = w2 * dzf;
                cs = fmaf(T.m[1], wf1, T.m[0] * wf0);
                c1 = fmaf(T.m[2], wf1, T.m[1] * wf0);
            }
            a2 = fmaf(cs, LU[jg][ig], a2);
            a2 = fmaf(c1, dzu, a2);
            cUU[jg][ig] = fmaf(nb, c1, cz);
            cUL[jg][ig] = fmaf(nb, cs, -cUU[jg][ig]);
        }
    }
#pragma unroll
    for (int jg = 0; jg < NGP; ++jg) {
        const float dv = UX[jg] - LX[jg];
        float sx = 0.f, tx = 0.f;
#pragma unroll
        for (int kg = 0; kg < NGP; ++kg) {
            const float ux = T.hs[0] * fmaf(T.b[kg][1], dv, LX[jg]);
            const float Qx = T.w[kg] * fmaf(T.b[kg][1], A1[jg], A0[jg]);
            const float qxu = Qx * ux;
            a1 = fmaf(qxu, ux, a1);
            const float cx = T.ahs[0] * qxu;
            sx += cx;
            tx = fmaf(T.b[kg][1], cx, tx);
        }
        cXU[jg] = tx;
        cXL[jg] = sx - tx;
    }
#pragma unroll
    for (int ig = 0; ig < NGP; ++ig) {
        const float dv = UY[ig] - LY[ig];
        float sy = 0.f, ty = 0.f;
#pragma unroll
        for (int kg = 0; kg < NGP; ++kg) {
            const float uy = T.hs[1] * fmaf(T.b[kg][1], dv, LY[ig]);
            const float Qy = T.w[kg] * fmaf(T.b[kg][1], B1[ig], B0[ig]);
            const float qyu = Qy * uy;
            a1 = fmaf(qyu, uy, a1);
            const float cy = T.ahs[1] * qyu;
            sy += cy;
            ty = fmaf(T.b[kg][1], cy, ty);
        }
        cYU[ig] = ty;
        cYL[ig] = sy - ty;
    }
    e1 = a1;
    e2 = a2;
}

// ---------------------------------------------------------------------------------------------
// 3-D Q1, marching form, second version (fewer instructions).  Differences to q1_layer_3d:
//   * the coefficient planes arrive WEIGHTED: VNw = w_j w_i nu, VFw = w_j w_i f at the in-plane Gauss points (weights applied
//     once per plane instead of once per layer, plane and term);
//   * raw differences are used throughout; 1/h^2, alpha and the user scale are applied once per element (T.kap, T.hs2);
//   * the cotangents of the lower plane are returned already summed with the carried cotangents of the layer below.
// UW: all weights of the rule are 1 (the exact 2-point rule): weight multiplications vanish at compile time.
// Outputs: tU/tX/tY = complete cotangents of the LOWER plane's stage values (carry included), cU/cX/cY (in/out) = carried
// cotangents: in = from the layer below, out = this layer's contribution to the UPPER plane.  e1, e2 without wscale.
// ---------------------------------------------------------------------------------------------
template <int NGP, bool FGP, bool HAS_F, bool UW, typename V = float>
__device__ __forceinline__ void q1_layer_3d_w(const ElemTab& T, const V (&LU)[NGP][NGP], const V (&UU)[NGP][NGP],
                                              const V (&LX)[NGP], const V (&UX)[NGP], const V (&LY)[NGP],
                                              const V (&UY)[NGP], const V (&LN)[NGP][NGP], const V (&UN)[NGP][NGP],
                                              const V (&LF)[NGP][NGP], const V (&UF)[NGP][NGP],
                                              const float* fg, V (&cU)[NGP][NGP], V (&cX)[NGP], V (&cY)[NGP],
                                              V (&tU)[NGP][NGP], V (&tX)[NGP], V (&tY)[NGP], V& e1, V& e2) {
    V a1z = 0.f, a1x = 0.f, a1y = 0.f, a2 = 0.f;
    const float nb = T.nbw;
    // row / column sums of the weighted nu planes (x- and y-moments of nu at the y- / x-Gauss points)
    V LA[NGP], UA[NGP], LB[NGP], UB[NGP];
#pragma unroll
    for (int j = 0; j < NGP; ++j) {
        V la = 0.f, ua = 0.f, lb = 0.f, ub = 0.f;
#pragma unroll
        for (int i = 0; i < NGP; ++i) { la += LN[j][i]; ua += UN[j][i]; lb += LN[i][j]; ub += UN[i][j]; }
        LA[j] = la; UA[j] = ua; LB[j] = lb; UB[j] = ub;
    }
#pragma unroll
    for (int j = 0; j < NGP; ++j) {
#pragma unroll
        for (int i = 0; i < NGP; ++i) {
            const V dzu = UU[j][i] - LU[j][i];
            const V Qz = vfma(T.m[1], UN[j][i], T.m01 * LN[j][i]);          // sum_k w_k nu(i,j,k), in-plane weights inside
            const V qz = Qz * dzu;
            a1z = vfma(qz, dzu, a1z);
            V up = T.kap[2] * qz;                                            // cotangent of UU from the z-derivative term
            V lo = cU[j][i] - up;
            if constexpr (FGP) {
                V cs = 0.f, c1 = 0.f;
#pragma unroll
                for (int k = 0; k < NGP; ++k) {
                    const float wf = T.w[k] * T.w[j] * T.w[i] * fg[(k * NGP + j) * NGP + i];
                    cs += wf;
                    c1 = vfma(T.b[k][1], wf, c1);
                }
                a2 = vfma(cs, LU[j][i], a2);
                a2 = vfma(c1, dzu, a2);
                up = vfma(nb, c1, up);
                lo = vfma(nb, cs - c1, lo);
            } else if constexpr (HAS_F) {
                const V cs = vfma(T.m[1], UF[j][i], T.m01 * LF[j][i]);       // m0 FL + m1 (FU - FL)
                const V c1 = vfma(T.m[2], UF[j][i], T.m12 * LF[j][i]);       // m1 FL + m2 (FU - FL)
                a2 = vfma(cs, LU[j][i], a2);
                a2 = vfma(c1, dzu, a2);
                up = vfma(nb, c1, up);
                lo = vfma(nb, cs - c1, lo);
            }
            tU[j][i] = lo;
            cU[j][i] = up;
        }
    }
#pragma unroll
    for (int j = 0; j < NGP; ++j) {
        const V dX = UX[j] - LX[j], dA = UA[j] - LA[j];
        V sx = 0.f, tx = 0.f;
#pragma unroll
        for (int k = 0; k < NGP; ++k) {
            const V ux = vfma(T.b[k][1], dX, LX[j]);
            const V Qx = vfma(T.b[k][1], dA, LA[j]);
            const V qx = UW ? Qx * ux : (T.w[k] * Qx) * ux;
            a1x = vfma(qx, ux, a1x);
            sx += qx;
            tx = vfma(T.b[k][1], qx, tx);
        }
        const V up = T.kap[0] * tx;
        tX[j] = vfma(T.kap[0], sx, cX[j] - up);
        cX[j] = up;
    }
#pragma unroll
    for (int i = 0; i < NGP; ++i) {
        const V dY = UY[i] - LY[i], dB = UB[i] - LB[i];
        V sy = 0.f, ty = 0.f;
#pragma unroll
        for (int k = 0; k < NGP; ++k) {
            const V uy = vfma(T.b[k][1], dY, LY[i]);
            const V Qy = vfma(T.b[k][1], dB, LB[i]);
            const V qy = UW ? Qy * uy : (T.w[k] * Qy) * uy;
            a1y = vfma(qy, uy, a1y);
            sy += qy;
            ty = vfma(T.b[k][1], qy, ty);
        }
        const V up = T.kap[1] * ty;
        tY[i] = vfma(T.kap[1], sy, cY[i] - up);
        cY[i] = up;
    }
    e1 = vfma(T.hs2[2], a1z, vfma(T.hs2[0], a1x, T.hs2[1] * a1y));
    e2 = a2;
}

}  // namespace dn
