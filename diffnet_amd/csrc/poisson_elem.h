// Per-element arithmetic of the fused Poisson operator (energy density, its gradient / the weak-form
// residual) for one structured Q_P element, evaluated entirely in registers.
//
// What the reference does per element and Gauss point with dense nd tables
// (DiffNet/DiffNetFEM.py:196-227, 405-453 tables; IBN_2D.py:123-131, 12_klsum.py:96-122 use) is done
// here by sum factorisation: the 1-D basis tables are applied axis by axis.  For Q1 the 1-D
// basis is a lerp, phi(xi) = (1-b, b) with b = (1+xi)/2 and phi' = (-1/2, +1/2), so every 1-D
// contraction is one subtraction plus one FMA per Gauss point and the derivative is the
// difference itself; that specialisation is what keeps the kernel under the HBM roofline
// instead of the VALU roofline.
#pragma once
#include "dn_common.h"

namespace dn {

// Kernel-argument resident (SGPR) tables: everything wave-uniform the element code needs.
struct ElemTab {
    float b[4][4];       // b[ig][ib]  = phi_ib(xi_ig)
    float dx[4][4];      // dbasis[ig][ib] * 2/hx   (generic-degree path)
    float dy[4][4];
    float dz[4][4];
    float w[4];          // 1-D Gauss weights (wscale folded into w of the x axis: see wx)
    float wx[4];         // w[ig] * wscale
    float hs[3];         // Q1 path: 0.5 * 2/h_d = 1/h_d
    float alpha, beta, c;
};

enum { F_NONE = 0, F_NODAL = 1, F_GP = 2 };

// ---------------------------------------------------------------------------------------------
// 2-D element.  u/nu/f: nodal values [jb][ib]; fg: forcing at Gauss points [jg*NGP+ig] (F_GP).
// g: contributions to the (P+1)^2 nodes;  returns the element's energy.
// ---------------------------------------------------------------------------------------------
template <int P, int NGP>
__device__ __forceinline__ float elem2d(const ElemTab& T, const bool has_nu, const int fmode,
                                        const float (&u)[P + 1][P + 1], const float (&nu)[P + 1][P + 1],
                                        const float (&f)[P + 1][P + 1], const float* fg, float (&g)[P + 1][P + 1]) {
    constexpr int NB = P + 1;
    float e = 0.f;
    if constexpr (P == 1) {
        float b1[NGP];
#pragma unroll
        for (int i = 0; i < NGP; ++i) b1[i] = T.b[i][1];
        const float dx0 = u[0][1] - u[0][0], dx1 = u[1][1] - u[1][0], ddx = dx1 - dx0;
        float tv0[NGP], dyv[NGP], uy[NGP], uy2[NGP], ux[NGP], ux2[NGP];
        float tn0[NGP], dyn[NGP], tf0[NGP], dyf[NGP];
#pragma unroll
        for (int i = 0; i < NGP; ++i) {
            tv0[i] = fmaf(b1[i], dx0, u[0][0]);
            dyv[i] = fmaf(b1[i], dx1, u[1][0]) - tv0[i];
            uy[i] = T.hs[1] * dyv[i];
            uy2[i] = uy[i] * uy[i];
            ux[i] = T.hs[0] * fmaf(b1[i], ddx, dx0);   // index i is jg here
            ux2[i] = ux[i] * ux[i];
        }
        if (has_nu) {
            const float n0 = nu[0][1] - nu[0][0], n1 = nu[1][1] - nu[1][0];
#pragma unroll
            for (int i = 0; i < NGP; ++i) {
                tn0[i] = fmaf(b1[i], n0, nu[0][0]);
                dyn[i] = fmaf(b1[i], n1, nu[1][0]) - tn0[i];
            }
        }
        if (fmode == F_NODAL) {
            const float f0 = f[0][1] - f[0][0], f1 = f[1][1] - f[1][0];
#pragma unroll
            for (int i = 0; i < NGP; ++i) {
                tf0[i] = fmaf(b1[i], f0, f[0][0]);
                dyf[i] = fmaf(b1[i], f1, f[1][0]) - tf0[i];
            }
        }
        float Qx[NGP], Qy[NGP], c1[NGP], cs[NGP];
#pragma unroll
        for (int i = 0; i < NGP; ++i) Qx[i] = Qy[i] = c1[i] = cs[i] = 0.f;
#pragma unroll
        for (int jg = 0; jg < NGP; ++jg) {
#pragma unroll
            for (int ig = 0; ig < NGP; ++ig) {
                const float W = T.w[jg] * T.wx[ig];
                const float val = fmaf(b1[jg], dyv[ig], tv0[ig]);
                const float nuv = has_nu ? fmaf(b1[jg], dyn[ig], tn0[ig]) : 1.f;
                float fv = 0.f;
                if (fmode == F_NODAL) fv = fmaf(b1[jg], dyf[ig], tf0[ig]);
                else if (fmode == F_GP) fv = fg[jg * NGP + ig];
                const float Wn = W * nuv;
                const float Wf = W * fv;
                e = fmaf(T.c * Wn, ux2[jg] + uy2[ig], e);
                e = fmaf(-Wf, val, e);
                Qx[jg] += Wn;
                Qy[ig] += Wn;
                const float qv = -T.beta * Wf;
                cs[ig] += qv;
                c1[ig] = fmaf(b1[jg], qv, c1[ig]);
            }
        }
        // transposes
        float cdx1 = 0.f, cdxs = 0.f;
#pragma unroll
        for (int jg = 0; jg < NGP; ++jg) {
            const float cx = (T.alpha * T.hs[0]) * (Qx[jg] * ux[jg]);
            cdx1 = fmaf(b1[jg], cx, cdx1);
            cdxs += cx;
        }
        const float cdx0 = cdxs - cdx1;
        float s0 = 0.f, s1 = 0.f, t0 = 0.f, t1 = 0.f;   // row sums / b-weighted sums of cot(tv[jb][ig])
#pragma unroll
        for (int ig = 0; ig < NGP; ++ig) {
            const float cyd = (T.alpha * T.hs[1]) * (Qy[ig] * uy[ig]);
            const float ct1 = c1[ig] + cyd;      // cot of tv[1][ig]
            const float ct0 = cs[ig] - ct1;      // cot of tv[0][ig]
            s0 += ct0; s1 += ct1;
            t0 = fmaf(b1[ig], ct0, t0);
            t1 = fmaf(b1[ig], ct1, t1);
        }
        g[0][1] = t0 + cdx0; g[0][0] = s0 - g[0][1];
        g[1][1] = t1 + cdx1; g[1][0] = s1 - g[1][1];
    } else {
        // generic degree: table driven sum factorisation
        float tv[NB][NGP], td[NB][NGP], tn[NB][NGP], tf[NB][NGP];
#pragma unroll
        for (int jb = 0; jb < NB; ++jb)
#pragma unroll
            for (int ig = 0; ig < NGP; ++ig) {
                float a = 0.f, d = 0.f, n = 0.f, q = 0.f;
#pragma unroll
                for (int ib = 0; ib < NB; ++ib) {
                    a = fmaf(T.b[ig][ib], u[jb][ib], a);
                    d = fmaf(T.dx[ig][ib], u[jb][ib], d);
                    if (has_nu) n = fmaf(T.b[ig][ib], nu[jb][ib], n);
                    if (fmode == F_NODAL) q = fmaf(T.b[ig][ib], f[jb][ib], q);
                }
                tv[jb][ig] = a; td[jb][ig] = d; tn[jb][ig] = n; tf[jb][ig] = q;
            }
        float rv[NB][NGP], rd[NB][NGP];
#pragma unroll
        for (int jb = 0; jb < NB; ++jb)
#pragma unroll
            for (int ig = 0; ig < NGP; ++ig) rv[jb][ig] = rd[jb][ig] = 0.f;
#pragma unroll
        for (int jg = 0; jg < NGP; ++jg) {
#pragma unroll
            for (int ig = 0; ig < NGP; ++ig) {
                float val = 0.f, ux = 0.f, uy = 0.f, nuv = 0.f, fv = 0.f;
#pragma unroll
                for (int jb = 0; jb < NB; ++jb) {
                    val = fmaf(T.b[jg][jb], tv[jb][ig], val);
                    ux = fmaf(T.b[jg][jb], td[jb][ig], ux);
                    uy = fmaf(T.dy[jg][jb], tv[jb][ig], uy);
                    if (has_nu) nuv = fmaf(T.b[jg][jb], tn[jb][ig], nuv);
                    if (fmode == F_NODAL) fv = fmaf(T.b[jg][jb], tf[jb][ig], fv);
                }
                if (!has_nu) nuv = 1.f;
                if (fmode == F_GP) fv = fg[jg * NGP + ig];
                const float W = T.w[jg] * T.wx[ig];
                const float Wn = W * nuv, Wf = W * fv;
                e = fmaf(T.c * Wn, ux * ux + uy * uy, e);
                e = fmaf(-Wf, val, e);
                const float qx = T.alpha * Wn * ux, qy = T.alpha * Wn * uy, qv = -T.beta * Wf;
#pragma unroll
                for (int jb = 0; jb < NB; ++jb) {
                    rv[jb][ig] = fmaf(T.b[jg][jb], qv, rv[jb][ig]);
                    rv[jb][ig] = fmaf(T.dy[jg][jb], qy, rv[jb][ig]);
                    rd[jb][ig] = fmaf(T.b[jg][jb], qx, rd[jb][ig]);
                }
            }
        }
#pragma unroll
        for (int jb = 0; jb < NB; ++jb)
#pragma unroll
            for (int ib = 0; ib < NB; ++ib) {
                float a = 0.f;
#pragma unroll
                for (int ig = 0; ig < NGP; ++ig) {
                    a = fmaf(T.b[ig][ib], rv[jb][ig], a);
                    a = fmaf(T.dx[ig][ib], rd[jb][ig], a);
                }
                g[jb][ib] = a;
            }
    }
    return e;
}

// ---------------------------------------------------------------------------------------------
// 3-D Q1 element (trilinear hexahedron), lerp form.  u/nu/f: [kb][jb][ib]; fg: [(kg*NGP+jg)*NGP+ig].
// ---------------------------------------------------------------------------------------------
template <int NGP>
__device__ __forceinline__ float elem3d_q1(const ElemTab& T, const bool has_nu, const int fmode,
                                           const float (&u)[2][2][2], const float (&nu)[2][2][2],
                                           const float (&f)[2][2][2], const float* fg, float (&g)[2][2][2]) {
    float b1[NGP];
#pragma unroll
    for (int i = 0; i < NGP; ++i) b1[i] = T.b[i][1];
    float e = 0.f;
    // ---- forward: x stage, y stage
    float dx[2][2];            // u differences along x            [kb][jb]
    float tv0[2][NGP];         // tv[kb][0][ig]
    float dyv[2][NGP];         // tv[kb][1][ig] - tv[kb][0][ig]
    float vx0[2], ddx[2];      // vx[kb][jg] = fma(b1[jg], ddx[kb], vx0[kb])
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        dx[kb][0] = u[kb][0][1] - u[kb][0][0];
        dx[kb][1] = u[kb][1][1] - u[kb][1][0];
        vx0[kb] = dx[kb][0];
        ddx[kb] = dx[kb][1] - dx[kb][0];
#pragma unroll
        for (int ig = 0; ig < NGP; ++ig) {
            tv0[kb][ig] = fmaf(b1[ig], dx[kb][0], u[kb][0][0]);
            dyv[kb][ig] = fmaf(b1[ig], dx[kb][1], u[kb][1][0]) - tv0[kb][ig];
        }
    }
    float tn0[2][NGP], dyn[2][NGP], tf0[2][NGP], dyf[2][NGP];
    if (has_nu) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const float n0 = nu[kb][0][1] - nu[kb][0][0], n1 = nu[kb][1][1] - nu[kb][1][0];
#pragma unroll
            for (int ig = 0; ig < NGP; ++ig) {
                tn0[kb][ig] = fmaf(b1[ig], n0, nu[kb][0][0]);
                dyn[kb][ig] = fmaf(b1[ig], n1, nu[kb][1][0]) - tn0[kb][ig];
            }
        }
    }
    if (fmode == F_NODAL) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const float f0 = f[kb][0][1] - f[kb][0][0], f1 = f[kb][1][1] - f[kb][1][0];
#pragma unroll
            for (int ig = 0; ig < NGP; ++ig) {
                tf0[kb][ig] = fmaf(b1[ig], f0, f[kb][0][0]);
                dyf[kb][ig] = fmaf(b1[ig], f1, f[kb][1][0]) - tf0[kb][ig];
            }
        }
    }
    // gradient components depend on two of the three Gauss indices only
    float ux[NGP][NGP], uy[NGP][NGP];   // ux[kg][jg], uy[kg][ig]
#pragma unroll
    for (int jg = 0; jg < NGP; ++jg) {
        const float a0 = fmaf(b1[jg], ddx[0], vx0[0]), a1 = fmaf(b1[jg], ddx[1], vx0[1]);
#pragma unroll
        for (int kg = 0; kg < NGP; ++kg) ux[kg][jg] = T.hs[0] * fmaf(b1[kg], a1 - a0, a0);
    }
#pragma unroll
    for (int ig = 0; ig < NGP; ++ig) {
        const float d = dyv[1][ig] - dyv[0][ig];
#pragma unroll
        for (int kg = 0; kg < NGP; ++kg) uy[kg][ig] = T.hs[1] * fmaf(b1[kg], d, dyv[0][ig]);
    }
    float Qx[NGP][NGP], Qy[NGP][NGP];
#pragma unroll
    for (int a = 0; a < NGP; ++a)
#pragma unroll
        for (int b = 0; b < NGP; ++b) Qx[a][b] = Qy[a][b] = 0.f;
    // cotangents of vv[kb][jg][ig] accumulated as (sum over kg, b1-weighted sum over kg)
    float cT0[2][NGP], cD[2][NGP];   // after the jg loop: cot of tv[kb][0][ig] partial sums and cot of dyv[kb][ig]
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ig = 0; ig < NGP; ++ig) cT0[kb][ig] = cD[kb][ig] = 0.f;
#pragma unroll
    for (int jg = 0; jg < NGP; ++jg) {
#pragma unroll
        for (int ig = 0; ig < NGP; ++ig) {
            const float vv0 = fmaf(b1[jg], dyv[0][ig], tv0[0][ig]);
            const float vv1 = fmaf(b1[jg], dyv[1][ig], tv0[1][ig]);
            const float dz = vv1 - vv0;
            const float uz = T.hs[2] * dz, uz2 = uz * uz;
            float nn0 = 1.f, dn = 0.f, ff0 = 0.f, df = 0.f;
            if (has_nu) {
                nn0 = fmaf(b1[jg], dyn[0][ig], tn0[0][ig]);
                dn = fmaf(b1[jg], dyn[1][ig], tn0[1][ig]) - nn0;
            }
            if (fmode == F_NODAL) {
                ff0 = fmaf(b1[jg], dyf[0][ig], tf0[0][ig]);
                df = fmaf(b1[jg], dyf[1][ig], tf0[1][ig]) - ff0;
            }
            float Qz = 0.f, cs = 0.f, c1 = 0.f;
#pragma unroll
            for (int kg = 0; kg < NGP; ++kg) {
                const float W = T.w[kg] * T.w[jg] * T.wx[ig];
                const float val = fmaf(b1[kg], dz, vv0);
                const float nuv = has_nu ? fmaf(b1[kg], dn, nn0) : 1.f;
                float fv = 0.f;
                if (fmode == F_NODAL) fv = fmaf(b1[kg], df, ff0);
                else if (fmode == F_GP) fv = fg[(kg * NGP + jg) * NGP + ig];
                const float Wn = W * nuv, Wf = W * fv;
                e = fmaf(T.c * Wn, ux[kg][jg] * ux[kg][jg] + uy[kg][ig] * uy[kg][ig] + uz2, e);
                e = fmaf(-Wf, val, e);
                Qx[kg][jg] += Wn;
                Qy[kg][ig] += Wn;
                Qz += Wn;
                const float qv = -T.beta * Wf;
                cs += qv;
                c1 = fmaf(b1[kg], qv, c1);
            }
            // cot of vv1 / vv0 (value path + z-derivative path)
            const float cz = (T.alpha * T.hs[2]) * (Qz * uz);
            const float cv1 = c1 + cz, cv0 = cs - cv1;
            // y-stage transpose: vv[kb] = fma(b1[jg], dyv[kb][ig], tv0[kb][ig])
            cT0[0][ig] += cv0; cD[0][ig] = fmaf(b1[jg], cv0, cD[0][ig]);
            cT0[1][ig] += cv1; cD[1][ig] = fmaf(b1[jg], cv1, cD[1][ig]);
        }
    }
    // y-derivative path: uy[kg][ig] = hs1 * fma(b1[kg], dyv1-dyv0, dyv0)
#pragma unroll
    for (int ig = 0; ig < NGP; ++ig) {
        float s = 0.f, t = 0.f;
#pragma unroll
        for (int kg = 0; kg < NGP; ++kg) {
            const float cy = (T.alpha * T.hs[1]) * (Qy[kg][ig] * uy[kg][ig]);
            s += cy; t = fmaf(b1[kg], cy, t);
        }
        cD[1][ig] += t;
        cD[0][ig] += s - t;
    }
    // x-derivative path: ux[kg][jg] = hs0 * fma(b1[kg], a1-a0, a0), a_kb = fma(b1[jg], ddx[kb], vx0[kb])
    float cX0[2] = {0.f, 0.f}, cDD[2] = {0.f, 0.f};   // cot of vx0[kb] (= dx[kb][0]) and of ddx[kb]
#pragma unroll
    for (int jg = 0; jg < NGP; ++jg) {
        float s = 0.f, t = 0.f;
#pragma unroll
        for (int kg = 0; kg < NGP; ++kg) {
            const float cx = (T.alpha * T.hs[0]) * (Qx[kg][jg] * ux[kg][jg]);
            s += cx; t = fmaf(b1[kg], cx, t);
        }
        const float ca1 = t, ca0 = s - t;
        cX0[0] += ca0; cDD[0] = fmaf(b1[jg], ca0, cDD[0]);
        cX0[1] += ca1; cDD[1] = fmaf(b1[jg], ca1, cDD[1]);
    }
    // x-stage transpose
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        // cot of tv[kb][1][ig] = cD[kb][ig]; cot of tv[kb][0][ig] = cT0[kb][ig] - cD[kb][ig]
        float s0 = 0.f, t0 = 0.f, s1 = 0.f, t1 = 0.f;
#pragma unroll
        for (int ig = 0; ig < NGP; ++ig) {
            const float c1_ = cD[kb][ig], c0_ = cT0[kb][ig] - c1_;
            s0 += c0_; t0 = fmaf(b1[ig], c0_, t0);
            s1 += c1_; t1 = fmaf(b1[ig], c1_, t1);
        }
        // cot of dx[kb][1] = cDD[kb]; cot of dx[kb][0] = cX0[kb] - cDD[kb]
        const float cdx1 = cDD[kb], cdx0 = cX0[kb] - cDD[kb];
        g[kb][0][1] = t0 + cdx0; g[kb][0][0] = s0 - g[kb][0][1];
        g[kb][1][1] = t1 + cdx1; g[kb][1][0] = s1 - g[kb][1][1];
    }
    return e;
}

}  // namespace dn
