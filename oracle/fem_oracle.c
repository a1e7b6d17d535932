/* Plain-C float64 restatement of the reference's Poisson energy loss and its gradient -- TEST INFRASTRUCTURE ONLY.
 *
 * Independent second checker next to oracle/fem_oracle.py: direct loops over (sample, element, Gauss point, basis)
 * with dense nd tables formed as in DiffNet/DiffNetFEM.py:196-227 / :405-453 (tensor products of the 1-D bases and
 * the 2/h scales, here kept in double), the integrand of IBN_2D.py:123-131 / solve_in_object_3d.py:90-100
 *     L = 1/(B*nel) * sum_{b,e,g} gpw_g * jac * ( c * nu_g * |grad u|_g^2 - u_g * f_g ),
 * Dirichlet replacement u <- where(mask > 0.5, value, u) first (IBN_2D.py:119-121), and dL/du by the chain rule
 * (what autograd does for the reference).  No sum factorisation, no float32: it tells which of two float32
 * implementations is closer to the exact quadrature sum.  Pinned against oracle/fem_oracle.py (itself pinned
 * against the reference's golden vectors) in tests/test_oracle_c.py.  Nothing under diffnet_amd/ links this.
 *
 * nsd in {2,3}; nb = degree+1 (2..4); ng = ngp_1d (1..4); n[] nodes per axis (x,y,z); basis/dbasis: [ng][nb]
 * row-major (dbasis in reference coordinates); scale[d] = 2/h_d.  Fields are (B,1,nz,ny,nx) doubles; nu/f/mask may
 * be NULL.  Returns the loss; grad (same shape as u) is overwritten.
 */
#include <stdlib.h>
#include <string.h>

double dn_oracle_energy(int nsd, int nb, int ng, int batch, const int n[3], const double *basis, const double *dbasis,
                        const double *gpw, const double scale[3], const double *u_in, const double *nu, const double *f,
                        const double *mask, double mask_value, double c, double jac, double *grad) {
    const int deg = nb - 1;
    const int nx = n[0], ny = n[1], nz = nsd == 3 ? n[2] : 1;
    const int ex = (nx - 1) / deg, ey = (ny - 1) / deg, ez = nsd == 3 ? (nz - 1) / deg : 1;
    const int nbz = nsd == 3 ? nb : 1, ngz = nsd == 3 ? ng : 1;
    const long nps = (long)nx * ny * nz;
    const long nel = (long)ex * ey * ez;
    const int NB = nb * nb * nbz, NG = ng * ng * ngz;
    double *N = malloc(sizeof(double) * NG * NB), *Dx = malloc(sizeof(double) * NG * NB);
    double *Dy = malloc(sizeof(double) * NG * NB), *Dz = malloc(sizeof(double) * NG * NB), *W = malloc(sizeof(double) * NG);
    double *u = malloc(sizeof(double) * nps * batch);
    for (int kg = 0; kg < ngz; ++kg)
        for (int jg = 0; jg < ng; ++jg)
            for (int ig = 0; ig < ng; ++ig) {
                const int g = (kg * ng + jg) * ng + ig;
                W[g] = gpw[ig] * gpw[jg] * (nsd == 3 ? gpw[kg] : 1.0);
                for (int kb = 0; kb < nbz; ++kb)
                    for (int jb = 0; jb < nb; ++jb)
                        for (int ib = 0; ib < nb; ++ib) {
                            const int a = (kb * nb + jb) * nb + ib;
                            const double bi = basis[ig * nb + ib], bj = basis[jg * nb + jb];
                            const double bk = nsd == 3 ? basis[kg * nb + kb] : 1.0;
                            const double di = dbasis[ig * nb + ib], dj = dbasis[jg * nb + jb];
                            const double dk = nsd == 3 ? dbasis[kg * nb + kb] : 0.0;
                            N[g * NB + a] = bi * bj * bk;
                            Dx[g * NB + a] = di * bj * bk * scale[0];
                            Dy[g * NB + a] = bi * dj * bk * scale[1];
                            Dz[g * NB + a] = nsd == 3 ? bi * bj * dk * scale[2] : 0.0;
                        }
            }
    for (long i = 0; i < nps * batch; ++i) {
        u[i] = (mask && mask[i] > 0.5) ? mask_value : u_in[i];
        grad[i] = 0.0;
    }
    const double norm = 1.0 / ((double)batch * (double)nel);
    double total = 0.0;
    double *lu = malloc(sizeof(double) * NB), *ln = malloc(sizeof(double) * NB), *lf = malloc(sizeof(double) * NB);
    double *lg = malloc(sizeof(double) * NB);
    for (int b = 0; b < batch; ++b)
        for (int kz = 0; kz < ez; ++kz)
            for (int ky = 0; ky < ey; ++ky)
                for (int kx = 0; kx < ex; ++kx) {
                    for (int kb = 0; kb < nbz; ++kb)
                        for (int jb = 0; jb < nb; ++jb)
                            for (int ib = 0; ib < nb; ++ib) {
                                const long nd = (long)b * nps + ((long)(kz * deg + kb) * ny + (ky * deg + jb)) * nx + kx * deg + ib;
                                const int a = (kb * nb + jb) * nb + ib;
                                lu[a] = u[nd];
                                ln[a] = nu ? nu[nd] : 1.0;
                                lf[a] = f ? f[nd] : 0.0;
                                lg[a] = 0.0;
                            }
                    for (int g = 0; g < NG; ++g) {
                        double ug = 0, ux = 0, uy = 0, uz = 0, ng_ = 0, fg = 0;
                        for (int a = 0; a < NB; ++a) {
                            ug += N[g * NB + a] * lu[a];
                            ux += Dx[g * NB + a] * lu[a];
                            uy += Dy[g * NB + a] * lu[a];
                            uz += Dz[g * NB + a] * lu[a];
                            ng_ += N[g * NB + a] * ln[a];
                            fg += N[g * NB + a] * lf[a];
                        }
                        const double w = W[g] * jac;
                        total += w * (c * ng_ * (ux * ux + uy * uy + uz * uz) - ug * fg);
                        for (int a = 0; a < NB; ++a)
                            lg[a] += w * (2.0 * c * ng_ * (ux * Dx[g * NB + a] + uy * Dy[g * NB + a] + uz * Dz[g * NB + a]) - fg * N[g * NB + a]);
                    }
                    for (int kb = 0; kb < nbz; ++kb)
                        for (int jb = 0; jb < nb; ++jb)
                            for (int ib = 0; ib < nb; ++ib) {
                                const long nd = (long)b * nps + ((long)(kz * deg + kb) * ny + (ky * deg + jb)) * nx + kx * deg + ib;
                                grad[nd] += lg[(kb * nb + jb) * nb + ib] * norm;
                            }
                }
    if (mask)
        for (long i = 0; i < nps * batch; ++i)
            if (mask[i] > 0.5) grad[i] = 0.0;
    free(N); free(Dx); free(Dy); free(Dz); free(W); free(u); free(lu); free(ln); free(lf); free(lg);
    return total * norm;
}
