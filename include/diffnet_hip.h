/*
 * diffnet_hip.h -- C ABI of libdiffnet_hip.so, the MI355X (gfx950) implementation of the
 * DiffNet FEM Gauss-quadrature hot path.
 *
 * Plain pointers and sizes only: no torch types cross this boundary.  Every pointer is a
 * DEVICE pointer (HBM resident) unless stated otherwise; tensors are contiguous fp32 in the
 * reference's layouts: nodal fields (B,1,Ny,Nx) / (B,1,Nz,Ny,Nx), Gauss-point fields
 * (B,G,nelY,nelX) / (B,G,nelZ,nelY,nelX) with G = ngp_1d^nsd and the Gauss-point id
 * g = (kg*ngp + jg)*ngp + ig (x fastest), local basis id a = (kb*nbf + jb)*nbf + ib.
 * `stream` is a hipStream_t passed as void* (NULL = default stream).  All entry points are
 * asynchronous on `stream`, allocate nothing, never synchronise, and are graph-capturable.
 * Return value: 0 on success, a positive hipError_t from the launch, or a negative DN_E_* code
 * for an argument the kernels do not support (nothing is launched in that case).
 *
 * Each entry point names the reference interface it replaces (paths relative to the
 * reference repository root).
 */
#ifndef DIFFNET_HIP_H
#define DIFFNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DN_ABI_VERSION 9

#define DN_E_BADARG (-1)    /* null pointer / non-positive size / unsupported combination   */
#define DN_E_UNSUPPORTED (-2) /* (nsd, degree, ngp) outside the compiled instantiations      */
#define DN_E_WORKSPACE (-3)  /* workspace too small                                           */
#define DN_E_HANDOVER (-4)   /* dn_workspace_status: a launch that used the workspace ran a bounded LDS hand-over poll to its
                                limit (chained-strip plans of dn_poisson_apply / dn_fsdt_apply); its results are NaN        */

/* Process-wide tuning / A-B switches (launch geometry overrides, kernel-variant selection).  They are NOT part of the
 * numerical contract: every setting yields the same results to rounding.  The table is initialised ONCE when the library
 * is loaded, from the environment variables DN_<KEY>; afterwards the environment is never read again (no getenv on the
 * launch path) and the only way to change a switch is this call.  Keys (the table kKeys in csrc/dn_api.hip):
 *   "PLAN2D" ("T,E,R[,W]": threads per strip, elements per thread, node rows per strip, W >= 2 = chain strips per workgroup
 *   where the closed-form Q1 kernel can), "PLAN3D" ("TX,TY,E,R"), "PLAN_FSDT" ("T,R"),
 *   "Q1_RULE_KERNEL" (non-empty: per-Gauss-point 2-D Q1 kernels instead of the closed form),
 *   "GPE_GATHER" (non-empty: per-node gather adjoint of gauss_pt_eval), "GPE_TILED" (non-empty: tiled LDS adjoint instead of the
 *   marching 3-D Q1 adjoint), "Q1_3D_T16" (non-empty: the 3-D Q1 kernel form in which every thread loads its own nodes),
 *   "Q1_3D_E1SUM" (non-empty: 3-D stiffness energy summed Gauss point by Gauss point), "Q1_3D_E1" (non-empty: the 3-D node-owner kernel with one element per thread also where the two-element form applies),
 *   "FSDT_GENERIC" (non-empty: the table-driven
 *   FSDT element also for Q2 with the symmetric 3-point rule, whose middle point otherwise runs a specialised form),
 *   "PLAN_FSDT" also takes "64,R,W": W one-wave sub-strips chained per workgroup,
 *   "HANDOVER_SPIN_LIMIT" (integer: bound of the chained plans' LDS hand-over polls instead of the built-in 2^20 -- a test hook for the
 *   error path of dn_workspace_status; a small value makes healthy launches fail), "CONV2D_V1" (non-empty: the round-2 im2col forms of
 *   dn_conv2d_k4s2_down / _up / _wrw also where the raw-row-tile forms of round 4 apply), "CONV_WRW_WGS" (integer: workgroups a
 *   dn_conv2d_k4s2_wrw launch aims at when it splits K; default 1024), "Q1_3D_N2" (non-empty: the round-3 per-Gauss-point form of the 3-D Q1
 *   two-element kernel also where the closed-form-in-z kernel of round 4, csrc/poisson3d_q1_cf.hip, applies),
 *   "FSDT_FORM" ("elem": dn_fsdt_apply runs the element form of rounds 1-3, csrc/fsdt.hip, instead of the assembled-stencil form of round 4,
 *   csrc/fsdt_st.hip, which is the default for Q1 and Q2 meshes; "stencil": the stencil form also for Q3 meshes, where the element form is the default).  value NULL or "" clears the switch.
 * Returns 0, or DN_E_BADARG for an unknown key / over-long value.  Not thread-safe against concurrent launches.
 * No reference counterpart (the reference has no tuning surface). */
int dn_config_set(const char *key, const char *value);
/* Current value of a switch ("" when unset), NULL for an unknown key. */
const char *dn_config_get(const char *key);

/* Geometry + 1-D reference-element tables of one structured mesh.
 * Mirrors what DiffNetFEM.__init__ derives (DiffNet/DiffNetFEM.py:25-126):
 *   basis[ig][ib]  = phi_ib(xi_ig)          1-D Lagrange basis at the 1-D Gauss points
 *   dbasis[ig][ib] = phi'_ib(xi_ig)         derivative in reference coordinates
 *   gpw[ig]        1-D Gauss weights,  scale[d] = 2/h_d  (d = x,y,z)
 * The nd tables of the reference (N_gp, dN_x_gp, ...) are the tensor products of these; the
 * fused kernels apply them axis by axis (sum factorisation) instead of materialising them. */
typedef struct dn_mesh {
    int32_t nsd;          /* 2 or 3                                                     */
    int32_t degree;       /* fem_basis_deg: 1, 2 or 3                                   */
    int32_t ngp;          /* ngp_1d: 1..4                                               */
    int32_t batch;        /* B                                                          */
    int32_t nx, ny, nz;   /* nodes per axis (nz = 1 when nsd == 2)                      */
    float scale[3];       /* 2/hx, 2/hy, 2/hz                                           */
    float gpw[4];
    float basis[4][4];
    float dbasis[4][4];
} dn_mesh;

/* One Dirichlet condition: u <- where(mask > 0.5, value, u) with value = field[node] when
 * `field` is non-null (broadcast over the batch when field_batched == 0) else the constant.
 * Mirrors the torch.where lines of the loss bodies, e.g. IBN/poisson-2d/parametric/IBN_2D.py:119-121,
 * examples/poisson/single_instance/e8_2d_poisson_mms.py:120.
 * Mask formats (`mask_kind`):
 *   DN_MASK_F32  the reference's fp32 image, compared with 0.5            (4 B per node)
 *   DN_MASK_U8   uint8 / bool, non-zero = set                             (1 B per node)
 *   DN_MASK_BITS one bit per node: node x of a node row is bit (x & 31) of 32-bit word (x >> 5) of that row, rows
 *                `row_words` words apart (>= ceil(nx / 32)), rows of a sample contiguous, samples ny*[nz*]row_words words
 *                apart -- what dn_pack_mask_bits() writes                 (1/8 B per node)
 *   DN_MASK_BOX  no array at all: the condition holds on the domain faces named in `box_faces` (IBN_2D.py:69-73 builds exactly
 *                this mask as an image, DiffNet/datasets/single_instances/rectangles.py:16,232-233 likewise); `mask` is ignored
 * BITS and BOX take a constant `value` (field must be NULL) and are implemented by the 2-D Q1 kernels with nodal / absent
 * forcing (the IBN_2D / bench path); every other kernel returns DN_E_UNSUPPORTED for them -- expand with dn_unpack_mask_bits(). */
enum { DN_MASK_F32 = 0, DN_MASK_U8 = 1, DN_MASK_BITS = 2, DN_MASK_BOX = 3 };
enum { DN_FACE_XLO = 1, DN_FACE_XHI = 2, DN_FACE_YLO = 4, DN_FACE_YHI = 8, DN_FACE_ZLO = 16, DN_FACE_ZHI = 32 };
typedef struct dn_dirichlet {
    const void *mask;     /* (B,1,*N) or (1,1,*N); NULL = condition absent (DN_MASK_BOX: ignored) */
    const float *field;   /* optional Dirichlet values                                  */
    float value;
    int32_t mask_kind;    /* DN_MASK_*  (0 / 1 keep their round-1 meaning "mask_is_u8") */
    int32_t mask_batched; /* 1: (B,...), 0: one mask shared by the whole batch          */
    int32_t field_batched;
    int32_t box_faces;    /* DN_MASK_BOX: bit set of DN_FACE_*                          */
    int32_t row_words;    /* DN_MASK_BITS: 32-bit words per node row                    */
} dn_dirichlet;

/* uint8 / fp32 mask image -> DN_MASK_BITS layout and back (one launch each; rows = B*ny[*nz] node rows of nx nodes).
 * The reference keeps masks as fp32 images in its datasets (DiffNet/datasets/parametric/images.py:30); packing them once when
 * the dataset is placed in HBM turns a 4 (or 1) B/node stream of the loss into 1/8 B/node. */
int dn_pack_mask_bits(const void *mask, int32_t mask_kind, int64_t rows, int32_t nx, int32_t row_words, uint32_t *bits, void *stream);
int dn_unpack_mask_bits(const uint32_t *bits, int64_t rows, int32_t nx, int32_t row_words, uint8_t *mask_u8, void *stream);

/* Arguments of the fused Poisson operator
 *     out_a = zero_on_dirichlet( sum_e sum_g W_g ( alpha * nu_g * gradN_a . grad u_g  -  beta * N_a * f_g ) )
 *     energy = sum_{b,e,g} W_g ( c * nu_g * |grad u_g|^2 - u_g * f_g ),      W_g = gpw_g * wscale
 * evaluated in ONE pass over the nodal fields with element->node assembly in gather form
 * (deterministic, no atomics).  It replaces, per call:
 *   - energy loss + its gradient  (alpha = 2c, beta = 1):  IBN_2D.py:116-134, IBN_3D.py:114-136,
 *     solve_in_object_3d.py:75-102, 12_klsum.py:53-78, e8_2d_poisson_mms.py:152-180 -- i.e. 5-6
 *     gauss_pt_eval calls (DiffNet/DiffNetFEM.py:7-18) + ~8 elementwise ops + autograd backward;
 *   - weak-form residual + Q1_*_vector_assembly + BC mask + sum(R^2) (alpha = beta = 1):
 *     12_klsum.py:80-132, e8_2d_poisson_mms.py:92-150, e8_3d_poisson_mms.py:89-139, tests/test.py:43-79;
 *   - the backward of the residual loss: the operator is symmetric, so grad_u = apply(v = 2R, f absent). */
typedef struct dn_poisson_args {
    const float *u;        /* (B,1,*N) nodal field                                      */
    const float *nu;       /* (B,1,*N) or NULL (nu == 1)                                */
    const float *f;        /* (B,1,*N) nodal forcing or NULL                            */
    const float *f_gp;     /* (Bf,G,*nel) forcing at Gauss points or NULL (exclusive with f) */
    int32_t nu_batched;    /* 0: one (1,1,*N) field for the whole batch                 */
    int32_t f_batched;     /* same for f / f_gp                                         */
    dn_dirichlet bc[2];    /* applied in order                                          */
    float alpha, beta, c, wscale;
    float out_scale;       /* out is multiplied by this (e.g. 1/(B*nel) of torch.mean)  */
    float *out;            /* (B,1,*N) or NULL                                          */
    double *energy;        /* device scalar: sum (unscaled) or NULL                     */
    double *sumsq;         /* device scalar: sum over nodes of (out/out_scale)^2 or NULL */
    void *workspace;       /* dn_poisson_workspace_bytes() bytes, needed when energy or sumsq is set. Must be
                              zero-filled ONCE before its first use; every call leaves it ready for the next
                              (self-resetting arrival counter). One workspace per concurrently used stream. */
    int64_t workspace_bytes;
    float *energy_f32;     /* optional device scalar: (float)(energy * energy_scale), e.g. the mean loss, written by the
                              same launch (saves the caller two elementwise kernels per evaluation); needs workspace  */
    double energy_scale;
    /* Split evaluation (no reference counterpart; the slab-parallel path, diffnet_amd/slab.py): the marched axis (y in 2-D, z in 3-D) is
     * cut into strips by the launch plan.  strip_select 1 launches only the first and the last strip -- the node layers next to the two
     * faces across the marched axis are complete after it, so their exchange with the neighbouring ranks can start --, 2 launches all the
     * other strips; 0 (default) all of them.  The two launches write disjoint parts of `out`.  accumulate_sums != 0: the final scalars
     * are added to what `energy` / `sumsq` (both required then) already hold and energy_f32 is formed from the running energy: launch 1
     * with accumulate_sums = 0, then launch 2 with 1, on one stream, gives the sums of the whole mesh in a fixed order. */
    int32_t strip_select, accumulate_sums;
    /* defer_sums != 0: the launch leaves its per-workgroup partial sums in the workspace and writes none of energy / sumsq / energy_f32;
     * dn_poisson_finish_sums(mesh, args, stream) adds them up (one small kernel, fixed order) on ANY stream ordered after the launch --
     * e.g. a side stream, under the next launch: the in-kernel final reduction is a ~3 us serial tail at the end of a ~56 us launch
     * (its last workgroup waits for every other one), the deferred one overlaps the next evaluation.  The workspace must not be handed
     * to another launch before the finish kernel has run (one workspace per evaluation in flight). */
    int32_t defer_sums;
    /* fold_prev != NULL (round 4): the arguments of an EARLIER dn_poisson_apply call on the same mesh that ran with defer_sums (its
     * partial sums still lie in ITS workspace).  This launch's first workgroup adds them up and writes that call's energy / sumsq /
     * energy_f32 -- what dn_poisson_finish_sums would do -- before it starts its own strip: the final reduction of evaluation k leaves
     * the critical path (no tail at the end of launch k, no extra kernel, no event) and rides at the start of launch k + 1, where one
     * of ~2000 workgroups doing 1-2 us of extra work delays nothing.  A loop therefore gets the loss of step k when launch k + 1 has
     * run (a training loop only logs it); the last evaluation is closed with dn_poisson_finish_sums.  Kernels without the path (anything
     * but the 2-D closed-form Q1 kernel and the 3-D Q1 two-element node-owner kernel) return DN_E_UNSUPPORTED.  Read at call time only. */
    const struct dn_poisson_args *fold_prev;
    /* f_is_load != 0 (round 4): `f` does not hold nodal forcing values but the ASSEMBLED load vector b_a = sum_e sum_g w_g N_a(g) f_g
     * (the rule's own weights, no wscale / beta: what this operator returns as -out for u = 0, alpha = 0, beta = 1, wscale = 1, no
     * conditions).  The forcing of a sample does not change from one training step to the next (the reference re-interpolates it to
     * the Gauss points every step, e.g. IBN/poisson-2d/parametric/IBN_2D.py:128-130), so it can be assembled once when the dataset is
     * placed on the device; the operator then spends one FMA per node on it instead of the element's forcing arithmetic (a fifth of
     * the 3-D Q1 kernel's time): out_a -= beta wscale b_a,  sum W f u = sum_a u_a b_a -- equal to the nodal-forcing result to rounding.
     * Only the 3-D Q1 two-element node-owner kernel (even nx, 2-point rule, constant-value uint8 / fp32 masks) takes it so far; every
     * other launch returns DN_E_UNSUPPORTED. */
    int32_t f_is_load;
} dn_poisson_args;

int dn_abi_version(void);
/* Human-readable build info ("gfx950 hipcc <ver> ..."), static storage. */
const char *dn_build_info(void);

/* Measurement probe, no reference counterpart: out = a * b + c over n floats (n % 4 == 0, 16-byte aligned arrays) as a plain streaming
 * kernel -- three arrays read once, one written once: the access mix of the fused energy loss + gradient without its halo rows, its
 * reduction and its arithmetic.  bench.py runs it over the arrays of the timed launches and reports the rate as
 * roofline.stream_ceiling.  mode: bit 0 non-temporal stores, bit 1 non-temporal loads, bits 2.. the form (0 one 16-byte vector per
 * thread, 1 blocks of four vectors per thread, 2 the same with 2048 persistent workgroups). */
int dn_probe_stream(const float *a, const float *b, const float *c, float *out, int64_t n, int32_t mode, void *stream);
/* Second probe: the MARCHING access pattern of the fused 2-D kernel without its arithmetic (arrays (B, ny, 512) fp32): a 128-thread
 * workgroup walks a strip of R node rows row by row with `rows_ahead` (1..4) rows requested ahead of the one it consumes.  flags: bit 0
 * read one halo row on either side of the strip, bit 1 also load the dword of the node shared with the right neighbour, bit 2
 * non-temporal stores, bit 3 non-temporal vector loads, bit 4 shared node taken from the neighbouring lane (one lane per wave loads it).
 * tools/march_probe.py: which (R, rows_ahead) the memory system rewards. */
int dn_probe_march(const float *a, const float *b, const float *c, float *out, int32_t B, int32_t ny, int32_t R, int32_t rows_ahead,
                   int32_t flags, void *stream);
/* Third probe: the TILE access pattern -- a workgroup requests all rows of a tile of rows_per_tile (4 | 8 | 16) node rows plus one halo row
 * on either side at once, parks them in LDS and walks them from there (threads 256 | 512; flags bit 3 non-temporal loads, bit 6
 * consecutive tiles on one XCD). */
int dn_probe_tile(const float *a, const float *b, const float *c, float *out, int32_t B, int32_t ny, int32_t rows_per_tile, int32_t threads,
                  int32_t flags, void *stream);

/* Meshes: nsd = 2, degree 1..3 (ngp 2..4, >= 3 for degree > 1); nsd = 3, degree 1 (marching kernels) and degree 2 / 3 with ngp 3 / 4 (element
 * vectors + fixed-order gather assembly: the workspace then also holds (degree + 1)^3 floats per element and is needed for every call, with
 * or without sums; no strip_select / accumulate_sums / defer_sums in that form).  dn_poisson_workspace_bytes says what a mesh needs. */
int64_t dn_poisson_workspace_bytes(const dn_mesh *mesh);
int dn_poisson_apply(const dn_mesh *mesh, const dn_poisson_args *args, void *stream);
/* Second half of a dn_poisson_apply launch with args->defer_sums set (same mesh, same args): the final scalars from the partial sums. */
int dn_poisson_finish_sums(const dn_mesh *mesh, const dn_poisson_args *args, void *stream);

/* Sticky error state of the launches that used `workspace` (dn_poisson_apply, dn_fsdt_apply).  The chained-strip launch plans hand rows
 * from wave to wave through flag-guarded LDS words; every poll of such a flag is bounded so that a producer wave that never arrives
 * cannot hang the GPU.  A poll that reaches its bound is never silent: the wave adds NaN to every value it writes afterwards (its rows
 * of `out`, its partial sums, hence the launch's energy / sumsq / norms) and sets bit 0 of the workspace's error word.  This call waits
 * for `stream`, reads the word, clears it, and returns 0 or DN_E_HANDOVER (synchronous: for health checks, not for the launch path).
 * The reference has no such state: its loss bodies are synchronous torch ops (IBN/poisson-2d/parametric/IBN_2D.py:116-134). */
int dn_workspace_status(void *workspace, void *stream);

/* gauss_pt_eval (DiffNet/DiffNetFEM.py:7-18) for an arbitrary table list, and its adjoint.
 *   out[b,g,e] = sum_a tables[g][a] * in[b, node(e,a)]          (conv_nd with stride `degree`)
 * in: (B,1,*N); tables: (G, nbf^nsd) device fp32; out: (B,G,*nel). nsd in {1,2,3}, nbf in 2..4.
 * n[] = nodes per axis in (x,y,z) order (unused axes 1).  The adjoint writes every element of
 * grad_in exactly once (gather form).  Replaces the 12 gauss_pt_evaluation* wrappers
 * (DiffNet/DiffNetFEM.py:143-174) and their autograd backward (conv_backward_input). */
int dn_gauss_pt_eval_fwd(const float *in, const float *tables, float *out, int32_t batch, int32_t nsd,
                         const int32_t n[3], int32_t nbf, int32_t stride, int32_t G, void *stream);
int dn_gauss_pt_eval_bwd(const float *grad_out, const float *tables, float *grad_in, int32_t batch, int32_t nsd,
                         const int32_t n[3], int32_t nbf, int32_t stride, int32_t G, void *stream);

/* Q1_2D_vector_assembly / Q1_3D_vector_assembly (e8_2d_poisson_mms.py:85-90, e8_3d_poisson_mms.py:78-87)
 * generalised to any degree: out[b, node] (+)= sum over the elements e and local ids a that map to node
 * of r_split[b,a,e], in the reference's a-ascending order.  accumulate = 0 overwrites `out`.
 * dn_assemble_bwd is its adjoint: grad_split[b,a,e] = grad_out[b, node(e,a)]. */
int dn_assemble(const float *r_split, float *out, int32_t batch, int32_t nsd, const int32_t n[3], int32_t nbf,
                int32_t stride, int32_t accumulate, void *stream);
int dn_assemble_bwd(const float *grad_out, float *grad_split, int32_t batch, int32_t nsd, const int32_t n[3],
                    int32_t nbf, int32_t stride, void *stream);

/* Winding-number inside/outside field: compute_winding_nodes of IBN/poisson-2d/parametric/IBN_2D.py:89-104.
 * points, normals: (B, Npts, 2); nodes: (2, Ny, Nx) node coordinates (xx, yy); out: (B, 1, Nx, Ny) in the reference's
 * (Nx, Ny) order: out[b,0,ix,iy] = sum_p ((p - q).n_p) / (4 pi |p - q|_1)^3 with q = nodes[:, iy, ix]. */
int dn_winding_nodes(const float *points, const float *normals, const float *nodes, float *out, int32_t batch,
                     int32_t npts, int32_t ny, int32_t nx, void *stream);

/* DiffNetFDM stencil derivatives (DiffNet/DiffNetFDM.py:158-199): out = Corr( conv2d(g_padded, kernel9) ) with
 * g_padded (B,1,ny+2,nx+2) replicate-padded by the caller as in the reference, out (B,1,ny,nx), kernel9 a HOST pointer to
 * the 3x3 correlation kernel (row-major), and the reference's dense correction matmul (:63-119) reduced to what it does:
 * along `axis` (0 = x: columns, 1 = y: rows) the two boundary lines become a*d[edge] + b*d[edge+-1].
 * dn_fdm_stencil_bwd is the adjoint (gradient wrt g_padded). */
int dn_fdm_stencil_fwd(const float *g_padded, float *out, int32_t batch, int32_t ny, int32_t nx, const float *kernel9,
                       int32_t axis, float a, float b, void *stream);
int dn_fdm_stencil_bwd(const float *grad_out, float *grad_g_padded, int32_t batch, int32_t ny, int32_t nx,
                       const float *kernel9, int32_t axis, float a, float b, void *stream);

/* ---- fused FSDT (Mindlin) plate residuals ---------------------------------------------------------------------
 * Replaces the loss body of examples/elasticity/single_instance/e1_plate_bending_fsdt.py:128-232 (9 Gauss-point
 * evaluations of w, phi_x, phi_y, the constitutive combinations, three weak-form residuals, their assembly, the
 * Dirichlet rows and the three Frobenius sums) by one launch; mesh: nsd = 2, degree 1..3, ngp 2..4.
 *   fields (B,1,ny,nx) fp32; bc_mask marks Dirichlet nodes (fp32: >= 0.5 as in the script, u8: != 0; NULL = none);
 *   on those nodes the inputs are replaced by bc_field[k] (or bc_value[k] when the pointer is NULL) and so are the
 *   rows of the k-th residual.  A44, A55 include the shear correction K_s; wscale = (hx/2)(hy/2).
 *   out[k]: assembled residual k (may be NULL); sumsq: 3 doubles, sum of squares of each residual field
 *   (deterministic in-kernel reduction; needs the workspace, zero-filled once before first use).
 * The operator is the gradient of the plate energy (symmetric Jacobian): the VJP wrt the fields is the same call on
 * the masked cotangents with q = 0 and zero Dirichlet values. */
typedef struct dn_fsdt_args {
    const float *w, *phi_x, *phi_y;
    const void *bc_mask;
    int32_t mask_is_u8, mask_batched;
    const float *bc_field[3];
    int32_t bc_field_batched[3];
    float bc_value[3];
    float D11, D12, D22, D66, A44, A55, q;
    float wscale;
    float *out[3];
    double *sumsq;
    void *workspace;
    int64_t workspace_bytes;
    const float *in_scale; /* optional: 3 device floats, field k is multiplied by in_scale[k] as it is loaded (before the
                              Dirichlet substitution) -- lets the VJP of the three norms run without a scaling pass */
    float *norms;          /* optional: 3 device floats, norms[k] = sqrt(sum of squares of residual k) written by the same launch -- the
                              three Frobenius norms of e1_plate_bending_fsdt.py:230-232 without a torch op behind the kernel */
    const float *in_num;   /* optional, both or neither (not with in_scale): field k is multiplied by in_num[k] / in_den[k] as it is */
    const float *in_den;   /* loaded, by 0 where in_den[k] <= 0: the VJP of the norms is this call on the saved residuals with
                              in_num = cotangents of the norms, in_den = the norms (torch's zero subgradient at a zero norm) */
    int32_t defer_sums;    /* (round 4) non-zero: the launch leaves its per-workgroup partial sums of squares in `workspace` and does NOT form
                              sumsq / norms (the arrival protocol and the last workgroup's reduction are 4.5-6.6 us at the end of every launch:
                              a third of the 1025^2 Q2 launch at one sample); sumsq and norms are ignored.  The value is the pair's TICKET: it is
                              left in the workspace next to the partials, and any launch that reduces in the kernel clears it */
    int32_t den_ticket;    /* with den_workspace: the ticket the producer was launched with; if the workspace holds another one (some other reducing
                              launch used it in between) the scales, this call's outputs and the norms come out NaN -- never silently stale */
    const void *den_workspace; /* (round 4) the `workspace` of an earlier launch with defer_sums on the same mesh, stream and launch plan: every
                              workgroup of THIS launch reduces those partials (fixed order: bitwise the sums the producer would have formed)
                              and uses their square roots where it would have read in_den (in_num required, in_den NULL); if sumsq / norms of
                              this call are non-NULL they receive the producer's sums / norms.  The loss + gradient of the plate is then two
                              launches with no reduction on the critical path of either */
} dn_fsdt_args;
int64_t dn_fsdt_workspace_bytes(const dn_mesh *mesh);
int dn_fsdt_apply(const dn_mesh *mesh, const dn_fsdt_args *args, void *stream);

/* ---- fused output block of the 2-D U-Net generator ----------------------------------------------------------
 * Upsample(x2, nearest) -> ZeroPad2d((1,0,1,0)) -> Conv2d(C -> 1, 4x4, padding 1, bias) -> Sigmoid
 * (DiffNet/networks/unets.py:68-74, `self.final`) without materialising the upsampled tensor.
 *   in (B,C,h,w), weight (1,C,4,4), bias (1) device pointer or NULL, out / grad_out (B,1,2h,2w); act: 1 = sigmoid, 0 = none.
 * bwd: grad_in (B,C,h,w), grad_weight (1,C,4,4), grad_bias (1); any of the three may be NULL (grad_bias needs
 * grad_weight).  `out` is the forward result (needed when act = 1).  workspace: dn_upconv_out_workspace_bytes, no
 * initialisation required.  Weight-gradient partial sums are combined in a fixed order (bitwise repeatable). */
int64_t dn_upconv_out_workspace_bytes(int64_t B, int64_t C, int64_t h, int64_t w);
int dn_upconv_out_fwd(const float *in, const float *weight, const float *bias, float *out, int64_t B, int64_t C, int64_t h,
                      int64_t w, int act, void *workspace, int64_t workspace_bytes, void *stream);
int dn_upconv_out_bwd(const float *in, const float *weight, const float *out, const float *grad_out, float *grad_in,
                      float *grad_weight, float *grad_bias, int64_t B, int64_t C, int64_t h, int64_t w, int act, void *workspace,
                      int64_t workspace_bytes, void *stream);

/* 3-D counterpart: Upsample(x2, nearest) -> Conv3d(C -> 1, 3x3x3, padding 1, bias) -> Sigmoid
 * (DiffNet/networks/wgan3d.py:88-92, `GoodGenerator.final`).  in (B,C,d,h,w), weight (1,C,3,3,3), out (B,1,2d,2h,2w);
 * arguments as for dn_upconv_out_*. */
int64_t dn_upconv3d_out_workspace_bytes(int64_t B, int64_t C, int64_t d, int64_t h, int64_t w);
int dn_upconv3d_out_fwd(const float *in, const float *weight, const float *bias, float *out, int64_t B, int64_t C, int64_t d,
                        int64_t h, int64_t w, int act, void *workspace, int64_t workspace_bytes, void *stream);
int dn_upconv3d_out_bwd(const float *in, const float *weight, const float *out, const float *grad_out, float *grad_in,
                        float *grad_weight, float *grad_bias, int64_t B, int64_t C, int64_t d, int64_t h, int64_t w, int act,
                        void *workspace, int64_t workspace_bytes, void *stream);

/* Weight gradient of the 4x4x4 / stride 2 / padding 1 3-D convolutions and transposed convolutions of the generator
 * (DiffNet/networks/wgan3d.py:23-55):
 *     grad_weight[m][cn][kz][ky][kx] = sum_{b,i,j,k} coarse[b][m][i][j][k] * fine[b][cn][2i+kz-1][2j+ky-1][2k+kx-1]
 *   Conv3d:          coarse = grad_out (B,cout,d,h,w), fine = input    (B,cin,2d,2h,2w),  grad_weight (cout,cin,4,4,4)
 *   ConvTranspose3d: coarse = input    (B,cin,d,h,w),  fine = grad_out (B,cout,2d,2h,2w), grad_weight (cin,cout,4,4,4)
 * M (channels of `coarse`) <= 128.  Fixed-order partial sums (bitwise repeatable); workspace needs no initialisation. */
int64_t dn_conv3d_k4s2_wrw_workspace_bytes(int64_t B, int64_t CN, int64_t M, int64_t d, int64_t h, int64_t w);
int dn_conv3d_k4s2_wrw(const float *fine, const float *coarse, float *grad_weight, int64_t B, int64_t CN, int64_t M, int64_t d,
                       int64_t h, int64_t w, void *workspace, int64_t workspace_bytes, void *stream);

/* Fused InstanceNorm (affine = False, biased variance) + LeakyReLU/ReLU of the generator blocks
 * (DiffNet/networks/unets.py:13-45, autoencoders.py:7-70, wgan3d.py:23-55): x, y (n_inst, spatial) contiguous with
 * n_inst = B*C; mean, rstd (n_inst) are written by fwd and consumed by bwd.  slope: 0 = ReLU, 0.2 = LeakyReLU(0.2),
 * 1 = no activation.  With few, large instances the work of one instance is sliced over several workgroups and the
 * fp64 partial sums go through `workspace` (dn_instnorm_workspace_bytes(n_inst, spatial) bytes, 0 when not needed;
 * no initialisation required).  Results are bitwise repeatable.  bwd: grad_y may be a channel slice of a wider
 * (B, C_total, spatial) tensor, e.g. the U-Net's skip concatenation: `channels` = C of this layer and
 * grad_y_batch_stride = elements between samples (0 = contiguous). */
int64_t dn_instnorm_workspace_bytes(int64_t n_inst, int64_t spatial);
int dn_instnorm_act_fwd(const float *x, float *y, float *mean, float *rstd, int64_t n_inst, int64_t spatial, float eps,
                        float slope, void *workspace, int64_t workspace_bytes, void *stream);
int dn_instnorm_act_bwd(const float *x, const float *mean, const float *rstd, const float *grad_y, float *grad_x,
                        int64_t n_inst, int64_t spatial, float slope, int64_t channels, int64_t grad_y_batch_stride,
                        void *workspace, int64_t workspace_bytes, void *stream);

/* 4 x 4 / stride 2 / padding 1 convolution family of the 2-D networks (no bias), NCHW fp32, on the fp32 matrix cores.
 * Replaces the arithmetic of `nn.Conv2d(cin, cout, 4, 2, 1, bias=False)` (UNetDown, DiffNet/networks/unets.py:13-24;
 * autoencoders.py:29-34) and `nn.ConvTranspose2d(cin, cout, 4, 2, 1, bias=False)` (UNetUp, unets.py:27-45; autoencoders.py:39-45)
 * in all their passes.  `fine` lives on the 2H x 2W grid, `coarse` on the H x W grid, and w[m][c][ky][kx] (m = coarse channel,
 * c = fine channel) is Conv2d's (cout, cin, 4, 4) weight and ConvTranspose2d's (cin, cout, 4, 4) weight as they stand:
 *   down: coarse[b,m,i,j] = sum w[m,c,ky,kx] fine[b,c,2i+ky-1,2j+kx-1]      Conv2d forward / ConvTranspose2d input gradient
 *   up  : fine[b,c,y,x]   = sum w[m,c,ky,kx] coarse[b,m,(y+1-ky)/2,(x+1-kx)/2]  ConvTranspose2d forward / Conv2d input gradient
 *   wrw : gw[m,c,ky,kx]   = sum coarse[b,m,i,j] fine[b,c,2i+ky-1,2j+kx-1]   weight gradient of both (deterministic two-stage sum)
 * Shapes: fine (B,C,2H,2W), coarse (B,M,H,W), w / grad_weight (M,C,4,4). */
int dn_conv2d_k4s2_down(const float *fine, const float *w, float *coarse, int64_t B, int64_t C, int64_t M, int64_t H, int64_t W,
                        void *stream);
int dn_conv2d_k4s2_up(const float *coarse, const float *w, float *fine, int64_t B, int64_t C, int64_t M, int64_t H, int64_t W,
                      void *stream);
int64_t dn_conv2d_k4s2_wrw_workspace_bytes(int64_t B, int64_t C, int64_t M, int64_t H, int64_t W);
int dn_conv2d_k4s2_wrw(const float *fine, const float *coarse, float *grad_weight, int64_t B, int64_t C, int64_t M, int64_t H,
                       int64_t W, void *workspace, int64_t workspace_bytes, void *stream);

/* 4 x 4 x 4 / stride 2 / padding 1 convolution family of the 3-D generator (no bias), NCDHW fp32, fp32 matrix cores: the `down`
 * and `up` contractions of `nn.Conv3d(cin, cout, 4, 2, 1, bias=False)` / `nn.ConvTranspose3d(cin, cout, 4, 2, 1, bias=False)`
 * (DiffNet/networks/wgan3d.py:23-55), same conventions as dn_conv2d_k4s2_*: fine (B,C,2D,2H,2W), coarse (B,M,D,H,W), w (M,C,4,4,4).
 * The weight gradient of both layers is dn_conv3d_k4s2_wrw. */
int dn_conv3d_k4s2_down(const float *fine, const float *w, float *coarse, int64_t B, int64_t C, int64_t M, int64_t D, int64_t H,
                        int64_t W, void *stream);
int dn_conv3d_k4s2_up(const float *coarse, const float *w, float *fine, int64_t B, int64_t C, int64_t M, int64_t D, int64_t H,
                      int64_t W, void *stream);
/* The same contractions with a workspace (round 4): the deep, narrow layers of the generator (128 -> 128 channels on 4^3 positions ...) fill
 * only a few position tiles, so the contraction itself is split over workgroups -- slices of fine channels (`down`) / coarse channel
 * pairs (`up`) write partial results into the workspace and a second launch adds them in slice order (deterministic).  The library
 * decides the split from the shapes; dn_conv3d_k4s2_workspace_bytes (up = 0 / 1) returns what that split needs (0: no split).  A NULL
 * workspace runs the unsplit launch, which is what dn_conv3d_k4s2_down / _up are. */
int64_t dn_conv3d_k4s2_workspace_bytes(int32_t up, int64_t B, int64_t C, int64_t M, int64_t D, int64_t H, int64_t W);
int dn_conv3d_k4s2_down_ws(const float *fine, const float *w, float *coarse, int64_t B, int64_t C, int64_t M, int64_t D, int64_t H,
                           int64_t W, void *workspace, int64_t workspace_bytes, void *stream);
int dn_conv3d_k4s2_up_ws(const float *coarse, const float *w, float *fine, int64_t B, int64_t C, int64_t M, int64_t D, int64_t H,
                         int64_t W, void *workspace, int64_t workspace_bytes, void *stream);

/* Stride-1 "valid" k x k convolutions (k <= 7) with optional bias: the auto-encoder's stem / head layers behind an explicit
 * ReflectionPad2d (DiffNet/networks/autoencoders.py:13 `nn.Conv2d(in_channels, dim*2, 7)`, :75 `nn.Conv2d(.., out_channels, 3)`,
 * `nn.Conv2d(out_channels, out_channels, 7)`).  x (B,Ci,H,W), w (Co,Ci,K,K), y / gy (B,Co,H-K+1,W-K+1); gbias may be NULL. */
int dn_conv2d_valid_fwd(const float *x, const float *w, const float *bias, float *y, int64_t B, int64_t Ci, int64_t Co, int64_t H, int64_t W,
                        int64_t K, void *stream);
int dn_conv2d_valid_bwd_data(const float *gy, const float *w, float *gx, int64_t B, int64_t Ci, int64_t Co, int64_t H, int64_t W, int64_t K,
                             void *stream);
int dn_conv2d_valid_bwd_weight(const float *x, const float *gy, float *gw, float *gbias, int64_t B, int64_t Ci, int64_t Co, int64_t H, int64_t W,
                               int64_t K, void *stream);

/* dn_fdm_stencil_* with the replicate padding folded in (`self.pad(u)` of DiffNet/DiffNetFDM.py:128-199 becomes clamped indexing):
 * u and out are both (B,1,ny,nx); the backward pass returns the gradient on the unpadded grid (adjoint of pad + stencil + fix-up). */
int dn_fdm_fused_fwd(const float *u, float *out, int32_t batch, int32_t ny, int32_t nx, const float *kernel9, int32_t axis, float a,
                     float b, void *stream);
int dn_fdm_fused_bwd(const float *grad_out, float *grad_u, int32_t batch, int32_t ny, int32_t nx, const float *kernel9, int32_t axis,
                     float a, float b, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFNET_HIP_H */
