/*
 * miosqp_amd.h -- C ABI of the MI355X QP-relaxation engine (libmiosqp_hip.so).
 *
 * Drop-in boundary for ONE path of miOSQP: the relaxation solve that Node.solve() triggers.
 * The reference reaches that path through five Python calls on one shared `osqp.OSQP` object;
 * each entry point below names the reference call it replaces.  Plain pointers and sizes only;
 * every array argument is copied during the call (the reference keeps mutating its numpy
 * arrays afterwards: /root/reference/miosqp/data.py:120,126), outputs are written into
 * caller-owned buffers.  All functions return 0 on success, a negative MIOSQP_E* code on error,
 * or a positive value where documented.  Not re-entrant per handle (the reference is
 * single-threaded: one solver object mutated in place, workspace.py:63).
 *
 * Matrices: CSC, 32-bit indices, fp64 values (scipy's native layout, README.md:31 of the
 * reference).  P may be the full symmetric matrix or its upper triangle; only entries with
 * row <= col are read.  A is the EXTENDED constraint matrix [A; I[i_idx,:]] built by
 * /root/reference/miosqp/data.py:5-33, M = m + n_int rows.
 */
#ifndef MIOSQP_AMD_H
#define MIOSQP_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* solve outcome (info.status_val); values follow OSQP 0.6.x, looked up by name through
 * miosqp_qp_constant() exactly as the reference does with osqp.constant(name)
 * (/root/reference/miosqp/node.py:88,128-129; workspace.py:294-295,403-404,419). */
#define MIOSQP_QP_SOLVED 1
#define MIOSQP_QP_MAX_ITER_REACHED (-2)
#define MIOSQP_QP_PRIMAL_INFEASIBLE (-3)
#define MIOSQP_QP_DUAL_INFEASIBLE (-4)
#define MIOSQP_QP_UNSOLVED (-10)

/* error codes */
#define MIOSQP_EARG (-1)      /* bad argument (NULL, negative size, crossed bounds at setup) */
#define MIOSQP_EHIP (-2)      /* a HIP runtime call failed; see miosqp_qp_last_error() */
#define MIOSQP_EFACTOR (-3)   /* KKT factorisation broke down (non-convex P?) */
#define MIOSQP_ENODEV (-4)    /* no usable gfx950 device */
#define MIOSQP_EUNSUPPORTED (-5) /* the entry point does not cover this problem size / engine form (caller falls back) */
#define MIOSQP_EFULL (-6)     /* the leaf store of miosqp_qp_search_* has no free slot (raise the capacity) */
#define MIOSQP_EBOUNDS 1      /* update_bounds: some l[i] > u[i]; nothing was changed */

/* Solver parameters = the keyword arguments the reference forwards verbatim to
 * osqp.OSQP.setup (/root/reference/miosqp/workspace.py:67-68).  The reference's examples set
 * only eps_abs, eps_rel, eps_prim_inf, verbose (examples/random_miqp/run_example.py:113-116);
 * everything else is this table (DESIGN.md "frozen spec"). rho is ONE scalar for all rows and
 * is never adapted, so the factor is shared by every branch-and-bound node. */
typedef struct miosqp_qp_settings {
  double rho;          /* 0.1  */
  double sigma;        /* 1e-6 */
  double alpha;        /* 1.6  */
  double eps_abs;      /* 1e-3 */
  double eps_rel;      /* 1e-3 */
  double eps_prim_inf; /* 1e-4 */
  double eps_dual_inf; /* 1e-4 */
  int32_t max_iter;          /* 4000 */
  int32_t scaling;           /* 10 Ruiz passes, 0 = off */
  int32_t check_termination; /* 25: residuals are tested every this many iterations */
  int32_t warm_start;        /* 1: solve() continues from the stored iterates */
  int32_t device;            /* HIP device ordinal; -1 = current device */
  int32_t max_batch;         /* capacity of miosqp_qp_solve_batch (>= 1) */
  int32_t fold;              /* -1 auto, 0 factor form L (4 kernels/iteration), 1 product form L^-1
                                (2 kernels/iteration; auto picks it for panels denser than 30 %) */
  int32_t resident;          /* -1 auto, 0 off, 1 on: whole solve in ONE workgroup: iterates in LDS; the factor as the
                                explicit KKT inverse in the workgroup's registers (n+M <= 192, the automatic
                                choice there) or in product form in LDS (when that fits 160 KB) */
  int32_t setup_on_device;   /* -1 auto (n >= 1024), 0 host, 1 device: dense LDL^T of the reduced Hessian and
                                the inverse of its triangular factor computed on the GPU at setup */
  int32_t coop;              /* -1 auto, 0 off, 1 on: cooperative register-resident solver -- the explicit KKT
                                inverse (n+M)^2 spread over the register files of up to one workgroup per
                                CU, ONE exchange per iteration (n+M <= 2048; auto from n+M = 193 on) */
  int32_t pers;              /* -1 auto, 0 off, 1 on: persistent streaming solver -- ONE launch per solve that reads the
                                factor (product or factor form, whichever `fold` selects) from memory in every
                                iteration; the kernel boundaries of the multi-kernel forms become tagged exchanges
                                between co-resident workgroups.  Any size that fits one workgroup's LDS operand
                                buffers (n+M <= ~17 000 in product form, n <= ~9 000 in factor form); auto: problems
                                beyond the cooperative solver's n+M <= 2048.  2 (factor form): the dense tail as the
                                explicit inverse of the reduced Hessian, S^-1 = L22^-T D22^-1 L22^-1 -- the bytes of the
                                two triangles, ONE dense phase per iteration instead of two */
  int32_t batch_pers;        /* -1 auto, 0 off, 1 on: batched mode (solve_batch, the leaf pool's stream) runs the lock-step
                                iterations of a chunk as ONE persistent launch with the product-form factor held in the
                                registers and LDS of 256 co-resident workgroups (n <= 512, rows in the products <= 1024;
                                auto: n >= 256).  Called off -- the chunk goes through two launches per iteration
                                instead -- when the device is shared and the workgroups are not co-resident in time */
  int32_t rho_auto;          /* 0 (default): `rho` as given.  1: rho is chosen ONCE per problem at setup and then frozen
                                (opt-in; Python: rho="auto").  The reference hands only eps_* to osqp.setup
                                (/root/reference/miosqp/workspace.py:67-68, examples/random_miqp/run_example.py:113-116),
                                i.e. OSQP's defaults, which adapt rho while solving; here one factor serves every node of
                                a tree, so OSQP's update rule -- rho <- rho sqrt(normalised primal residual / normalised
                                dual residual), scaled quantities, clamped to [1e-6, 1e6] -- is applied once, to the
                                iterates 50 iterations from zero on the setup's bounds, rounded to two significant
                                digits, and the KKT matrix is factorised for that value.  Config 2: 0.1 -> 0.013,
                                a third of the iterations per node.  miosqp_qp_get_rho reports the value in use */
} miosqp_qp_settings;

/* What the reference reads from `results.info` (/root/reference/miosqp/node.py:111-125) plus
 * residuals and the fused node epilogue. */
typedef struct miosqp_qp_info {
  int32_t status_val; /* node.py:111 */
  int32_t iter;       /* node.py:118 */
  double run_time;    /* node.py:121: host wall seconds of this solve call */
  double obj_val;     /* relaxation objective at the returned x (unused by the reference) */
  double pri_res;
  double dua_res;
  double device_time; /* seconds between HIP events around the device work of this call */
  double lower;       /* solve_node/solve_batch only: objective at the clamped x (node.py:143) */
  /* node digest (after miosqp_qp_set_root; -1 / NaN otherwise): what bound_and_branch needs next */
  int32_t int_inf;    /* number of integer variables off by more than eps_int_feas (workspace.py:245-264) */
  int32_t nextvar;    /* position in i_idx of the most fractional one (workspace.py:205-230) */
  double heur_viol;   /* worst violation of the root bounds by the rounded candidate, eps_abs slack
                         included: <= 0 means satisfies_lin_constraints (workspace.py:232-243, 321-323) */
  double heur_obj;    /* objective of the rounded candidate (workspace.py:324) */
} miosqp_qp_info;

typedef struct miosqp_qp_engine miosqp_qp_engine;

/* fills *s with the defaults tabulated above */
int miosqp_qp_default_settings(miosqp_qp_settings *s);

/* osqp.constant(name) -- /root/reference/miosqp/node.py:88.  Unknown name -> 0. */
int miosqp_qp_constant(const char *name);

/* osqp.OSQP().setup(P, q, A, l, u, **qp_settings) -- /root/reference/miosqp/workspace.py:63-68.
 * Scales the problem, factorises the KKT matrix once and places factor + matrices in HBM. */
int miosqp_qp_setup(miosqp_qp_engine **out, int32_t n, int32_t M,
                    const int32_t *P_colptr, const int32_t *P_rowidx, const double *P_val,
                    const int32_t *A_colptr, const int32_t *A_rowidx, const double *A_val,
                    const double *q, const double *l, const double *u,
                    const miosqp_qp_settings *settings);

/* solver.update(l=l, u=u) -- /root/reference/miosqp/node.py:102.  l, u: M doubles each. */
int miosqp_qp_update_bounds(miosqp_qp_engine *e, const double *l, const double *u);

/* solver.update(q=q) -- /root/reference/miosqp/solver.py:183-185.  q: n doubles. */
int miosqp_qp_update_lin_cost(miosqp_qp_engine *e, const double *q);

/* solver.warm_start(x=x, y=y) -- /root/reference/miosqp/node.py:105.  x: n, y: M doubles. */
int miosqp_qp_warm_start(miosqp_qp_engine *e, const double *x, const double *y);

/* results = solver.solve() -- /root/reference/miosqp/node.py:108-125.
 * x_out: n doubles, y_out: M doubles (fresh copies), info filled. */
int miosqp_qp_solve(miosqp_qp_engine *e, double *x_out, double *y_out, miosqp_qp_info *info);

/* Declares which rows of A are the integer-bound rows so that the node epilogue can run on
 * the device: rows m .. m+n_int-1 of A are I[i_idx,:] (/root/reference/miosqp/data.py:19-29).
 * i_idx: n_int variable indices.  Must be called before solve_node / solve_batch. */
int miosqp_qp_set_integer_rows(miosqp_qp_engine *e, int32_t n_int, const int32_t *i_idx,
                               int32_t m_orig);

/* Enables the node digest: the part of Workspace.bound_and_branch that only needs the node's x
 * (/root/reference/miosqp/workspace.py:245-272, 321-324) is evaluated on the device at the end of
 * solve_node.  l_root/u_root: the ROOT bounds data.l / data.u (M doubles); eps_int_feas from the
 * B&B settings; eps_lin = qp_settings['eps_abs'].  Call again after MIOSQP.update_vectors. */
int miosqp_qp_set_root(miosqp_qp_engine *e, const double *l_root, const double *u_root,
                       double eps_int_feas, double eps_lin);

/* Whole body of Node.solve() in one call -- /root/reference/miosqp/node.py:96-143:
 * update(l,u) -> warm_start(x0,y0) -> solve -> x[i_idx] clamped into [l[-n_int:], u[-n_int:]]
 * (node.py:131-136) -> lower = .5 x'Px + q'x at the clamped x (node.py:143, data.py:99-103).
 * For infeasible outcomes x/y hold the certificate as solve() returns it and lower is NaN. */
int miosqp_qp_solve_node(miosqp_qp_engine *e, const double *l, const double *u,
                         const double *x0, const double *y0, double *x_out, double *y_out,
                         miosqp_qp_info *info);

/* B independent nodes sharing the factor (leaves of Workspace.leaves,
 * /root/reference/miosqp/workspace.py:83), node-major arrays: l,u,y0,y_out [B][M];
 * x0,x_out [B][n]; info [B].  Each node's result equals what solve_node returns for it. */
int miosqp_qp_solve_batch(miosqp_qp_engine *e, int32_t B, const double *l, const double *u,
                          const double *x0, const double *y0, double *x_out, double *y_out,
                          miosqp_qp_info *info);

/* ---- a whole tree search in one launch (small problems) ------------------------------------------------
 * the MPC re-solve path (/root/reference/miosqp/solver.py:65-172 per MIQP,
 * examples/power_converter/power_converter.py:421-508 per sampling step).  For problems the LDS-resident solver
 * handles, the loop `while can_continue: choose_leaf -> Node.solve -> bound_and_branch` runs inside ONE launch of
 * one workgroup with the host logic's exact decisions (miosqp_amd/csrc/kernels_tree.inc); the host sends the root
 * and gets the incumbent back.  Returns MIOSQP_EUNSUPPORTED when the problem is too large for that form. */
typedef struct miosqp_tree_info {
  int32_t nodes;        /* nodes visited = iter_num - 1 */
  int32_t osqp_iter;    /* ADMM iterations over all of them */
  int32_t leaves_left;  /* open leaves when the loop ended (0: the tree is closed) */
  int32_t overflow;     /* 1: more than 1024 leaves alive -- the result is not usable, redo on the host */
  int32_t max_leaves;
  int32_t found;        /* 1: an incumbent was found in this launch (x_out holds it) */
  double upper_glob, lower_glob;
  double device_time, run_time;
} miosqp_tree_info;

/* l, u, x0, y0: the root node (M, M, n, M doubles); upper0 / x_inc0: incumbent from MIOSQP.set_x0 (x_inc0 NULL or
 * upper0 >= 1.7e308: none); tree_explor_rule, max_iter_bb: the B&B settings; branching_rule 0 is implied. */
int miosqp_qp_solve_tree(miosqp_qp_engine *e, const double *l, const double *u, const double *x0, const double *y0,
                         double upper0, const double *x_inc0, int32_t tree_explor_rule, int32_t max_iter_bb,
                         double *x_out, miosqp_tree_info *info);

/* B independent MIQPs that share P and A -- hence the factor -- and differ in q, l, u, warm start and incumbent (the
 * reference's MPC pattern: MIOSQP.update_vectors + set_x0 + solve per instance, /root/reference/miosqp/solver.py:174-212),
 * every tree in ONE launch: workgroup (or wavefront, n + M <= 64) b runs instance b exactly as miosqp_qp_solve_tree would
 * after miosqp_qp_update_lin_cost(q_b).  All arrays are instance-major: q, x0, x_inc0, x_out B x n; l, u, y0 B x M;
 * upper0 B (>= 1.7e308: no incumbent; x_inc0 may be NULL when none has one); info B.  The engine's own linear cost and
 * bounds are not touched.  MIOSQP_EUNSUPPORTED as for miosqp_qp_solve_tree. */
int miosqp_qp_solve_trees(miosqp_qp_engine *e, int32_t B, const double *q, const double *l, const double *u,
                          const double *x0, const double *y0, const double *upper0, const double *x_inc0,
                          int32_t tree_explor_rule, int32_t max_iter_bb, double *x_out, miosqp_tree_info *info);

/* ---- node-at-a-time branch and bound, driven from the host in C++ -------------------------------------
 * The loop of /root/reference/miosqp/solver.py:65-172 (choose_leaf -> Node.solve -> bound_and_branch, workspace.py:
 * 113-155, 274-334) for problems of any size, one relaxation at a time in whatever form the engine uses for single
 * nodes.  The open leaves are device slots (integer-row bounds + solution = the children's warm start), the children
 * are written on the device (workspace.py:157-203), a node's outcome reaches the host as a 96-byte record: no vector
 * crosses PCIe per node and no interpreter runs between two nodes.  Same list semantics as the Python mirror
 * (creation order, first maximum, the prune traversal of workspace.py:278-280); the heuristic incumbent's value is
 * the device's.  Needs set_integer_rows + set_root. */
typedef struct miosqp_search_info {
  int64_t nodes;        /* nodes solved in this call */
  int64_t osqp_iter;    /* ADMM iterations over them */
  int32_t open_leaves;  /* leaves still open (0: the tree is closed) */
  int32_t free_slots;
  int32_t improved;     /* the incumbent improved in this call: 1 = last by an integer-feasible node, 2 = last by the
                           rounding heuristic (its value is the device's sum: recompute on the host and hand it back
                           through miosqp_qp_search_set_incumbent(e, value, NULL) for workspace.py:321-327's number) */
  int32_t reserved;
  double upper_glob, lower_glob;
  double device_time;   /* seconds inside the ADMM loops of these nodes (device events) */
  double run_time;
} miosqp_search_info;

int miosqp_qp_search_create(miosqp_qp_engine *e, int32_t capacity);
/* new MIQP on the same factor: no leaves, no incumbent */
int miosqp_qp_search_reset(miosqp_qp_engine *e);
/* appends a leaf given with explicit vectors (the root, or one from another rank): l_int, u_int (n_int), x0 (n), y0 (M).
 * Here and in take_leaf / miosqp_qp_stream_add_leaf / _take_leaf / miosqp_qp_pool_write_node / _read_node the four vectors
 * may live in HOST or in DEVICE memory (unified addressing; the copies are hipMemcpyDefault): between ranks a leaf
 * travels as one device buffer -- slot store -> RCCL broadcast -> slot store -- and never visits the host
 * (miosqp_amd/dist.py: ShardedStream; the l <= u check is the host's and is skipped for device memory). */
int miosqp_qp_search_add_leaf(miosqp_qp_engine *e, const double *l_int, const double *u_int, const double *x0,
                              const double *y0, int32_t depth, double lower);
/* removes the shallowest open leaf and returns it with explicit vectors; 1 when there is none */
int miosqp_qp_search_take_leaf(miosqp_qp_engine *e, double *l_int, double *u_int, double *x0, double *y0,
                               int32_t *depth, double *lower);
/* adopts an incumbent from outside when it is better (MIOSQP.set_x0, another rank) and prunes against it;
 * x == NULL: only the VALUE of the incumbent the search already holds is replaced (no comparison, no pruning) */
int miosqp_qp_search_set_incumbent(miosqp_qp_engine *e, double upper, const double *x);
/* *upper >= 1.7e308: none yet (x untouched) */
int miosqp_qp_search_get_incumbent(miosqp_qp_engine *e, double *upper, double *x);
/* solves nodes until the list is empty, max_nodes are done or budget_s seconds have passed (<= 0: no time limit).
 * The slot store grows by itself (the capacity of search_create is a starting size); MIOSQP_EFULL only when the
 * device has no memory left for it -- *info is filled with what was done up to then in that case too. */
int miosqp_qp_search_run(miosqp_qp_engine *e, int32_t tree_explor_rule, int64_t max_nodes, double budget_s,
                         miosqp_search_info *info);

/* ---- device-resident leaf pool + streaming batch -------------------------------------------------------
 * Workspace.leaves and child generation (/root/reference/miosqp/workspace.py:83,
 * 157-203) kept on the device.  A node = its integer-row bounds (rows m .. M-1; workspace.py:227) + a warm
 * start = the solution of its parent's slot (workspace.py:174-176, 198-200).  The host owns slot numbers and
 * the search logic (choose_leaf / prune / incumbent: workspace.py:128-155, 274-334) on 64-byte digests; the
 * only vectors that cross PCIe are an incumbent's x when one is found.  The batch's columns are refilled
 * from a ready ring between chunks (check_termination iterations + one test), so a finished node's column
 * does not wait for the slowest node of a wave; every node still gets exactly the iterations and tests
 * miosqp_qp_solve_node would give it.  Needs set_integer_rows + set_root, and max_iter % check_termination == 0. */
typedef struct miosqp_pool_digest {
  int32_t slot;        /* pool slot of the node */
  int32_t status_val;  /* as miosqp_qp_info.status_val; -100 = dropped at refill: its bound exceeded the incumbent */
  int32_t iter;
  int32_t int_inf;     /* -1 when the relaxation was infeasible */
  int32_t nextvar;     /* branching variable (position in i_idx); the children were written when int_inf > 0 */
  int32_t reserved;
  double lower;        /* objective at the clamped x (node.py:143); NaN when infeasible */
  double heur_viol;    /* <= 0: the rounded candidate satisfies the root constraints */
  double heur_obj;
  double pri_res, dua_res;
} miosqp_pool_digest;

/* capacity: node slots; columns: width of the streaming batch (<= 1024, rounded up to 64) */
int miosqp_qp_pool_create(miosqp_qp_engine *e, int32_t capacity, int32_t columns);
/* empties columns, ready ring and digests (new MIQP on the same factor: after update_vectors / set_root) */
int miosqp_qp_pool_reset(miosqp_qp_engine *e);
/* a node given by the host (root, or a leaf received from another rank): integer-row bounds (n_int each) and an
 * explicit warm start x0 (n), y0 (M) */
int miosqp_qp_pool_write_node(miosqp_qp_engine *e, int32_t slot, const double *l_int, const double *u_int,
                              const double *x0, const double *y0);
/* integer-row bounds and, for a solved node, its solution (any pointer may be NULL) */
int miosqp_qp_pool_read_node(miosqp_qp_engine *e, int32_t slot, double *l_int, double *u_int, double *x, double *y);
/* appends nodes to the ready ring in the order they should be solved: slot, the two slots its children are to be
 * written into (-1: none), and the bound it inherited (dropped unsolved once that exceeds the incumbent) */
int miosqp_qp_pool_push(miosqp_qp_engine *e, int32_t count, const int32_t *slot, const int32_t *child0,
                        const int32_t *child1, const double *lower);
/* incumbent value for the refill-time bound test */
int miosqp_qp_pool_set_upper(miosqp_qp_engine *e, double upper);
/* enqueues `chunks` x (refill, check_termination iterations, test, harvest) and returns at once */
int miosqp_qp_pool_launch(miosqp_qp_engine *e, int32_t chunks);
/* waits until at most `keep_in_flight` launches are still running (oldest first: with 1 the device works on the
 * newest launch while the host handles the results of the one before); digests of the nodes decided (or dropped)
 * since the last call, at most max_out; *active = columns holding a node after the last refill seen,
 * *ready_left = ready-ring entries not yet taken */
int miosqp_qp_pool_collect(miosqp_qp_engine *e, int32_t keep_in_flight, miosqp_pool_digest *out, int32_t max_out,
                           int32_t *n_out, int32_t *active, int64_t *ready_left);

/* frees device and host memory */
int miosqp_qp_cleanup(miosqp_qp_engine *e);

/* ---- introspection used by bench.py and the parity tests (not part of the reference API) -- */

/* human-readable text of the last failure on this thread */
const char *miosqp_qp_last_error(void);

/* scaled iterates after running exactly k ADMM iterations from the current state with the
 * termination test disabled (iterate-level parity against the oracle). x: n, z,y: M. */
int miosqp_qp_debug_iterate(miosqp_qp_engine *e, int32_t k, double *x, double *z, double *y);

/* D (n), E (M), c of the Ruiz equilibration */
int miosqp_qp_get_scaling(miosqp_qp_engine *e, double *D, double *E, double *c);

/* sizes of the factor: out[0]=nnz(L) strict (panel + tail), out[1]=nnz panel, out[2]=tail order,
 * out[3]=algorithmic bytes per ADMM iteration (SURVEY.md sec. 8d formula), out[4..6] threads per
 * row of the panel/tail kernels, out[7] bit 0 = product-form factor in use, bit 1 = LDS-resident solver in use,
 * bit 2 = dense setup stages ran on the device, bit 3 = cooperative solver in use, bits 8..15 = its calibrated
 * poll delay (64-clock units); out[8] = bytes per iteration the kernels of the form in use actually request (dense
 * blocks carry no index array: 8 B per entry instead of the formula's 12); out[9] = times the engine fell back from
 * the cooperative form; out[7] bit 17 = the explicit KKT inverse failed its residual check at set-up and the engine
 * iterates with the factor's sweeps instead (see miosqp_qp_get_inverse_guard); bit 18 = launches of the hosted search counted
 * by miosqp_qp_get_loop_launches since the last reset were launches of the resident grid (one per miosqp_qp_search_run).
 * out must hold 10 values. */
int miosqp_qp_get_factor_stats(miosqp_qp_engine *e, int64_t *out);

/* rho in use (differs from settings.rho when settings.rho_auto chose it) */
int miosqp_qp_get_rho(miosqp_qp_engine *e, double *rho);

/* The register-resident solvers iterate on the explicit inverse of the scaled KKT matrix (sigma = 1e-6), which -- unlike
 * the sweeps with the LDL^T factor the reference's linear solver performs at /root/reference/miosqp/workspace.py:63-68
 * (osqp.setup) and node.py:108 (solve) -- is not backward stable.  At set-up the device measures
 * max |K W r - r| / max |r| over three probe vectors; above the threshold (1e-9; MIOSQP_GUARD_TOL overrides, negative
 * switches the check off) the engine does not use the inverse.  out[0] = measured residual (-1: no inverse was built),
 * out[1] = threshold, out[2] = 1 when the engine fell back.  out must hold 3 values. */
int miosqp_qp_get_inverse_guard(miosqp_qp_engine *e, double *out);

/* Times `reps` back-to-back launches of one hot-path kernel with HIP events on the engine's
 * own stream and returns the mean duration in microseconds in *usec and the kernel's
 * algorithmic bytes per launch in *bytes.  which: 0 panel-forward, 1 tail-forward,
 * 2 tail-backward, 3 panel-backward+update, 4 one whole ADMM iteration (all four); with the
 * product-form factor 0 = forward sweep, 1 = backward sweep + update, 2 and 3 are empty;
 * 10..14 the same for the batched kernels at full batch capacity (after a solve_batch). */
int miosqp_qp_time_kernel(miosqp_qp_engine *e, int32_t which, int32_t reps, double *usec,
                          double *bytes);

/* Device time spent in the ADMM loop since the last reset, measured with HIP events recorded on
 * the engine's stream around every chunk (check_termination iterations + one termination test)
 * of every solve: *ms = total milliseconds, *iters = ADMM iterations executed in them. */
int miosqp_qp_get_loop_stats(miosqp_qp_engine *e, double *ms, int64_t *iters, int32_t reset);

/* Per node of the node-at-a-time search in the host library (miosqp_qp_search_run) since the last reset of the loop
 * statistics: microseconds of device time per ADMM iteration of a node -- out[0] minimum, out[1] median, out[2] maximum
 * over the nodes -- and *nodes, how many nodes they are taken over (bench.py prints them next to the mean: a single slow
 * node cannot move the headline unnoticed).  No reference counterpart: the reference only sums OSQP's run_time
 * (/root/reference/miosqp/node.py:118). */
int miosqp_qp_get_node_stats(miosqp_qp_engine *e, double *us_per_iter_min_med_max, int32_t *nodes);

/* Launches of the hosted search's solver kernel since the last reset of the loop statistics.  A node of
 * miosqp_qp_search_run is one cooperative launch (k_coop) -- or, where the cooperative grid can stay resident, the whole
 * call is ONE launch (k_coop_run) that takes its nodes from a mailbox the host writes: the loop
 * `while can_continue: choose_leaf -> solve -> bound_and_branch` of /root/reference/miosqp/solver.py:85-123 with the
 * `solve` of /root/reference/miosqp/node.py:96-143 resident on the device between nodes.  bench.py divides the loop's
 * device time by this count for the kernel's average launch. */
int miosqp_qp_get_loop_launches(miosqp_qp_engine *e, int64_t *launches);

/* Same for solve_batch: *ms device milliseconds in batched chunks, *batch_iters lock-step
 * iterations executed, *node_iters = sum over those iterations of the columns still iterating. */
int miosqp_qp_get_batch_stats(miosqp_qp_engine *e, double *ms, int64_t *batch_iters,
                              int64_t *node_iters, int32_t reset);

/* debug counters: which = 0 -> wave compactions solve_batch has performed; 1 -> times a chunk's persistent launch was
 * called off; 2 -> whole-chip launches on this engine's device that were ordered behind another engine's (engines of
 * one process take turns with k_coop / k_pers / kbp); 3 -> the control block's call-off word once the engine's stream
 * is idle (blocks); 4 -> engines of this device that take turns; 5 -> columns of every row of the tail's inverse that
 * the persistent streaming solver keeps in LDS for a whole launch (0: none); 6 -> tiles of the tail's inverse the
 * persistent streaming solver reads per iteration when it takes it as a symmetric matrix (0: it reads whole rows);
 * 7 -> launches of the stream's persistent kernel (kbs) the leaf pool has queued, 8 -> chunks queued that way, 9 -> chunks
 * queued as the chunk graph / kernel by kernel instead (a launch of kbs that is called off leaves its chunks undone: 1);
 * 10 -> the poll delay the resident search grid ran with last (-1: it has not run), 11 -> synthetic nodes the calibration of
 * that delay has run on this engine (0: looked up) */
int64_t miosqp_qp_debug_counter(miosqp_qp_engine *e, int32_t which);

/* debug: per-workgroup (start, end) stamps (100 MHz wall clock) of ONE launch of a product-form
 * kernel (which: 0 forward, 1 backward); out holds 2 * max_blocks values */
int miosqp_qp_debug_timeline(miosqp_qp_engine *e, int32_t which, uint64_t *out, int32_t max_blocks,
                             int32_t *nblocks);

/* debug: shader cycles and 100 MHz ticks recorded by the last LDS-resident launch */
int miosqp_qp_debug_clock(miosqp_qp_engine *e, double *cycles, double *ticks);

/* ---- the host side of the streaming search, compiled ------------------------------------------------------
 * What miosqp_amd/stream.py: StreamSearch does per round -- push the next leaves in the exploration rule's order
 * (workspace.py:128-149), launch, collect the launch before, bound_and_branch on every digest (workspace.py:282-334),
 * prune -- as one call per `rounds` rounds on top of miosqp_qp_pool_*: the same exploration order and node counts,
 * no interpreter between two chunks (several pools on several host threads stay bound by the device).  The value of
 * an incumbent found by the rounding heuristic is the device's.  Needs set_integer_rows + set_root. */
typedef struct miosqp_stream_info {
  int64_t alive;        /* leaves open on the host, waiting in the ring or being solved (0: the tree is closed) */
  int64_t open_leaves, in_flight, free_slots;
  int64_t nodes, osqp_iter, chunks, dropped;   /* totals since the driver was created */
  int32_t improved;     /* 1: the incumbent improved during this call */
  int32_t active;       /* columns holding a node after the last refill seen */
  double upper_glob;
} miosqp_stream_info;

/* creates the pool if there is none; ring_margin 0 = max(32, columns / 2) */
int miosqp_qp_stream_create(miosqp_qp_engine *e, int32_t capacity, int32_t columns, int32_t ring_margin);
int miosqp_qp_stream_begin(miosqp_qp_engine *e);
int miosqp_qp_stream_add_leaf(miosqp_qp_engine *e, const double *l_int, const double *u_int, const double *x0,
                              const double *y0, int32_t depth, double lower);
int miosqp_qp_stream_take_leaf(miosqp_qp_engine *e, double *l_int, double *u_int, double *x0, double *y0, int32_t *depth,
                               double *lower);
int miosqp_qp_stream_set_incumbent(miosqp_qp_engine *e, double upper, const double *x);
int miosqp_qp_stream_get_incumbent(miosqp_qp_engine *e, double *upper, double *x);
int miosqp_qp_stream_step(miosqp_qp_engine *e, int32_t tree_explor_rule, int32_t chunks, int32_t rounds,
                          int64_t max_nodes, miosqp_stream_info *info);

#ifdef __cplusplus
}
#endif
#endif /* MIOSQP_AMD_H */
