// MI355X (gfx950) relaxation engine: device kernels + host driver + C ABI (include/miosqp_amd.h).
//
// One ADMM iteration of the OSQP algorithm (paper Algorithm 1) = one application of K^-1
// through the block factor of factor.hpp plus the fused vector updates, as FOUR row-parallel
// kernels (each row = one CSR / triangular row handled by a wavefront group: lanes stride the
// row with coalesced loads, partial sums meet in a DPP butterfly and, for multi-wave rows, in
// LDS).  Every reduction has a fixed order, so reruns are bit-identical.
//
//   k_panel_fwd   c  = sigma x - q - L21 wh                    (n rows, pattern of A^T)
//   k_tail_fwd    ut = D22^-1 (c + strict_lower(Linv) c)       (n rows, triangular)
//   k_tail_bwd    xt = ut + strict_upper(Linv^T) ut ;  x, dx   (n rows, triangular)
//   k_panel_bwd   nu = -rho wh - L21^T xt ; z~, z, y, dy, wh   (M rows, pattern of A)
//
// with wh = z - y/rho (the permuted right-hand side of the constraint block).  The termination
// test (every check_termination iterations) is three more kernels and one device-side decision;
// the host only reads a 64-byte control block per chunk.  The per-chunk kernel sequence is
// captured once in a hipGraph.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <queue>
#include <deque>
#include <tuple>
#include <type_traits>
#include <string>
#include <vector>

#include "../../include/miosqp_amd.h"
#include "factor.hpp"

#define QP_INFTY 1e30
#define QP_MIN_SCALING 1e-4
#define QP_DIVISION_TOL (1.0 / QP_INFTY)

// dense_setup.hip: dense LDL^T + triangular inverse on the device (miosqp::DenseLdlInv)
int miosqp_device_ldl_inverse(int n, int ld, const double *S, double *d, double *Linv, double *LinvT, void *ctx);
void miosqp_device_ruiz_ops(miosqp::RuizOps *ops, void **storage, void *stream);
void miosqp_device_ruiz_free(void *storage);
// dense_setup.hip: explicit KKT inverse W = F^T D22^-1 F from the product-form rows (device pointers)
int miosqp_device_kkt_inverse(const double *F, int ldf, const double *dinv, int n, int M, double *W, int ldw,
                              hipStream_t stream);

namespace {

thread_local std::string g_err;

void set_err(const char *what, hipError_t e, const char *file, int line) {
  char buf[512];
  snprintf(buf, sizeof buf, "%s: %s (%s:%d)", what, hipGetErrorString(e), file, line);
  g_err = buf;
}

#define HIPCHK(call)                                  \
  do {                                                \
    hipError_t e__ = (call);                          \
    if (e__ != hipSuccess) {                          \
      set_err(#call, e__, __FILE__, __LINE__);        \
      return MIOSQP_EHIP;                             \
    }                                                 \
  } while (0)

// The HIP current device is per host thread: every entry point makes the engine's device current
// first, so a handle can be driven from any thread (e.g. the worker thread of a pipelined wave).
#define ENTER(e)                                       \
  do {                                                 \
    if ((e)->device >= 0) HIPCHK(hipSetDevice((e)->device)); \
  } while (0)

struct Ctrl {
  int done, status, iter, pad;
  int B, ndone, pad3[2];  // batched mode: active columns, columns already decided
  double pri_res, dua_res, obj_val, lower;
  double nrm_dy, nrm_dx;  // certificate normalisers
  double pad2[2];
  // node digest (branching epilogue, workspace.py:245-272 on the device)
  int int_inf, nextvar, pad4[2];
  double heur_viol, heur_obj;
};

struct PoolCtl;
struct ReadyEntry;

// everything a kernel needs, passed by value
struct Dev {
  int n, M, ld, n_int, m_orig;
  double rho, rho_inv, sigma, alpha, eps_abs, eps_rel, eps_pinf, eps_dinf, c, cinv;
  // panel by variable (n rows) / by constraint (M rows)
  const int *pv_ptr, *pv_idx;
  const double *pv_L, *pv_At;
  const int *pc_ptr, *pc_idx;
  const double *pc_L, *pc_A;
  // tail
  const double *Linv, *LinvT, *d2inv;
  // symmetric matrices by row
  const int *pb_ptr, *pb_idx;
  const double *pb_val;
  const int *pr_ptr, *pr_idx;
  const double *pr_val;
  const double *D, *Dinv, *E, *Einv;
  const int *i_idx;
  // scaled vectors
  double *q, *l, *u, *x, *z, *y, *wh, *cv, *ut, *xt, *dx, *dy;
  double *qraw;
  // scratch for the termination test: sm = 8 x M, sn = 10 x n
  double *sm, *sn;
  // staging: raw (unscaled) inputs and outputs
  double *raw_l, *raw_u, *raw_x, *raw_y, *out_x, *out_y;
  Ctrl *ctrl;
  // ---- folded (product-form) factor: rows of L^-1, see factor.hpp ----
  const double *f_rows, *f_GmT, *f_Ad, *f_Atd, *f_Pd;
  int ldf, ldn, ldm;
  double *rx;  // sigma x - q, kept right behind wh so that [wh | rx] is one contiguous vector
  // node digest: rounded candidate (unscaled / scaled), root bounds, tolerances
  double *xi, *xis, *root_l, *root_u;
  double eps_int, eps_lin;
  int digest;
  int bm_ablate;  // debug: 1 = no operand loads, 2 = no matrix-core instructions
  unsigned long long *prof;  // debug timeline (per-block start/end, 100 MHz wall clock) or nullptr
  // ---- cooperative register-resident solver (k_coop): explicit KKT inverse, exchange buffers ----
  const double *W;               // N x ldw, N = M + n, ordering [constraints ; variables]
  int ldw;
  unsigned *coop_tag;            // tag of the last exchange round that completed
  unsigned long long *coop_buf;  // 2 parities x N x {lo32|tag, hi32|tag}
  unsigned long long *coop_chk;  // 2 x coop_half: the test's operands [y ; x] and [proj(dy) ; dx]
  unsigned long long *coop_q;    // (256 + 16) x COOP_QS: per-workgroup, then per-group norms / sums of the test
  unsigned long long *coop_reg;  // start-up registration counter (grows by the grid size per launch); +64: launches with testers
  unsigned long long *coop_chz;  // coop_half: z, the third operand of the test when it runs on tester workgroups
  unsigned long long *coop_dec;  // the testers' decision {status | tag}
  int coop_nt;                   // tester workgroups behind the exchange grid (0: the test runs inside the grid)
  int coop_lag;                  // iterations between a test and the point where the grid waits for its decision
  int coop_tres;                 // the testers keep their rows of Kc in registers (kernels_coop.inc: coop_tester<.., RES>)
  const double *Kc;              // [ 0 Abar ; Abar^T Pbar ], N x ldw
  int coop_dbg;                  // debug: 1 = no gather (ablation), 64 = workgroup 1 never starts (fault injection)
  int coop_nap;                  // 64-clock naps between publishing and the first poll of a round (calibrated)
  int coop_stride;               // 8-byte words between the blocks of consecutive workgroups (= 2 RW: entries are flat, 2 words each)
  size_t coop_half;              // words per parity
  // the exchange's row layout (kernels_coop.inc "reduced indices"): W' = W without the identity (integer-bound) rows
  const double *coop_W;          // NR x coop_ldw, NR = coop_mg + n  (== W, ldw while nothing is dropped)
  int coop_ldw, coop_mg;         // coop_mg: general rows kept in W' (M while nothing is dropped)
  const int *coop_brow;          // per variable: FULL index of its bound row, -1 without one (nullptr: none anywhere)
  const double *coop_aj;         // per variable: scaled entry of its bound row, 0 without one
  // ---- batched mode: B nodes share the factor; vectors are [len][Bs], batch index fastest ----
  int Bs;  // column stride, multiple of 64
  double *b_l, *b_u, *b_x, *b_z, *b_y, *b_wh, *b_rx, *b_cv, *b_ut, *b_xt, *b_dx, *b_dy;
  double *b_xfs;  // harvest tile: D^-1 (clamped x), next to b_xis = D^-1 (rounded candidate)
  double *b_sm, *b_sn;      // 8 x M x Bs, 4 x n x Bs
  double *b_xfin, *b_yfin;  // unscaled answers, batch-fastest
  double *b_xi, *b_xis;     // rounded candidates (node digest), unscaled / scaled
  double *b_part;           // partial reductions of the batched termination test
  const double *b_zero;     // a page of zeros (matrix-core tiles: operand of a chunk beyond the end)
  // ---- leaf pool + streaming batch (kernels_pool.inc) ----
  int stream;               // 1: columns are refilled from the pool between chunks and count their own iterations
  int max_iter_s, ring;
  int *c_start, *c_child0, *c_child1, *c_harv, *t_has;
  int *fmap, *hmap;         // columns refilled / harvested in the current chunk
  double *lt_x;             // load tile: scaled x of the refilled columns, [n][64]
  const int *int_pos;       // variable -> its position in i_idx, or -1
  // ---- integer-bound rows taken out of the batched products (kernels_batched.inc "identity rows") ----
  int wh_m;                 // rows of the constraint block that go through the matrix products (m, or M when off)
  const double *f_rows2;    // [ -G(general rows) | strict_lower(Linv) ], n x ldf2 (== f_rows, ldf when off)
  int ldf2;
  const double *a_int;      // per variable: rho * (scaled entry of its bound row), 0 for continuous variables
  double *pl_lo, *pl_hi;    // [cap][n_int] integer-row bounds of every node
  int *pl_ws;               // slot whose solution is the node's warm start
  double *sol_x, *sol_y;    // [cap][n], [cap][M] solutions of solved nodes (x clamped as node.py:131-136)
  PoolCtl *pctl;            // device-side counters
  PoolCtl *hctl;            // host memory: [0] host -> device (tail, upper), [1] device -> host (head, fin, active)
  ReadyEntry *ready;        // host memory: ready ring
  miosqp_pool_digest *dg_ring;  // host memory: digests of decided nodes
  int *c_intinf, *c_nextvar;
  int *c_node;   // column position -> node of the wave (columns are swapped when the wave is compacted)
  int *c_pairs;  // swap list of the current compaction
  double *c_hviol, *c_hobj;
  double *b_raw;            // node-major staging in:  l[B][M] | u[B][M] | x0[B][n] | y0[B][M]
  double *b_out;            // node-major staging out: x[B][n] | y[B][M]
  int *c_done, *c_status, *c_iter;
  double *c_pri, *c_dua, *c_obj, *c_lower;
};

struct SearchDigest {  // coherent host memory
  int status, iter, int_inf, nextvar, action, branched, pad;
  int todo;  // 1: the launch only ran the relaxation (or was called off): the host still has to launch the epilogue kernels
  double lower, heur_viol, heur_obj, pri_res, dua_res, obj_val;
  unsigned long long t0, t1;  // device wall clock (100 MHz): the first tester took the node up / wrote this record (0: not stamped)
  unsigned long long seq;  // written last (release, system scope): the record of node `seq` is complete
};
struct SearchArgs {
  const double *lo_s, *hi_s;  // this node's integer-row bounds
  const double *x_w, *y_w;    // its warm start: the parent's solution
  double *x_s, *y_s;          // where its own solution goes
  double *lo0, *hi0, *lo1, *hi1;  // the children's bounds (slots reserved by the host)
  double *inc_x;              // the incumbent
  double upper;               // the incumbent's value as the host knows it
  SearchDigest *dg;
  unsigned long long seq;
};

// node mode of k_coop (hosted search, host_search.inc): prologue and epilogue of a node inside the cooperative launch
struct CoopNode {
  int on;            // 0: plain solve -- iterates in d.*, prologue / epilogue by the host's kernels
  int epi;           // 1: the tester workgroups run the epilogue (unscale, clamp, digest, heuristic, objective, commit)
  int tpr_h, tpr_o;  // threads per row of the sparse rows of Abar / of the unscaled P, as the launches use them
  SearchArgs a;
  unsigned long long *epi_buf;  // exchange buffer of the epilogue: 2 n + (workgroups of the launch) tagged entries
  unsigned long long *stamps;   // debug (MIOSQP_SEARCH_STAMPS=1): 16 wall-clock stamps (100 MHz) of this node, or nullptr
};

// The device code and the host driver live in the .inc files below: ONE translation unit (everything
// sits in the same anonymous namespace and the kernels are templates instantiated by the host code),
// split only for reading.
#include "kernels_iter.inc"  // device helpers; the four kernels of the factor form and the two of the product form
#include "kernels_test.inc"  // termination test of the multi-kernel forms (k_check_*), Norms / decide_status shared by all forms
#include "kernels_coop.inc"  // cooperative register-resident solver (k_coop) and its in-kernel termination test
#include "kernels_guard.inc"  // residual check of the explicit KKT inverse at set-up (falls back to the sweeps when it fails)
#include "kernels_pers.inc"  // persistent streaming solver (k_pers): one launch per solve, the factor read from memory every iteration
#include "kernels_resident.inc"  // LDS-resident single-workgroup solver (k_resident)
#include "kernels_node.inc"  // per-solve prologue / epilogue kernels (scaling, warm start, finish, node digest, objective)
#include "kernels_batched.inc"  // batched mode: sparse row kernels, dense vector-FMA tiles, fp64 matrix-core tiles, batched test
#include "kernels_bpers.inc"  // batched mode: a chunk's lock-step iterations as one persistent launch, the factor in registers and LDS (kbp)
#include "kernels_tree.inc"  // a whole branch-and-bound tree in one launch (LDS-resident problems)
#include "kernels_pool.inc"  // device-resident leaf pool, streaming batch (refill / harvest between chunks)
#include "kernels_bstream.inc"  // the streaming batch as ONE persistent launch: iterations, test, harvest and refill per column group (kbs)
#include "host.inc"  // host side: engine object, allocation, launches, graph capture, solve loops
#include "host_pool.inc"  // host side of the leaf pool (C ABI miosqp_qp_pool_*)
#include "host_search.inc"  // node-at-a-time branch and bound driven from the host in C++ (C ABI miosqp_qp_search_*)
#include "host_stream.inc"  // the host side of the streaming search in C++ (C ABI miosqp_qp_stream_*)

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
namespace {
// the dynamic-LDS ceiling of a kernel is a property of (process, device, kernel): raised to the chip's 160 KB once instead of
// at every set-up (the call costs ~40 us, a fifth of a small problem's set-up)
hipError_t lds_limit_once(const void *fn, int which) {
  static bool done[12][64] = {};
  int dev = 0;
  hipError_t rc = hipGetDevice(&dev);
  if (rc != hipSuccess) return rc;
  if (dev >= 0 && dev < 64 && done[which][dev]) return hipSuccess;
  rc = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (rc == hipSuccess && dev >= 0 && dev < 64) done[which][dev] = true;
  return rc;
}
}  // namespace

extern "C" {

const char *miosqp_qp_last_error(void) { return g_err.c_str(); }

int miosqp_qp_default_settings(miosqp_qp_settings *s) {
  if (!s) return MIOSQP_EARG;
  memset(s, 0, sizeof *s);
  s->rho = 0.1;
  s->sigma = 1e-6;
  s->alpha = 1.6;
  s->eps_abs = 1e-3;
  s->eps_rel = 1e-3;
  s->eps_prim_inf = 1e-4;
  s->eps_dual_inf = 1e-4;
  s->max_iter = 4000;
  s->scaling = 10;
  s->check_termination = 25;
  s->warm_start = 1;
  s->device = -1;
  s->max_batch = 1;
  s->fold = -1;
  s->resident = -1;
  s->setup_on_device = -1;
  s->coop = -1;
  s->pers = -1;
  s->batch_pers = -1;
  return 0;
}

int miosqp_qp_constant(const char *name) {
  if (!name) return 0;
  if (!strcmp(name, "OSQP_SOLVED")) return MIOSQP_QP_SOLVED;
  if (!strcmp(name, "OSQP_MAX_ITER_REACHED")) return MIOSQP_QP_MAX_ITER_REACHED;
  if (!strcmp(name, "OSQP_PRIMAL_INFEASIBLE")) return MIOSQP_QP_PRIMAL_INFEASIBLE;
  if (!strcmp(name, "OSQP_DUAL_INFEASIBLE")) return MIOSQP_QP_DUAL_INFEASIBLE;
  if (!strcmp(name, "OSQP_UNSOLVED")) return MIOSQP_QP_UNSOLVED;
  return 0;
}

int miosqp_qp_cleanup(miosqp_qp_engine *e) {
  if (!e) return 0;
  if (e->device >= 0) (void)hipSetDevice(e->device);
  kbs_stop(e);  // (a persistent launch of the stream still queued leaves at its next chunk boundary)
  if (e->stream) hipStreamSynchronize(e->stream);
  if (e->side) {
    hipStreamSynchronize(e->side);
    hipStreamDestroy(e->side);
    e->side = nullptr;
  }
  chip_quit(e);
  if (e->x_full) hipGraphExecDestroy(e->x_full);
  if (e->x_tail) hipGraphExecDestroy(e->x_tail);
  if (e->g_full) hipGraphDestroy(e->g_full);
  if (e->g_tail) hipGraphDestroy(e->g_tail);
  {
    // small pool chunks are parked for the next engine of the process, zero-filled HERE (outside anybody's set-up time):
    // fill on the engine's stream, wait, then offer them
    std::vector<ParkedChunk> park;
    for (void *p : e->allocs) {
      bool keep = false;
      for (const ParkedChunk &c : e->chunks)
        if (c.p == p && c.bytes <= CHUNK_PARK_MAX && park.size() < 8 && !getenv("MIOSQP_NO_CACHE")) {
          keep = hipMemsetAsync(c.p, 0, c.bytes, e->stream) == hipSuccess;
          if (keep) park.push_back(c);
          break;
        }
      if (!keep) hipFree(p);
    }
    if (!park.empty()) {
      const bool filled = hipStreamSynchronize(e->stream) == hipSuccess;
      std::lock_guard<std::mutex> lk(g_rt_mutex);
      for (const ParkedChunk &c : park) {
        if (filled && g_chunk_cache.size() < 8) g_chunk_cache.push_back(c);
        else hipFree(c.p);
      }
    }
  }
  drop_stream_graph(e);
  for (hipEvent_t ev : e->ev_pool)
    if (ev) hipEventDestroy(ev);
  if (e->sdriver) delete static_cast<StreamDriver *>(e->sdriver);
  if (e->search) {
    NodeSearch *S = static_cast<NodeSearch *>(e->search);
    if (S->dg) hipHostFree(S->dg);
    if (S->own_block) hipFree(S->own_block);
    for (hipEvent_t ev : S->ev)
      if (ev) hipEventDestroy(ev);
    for (hipEvent_t ev : S->ev_run)
      if (ev) hipEventDestroy(ev);
    if (S->mail_host) hipHostFree(S->mail_host);
    delete S;
  }
  if (e->h_ready) hipHostFree(e->h_ready);
  if (e->h_dg) hipHostFree(e->h_dg);
  if (e->h_pctl) hipHostFree(e->h_pctl);
  if (e->hb_in) hipHostFree(e->hb_in);
  if (e->hb_out) hipHostFree(e->hb_out);
  if (e->hb_int) hipHostFree(e->hb_int);
  if (e->hb_dbl) hipHostFree(e->hb_dbl);
  for (int k = 0; k < 16; k++) {
    if (e->xb_full[k]) hipGraphExecDestroy(e->xb_full[k]);
    if (e->xb_tail[k]) hipGraphExecDestroy(e->xb_tail[k]);
    if (e->gb_full[k]) hipGraphDestroy(e->gb_full[k]);
    if (e->gb_tail[k]) hipGraphDestroy(e->gb_tail[k]);
  }
  rt_release(e->rt);  // stream, events and pinned staging go back to the per-process cache
  delete e;
  return 0;
}

static int stage_wait(miosqp_qp_engine *e, int k);
static int stage_mark(miosqp_qp_engine *e, int k);

// rho chosen once per problem (settings.rho_auto): OSQP's update rule on the scaled iterates after RHO_ONCE_ITERS
// iterations from zero, rounded to two significant digits -- loop for loop the arithmetic the CPU restatement under the
// tests uses for the same setting (its rho_estimate), on the host's copy of the scaled matrices (the iterates come from
// the device).
constexpr int RHO_ONCE_ITERS = 50;
static double rho_round2(double r) {
  if (!(r > 0)) return r;
  const double e = floor(log10(r)), p = pow(10.0, e);
  return floor(r / p * 10.0 + 0.5) / 10.0 * p;
}
static double rho_estimate_host(const miosqp::Scaled &sc, double rho, const double *x, const double *z, const double *y) {
  const int n = sc.n, M = sc.M;
  std::vector<double> Ax(M, 0.0), Px(n, 0.0), Aty(n, 0.0);
  for (int j = 0; j < n; j++)
    for (int p = sc.Ap[j]; p < sc.Ap[j + 1]; p++) Ax[sc.Ai[p]] += sc.Ax[p] * x[j];
  for (int j = 0; j < n; j++)
    for (int p = sc.Pp[j]; p < sc.Pp[j + 1]; p++) {
      const int i = sc.Pi[p];
      Px[i] += sc.Px[p] * x[j];
      if (i != j) Px[j] += sc.Px[p] * x[i];
    }
  for (int j = 0; j < n; j++) {
    double t = 0;
    for (int p = sc.Ap[j]; p < sc.Ap[j + 1]; p++) t += sc.Ax[p] * y[sc.Ai[p]];
    Aty[j] = t;
  }
  auto ninf = [](const double *v, int k) { double r = 0; for (int i = 0; i < k; i++) r = fmax(r, fabs(v[i])); return r; };
  double pri = 0, dua = 0;
  for (int i = 0; i < M; i++) pri = fmax(pri, fabs(Ax[i] - z[i]));
  for (int j = 0; j < n; j++) dua = fmax(dua, fabs(Px[j] + sc.q[j] + Aty[j]));
  const double pn = fmax(ninf(z, M), ninf(Ax.data(), M));
  const double dn = fmax(ninf(sc.q.data(), n), fmax(ninf(Aty.data(), n), ninf(Px.data(), n)));
  pri /= (pn + 1e-10);
  dua /= (dn + 1e-10);
  double r = rho * sqrt(pri / (dua + 1e-10));
  r = fmin(fmax(r, 1e-6), 1e6);
  return rho_round2(r);
}

int miosqp_qp_setup(miosqp_qp_engine **out, int32_t n, int32_t M, const int32_t *Pp, const int32_t *Pi,
                    const double *Px, const int32_t *Ap, const int32_t *Ai, const double *Ax,
                    const double *q, const double *l, const double *u, const miosqp_qp_settings *s_in) {
  if (!out || n <= 0 || M < 0 || !Pp || !Ap || !q || !s_in || (M > 0 && (!l || !u))) {
    g_err = "setup: bad argument";
    return MIOSQP_EARG;
  }
  miosqp_qp_settings s_own = *s_in;
  miosqp_qp_settings *s = &s_own;
  // rho chosen at set-up (rho_auto), r05: ONE set-up.  The probing iterations run on THIS engine, in the plain four-kernel
  // factor form, the moment its factor at the starting rho is on the device (before the product form, the explicit
  // inverse, graphs and calibrations exist); then only what depends on rho is built again -- the factor, on the host --
  // and copied over the first one.  Measured at config 2: 153 ms (two set-ups) -> see DESIGN.md; the equilibration, the
  // allocations, the runtime objects, the product form, the inverse, captures and calibrations happen once.
  // Kept on the two-set-up path: problems factorised on the device (n >= 1024: the factor never visits the host) and
  // the small ones whose probe engine is the one-workgroup solver (fifty iterations there cost less than 200 launches).
  bool probe_inline = false;
  if (s->rho_auto && M > 0) {
    int on_dev_pre = s->setup_on_device;
    if (on_dev_pre < 0) on_dev_pre = n >= 1024 ? 1 : 0;
    probe_inline = !on_dev_pre && n + M > 400 && !getenv("MIOSQP_RHO_TWO_SETUPS");
    if (probe_inline) s->rho_auto = 0;
  }
  if (s->rho_auto && M > 0) {
    // a throw-away engine in the plain multi-kernel form for the probing iterations, then the real one at the chosen
    // rho (the equilibration does not depend on rho and is simply repeated: set-up runs once per MIQP)
    miosqp_qp_settings s1 = *s;
    s1.rho_auto = 0;
    miosqp_qp_settings sp = s1;
    sp.coop = 0;
    sp.pers = 0;
    sp.max_batch = 1;
    miosqp_qp_engine *probe = nullptr;
    int rc = miosqp_qp_setup(&probe, n, M, Pp, Pi, Px, Ap, Ai, Ax, q, l, u, &sp);
    if (rc) return rc;
    std::vector<double> xs(n), zs(M), ys(M);
    {
      const int big = n > M ? n : M;
      hipLaunchKernelGGL(k_zero_iterates, dim3((big + 255) / 256), dim3(256), 0, probe->stream, probe->d);
    }
    rc = miosqp_qp_debug_iterate(probe, RHO_ONCE_ITERS, xs.data(), zs.data(), ys.data());
    double rho_new = s1.rho;
    if (!rc) rho_new = rho_estimate_host(probe->sc, s1.rho, xs.data(), zs.data(), ys.data());
    miosqp_qp_cleanup(probe);
    if (rc) return rc;
    if (rho_new > 0) s1.rho = rho_new;
    rc = miosqp_qp_setup(out, n, M, Pp, Pi, Px, Ap, Ai, Ax, q, l, u, &s1);
    if (!rc) (*out)->st.rho_auto = 1;
    return rc;
  }
  if (M == 0) {
    // miOSQP always has rows: data.py:5-33 appends one identity row per integer variable
    g_err = "setup: the engine needs at least one constraint row (M >= 1)";
    return MIOSQP_EARG;
  }
  for (int i = 0; i < M; i++)
    if (l[i] > u[i]) {
      g_err = "setup: lower bound above upper bound";
      return MIOSQP_EARG;
    }
  if (!(s->rho > 0) || !(s->sigma > 0) || !(s->alpha > 0 && s->alpha < 2) || s->max_iter <= 0) {
    g_err = "setup: settings out of range";
    return MIOSQP_EARG;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    g_err = "no HIP device visible: the relaxation engine has no CPU fallback";
    return MIOSQP_ENODEV;
  }
  if (s->device >= 0) HIPCHK(hipSetDevice(s->device));
  miosqp_qp_engine *e = new miosqp_qp_engine();
  HIPCHK(hipGetDevice(&e->device));
  const bool t_on = getenv("MIOSQP_SETUP_TIMING") != nullptr;
  double t_last = wall();
  const double t_begin = t_last;
  auto tick = [&](const char *what) {
    if (!t_on) return;
    const double now = wall();
    fprintf(stderr, "[miosqp setup] %-32s %8.3f ms\n", what, 1e3 * (now - t_last));
    t_last = now;
  };
  e->n = n;
  e->M = M;
  e->st = *s;
  if (e->st.check_termination <= 0 || e->st.check_termination > e->st.max_iter)
    e->st.check_termination = e->st.max_iter;
  std::string err;
  // the array-sized parts of the setup on the device for large problems: the equilibration's
  // maxima and products, the Schur complement, the dense LDL^T and its triangular inverse
  int on_dev = s->setup_on_device;
  if (on_dev < 0) on_dev = n >= 1024 ? 1 : 0;
  e->setup_on_device = on_dev != 0;
  {
    bool done = false;
    if (on_dev && s->scaling > 0 && !getenv("MIOSQP_SETUP_HOST_RUIZ")) {
      miosqp::RuizOps ops;
      void *storage = nullptr;
      miosqp_device_ruiz_ops(&ops, &storage, nullptr);
      done = miosqp::scale_problem(n, M, Pp, Pi, Px, Ap, Ai, Ax, q, s->scaling, e->sc, &ops);
      miosqp_device_ruiz_free(storage);
      if (!done) fprintf(stderr, "miosqp: equilibration on the device failed, redone on the host\n");
    }
    if (!done) miosqp::scale_problem(n, M, Pp, Pi, Px, Ap, Ai, Ax, q, s->scaling, e->sc);
  }
  // Will the product form be built (on the host, from Linv)?  Decided here because the device stage of the setup may
  // then leave Linv / LinvT on the device instead of sending them down and up again.
  int want_fold = s->fold;
  bool coop_pref = false;
  {
    const double dens = M > 0 ? (double)Ap[n] / ((double)n * M) : 0.0;
    const double fold_bytes = 8.0 * ((double)n * (M + n) + (double)M * n);
    const bool res_w_on = !(getenv("MIOSQP_RES_W") && atoi(getenv("MIOSQP_RES_W")) == 0);
    const bool fits_lds = resident_lds_doubles(n, M, res_w_on && M > 0 && n + M <= RES_W_MAX) * sizeof(double) <= 150 * 1024;
    // the cooperative solver only needs the product form to build its explicit inverse from, whatever
    // the sparsity; it is preferred from n + M = 64 on (measured: equal to the LDS-resident workgroup
    // below 100, 1.8x at 160, 2.2x at 240)
    int coop_req = s->coop;
    if (const char *ev = getenv("MIOSQP_COOP")) coop_req = atoi(ev);
    const bool coop_fits = M > 0 && n + M <= 2048;
    // ... and from n + M = 193 on since the LDS-resident workgroup runs its loop on the explicit inverse in registers
    // (res_admm_w: 0.9-1.4 us per iteration at n + M = 85-190 against 2.1-2.2 for the cooperative grid)
    const bool res_w_here = res_w_on && M > 0 && n + M <= RES_W_MAX && s->resident != 0;
    coop_pref = coop_fits && (coop_req == 1 || (coop_req < 0 && n + M >= 64 && !res_w_here)) && s->resident != 1;
    if (want_fold < 0)
      want_fold = (M > 0 && ((dens >= 0.30 && fold_bytes <= 4.0e9) || (fits_lds && s->resident != 0) || coop_pref)) ? 1 : 0;
  }
  miosqp::DenseAccelCtx actx;
  actx.keep_on_device = on_dev && !(want_fold && M > 0) && !getenv("MIOSQP_SETUP_ROUNDTRIP");
  actx.schur_on_device = on_dev && M > 0 && !getenv("MIOSQP_SETUP_HOST_SCHUR");
  if (!miosqp::build_factor(e->sc, Pp, Pi, Px, s->rho, s->sigma, e->fa, err,
                            on_dev ? miosqp_device_ldl_inverse : nullptr, &actx)) {
    g_err = err;
    if (actx.dLinv) hipFree(actx.dLinv);
    if (actx.dLinvT) hipFree(actx.dLinvT);
    delete e;
    return MIOSQP_EFACTOR;
  }
  if (actx.dLinv) e->allocs.push_back(actx.dLinv);    // freed with the engine
  if (actx.dLinvT) e->allocs.push_back(actx.dLinvT);
  tick("host: scaling + factor (total)");
  e->nnzA = Ap[n];
  e->nnzPtriu = (int64_t)e->sc.Pi.size();
  const miosqp::Factor &f = e->fa;
  Dev &d = e->d;
  d.n = n; d.M = M; d.ld = f.ld; d.n_int = 0; d.m_orig = M; d.wh_m = M;
  d.rho = s->rho; d.rho_inv = 1.0 / s->rho; d.sigma = s->sigma; d.alpha = s->alpha; d.eps_abs = s->eps_abs; d.eps_rel = s->eps_rel;
  d.eps_pinf = s->eps_prim_inf; d.eps_dinf = s->eps_dual_inf; d.c = e->sc.c; d.cinv = e->sc.cinv;
  {
    // one chunk for everything this setup will place on the device
    const size_t np2 = f.panel_by_var.idx.size() + f.panel_by_con.idx.size();
    size_t est = np2 * (4 + 16) + 2 * f.Linv.size() * 8 + (f.Pbar.idx.size() + f.Praw.idx.size()) * 12 +
                 ((size_t)60 * n + (size_t)60 * M) * 8 + ((size_t)n * ((M + n + 8)) + (size_t)M * (n + 8)) * 8 +
                 200 * 512 + ((size_t)1 << 16);
    int rc0 = pool_reserve(e, est);
    if (rc0) { miosqp_qp_cleanup(e); return rc0; }
  }
  tick("device pool: malloc + zero fill");
#define UP(vec, field)                                   \
  do {                                                   \
    int rc__ = dupload(e, vec, &d.field);                \
    if (rc__) { miosqp_qp_cleanup(e); return rc__; }     \
  } while (0)
  upload_begin(e);
  UP(f.panel_by_var.ptr, pv_ptr); UP(f.panel_by_var.idx, pv_idx); UP(f.panel_by_var.val, pv_L); UP(f.At_val, pv_At);
  UP(f.panel_by_con.ptr, pc_ptr); UP(f.panel_by_con.idx, pc_idx); UP(f.panel_by_con.val, pc_L); UP(f.A_val, pc_A);
  if (actx.dLinv) {  // computed on the device and left there
    d.Linv = actx.dLinv;
    d.LinvT = actx.dLinvT;
  } else {
    UP(f.Linv, Linv); UP(f.LinvT, LinvT);
  }
  UP(f.d2inv, d2inv);
  UP(f.Pbar.ptr, pb_ptr); UP(f.Pbar.idx, pb_idx); UP(f.Pbar.val, pb_val);
  UP(f.Praw.ptr, pr_ptr); UP(f.Praw.idx, pr_idx); UP(f.Praw.val, pr_val);
  UP(e->sc.D, D); UP(e->sc.Dinv, Dinv); UP(e->sc.E, E); UP(e->sc.Einv, Einv);
#undef UP
  {
    int rcu = upload_end(e);
    if (rcu) { miosqp_qp_cleanup(e); return rcu; }
  }
#define AL(field, count)                                 \
  do {                                                   \
    int rc__ = dalloc(e, &d.field, (size_t)(count));     \
    if (rc__) { miosqp_qp_cleanup(e); return rc__; }     \
  } while (0)
  AL(q, n); AL(qraw, n); AL(l, M); AL(u, M); AL(x, n); AL(z, M); AL(y, M); AL(wh, (size_t)M + n); AL(cv, n); AL(ut, n);
  AL(xt, n); AL(dx, n); AL(dy, M); AL(sm, 8 * (size_t)M); AL(sn, 10 * (size_t)n);
  d.rx = d.wh + M;
  AL(xi, n); AL(xis, n); AL(root_l, M); AL(root_u, M);
  d.digest = 0;
  AL(ctrl, 1);
  // staging block: raw_l | raw_u | raw_x | raw_y contiguous, out_x | out_y contiguous
  AL(raw_l, 2 * (size_t)M + n + M);
  d.raw_u = d.raw_l + M; d.raw_x = d.raw_u + M; d.raw_y = d.raw_x + n;
  AL(out_x, (size_t)n + M);
  d.out_y = d.out_x + n;
  {
    int *ii = nullptr;
    int rc = dalloc(e, &ii, (size_t)n);
    if (rc) { miosqp_qp_cleanup(e); return rc; }
    d.i_idx = ii;
  }
#undef AL
  tick("uploads (factor, matrices)");
  {
    int rcb = rt_acquire(e->device, 2 * (size_t)M + n + M + 1, (size_t)n + M + 1, e->rt);
    if (rcb) { miosqp_qp_cleanup(e); return rcb; }
    e->stream = e->rt.stream;
    e->ev0 = e->rt.ev0; e->ev1 = e->rt.ev1; e->evc0 = e->rt.evc0; e->evc1 = e->rt.evc1;
    e->ev_chunk[0] = e->rt.ev_chunk[0]; e->ev_chunk[1] = e->rt.ev_chunk[1];
    e->h_in = e->rt.h_in; e->h_out = e->rt.h_out; e->h_ctrl = e->rt.h_ctrl; e->h_ctrl2 = e->rt.h_ctrl2;
  }
  tick("stream, events, pinned buffers");
  // scaled q, raw q, scaled bounds
  if (e->rt.in_cap >= 2 * (size_t)M + 2 * (size_t)n) {
    // through the pinned staging block on the engine's stream (four blocking copies from pageable memory were ~40 us of
    // a small problem's set-up): [ l | u ] in region 0, [ scaled q | q ] in region 1, marked busy until the copies are out
    double *h = e->h_in;
    memcpy(h, l, sizeof(double) * M);
    memcpy(h + M, u, sizeof(double) * M);
    memcpy(h + 2 * (size_t)M, e->sc.q.data(), sizeof(double) * n);
    memcpy(h + 2 * (size_t)M + n, q, sizeof(double) * n);
    if (M > 0) {
      HIPCHK(hipMemcpyAsync(d.raw_l, h, sizeof(double) * M, hipMemcpyHostToDevice, e->stream));
      HIPCHK(hipMemcpyAsync(d.raw_u, h + M, sizeof(double) * M, hipMemcpyHostToDevice, e->stream));
    }
    HIPCHK(hipMemcpyAsync(d.q, h + 2 * (size_t)M, sizeof(double) * n, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(d.qraw, h + 2 * (size_t)M + n, sizeof(double) * n, hipMemcpyHostToDevice, e->stream));
    if (int rcm = stage_mark(e, 0)) { miosqp_qp_cleanup(e); return rcm; }
    if (int rcm = stage_mark(e, 1)) { miosqp_qp_cleanup(e); return rcm; }
  } else {
    HIPCHK(hipMemcpy(d.q, e->sc.q.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d.qraw, q, sizeof(double) * n, hipMemcpyHostToDevice));
    if (M > 0) {
      HIPCHK(hipMemcpy(d.raw_l, l, sizeof(double) * M, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(d.raw_u, u, sizeof(double) * M, hipMemcpyHostToDevice));
    }
  }
  if (M > 0) hipLaunchKernelGGL(k_scale_bounds, dim3((M + 255) / 256), dim3(256), 0, e->stream, d);
  hipLaunchKernelGGL(k_reset_ctrl, dim3(1), dim3(1), 0, e->stream, d);
  e->tpr_pv = pick_tpr((double)f.nnz_panel / n);
  e->tpr_pc = pick_tpr(M > 0 ? (double)f.nnz_panel / M : 1.0);
  e->tpr_tail = pick_tpr(0.5 * n);
  e->tpr_pb = pick_tpr((double)f.Pbar.nnz / n);
  e->tpr_pr = pick_tpr((double)f.Praw.nnz / n);
  if (probe_inline) {
    // the rule's recipe: RHO_ONCE_ITERS iterations from zero on the set-up's bounds, the rule, a new factor
    std::vector<double> xs(n), zs(M), ys(M);
    {
      const int big = n > M ? n : M;
      hipLaunchKernelGGL(k_zero_iterates, dim3((big + 255) / 256), dim3(256), 0, e->stream, d);
    }
    int rc = miosqp_qp_debug_iterate(e, RHO_ONCE_ITERS, xs.data(), zs.data(), ys.data());
    if (rc) { miosqp_qp_cleanup(e); return rc; }
    double rho_new = rho_estimate_host(e->sc, s->rho, xs.data(), zs.data(), ys.data());
    tick("rho chosen at set-up: probing iterations + the rule");
    if (rho_new > 0 && rho_new != s->rho) {
      std::string err2;
      // (the dense part on the device -- a third of the host's time at config 2 --, the result back on the host for the
      //  product form's rows; MIOSQP_RHO_HOST_REFACTOR=1 keeps it on the host)
      miosqp::DenseAccelCtx actx2;
      actx2.stream = e->stream;
      const bool dev2 = !getenv("MIOSQP_RHO_HOST_REFACTOR");
      if (!miosqp::build_factor(e->sc, Pp, Pi, Px, rho_new, s->sigma, e->fa, err2, dev2 ? miosqp_device_ldl_inverse : nullptr,
                                dev2 ? &actx2 : nullptr, true)) {
        // the factor at the chosen rho failed (a pivot, the device): the factor is rebuilt in place, so the one at the
        // starting rho -- which exists: the probing iterations ran on it -- is built again and the choice is dropped
        fprintf(stderr, "miosqp: rho chosen at set-up (%g): the factor failed (%s); keeping rho = %g\n", rho_new, err2.c_str(), s->rho);
        rho_new = s->rho;
        std::string err3;
        if (!miosqp::build_factor(e->sc, Pp, Pi, Px, rho_new, s->sigma, e->fa, err3, nullptr, nullptr, true)) {
          g_err = err3;
          miosqp_qp_cleanup(e);
          return MIOSQP_EFACTOR;
        }
      }
      const miosqp::Factor &f2 = e->fa;  // same patterns, same sizes: copied over the first factor
      HIPCHK(hipMemcpy(const_cast<double *>(d.pv_L), f2.panel_by_var.val.data(), sizeof(double) * f2.panel_by_var.val.size(), hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(const_cast<double *>(d.pc_L), f2.panel_by_con.val.data(), sizeof(double) * f2.panel_by_con.val.size(), hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(const_cast<double *>(d.Linv), f2.Linv.data(), sizeof(double) * f2.Linv.size(), hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(const_cast<double *>(d.LinvT), f2.LinvT.data(), sizeof(double) * f2.LinvT.size(), hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(const_cast<double *>(d.d2inv), f2.d2inv.data(), sizeof(double) * f2.d2inv.size(), hipMemcpyHostToDevice));
      s->rho = rho_new;
      e->st.rho = rho_new;
      d.rho = rho_new;
      d.rho_inv = 1.0 / rho_new;
      tick("rho chosen at set-up: the factor again");
    }
    e->st.rho_auto = 1;
    {
      const int big = n > M ? n : M;
      hipLaunchKernelGGL(k_zero_iterates, dim3((big + 255) / 256), dim3(256), 0, e->stream, d);
    }
    hipLaunchKernelGGL(k_reset_ctrl, dim3(1), dim3(1), 0, e->stream, d);
  }
  {
    // product-form factor: worth it when the panel is dense (bytes no worse, half the launches); decided above
    const int want = want_fold;
    if (want && M > 0) {
      miosqp::build_folded(f, e->fo);
      upload_begin(e);
      int rc = dupload(e, e->fo.rows, &d.f_rows);
      if (!rc) rc = dupload(e, e->fo.GmT, &d.f_GmT);
      if (!rc) rc = dupload(e, e->fo.Ad, &d.f_Ad);
      if (!rc) rc = dupload(e, e->fo.Atd, &d.f_Atd);
      if (!rc) rc = dupload(e, e->fo.Pd, &d.f_Pd);
      if (!rc) rc = upload_end(e);
      else e->up_on = false;
      d.ldm = e->fo.ldm;
      if (rc) { miosqp_qp_cleanup(e); return rc; }
      d.ldf = e->fo.ldf;
      d.ldn = e->fo.ldn;
      d.f_rows2 = d.f_rows;
      d.ldf2 = d.ldf;
      e->fold = true;
      e->tpr_ff = pick_tpr(M + 0.5 * n);
      e->tpr_fx = pick_tpr(0.5 * n);
      e->tpr_fc = pick_tpr((double)n);
      {
        const bool res_w = n + M <= RES_W_MAX && M > 0 && !(getenv("MIOSQP_RES_W") && atoi(getenv("MIOSQP_RES_W")) == 0);
        size_t need = resident_lds_doubles(n, M, res_w) * sizeof(double);
        e->sp_nnzA = f.panel_by_con.idx.size();
        e->sp_nnzP = f.Pbar.idx.size();
        const bool sp_on = res_w && !(getenv("MIOSQP_RES_SP") && atoi(getenv("MIOSQP_RES_SP")) == 0);
        const size_t sp_bytes = res_sp_doubles(e->sp_nnzA, e->sp_nnzP, n, M) * sizeof(double);
        if (sp_on && need + sp_bytes <= 150 * 1024) {  // room for the sparse rows of the termination test in LDS
          e->res_sp_off = (int)(need / sizeof(double));
          need += sp_bytes;
        }
        int wantr = s->resident;
        if (wantr < 0) wantr = (need <= 150 * 1024 && !coop_pref) ? 1 : 0;
        if (wantr && need <= 160 * 1024) {
          e->resident = true;
          e->res_lds = need;
          if (res_w) {
            // the explicit KKT inverse for the register-resident loop (res_admm_w, kernels_resident.inc)
            d.ldw = (n + M + 7) & ~7;
            double *Wd = nullptr;
            rc = dalloc(e, &Wd, (size_t)(n + M) * d.ldw + 64);
            if (!rc) rc = miosqp_device_kkt_inverse(d.f_rows, d.ldf, d.d2inv, n, M, Wd, d.ldw, e->stream);
            double *Kc = nullptr;  // dense [0 Abar ; Abar^T Pbar]: the rows of the termination test in the same form
            if (!rc) rc = dalloc(e, &Kc, (size_t)(n + M) * d.ldw + 64);
            if (rc) { miosqp_qp_cleanup(e); return rc; }
            d.W = Wd;
            d.Kc = Kc;
            hipLaunchKernelGGL(k_build_kc, dim3((n + M + 255) / 256, n + M), dim3(256), 0, e->stream, d, Kc);
            // the explicit inverse is checked before it is used (kernels_guard.inc); when it fails the workgroup keeps
            // the product-form sweeps in LDS, if they fit
            rc = inverse_guard(e);
            if (rc) { miosqp_qp_cleanup(e); return rc; }
            if (e->guard_tripped) {
              d.W = nullptr;
              d.Kc = nullptr;
              e->res_sp_off = -1;
              const size_t need2 = resident_lds_doubles(n, M, false) * sizeof(double);
              if (need2 <= 160 * 1024) e->res_lds = need2;
              else e->resident = false;
            }
          }
          if (e->resident) {
            HIPCHK(lds_limit_once((const void *)k_resident, 0));
            // lanes per row: as many as keep every row of a sweep in flight at once
            auto pow2_floor = [](int v) { int p = 1; while (2 * p <= v) p *= 2; return p; };
            e->res_tg1 = std::min(64, std::max(1, pow2_floor(RES_THREADS / n)));
            e->res_tg2 = std::min(64, std::max(1, pow2_floor(RES_THREADS / (n + M))));
            if (const char *ev = getenv("MIOSQP_RES_TG")) {
              int a1 = 0, b1 = 0;
              if (sscanf(ev, "%d,%d", &a1, &b1) == 2) { e->res_tg1 = a1; e->res_tg2 = b1; }
            }
          }
        }
      }
      {
        // cooperative register-resident solver: the explicit KKT inverse spread over the register
        // files of up to one workgroup per CU
        const int N = n + M;
        int wantc = coop_pref ? 1 : 0;
        int dev_now = 0;
        HIPCHK(hipGetDevice(&dev_now));
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, dev_now));
        int rw = coop_pick_rw(N);  // rows of W per workgroup
        if (const char *ev = getenv("MIOSQP_COOP_DBG")) d.coop_dbg = atoi(ev);
        const int T = (N + rw - 1) / rw;
        bool can = !e->resident && N <= 2048 && T <= prop.multiProcessorCount;
        e->coop_cus = prop.multiProcessorCount;
        if (wantc && can) {
          // the grid must fit the device with one workgroup per CU: ask the runtime instead of assuming it
          if (coop_occupancy(rw, coop_pick_cpt(N), N < 1024 ? 2 : 4) < 1) can = false;
        }
        if (wantc && can) {
          e->coop = true;
          e->coop_capable = true;
          e->coop_rw = rw;
          e->coop_cpt = coop_pick_cpt(N);
          e->coop_cptf = N < 1024 ? 2 : 4;
          e->coop_T = T;
          d.coop_mg = M;
          d.ldw = (N + 7) & ~7;
          double *Wd = nullptr;
          rc = dalloc(e, &Wd, (size_t)N * d.ldw + 64);
          if (!rc) rc = dalloc(e, &d.coop_tag, 64);
          d.coop_stride = 2 * rw;
          d.coop_half = 2 * (size_t)((N + 1 + 15) & ~15);  // two words per entry (one more than rows: the testers' decision), room for either row-block size
          if (!rc) rc = dalloc(e, &d.coop_buf, 2 * d.coop_half + 64);
          if (!rc) rc = dalloc(e, &d.coop_chk, 2 * d.coop_half + 64);
          if (!rc) rc = dalloc(e, &d.coop_q, (size_t)(256 + 16) * COOP_QS + 64);
          if (!rc) rc = dalloc(e, &d.coop_reg, 128);
          if (!rc) rc = dalloc(e, &d.coop_chz, d.coop_half + 64);
          if (!rc) rc = dalloc(e, &d.coop_dec, 64);
          if (!rc) rc = dalloc(e, &e->coop_epi, 2 * (size_t)COOP_EPI_ENTRIES + 64);
          {
            // the test on workgroups of its own when the exchange grid leaves enough CUs free
            d.coop_nt = coop_decision_slot(N, e->coop_cpt) ? coop_pick_testers(prop.multiProcessorCount - T) : 0;
            coop_pick_lag(e);
          }
          double *Kc = nullptr;
          if (!rc) rc = dalloc(e, &Kc, (size_t)N * d.ldw + 64);
          const bool timing = getenv("MIOSQP_SETUP_TIMING") != nullptr;
          if (timing) HIPCHK(hipStreamSynchronize(e->stream));
          const double tk0 = wall();
          if (!rc) rc = miosqp_device_kkt_inverse(d.f_rows, d.ldf, d.d2inv, n, M, Wd, d.ldw, e->stream);
          if (rc) { miosqp_qp_cleanup(e); return rc; }
          if (timing) {
            HIPCHK(hipStreamSynchronize(e->stream));
            fprintf(stderr, "[miosqp setup] %-32s %8.1f ms\n", "explicit KKT inverse (device)", 1e3 * (wall() - tk0));
          }
          d.W = Wd;
          d.Kc = Kc;
          d.coop_W = Wd;
          d.coop_ldw = d.ldw;
          hipLaunchKernelGGL(k_build_kc, dim3((N + 255) / 256, N), dim3(256), 0, e->stream, d, Kc);
          // the explicit inverse is checked before it is used (kernels_guard.inc); when it fails this engine iterates
          // with the sweeps of the product form (two launches per iteration) instead
          rc = inverse_guard(e);
          if (rc) { miosqp_qp_cleanup(e); return rc; }
          if (e->guard_tripped) {
            e->coop = false;
            e->coop_capable = false;
            d.W = nullptr;
            d.Kc = nullptr;
          }
        }
      }
      if (const char *ev = getenv("MIOSQP_BD_CFG")) e->bd_cfg = atoi(ev);
      if (const char *ev = getenv("MIOSQP_BM_VAR")) e->bm_var = atoi(ev);
      if (const char *ev = getenv("MIOSQP_SPIN_WAIT")) e->spin_wait = atoi(ev) != 0;
      if (const char *ev = getenv("MIOSQP_COMPACT")) e->compact = atoi(ev) != 0;
      if (const char *ev = getenv("MIOSQP_BM_ABLATE")) d.bm_ablate = atoi(ev);
      if (const char *ev = getenv("MIOSQP_FOLD_TPR")) {  // tuning hook: "fwd,x,c"
        int a = 0, b = 0, c = 0;
        if (sscanf(ev, "%d,%d,%d", &a, &b, &c) == 3) {
          e->tpr_ff = a;
          e->tpr_fx = b;
          e->tpr_fc = c;
        }
      }
      std::vector<double>().swap(e->fo.rows);
      std::vector<double>().swap(e->fo.GmT);
      std::vector<double>().swap(e->fo.Ad);
      std::vector<double>().swap(e->fo.Atd);
      std::vector<double>().swap(e->fo.Pd);
    }
  }
  {
    // persistent streaming solver (kernels_pers.inc): for what the cooperative solver cannot hold, or on request
    int want = s->pers;
    if (const char *ev = getenv("MIOSQP_PERS")) want = atoi(ev);
    // (auto, factor form: with the tail as S^-1 -- 42.8 us per iteration at config 5 against 48.7 with the two triangular
    //  sweeps and 50.7 with four launches; pers = 1 keeps the sweeps)
    if (want < 0) want = (!e->resident && !e->coop && n + M > 2048) ? (e->fold ? 1 : 2) : 0;
    if (want && !e->resident && !e->coop && M > 0) {
      e->st.pers = want;  // (2: with the tail as the explicit inverse of the reduced Hessian, see pers_setup)
      int rc = pers_setup(e);
      if (rc) { miosqp_qp_cleanup(e); return rc; }
    }
  }
  tick("product form, resident / cooperative / persistent set-up");
  e->chunk = e->st.check_termination;
  e->tail_iters = e->st.max_iter % e->chunk;
  HIPCHK(hipStreamSynchronize(e->stream));
  if (!e->resident && !e->coop && !e->pers) {  // the single-launch solvers need no captured chunk
    int rc = capture_chunk(e, e->chunk, &e->g_full, &e->x_full);
    if (!rc && e->tail_iters > 0) rc = capture_chunk(e, e->tail_iters, &e->g_tail, &e->x_tail);
    if (rc) { miosqp_qp_cleanup(e); return rc; }
  }
  if (e->coop_capable || e->pers_capable) {  // whole-chip launches of one process take turns (host.inc: ChipTurn)
    int rc = chip_join(e);
    if (rc) { miosqp_qp_cleanup(e); return rc; }
  }
  if (e->coop) {
    int rc = calibrate_coop_nap(e);
    if (rc) { miosqp_qp_cleanup(e); return rc; }
  }
  tick("graph capture / calibration");
  if (t_on) fprintf(stderr, "[miosqp setup] %-32s %8.3f ms\n", "miosqp_qp_setup, all of it", 1e3 * (wall() - t_begin));
  *out = e;
  return 0;
}

// The update / warm-start calls return as soon as their work is queued (the solve that follows is ordered behind it
// on the engine's stream); what must not happen is the NEXT upload overwriting the pinned staging area while the
// previous copy out of it is still under way: one event per staging region, waited for before the region is rewritten.
static int stage_wait(miosqp_qp_engine *e, int k) {
  if (e->stage_busy[k]) {
    HIPCHK(hipEventSynchronize(e->rt.ev_stage[k]));
    e->stage_busy[k] = false;
  }
  return 0;
}
static int stage_mark(miosqp_qp_engine *e, int k) {
  HIPCHK(hipEventRecord(e->rt.ev_stage[k], e->stream));
  e->stage_busy[k] = true;
  return 0;
}

int miosqp_qp_update_bounds(miosqp_qp_engine *e, const double *l, const double *u) {
  if (!e || !l || !u) return MIOSQP_EARG;
  ENTER(e);
  for (int i = 0; i < e->M; i++)
    if (l[i] > u[i]) return MIOSQP_EBOUNDS;
  if (int rc = stage_wait(e, 0)) return rc;
  memcpy(e->h_in, l, sizeof(double) * e->M);
  memcpy(e->h_in + e->M, u, sizeof(double) * e->M);
  HIPCHK(hipMemcpyAsync(e->d.raw_l, e->h_in, sizeof(double) * 2 * e->M, hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(k_scale_bounds, dim3((e->M + 255) / 256), dim3(256), 0, e->stream, e->d);
  return stage_mark(e, 0);
}

int miosqp_qp_update_lin_cost(miosqp_qp_engine *e, const double *q) {
  if (!e || !q) return MIOSQP_EARG;
  ENTER(e);
  if (int rc = stage_wait(e, 1)) return rc;
  double *hx = e->h_in + 2 * (size_t)e->M;
  memcpy(hx, q, sizeof(double) * e->n);
  HIPCHK(hipMemcpyAsync(e->d.raw_x, hx, sizeof(double) * e->n, hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(k_scale_q, dim3((e->n + 255) / 256), dim3(256), 0, e->stream, e->d);
  return stage_mark(e, 1);
}

static int enqueue_warm(miosqp_qp_engine *e) {
  const int big = e->n > e->M ? e->n : e->M;
  hipLaunchKernelGGL(k_scale_warm, dim3((big + 255) / 256), dim3(256), 0, e->stream, e->d);
  DISPATCH_TPR(e->tpr_pc, k_warm_z, e->M, e->stream, e->d);
  return 0;
}

int miosqp_qp_warm_start(miosqp_qp_engine *e, const double *x, const double *y) {
  if (!e || !x || !y) return MIOSQP_EARG;
  ENTER(e);
  if (int rc = stage_wait(e, 1)) return rc;
  double *hx = e->h_in + 2 * (size_t)e->M;
  memcpy(hx, x, sizeof(double) * e->n);
  memcpy(hx + e->n, y, sizeof(double) * e->M);
  HIPCHK(hipMemcpyAsync(e->d.raw_x, hx, sizeof(double) * (e->n + e->M), hipMemcpyHostToDevice, e->stream));
  enqueue_warm(e);
  return stage_mark(e, 1);
}

static int begin_solve(miosqp_qp_engine *e) {
  const int big = e->n > e->M ? e->n : e->M;
  HIPCHK(hipEventRecord(e->ev0, e->stream));
  if (!e->st.warm_start) hipLaunchKernelGGL(k_zero_iterates, dim3((big + 255) / 256), dim3(256), 0, e->stream, e->d);
  hipLaunchKernelGGL(k_reset_ctrl, dim3(1), dim3(1), 0, e->stream, e->d);
  hipLaunchKernelGGL(k_init_wh, dim3(((e->M > e->n ? e->M : e->n) + 255) / 256), dim3(256), 0, e->stream, e->d);
  return 0;
}

int miosqp_qp_solve(miosqp_qp_engine *e, double *x_out, double *y_out, miosqp_qp_info *info) {
  if (!e || !x_out || !y_out || !info) return MIOSQP_EARG;
  ENTER(e);
  const double t0 = wall();
  maybe_rejoin_coop(e);
  for (;;) {
    int rc = begin_solve(e);
    if (!rc) rc = run_loop(e);
    if (!rc) rc = finish_and_fetch(e, 0, x_out, y_out, info, t0);
    if (rc != COOP_RETRY) return rc;
    // called off before any iterate was touched: the same solve again in the two-kernel form
    rc = leave_coop(e);
    if (rc) return rc;
  }
}

int miosqp_qp_set_integer_rows(miosqp_qp_engine *e, int32_t n_int, const int32_t *i_idx, int32_t m_orig) {
  if (!e || n_int < 0 || n_int > e->n || m_orig < 0 || m_orig + n_int != e->M || (n_int && !i_idx)) {
    g_err = "set_integer_rows: need m_orig + n_int == M";
    return MIOSQP_EARG;
  }
  ENTER(e);
  for (int k = 0; k < n_int; k++)
    if (i_idx[k] < 0 || i_idx[k] >= e->n) return MIOSQP_EARG;
  e->h_iidx.assign(i_idx, i_idx + n_int);  // host copy (the streaming driver rounds an incumbent's integer entries)
  // uploads through region 1 of the pinned staging block when they fit (three blocking copies were ~30 us of a small set-up):
  // [ i_idx (n_int ints) | int_pos (n ints) | a_int (n doubles, below) ]
  const size_t stage_doubles = e->rt.in_cap > 2 * (size_t)e->M ? e->rt.in_cap - 2 * (size_t)e->M : 0;
  const size_t ints_doubles = ((size_t)n_int + (size_t)e->n + 1) / 2;
  const bool staged = stage_doubles >= ints_doubles + (size_t)e->n;
  double *const stage = e->h_in + 2 * (size_t)e->M;
  if (staged) {
    if (int rcw = stage_wait(e, 1)) return rcw;
  }
  {
    std::vector<int> pos(e->n, -1);
    for (int k = 0; k < n_int; k++) pos[i_idx[k]] = k;
    if (!e->d.int_pos) {
      int *ip = nullptr;
      int rc = dalloc(e, &ip, (size_t)e->n);
      if (rc) return rc;
      e->d.int_pos = ip;
    }
    if (staged) {
      int *hi = reinterpret_cast<int *>(stage);
      memcpy(hi, i_idx, sizeof(int) * n_int);
      memcpy(hi + n_int, pos.data(), sizeof(int) * e->n);
      if (n_int) HIPCHK(hipMemcpyAsync((void *)e->d.i_idx, hi, sizeof(int) * n_int, hipMemcpyHostToDevice, e->stream));
      HIPCHK(hipMemcpyAsync((void *)e->d.int_pos, hi + n_int, sizeof(int) * e->n, hipMemcpyHostToDevice, e->stream));
    } else {
      if (n_int) HIPCHK(hipMemcpy((void *)e->d.i_idx, i_idx, sizeof(int) * n_int, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy((void *)e->d.int_pos, pos.data(), sizeof(int) * e->n, hipMemcpyHostToDevice));
    }
  }
  e->d.n_int = n_int;
  e->d.m_orig = m_orig;
  e->have_int = true;
  // Identity rows (kernels_batched.inc): when every bound row has exactly one entry, at its variable's column, and
  // no variable has two, the matrix-core sweeps leave those rows out.  Decided once, before the batched arrays exist.
  if (e->fold && e->bd_cfg == 0 && n_int > 0 && e->Bcap == 0 && !getenv("MIOSQP_NO_IDROWS")) {
    const miosqp::PCsr &C = e->fa.panel_by_con;
    std::vector<double> aint(e->n, 0.0);
    bool ok = true;
    for (int k = 0; k < n_int && ok; k++) {
      const int j = m_orig + k;
      int cnt = 0, col = -1;
      double val = 0.0;
      for (int t = C.ptr[j]; t < C.ptr[j + 1]; t++)
        if (e->fa.A_val[t] != 0.0) { cnt++; col = C.idx[t]; val = e->fa.A_val[t]; }
      ok = cnt == 1 && col == i_idx[k] && aint[col] == 0.0;
      if (ok) aint[col] = e->st.rho * val;
    }
    if (ok) {
      const int n = e->n, m = m_orig;
      const int ld2 = (m + n + 15) & ~15;
      double *f2 = nullptr, *ai = nullptr;
      int rc = dalloc(e, &f2, (size_t)n * ld2 + 64);
      if (!rc) rc = dalloc(e, &ai, (size_t)n);
      if (rc) return rc;
      if (staged) {
        memcpy(stage + ints_doubles, aint.data(), sizeof(double) * n);
        HIPCHK(hipMemcpyAsync(ai, stage + ints_doubles, sizeof(double) * n, hipMemcpyHostToDevice, e->stream));
      } else {
        HIPCHK(hipMemcpy(ai, aint.data(), sizeof(double) * n, hipMemcpyHostToDevice));
      }
      hipLaunchKernelGGL(k_drop_bound_columns, dim3((ld2 + 255) / 256, n), dim3(256), 0, e->stream, e->d.f_rows, e->d.ldf,
                         f2, ld2, m, n_int, n);
      // (no wait: every reader of f2 is a later kernel on this stream)
      e->d.f_rows2 = f2;
      e->d.ldf2 = ld2;
      e->d.a_int = ai;
      e->d.wh_m = m;
    }
  }
  // the captured graphs hold Dev by value but never read n_int / m_orig / i_idx contents
  if (staged) {
    if (int rcm = stage_mark(e, 1)) return rcm;
  }
  return coop_drop_identity_rows(e, n_int, i_idx, m_orig);
}

int miosqp_qp_set_root(miosqp_qp_engine *e, const double *l_root, const double *u_root, double eps_int_feas,
                       double eps_lin) {
  if (!e || !l_root || !u_root || !e->have_int) {
    g_err = "set_root: call miosqp_qp_set_integer_rows first";
    return MIOSQP_EARG;
  }
  ENTER(e);
  if (e->M > 0) {
    // through the pinned staging block, on the engine's stream, no wait: whatever reads the root bounds is queued
    // behind these copies (two blocking copies from pageable memory and two stream synchronisations were 40 us of an
    // MPC step's 90)
    if (int rcw = stage_wait(e, 0)) return rcw;
    memcpy(e->h_in, l_root, sizeof(double) * e->M);
    memcpy(e->h_in + e->M, u_root, sizeof(double) * e->M);
    HIPCHK(hipMemcpyAsync(e->d.root_l, e->h_in, sizeof(double) * e->M, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->d.root_u, e->h_in + e->M, sizeof(double) * e->M, hipMemcpyHostToDevice, e->stream));
    if (int rcm = stage_mark(e, 0)) return rcm;
  }
  if ((e->x_stream || e->x_stream_a) && (e->d.eps_int != eps_int_feas || e->d.eps_lin != eps_lin)) drop_stream_graph(e);  // captured by value
  e->d.eps_int = eps_int_feas;
  e->d.eps_lin = eps_lin;
  e->d.digest = 1;
  e->dh.eps_int = eps_int_feas;  // the pool's harvest tile carries its own copy
  e->dh.eps_lin = eps_lin;
  e->dh.digest = 1;
  return 0;
}

int miosqp_qp_solve_node(miosqp_qp_engine *e, const double *l, const double *u, const double *x0,
                         const double *y0, double *x_out, double *y_out, miosqp_qp_info *info) {
  if (!e || !l || !u || !x0 || !y0 || !x_out || !y_out || !info) return MIOSQP_EARG;
  ENTER(e);
  if (!e->have_int) {
    g_err = "solve_node: call miosqp_qp_set_integer_rows first";
    return MIOSQP_EARG;
  }
  const double t0 = wall();
  const int n = e->n, M = e->M;
  for (int i = 0; i < M; i++)
    if (l[i] > u[i]) return MIOSQP_EBOUNDS;
  if (int rcw = stage_wait(e, 0)) return rcw;
  if (int rcw = stage_wait(e, 1)) return rcw;
  memcpy(e->h_in, l, sizeof(double) * M);
  memcpy(e->h_in + M, u, sizeof(double) * M);
  memcpy(e->h_in + 2 * (size_t)M, x0, sizeof(double) * n);
  memcpy(e->h_in + 2 * (size_t)M + n, y0, sizeof(double) * M);
  maybe_rejoin_coop(e);
  for (;;) {
    HIPCHK(hipEventRecord(e->ev0, e->stream));
    HIPCHK(hipMemcpyAsync(e->d.raw_l, e->h_in, sizeof(double) * (3 * (size_t)M + n), hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(k_node_pre, dim3(((M > n ? M : n) + 255) / 256), dim3(256), 0, e->stream, e->d);
    DISPATCH_TPR(e->tpr_pc, k_warm_zw, M, e->stream, e->d);
    int rc = run_loop(e);
    if (!rc) rc = finish_and_fetch(e, 1, x_out, y_out, info, t0);
    if (rc != COOP_RETRY) return rc;
    rc = leave_coop(e);
    if (rc) return rc;
  }
}

int miosqp_qp_solve_batch(miosqp_qp_engine *e, int32_t B, const double *l, const double *u, const double *x0,
                          const double *y0, double *x_out, double *y_out, miosqp_qp_info *info) {
  if (!e || B < 0 || (B > 0 && (!l || !u || !x0 || !y0 || !x_out || !y_out || !info))) return MIOSQP_EARG;
  ENTER(e);
  if (!e->have_int) {
    g_err = "solve_batch: call miosqp_qp_set_integer_rows first";
    return MIOSQP_EARG;
  }
  const size_t n = e->n, M = e->M;
  for (size_t k = 0; k < (size_t)B * M; k++)
    if (l[k] > u[k]) return MIOSQP_EBOUNDS;
  if (e->Bcap == 0) {
    int cap = e->st.max_batch > 1 ? e->st.max_batch : 64;
    if (cap > 1024) cap = 1024;
    int rc = alloc_batch(e, cap);
    if (rc) return rc;
  }
  for (int s0 = 0; s0 < B; s0 += e->Bcap) {
    const int nb = B - s0 < e->Bcap ? B - s0 : e->Bcap;
    int rc = solve_slice(e, nb, l + s0 * M, u + s0 * M, x0 + s0 * n, y0 + s0 * M, x_out + s0 * n, y_out + s0 * M,
                         info + s0);
    if (rc) return rc;
  }
  return 0;
}

int miosqp_qp_solve_tree(miosqp_qp_engine *e, const double *l, const double *u, const double *x0, const double *y0,
                         double upper0, const double *x_inc0, int32_t tree_explor_rule, int32_t max_iter_bb,
                         double *x_out, miosqp_tree_info *info) {
  if (!e || !l || !u || !x0 || !y0 || !x_out || !info || max_iter_bb < 1 || tree_explor_rule < 0 || tree_explor_rule > 1)
    return MIOSQP_EARG;
  ENTER(e);
  if (!e->have_int || !e->d.digest) {
    g_err = "solve_tree: call miosqp_qp_set_integer_rows and miosqp_qp_set_root first";
    return MIOSQP_EARG;
  }
  const int n = e->n, M = e->M, p = e->d.n_int;
  // (the tree kernels decide the same way on the device: explicit inverse there and n + M small enough -> no product
  //  form in LDS; an engine without W yet gets it below when n + M <= 64, which only shrinks what the kernel uses)
  const bool tree_w = e->d.W != nullptr && n + M <= RES_W_MAX;
  size_t lds = tree_lds_doubles(n, M, tree_w) * sizeof(double);
  int tree_sp_off = -1;
  {
    const size_t sp_bytes = res_sp_doubles(e->sp_nnzA, e->sp_nnzP, n, M) * sizeof(double);
    if (tree_w && e->sp_nnzA > 0 && lds + sp_bytes <= 160 * 1024 &&
        !(getenv("MIOSQP_RES_SP") && atoi(getenv("MIOSQP_RES_SP")) == 0)) {
      tree_sp_off = (int)(lds / sizeof(double));
      lds += sp_bytes;
    }
  }
  if (!e->fold || lds > 160 * 1024 || p < 1) {  // whatever form single nodes use: the product-form rows must exist and fit
    g_err = "solve_tree: only for problems whose product-form factor, iterates and leaf list fit 160 KB of LDS";
    return MIOSQP_EUNSUPPORTED;
  }
  for (int i = 0; i < M; i++)
    if (l[i] > u[i]) return MIOSQP_EBOUNDS;
  const double t0 = wall();
  // n + M <= 64: the whole search in ONE wavefront on the explicit KKT inverse (k_tree_w); MIOSQP_TREE_WAVE=0 keeps k_tree
  bool wave = n + M <= TW && !e->guard_tripped && !(getenv("MIOSQP_TREE_WAVE") && atoi(getenv("MIOSQP_TREE_WAVE")) == 0);
  if (wave && !e->d.Kc) {  // an engine in the LDS-resident form has not built Kc (and perhaps W) yet
    Dev &d = e->d;
    const int N = n + M;
    d.ldw = (N + 7) & ~7;
    double *Wd = const_cast<double *>(d.W), *Kc = nullptr;
    int rcw = 0;
    if (!Wd) {
      rcw = dalloc(e, &Wd, (size_t)N * d.ldw + 64);
      if (!rcw) rcw = miosqp_device_kkt_inverse(d.f_rows, d.ldf, d.d2inv, n, M, Wd, d.ldw, e->stream);
    }
    if (!rcw) rcw = dalloc(e, &Kc, (size_t)N * d.ldw + 64);
    if (rcw) return rcw;
    d.W = Wd;
    d.Kc = Kc;
    hipLaunchKernelGGL(k_build_kc, dim3((N + 255) / 256, N), dim3(256), 0, e->stream, d, Kc);
    if (int rcg = inverse_guard(e)) return rcg;
    if (e->guard_tripped) {  // (kernels_guard.inc) the explicit inverse failed its check: the workgroup kernel with the sweeps
      d.W = nullptr;
      d.Kc = nullptr;
      wave = false;
    }
  }
  if (!e->tree_ready) {
    HIPCHK(lds_limit_once((const void *)k_tree, 1));
    int rc = dalloc(e, &e->ta.lf_lo, (size_t)TREE_CAP * p);
    if (!rc) rc = dalloc(e, &e->ta.lf_hi, (size_t)TREE_CAP * p);
    if (!rc) rc = dalloc(e, &e->ta.lf_x, (size_t)TREE_CAP * n);
    if (!rc) rc = dalloc(e, &e->ta.lf_y, (size_t)TREE_CAP * M);
    if (!rc) rc = dalloc(e, &e->ta.inc_x, (size_t)n);
    if (!rc) rc = dalloc(e, &e->ta.out, 1);
    if (rc) return rc;
    e->tree_ready = true;
  }
  if (int rcw = stage_wait(e, 0)) return rcw;
  if (int rcw = stage_wait(e, 1)) return rcw;
  memcpy(e->h_in, l, sizeof(double) * M);
  memcpy(e->h_in + M, u, sizeof(double) * M);
  memcpy(e->h_in + 2 * (size_t)M, x0, sizeof(double) * n);
  memcpy(e->h_in + 2 * (size_t)M + n, y0, sizeof(double) * M);
  const bool have_inc = x_inc0 != nullptr && upper0 < 1.7e308;
  TreeArgs ta = e->ta;
  ta.rule = tree_explor_rule;
  ta.max_nodes = max_iter_bb;
  ta.max_iter = e->st.max_iter;
  ta.check_every = e->st.check_termination;
  {  // lanes per row of the two sweeps, as the LDS-resident solver picks them
    auto pow2_floor = [](int v) { int q = 1; while (2 * q <= v) q *= 2; return q; };
    ta.TG1 = std::min(64, std::max(1, pow2_floor(RES_THREADS / n)));
    ta.TG2 = std::min(64, std::max(1, pow2_floor(RES_THREADS / (n + M))));
  }
  ta.upper0 = have_inc ? upper0 : 1.0 / 0.0;
  ta.done = nullptr;
  ta.sp_off = tree_sp_off;
  const TreeOut *o = nullptr;
  const double *x_found = nullptr;
  float ms = 0;
  if (wave) {
    // The one-wavefront search reads the root straight from the pinned staging block and writes incumbent and
    // outcome into coherent host memory, the completion word last: ONE launch and no copy per MIQP (three copies and
    // two events around a 60-400 us kernel were 25 us).  Device time: the kernel's own 100 MHz stamps.
    if (!e->h_tree) {  // lives in the runtime bundle: a pinned allocation per engine was ~80 us of a 130 us first solve
      if (e->rt.tree_cap < (size_t)n + 16) {
        if (e->rt.h_tree) hipHostFree(e->rt.h_tree);
        e->rt.h_tree = nullptr;
        e->rt.tree_cap = 0;
        const size_t cap = std::max<size_t>(512, (size_t)n + 16);
        HIPCHK(hipHostMalloc((void **)&e->rt.h_tree, sizeof(double) * cap, hipHostMallocCoherent));
        e->rt.tree_cap = cap;
      }
      e->h_tree = e->rt.h_tree;
      memset(e->h_tree, 0, sizeof(double) * ((size_t)n + 16));
    }
    unsigned long long *done = e->h_tree;
    TreeOut *out = reinterpret_cast<TreeOut *>(e->h_tree + 4);
    double *inc = reinterpret_cast<double *>(e->h_tree + 16);
    static_assert(sizeof(TreeOut) <= 12 * sizeof(unsigned long long), "TreeOut fits its place in h_tree");
    __atomic_store_n(&done[0], 0ull, __ATOMIC_RELAXED);
    if (have_inc) memcpy(inc, x_inc0, sizeof(double) * n);
    Dev dp = e->d;
    dp.raw_l = e->h_in;
    dp.raw_u = e->h_in + M;
    dp.raw_x = e->h_in + 2 * (size_t)M;
    dp.raw_y = e->h_in + 2 * (size_t)M + n;
    ta.inc_x = inc;
    ta.out = out;
    ta.done = done;
    static unsigned long long *tree_prof = nullptr;  // debug (MIOSQP_TREE_PROF=1): ticks per phase, summed over launches
    static const bool tree_prof_on = getenv("MIOSQP_TREE_PROF") != nullptr;
    if (tree_prof_on && !tree_prof) {
      HIPCHK(hipMalloc((void **)&tree_prof, 64));
      HIPCHK(hipMemset(tree_prof, 0, 64));
    }
    if (tree_prof_on) dp.prof = tree_prof;
    hipLaunchKernelGGL(k_tree_w, dim3(1), dim3(TW), 0, e->stream, dp, ta);
    if (e->spin_wait) {
      unsigned long long spins = 0;
      while (__atomic_load_n(&done[0], __ATOMIC_ACQUIRE) == 0ull) {
        if ((++spins & 0xfffff) == 0) {
          const hipError_t q = hipStreamQuery(e->stream);
          if (q != hipSuccess && q != hipErrorNotReady) HIPCHK(q);
          if (q == hipSuccess && __atomic_load_n(&done[0], __ATOMIC_ACQUIRE) == 0ull) {
            g_err = "solve_tree: the stream is idle and the outcome never arrived";
            return MIOSQP_EHIP;
          }
        }
      }
    } else {
      HIPCHK(hipStreamSynchronize(e->stream));
    }
    if (tree_prof_on) {
      unsigned long long h[6];
      HIPCHK(hipMemcpy(h, tree_prof, sizeof h, hipMemcpyDeviceToHost));
      fprintf(stderr, "[tree prof] cumulative us: prologue %.1f  iterations %.1f  tests %.1f  epilogue %.1f  branch %.1f\n",
              h[0] / 100.0, h[1] / 100.0, h[2] / 100.0, h[3] / 100.0, h[4] / 100.0);
    }
    o = out;
    x_found = inc;
    ms = (float)(1e-5 * (double)(done[2] - done[1]));  // 100 MHz ticks -> ms
  } else {
    HIPCHK(hipEventRecord(e->ev0, e->stream));
    HIPCHK(hipMemcpyAsync(e->d.raw_l, e->h_in, sizeof(double) * (3 * (size_t)M + n), hipMemcpyHostToDevice, e->stream));
    if (have_inc) {
      memcpy(e->h_out, x_inc0, sizeof(double) * n);
      HIPCHK(hipMemcpyAsync(e->ta.inc_x, e->h_out, sizeof(double) * n, hipMemcpyHostToDevice, e->stream));
    }
    static unsigned long long *ktree_prof = nullptr;  // debug (MIOSQP_TREE_PROF=1), as for the one-wavefront kernel
    static const bool ktree_prof_on = getenv("MIOSQP_TREE_PROF") != nullptr;
    if (ktree_prof_on && !ktree_prof) {
      HIPCHK(hipMalloc((void **)&ktree_prof, 128));
      HIPCHK(hipMemset(ktree_prof, 0, 128));
    }
    Dev dk = e->d;
    if (ktree_prof_on) dk.prof = ktree_prof;
    hipLaunchKernelGGL(k_tree, dim3(1), dim3(RES_THREADS), lds, e->stream, dk, ta);
    if (ktree_prof_on) {
      unsigned long long h[16];
      HIPCHK(hipMemcpyAsync(h, ktree_prof, sizeof h, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
      fprintf(stderr, "[tree prof] k_tree cumulative us: choose+prologue %.1f  relaxation %.1f  epilogue %.1f  branch+children %.1f\n",
              h[8] / 100.0, h[9] / 100.0, h[10] / 100.0, h[11] / 100.0);
    }
    HIPCHK(hipMemcpyAsync(e->h_out, e->ta.inc_x, sizeof(double) * n, hipMemcpyDeviceToHost, e->stream));
    static_assert(sizeof(TreeOut) <= sizeof(Ctrl), "TreeOut travels through the pinned control block");
    HIPCHK(hipMemcpyAsync(e->h_ctrl, e->ta.out, sizeof(TreeOut), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipEventRecord(e->ev1, e->stream));
    if (e->spin_wait) {
      hipError_t q;
      while ((q = hipEventQuery(e->ev1)) == hipErrorNotReady) {}
      if (q != hipSuccess) HIPCHK(q);
    } else {
      HIPCHK(hipStreamSynchronize(e->stream));
    }
    o = reinterpret_cast<const TreeOut *>(e->h_ctrl);
    x_found = e->h_out;
    HIPCHK(hipEventElapsedTime(&ms, e->ev0, e->ev1));
  }
  info->nodes = o->nodes;
  info->osqp_iter = o->osqp_iter;
  info->leaves_left = o->leaves_left;
  info->overflow = o->overflow;
  info->max_leaves = o->max_leaves;
  info->found = o->found;
  info->upper_glob = o->upper;
  info->lower_glob = o->lower_glob;
  info->device_time = 1e-3 * ms;
  info->run_time = wall() - t0;
  if (o->found || have_inc) memcpy(x_out, o->found ? x_found : x_inc0, sizeof(double) * n);
  e->loop_ms += ms;
  e->loop_iters += o->osqp_iter;
  return 0;
}

int miosqp_qp_solve_trees(miosqp_qp_engine *e, int32_t B, const double *q, const double *l, const double *u,
                          const double *x0, const double *y0, const double *upper0, const double *x_inc0,
                          int32_t tree_explor_rule, int32_t max_iter_bb, double *x_out, miosqp_tree_info *info) {
  if (!e || B < 1 || !q || !l || !u || !x0 || !y0 || !upper0 || !x_out || !info || max_iter_bb < 1 || tree_explor_rule < 0 ||
      tree_explor_rule > 1)
    return MIOSQP_EARG;
  ENTER(e);
  if (!e->have_int || !e->d.digest) {
    g_err = "solve_trees: call miosqp_qp_set_integer_rows and miosqp_qp_set_root first";
    return MIOSQP_EARG;
  }
  const int n = e->n, M = e->M, p = e->d.n_int;
  const size_t blk = 3 * (size_t)M + n;
  const bool tree_w = e->d.W != nullptr && n + M <= RES_W_MAX;
  size_t lds = tree_lds_doubles(n, M, tree_w) * sizeof(double);
  int tree_sp_off = -1;
  {
    const size_t sp_bytes = res_sp_doubles(e->sp_nnzA, e->sp_nnzP, n, M) * sizeof(double);
    if (tree_w && e->sp_nnzA > 0 && lds + sp_bytes <= 160 * 1024 &&
        !(getenv("MIOSQP_RES_SP") && atoi(getenv("MIOSQP_RES_SP")) == 0)) {
      tree_sp_off = (int)(lds / sizeof(double));
      lds += sp_bytes;
    }
  }
  if (!e->fold || lds > 160 * 1024 || p < 1) {
    g_err = "solve_trees: only for problems whose product-form factor, iterates and leaf list fit 160 KB of LDS";
    return MIOSQP_EUNSUPPORTED;
  }
  for (size_t k = 0; k < (size_t)B * M; k++)
    if (l[k] > u[k]) return MIOSQP_EBOUNDS;
  const double t0 = wall();
  bool wave = n + M <= TW && !e->guard_tripped && !(getenv("MIOSQP_TREE_WAVE") && atoi(getenv("MIOSQP_TREE_WAVE")) == 0);
  if (wave && !e->d.Kc) {  // as miosqp_qp_solve_tree: an engine in the LDS-resident form has not built Kc (and perhaps W) yet
    Dev &d = e->d;
    const int N = n + M;
    d.ldw = (N + 7) & ~7;
    double *Wd = const_cast<double *>(d.W), *Kc = nullptr;
    int rcw = 0;
    if (!Wd) {
      rcw = dalloc(e, &Wd, (size_t)N * d.ldw + 64);
      if (!rcw) rcw = miosqp_device_kkt_inverse(d.f_rows, d.ldf, d.d2inv, n, M, Wd, d.ldw, e->stream);
    }
    if (!rcw) rcw = dalloc(e, &Kc, (size_t)N * d.ldw + 64);
    if (rcw) return rcw;
    d.W = Wd;
    d.Kc = Kc;
    hipLaunchKernelGGL(k_build_kc, dim3((N + 255) / 256, N), dim3(256), 0, e->stream, d, Kc);
    if (int rcg = inverse_guard(e)) return rcg;
    if (e->guard_tripped) {  // (kernels_guard.inc) the explicit inverse failed its check: the workgroup kernel with the sweeps
      d.W = nullptr;
      d.Kc = nullptr;
      wave = false;
    }
  }
  if (!wave) HIPCHK(lds_limit_once((const void *)k_tree, 1));
  if (B > e->tb_cap) {  // (a larger batch takes new arrays; the old ones stay with the pool until cleanup)
    const size_t cap = (size_t)std::max(B, 2 * e->tb_cap);
    int rc = pool_reserve(e, cap * ((size_t)TREE_CAP * (2 * (size_t)p + n + M) + blk + 3 * (size_t)n + 16) * sizeof(double) + 4096);
    if (!rc) rc = dalloc(e, &e->tb_q, cap * n);
    if (!rc) rc = dalloc(e, &e->tb_qraw, cap * n);
    if (!rc) rc = dalloc(e, &e->tb_raw, cap * blk);
    if (!rc) rc = dalloc(e, &e->tb_lo, cap * TREE_CAP * p);
    if (!rc) rc = dalloc(e, &e->tb_hi, cap * TREE_CAP * p);
    if (!rc) rc = dalloc(e, &e->tb_x, cap * TREE_CAP * n);
    if (!rc) rc = dalloc(e, &e->tb_y, cap * TREE_CAP * M);
    if (!rc) rc = dalloc(e, &e->tb_inc, cap * n);
    if (!rc) rc = dalloc(e, &e->tb_upper, cap);
    if (!rc) rc = dalloc(e, &e->tb_out, cap);
    if (rc) return rc;
    e->tb_cap = (int)cap;
  }
  // one staging vector: [raw blocks | q | incumbents | upper] in
  const size_t nin = (size_t)B * (blk + 2 * (size_t)n + 1);
  if (e->tb_host.size() < nin) e->tb_host.resize(nin);
  double *h_raw = e->tb_host.data(), *h_q = h_raw + (size_t)B * blk, *h_inc = h_q + (size_t)B * n, *h_up = h_inc + (size_t)B * n;
  bool any_inc = false;
  for (int b = 0; b < B; b++) {
    double *r = h_raw + (size_t)b * blk;
    memcpy(r, l + (size_t)b * M, sizeof(double) * M);
    memcpy(r + M, u + (size_t)b * M, sizeof(double) * M);
    memcpy(r + 2 * (size_t)M, x0 + (size_t)b * n, sizeof(double) * n);
    memcpy(r + 2 * (size_t)M + n, y0 + (size_t)b * M, sizeof(double) * M);
    const bool have = x_inc0 != nullptr && upper0[b] < 1.7e308;
    h_up[b] = have ? upper0[b] : 1.0 / 0.0;
    if (have) memcpy(h_inc + (size_t)b * n, x_inc0 + (size_t)b * n, sizeof(double) * n);
    else memset(h_inc + (size_t)b * n, 0, sizeof(double) * n);
    any_inc = any_inc || have;
  }
  memcpy(h_q, q, sizeof(double) * (size_t)B * n);
  HIPCHK(hipEventRecord(e->ev0, e->stream));
  HIPCHK(hipMemcpyAsync(e->tb_raw, h_raw, sizeof(double) * (size_t)B * blk, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(e->tb_qraw, h_q, sizeof(double) * (size_t)B * n, hipMemcpyHostToDevice, e->stream));
  if (any_inc) HIPCHK(hipMemcpyAsync(e->tb_inc, h_inc, sizeof(double) * (size_t)B * n, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(e->tb_upper, h_up, sizeof(double) * B, hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(k_scale_q_batch, dim3((unsigned)(((size_t)B * n + 255) / 256)), dim3(256), 0, e->stream, e->d, e->tb_qraw,
                     e->tb_q, B);
  TreeArgs ta{};
  ta.rule = tree_explor_rule;
  ta.max_nodes = max_iter_bb;
  ta.max_iter = e->st.max_iter;
  ta.check_every = e->st.check_termination;
  {
    auto pow2_floor = [](int v) { int q2 = 1; while (2 * q2 <= v) q2 *= 2; return q2; };
    ta.TG1 = std::min(64, std::max(1, pow2_floor(RES_THREADS / n)));
    ta.TG2 = std::min(64, std::max(1, pow2_floor(RES_THREADS / (n + M))));
  }
  ta.upper0 = 1.0 / 0.0;
  ta.lf_lo = e->tb_lo; ta.lf_hi = e->tb_hi; ta.lf_x = e->tb_x; ta.lf_y = e->tb_y;
  ta.inc_x = e->tb_inc;
  ta.out = e->tb_out;
  ta.done = nullptr;
  ta.sp_off = tree_sp_off;
  ta.batch = 1;
  ta.upper_b = e->tb_upper;
  Dev db = e->d;  // instance 0's arrays; workgroup b moves on from there (tree_instance)
  db.q = e->tb_q;
  db.qraw = e->tb_qraw;
  db.raw_l = e->tb_raw;
  db.raw_u = e->tb_raw + M;
  db.raw_x = e->tb_raw + 2 * (size_t)M;
  db.raw_y = e->tb_raw + 2 * (size_t)M + n;
  db.prof = nullptr;
  if (wave) hipLaunchKernelGGL(k_tree_w, dim3(B), dim3(TW), 0, e->stream, db, ta);
  else hipLaunchKernelGGL(k_tree, dim3(B), dim3(RES_THREADS), lds, e->stream, db, ta);
  if (e->tb_hout.size() < (size_t)B) e->tb_hout.resize(B);
  HIPCHK(hipMemcpyAsync(e->tb_hout.data(), e->tb_out, sizeof(TreeOut) * B, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(h_inc, e->tb_inc, sizeof(double) * (size_t)B * n, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipEventRecord(e->ev1, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, e->ev0, e->ev1));
  const double wall_s = wall() - t0;
  int64_t iters = 0;
  for (int b = 0; b < B; b++) {
    const TreeOut &o = e->tb_hout[b];
    info[b].nodes = o.nodes;
    info[b].osqp_iter = o.osqp_iter;
    info[b].leaves_left = o.leaves_left;
    info[b].overflow = o.overflow;
    info[b].max_leaves = o.max_leaves;
    info[b].found = o.found;
    info[b].upper_glob = o.upper;
    info[b].lower_glob = o.lower_glob;
    info[b].device_time = 1e-3 * ms / B;  // the launch's device time, shared equally
    info[b].run_time = wall_s / B;
    const bool have = x_inc0 != nullptr && upper0[b] < 1.7e308;
    if (o.found) memcpy(x_out + (size_t)b * n, h_inc + (size_t)b * n, sizeof(double) * n);
    else if (have) memcpy(x_out + (size_t)b * n, x_inc0 + (size_t)b * n, sizeof(double) * n);
    iters += o.osqp_iter;
  }
  e->loop_ms += ms;
  e->loop_iters += iters;
  return 0;
}

int miosqp_qp_debug_iterate(miosqp_qp_engine *e, int32_t k, double *x, double *z, double *y) {
  if (!e || k < 0) return MIOSQP_EARG;
  ENTER(e);
  hipLaunchKernelGGL(k_reset_ctrl, dim3(1), dim3(1), 0, e->stream, e->d);
  hipLaunchKernelGGL(k_init_wh, dim3(((e->M > e->n ? e->M : e->n) + 255) / 256), dim3(256), 0, e->stream, e->d);
  if (e->resident) {
    if (k > 0) {
      int rc = launch_resident(e, k, 0, 0);
      if (rc) return rc;
    }
  } else if (e->coop || e->pers) {
    if (k > 0) {
      if (e->coop) launch_coop(e, k, 0, 0);
      else launch_pers(e, k, 0, 0);
      HIPCHK(hipMemcpyAsync(e->h_ctrl, e->d.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
      if (e->h_ctrl->pad == 1) {  // called off, iterates untouched: the same k iterations in the two-kernel form
        reset_registration(e);
        g_err = "cooperative solver: the grid was not co-resident within 100 ms (device shared?), stage 1";
        int rc = leave_coop(e);
        if (rc) return rc;
        hipLaunchKernelGGL(k_reset_ctrl, dim3(1), dim3(1), 0, e->stream, e->d);
        for (int i = 0; i < k; i++) launch_iteration(e);
      } else if (e->h_ctrl->pad) {
        g_err = "cooperative solver: exchange timed out, stage " + std::to_string(e->h_ctrl->pad);
        return MIOSQP_EHIP;
      }
    }
  } else {
    for (int i = 0; i < k; i++) launch_iteration(e);
  }
  HIPCHK(hipStreamSynchronize(e->stream));
  if (x) HIPCHK(hipMemcpy(x, e->d.x, sizeof(double) * e->n, hipMemcpyDeviceToHost));
  if (z) HIPCHK(hipMemcpy(z, e->d.z, sizeof(double) * e->M, hipMemcpyDeviceToHost));
  if (y) HIPCHK(hipMemcpy(y, e->d.y, sizeof(double) * e->M, hipMemcpyDeviceToHost));
  return 0;
}

int miosqp_qp_get_scaling(miosqp_qp_engine *e, double *D, double *E, double *c) {
  if (!e) return MIOSQP_EARG;
  if (D) memcpy(D, e->sc.D.data(), sizeof(double) * e->n);
  if (E) memcpy(E, e->sc.E.data(), sizeof(double) * e->M);
  if (c) *c = e->sc.c;
  return 0;
}

static void kernel_bytes(const miosqp_qp_engine *e, double b[6]) {
  const double n = e->n, M = e->M, np = (double)e->fa.nnz_panel, nt = (double)e->fa.nnz_tail;
  // 12 B per factor entry (value + index), 4 B row pointers, 8 B vectors
  b[0] = np * 12 + (n + 1) * 4 + (M + 3 * n) * 8;          // panel forward: wh in; x, q in; c out
  b[1] = nt * 12 + (n + 1) * 4 + (3 * n) * 8;              // tail forward: c in, d2inv in, ut out
  b[2] = nt * 12 + (n + 1) * 4 + (5 * n) * 8;              // tail backward: ut in, x in; xt, x, dx out
  b[3] = np * 12 + (M + 1) * 4 + (n + 10 * M) * 8;         // panel backward + z/y update
  const double k = e->st.check_termination;
  const double NK = n + M;
  const double rest = 2 * (NK + 1) * 4 + NK * 8 + NK * 20 + (6 * n + 16 * M) * 8 + (2.0 * e->nnzA + e->nnzPtriu) * 12 / k;
  b[4] = 2 * (np + nt) * 12 + rest;
  // what the kernels of the form in use REQUEST per iteration: the dense tail (and, in the product form, the
  // dense G block) carries no index array, so those entries move 8 bytes, not the 12 of the formula above
  b[5] = (e->fold ? 2 * ((double)n * M + nt) * 8 : 2 * (np * 12 + nt * 8)) + rest;
  if (e->pers && e->pp.sinv && e->pp.res_w) b[5] -= n * (n - e->pp.res_c0) * 8;  // columns of S^-1 that stay in LDS
  if (e->pers && e->pp.sinv && e->pp.sym) {  // the tiles on and above the diagonal instead of the whole of S^-1
    double tiles = 0;
    for (int I = 0; I < e->pp.sym_T; I++)
      for (int J = I; J < e->pp.sym_T; J++)
      {
        const int nr = std::min(e->pp.sym_C, e->n - I * e->pp.sym_C), Rw = (nr + PERS_NW - 1) / PERS_NW;
        int streamed = 0;  // (every wave keeps the first sym_res rows of its share in LDS)
        for (int w = 0; w < PERS_NW; w++) streamed += std::max(0, std::min(Rw, nr - w * Rw) - e->pp.sym_res);
        tiles += (double)streamed * std::min(e->pp.sym_C, e->n - J * e->pp.sym_C);
      }
    b[5] += (tiles - 2 * nt) * 8;
  }
}

int miosqp_qp_get_factor_stats(miosqp_qp_engine *e, int64_t *out) {
  if (!e || !out) return MIOSQP_EARG;
  double b[6];
  kernel_bytes(e, b);
  out[8] = (int64_t)b[5];
  out[9] = e->coop_fallbacks;
  out[0] = e->fa.nnz_panel + e->fa.nnz_tail;
  out[1] = e->fa.nnz_panel;
  out[2] = e->n;
  out[3] = (int64_t)b[4];
  out[4] = e->tpr_pv; out[5] = e->tpr_pc; out[6] = e->tpr_tail; out[7] = (e->fold ? 1 : 0) | (e->resident ? 2 : 0) | (e->setup_on_device ? 4 : 0) | (e->coop ? 8 : 0) |
           (e->pers ? 16 : 0) | ((e->pers && e->pp.sinv) ? 32 : 0) | ((e->pers && e->pp.small) ? 64 : 0) |
           ((e->d.coop_nap & 0xff) << 8) | (e->kbp ? (1 << 16) : 0) | (e->guard_tripped ? (1 << 17) : 0) |
           (e->run_launches > 0 ? (1 << 18) : 0);
  return 0;
}

int miosqp_qp_get_rho(miosqp_qp_engine *e, double *rho) {
  if (!e || !rho) return MIOSQP_EARG;
  *rho = e->d.rho;
  return 0;
}

int miosqp_qp_get_inverse_guard(miosqp_qp_engine *e, double *out) {
  if (!e || !out) return MIOSQP_EARG;
  out[0] = e->guard_resid;
  out[1] = e->guard_tol;
  out[2] = e->guard_tripped ? 1.0 : 0.0;
  return 0;
}

int miosqp_qp_get_loop_stats(miosqp_qp_engine *e, double *ms, int64_t *iters, int32_t reset) {
  if (!e) return MIOSQP_EARG;
  if (ms) *ms = e->loop_ms;
  if (iters) *iters = e->loop_iters;
  if (reset) {
    e->loop_ms = 0.0;
    e->loop_iters = 0;
    e->loop_launches = 0;
    e->run_launches = 0;
    e->node_ms.clear();
    e->node_it.clear();
  }
  return 0;
}

// launches of the hosted search's solver kernel since the last reset of the loop statistics: one per node, or -- the
// cooperative grid kept resident (k_coop_run) -- one per miosqp_qp_search_run
int miosqp_qp_get_loop_launches(miosqp_qp_engine *e, int64_t *launches) {
  if (!e || !launches) return MIOSQP_EARG;
  *launches = e->loop_launches;
  return 0;
}

// per node of the hosted search since the last reset of the loop statistics: microseconds of device time per ADMM
// iteration of a node's launch -- minimum, median, maximum over the nodes -- and the number of nodes they are taken over
int miosqp_qp_get_node_stats(miosqp_qp_engine *e, double *us_per_iter_min_med_max, int32_t *nodes) {
  if (!e || !us_per_iter_min_med_max || !nodes) return MIOSQP_EARG;
  const size_t k = std::min(e->node_ms.size(), e->node_it.size());
  std::vector<double> v;
  for (size_t i = 0; i < k; i++)
    if (e->node_it[i] > 0) v.push_back(1e3 * (double)e->node_ms[i] / (double)e->node_it[i]);
  *nodes = (int32_t)v.size();
  us_per_iter_min_med_max[0] = us_per_iter_min_med_max[1] = us_per_iter_min_med_max[2] = 0.0;
  if (v.empty()) return 0;
  std::sort(v.begin(), v.end());
  us_per_iter_min_med_max[0] = v.front();
  us_per_iter_min_med_max[1] = v[v.size() / 2];
  us_per_iter_min_med_max[2] = v.back();
  return 0;
}

// debug: per-block (start, end) wall-clock stamps (100 MHz) of ONE launch of a product-form kernel
int miosqp_qp_debug_timeline(miosqp_qp_engine *e, int32_t which, uint64_t *out, int32_t max_blocks,
                             int32_t *nblocks) {
  if (which == 6) {
    // the persistent stream's own phase clocks (MIOSQP_KBS_PROF=1), summed over the launches since the last call; per
    // workgroup 8 words: shader clocks of thread 0 in {iterations, test, harvest + refill, -}, chunks, -, -, -; then per
    // workgroup 8 more: the phases inside the iterations {forward sweep, reduce + store, barrier, x sweep, x epilogue,
    // constraint tiles, barrier}, -; then per workgroup 16: the phases inside the boundaries {test jobs, row pieces + fold,
    // barrier, members' fold + decision, harvest rows + claim, barrier, table + harvest jobs + fold, prepare, barrier, commit,
    // z jobs + stores, barrier, counters}
    if (!e || !out || !e->kbs_prof || max_blocks < 16 * KBP_WGS) return MIOSQP_EARG;
    ENTER(e);
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(out, e->kbs_prof, sizeof(unsigned long long) * 32 * KBP_WGS, hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(e->kbs_prof, 0, sizeof(unsigned long long) * 32 * KBP_WGS));
    if (nblocks) *nblocks = 16 * KBP_WGS;
    return 0;
  }
  if (!e || !out || which < 0 || which > 5 || (which < 4 && !e->fold) || (which == 2 && !e->coop) || (which == 3 && e->Bcap == 0) ||
      (which == 4 && !e->pers) || (which == 5 && !e->kbp))
    return MIOSQP_EARG;
  ENTER(e);
  unsigned long long *buf = nullptr;
  HIPCHK(hipMalloc((void **)&buf, sizeof(unsigned long long) * 2 * 8192));
  HIPCHK(hipMemsetAsync(buf, 0, sizeof(unsigned long long) * 2 * 8192, e->stream));
  if (which == 5) {
    // every column live for the measurement (the columns of a finished wave are all decided); the caller solves anew
    HIPCHK(hipMemsetAsync(e->d.c_done, 0, sizeof(int) * e->d.Bs, e->stream));
    launch_kbp(e, e->d, e->Bcap / 64, 5);
  } else if (which == 4) {
    launch_pers(e, 20, 0, 0);
  } else if (which == 3) {
    for (int i = 0; i < 5; i++) launch_iteration_b(e, e->Bcap / 64);
  } else {
    for (int i = 0; i < 20; i++) launch_iteration(e);
  }
  Dev saved = e->d;
  e->d.prof = buf;
  // which == 2: 1000 iterations of the cooperative solver; per workgroup 8 words: shader clocks of
  // thread 0 in {reduce, update+publish, gather}, iterations, {test operands+rows, test norms}, tests, -
  // which == 3: the batched forward sweep (kbm_fwd) at full width; per workgroup 8 words (100 MHz stamps of wave 0):
  // start, first loads issued, first operands arrived, sweep done, reduced, stored
  // which == 4: 1000 iterations of the persistent streaming solver, tests every 25; per workgroup 16 words: shader
  // clocks of thread 0 in {sparse forward, forward wait, forward rows, forward epilogue, backward wait, backward rows,
  // backward epilogue (+ sparse backward), tests}, iterations
  // which == 5: 100 lock-step iterations of kbp at full width; per workgroup 16 words: shader clocks of thread 0 in
  // {forward sweep, forward reduce + store, barrier, x sweep, x reduce + epilogue, constraint tiles, barrier}, iterations
  if (which == 5) launch_kbp(e, e->d, e->Bcap / 64, 100, true);
  else if (which == 4) launch_pers(e, 1000, 25, 0);
  else if (which == 3) launch_bd(e, e->Bcap / 64, 0);
  else if (which == 2) launch_coop(e, 1000, 25, 0);
  else if (which == 0) launch_fold_fwd(e); else launch_fold_bwd(e);
  e->d = saved;
  HIPCHK(hipStreamSynchronize(e->stream));
  const int nb = max_blocks < 8192 ? max_blocks : 8192;
  HIPCHK(hipMemcpy(out, buf, sizeof(unsigned long long) * 2 * nb, hipMemcpyDeviceToHost));
  if (nblocks) *nblocks = nb;
  hipFree(buf);
  return 0;
}

// debug: shader cycles and 100 MHz wall ticks recorded by the last LDS-resident launch
int miosqp_qp_debug_clock(miosqp_qp_engine *e, double *cycles, double *ticks) {
  if (!e) return MIOSQP_EARG;
  ENTER(e);
  HIPCHK(hipStreamSynchronize(e->stream));
  Ctrl c;
  HIPCHK(hipMemcpy(&c, e->d.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost));
  *cycles = c.nrm_dy;
  *ticks = c.nrm_dx;
  return 0;
}

// debug counters: 0 = wave compactions performed by solve_batch so far
int64_t miosqp_qp_debug_counter(miosqp_qp_engine *e, int32_t which) {
  if (!e) return -1;
  if (which == 2) {  // whole-chip launches of this device that were ordered behind another stream's (host.inc: ChipTurn)
    ChipTurn *t = chip_of(e);
    if (!t) return -1;
    std::lock_guard<std::mutex> lk(t->mu);
    return t->waits;
  }
  if (which == 3) {  // the control block's call-off / time-out word, after everything queued on the engine's stream
    if (e->device >= 0 && hipSetDevice(e->device) != hipSuccess) return -1;
    int pad = -1;
    if (hipStreamSynchronize(e->stream) != hipSuccess) return -1;
    if (hipMemcpy(&pad, &e->d.ctrl->pad, sizeof pad, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return pad;
  }
  if (which == 4) {  // engines of this device that take turns
    ChipTurn *t = chip_of(e);
    if (!t) return -1;
    std::lock_guard<std::mutex> lk(t->mu);
    return t->users;
  }
  if (which == 5) return (e->pers_capable && e->pp.sinv && e->pp.res_w) ? e->n - e->pp.res_c0 : 0;
  if (which == 6) return (e->pers_capable && e->pp.sinv && e->pp.sym) ? e->pp.sym_tiles : 0;
  if (which == 7) return e->kbs_launches;
  if (which == 8) return e->kbs_chunks_run;
  if (which == 9) return e->graph_chunks_run;
  if (which == 10) return e->run_nap_in_use;
  if (which == 11) return e->run_cal_nodes;
  return which == 0 ? e->compactions : which == 1 ? e->kbp_fallbacks : -1;
}

int miosqp_qp_get_batch_stats(miosqp_qp_engine *e, double *ms, int64_t *batch_iters, int64_t *node_iters,
                              int32_t reset) {
  if (!e) return MIOSQP_EARG;
  if (ms) *ms = e->bloop_ms;
  if (batch_iters) *batch_iters = e->bloop_iters;
  if (node_iters) *node_iters = e->bloop_node_iters;
  if (reset) {
    e->bloop_ms = 0.0;
    e->bloop_iters = e->bloop_node_iters = 0;
  }
  return 0;
}

int miosqp_qp_time_kernel(miosqp_qp_engine *e, int32_t which, int32_t reps, double *usec, double *bytes) {
  if (!e || which < 0 || (which > 4 && which < 10) || which > 15 || reps <= 0 || !usec) return MIOSQP_EARG;
  ENTER(e);
  if (which >= 10 && e->Bcap == 0) {
    g_err = "time_kernel: batched kernels need a prior solve_batch";
    return MIOSQP_EARG;
  }
  const Dev &d = e->d;
  const int ntiles = e->Bcap / 64;
  if (which >= 10) hipLaunchKernelGGL(kb_reset, dim3((d.Bs + 255) / 256), dim3(256), 0, e->stream, d, e->Bcap);
  hipLaunchKernelGGL(k_reset_ctrl, dim3(1), dim3(1), 0, e->stream, d);
  auto one = [&]() {
    if (e->fold && which < 4) {
      if (which == 0) launch_fold_fwd(e);
      if (which == 1) launch_fold_bwd(e);
      return;
    }
    if (e->fold && which >= 10 && which < 14) {
      if (which == 10) launch_bd(e, ntiles, 0);
      if (which == 11) launch_bd(e, ntiles, 1);
      return;
    }
    switch (which) {
      case 10: hipLaunchKernelGGL(kb_panel_fwd, dim3((d.n + 3) / 4, ntiles), dim3(256), 0, e->stream, d); break;
      case 11: hipLaunchKernelGGL(kb_tail_fwd, dim3((d.n + 3) / 4, ntiles), dim3(256), 0, e->stream, d); break;
      case 12: hipLaunchKernelGGL(kb_tail_bwd, dim3((d.n + 3) / 4, ntiles), dim3(256), 0, e->stream, d); break;
      case 13: hipLaunchKernelGGL(kb_panel_bwd, dim3((d.M + 3) / 4, ntiles), dim3(256), 0, e->stream, d); break;
      case 14: launch_iteration_b(e, ntiles); break;
      case 0: DISPATCH_TPR(e->tpr_pv, k_panel_fwd, d.n, e->stream, d); break;
      case 1: DISPATCH_TPR(e->tpr_tail, k_tail_fwd, d.n, e->stream, d); break;
      case 2: DISPATCH_TPR(e->tpr_tail, k_tail_bwd, d.n, e->stream, d); break;
      case 3: DISPATCH_TPR(e->tpr_pc, k_panel_bwd, d.M, e->stream, d); break;
      default: launch_iteration(e); break;
    }
  };
  if (which == 15) {  // `reps` lock-step iterations of the batched sweeps in ONE persistent launch (kbp), every column live
    if (!e->kbp) {
      g_err = "time_kernel: the persistent batched sweeps are not in use on this engine";
      return MIOSQP_EARG;
    }
    launch_kbp(e, d, ntiles, 5);
    HIPCHK(hipEventRecord(e->ev0, e->stream));
    launch_kbp(e, d, ntiles, reps < 200 ? reps : 200);
    HIPCHK(hipEventRecord(e->ev1, e->stream));
    reps = reps < 200 ? reps : 200;
  } else if ((e->coop || e->pers) && which == 4) {  // `reps` iterations of the single-launch solver in ONE launch, no tests
    if (e->coop) launch_coop(e, 5, 0, 0); else launch_pers(e, 5, 0, 0);
    HIPCHK(hipEventRecord(e->ev0, e->stream));
    if (e->coop) launch_coop(e, reps, 0, 0); else launch_pers(e, reps, 0, 0);
    HIPCHK(hipEventRecord(e->ev1, e->stream));
  } else {
    for (int i = 0; i < 5; i++) one();
    HIPCHK(hipEventRecord(e->ev0, e->stream));
    for (int i = 0; i < reps; i++) one();
    HIPCHK(hipEventRecord(e->ev1, e->stream));
  }
  HIPCHK(hipStreamSynchronize(e->stream));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, e->ev0, e->ev1));
  *usec = 1e3 * ms / reps;
  if (bytes) {
    double b[6];
    kernel_bytes(e, b);
    if (which < 10) {
      *bytes = b[which];
      if (e->fold && which < 4)  // forward sweep = panel + tail forward, backward likewise
        *bytes = which == 0 ? b[0] + b[1] : which == 1 ? b[2] + b[3] : 0.0;
    } else {
      // batched: matrix terms once per launch, per-node vector terms times the columns
      const double n = e->n, M = e->M, np = (double)e->fa.nnz_panel, nt = (double)e->fa.nnz_tail, B = e->Bcap;
      const double mat[5] = {np * 12 + (n + 1) * 4, nt * 12 + (n + 1) * 4, nt * 12 + (n + 1) * 4,
                             np * 12 + (M + 1) * 4, 2 * (np + nt) * 12 + 2 * (n + M + 1) * 4 + (n + M) * 12};
      const double vec[5] = {(M + 3 * n) * 8, 3 * n * 8, 5 * n * 8, (n + 10 * M) * 8, (6 * n + 16 * M) * 8};
      const int wb = which == 15 ? 14 : which;
      *bytes = mat[wb - 10] + B * vec[wb - 10];
      if (e->fold && which < 14)
        *bytes = which == 10 ? mat[0] + mat[1] + B * (vec[0] + vec[1])
                 : which == 11 ? mat[2] + mat[3] + B * (vec[2] + vec[3]) : 0.0;
    }
  }
  return 0;
}

}  // extern "C"
