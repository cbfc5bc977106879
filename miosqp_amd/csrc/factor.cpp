// Host-side one-time preparation: equilibration and block LDL^T factor.  See factor.hpp.
#include "factor.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <functional>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <pthread.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace miosqp {

namespace {

struct StageTimer {  // MIOSQP_SETUP_TIMING=1 prints the host setup stages
  bool on = getenv("MIOSQP_SETUP_TIMING") != nullptr;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void lap(const char *what) {
    if (!on) return;
    auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[miosqp setup] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
    t = now;
  }
};

constexpr double kMinScaling = 1e-4, kMaxScaling = 1e4;

inline double clamp_scaling(double v) {
  if (v < kMinScaling) v = 1.0;
  if (v > kMaxScaling) v = kMaxScaling;
  return v;
}

// dynamic-chunk parallel loop over [0, count).  The workers are created once per process and reused: a setup at
// config 5 runs ~80 of these loops, and spawning 32 threads for each cost more than the loops themselves (60 of the
// 85 ms of the equilibration).  One loop at a time (a mutex serialises concurrent setups); a loop started from inside a
// worker runs serially.
class WorkerPool {
 public:
  static WorkerPool &get() {
    static WorkerPool p;
    return p;
  }
  int size() const { return (int)workers_.size(); }
  void run(int64_t count, const std::function<void(int64_t)> &fn) {
    if (workers_.empty() || in_worker() || forked().load()) {  // (a forked child has no workers: serial)
      for (int64_t i = 0; i < count; i++) fn(i);
      return;
    }
    std::lock_guard<std::mutex> region(region_);
    {
      std::lock_guard<std::mutex> lk(m_);
      fn_ = &fn;
      count_ = count;
      next_.store(0);
      busy_ = (int)workers_.size();
      epoch_++;
    }
    cv_.notify_all();
    work();  // the caller takes chunks too
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [&] { return busy_ == 0; });
    fn_ = nullptr;
  }

 private:
  static std::atomic<bool> &forked() {
    static std::atomic<bool> f{false};
    return f;
  }
  WorkerPool() {
    pthread_atfork(nullptr, nullptr, []() { forked().store(true); });
    unsigned hw = std::thread::hardware_concurrency();
    const int nthreads = (int)std::min<unsigned>(hw ? hw : 1, 32);
    for (int t = 0; t + 1 < nthreads; t++)
      workers_.emplace_back([this]() {
        flag() = true;
        uint64_t seen = 0;
        for (;;) {
          {
            std::unique_lock<std::mutex> lk(m_);
            cv_.wait(lk, [&] { return stop_ || epoch_ != seen; });
            if (stop_) return;
            seen = epoch_;
          }
          work();
          std::lock_guard<std::mutex> lk(m_);
          if (--busy_ == 0) done_.notify_all();
        }
      });
  }
  ~WorkerPool() {
    if (forked().load()) {  // the threads do not exist in this process
      for (auto &th : workers_) th.detach();
      return;
    }
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto &th : workers_) th.join();
  }
  static bool &flag() {
    static thread_local bool f = false;
    return f;
  }
  static bool in_worker() { return flag(); }
  void work() {
    for (;;) {
      const int64_t i = next_.fetch_add(1);
      if (i >= count_) break;
      (*fn_)(i);
    }
  }
  std::vector<std::thread> workers_;
  std::mutex m_, region_;
  std::condition_variable cv_, done_;
  const std::function<void(int64_t)> *fn_ = nullptr;
  int64_t count_ = 0;
  std::atomic<int64_t> next_{0};
  int busy_ = 0;
  uint64_t epoch_ = 0;
  bool stop_ = false;
};

void parallel_chunks(int64_t count, int64_t work_per_item, const std::function<void(int64_t)> &fn) {
  if (count < 2 || count * work_per_item < (int64_t)1 << 22) {
    for (int64_t i = 0; i < count; i++) fn(i);
    return;
  }
  WorkerPool::get().run(count, fn);
}

// contiguous column ranges [lo, hi) for the parallel loops of the equilibration (max and element-wise
// products only: the results do not depend on the split)
constexpr int kColChunks = 32;
inline void col_range(int n, int64_t c, int &lo, int &hi) {
  lo = (int)((int64_t)n * c / kColChunks);
  hi = (int)((int64_t)n * (c + 1) / kColChunks);
}

// column inf-norms of a symmetric matrix given by its upper triangle
void sym_col_norms(int n, const std::vector<int> &Pp, const std::vector<int> &Pi,
                   const std::vector<double> &Px, std::vector<double> &out) {
  const int64_t nnz = Pp[n];
  if (nnz < ((int64_t)1 << 20)) {
    std::fill(out.begin(), out.end(), 0.0);
    for (int j = 0; j < n; j++)
      for (int p = Pp[j]; p < Pp[j + 1]; p++) {
        double a = std::fabs(Px[p]);
        out[j] = std::max(out[j], a);
        out[Pi[p]] = std::max(out[Pi[p]], a);
      }
    return;
  }
  // every chunk of columns fills its own array (an entry touches its column and its row), then a max-merge
  std::vector<double> part((size_t)kColChunks * n, 0.0);
  parallel_chunks(kColChunks, nnz / kColChunks + 1, [&](int64_t c) {
    int lo, hi;
    col_range(n, c, lo, hi);
    double *o = part.data() + (size_t)c * n;
    for (int j = lo; j < hi; j++)
      for (int p = Pp[j]; p < Pp[j + 1]; p++) {
        double a = std::fabs(Px[p]);
        o[j] = std::max(o[j], a);
        o[Pi[p]] = std::max(o[Pi[p]], a);
      }
  });
  parallel_chunks(kColChunks, (int64_t)n, [&](int64_t c) {
    int lo, hi;
    col_range(n, c, lo, hi);
    for (int j = lo; j < hi; j++) {
      double m = 0.0;
      for (int k = 0; k < kColChunks; k++) m = std::max(m, part[(size_t)k * n + j]);
      out[j] = m;
    }
  });
}

// rows -> padded compressed rows
struct Triplet {
  int col;
  double v0, v1;
};

void pack_rows(int rows, int cols, std::vector<std::vector<Triplet>> &r, PCsr &out,
               std::vector<double> *second) {
  out.rows = rows;
  out.cols = cols;
  out.ptr.assign(rows + 1, 0);
  out.nnz = 0;
  int64_t total = 0;
  for (int i = 0; i < rows; i++) {
    out.nnz += (int64_t)r[i].size();
    total += ((int64_t)r[i].size() + 1) & ~(int64_t)1;
  }
  out.idx.assign(total, 0);
  out.val.assign(total, 0.0);
  if (second) second->assign(total, 0.0);
  int64_t k = 0;
  for (int i = 0; i < rows; i++) {
    out.ptr[i] = (int)k;
    std::sort(r[i].begin(), r[i].end(), [](const Triplet &a, const Triplet &b) { return a.col < b.col; });
    for (const Triplet &t : r[i]) {
      out.idx[k] = t.col;
      out.val[k] = t.v0;
      if (second) (*second)[k] = t.v1;
      k++;
    }
    if (k & 1) {  // pad: zero value, repeat a valid column
      out.idx[k] = r[i].empty() ? 0 : r[i].back().col;
      k++;
    }
  }
  out.ptr[rows] = (int)k;
}

}  // namespace

bool scale_problem(int n, int M, const int32_t *Pp, const int32_t *Pi, const double *Px,
                   const int32_t *Ap, const int32_t *Ai, const double *Ax, const double *q,
                   int passes, Scaled &s, const RuizOps *ops) {
  StageTimer tm;
  s.n = n;
  s.M = M;
  s.Pp.assign(n + 1, 0);
  s.Pi.clear();
  s.Px.clear();
  for (int j = 0; j < n; j++) {
    s.Pp[j] = (int)s.Pi.size();
    for (int p = Pp[j]; p < Pp[j + 1]; p++)
      if (Pi[p] <= j) {
        s.Pi.push_back(Pi[p]);
        s.Px.push_back(Px[p]);
      }
  }
  s.Pp[n] = (int)s.Pi.size();
  s.Ap.assign(Ap, Ap + n + 1);
  s.Ai.assign(Ai, Ai + Ap[n]);
  s.Ax.assign(Ax, Ax + Ap[n]);
  s.q.assign(q, q + n);
  s.D.assign(n, 1.0);
  s.E.assign(M, 1.0);
  s.c = 1.0;
  std::vector<double> dt(n), et(M);
  const bool dev = ops && passes > 0;
  bool dev_ok = true;
  if (dev) dev_ok = ops->begin(ops->ctx, n, M, s.Pp.data(), s.Pi.data(), s.Px.data(), s.Ap.data(), s.Ai.data(), s.Ax.data()) == 0;
  for (int pass = 0; pass < passes && dev_ok; pass++) {
    if (dev) {
      if (ops->norms(ops->ctx, dt.data(), et.data(), 1)) { dev_ok = false; break; }
    } else {
    sym_col_norms(n, s.Pp, s.Pi, s.Px, dt);
    std::fill(et.begin(), et.end(), 0.0);
    for (int j = 0; j < n; j++)
      for (int p = s.Ap[j]; p < s.Ap[j + 1]; p++) {
        double a = std::fabs(s.Ax[p]);
        dt[j] = std::max(dt[j], a);
        et[s.Ai[p]] = std::max(et[s.Ai[p]], a);
      }
    }
    for (int j = 0; j < n; j++) dt[j] = 1.0 / std::sqrt(clamp_scaling(dt[j]));
    for (int i = 0; i < M; i++) et[i] = 1.0 / std::sqrt(clamp_scaling(et[i]));
    if (dev) {
      if (ops->scale(ops->ctx, dt.data(), et.data())) { dev_ok = false; break; }
    } else {
    parallel_chunks(kColChunks, (int64_t)s.Pp[n] / kColChunks + 1, [&](int64_t c) {
      int lo, hi;
      col_range(n, c, lo, hi);
      for (int j = lo; j < hi; j++)
        for (int p = s.Pp[j]; p < s.Pp[j + 1]; p++) s.Px[p] *= dt[j] * dt[s.Pi[p]];
    });
    for (int j = 0; j < n; j++)
      for (int p = s.Ap[j]; p < s.Ap[j + 1]; p++) s.Ax[p] *= dt[j] * et[s.Ai[p]];
    }
    for (int j = 0; j < n; j++) {
      s.q[j] *= dt[j];
      s.D[j] *= dt[j];
    }
    for (int i = 0; i < M; i++) s.E[i] *= et[i];
    // cost normalisation
    if (dev) {
      if (ops->norms(ops->ctx, dt.data(), nullptr, 0)) { dev_ok = false; break; }
    } else {
      sym_col_norms(n, s.Pp, s.Pi, s.Px, dt);
    }
    double mean = 0;
    for (int j = 0; j < n; j++) mean += dt[j];
    mean /= n;
    double nq = 0;
    for (int j = 0; j < n; j++) nq = std::max(nq, std::fabs(s.q[j]));
    double ct = 1.0 / clamp_scaling(std::max(mean, clamp_scaling(nq)));
    if (dev) {
      if (ops->scale_cost(ops->ctx, ct)) { dev_ok = false; break; }
    } else {
    parallel_chunks(kColChunks, (int64_t)s.Pp[n] / kColChunks + 1, [&](int64_t c) {
      int lo, hi;
      col_range(n, c, lo, hi);
      for (int p = s.Pp[lo]; p < s.Pp[hi]; p++) s.Px[p] *= ct;
    });
    }
    for (double &v : s.q) v *= ct;
    s.c *= ct;
  }
  if (dev) {
    if (ops->end(ops->ctx, s.Px.data(), s.Ax.data())) dev_ok = false;
    if (!dev_ok) return false;  // the caller repeats the call on the host
  }
  tm.lap(dev ? "copy + Ruiz equilibration (device)" : "copy + Ruiz equilibration");
  s.Dinv.resize(n);
  s.Einv.resize(M);
  for (int j = 0; j < n; j++) s.Dinv[j] = 1.0 / s.D[j];
  for (int i = 0; i < M; i++) s.Einv[i] = 1.0 / s.E[i];
  s.cinv = 1.0 / s.c;
  return true;
}

bool build_factor(const Scaled &s, const int32_t *Pp_raw, const int32_t *Pi_raw,
                  const double *Px_raw, double rho, double sigma, Factor &f, std::string &err,
                  DenseLdlInv accel, void *accel_ctx, bool reuse_rows) {
  const int n = s.n, M = s.M;
  StageTimer tm;
  f.n = n;
  f.M = M;
  f.rho = rho;
  f.sigma = sigma;
  f.ld = (n + 15) & ~15;  // multiples of 16: the batched matrix-core tiles read whole 16-deep chunks of a row
  const int ld = f.ld;

  if (reuse_rows) {
    // (a padding entry is 0.0 in a fresh build, not -rho * 0.0 = -0.0)
    for (size_t k = 0; k < f.At_val.size(); k++) f.panel_by_var.val[k] = f.At_val[k] == 0.0 ? 0.0 : -rho * f.At_val[k];
    for (size_t k = 0; k < f.A_val.size(); k++) f.panel_by_con.val[k] = f.A_val[k] == 0.0 ? 0.0 : -rho * f.A_val[k];
  }
  // ---- panel: rows of Abar^T (per variable) and rows of Abar (per constraint) ----------
  if (!reuse_rows) {
    std::vector<std::vector<Triplet>> byvar(n), bycon(M);
    for (int i = 0; i < n; i++) byvar[i].reserve(s.Ap[i + 1] - s.Ap[i]);
    for (int i = 0; i < n; i++)
      for (int p = s.Ap[i]; p < s.Ap[i + 1]; p++) {
        int j = s.Ai[p];
        double a = s.Ax[p];
        byvar[i].push_back({j, -rho * a, a});
        bycon[j].push_back({i, -rho * a, a});
      }
    pack_rows(n, M, byvar, f.panel_by_var, &f.At_val);
    pack_rows(M, n, bycon, f.panel_by_con, &f.A_val);
    f.nnz_panel = f.panel_by_var.nnz;
  }
  tm.lap("panel rows");
  // ---- symmetric matrices by row (counting transpose of the upper triangle; columns ascend) ----
  if (!reuse_rows) {
    // Row r of the full symmetric matrix = [column r of the upper triangle: entries (i, r), i <= r, as columns i]
    // followed by [row r of the strict upper triangle: entries (r, j), j > r] -- both ascending, so the row comes out
    // sorted.  The first part is copied per row; the second is a transpose of the strict upper triangle done in
    // kColChunks column chunks with per-chunk histograms and cursors (no atomics, the result does not depend on the
    // number of threads).
    auto sym_rows = [&](const int *Pp_, const int *Pi_, const double *Px_, bool upper_only_input, PCsr &out) {
      (void)upper_only_input;
      out.rows = out.cols = n;
      const int64_t nnz_in = Pp_[n];
      std::vector<int> colcnt(n, 0);                           // entries of column r with i <= r
      std::vector<int> hist((size_t)kColChunks * n, 0);        // [chunk][row]: strict upper entries of that row in the chunk
      parallel_chunks(kColChunks, nnz_in / kColChunks + 1, [&](int64_t c) {
        int lo, hi;
        col_range(n, c, lo, hi);
        int *h = &hist[(size_t)c * n];
        for (int j = lo; j < hi; j++) {
          int cc = 0;
          for (int p = Pp_[j]; p < Pp_[j + 1]; p++) {
            const int i = Pi_[p];
            if (i > j) continue;  // only the upper triangle is read
            cc++;
            if (i != j) h[i]++;
          }
          colcnt[j] = cc;
        }
      });
      out.ptr.assign(n + 1, 0);
      out.nnz = 0;
      std::vector<int> cnt(n);
      for (int i = 0; i < n; i++) {
        int c2 = colcnt[i];
        for (int c = 0; c < kColChunks; c++) {
          const int v = hist[(size_t)c * n + i];
          hist[(size_t)c * n + i] = c2;  // becomes the chunk's cursor inside row i
          c2 += v;
        }
        cnt[i] = c2;
        out.nnz += c2;
        out.ptr[i + 1] = out.ptr[i] + ((c2 + 1) & ~1);
      }
      out.idx.assign(out.ptr[n], 0);
      out.val.assign(out.ptr[n], 0.0);
      parallel_chunks(kColChunks, nnz_in / kColChunks + 1, [&](int64_t c) {
        int lo, hi;
        col_range(n, c, lo, hi);
        int *cur = &hist[(size_t)c * n];
        for (int j = lo; j < hi; j++) {
          int w = out.ptr[j];  // first part of row j: its own column
          for (int p = Pp_[j]; p < Pp_[j + 1]; p++) {
            const int i = Pi_[p];
            if (i > j) continue;
            out.idx[w] = i;
            out.val[w++] = Px_[p];
            if (i != j) {  // second part of row i
              const int q = out.ptr[i] + cur[i]++;
              out.idx[q] = j;
              out.val[q] = Px_[p];
            }
          }
        }
      });
      for (int i = 0; i < n; i++) {
        const int e = out.ptr[i] + cnt[i];
        if (e < out.ptr[i + 1]) out.idx[e] = cnt[i] ? out.idx[e - 1] : 0;  // zero-valued pad
      }
    };
    sym_rows(s.Pp.data(), s.Pi.data(), s.Px.data(), true, f.Pbar);
    sym_rows(Pp_raw, Pi_raw, Px_raw, false, f.Praw);
  }
  tm.lap("P rows");
  // ---- Schur complement S = Pbar + sigma I + rho Abar^T Abar (dense, lower, row-major) -----
  DenseAccelCtx *actx_s = static_cast<DenseAccelCtx *>(accel_ctx);
  bool schur_dev = accel && actx_s && actx_s->schur_on_device && (size_t)n * sizeof(double) <= 150 * 1024;
  if (schur_dev) {
    // the device scatter of Pbar's columns needs distinct row indices within a column
    for (int j = 0; j < n && schur_dev; j++)
      for (int p = s.Pp[j] + 1; p < s.Pp[j + 1]; p++)
        if (s.Pi[p] <= s.Pi[p - 1]) { schur_dev = false; break; }
  }
  std::vector<double> S;
  if (schur_dev) {
    actx_s->rho = rho; actx_s->sigma = sigma; actx_s->M = M;
    actx_s->Pp = s.Pp.data(); actx_s->Pi = s.Pi.data(); actx_s->Px = s.Px.data(); actx_s->nnzP = (int64_t)s.Pi.size();
    actx_s->Ap = s.Ap.data(); actx_s->Ai = s.Ai.data(); actx_s->Ax = s.Ax.data(); actx_s->nnzA = (int64_t)s.Ai.size();
    actx_s->Rptr = f.panel_by_con.ptr.data(); actx_s->Ridx = f.panel_by_con.idx.data(); actx_s->Rval = f.A_val.data();
    actx_s->nnzR = (int64_t)f.panel_by_con.idx.size();
  } else if (actx_s) {
    actx_s->schur_on_device = 0;
  }
  auto assemble_host = [&]() {
  S.assign((size_t)n * ld, 0.0);
  for (int j = 0; j < n; j++)
    for (int p = s.Pp[j]; p < s.Pp[j + 1]; p++) S[(size_t)j * ld + s.Pi[p]] += s.Px[p];  // (j >= i)
  for (int j = 0; j < n; j++) S[(size_t)j * ld + j] += sigma;
  {
    // accumulate rho * a_r a_r^T over constraint rows r, parallel over output rows of S:
    // S[i1][i2] += rho * sum_r A[r][i1] A[r][i2].  Use column form: for variable i1 (col of A)
    // scatter into a dense accumulator through the rows it touches.
    const PCsr &R = f.panel_by_con;  // rows of Abar with A_val
    parallel_chunks(n, (int64_t)f.nnz_panel, [&](int64_t i1) {
      double *out = &S[(size_t)i1 * ld];
      for (int p = s.Ap[i1]; p < s.Ap[i1 + 1]; p++) {
        int r = s.Ai[p];
        double w = rho * s.Ax[p];
        for (int k = R.ptr[r]; k < R.ptr[r + 1]; k++) {
          int i2 = R.idx[k];
          if (i2 > i1) break;  // row sorted by column; padding repeats the last column
          out[i2] = std::fma(w, f.A_val[k], out[i2]);  // (explicitly fused: the device assembly does the same)
        }
      }
    });
    // padding entries carry value 0, so a repeated last column adds nothing
  }
  };
  if (!schur_dev) assemble_host();
  tm.lap("Schur complement");
  if (accel) {
    // dense LDL^T, triangular inverse and transpose on the device
    std::vector<double> dd(n);
    const DenseAccelCtx *actx = static_cast<const DenseAccelCtx *>(accel_ctx);
    const bool keep = actx && actx->keep_on_device;
    f.Linv.clear();
    f.LinvT.clear();
    if (!keep) {
      f.Linv.assign((size_t)n * ld, 0.0);
      f.LinvT.assign((size_t)n * ld, 0.0);
    }
    const int rc = accel(n, ld, schur_dev ? nullptr : S.data(), dd.data(), keep ? nullptr : f.Linv.data(),
                         keep ? nullptr : f.LinvT.data(), accel_ctx);
    if (rc == 1) {
      err = "KKT factorisation: non-positive pivot in the reduced Hessian (P not PSD?)";
      return false;
    }
    if (rc == 0) {
      f.d2inv.resize(n);
      for (int j = 0; j < n; j++) f.d2inv[j] = 1.0 / dd[j];
      f.nnz_tail = (int64_t)n * (n - 1) / 2;
      tm.lap("device LDL^T + inverse");
      return true;
    }
    // device error: fall through to the host path
    if (S.empty()) assemble_host();
  }
  // ---- blocked right-looking LDL^T of S (in place: strict lower = L22, diag = D22) ------
  std::vector<double> d(n);
  const int nb = 64;
  std::vector<double> W;  // scaled block columns
  bool ok = true;
  for (int jb = 0; jb < n && ok; jb += nb) {
    int je = std::min(n, jb + nb), w = je - jb;
    // diagonal block, unblocked
    for (int j = jb; j < je; j++) {
      double *Sj = &S[(size_t)j * ld];
      double dj = Sj[j];
      for (int k = jb; k < j; k++) dj -= Sj[k] * Sj[k] * d[k];
      if (!(dj > 0.0) || !std::isfinite(dj)) {
        ok = false;
        err = "KKT factorisation: non-positive pivot in the reduced Hessian (P not PSD?)";
        break;
      }
      d[j] = dj;
      for (int i = j + 1; i < je; i++) {
        double *Si = &S[(size_t)i * ld];
        double v = Si[j];
        for (int k = jb; k < j; k++) v -= Si[k] * d[k] * Sj[k];
        Si[j] = v / dj;
      }
    }
    if (!ok) break;
    if (je >= n) break;
    // panel below the diagonal block
    parallel_chunks(n - je, (int64_t)w * w, [&](int64_t ii) {
      double *Si = &S[(size_t)(je + ii) * ld];
      for (int j = jb; j < je; j++) {
        const double *Sj = &S[(size_t)j * ld];
        double v = Si[j];
        for (int k = jb; k < j; k++) v -= Si[k] * d[k] * Sj[k];
        Si[j] = v / d[j];
      }
    });
    // trailing update: S[i][j] -= sum_k L[i][k] d[k] L[j][k],  je <= j <= i
    W.assign((size_t)(n - je) * nb, 0.0);
    for (int i = je; i < n; i++)
      for (int k = 0; k < w; k++) W[(size_t)(i - je) * nb + k] = S[(size_t)i * ld + jb + k] * d[jb + k];
    parallel_chunks(n - je, (int64_t)(n - je) * w / 2 + 1, [&](int64_t ii) {
      int i = je + (int)ii;
      double *Si = &S[(size_t)i * ld];
      const double *Wi = &W[(size_t)ii * nb];
      for (int j = je; j <= i; j++) {
        const double *Lj = &S[(size_t)j * ld + jb];
        double acc = 0;
        for (int k = 0; k < w; k++) acc += Wi[k] * Lj[k];
        Si[j] -= acc;
      }
    });
  }
  if (!ok) return false;
  tm.lap("dense LDL^T");
  f.d2inv.resize(n);
  for (int j = 0; j < n; j++) f.d2inv[j] = 1.0 / d[j];
  f.nnz_tail = (int64_t)n * (n - 1) / 2;
  // ---- Linv = L22^-1 (unit lower).  Columns are independent: X[i][j] depends only on
  //      X[k][j], k < i, so column chunks are computed in parallel with no synchronisation.
  f.Linv.assign((size_t)n * ld, 0.0);
  {
    const int cw = 16;
    int nchunks = (n + cw - 1) / cw;
    double *X = f.Linv.data();
    parallel_chunks(nchunks, (int64_t)n * n / 4 + 1, [&](int64_t ch) {
      int c0 = (int)ch * cw, c1 = std::min(n, c0 + cw);
      double acc[cw];
      for (int i = c0 + 1; i < n; i++) {
        const double *Li = &S[(size_t)i * ld];
        int hi = std::min(c1, i);  // columns c0 .. hi-1 of row i are below the diagonal
        for (int t = 0; t < cw; t++) acc[t] = 0;
        // X[i][j] = -L[i][j] - sum_{k=j+1}^{i-1} L[i][k] X[k][j]
        for (int k = c0 + 1; k < i; k++) {
          double lik = Li[k];
          const double *Xk = &X[(size_t)k * ld];
          int jh = std::min(hi, k);  // X[k][j] nonzero (strict lower) only for j < k
          for (int j = c0; j < jh; j++) acc[j - c0] += lik * Xk[j];
        }
        double *Xi = &X[(size_t)i * ld];
        for (int j = c0; j < hi; j++) Xi[j] = -Li[j] - acc[j - c0];
      }
    });
  }
  tm.lap("triangular inverse");
  f.LinvT.assign((size_t)n * ld, 0.0);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < i; j++) f.LinvT[(size_t)j * ld + i] = f.Linv[(size_t)i * ld + j];
  tm.lap("transpose");
  return true;
}

void build_folded(const Factor &f, Folded &o) {
  const int n = f.n, M = f.M, ld = f.ld;
  o.n = n;
  o.M = M;
  o.ldf = (M + n + 15) & ~15;
  o.ldn = (n + 15) & ~15;
  o.rows.assign((size_t)n * o.ldf, 0.0);
  o.GmT.assign((size_t)(M > 0 ? M : 1) * o.ldn, 0.0);
  const PCsr &R = f.panel_by_var;  // rows of L21
  // row i of -G = -( L21[i][:] + sum_{j<i} Linv[i][j] L21[j][:] ), accumulated densely
  parallel_chunks(n, (int64_t)n * 200 + 1, [&](int64_t ii) {
    const int i = (int)ii;
    double *g = &o.rows[(size_t)i * o.ldf];
    for (int k = R.ptr[i]; k < R.ptr[i + 1]; k++) g[R.idx[k]] -= R.val[k];
    const double *Li = &f.Linv[(size_t)i * ld];
    for (int j = 0; j < i; j++) {
      const double w = Li[j];
      if (w == 0.0) continue;
      for (int k = R.ptr[j]; k < R.ptr[j + 1]; k++) g[R.idx[k]] -= w * R.val[k];
    }
    for (int j = 0; j < i; j++) g[M + j] = Li[j];
  });
  for (int i = 0; i < n; i++)
    for (int j = 0; j < M; j++) o.GmT[(size_t)j * o.ldn + i] = o.rows[(size_t)i * o.ldf + j];
  // dense Abar, Abar^T, Pbar
  o.ldm = (M + 15) & ~15;
  o.Ad.assign((size_t)(M > 0 ? M : 1) * o.ldn, 0.0);
  o.Atd.assign((size_t)n * o.ldm, 0.0);
  o.Pd.assign((size_t)n * o.ldn, 0.0);
  const PCsr &C = f.panel_by_con;
  for (int j = 0; j < M; j++)
    for (int k = C.ptr[j]; k < C.ptr[j + 1]; k++)
      if (f.A_val[k] != 0.0) {
        o.Ad[(size_t)j * o.ldn + C.idx[k]] = f.A_val[k];
        o.Atd[(size_t)C.idx[k] * o.ldm + j] = f.A_val[k];
      }
  for (int i = 0; i < n; i++)
    for (int k = f.Pbar.ptr[i]; k < f.Pbar.ptr[i + 1]; k++)
      if (f.Pbar.val[k] != 0.0) o.Pd[(size_t)i * o.ldn + f.Pbar.idx[k]] = f.Pbar.val[k];
}

}  // namespace miosqp
