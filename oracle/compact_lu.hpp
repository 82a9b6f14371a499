// ORACLE-SIDE MODEL -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Sequential restatement of the structure-exploiting LU that the device runs for the bases of
// LCP::lcp_lemke (moby_amd/csrc/mh_lu_compact.h).  It exists so that tests/test_oracle_compact_lu.py can hold
// the ALGORITHM to oracle::lu_solve -- dgesv = dgetf2 + dgetrs on the basis as LCP.cpp:837-838 assembles it --
// on the CPU, before and independently of any device run: same pivots, same factors, same solution, bit for
// bit up to the sign of a zero.
//
// The matrix is given by columns, in POSITION order (position = column index of the assembled basis):
//   kind[c] == CL_UNIT : the column is -e_r with r = idx[c]      (a slack variable of Lemke's basis)
//   kind[c] == CL_DENSE: the column is dense[:, idx[c]]          (a column of M, or the artificial column)
//
// What dgetf2 does to such a matrix, and what is therefore skipped (every skipped operation is a - l*u with
// l == 0 or u == 0 and the other factor finite, which leaves a unchanged):
//   * a unit column whose row r has not been taken as a pivot row yet: its pivot is the -1 itself, every
//     multiplier is 0, no other column changes ("trivial step").  If r sits at position c (its own row is
//     still where it started: "home") not even a row exchange happens.
//   * a unit column whose row r WAS taken by an earlier dense step k: from step k on it is the multiplier
//     column of step k and further elimination steps change it ("fill-in").  It is materialised as the dense
//     column -e_r at the moment step k picks r; every earlier step saw a zero in its pivot row.
//   * rows are never exchanged physically: pos_of_row / row_at_pos carry dgetf2's row order (it decides ties
//     of the pivot search: idamax returns the FIRST maximum in position order), pivstep[row] the step that
//     took the row.
// Dense steps are processed in panels of up to NB columns, right-looking: the in-panel elimination keeps the
// panel's columns current, the other dense columns to the right (and the right-hand side, which is just one
// more column: dgetrs' forward substitution performs the same subtractions) receive the panel's updates
// afterwards, per element in ascending step order, each product rounded on its own -- the order of dgetf2.
// A panel ends early when one of its steps creates a fill-in column that lies between its remaining columns.
//
// Returns LAPACK's info (0, or j > 0 for an exactly zero pivot at step j - 1), or CL_FALLBACK when a
// non-finite value was met: the skips above are then not exact and the caller has to run the dense routine.
#ifndef ORACLE_COMPACT_LU_HPP
#define ORACLE_COMPACT_LU_HPP
#include <cfloat>
#include <cmath>
#include <vector>

namespace oracle {

enum { CL_UNIT = 0, CL_DENSE = 1 };
enum { CL_FALLBACK = -1 };

struct CompactLuStats { int dense_steps = 0, fill_ins = 0, panels = 0, truncated_panels = 0, nonhome = 0; };

inline int lu_solve_compact(int n, const int* kind, const int* idx, const double* dense, int ld_dense, double* b,
                            int NB = 8, int* ipiv_rows = nullptr, CompactLuStats* stats = nullptr)
{
  enum { HOME = 0, NONHOME = 1, DENSE = 2 };
  const int INF = 0x7fffffff;
  auto fin = [](double x) { return std::fabs(x) <= DBL_MAX; };
  std::vector<int> ptype(n), colslot(n, -1), slackpos(n, -1), pos_of_row(n), row_at_pos(n), pivstep(n, INF), rowofstep(n, -1);
  std::vector<double> D;                       // n x nslots, column-major, rows by ORIGINAL row index
  std::vector<int> slot_pos;
  std::vector<int> dpos, drow, dslot;          // the dense steps, in order
  CompactLuStats st;
  for (int i = 0; i < n; i++) { pos_of_row[i] = i; row_at_pos[i] = i; }
  for (int c = 0; c < n; c++) {
    if (kind[c] == CL_UNIT) {
      const int r = idx[c];
      slackpos[r] = c;
      ptype[c] = (r == c) ? HOME : NONHOME;
      if (r == c) { pivstep[c] = c; rowofstep[c] = c; } else st.nonhome++;
    } else {
      ptype[c] = DENSE; colslot[c] = (int)slot_pos.size(); slot_pos.push_back(c);
      D.insert(D.end(), dense + (size_t)ld_dense * idx[c], dense + (size_t)ld_dense * idx[c] + n);
    }
  }
  auto exchange = [&](int c, int r) {          // the row exchange of step c when it takes row r
    const int jp = pos_of_row[r];
    if (jp != c) { const int displaced = row_at_pos[c]; row_at_pos[jp] = displaced; pos_of_row[displaced] = jp; row_at_pos[c] = r; pos_of_row[r] = c; }
  };
  int cur = 0;
  std::vector<double> P((size_t)n * NB), Lpp((size_t)NB * NB), u(NB);
  std::vector<int> pcols, freerows, reachv(n), rowsv;
  while (true) {
    pcols.clear();
    for (int p = cur; p < n && (int)pcols.size() < NB; p++) if (ptype[p] == DENSE) pcols.push_back(p);
    if (pcols.empty()) {
      for (int p = cur; p < n; p++) if (ptype[p] == NONHOME) { const int r = idx[p]; exchange(p, r); pivstep[r] = p; rowofstep[p] = r; }
      break;
    }
    st.panels++;
    const int npc = (int)pcols.size(), last = pcols.back(), first_step = (int)dpos.size();
    for (int jj = 0; jj < npc; jj++) for (int i = 0; i < n; i++) P[i + (size_t)n * jj] = D[i + (size_t)n * colslot[pcols[jj]]];
    int nbk = 0; bool trunc = false;
    int p = cur;
    for (; p <= last && !trunc; p++) {
      if (ptype[p] == HOME) continue;
      if (ptype[p] == NONHOME) { const int r = idx[p]; exchange(p, r); pivstep[r] = p; rowofstep[p] = r; continue; }
      const int jj = nbk;                                          // pcols[jj] == p
      // idamax over the rows at positions >= p: first maximum in position order
      double amax = -1.0; int bpos = INF;
      for (int i = 0; i < n; i++) if (pivstep[i] > p) {
        const double a = std::fabs(P[i + (size_t)n * jj]);
        if (!fin(a)) return CL_FALLBACK;
        if (a > amax || (a == amax && pos_of_row[i] < bpos)) { amax = a; bpos = pos_of_row[i]; }
      }
      if (!(amax != 0.0)) return p + 1;
      const int r = row_at_pos[bpos];
      exchange(p, r);
      pivstep[r] = p; rowofstep[p] = r;
      dpos.push_back(p); drow.push_back(r); dslot.push_back(colslot[p]);
      const int sp = slackpos[r];
      if (sp > p) {                                                // the slack column of row r lies to the right: fill-in
        ptype[sp] = DENSE; colslot[sp] = (int)slot_pos.size(); slot_pos.push_back(sp);
        D.insert(D.end(), (size_t)n, 0.0); D[D.size() - n + r] = -1.0;
        st.fill_ins++;
        if (sp < last) { trunc = true; st.truncated_panels++; }
      }
      const double piv = P[r + (size_t)n * jj];
      const bool big = std::fabs(piv) >= DBL_MIN;
      const double rcp = 1.0 / piv;
      // (column by column: the same per-element operations in an order the CPU's caches like)
      freerows.clear();
      for (int i = 0; i < n; i++) if (pivstep[i] > p) freerows.push_back(i);
      { double* pj = &P[(size_t)n * jj];
        for (int i : freerows) { double l = pj[i]; l = big ? l * rcp : l / piv; if (!fin(l)) return CL_FALLBACK; pj[i] = l; }
        for (int c2 = jj + 1; c2 < npc; c2++) {
          double* pc = &P[(size_t)n * c2]; const double uu = pc[r];
          if (uu != 0.0) for (int i : freerows) pc[i] = lu_upd(pc[i], pj[i], uu);
        } }
      nbk++;
    }
    cur = p;
    for (int jj = 0; jj < npc; jj++) for (int i = 0; i < n; i++) D[i + (size_t)n * colslot[pcols[jj]]] = P[i + (size_t)n * jj];
    // the multipliers of the pivot rows among themselves, and how many of the panel's steps reach each row
    for (int s = 0; s < nbk; s++) for (int t = 0; t < s; t++) Lpp[s + (size_t)NB * t] = P[drow[first_step + s] + (size_t)n * t];
    auto reach = [&](int i) { int c = 0; while (c < nbk && dpos[first_step + c] < pivstep[i]) c++; return c; };
    const int plast = dpos.back();
    // per row: how many of the panel's steps reach it (0 for the panel's own pivot rows, whose prefix is the U12 part)
    rowsv.clear();
    for (int i = 0; i < n; i++) {
      bool is_dense_pivot_row = false;
      if (pivstep[i] <= plast && pivstep[i] >= dpos[first_step]) for (int s = 0; s < nbk; s++) if (drow[first_step + s] == i) is_dense_pivot_row = true;
      reachv[i] = is_dense_pivot_row ? 0 : reach(i);
      if (reachv[i] > 0) rowsv.push_back(i);
    }
    auto update_column = [&](double* x) -> bool {
      bool any = false;
      for (int s = 0; s < nbk; s++) {
        double a = x[drow[first_step + s]];
        for (int t = 0; t < s; t++) a = lu_upd(a, Lpp[s + (size_t)NB * t], u[t]);
        if (!fin(a)) return false;
        u[s] = a; x[drow[first_step + s]] = a;
        any = any || (a != 0.0);
      }
      if (!any) return true;                                       // every update is a - l * 0 with finite l
      for (int s = 0; s < nbk; s++) {                              // (per element still ascending in s)
        const double us = u[s]; if (us == 0.0) continue;
        const double* ls = &P[(size_t)n * s];
        for (int i : rowsv) if (s < reachv[i]) x[i] = lu_upd(x[i], ls[i], us);
      }
      return true;
    };
    std::vector<char> inpanel(slot_pos.size(), 0);
    for (int jj = 0; jj < npc; jj++) inpanel[colslot[pcols[jj]]] = 1;
    for (size_t sl = 0; sl < slot_pos.size(); sl++) if (!inpanel[sl] && slot_pos[sl] > plast) if (!update_column(&D[(size_t)n * sl])) return CL_FALLBACK;
    if (!update_column(b)) return CL_FALLBACK;                     // dgetrs: L y = P b
  }
  // dgetrs: U x = y.  A trivial step's column of U is its diagonal -1 alone: x = y / -1 once every dense column to its right has
  // been subtracted; the dense columns in descending order.
  const int nd = (int)dpos.size();
  for (int s = nd - 1; s >= 0; s--) {
    const double* col = &D[(size_t)n * dslot[s]];
    const double xs = b[drow[s]] / col[drow[s]];
    b[drow[s]] = xs;
    for (int i = 0; i < n; i++) if (pivstep[i] < dpos[s]) b[i] = lu_upd(b[i], xs, col[i]);
  }
  std::vector<double> x(n);
  for (int j = 0; j < n; j++) {
    const int r = rowofstep[j];
    const bool dense_step = ptype[j] == DENSE;
    x[j] = dense_step ? b[r] : b[r] / -1.0;
    if (!fin(x[j])) return CL_FALLBACK;
    if (ipiv_rows) ipiv_rows[j] = r;
  }
  for (int j = 0; j < n; j++) b[j] = x[j];
  st.dense_steps = nd;
  if (stats) *stats = st;
  return 0;
}

// ---- reuse across the pivots of lcp_lemke (the model of mh_lu_compact.inc's rec / prev_nd / cpos) --------------------------------
// Consecutive bases differ in ONE column, at position cpos.  Everything dgetf2 does at the positions before cpos is what it did for
// the previous basis, so the routine keeps its dense columns BY POSITION (K.D, n x n) together with position and row of every dense
// step, and on the next call
//   * replays the index effects of the kept steps before cpos (row order, step records, the fill-ins they cause),
//   * rebuilds only the columns from cpos on (gathered, or -e_r for a fill-in) and lets the kept steps update them and the new
//     right-hand side, NB steps at a time as one panel whose multipliers are read back from K.D (the grouping of steps into panels is
//     free: every element still receives its subtractions in step order),
//   * factorises from cpos on as lu_solve_compact does.
// cpos < 0 (or nothing kept): from scratch.  Anything that does not fit drops what was kept and starts over.
struct CompactLuKeep { int n = 0; bool valid = false; std::vector<double> D; std::vector<int> dpos, drow; int reused_steps = 0; };

inline int lu_solve_compact_keep(CompactLuKeep& K, int n, const int* kind, const int* idx, const double* dense, int ld_dense, double* b,
                                 int NB, int cpos, CompactLuStats* stats = nullptr)
{
  enum { HOME = 0, NONHOME = 1, DENSE = 2 };
  const int INF = 0x7fffffff;
  auto fin = [](double x) { return std::fabs(x) <= DBL_MAX; };
  const bool have = K.valid && K.n == n && cpos > 0;
  K.valid = false;                                                  // (whatever goes wrong below: the next call starts from scratch)
  if (K.n != n || (int)K.D.size() != n * n) { K.n = n; K.D.assign((size_t)n * n, 0.0); }
  std::vector<double>& D = K.D;                                     // column p = the dense column at POSITION p
  std::vector<int> ptype(n), slackpos(n, -1), pos_of_row(n), row_at_pos(n), pivstep(n, INF), rowofstep(n, -1);
  std::vector<int> dpos, drow;
  CompactLuStats st;
  for (int i = 0; i < n; i++) { pos_of_row[i] = i; row_at_pos[i] = i; }
  for (int c = 0; c < n; c++) {
    if (kind[c] == CL_UNIT) {
      const int r = idx[c];
      slackpos[r] = c;
      ptype[c] = (r == c) ? HOME : NONHOME;
      if (r == c) { pivstep[c] = c; rowofstep[c] = c; } else st.nonhome++;
    } else ptype[c] = DENSE;
  }
  auto exchange = [&](int c, int r) {
    const int jp = pos_of_row[r];
    if (jp != c) { const int displaced = row_at_pos[c]; row_at_pos[jp] = displaced; pos_of_row[displaced] = jp; row_at_pos[c] = r; pos_of_row[r] = c; }
  };
  int q = 0;
  if (have) for (size_t s2 = 0; s2 < K.dpos.size(); s2++) if (K.dpos[s2] < cpos) q++;
  const int cstart = (q > 0) ? cpos : 0;
  std::vector<char> rebuilt_fill(n, 0);
  if (q > 0) {                                                      // the index effects of every step before cstart, in position order
    int s2 = 0; bool ok = true;
    for (int p = 0; p < cstart && ok; p++) {
      if (ptype[p] == HOME) continue;
      int r;
      if (ptype[p] == NONHOME) r = idx[p];
      else { if (s2 >= q || K.dpos[s2] != p) { ok = false; break; } r = K.drow[s2]; }
      if (!(pivstep[r] > p)) { ok = false; break; }
      exchange(p, r); pivstep[r] = p; rowofstep[p] = r;
      if (ptype[p] == DENSE) {
        dpos.push_back(p); drow.push_back(r);
        const int sp = slackpos[r];
        if (sp > p) {
          ptype[sp] = DENSE; st.fill_ins++;
          if (sp >= cstart) { for (int i = 0; i < n; i++) D[i + (size_t)n * sp] = (i == r) ? -1.0 : 0.0; rebuilt_fill[sp] = 1; }
        }
        s2++;
      }
    }
    if (!ok || s2 != q) { CompactLuKeep fresh; fresh.n = n; fresh.D.swap(K.D); K = fresh; return lu_solve_compact_keep(K, n, kind, idx, dense, ld_dense, b, NB, -1, stats); }
  }
  for (int c = cstart; c < n; c++)
    if (kind[c] == CL_DENSE) for (int i = 0; i < n; i++) D[i + (size_t)n * c] = dense[i + (size_t)ld_dense * idx[c]];
  std::vector<double> P((size_t)n * NB), Lpp((size_t)NB * NB), u(NB);
  std::vector<int> pcols, freerows, reachv(n), rowsv;
  int first_step = 0, nbk = 0;
  auto prepare_rows = [&]() {                                       // Lpp, reachv, rowsv of the panel (first_step, nbk) whose multipliers are in P
    for (int s = 0; s < nbk; s++) for (int t = 0; t < s; t++) Lpp[s + (size_t)NB * t] = P[drow[first_step + s] + (size_t)n * t];
    const int plast = dpos[first_step + nbk - 1];
    rowsv.clear();
    for (int i = 0; i < n; i++) {
      bool is_dense_pivot_row = false;
      if (pivstep[i] <= plast && pivstep[i] >= dpos[first_step]) for (int s = 0; s < nbk; s++) if (drow[first_step + s] == i) is_dense_pivot_row = true;
      int c = 0; while (c < nbk && dpos[first_step + c] < pivstep[i]) c++;
      reachv[i] = is_dense_pivot_row ? 0 : c;
      if (reachv[i] > 0) rowsv.push_back(i);
    }
  };
  auto update_column = [&](double* x) -> bool {
    bool any = false;
    for (int s = 0; s < nbk; s++) {
      double a = x[drow[first_step + s]];
      for (int t = 0; t < s; t++) a = lu_upd(a, Lpp[s + (size_t)NB * t], u[t]);
      if (!fin(a)) return false;
      u[s] = a; x[drow[first_step + s]] = a;
      any = any || (a != 0.0);
    }
    if (!any) return true;
    for (int s = 0; s < nbk; s++) {
      const double us = u[s]; if (us == 0.0) continue;
      const double* ls = &P[(size_t)n * s];
      for (int i : rowsv) if (s < reachv[i]) x[i] = lu_upd(x[i], ls[i], us);
    }
    return true;
  };
  // the kept steps, NB at a time, on the rebuilt columns and the right-hand side
  for (first_step = 0; first_step < q; first_step += NB) {
    nbk = (q - first_step < NB) ? q - first_step : NB;
    for (int jj = 0; jj < nbk; jj++) for (int i = 0; i < n; i++) P[i + (size_t)n * jj] = D[i + (size_t)n * dpos[first_step + jj]];
    prepare_rows();
    for (int c = cstart; c < n; c++) if (ptype[c] == DENSE) if (!update_column(&D[(size_t)n * c])) return CL_FALLBACK;
    if (!update_column(b)) return CL_FALLBACK;
  }
  K.reused_steps = q;
  int cur = cstart;
  while (true) {
    pcols.clear();
    for (int p = cur; p < n && (int)pcols.size() < NB; p++) if (ptype[p] == DENSE) pcols.push_back(p);
    if (pcols.empty()) {
      for (int p = cur; p < n; p++) if (ptype[p] == NONHOME) { const int r = idx[p]; exchange(p, r); pivstep[r] = p; rowofstep[p] = r; }
      break;
    }
    st.panels++;
    const int npc = (int)pcols.size(), last = pcols.back();
    first_step = (int)dpos.size();
    for (int jj = 0; jj < npc; jj++) for (int i = 0; i < n; i++) P[i + (size_t)n * jj] = D[i + (size_t)n * pcols[jj]];
    nbk = 0; bool trunc = false;
    int p = cur;
    for (; p <= last && !trunc; p++) {
      if (ptype[p] == HOME) continue;
      if (ptype[p] == NONHOME) { const int r = idx[p]; exchange(p, r); pivstep[r] = p; rowofstep[p] = r; continue; }
      const int jj = nbk;
      double amax = -1.0; int bpos = INF;
      for (int i = 0; i < n; i++) if (pivstep[i] > p) {
        const double a = std::fabs(P[i + (size_t)n * jj]);
        if (!fin(a)) return CL_FALLBACK;
        if (a > amax || (a == amax && pos_of_row[i] < bpos)) { amax = a; bpos = pos_of_row[i]; }
      }
      if (!(amax != 0.0)) return p + 1;
      const int r = row_at_pos[bpos];
      exchange(p, r);
      pivstep[r] = p; rowofstep[p] = r;
      dpos.push_back(p); drow.push_back(r);
      const int sp = slackpos[r];
      if (sp > p) {
        ptype[sp] = DENSE; for (int i = 0; i < n; i++) D[i + (size_t)n * sp] = (i == r) ? -1.0 : 0.0;
        st.fill_ins++;
        if (sp < last) { trunc = true; st.truncated_panels++; }
      }
      const double piv = P[r + (size_t)n * jj];
      const bool big = std::fabs(piv) >= DBL_MIN;
      const double rcp = 1.0 / piv;
      freerows.clear();
      for (int i = 0; i < n; i++) if (pivstep[i] > p) freerows.push_back(i);
      { double* pj = &P[(size_t)n * jj];
        for (int i : freerows) { double l = pj[i]; l = big ? l * rcp : l / piv; if (!fin(l)) return CL_FALLBACK; pj[i] = l; }
        for (int c2 = jj + 1; c2 < npc; c2++) {
          double* pc = &P[(size_t)n * c2]; const double uu = pc[r];
          if (uu != 0.0) for (int i : freerows) pc[i] = lu_upd(pc[i], pj[i], uu);
        } }
      nbk++;
    }
    cur = p;
    for (int jj = 0; jj < npc; jj++) for (int i = 0; i < n; i++) D[i + (size_t)n * pcols[jj]] = P[i + (size_t)n * jj];
    prepare_rows();
    const int plast = dpos.back();
    for (int c = plast + 1; c < n; c++) {
      if (ptype[c] != DENSE) continue;
      bool inpanel = false; for (int jj = 0; jj < npc; jj++) if (pcols[jj] == c) inpanel = true;
      if (!inpanel && !update_column(&D[(size_t)n * c])) return CL_FALLBACK;
    }
    if (!update_column(b)) return CL_FALLBACK;
  }
  const int nd = (int)dpos.size();
  for (int s = nd - 1; s >= 0; s--) {
    const double* col = &D[(size_t)n * dpos[s]];
    const double xs = b[drow[s]] / col[drow[s]];
    b[drow[s]] = xs;
    for (int i = 0; i < n; i++) if (pivstep[i] < dpos[s]) b[i] = lu_upd(b[i], xs, col[i]);
  }
  std::vector<double> x(n);
  for (int j = 0; j < n; j++) {
    const int r = rowofstep[j];
    x[j] = (ptype[j] == DENSE) ? b[r] : b[r] / -1.0;
    if (!fin(x[j])) return CL_FALLBACK;
  }
  for (int j = 0; j < n; j++) b[j] = x[j];
  st.dense_steps = nd;
  if (stats) *stats = st;
  K.dpos = dpos; K.drow = drow; K.valid = true;
  return 0;
}

} // namespace oracle
#endif
