// rsr_solver.hpp -- Newton constraint solver, integrator and forward pass for one env / one wavefront.
// Restates MJX solver.solve / _linesearch / _update_constraint / _update_gradient
// with per-row scalars in lane registers, J in LDS, the Hessian assembled as 2x2 blocks (one block per
// lane) and factored by the in-register Cholesky of rsr_device.hpp.
#pragma once
#include "rsr_device.hpp"

namespace rsr {

// per-dof vectors live in registers of lane i (i < NV); lanes >= NV carry zeros
template <class C>
struct DofRegs { float qacc, Ma, grad, search, mv, qfc; };

// J * v for the pyramid rows owned by this lane (vb: the per-dof vector broadcast to every lane, vec_bcast): dot products with the
// base rows (lane = base row), published through LDS, then combined as base[bn] + mu * base[bk].
template <class C>
__device__ __forceinline__ void jdot(Smem<C>& s, int lane, int nefc, int nbase, const RowRegs (&rr)[C::NCHUNK], const float (&vb)[NVP<C>],
                                     float (&out)[C::NCHUNK]) {
  WSYNC();
#pragma unroll
  for (int ch = 0; ch < C::NCHB; ++ch) {
    if (64 * ch >= nbase) continue;             // wave-uniform
    int b = lane + 64 * ch, bb = b < nbase ? b : C::NBASE;
    // the lane's whole row in LDJ/4 ds_read_b128, all in flight before the first multiply
    const float4* row = reinterpret_cast<const float4*>(&s.x.b.J[bb * C::LDJ]);
    float4 q[C::LDJ / 4];
#pragma unroll
    for (int k = 0; k < C::LDJ / 4; ++k) q[k] = row[k];
    v2f acc2 = {0.0f, 0.0f};         // even / odd columns in the halves of a packed accumulator (see row_dot)
#pragma unroll
    for (int i = 0; i < C::NV; i += 2) {
      const float4& t = q[i / 4];
      const v2f jr = (i % 4 == 0) ? (v2f){t.x, t.y} : (v2f){t.z, t.w};
      acc2 = __builtin_elementwise_fma(jr, (v2f){vb[i], vb[i + 1]}, acc2);
    }
    const float acc = acc2.x + acc2.y;
    if (b < nbase) s.bval[b] = acc;
  }
  WSYNC();
#pragma unroll
  for (int ch = 0; ch < C::NCHUNK; ++ch) {
    out[ch] = 0.0f;
    if (64 * ch >= nefc) continue;
    out[ch] = s.bval[rr[ch].bn] + rr[ch].mu * s.bval[rr[ch].bk];
  }
}

// row classification by index
template <class C>
__device__ __forceinline__ int row_kind(int r) { return r < C::NEQ ? 0 : (r < C::NEQ + C::NF ? 1 : 2); }

// cost of the constraint rows at Jaref; optionally emits force and Hessian weight per row
template <class C, bool REDUCE = true>
__device__ __forceinline__ float rows_cost(int lane, int nefc, const float (&jaref)[C::NCHUNK], const RowRegs (&rr)[C::NCHUNK],
                                           float (&force)[C::NCHUNK], float (&hw)[C::NCHUNK]) {
  float cost = 0;
#pragma unroll
  for (int ch = 0; ch < C::NCHUNK; ++ch) {
    force[ch] = 0.0f; hw[ch] = 0.0f;
    if (64 * ch >= nefc) continue;            // wave-uniform
    int r = lane + 64 * ch, kind = row_kind<C>(r);
    float x = jaref[ch], D = rr[ch].D, f = 0, w = 0;
    bool act;
    if (kind == 0) act = true;
    else if (kind == 1) {
      float fl = rr[ch].floss, rf = rr[ch].R * fl;
      if (x <= -rf) { act = false; f = fl; cost += fl * (-0.5f * rf - x); }
      else if (x >= rf) { act = false; f = -fl; cost += fl * (-0.5f * rf + x); }
      else act = true;
    } else act = x < 0.0f;
    if (act) { f = -D * x; cost += 0.5f * D * x * x; w = D; }
    force[ch] = f; hw[ch] = w;
  }
  return REDUCE ? wave_sum(cost) : cost;      // !REDUCE: this lane's share, for a caller that reduces several sums together
}

// One step size of a line search: the 1-D model's derivatives there, and the summed linear / quadratic coefficients from which
// the cost is formed once the search has ended (ls_costs).
// (The derivatives are functions of three numbers and are formed where they are read: the line-search loop is the kernel's register
// peak, and its six points are wave-uniform values in vector registers.)
struct LSPoint {
  float alpha, q1, q2;
  __device__ __forceinline__ float d0() const { return 2.0f * alpha * q2 + q1; }
  __device__ __forceinline__ float d1() const { return 2.0f * q2 + (q2 == 0.0f ? RSR_MINVAL : 0.0f); }
};

// Per-row constants of one line search (they do not depend on the step size): the quadratic piece b0 + b1 a + b2 a^2 of an
// active row, and for the first chunk -- the only one that can hold equality and friction rows -- the two linear pieces
// and their thresholds.  Row kinds are unified there as "lower linear piece if x <= -tlo, upper one if x >= thi, else
// quadratic": friction rows have tlo = thi = R*floss; a unilateral row (limit, contact) has no lower piece (tlo = inf)
// and a zero upper piece from thi = 0 on; an equality row has neither (tlo = thi = inf).
template <class C>
struct LSRows {
  float ja[C::NCHUNK], v[C::NCHUNK], b0[C::NCHUNK], b1[C::NCHUNK], b2[C::NCHUNK];
  float tlo, thi, c0m, c1m, c0p, c1p;        // chunk 0 only
};
template <class C>
__device__ __forceinline__ void ls_prepare(int lane, int nefc, const float (&jaref)[C::NCHUNK], const float (&jv)[C::NCHUNK],
                                           const RowRegs (&rr)[C::NCHUNK], LSRows<C>& o) {
  static_assert(C::NEQ + C::NF <= 64, "equality and friction rows must sit in the first chunk");
#pragma unroll
  for (int ch = 0; ch < C::NCHUNK; ++ch) {
    float ja = jaref[ch], v = jv[ch], D = rr[ch].D;
    o.ja[ch] = ja; o.v[ch] = v;
    o.b0[ch] = 0.5f * ja * ja * D; o.b1[ch] = v * ja * D; o.b2[ch] = 0.5f * v * v * D;
    if (lane + 64 * ch >= nefc) { o.b0[ch] = 0.0f; o.b1[ch] = 0.0f; o.b2[ch] = 0.0f; o.v[ch] = 0.0f; o.ja[ch] = 1.0f; }   // padding rows: inactive
  }
  const int kind = row_kind<C>(lane);
  const float f = rr[0].floss, rf = rr[0].R * f, ja = jaref[0], v = jv[0];
  o.tlo = kind == 1 ? rf : INFINITY;
  o.thi = kind == 1 ? rf : (kind == 2 ? 0.0f : INFINITY);
  o.c0m = f * (-0.5f * rf - ja); o.c1m = -f * v;
  o.c0p = kind == 1 ? f * (-0.5f * rf + ja) : 0.0f; o.c1p = kind == 1 ? f * v : 0.0f;
  if (lane >= nefc) { o.tlo = INFINITY; o.thi = INFINITY; }
}

// The 1-D model's derivatives at NPT step sizes at once: the linear and quadratic pieces of the rows owned by the lane are summed,
// then the 2*NPT partial sums are reduced three chains at a time (independent DPP chains overlap).  The constant pieces -- the
// cost itself -- are NOT summed here: the bracket update only looks at derivatives, and the cost is needed for three step sizes of
// a search (its start and its two final ends), not for the three candidates of every iteration (ls_costs; a third of the row
// work and of the reductions of an iteration before).
template <class C, int NPT>
__device__ __forceinline__ void ls_eval(int nefc, const float (&alpha)[NPT], const LSRows<C>& w, float g1, float g2, LSPoint (&out)[NPT]) {
  static_assert(NPT == 1 || NPT == 3, "one step size or the three candidates of an iteration");
  float q[NPT][2];
  // (copies first: a ?: between struct members is an lvalue select, which would pin the struct in scratch memory)
  const float ja0 = w.ja[0], v0 = w.v[0], tlo = w.tlo, thi = w.thi, c1m = w.c1m, c1p = w.c1p;
  const float b10 = w.b1[0], b20 = w.b2[0];
#pragma unroll
  for (int p = 0; p < NPT; ++p) {
    // chunk 0: three pieces
    float x = ja0 + alpha[p] * v0;
    bool lo = x <= -tlo, hi = x >= thi;
    float u1 = hi ? c1p : b10;
    q[p][0] = lo ? c1m : u1;
    q[p][1] = (lo || hi) ? 0.0f : b20;
  }
#pragma unroll
  for (int ch = 1; ch < C::NCHUNK; ++ch) {
    if (64 * ch >= nefc) continue;            // wave-uniform
#pragma unroll
    for (int p = 0; p < NPT; ++p) {           // unilateral rows only: quadratic while x < 0
      float x = w.ja[ch] + alpha[p] * w.v[ch];
      bool act = x < 0.0f;
      q[p][0] += act ? w.b1[ch] : 0.0f; q[p][1] += act ? w.b2[ch] : 0.0f;
    }
  }
  if constexpr (NPT == 3) { wave_sum3(q[0][0], q[1][0], q[2][0]); wave_sum3(q[0][1], q[1][1], q[2][1]); }
  else { float z = 0.0f; wave_sum3(q[0][0], q[0][1], z); }
#pragma unroll
  for (int p = 0; p < NPT; ++p) {
    float q1 = q[p][0] + g1, q2 = q[p][1] + g2, al = alpha[p];
    out[p].alpha = al; out[p].q1 = q1; out[p].q2 = q2;
  }
}
template <class C>
__device__ __forceinline__ LSPoint ls_point(int nefc, float alpha, const LSRows<C>& w, float g1, float g2) {
  float al[1] = {alpha};
  LSPoint o[1];
  ls_eval<C, 1>(nefc, al, w, g1, g2, o);
  return o[0];
}
// Cost of the 1-D model at three points that have been evaluated: the constant pieces of the rows at each step size, one
// reduction for the three, and al^2 q2 + al q1 + q0 with the point's own q1 / q2 (the same sums, the same expression, the same
// bits as when every evaluation carried its cost along).
template <class C>
__device__ __forceinline__ void ls_costs(int nefc, const LSRows<C>& w, float g0, const LSPoint& a, const LSPoint& b, const LSPoint& c,
                                         float& cost_a, float& cost_b, float& cost_c) {
  const float al[3] = {a.alpha, b.alpha, c.alpha};
  float q0[3];
  const float ja0 = w.ja[0], v0 = w.v[0], tlo = w.tlo, thi = w.thi, c0m = w.c0m, c0p = w.c0p, b00 = w.b0[0];
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    float x = ja0 + al[p] * v0;
    bool lo = x <= -tlo, hi = x >= thi;
    float u0 = hi ? c0p : b00;
    q0[p] = lo ? c0m : u0;
  }
#pragma unroll
  for (int ch = 1; ch < C::NCHUNK; ++ch) {
    if (64 * ch >= nefc) continue;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      float x = w.ja[ch] + al[p] * w.v[ch];
      q0[p] += x < 0.0f ? w.b0[ch] : 0.0f;
    }
  }
  wave_sum3(q0[0], q0[1], q0[2]);
  auto cost = [&](const LSPoint& pt, float s0) { const float q0s = s0 + g0, q1 = pt.q1, q2 = pt.q2, a_ = pt.alpha; return a_ * a_ * q2 + a_ * q1 + q0s; };
  cost_a = cost(a, q0[0]); cost_b = cost(b, q0[1]); cost_c = cost(c, q0[2]);
}

// qfrc_constraint = J^T force.  Friction and limit rows have one +-1 entry: their force goes straight to that dof (one LDS
// float add per row).  The pyramid forces of a contact are folded onto its base rows (normal: sum of the edges;
// direction k: mu_k * (f_k+ - f_k-)), then J_base^T g runs over the contact base rows four per trip with addresses that
// are affine in the trip counter, so the loads of a trip (and of the next) are independent of each other.
template <class C>
__device__ __forceinline__ float jt_force(Smem<C>& s, int lane, int nefc, int nbase, const float (&force)[C::NCHUNK]) {
  const int ncon = s.ncon, rcon = nefc - C::NPYR * ncon;     // first contact row (pyramid and base numbering agree below it)
  WSYNC();
#pragma unroll
  for (int ch = 0; ch < C::NCHUNK; ++ch) { int r = lane + 64 * ch; if (r >= rcon && r < nefc) s.rw[r] = force[ch]; }
  if (lane < C::NV) s.dgw[lane] = 0.0f;
  WSYNC();
  if (lane >= C::NEQ && lane < rcon && force[0] != 0.0f) {
    const int d = s.sdof[lane];
    atomicAdd(&s.dgw[d], force[0] * s.x.b.J[lane * C::LDJ + d]);
  }
  constexpr int NCB = (C::NBC * C::NCON + 63) / 64;
#pragma unroll
  for (int ch = 0; ch < NCB; ++ch) {
    int t = lane + 64 * ch;
    if (t < C::NBC * ncon) {
      int c = t / C::NBC, k = t - C::NBC * c, r0 = rcon + C::NPYR * c;
      float g = 0.0f;
      if (k == 0) {
#pragma unroll
        for (int e = 0; e < C::NPYR; e += 2) g += s.rw[r0 + e] + s.rw[r0 + e + 1];
      } else g = s.bmu[rcon + t] * (s.rw[r0 + 2 * (k - 1)] - s.rw[r0 + 2 * (k - 1) + 1]);
      s.bval[rcon + t] = g;
    }
  }
  WSYNC();
  if constexpr (C::NBC * C::NCON <= 16) {       // few contact rows (Go2): one group, no exchange
    const int col = lane < C::NV ? lane : 0;
    float acc0 = s.dgw[col], acc1 = 0.0f;
#pragma unroll
    for (int r = 0; r < C::NEQ; ++r) acc1 += s.x.b.J[r * C::LDJ + col] * rdlane(force[0], r);
    for (int b = rcon; b < nbase; b += 4) {
      const int r0 = b, r1 = b + 1 < nbase ? b + 1 : C::NBASE, r2 = b + 2 < nbase ? b + 2 : C::NBASE, r3 = b + 3 < nbase ? b + 3 : C::NBASE;
      acc0 += s.x.b.J[r0 * C::LDJ + col] * s.bval[r0]; acc1 += s.x.b.J[r1 * C::LDJ + col] * s.bval[r1];
      acc0 += s.x.b.J[r2 * C::LDJ + col] * s.bval[r2]; acc1 += s.x.b.J[r3 * C::LDJ + col] * s.bval[r3];
    }
    return lane < C::NV ? acc0 + acc1 : 0.0f;
  }
  // lane = (row group, dof): the 64 / NV groups walk interleaved quartets of base rows, so the loop is 1 / groups as many
  // dependent LDS round trips long; the per-group partial sums meet in LDS
  constexpr int G = 64 / C::NV;
  const int grp = lane / C::NV, col = lane - grp * C::NV;
  float acc0 = 0.0f, acc1 = 0.0f;
  if (grp < G) {
    for (int b = rcon + 4 * grp; b < nbase; b += 4 * G) {
      const int r0 = b, r1 = b + 1 < nbase ? b + 1 : C::NBASE, r2 = b + 2 < nbase ? b + 2 : C::NBASE, r3 = b + 3 < nbase ? b + 3 : C::NBASE;
      acc0 += s.x.b.J[r0 * C::LDJ + col] * s.bval[r0]; acc1 += s.x.b.J[r1 * C::LDJ + col] * s.bval[r1];
      acc0 += s.x.b.J[r2 * C::LDJ + col] * s.bval[r2]; acc1 += s.x.b.J[r3 * C::LDJ + col] * s.bval[r3];
    }
  }
  s.jtp[lane] = acc0 + acc1;
  WSYNC();
  if (lane >= C::NV) return 0.0f;
  float tot = s.dgw[lane];
#pragma unroll
  for (int r = 0; r < C::NEQ; ++r) tot += s.x.b.J[r * C::LDJ + lane] * rdlane(force[0], r);
#pragma unroll
  for (int g = 0; g < G; ++g) tot += s.jtp[g * C::NV + lane];
  return tot;
}

// H = M + J^T diag(hw) J as 2x2 blocks (lane = block of the lower triangle), then factor.
//  * friction and limit rows have a single +-1 entry, so their J^T hw J is just hw added to that dof's diagonal: one
//    LDS float add per row into dgw[] (at most two rows share a dof, so the sum does not depend on their order);
//  * equality rows (two entries) stay rank-1 updates;
//  * a contact adds B^T W B with B its base rows and W the arrow matrix
//      W_nn = sum hw,  W_nk = mu_k (hw_k+ - hw_k-),  W_kk = mu_k^2 (hw_k+ + hw_k-)   (k = tangent 1, tangent 2, torsion);
//    all contacts are visited (an all-inactive pyramid has W = 0), which keeps the loop free of index loads.
template <class C>
__device__ __forceinline__ float hessian_factor(Smem<C>& s, int lane, int nefc, int nbase, const float (&hw)[C::NCHUNK],
                                               float (&a)[C::NCH], float (&lt)[C::NCH], bool joined PROF_ARG) {
  constexpr int NBLK = (C::NV + 1) / 2;
  static_assert(NBLK * (NBLK + 1) / 2 <= 64, "Hessian blocks exceed one wave");
  static_assert(C::NV % 2 == 0, "2x2 Hessian blocking assumes an even dof count");
  static_assert(C::NSP <= 64 && C::NCON <= 64, "one lane per sparse row / contact");
  const int ncon = s.ncon, rcon = nefc - C::NPYR * ncon;
  WSYNC();
#pragma unroll
  for (int ch = 0; ch < C::NCHUNK; ++ch) { int r = lane + 64 * ch; if (r >= rcon && r < nefc) s.rw[r] = hw[ch]; }
  if (lane < C::NV) s.dgw[lane] = 0.0f;
  WSYNC();
  if (lane >= C::NEQ && lane < rcon && hw[0] != 0.0f) atomicAdd(&s.dgw[s.sdof[lane]], hw[0]);
  if (lane < ncon) {
    int r0 = rcon + C::NPYR * lane, b0 = rcon + C::NBC * lane;
    float* w = &s.wc[8 * lane];
    float wnn = 0.0f;
#pragma unroll
    for (int k = 1; k < C::NBC; ++k) {
      float hp = s.rw[r0 + 2 * (k - 1)], hm = s.rw[r0 + 2 * (k - 1) + 1], mu = s.bmu[b0 + k];
      wnn += hp + hm;
      w[k] = mu * (hp - hm);
      w[C::NBC - 1 + k] = mu * mu * (hp + hm);
    }
    w[0] = wnn;
  }
  WSYNC();
  PROF(PS_H_PREP)
  // block row of this lane in the packed lower triangle: the largest bi with bi (bi + 1) / 2 <= lane (closed form + fix-up
  // instead of a counting loop of up to NBLK trips per Hessian)
  int bi = 0;
  if constexpr (C::ROWCHOL) {        // (measured: +0.9 % on the cube; the Go2 kernel is 2 % faster with the loop)
    bi = (int)((fsqrt(8.0f * (float)lane + 1.0f) - 1.0f) * 0.5f);      // (1 ulp is enough: the two fix-ups below)
    bi += ((bi + 1) * (bi + 2) / 2 <= lane) ? 1 : 0;
    bi -= (bi * (bi + 1) / 2 > lane) ? 1 : 0;
  } else {
    while ((bi + 1) * (bi + 2) / 2 <= lane) ++bi;
  }
  int bj = lane - bi * (bi + 1) / 2;
  bool blk = bi < NBLK;
  int i0 = blk ? 2 * bi : 0, j0 = blk ? 2 * bj : 0;
  float h00 = s.M[i0 * C::LD + j0], h01 = s.M[i0 * C::LD + j0 + 1];
  float h10 = s.M[(i0 + 1) * C::LD + j0], h11 = s.M[(i0 + 1) * C::LD + j0 + 1];
  if (bi == bj) { h00 += s.dgw[i0]; h11 += s.dgw[i0 + 1]; }
#pragma unroll
  for (int r = 0; r < C::NEQ; ++r) {
    const float* Jr = &s.x.b.J[r * C::LDJ];
    float w = rdlane(hw[0], r);
    float a0 = Jr[i0] * w, a1 = Jr[i0 + 1] * w, b0 = Jr[j0], b1 = Jr[j0 + 1];
    h00 += a0 * b0; h01 += a0 * b1; h10 += a1 * b0; h11 += a1 * b1;
  }
  PROF(PS_H_SPARSE)
  auto add_contact = [&](int c) {
    const float* B = &s.x.b.J[(rcon + C::NBC * c) * C::LDJ];
    const float* w = &s.wc[8 * c];
    float ni0 = B[i0], ni1 = B[i0 + 1], nj0 = B[j0], nj1 = B[j0 + 1];
    float u0n = w[0] * nj0, u1n = w[0] * nj1;          // (W b_j)[normal] for the two columns of the block
    float a00 = 0, a01 = 0, a10 = 0, a11 = 0;
#pragma unroll
    for (int d = 1; d < C::NBC; ++d) {
      float wn = w[d], wd = w[C::NBC - 1 + d];
      float di0 = B[d * C::LDJ + i0], di1 = B[d * C::LDJ + i0 + 1], dj0 = B[d * C::LDJ + j0], dj1 = B[d * C::LDJ + j0 + 1];
      u0n += wn * dj0; u1n += wn * dj1;
      float u0d = wn * nj0 + wd * dj0, u1d = wn * nj1 + wd * dj1;   // (W b_j)[direction d]
      a00 += di0 * u0d; a01 += di0 * u1d; a10 += di1 * u0d; a11 += di1 * u1d;
    }
    h00 += ni0 * u0n + a00; h01 += ni0 * u1n + a01; h10 += ni1 * u0n + a10; h11 += ni1 * u1n + a11;
  };
  bool by_tree = false;
  if constexpr (C::ROWTREE && C::tree_size(2) > 0) by_tree = !joined;      // (two trees, one of them the arm with few contacts: no gain, measured)
  if (by_tree) {
    // no contact joins two trees: a block lies in one tree (or between two, and is zero) and only that tree's contacts touch it,
    // so the lanes of the three trees walk their own contact lists side by side: max(contacts per tree) trips instead of ncon.
    // Within a block the contacts are added in the same order as in the full walk, which only adds zeros in between.
    if constexpr (C::ROWTREE) {
      const int ti = C::tree_of(i0), tj = C::tree_of(j0);
      const int n0 = s.tree_ncon[0], n1 = s.tree_ncon[1], n2 = s.tree_ncon[2];
      const int nmine = (blk && ti == tj) ? (ti == 0 ? n0 : (ti == 1 ? n1 : n2)) : 0;
      const int nmax = n0 > n1 ? (n0 > n2 ? n0 : n2) : (n1 > n2 ? n1 : n2);
      const unsigned char* list = &s.tree_con[ti * C::NCON];
      // the first eight entries of the list in two registers: no index load inside the loop for the usual case
      static_assert(C::NCON % 8 == 0, "tree contact lists are read as 8-byte words");
      const uint2 head = *reinterpret_cast<const uint2*>(list);
      for (int q = 0; q < nmax; ++q) {
        const unsigned word = q < 4 ? head.x : head.y;
        const int c = q < 8 ? (int)((word >> (8 * (q & 3))) & 0xFFu) : (int)list[q < C::NCON ? q : 0];
        if (q < nmine) add_contact(c);
      }
    }
  } else {
    for (int c = 0; c < ncon; ++c) add_contact(c);
  }
  PROF(PS_H_CONTACT)
  if (blk) {
    float* const Tb = s.scratch_b();
    Tb[i0 * C::LD + j0] = h00; Tb[(i0 + 1) * C::LD + j0] = h10; Tb[(i0 + 1) * C::LD + j0 + 1] = h11;
    if (bi != bj) Tb[i0 * C::LD + j0 + 1] = h01;     // diagonal blocks: (i0, i0+1) is upper, never read
  }
  WSYNC();
  // Row `lane` of T, unmasked: chol_factor never consumes a[j] of a lane < j before zeroing it, and lanes >= NV (which
  // read row 0) are never read by anyone.  (A select on the loaded value makes the compiler wrap every load in its own
  // exec-mask region with a wait inside: 20 serialised LDS round trips.)
  if constexpr (C::ROWCHOL) {
    // the row-blocked factorisation reads its (permuted) rows from T and later writes its transpose to T: one wave, LDS
    // operations in order, and the rows are in registers (waited for) before the factor loop that precedes the writes
    PROF(PS_H_XCHG)
    if constexpr (C::ROWTREE) {
      if (!joined) {                                  // no contact joins two trees: one tree per DPP row
        const float dinv = rowtree_factor<C>(s.scratch_b(), 0.0f, a, lt, s.scratch_b(), lane);
        if constexpr (!C::TALIAS && !C::TTAIL) { if (lane < 4) s.bval[C::NBASE + lane] = 0.0f; }
        PROF(PS_H_CHOL)
        return dinv;
      }
    }
    const float dinv = rowchol_factor<C, false>(s.scratch_b(), 0.0f, a, lt, s.scratch_b(), lane);
    if constexpr (!C::TALIAS && !C::TTAIL) { if (lane < 4) s.bval[C::NBASE + lane] = 0.0f; }      // the null row's value (the exchange ran over it)
    PROF(PS_H_CHOL)
    return dinv;
  } else if constexpr (C::ARROW) {
    PROF(PS_H_XCHG)
    const float dinv = arrow_factor<C>(s.scratch_b(), 0.0f, a, lt, s.scratch_b(), lane);
    PROF(PS_H_CHOL)
    return dinv;
  } else {
  const float* Trow = &s.scratch_b()[(lane < C::NV ? lane : 0) * C::LD];
#pragma unroll
  for (int j = 0; j < C::NV; ++j) a[j] = Trow[j];
  WSYNC();
  PROF(PS_H_XCHG)
  const float dinv = chol_factor<C>(a, lt, s.scratch_b(), lane);
  if constexpr (!C::TALIAS && !C::TTAIL) { if (lane < 4) s.bval[C::NBASE + lane] = 0.0f; }
  PROF(PS_H_CHOL)
  return dinv;
  }
}

struct SolveStats { int niter, ls_total; };

// Row `lane` of the mass matrix (lanes >= NV: row 0, every use is masked).  Dims::MROW_LDS models call this at every use of
// the row instead of keeping it in registers across the solve.
template <class C>
__device__ __forceinline__ void load_mrow(const Smem<C>& s, int lane, float (&Mrow)[C::NV]) {
  const float* r = &s.M[(lane < C::NV ? lane : 0) * C::LD];
#pragma unroll
  for (int j = 0; j < C::NV; ++j) Mrow[j] = r[j];
}

// Whether integrate() will solve (M + dt*D) qacc = qfrc_smooth + qfrc_constraint (implicitfast, or Euler with damping
// folded in): the only consumer of qfrc_constraint on the path.  Wave-uniform.
template <class C>
__device__ __forceinline__ bool implicit_integration(const Hot& m, const Smem<C>& s, int lane) {
  bool implicit = m.integrator == INT_IMPLICITFAST;
  if (m.integrator == INT_EULER && !m.disable_eulerdamp) {
    float dm = lane < C::NV ? fabsf(s.damp[lane]) : 0.0f;
    implicit = uniform_i(__ballot(dm != 0.0f) != 0ull);
  }
  return implicit;
}

// Newton solve.  In: Mrow (row i of M in lane i), fs = qfrc_smooth_i, a0 = qacc_smooth_i, warm_i.
// Out: qacc_i, qfrc_constraint_i.
template <class C>
__device__ __forceinline__ void solve(const Hot& m, Smem<C>& s, int lane, int nefc, int nbase, const RowRegs (&rr)[C::NCHUNK],
                      float (&Mrow)[C::NV], float fs, float a0, float warm, bool need_force, float& qacc_out, float& qfc_out,
                      SolveStats& st, float* dbg PROF_ARG) {
  const bool dofl = lane < C::NV;
  // Work-queue kernels (two persistent waves per SIMD): the solver and the integrator -- chains of dependent reductions, pivots and
  // line-search steps, where an issue slot lost to the other wave lengthens the critical path -- ask for the right of way over the
  // geometry stages, whose independent instructions fill whatever slots are left.  Timing only; measured +0.55 % (cube), +0.8 %
  // (T-shape); the reverse order -0.4 %; finer grades (line search above the rest of the solver) add nothing.  The Go2-family
  // kernels carry a priority schedule of their own (prio_substep).
  if constexpr (!C::ARROW) __builtin_amdgcn_s_setprio(1);
  float force[C::NCHUNK], hw[C::NCHUNK], jaref[C::NCHUNK], jv[C::NCHUNK], tmp[C::NCHUNK];
  float a[C::NCH], lt[C::NCH];
  // --- warm start: the cheaper of qacc_warmstart and qacc_smooth (cost only) ---
  // qacc_smooth is costed first so that the force / weight registers hold the warm-start point afterwards: the warm start
  // wins almost always, and its context (row cost, Gauss term, force, hw) is then already there instead of being
  // evaluated a third time.
  if constexpr (C::MROW_LDS) load_mrow<C>(s, lane, Mrow);
  float vb[NVP<C>];
  vec_bcast<C>(s, lane, a0, vb);
  float Ma_s = dofl ? row_dot<C>(Mrow, vb) : 0.0f;
  float jar_s[C::NCHUNK];
  jdot<C>(s, lane, nefc, nbase, rr, vb, tmp);
#pragma unroll
  for (int ch = 0; ch < C::NCHUNK; ++ch) jar_s[ch] = tmp[ch] - rr[ch].aref;
  const float cost_s = rows_cost<C>(lane, nefc, jar_s, rr, force, hw);      // (its Gauss term (Ma_s - fs).(a0 - a0) is zero)
  vec_bcast<C>(s, lane, warm, vb);
  float Ma_w = dofl ? row_dot<C>(Mrow, vb) : 0.0f;
  jdot<C>(s, lane, nefc, nbase, rr, vb, tmp);
#pragma unroll
  for (int ch = 0; ch < C::NCHUNK; ++ch) jaref[ch] = tmp[ch] - rr[ch].aref;
  float rows_w = rows_cost<C, false>(lane, nefc, jaref, rr, force, hw), gs_w = dofl ? (Ma_w - fs) * (warm - a0) : 0.0f, unused = 0.0f;
  wave_sum3(rows_w, gs_w, unused);
  const float cost_w = rows_w + 0.5f * gs_w;
  const bool use_warm = cost_w < cost_s;
  float qacc = use_warm ? warm : a0, Ma = use_warm ? Ma_w : Ma_s;
  // --- context at the start point ---
  float gauss = 0.5f * gs_w, rc = rows_w, cost, prev_cost = INFINITY;
  if (!uniform_i(use_warm)) {
#pragma unroll
    for (int ch = 0; ch < C::NCHUNK; ++ch) jaref[ch] = jar_s[ch];
    rc = rows_cost<C>(lane, nefc, jaref, rr, force, hw);
    gauss = 0.0f;
  }
  cost = rc + gauss;
  PROF(PS_S_COST)
  float qfc = jt_force<C>(s, lane, nefc, nbase, force);
  float grad = dofl ? Ma - fs - qfc : 0.0f;
  PROF(PS_S_JTF)
  // the solver's options, fetched once: read where they are used, each is a scalar load from the model struct with its own
  // wait (the struct is passed by pointer so that its ~180 pointers do not sit in SGPRs), a dozen per Newton iteration
  bool joined = true;
  if constexpr (C::ROWTREE) joined = uniform_i(s.trees_joined) != 0;
  const int opt_iterations = m.iterations, opt_ls_iterations = m.ls_iterations;
  const float opt_tolerance = m.tolerance, opt_ls_tolerance = m.ls_tolerance, opt_meaninertia = m.meaninertia;
  const float scale = 1.0f / (opt_meaninertia * (float)(C::NV > 1 ? C::NV : 1));
  int iter = 0, ls_total = 0;
  if (dbg && lane == 0) { dbg[7400] = cost_s; dbg[7401] = cost_w; dbg[7402] = cost; }      // parity dump: the solver's trajectory
  float dinv = 0.0f, hw_fact[C::NCHUNK];
  bool have_factor = false;
#pragma unroll
  for (int ch = 0; ch < C::NCHUNK; ++ch) hw_fact[ch] = 0.0f;
  // |grad|^2: summed here for the first exit test, afterwards in the same reduction as the cost of the new point (one reduction per
  // Newton iteration less; the chains of a three-value reduction pair their additions as a single one does: same bits)
  float gn2 = opt_iterations != 1 ? wave_sum(grad * grad) : 0.0f;
  while (true) {
    bool done;
    if (opt_iterations != 1) {
      float gn = fsqrt(gn2);
      done = iter >= opt_iterations;
      done |= scale * (prev_cost - cost) < opt_tolerance;
      done |= scale * gn < opt_tolerance;
    } else done = iter >= 1;
    if (uniform_i(done)) break;
    // Newton direction at the current point.  MJX's loop body ends with _update_gradient (gradient, Hessian, factor,
    // search), so the reference factors once more after the last iteration and never uses the result; the exit test only
    // needs cost and gradient, so the Hessian is built here, at the top of an iteration that is known to run: one
    // assembly + factorisation + solve fewer per solve (of ~4 on the Airbot models, of 2 on Go2), same iterates.
    // H = M + J^T diag(hw) J depends on the iterate only through the row weights hw (D on the quadratic piece of a row,
    // 0 elsewhere): while the active set does not change between iterations -- the usual case close to convergence --
    // the factorisation of the previous iteration is still the factorisation of H.
    bool changed = !have_factor;
#pragma unroll
    for (int ch = 0; ch < C::NCHUNK; ++ch) changed |= hw[ch] != hw_fact[ch];
    if (uniform_i(__ballot(changed) != 0ull)) {
      dinv = hessian_factor<C>(s, lane, nefc, nbase, hw, a, lt, joined PROF_PASS);
#pragma unroll
      for (int ch = 0; ch < C::NCHUNK; ++ch) hw_fact[ch] = hw[ch];
      have_factor = true;
    }
    float search;
    if constexpr (C::ROWTREE) search = joined ? -rowchol_solve<C>(a, lt, dinv, grad, lane) : -rowtree_solve<C>(a, lt, dinv, grad, lane);
    else if constexpr (C::ROWCHOL) search = -rowchol_solve<C>(a, lt, dinv, grad, lane);
    else if constexpr (C::ARROW) search = -arrow_solve<C>(a, lt, dinv, grad, lane);
    else search = dofl ? -chol_solve<C>(a, lt, dinv, grad, lane) : 0.0f;
    search = dofl ? search : 0.0f;
    PROF(PS_HESS)
    // ---------------- line search ----------------
    if constexpr (C::MROW_LDS) load_mrow<C>(s, lane, Mrow);
    float sb[NVP<C>];
    vec_bcast<C>(s, lane, search, sb);
    float mv = dofl ? row_dot<C>(Mrow, sb) : 0.0f;
    jdot<C>(s, lane, nefc, nbase, rr, sb, jv);
    float snorm = search * search, g1a = search * Ma, g1b = search * fs;
    wave_sum3(snorm, g1a, g1b);
    snorm = fsqrt(snorm);
    float gtol = opt_tolerance * opt_ls_tolerance * snorm * opt_meaninertia * (float)(C::NV > 1 ? C::NV : 1);
    float g1 = g1a - g1b;
    // fp32 noise floor of the 1-D derivative: d0(alpha) = 2 alpha q2 + q1 is a sum of up to NEFC terms, so values
    // below eps * (sum |q1 terms| + 2 |alpha| sum |q2 terms|) are indistinguishable from zero.  MJX's gtol
    // (tolerance * ls_tolerance * |search| * scale ~ 1e-6) is below that floor in fp32 and the reference loop
    // then dithers until no bracket update happens; stopping at the floor changes nothing above rounding noise.
    float n1 = 0.0f, n2 = 0.0f;
#pragma unroll
    for (int ch = 0; ch < C::NCHUNK; ++ch) {
      n1 += fabsf(jv[ch] * jaref[ch] * rr[ch].D) + (rr[ch].floss > 0.0f ? fabsf(rr[ch].floss * jv[ch]) : 0.0f);
      n2 += 0.5f * jv[ch] * jv[ch] * rr[ch].D;
    }
    float g2 = search * mv, n1a = fabsf(search * Ma), n1b = fabsf(search * fs);
    wave_sum3(g2, n1a, n1b);
    g2 *= 0.5f;
    float zero = 0.0f;
    wave_sum3(n1, n2, zero);
    n1 = (n1 + n1a) + n1b;
    n2 += fabsf(g2);
    // Only for converging solves: with iterations == 1 (Go2) the reference's truncated procedure IS the answer.
    const float NOISE = opt_iterations > 1 ? 1.1920929e-7f : 0.0f;
    PROF(PS_LS_SETUP)
    LSRows<C> lw;
    ls_prepare<C>(lane, nefc, jaref, jv, rr, lw);
    PROF(PS_L_PREP)
    LSPoint p0 = ls_point<C>(nefc, 0.0f, lw, g1, g2);
    PROF(PS_L_P0)
    const float p0_d0 = p0.d0();
    LSPoint lo = ls_point<C>(nefc, p0.alpha - p0_d0 * __builtin_amdgcn_rcpf(p0.d1()), lw, g1, g2), hi;
    if (lo.d0() < p0_d0) { hi = p0; } else { hi = lo; lo = p0; }
    PROF(PS_L_LO)
    bool swap = true; int it = 0;
    // Limit cycles of the bracket update, cut short exactly.  A Newton step from `lo` that overshoots is accepted as the new
    // `lo` although its derivative is positive; the two ends then keep trading roles and only the iteration cap ends the
    // search (4-15 % of the searches on the Airbot models, 50 iterations each: they were the long tail of the wave
    // lifetimes).  The loop state is the pair (lo.alpha, hi.alpha) -- every other field is a function of alpha -- so once
    // the pair repeats bit for bit with period P the rest is known: (cap - it) mod P more iterations leave exactly the state
    // the cap would have left.  Results are bit-identical to running the cap out (the oracle has the same switch).
    const int max_it = opt_ls_iterations;
    int cap = max_it;
    // the history lives in lanes (lane k = the pair k + 1 iterations ago): two VGPRs, one ballot per iteration, no scalar
    // arrays (the kernel is short of SGPRs: sixteen more of them turned into v_writelane / v_readlane spill traffic in this loop)
    constexpr int LS_HIST = 8;
    float hist_lo = 0.0f, hist_hi = 0.0f;
    const bool cut_cycles = opt_iterations > 1;
    while (true) {
      if (cut_cycles && cap == max_it) {
        unsigned long long same = __ballot(hist_lo == lo.alpha && hist_hi == hi.alpha);
        same &= it >= LS_HIST ? ((1ull << LS_HIST) - 1ull) : ((1ull << it) - 1ull);
        if (same) { const int P = __builtin_ctzll(same) + 1; cap = it + (max_it - it) % P; }      // smallest period first
        hist_lo = dpp_mov<0x111>(hist_lo); hist_hi = dpp_mov<0x111>(hist_hi);                       // row_shr:1
        if (lane == 0) { hist_lo = lo.alpha; hist_hi = hi.alpha; }
      }
      const float lo_d0 = lo.d0(), hi_d0 = hi.d0();
      bool ldone = it >= cap;
      ldone |= !swap;
      // MJX's own test, signs included, for its tolerance; below the rounding noise of its own sum the derivative has no sign, so the
      // noise-floor term asks for the magnitude only (NOISE = 0, the single-iteration models: MJX's rule alone)
      const float nz_lo = NOISE * (n1 + 2.0f * fabsf(lo.alpha) * n2), nz_hi = NOISE * (n1 + 2.0f * fabsf(hi.alpha) * n2);
      ldone |= ((lo_d0 < 0.0f) && (lo_d0 > -gtol)) || (fabsf(lo_d0) < nz_lo);
      ldone |= ((hi_d0 > 0.0f) && (hi_d0 < gtol)) || (fabsf(hi_d0) < nz_hi);
      if (uniform_i(ldone)) break;
      float al3[3] = {lo.alpha - lo_d0 * __builtin_amdgcn_rcpf(lo.d1()), hi.alpha - hi_d0 * __builtin_amdgcn_rcpf(hi.d1()), 0.5f * (lo.alpha + hi.alpha)};
      LSPoint p3[3];
      ls_eval<C, 3>(nefc, al3, lw, g1, g2, p3);
      const LSPoint lo_next = p3[0], hi_next = p3[1], mid = p3[2];
      const float mid_d0 = mid.d0();
      float lo_cur = lo_d0, hi_cur = hi_d0;                   // d0 of the ends as the four updates go along
      bool s1 = (lo_cur > 0.0f) || (lo_cur < lo_next.d0());
      if (s1) { lo = lo_next; lo_cur = lo_next.d0(); }
      bool s2 = (mid_d0 < 0.0f) && (lo_cur < mid_d0);
      if (s2) lo = mid;
      bool s3 = (hi_cur < 0.0f) || (hi_cur > hi_next.d0());
      if (s3) { hi = hi_next; hi_cur = hi_next.d0(); }
      bool s4 = (mid_d0 > 0.0f) && (hi_cur > mid_d0);
      if (s4) hi = mid;
      swap = s1 | s2 | s3 | s4;
      ++it;
    }
    ls_total += it;
    PROF(PS_L_ITER)
    float cost_p0, cost_lo, cost_hi;
    ls_costs<C>(nefc, lw, gauss, p0, lo, hi, cost_p0, cost_lo, cost_hi);
    bool improved = (cost_lo < cost_p0) || (cost_hi < cost_p0);
    float alpha = cost_lo < cost_hi ? lo.alpha : hi.alpha;
    if (improved) {
      qacc += alpha * search; Ma += alpha * mv;
#pragma unroll
      for (int ch = 0; ch < C::NCHUNK; ++ch) jaref[ch] += alpha * jv[ch];
    }
    // A single-iteration solve (opt.iterations == 1, Go2) has no exit test to feed, so when nobody reads qfrc_constraint
    // either (plain Euler) the post-step force / cost / gradient evaluation is dead: stop at the new qacc.
    if (opt_iterations == 1 && !need_force) { ++iter; break; }
    // ---------------- update constraint + gradient ----------------
    rc = rows_cost<C, false>(lane, nefc, jaref, rr, force, hw);
    gauss = dofl ? (Ma - fs) * (qacc - a0) : 0.0f;
    PROF(PS_X6)
    qfc = jt_force<C>(s, lane, nefc, nbase, force);
    grad = dofl ? Ma - fs - qfc : 0.0f;
    gn2 = grad * grad;
    wave_sum3(rc, gauss, gn2);
    gauss *= 0.5f;
    prev_cost = cost; cost = rc + gauss;
    if (dbg && lane == 0 && iter < 16) { dbg[7410 + 4 * iter] = cost; dbg[7411 + 4 * iter] = improved ? alpha : 0.0f; dbg[7412 + 4 * iter] = (float)it; dbg[7413 + 4 * iter] = p0_d0; }
    PROF(PS_UPD)
    ++iter;
  }
  PROF(PS_U_JTF)
  st.niter = iter; st.ls_total = ls_total;
  qacc_out = qacc; qfc_out = qfc;
}

// per-lane registers that survive one forward pass
template <class C>
struct FwdOut { float qacc, qfc, fsmooth; int nefc; SolveStats st; };

// MJX forward(): position -> collision -> constraint rows -> velocity/actuation -> solve.
// warm_i is read and replaced by the solver's qacc (qacc_warmstart <- qacc).
template <class C>
__device__ __forceinline__ void forward(const DModel& m, const Hot& h, Smem<C>& s, int lane, float (&Mrow)[C::NV], float& warm, FwdOut<C>& out,
                        float* dbg PROF_ARG) {
  kinematics<C>(m, h, s, lane PROF_PASS);
  PROF(PS_KIN)
  com_crb_mass<C>(m, h, s, lane PROF_PASS);
  load_mrow<C>(s, lane, Mrow);
  PROF(PS_COMCRB)
  // velocity stage first: its scratch and the frames die before the Jacobian claims the shared LDS region
  float qvel_i = lane < C::NV ? s.qvel[lane] : 0.0f;
  float fs = smooth_forces<C>(m, h, s, lane, qvel_i, 0.0f PROF_PASS);
  PROF(PS_SMOOTH)
  // qacc_smooth = M^-1 qfrc_smooth
  float a[C::NCH], lt[C::NCH];
  float a0;
  if constexpr (C::ROWTREE) {
    const float dinv_m = rowtree_factor<C>(s.M, 0.0f, a, lt, s.scratch_a(), lane);
    a0 = rowtree_solve<C>(a, lt, dinv_m, fs, lane);
    a0 = lane < C::NV ? a0 : 0.0f;
  } else if constexpr (C::ROWCHOL) {
    const float dinv_m = rowchol_factor<C, true>(s.M, 0.0f, a, lt, s.scratch_a(), lane);
    a0 = rowchol_solve<C>(a, lt, dinv_m, fs, lane);
    a0 = lane < C::NV ? a0 : 0.0f;
  } else if constexpr (C::ARROW) {
    const float dinv_m = arrow_factor<C>(s.M, 0.0f, a, lt, s.scratch_a(), lane);
    a0 = arrow_solve<C>(a, lt, dinv_m, fs, lane);
    a0 = lane < C::NV ? a0 : 0.0f;
  } else {
#pragma unroll
    for (int j = 0; j < C::NV; ++j) a[j] = Mrow[j];        // entries j > lane are never consumed by chol_factor
    const float dinv_m = chol_factor<C, true>(a, lt, s.scratch_a(), lane);
    a0 = lane < C::NV ? chol_solve<C>(a, lt, dinv_m, fs, lane) : 0.0f;
  }
  PROF(PS_CHOLM)
  collision<C>(m, h, s, lane PROF_PASS);
  PROF(PS_COLL)
  RowRegs rr[C::NCHUNK];
  float bcoef[C::NCHUNK], jqv[C::NCHUNK];
  int nbase;
  int nefc = make_constraint<C>(m, h, s, lane, rr, bcoef, nbase PROF_PASS);
  {
    float qb[NVP<C>];
    vec_bcast<C>(s, lane, qvel_i, qb);
    jdot<C>(s, lane, nefc, nbase, rr, qb, jqv);                   // aref = -b (J.qvel) - k imp pos
  }
#pragma unroll
  for (int ch = 0; ch < C::NCHUNK; ++ch) rr[ch].aref -= bcoef[ch] * jqv[ch];
  PROF(PS_ROWS)
  out.fsmooth = fs; out.nefc = nefc;
  const bool need_force = dbg != nullptr || implicit_integration<C>(h, s, lane);
  solve<C>(h, s, lane, nefc, nbase, rr, Mrow, fs, a0, warm, need_force, out.qacc, out.qfc, out.st, dbg PROF_PASS);
  warm = out.qacc;
  if (dbg) {   // parity dump (layout: rsr_mjx_amd/_debug_layout in the Python binding)
    if (lane == 0) {
      dbg[0] = (float)nefc; dbg[1] = (float)C::NEQ; dbg[2] = (float)C::NF; dbg[3] = (float)s.ncon;
      dbg[4] = (float)out.st.niter; dbg[5] = (float)out.st.ls_total; dbg[6] = (float)s.nlim_act; dbg[7] = (float)s.ncon_drop;
    }
    for (int t = lane; t < C::NB * 3; t += 64) dbg[16 + t] = s.xpos[t];
    if constexpr (C::TALIAS) {        // (the mass matrix's LDS storage has been reused since the Hessian: dump the register copy)
      if (lane < C::NV) {
#pragma unroll
        for (int j = 0; j < C::NV; ++j) dbg[128 + lane * C::NV + j] = Mrow[j];
      }
    } else {
      for (int t = lane; t < C::NV * C::NV; t += 64) dbg[128 + t] = s.M[(t / C::NV) * C::LD + (t % C::NV)];
    }
    if (lane < C::NV) {
      dbg[736 + lane] = fs; dbg[768 + lane] = a0; dbg[800 + lane] = out.qacc; dbg[832 + lane] = out.qfc;
    }
    for (int t = lane; t < s.ncon; t += 64) {
      float* c = &dbg[864 + 8 * t];
      c[0] = s.cdist[t]; c[1] = s.cpos[3 * t]; c[2] = s.cpos[3 * t + 1]; c[3] = s.cpos[3 * t + 2];
      c[4] = s.cnrm[3 * t]; c[5] = s.cnrm[3 * t + 1]; c[6] = s.cnrm[3 * t + 2]; c[7] = (float)s.cpair[t];
    }
#pragma unroll
    for (int ch = 0; ch < C::NCHUNK; ++ch) {
      int r = lane + 64 * ch;
      if (r < nefc && r < 256) { dbg[1152 + r] = rr[ch].aref; dbg[1408 + r] = rr[ch].D; }
    }
    {
      const int rcon = nefc - C::NPYR * s.ncon;
      for (int t = lane; t < nefc * C::NV && t < 4300; t += 64) {
        int r = t / C::NV, i = t % C::NV;
        float v;
        if (r < rcon) v = s.x.b.J[r * C::LDJ + i];
        else {
          int c = (r - rcon) / C::NPYR, e = (r - rcon) % C::NPYR, bn = rcon + C::NBC * c, bk = bn + 1 + (e >> 1);
          v = s.x.b.J[bn * C::LDJ + i] + ((e & 1) ? -s.bmu[bk] : s.bmu[bk]) * s.x.b.J[bk * C::LDJ + i];
        }
        dbg[2048 + t] = v;
      }
    }
    for (int t = lane; t < C::NV * 6; t += 64) dbg[6528 + t] = s.cdof[t];
    for (int t = lane; t < C::NB * 3; t += 64) dbg[6800 + t] = s.com[t];
  }
}

// integrate one substep after forward(): implicitfast / Euler, then _advance
template <class C>
__device__ __forceinline__ void integrate(const DModel& mdl, const Hot& h, Smem<C>& s, int lane, float (&Mrow)[C::NV], const FwdOut<C>& f PROF_ARG) {
  const Hot& m = h;
  const int lr = lrec_lane(lane);
  const int4 rj_ids = lrec<C>(h, LQ_J_IDS, lr), rj_ax = lrec<C>(h, LQ_J_AX, lr);     // joint type; (axis z, qposadr, dofadr, -)
  float qacc = f.qacc;
  const bool implicit = implicit_integration<C>(m, s, lane);
  if (implicit) {
    float a[C::NCH], lt[C::NCH];
    if constexpr (C::ROWTREE) {
      const int dl = rowtree_dof<C>(lane);
      const float dd = dl >= 0 ? m.timestep * s.damp[dl] : 0.0f;
      const float dinv_i = rowtree_factor<C, true>(s.M, dd, a, lt, s.scratch_a(), lane);
      qacc = rowtree_solve<C>(a, lt, dinv_i, f.fsmooth + f.qfc, lane);
      qacc = lane < C::NV ? qacc : 0.0f;
    } else if constexpr (C::ROWCHOL) {
      const int dl = rowchol_dof<C>(lane);
      const float dd = dl >= 0 ? m.timestep * s.damp[dl] : 0.0f;
      const float dinv_i = rowchol_factor<C, true, true>(s.M, dd, a, lt, s.scratch_a(), lane);
      qacc = rowchol_solve<C>(a, lt, dinv_i, f.fsmooth + f.qfc, lane);
      qacc = lane < C::NV ? qacc : 0.0f;
    } else if constexpr (C::ARROW) {
      const int dl = arrow_dof<C>(lane);
      const float dd = dl >= 0 ? m.timestep * s.damp[dl] : 0.0f;
      const float dinv_i = arrow_factor<C, true>(s.M, dd, a, lt, s.scratch_a(), lane);
      qacc = arrow_solve<C>(a, lt, dinv_i, f.fsmooth + f.qfc, lane);
      qacc = lane < C::NV ? qacc : 0.0f;
    } else {
    float dd = lane < C::NV ? m.timestep * s.damp[lane] : 0.0f;
    if constexpr (C::MROW_LDS) load_mrow<C>(s, lane, Mrow);
#pragma unroll
    for (int j = 0; j < C::NV; ++j) a[j] = Mrow[j] + (j == lane ? dd : 0.0f);
    const float dinv_i = chol_factor<C, true>(a, lt, s.scratch_a(), lane);
    qacc = lane < C::NV ? chol_solve<C>(a, lt, dinv_i, f.fsmooth + f.qfc, lane) : 0.0f;
    }
  }
  PROF(PS_X7)
  WSYNC();
  if (lane < C::NV) s.qvel[lane] += qacc * m.timestep;
  WSYNC();
  if (lane < C::NJ) {
    const int qa = rj_ax.y, da = rj_ax.z;
    float dt = m.timestep;
    if (rj_ids.z == JNT_FREE) {
      s.qpos[qa] += dt * s.qvel[da]; s.qpos[qa + 1] += dt * s.qvel[da + 1]; s.qpos[qa + 2] += dt * s.qvel[da + 2];
      V3 w = ld3(&s.qvel[da + 3]);
      float n = fsqrt(dot(w, w));
      V3 ax = n > RSR_MINVAL ? w * frcp(n) : v3(0, 0, 0);
      float sn, cs;
      sincosf(0.5f * dt * n, &sn, &cs);
      Q4 q = qmul(ld4(&s.qpos[qa + 3]), Q4{cs, ax.x * sn, ax.y * sn, ax.z * sn});
      float qn = fsqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
      if (qn < RSR_MINVAL) q = Q4{1, 0, 0, 0};
      else { float inv = frcp(qn); q = Q4{q.w * inv, q.x * inv, q.y * inv, q.z * inv}; }
      st4(&s.qpos[qa + 3], q);
    } else {
      s.qpos[qa] += dt * s.qvel[da];
    }
  }
  WSYNC();
  if constexpr (!C::ARROW) __builtin_amdgcn_s_setprio(0);       // (raised at the top of solve())
  PROF(PS_INTEG)
}

}  // namespace rsr
