// rsr_mjx.hip -- fused env kernels (reset, step) and the C ABI of librsrmjx.so (include/rsr_mjx.h).
// gfx950 only.  One wavefront per environment; see rsr_device.hpp for the execution model.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rsr_mjx.h"
#include "rsr_solver.hpp"

namespace rsr {

// Airbot cube: nq 22, nv 20, nu 5, nbody 14, njnt 10, ngeom 23, nsite 1, npair 45, neq 1, nf 8, nl 8
#ifndef RSR_CUBE_NCON
#define RSR_CUBE_NCON 24
#endif
using CubeDims = Dims<22, 20, 5, 14, 10, 23, 1, 45, 1, 8, 8, /*NCON*/ RSR_CUBE_NCON, /*OBS*/ 23, /*NMET*/ 3, 0, 0, 4, 0,
                      /*ISO: the target body's free joint, dofs 8..13*/ 8, 14, false, false, false, /*NGA*/ 23, false, false,
                      /*TREE1, TREE2: arm | target | cube*/ 8, 14>;
// Airbot T-shape: nq 15, nv 14, njnt 9, ngeom 25, nsite 3, npair 60; 4 env geoms at env_ids[5..8]
// Unitree Go2 feet-only: nq 19, nv 18, nu 12, 13 joints, 39 geoms, 6 sites, 4 sphere-plane pairs of condim 3
using Go2Dims = Dims<19, 18, 12, 14, 13, 39, 6, 4, /*NEQ*/ 0, /*NF*/ 12, /*NL*/ 12, /*NCON*/ 4, /*OBS*/ 48, /*NMET*/ 22, 0, 0, /*CONDIM*/ 3,
                     /*NINFO*/ 144, /*ISO*/ 0, 0, /*DREX*/ true, /*HFIELD*/ true, /*TALIAS*/ false, /*NGA: floor or height field + four feet*/ 5, /*TTAIL*/ true, /*ARROW*/ true>;
// the same without the height-field narrow phase, for models whose floor is a plane (the flat-terrain joystick): the kernel is picked
// by the model (rsr_model_create: any PAIR_HFIELD_SPHERE pair)
using Go2FlatDims = Dims<19, 18, 12, 14, 13, 39, 6, 4, /*NEQ*/ 0, /*NF*/ 12, /*NL*/ 12, /*NCON*/ 4, /*OBS*/ 48, /*NMET*/ 22, 0, 0, /*CONDIM*/ 3,
                     /*NINFO*/ 144, /*ISO*/ 0, 0, /*DREX*/ true, /*HFIELD*/ false, /*TALIAS*/ false, /*NGA: floor or height field + four feet*/ 5, /*TTAIL*/ true, /*ARROW*/ true>;
// Unitree Go2 with every collision geom against the floor (go2_mjx.xml + scene_mjx_flat_terrain.xml, the Handstand / Footstand tasks): 44 geoms,
// 30 plane pairs of condim 3 (4 spheres, 20 capsules, 6 cylinders: up to 62 contact points, 12 kept active per env -- a state with
// more is a fall, which ends the episode in the same step)
using HandDims = Dims<19, 18, 12, 14, 13, 44, 6, 30, /*NEQ*/ 0, /*NF*/ 12, /*NL*/ 12, /*NCON*/ 12, /*OBS*/ 45, /*NMET*/ 11, 0, 0, /*CONDIM*/ 3,
                      /*NINFO*/ 144, /*ISO*/ 0, 0, /*DREX*/ true, /*HFIELD*/ false, /*TALIAS*/ false, /*NGA*/ 44, /*TTAIL*/ true, /*ARROW*/ true, 0, 0, /*CAPS*/ true>;
using TShapeDims = Dims<15, 14, 5, 14, 9, 25, 3, 60, 1, 8, 8, /*NCON*/ 32, /*OBS*/ 16, /*NMET*/ 5, /*NEG*/ 4, /*EG0*/ 5, /*CONDIM*/ 4, 0, 0, 0, false, false, false,
                        /*NGA*/ 25, false, false, /*TREE1, TREE2: arm | T block*/ 8, 14>;

// env_ids layout (rsr_mjx_amd/envs/config.py)
enum { ID_CUBE = 0, ID_TARGET = 1, ID_SITE = 2, ID_BOXQ = 3, ID_SITEQ = 4, ID_FINGERQ = 5, ID_JOINTQ = 6 };

// ---------------------------------------------------------------- threefry2x32 (jax.random default PRNG)
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ void threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t& o0, uint32_t& o1) {
  const int R[8] = {13, 15, 26, 6, 17, 29, 16, 24};
  uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1];
#pragma unroll
  for (int g = 0; g < 5; ++g) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { x0 += x1; x1 = rotl32(x1, R[(g & 1) * 4 + k]); x1 ^= x0; }
    x0 += ks[(g + 1) % 3];
    x1 += ks[(g + 2) % 3] + (uint32_t)(g + 1);
  }
  o0 = x0; o1 = x1;
}
// bits[0..n) = threefry_2x32(key, iota(n)) in jax's split-halves layout; lane-parallel, result in LDS
__device__ void random_bits(uint32_t k0, uint32_t k1, int n, uint32_t* bits, int lane) {
  int half = (n + 1) / 2;
  if (lane < half) {
    uint32_t c1 = (half + lane < n) ? (uint32_t)(half + lane) : 0u, o0, o1;
    threefry2x32(k0, k1, (uint32_t)lane, c1, o0, o1);
    bits[lane] = o0;
    if (half + lane < n) bits[half + lane] = o1;
  }
}
__device__ __forceinline__ float uniform_from_bits(uint32_t b, float lo, float hi) {
#pragma clang fp contract(off)   // jax does a separate multiply and add (HIP's __fmul_rn is a plain '*')
  float u = __uint_as_float((b >> 9) | 0x3F800000u) - 1.0f;
  float scale = hi - lo;
  float prod = u * scale;
  float v = prod + lo;
  return fmaxf(lo, v);
}

// ---------------------------------------------------------------- record I/O
template <class C>
__device__ void load_overrides(const DModel& m, Smem<C>& s, const StepArgs& a, int e, int lane) {
  if (lane < 4) s.rw[C::NEFC + lane] = 0.0f;                 // zero weight of the null row
  // Per-env leaf or the model's own: the source POINTER is selected, then every value is read in one batch of global loads and
  // stored to LDS after one wait.  (A branch per leaf -- `dr ? dr[..] : m.x[..]` -- made each leaf's load wait on its own: up
  // to nine global round trips in a row at the start of every work unit.)
  auto src = [&](const float* dr, gp_f own, int per_env) { return dr ? (gp_f)(dr + (size_t)e * per_env) : own; };
  const gp_f p_fric = src(a.dr_geom_friction, m.geom_friction, C::NG * 3), p_mass = src(a.dr_body_mass, m.body_mass, C::NB);
  const gp_f p_damp = src(a.dr_dof_damping, m.dof_damping, C::NV), p_floss = src(a.dr_dof_frictionloss, m.dof_frictionloss, C::NV);
  constexpr int NFR = (C::NGA * 3 + 63) / 64;
  float v_fric[NFR];
#pragma unroll
  for (int k = 0; k < NFR; ++k) {                             // friction of the geom slots (geoms of the contact pairs)
    const int t = lane + 64 * k, tt = t < C::NGA * 3 ? t : 0;
    const int sidx = C::NGA == C::NG ? tt : 3 * m.geom_slot_ids[tt / 3] + tt % 3;
    v_fric[k] = p_fric[sidx];
  }
  const float v_mass = p_mass[lane < C::NB ? lane : 0];
  const int dl = lane < C::NV ? lane : 0;
  const float v_damp = p_damp[dl], v_floss = p_floss[dl];
  float v_ipos[2] = {0, 0}, v_q0 = 0, v_arma = 0, v_gain = 0, v_bias = 0;
  if constexpr (C::DREX) {
    static_assert(C::NB * 3 <= 64 && C::NQ <= 64 && C::NU * 3 <= 64, "one lane per extended leaf entry");
    const gp_f p_ipos = src(a.dr_body_ipos, m.body_ipos, C::NB * 3), p_q0 = src(a.dr_qpos0, m.qpos0, C::NQ);
    const gp_f p_arma = src(a.dr_dof_armature, m.dof_armature, C::NV);
    const gp_f p_gain = src(a.dr_gainprm, m.actuator_gainprm, C::NU * 3), p_bias = src(a.dr_biasprm, m.actuator_biasprm, C::NU * 3);
    v_ipos[0] = p_ipos[lane < C::NB * 3 ? lane : 0]; v_q0 = p_q0[lane < C::NQ ? lane : 0]; v_arma = p_arma[dl];
    v_gain = p_gain[lane < C::NU * 3 ? lane : 0]; v_bias = p_bias[lane < C::NU * 3 ? lane : 0];
  }
#pragma unroll
  for (int k = 0; k < NFR; ++k) { const int t = lane + 64 * k; if (t < C::NGA * 3) s.fric[t] = v_fric[k]; }
  if (lane < C::NB) s.mass[lane] = v_mass;
  if (lane < C::NV) { s.damp[lane] = v_damp; s.floss[lane] = v_floss; }
  if constexpr (C::DREX) {
    if (lane < C::NB * 3) s.dx_ipos[lane] = v_ipos[0];
    if (lane < C::NQ) s.dx_qpos0[lane] = v_q0;
    if (lane < C::NV) s.dx_arma[lane] = v_arma;
    if (lane < C::NU * 3) { s.dx_gain[lane] = v_gain; s.dx_bias[lane] = v_bias; }
  }
}

// cube_env.py:215-229
template <class C>
__device__ void cube_obs(const DModel& m, const Smem<C>& s, const float* target_pos, const float* ncp, float* obs) {
  const int cube = m.env_ids[ID_CUBE], site = m.env_ids[ID_SITE];
  for (int i = 0; i < 6; ++i) obs[i] = s.qpos[m.env_ids[ID_JOINTQ + i]];
  for (int i = 0; i < 3; ++i) {
    float sp = s.spos[3 * site + i], cp = s.xpos[3 * cube + i], tp = target_pos[i];
    obs[6 + i] = sp; obs[9 + i] = tp; obs[12 + i] = cp; obs[17 + i] = tp - cp; obs[20 + i] = cp - sp;
  }
  obs[15] = ncp[0]; obs[16] = ncp[1];
}

// T-shape env_ids layout (rsr_mjx_amd/envs/config.py: tshape_env_fields); egeom[0..3] = base_block, vertical_block,
// base_target, vertical_target
enum { TID_T = 0, TID_TARGET = 1, TID_SITE = 2, TID_TAIL = 3, TID_TTAIL = 4, TID_GBASE = 5, TID_JOINTQ = 9 };

// T_shape_env.py:223-234
template <class C>
__device__ void tshape_obs(const DModel& m, const Smem<C>& s, const float* tb, const float* tv, float xita, const float* newT,
                           float* obs) {
  const int site = m.env_ids[TID_SITE];
  for (int i = 0; i < 6; ++i) obs[i] = s.qpos[m.env_ids[TID_JOINTQ + i]];
  obs[6] = s.spos[3 * site + 2];
  for (int i = 0; i < 3; ++i) { obs[7 + i] = tb[i] - s.egeom[i]; obs[10 + i] = tv[i] - s.egeom[3 + i]; }
  obs[13] = xita;
  obs[14] = newT[0] - s.spos[3 * site]; obs[15] = newT[1] - s.spos[3 * site + 1];
}

template <class C>
__device__ void store_pipeline(Smem<C>& s, float* rec, const Layout& L, int lane, float warm, float time) {
  for (int t = lane; t < C::NQ; t += 64) rec[L.qpos + t] = s.qpos[t];
  if (lane < C::NV) { rec[L.qvel + lane] = s.qvel[lane]; rec[L.warm + lane] = warm; }
  if (lane < C::NU) rec[L.ctrl + lane] = s.ctrl[lane];
  if (lane == 0) rec[L.time] = time;
  for (int t = lane; t < C::NB * 3; t += 64) rec[L.xpos + t] = s.xpos[t];
  for (int t = lane; t < C::NS * 3; t += 64) rec[L.site_xpos + t] = s.spos[t];
}

// ================================================================ Go2 joystick env (go2/joystick.py)
// ginfo layout = oracle enum G2_* ; env_go2f / env_go2i / env_ids as documented in rsr_mjx_amd/envs/config.py
enum { G2_CMD = 0, G2_STEPS_CMD = 3, G2_LAST_ACT = 4, G2_LAST_LAST_ACT = 16, G2_AIR = 28, G2_CONTACT_T = 32, G2_LAST_CONTACT = 36,
       G2_SWING = 40, G2_ACT_BUF = 44, G2_GYRO_BUF = 92, G2_LINVEL_BUF = 104, G2_GRAV_BUF = 116, G2_STEPS_PERT = 128,
       G2_PERT_DUR_S = 129, G2_PERT_DUR = 130, G2_SINCE_PERT = 131, G2_PERT_STEPS = 132, G2_PERT_DIR = 133, G2_PERT_MAG = 136,
       G2_RNG = 137, G2_XFRC = 139 /* data.xfrc_applied[torso, :3] */ };
constexpr int GO2_PRIV = 123;    // obs['privileged_state'], joystick.py:341-366
enum { RW_TRACK_LIN = 0, RW_TRACK_ANG, RW_LIN_VEL_Z, RW_ANG_VEL_XY, RW_ORIENT, RW_DOF_LIMITS, RW_POSE, RW_TERM, RW_STAND_STILL,
       RW_TORQUES, RW_ACTION_RATE, RW_ENERGY, RW_FEET_CLEAR, RW_FEET_HEIGHT, RW_FEET_SLIP, RW_FEET_AIR, RW_ALL_FEET_AIR,
       RW_SYM_GAIT, RW_LR_SYM, RW_FB_SYM, RW_FEET_OFF_STILL, RW_COUNT };

// jax.random.split(key, N): every lane receives all N keys (wave-uniform); bits = LDS scratch of >= 2N words
template <int N>
__device__ __forceinline__ void tf_split(uint32_t k0, uint32_t k1, uint32_t* bits, int lane, uint32_t (&out)[N][2]) {
  WSYNC();
  random_bits(k0, k1, 2 * N, bits, lane);
  WSYNC();
#pragma unroll
  for (int r = 0; r < N; ++r) { out[r][0] = bits[2 * r]; out[r][1] = bits[2 * r + 1]; }
}
// jax.random.uniform(key, (n,), lo, hi): lane i < n returns element i
__device__ __forceinline__ float tf_uniform(uint32_t k0, uint32_t k1, int n, float lo, float hi, uint32_t* bits, int lane) {
  WSYNC();
  random_bits(k0, k1, n, bits, lane);
  WSYNC();
  return lane < n ? uniform_from_bits(bits[lane], lo, hi) : 0.0f;
}

struct G2Sens { float gyro[3], linvel[3], gravity[3], up[3], glin[3], gang[3], accel[3]; };
// element i (0..2, a lane index) of a sensor triple by selects: indexing the register array with a lane index would put the
// whole struct into scratch memory
// (the three values pass through an empty asm: a select between loads of the struct would be rewritten into one load through
// a selected address, which pins the struct in memory just the same)
__device__ __forceinline__ float pick3(const float (&v)[3], int i) {
  float a = v[0], b = v[1], c = v[2];
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
  return i == 0 ? a : (i == 1 ? b : c);
}

template <class C>
__device__ __forceinline__ void go2_sensors(const DModel& m, const Smem<C>& s, G2Sens& o) {
  const int imu = m.env_ids[0];
  const float* R = &s.smat[9 * imu];
  V3 w = ld3(&s.sangvel[3 * imu]), v = ld3(&s.slinvel[3 * imu]);
  // site-frame quantities: R^T x
  o.gyro[0] = R[0] * w.x + R[3] * w.y + R[6] * w.z; o.gyro[1] = R[1] * w.x + R[4] * w.y + R[7] * w.z; o.gyro[2] = R[2] * w.x + R[5] * w.y + R[8] * w.z;
  o.linvel[0] = R[0] * v.x + R[3] * v.y + R[6] * v.z; o.linvel[1] = R[1] * v.x + R[4] * v.y + R[7] * v.z; o.linvel[2] = R[2] * v.x + R[5] * v.y + R[8] * v.z;
  o.gravity[0] = R[0] * 0.0f + R[3] * 0.0f + R[6] * -1.0f; o.gravity[1] = R[1] * 0.0f + R[4] * 0.0f + R[7] * -1.0f; o.gravity[2] = R[2] * 0.0f + R[5] * 0.0f + R[8] * -1.0f;
  o.up[0] = R[2]; o.up[1] = R[5]; o.up[2] = R[8];
  o.glin[0] = v.x; o.glin[1] = v.y; o.glin[2] = v.z; o.gang[0] = w.x; o.gang[1] = w.y; o.gang[2] = w.z;
}

// jax.random.split(key, N) in registers: the 2N output words are threefry(key, (j, N + j)) of lanes j < N -- word t is the first
// output of lane t for t < N and the second output of lane t - N otherwise -- fetched with v_readlane: no LDS, no barrier.
template <int N>
__device__ __forceinline__ void tf_split_reg(uint32_t k0, uint32_t k1, int lane, uint32_t (&out)[N][2]) {
  uint32_t o0, o1;
  threefry2x32(k0, k1, (uint32_t)lane, (uint32_t)(N + lane), o0, o1);
#pragma unroll
  for (int r = 0; r < N; ++r)
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const int t = 2 * r + w;
      out[r][w] = (uint32_t)(t < N ? rdlane_i((int)o0, t) : rdlane_i((int)o1, t - N));
    }
}
// One lane's share of a batch of jax.random.uniform draws evaluated together: this lane is word pair `idx` of a draw of n
// elements with key (k0, k1), whose bits go to bits[off .. off + n) (same split-halves layout as random_bits).
__device__ __forceinline__ void tf_bits_batched(uint32_t k0, uint32_t k1, int n, int idx, int off, bool on, uint32_t* bits) {
  const int half = (n + 1) / 2;
  const bool two = half + idx < n;
  uint32_t o0, o1;
  threefry2x32(k0, k1, (uint32_t)idx, two ? (uint32_t)(half + idx) : 0u, o0, o1);
  if (on) { bits[off + idx] = o0; if (two) bits[off + half + idx] = o1; }
}

// joystick.py:284-340: 48-dim "state" obs into obs_lds; advances ginfo rng by five splits.  The five splits are a serial
// chain of register-only evaluations; the five draws (3, 3, 3, 12, 12 elements) are then one evaluation with a lane per
// word pair and one barrier (ten evaluations with two barriers each when every split and draw went through LDS).
template <class C>
__device__ __forceinline__ void go2_obs(const DModel& m, Smem<C>& s, const G2Sens& sn, float* obs_lds, uint32_t* bits, int lane, float home_l) {
#pragma clang fp contract(off)
  const float* F = m.env_go2f;
  const bool idel = m.env_go2i[1] > 0;
  uint32_t rng0 = __float_as_uint(s.ginfo[G2_RNG]), rng1 = __float_as_uint(s.ginfo[G2_RNG + 1]);
  rng0 = (uint32_t)uniform_i((int)rng0); rng1 = (uint32_t)uniform_i((int)rng1);
  const float level = F[2];
  // order of the draws: gyro, gravity, linvel, joint angles, joint velocities
  uint32_t dk[5][2];
#pragma unroll
  for (int d = 0; d < 5; ++d) {
    uint32_t ks[2][2];
    tf_split_reg<2>(rng0, rng1, lane, ks);
    rng0 = ks[0][0]; rng1 = ks[0][1]; dk[d][0] = ks[1][0]; dk[d][1] = ks[1][1];
  }
  {
    // lanes [0,2) [2,4) [4,6): the three 3-element draws; [6,12) [12,18): the two 12-element draws
    const int d = lane < 6 ? (lane >> 1) : (lane < 12 ? 3 : 4);
    const int idx = lane < 6 ? (lane & 1) : (lane < 12 ? lane - 6 : lane - 12);
    uint32_t k0 = dk[4][0], k1 = dk[4][1];
#pragma unroll
    for (int q = 3; q >= 0; --q) if (d == q) { k0 = dk[q][0]; k1 = dk[q][1]; }
    WSYNC();
    tf_bits_batched(k0, k1, d < 3 ? 3 : 12, idx, d < 3 ? 3 * d : (d == 3 ? 9 : 21), lane < 18, bits);
    WSYNC();
  }
#pragma unroll
  for (int d = 0; d < 5; ++d) {
    const int n = d < 3 ? 3 : 12, off = d < 3 ? 3 * d : (d == 3 ? 9 : 21);
    if (lane < n) {
      const float u = uniform_from_bits(bits[off + lane], 0.0f, 1.0f);
      float src, scale; int dst;
      if (d == 0) { src = idel ? s.ginfo[G2_GYRO_BUF + lane] : pick3(sn.gyro, lane); scale = F[5]; dst = 3 + lane; }
      else if (d == 1) { src = idel ? s.ginfo[G2_GRAV_BUF + lane] : pick3(sn.gravity, lane); scale = F[6]; dst = 6 + lane; }
      else if (d == 2) { src = idel ? s.ginfo[G2_LINVEL_BUF + lane] : pick3(sn.linvel, lane); scale = F[7]; dst = lane; }
      else if (d == 3) { src = s.qpos[7 + lane]; scale = F[3]; dst = 9 + lane; }
      else { src = s.qvel[6 + lane]; scale = F[4]; dst = 21 + lane; }
      float a = 2.0f * u; float b = a - 1.0f; float c = b * level; float e = c * scale;
      float val = src + e;
      if (d == 3) val = val - home_l;             // home_l = env_go2_home[7 + lane] (lanes < 12), loaded by the caller ahead of time
      obs_lds[dst] = val;
    }
  }
  if (lane < 12) obs_lds[33 + lane] = s.ginfo[G2_LAST_ACT + lane];
  if (lane < 3) obs_lds[45 + lane] = s.ginfo[G2_CMD + lane];
  if (lane == 0) { s.ginfo[G2_RNG] = __uint_as_float(rng0); s.ginfo[G2_RNG + 1] = __uint_as_float(rng1); }
  WSYNC();
}

// accelerometer of the IMU site (MuJoCo sensor_acc: rne_postconstraint cacc + objectAcceleration, local frame):
// cacc = accb (velocity-product part saved by smooth_forces) + sum over the body's chain of cdof * qacc, moved to the site
// (lin + ang x dif), rotated into the site frame, plus w_local x v_local.  Wave-cooperative; every lane gets the result.
template <class C>
__device__ __forceinline__ void go2_accelerometer(const DModel& m, const Smem<C>& s, int lane, float qacc_i, G2Sens& o) {
  const int imu = m.env_ids[0], b = m.site_bodyid[imu];
  const bool on = lane < C::NV && ((m.body_dofmask[b] >> lane) & 1);
  float c6[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) c6[c] = on ? s.cdof[6 * lane + c] * qacc_i : 0.0f;
  wave_sum3(c6[0], c6[1], c6[2]); wave_sum3(c6[3], c6[4], c6[5]);
#pragma unroll
  for (int c = 0; c < 6; ++c) c6[c] += s.accb[c];
  V3 dif = ld3(&s.spos[3 * imu]) - ld3(&s.com[3 * m.body_rootid[b]]);
  V3 ang = v3(c6[0], c6[1], c6[2]), lin = v3(c6[3], c6[4], c6[5]) + cross(ang, dif);
  const float* R = &s.smat[9 * imu];
  V3 w = ld3(&s.sangvel[3 * imu]), v = ld3(&s.slinvel[3 * imu]);
  auto rt = [&](V3 x) { return v3(R[0] * x.x + R[3] * x.y + R[6] * x.z, R[1] * x.x + R[4] * x.y + R[7] * x.z, R[2] * x.x + R[5] * x.y + R[8] * x.z); };
  V3 al = rt(lin), wl = rt(w), vl = rt(v), cr = cross(wl, vl);
  o.accel[0] = al.x + cr.x; o.accel[1] = al.y + cr.y; o.accel[2] = al.z + cr.z;
}

// joystick.py:341-366: element t of obs["privileged_state"]; read right after go2_obs (info: old last_contact, air + dt)
template <class C>
__device__ __forceinline__ float go2_priv_elem(const DModel& m, const Smem<C>& s, const G2Sens& sn, const float* obs_lds, int t) {
  if (t < 48) return obs_lds[t];
  t -= 48;
  if (t < 3) return pick3(sn.gyro, t);
  if (t < 6) return pick3(sn.accel, t - 3);
  if (t < 9) return pick3(sn.gravity, t - 6);
  if (t < 12) return pick3(sn.linvel, t - 9);
  if (t < 15) return pick3(sn.gang, t - 12);
  if (t < 27) return s.qpos[7 + t - 15] - m.env_go2_home[7 + t - 15];
  if (t < 39) return s.qvel[6 + t - 27];
  if (t < 51) return s.aforce[t - 39];
  if (t < 55) return s.ginfo[G2_LAST_CONTACT + t - 51];
  if (t < 67) { int k = t - 55; return s.slinvel[3 * m.env_ids[1 + k / 3] + k % 3]; }
  if (t < 71) return s.ginfo[G2_AIR + t - 67];
  if (t < 74) return s.ginfo[G2_XFRC + t - 71];
  return s.ginfo[G2_SINCE_PERT] >= s.ginfo[G2_STEPS_PERT] ? 1.0f : 0.0f;
}

// The same element fetched as a gather: every element but the 15 sensor values (registers) and the kick flag is one LDS
// word, so the lanes compute an address with selects and issue ONE load (go2_priv_elem walks fifteen divergent branches,
// each waiting for its own load).  FIRST: t < 64 (the only elements that can be sensor values).  foot_site: env_ids[1..4].
template <class C, bool FIRST>
__device__ __forceinline__ float go2_priv_gather(const DModel& m, const Smem<C>& s, const G2Sens& sn, const float* obs_lds, int t,
                                                 const int (&foot_site)[4], float kick_flag, float home_l) {
#pragma clang fp contract(off)
  const int k = t - 48;
  const float* p = obs_lds + (t < 48 ? t : 0);
  p = (k >= 15 && k < 27) ? &s.qpos[7 + (k - 15)] : p;
  p = (k >= 27 && k < 39) ? &s.qvel[6 + (k - 27)] : p;
  p = (k >= 39 && k < 51) ? &s.aforce[k - 39] : p;
  p = (k >= 51 && k < 55) ? &s.ginfo[G2_LAST_CONTACT + (k - 51)] : p;
  {
    const int kk = k - 55, ft = kk / 3;
    const int site = ft == 0 ? foot_site[0] : ft == 1 ? foot_site[1] : ft == 2 ? foot_site[2] : foot_site[3];
    p = (k >= 55 && k < 67) ? &s.slinvel[3 * site + (kk - 3 * ft)] : p;
  }
  p = (k >= 67 && k < 71) ? &s.ginfo[G2_AIR + (k - 67)] : p;
  p = (k >= 71 && k < 74) ? &s.ginfo[G2_XFRC + (k - 71)] : p;
  const int hk = k - 15 < 0 ? 0 : (k - 15 > 11 ? 11 : k - 15);
  const float h = __shfl(home_l, hk);            // lane j holds home[7 + j]
  float v = *p;
  if (k >= 15 && k < 27) v = v - h;
  if constexpr (FIRST) {
    const int grp = k < 0 ? 0 : k / 3, comp = k < 0 ? 0 : k - 3 * grp;
    const float v_gy = pick3(sn.gyro, comp), v_ac = pick3(sn.accel, comp), v_gr = pick3(sn.gravity, comp), v_li = pick3(sn.linvel, comp), v_ga = pick3(sn.gang, comp);
    const float sv = grp == 0 ? v_gy : (grp == 1 ? v_ac : (grp == 2 ? v_gr : (grp == 3 ? v_li : v_ga)));
    if (k >= 0 && k < 15) v = sv;
  }
  if (k == 74) v = kick_flag;
  return v;
}

// ---------------------------------------------------------------- reset kernel
// cube / sf: cube_env.py:95-143 ; T-shape: T_shape_env.py:98-137 ; + Episode/AutoReset wrapper resets
template <class C, int ENV>
__global__ __launch_bounds__(64) void reset_kernel(const DModel* __restrict__ mp, Layout L, StepArgs a) {
  const DModel& m = *mp;
  const Hot hot = make_hot(m);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem<C>& s = *reinterpret_cast<Smem<C>*>(smem_raw);
  const int e = blockIdx.x, lane = threadIdx.x;
  if (e >= a.n) return;
  float* rec = a.state + (size_t)e * L.rec;
  const float* R = m.env_reset;
  constexpr int JQ = ENV == ENV_TSHAPE ? (int)TID_JOINTQ : (int)ID_JOINTQ;     // arm joint qpos addresses in env_ids
  constexpr int RCTRL = ENV == ENV_TSHAPE ? 7 : 8;                            // ctrl init in env_reset
  uint32_t* bits = reinterpret_cast<uint32_t*>(s.scratch_b());         // PRNG scratch
  load_overrides<C>(m, s, a, e, lane);
  const uint32_t k0 = a.keys[2 * e], k1 = a.keys[2 * e + 1];
  random_bits(k0, k1, 10, bits, lane);                        // rng, rng1..rng4 = split(rng, 5)
  WSYNC();
  uint32_t kk[5][2];
#pragma unroll
  for (int r = 0; r < 5; ++r) { kk[r][0] = bits[2 * r]; kk[r][1] = bits[2 * r + 1]; }
  WSYNC();
  const float lo = -R[0], hi = R[0];
  random_bits(kk[1][0], kk[1][1], C::NQ, bits, lane);
  WSYNC();
  if (lane < C::NQ) s.qpos[lane] = m.qpos0[lane] + uniform_from_bits(bits[lane], lo, hi);
  WSYNC();
  if (lane < 6) s.qpos[m.env_ids[JQ + lane]] += R[1 + lane];
  if (ENV != ENV_TSHAPE && lane == 6) s.qpos[m.env_ids[ID_FINGERQ]] = R[7];
  random_bits(kk[2][0], kk[2][1], C::NV, bits, lane);
  WSYNC();
  if (lane < C::NV) s.qvel[lane] = uniform_from_bits(bits[lane], lo, hi);
  WSYNC();
  random_bits(kk[3][0], kk[3][1], C::NU, bits, lane);
  WSYNC();
  float ctrl_init = lane < C::NU ? R[RCTRL + lane] + uniform_from_bits(bits[lane], lo, hi) : 0.0f;
  WSYNC();
  if constexpr (ENV != ENV_TSHAPE) {
    random_bits(kk[4][0], kk[4][1], 3, bits, lane);
    WSYNC();
    if (lane < 3) s.qpos[m.env_ids[ID_SITEQ] + lane] = uniform_from_bits(bits[lane], R[13 + lane], R[16 + lane]);
    WSYNC();
    random_bits(kk[0][0], kk[0][1], 3, bits, lane);
    WSYNC();
    if (lane < 3) s.qpos[m.env_ids[ID_BOXQ] + lane] = uniform_from_bits(bits[lane], R[19 + lane], R[22 + lane]);
  }
  if (lane < C::NU) s.ctrl[lane] = 0.0f;                      // pipeline_init runs forward with ctrl = 0
  WSYNC();
  float Mrow[C::NV], warm = 0.0f;
  FwdOut<C> f;
  PROF_DECL
  forward<C>(m, hot, s, lane, Mrow, warm, f, a.debug ? a.debug + (size_t)e * RSR_DEBUG_FLOATS : nullptr PROF_PASS);
  WSYNC();
  if (lane < C::NU) s.ctrl[lane] = ctrl_init;                 // data.replace(ctrl=joint_ctrl), no re-forward
  WSYNC();
  store_pipeline<C>(s, rec, L, lane, warm, 0.0f);
  if (lane == 0) {
    float obs[C::OBS];
    if constexpr (ENV == ENV_TSHAPE) {
      const int site = m.env_ids[TID_SITE], tb = m.env_ids[TID_T];
      float newT[2] = {R[12], R[13]};
      for (int i = 0; i < 3; ++i) {
        rec[L.target_base_pos + i] = s.egeom[6 + i]; rec[L.target_vertical_pos + i] = s.egeom[9 + i];
        rec[L.site_pos + i] = s.spos[3 * site + i]; rec[L.T_pos + i] = s.xpos[3 * tb + i];
      }
      rec[L.target_w] = s.xquat[4 * m.env_ids[TID_TARGET]] * 10.0f;
      rec[L.new_T_pos] = newT[0]; rec[L.new_T_pos + 1] = newT[1];
      rec[L.xita] = R[14];
      tshape_obs<C>(m, s, &s.egeom[6], &s.egeom[9], R[14], newT, obs);
    } else {
      const int cube = m.env_ids[ID_CUBE], tgt = m.env_ids[ID_TARGET], site = m.env_ids[ID_SITE];
      float tp[3], ncp[2] = {R[25], R[26]};
      for (int i = 0; i < 3; ++i) {
        tp[i] = s.xpos[3 * tgt + i];
        rec[L.target_pos + i] = tp[i];
        rec[L.site_pos + i] = s.spos[3 * site + i];
        rec[L.cube_pos + i] = s.xpos[3 * cube + i];
      }
      rec[L.new_cube_pos] = ncp[0]; rec[L.new_cube_pos + 1] = ncp[1];
      rec[L.last_action] = 0.0f;
      cube_obs<C>(m, s, tp, ncp, obs);
    }
    for (int i = 0; i < C::OBS; ++i) { rec[L.obs + i] = obs[i]; rec[L.f_obs + i] = obs[i]; }
    rec[L.reward] = 0.0f; rec[L.done] = 0.0f;
    for (int i = 0; i < C::NMET; ++i) rec[L.metrics + i] = 0.0f;
    rec[L.steps] = 0.0f; rec[L.truncation] = 0.0f; rec[L.episode_done] = 0.0f;
    for (int i = 0; i < 2 + C::NMET; ++i) rec[L.episode_metrics + i] = 0.0f;
    int* st = reinterpret_cast<int*>(rec + L.stats);
    st[0] = f.st.niter; st[1] = f.st.ls_total; st[2] = s.ncon; st[3] = s.ncon_drop;
  }
  // AutoResetWrapper.reset: cache first_pipeline_state (same field order as the live block)
  for (int t = lane; t < C::NQ; t += 64) rec[L.f_qpos + t] = s.qpos[t];
  if (lane < C::NV) { rec[L.f_qvel + lane] = s.qvel[lane]; rec[L.f_warm + lane] = warm; }
  if (lane < C::NU) rec[L.f_ctrl + lane] = s.ctrl[lane];
  if (lane == 0) rec[L.f_time] = 0.0f;
  for (int t = lane; t < C::NB * 3; t += 64) rec[L.f_xpos + t] = s.xpos[t];
  for (int t = lane; t < C::NS * 3; t += 64) rec[L.f_site_xpos + t] = s.spos[t];
}

// ---------------------------------------------------------------- step kernel
// cube / sf: cube_env.py:145-213, test/airbot.py:165-252 ; T-shape: T_shape_env.py:139-221 ; + wrappers.
// LDS is dynamic so that the register budget is set by RSR_WAVES_PER_EU below, not by the compiler's
// LDS-derived occupancy guess (which lands one register over the 2-waves/SIMD budget and halves residency).
#ifndef RSR_WAVES_PER_EU
#define RSR_WAVES_PER_EU 2
#endif
#ifndef RSR_DEFAULT_UNITS
#define RSR_DEFAULT_UNITS 4          // phases per env-step of the work-queue dispatch (measured: DESIGN.md 4)
#endif
template <class C, int ENV>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RSR_WAVES_PER_EU, RSR_WAVES_PER_EU)))
void step_kernel(const DModel* __restrict__ mp, Layout L, StepArgs a, Sched sc) {
  const DModel& m = *mp;
  const Hot hot = make_hot(m);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem<C>& s = *reinterpret_cast<Smem<C>*>(smem_raw);
  const int lane = threadIdx.x;
  const bool wrap_episode = m.wrap_flags & 1, wrap_autoreset = (m.wrap_flags & 2) != 0;
  constexpr int JQ = ENV == ENV_TSHAPE ? (int)TID_JOINTQ : (int)ID_JOINTQ;
  // Ticket space: the first n_whole envs are stepped as ONE unit each (all substeps: no hand-off, no flag, one ticket), the rest
  // as `units` phases each, phase-major.  Long units first, short units last: the launch still drains in short units, and only the
  // envs that start late pay the per-unit overhead (ticket round trip, flag poll, state round trip through memory, store drain).
  const int units = sc.units, n_whole = sc.n_whole, n_split = a.n - n_whole, total = n_whole + units * n_split;
  int* const ticket = sc.ticket + (sc.launch_id & 1u);
  // A wave's first ticket is its workgroup index where that ticket is a whole-env unit (which nobody waits for) -- 2048 waves
  // drawing from one counter at launch serialise at ~90 atomics per microsecond, ~20 us before the last wave has its first
  // unit -- and the counter hands out the tickets from n_static on.  (Tickets of split envs are only ever drawn from the counter,
  // in dependency order, so the wave that holds (env, phase - 1) is running whatever the residency of the grid.)
  const int n_static = (int)gridDim.x < n_whole ? (int)gridDim.x : n_whole;
  if (blockIdx.x == 0 && lane == 0) __hip_atomic_store(sc.ticket + ((sc.launch_id + 1u) & 1u), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the next launch's counter
  auto draw = [&]() {
    int t = 0;
    if (lane == 0) t = n_static + __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return t;
  };
  int tk_next = (int)blockIdx.x < n_static ? (int)blockIdx.x : draw();
  for (;;) {                                                   // persistent wave: one work unit per trip
  // The next unit's ticket is drawn when the current unit's substeps are done, ahead of its stores / epilogue: the atomic's round
  // trip (~2 us under load, 16 units per wave and launch) overlaps them.  (Drawn at the START of the current unit it binds the
  // last units of a launch to waves that are still busy for a whole unit while others idle: measured -3.4 % on the cube.)
  const int tk = uniform_i(tk_next);
  if (tk >= total) break;                                      // every wave reaches this: the queue only drains
  int phase = 0, e = tk, eu = 1;                               // eu = units of this env's class
  if (tk >= n_whole) { const int t2 = tk - n_whole; phase = t2 / n_split; e = n_whole + (t2 - phase * n_split); eu = units; }
  const bool first = phase == 0, last = phase == eu - 1;
  float* rec = a.state + (size_t)e * L.rec;
  PROF_DECL
  // ---- load the record ----
  // what no other unit of this launch writes first: the per-env model leaves and the env's bookkeeping words, so that their round
  // trip overlaps the flag poll below
  load_overrides<C>(m, s, a, e, lane);
  const float done_prev = rec[L.done];
  float steps = rec[L.steps];
  if (wrap_autoreset && done_prev != 0.0f) steps = 0.0f;     // AutoResetWrapper.step pre-step
  // env info read before it is updated
  float tp[3] = {0, 0, 0}, aux_old[2];
  if constexpr (ENV == ENV_TSHAPE) { aux_old[0] = rec[L.new_T_pos]; aux_old[1] = rec[L.new_T_pos + 1]; }
  else {
    tp[0] = rec[L.target_pos]; tp[1] = rec[L.target_pos + 1]; tp[2] = rec[L.target_pos + 2];
    aux_old[0] = rec[L.new_cube_pos]; aux_old[1] = rec[L.new_cube_pos + 1];
  }
  float warm = 0.0f, time;
  unsigned handoff_err = 0u;
  if (first) {
    for (int t = lane; t < C::NQ; t += 64) s.qpos[t] = rec[L.qpos + t];
    if (lane < C::NV) { s.qvel[lane] = rec[L.qvel + lane]; warm = rec[L.warm + lane]; }
    time = rec[L.time];
  } else {
    // the previous phase of this env (another wave, any CU) has published its state: poll its flag, then read every handed-off
    // word past the caches.  The spin is bounded; a timeout is sticky: counted in sc.err, carried to the env's later phases in
    // the flag's error bit, and reported by the last phase as stats[3] = -1 (the unit runs on whatever the record holds).
    const unsigned want = (sc.launch_id << 8) | (unsigned)phase;
    int spins = 0;
    unsigned fl;
    while (((fl = __hip_atomic_load(sc.flags + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & ~RSR_FLAG_ERR) != want && spins < sc.spin_cap) { __builtin_amdgcn_s_sleep(8); ++spins; }
    fl = (unsigned)uniform_i((int)fl);
    const bool timed_out = (fl & ~RSR_FLAG_ERR) != want;
    handoff_err = timed_out ? RSR_FLAG_ERR : (fl & RSR_FLAG_ERR);
    if (timed_out && lane == 0) {
      __hip_atomic_fetch_add(sc.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sc.err + 1, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int t = lane; t < C::NQ; t += 64) s.qpos[t] = ld_sc1(&rec[L.qpos + t]);
    if (lane < C::NV) { s.qvel[lane] = ld_sc1(&rec[L.qvel + lane]); warm = ld_sc1(&rec[L.warm + lane]); }
    if (lane < C::NU) s.ctrl[lane] = ld_sc1(&rec[L.ctrl + lane]);
    time = ld_sc1(&rec[L.time]);
  }
  // ---- prologue: ctrl shaping; uses the stale xpos / site_xpos of the previous forward pass ----
  if (first && lane < C::NU) {
#pragma clang fp contract(off)   // env algebra is evaluated op by op, as the reference's JAX-CPU path does
    float delta = m.env_action_scale[lane] * a.action[(size_t)e * C::NU + lane];
    float act = rec[L.ctrl + lane] + delta;
    if (lane == 3) act = -((1.57f + rec[L.qpos + m.env_ids[JQ + 1]]) + rec[L.qpos + m.env_ids[JQ + 2]]);
    float delta0 = m.env_action_scale[0] * a.action[(size_t)e * C::NU];
    float act0 = rec[L.ctrl] + delta0;
    if (lane == 4) {
      if constexpr (ENV == ENV_TSHAPE) {      // T_shape_env.py:146-153: aim from the end effector at the T's tail
        const int site = m.env_ids[TID_SITE], tail = m.env_ids[TID_TAIL];
        float dx = rec[L.site_xpos + 3 * tail] - rec[L.site_xpos + 3 * site];
        float dy = rec[L.site_xpos + 3 * tail + 1] - rec[L.site_xpos + 3 * site + 1];
        float ang = atan2f(dy, dx + 0.00001f);
        act = (-ang + act0) + 1.5708f;
      } else {                                // cube_env.py:152-159
        const int cube = m.env_ids[ID_CUBE];
        float dx = tp[0] - rec[L.xpos + 3 * cube], dy = tp[1] - rec[L.xpos + 3 * cube + 1];
        float ang = atan2f(dy, dx + 0.00001f);
        act = (-ang + act0) + 1.5708f;
        if (m.env_kind == ENV_AIRBOT_SF) {    // test/airbot.py:180-184: hold the wrist target within 3 cm of the goal
          float dz = tp[2] - rec[L.xpos + 3 * cube + 2];
          if (sqrtf(dx * dx + dy * dy + dz * dz) < 0.03f) act = rec[L.last_action];
          rec[L.last_action] = act;
        }
      }
    }
    s.ctrl[lane] = clampf(act, m.env_ctrl_lo[lane], m.env_ctrl_hi[lane]);
  }
  WSYNC();
  PROF(PS_LOAD)
  // ---- n_frames x mjx.step ----
  float Mrow[C::NV];
  FwdOut<C> f;
  for (int fr = phase * hot.n_frames / eu; fr < (phase + 1) * hot.n_frames / eu; ++fr) {
#if defined(RSR_PROFILE) || defined(RSR_TIMELINE)
    float* dbg = nullptr;
#else
    float* dbg = (a.debug && fr == m.n_frames - 1) ? a.debug + (size_t)e * RSR_DEBUG_FLOATS : nullptr;
#endif
    // the lane index passes through an opaque zero per substep: values derived from it (masks, LDS addresses) are then
    // recomputed in each substep instead of being hoisted out of the loop, kept live across the solver and spilled
    const int lane_s = lrec_lane(lane);
    forward<C>(m, hot, s, lane_s, Mrow, warm, f, dbg PROF_PASS);
    integrate<C>(m, hot, s, lane_s, Mrow, f PROF_PASS);
    time += hot.timestep;
  }
  tk_next = draw();
  if (!last) {
    // hand the pipeline state to the next phase: write-through stores, drained, then the flag (one wave = one workgroup)
    for (int t = lane; t < C::NQ; t += 64) st_sc1(&rec[L.qpos + t], s.qpos[t]);
    if (lane < C::NV) { st_sc1(&rec[L.qvel + lane], s.qvel[lane]); st_sc1(&rec[L.warm + lane], warm); }
    if (lane < C::NU) st_sc1(&rec[L.ctrl + lane], s.ctrl[lane]);
    if (lane == 0) st_sc1(&rec[L.time], time);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && !(first && e == sc.withhold_env))
      __hip_atomic_store(sc.flags + e, (sc.launch_id << 8) | (unsigned)(phase + 1) | handoff_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if defined(RSR_PROFILE) || defined(RSR_TIMELINE)
    if (a.debug && lane == 0) prof_timeline(a.debug + (size_t)e * RSR_DEBUG_FLOATS + 7300 + 8 * phase, prof_rt0_, prof_ct0_);
#endif
    WSYNC();
    continue;
  }
  // ---- epilogue: reward, done, obs, info; derived data are from the last forward pass ----
  float done = 0.0f;
  float* obs_lds = s.scratch_b();                                     // staged so that auto-reset can override it
  if (lane == 0) {
#pragma clang fp contract(off)
    const float* W = m.env_reward;
    float reward, met[C::NMET];
    // the record words the wrapper code below reads back, fetched here in one batch: left at their uses they follow the metric
    // stores (same base pointer, run-time offsets: the compiler must keep the order) and each waits out a memory round trip
    float prev_done = 0.0f, em_old[2 + C::NMET];
#pragma unroll
    for (int i = 0; i < 2 + C::NMET; ++i) em_old[i] = 0.0f;
    if (wrap_episode) {
      prev_done = rec[L.episode_done];
#pragma unroll
      for (int i = 0; i < 2 + C::NMET; ++i) em_old[i] = rec[L.episode_metrics + i];
    }
    const float met_kept = rec[L.metrics + (ENV == ENV_TSHAPE ? 3 : 1)];      // the metric this env never writes
    if constexpr (ENV == ENV_TSHAPE) {
      const int site = m.env_ids[TID_SITE], tail = m.env_ids[TID_TAIL], ttail = m.env_ids[TID_TTAIL], tbody = m.env_ids[TID_T];
      float sp[3] = {s.spos[3 * site], s.spos[3 * site + 1], s.spos[3 * site + 2]};
      const float* gb = &s.egeom[0]; const float* gv = &s.egeom[3];
      float tb[3], tv[3];
      for (int i = 0; i < 3; ++i) { tb[i] = rec[L.target_base_pos + i]; tv[i] = rec[L.target_vertical_pos + i]; }
      float a0 = tb[0] - gb[0], a1 = tb[1] - gb[1], a2 = tb[2] - gb[2];
      float dis_base = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
      if (dis_base < 0.005f) dis_base = 0.0f;
      float push_base = 1.0f / (1.0f + 10.0f * dis_base);
      float b0 = tv[0] - gv[0], b1 = tv[1] - gv[1], b2 = tv[2] - gv[2];
      float dis_vert = sqrtf(b0 * b0 + b1 * b1 + b2 * b2);
      if (dis_vert < 0.005f) dis_vert = 0.0f;
      float push_vert = 1.0f / (1.0f + 10.0f * dis_vert);
      float ba[3] = {gv[0] - gb[0], gv[1] - gb[1], gv[2] - gb[2]}, ta[3] = {tv[0] - tb[0], tv[1] - tb[1], tv[2] - tb[2]};
      float dotp = ba[0] * ta[0] + ba[1] * ta[1] + ba[2] * ta[2];
      float nb = sqrtf(ba[0] * ba[0] + ba[1] * ba[1] + ba[2] * ba[2]), nt = sqrtf(ta[0] * ta[0] + ta[1] * ta[1] + ta[2] * ta[2]);
      float xita = acosf(clampf(dotp / (nb * nt), -1.0f, 1.0f));
      float push_w = 1.0f / (1.0f + 6.0f * xita);
      float push = (0.1515f * push_base + 0.1515f * push_vert + 0.66f * push_w) * W[0];
      float site_z = sp[2] < 0.83f ? 1.0f : 0.0f;
      float z_reward = 4.0f / (1.0f + 3.0f * fabsf(sp[2] - 0.805f));
      site_z = site_z + z_reward;
      float tx = s.spos[3 * tail], ty = s.spos[3 * tail + 1];
      float dx = s.spos[3 * ttail] - tx, dy = s.spos[3 * ttail + 1] - ty;
      float ang = atan2f(dy, dx + 0.00001f);
      float dist = sqrtf(dx * dx + dy * dy) + 0.025f;
      float y_ = dist * sinf(ang), x_ = dist * cosf(ang);
      float newT[2] = {dx - x_ + tx, dy - y_ + ty};
      float e0 = sp[0] - aux_old[0], e1 = sp[1] - aux_old[1];
      float s2c = sqrtf(e0 * e0 + e1 * e1);
      s2c = s2c < 0.02f ? 0.0f : s2c - 0.02f;
      float siet = (1.0f - tanhf(5.0f * s2c)) * W[1];
      float health = W[2] * fabsf((sp[2] < W[3] ? 1.0f : 0.0f) - 1.0f);
      reward = clampf(push + siet + health + site_z, -100.0f, 100.0f);
      done = s.xpos[3 * tbody + 2] < 0.6f ? 1.0f : 0.0f;
      tshape_obs<C>(m, s, tb, tv, xita, newT, obs_lds);
      met[0] = push; met[1] = siet; met[2] = health; met[3] = met_kept; met[4] = site_z;
      rec[L.metrics + 0] = push; rec[L.metrics + 1] = siet; rec[L.metrics + 2] = health; rec[L.metrics + 4] = site_z;
      for (int i = 0; i < 3; ++i) { rec[L.site_pos + i] = sp[i]; rec[L.T_pos + i] = s.xpos[3 * tbody + i]; }
      rec[L.new_T_pos] = newT[0]; rec[L.new_T_pos + 1] = newT[1];
      rec[L.xita] = xita;
    } else {
      const int cube = m.env_ids[ID_CUBE], site = m.env_ids[ID_SITE];
      float cp[3] = {s.xpos[3 * cube], s.xpos[3 * cube + 1], s.xpos[3 * cube + 2]};
      float sp[3] = {s.spos[3 * site], s.spos[3 * site + 1], s.spos[3 * site + 2]};
      float d0 = tp[0] - cp[0], d1 = tp[1] - cp[1], d2 = tp[2] - cp[2];
      const bool sf = m.env_kind == ENV_AIRBOT_SF;
      float btd = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
      if (btd < W[4]) btd = 0.0f;                       // 0.005 (cube_env.py:166) / 0.003 (test/airbot.py:191)
      float push = (1.0f / (1.0f + 3.0f * btd)) * W[0];
      float task_complete = btd < W[4] ? W[5] : 0.0f;   // test/airbot.py:196
      float site_z = sp[2] < 0.82f ? 1.0f : 0.0f;
      float dx = tp[0] - cp[0], dy = tp[1] - cp[1];
      float ang = atan2f(dy, dx + 0.00001f);
      float dist = sqrtf(dx * dx + dy * dy) + 0.04f;
      float y_ = dist * sinf(ang), x_ = dist * cosf(ang);
      float ncp[2] = {dx - x_ + cp[0], dy - y_ + cp[1]};
      float e0 = sp[0] - aux_old[0], e1 = sp[1] - aux_old[1];
      float s2c = sqrtf(e0 * e0 + e1 * e1);
      s2c = s2c < 0.042f ? 0.0f : s2c - 0.042f;
      float siet = (1.0f - tanhf(5.0f * s2c)) * W[1];
      if (btd < 0.005f) siet = W[1];
      float hd = sp[2] < W[3] ? 1.0f : 0.0f;
      if (sf && (sp[0] > 1.0f || sp[0] < -0.6f || sp[1] > 0.3f || sp[1] < -0.3f || cp[2] < 0.6f)) hd = 1.0f;   // test/airbot.py:227-233
      float health = W[2] * fabsf(hd - 1.0f);
      reward = clampf(sf ? push + siet + health + task_complete + site_z : push + siet + health + site_z, -100.0f, 100.0f);
      done = sf ? (btd < W[4] ? 1.0f : 0.0f) : (cp[2] < 0.6f ? 1.0f : 0.0f);
      cube_obs<C>(m, s, tp, ncp, obs_lds);
      met[0] = push; met[1] = met_kept; met[2] = siet;
      rec[L.metrics + 0] = push; rec[L.metrics + 2] = siet;
      for (int i = 0; i < 3; ++i) { rec[L.site_pos + i] = sp[i]; rec[L.cube_pos + i] = cp[i]; }
      rec[L.new_cube_pos] = ncp[0]; rec[L.new_cube_pos + 1] = ncp[1];
    }
    rec[L.reward] = reward;
    // EpisodeWrapper.step (action_repeat = 1)
    if (wrap_episode) {
      steps += 1.0f;
      bool over = steps >= (float)m.episode_length;
      rec[L.truncation] = over ? 1.0f - done : 0.0f;
      // brax: metric = (metric + x) * (1 - prev_done).  Written as a select: the same value for finite metrics, and an env whose
      // simulation went non-finite once (a blow-up) starts its next episode's sums clean instead of carrying NaN * 0 = NaN forever.
      float* em = rec + L.episode_metrics;
      em[0] = prev_done != 0.0f ? 0.0f : em_old[0] + reward;
      em[1] = prev_done != 0.0f ? 0.0f : em_old[1] + 1.0f;
#pragma unroll
      for (int i = 0; i < C::NMET; ++i) em[2 + i] = prev_done != 0.0f ? 0.0f : em_old[2 + i] + met[i];
      if (over) done = 1.0f;
      rec[L.episode_done] = done;
    }
    rec[L.steps] = steps;
    rec[L.done] = done;
    int* st = reinterpret_cast<int*>(rec + L.stats);
    st[0] = f.st.niter; st[1] = f.st.ls_total; st[2] = s.ncon; st[3] = handoff_err ? -1 : s.ncon_drop;
  }
  WSYNC();
  done = rdlane(done, 0);
  if (wrap_autoreset && done != 0.0f) {
    // AutoResetWrapper.step post-step: the cached first state replaces the pipeline state and obs
    for (int t = lane; t < L.persist_end; t += 64) rec[t] = rec[L.f_qpos + t];
    for (int t = lane; t < C::OBS; t += 64) rec[L.obs + t] = rec[L.f_obs + t];
  } else {
    store_pipeline<C>(s, rec, L, lane, warm, time);
    for (int t = lane; t < C::OBS; t += 64) rec[L.obs + t] = obs_lds[t];
  }
#ifdef RSR_PROFILE
  PROF(PS_EPILOGUE)
  if (a.debug && lane == 0) {      // stage cycle counters leave the kernel only through the debug buffer
    float* d = a.debug + (size_t)e * RSR_DEBUG_FLOATS + 7200;
    for (int i = 0; i < PS_COUNT; ++i) d[i] = (float)prof_.acc[i];
  }
#endif
#if defined(RSR_PROFILE) || defined(RSR_TIMELINE)
  if (a.debug && lane == 0) prof_timeline(a.debug + (size_t)e * RSR_DEBUG_FLOATS + 7300 + 8 * phase, prof_rt0_, prof_ct0_);
#endif
  WSYNC();                                                     // the next unit reuses this wave's LDS image
  }
}

// ---------------------------------------------------------------- Go2 reset kernel (joystick.py:123-203 + wrappers)
template <class C>
__global__ __launch_bounds__(64) void go2_reset_kernel(const DModel* __restrict__ mp, Layout L, StepArgs a) {
  const DModel& m = *mp;
  const Hot hot = make_hot(m);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem<C>& s = *reinterpret_cast<Smem<C>*>(smem_raw);
  const int e = blockIdx.x, lane = threadIdx.x;
  if (e >= a.n) return;
  float* rec = a.state + (size_t)e * L.rec;
  const float* F = m.env_go2f;
  uint32_t* bits = reinterpret_cast<uint32_t*>(s.scratch_b());
  float* obs_lds = s.scratch_b() + 64;
  load_overrides<C>(m, s, a, e, lane);
  for (int t = lane; t < C::NINFO; t += 64) s.ginfo[t] = 0.0f;
  uint32_t rng0 = a.keys[2 * e], rng1 = a.keys[2 * e + 1], ks[4][2];
  if (lane < C::NQ) s.qpos[lane] = m.env_go2_home[lane];
  if (lane < C::NV) s.qvel[lane] = 0.0f;
  {
    uint32_t k2[2][2];
    tf_split<2>(rng0, rng1, bits, lane, k2); rng0 = k2[0][0]; rng1 = k2[0][1];
    float dxy = tf_uniform(k2[1][0], k2[1][1], 2, -0.5f, 0.5f, bits, lane);
    if (lane < 2) s.qpos[lane] = m.env_go2_home[lane] + dxy;
    tf_split<2>(rng0, rng1, bits, lane, k2); rng0 = k2[0][0]; rng1 = k2[0][1];
    float yaw = rdlane(tf_uniform(k2[1][0], k2[1][1], 1, -3.14f, 3.14f, bits, lane), 0);
    if (lane == 0) {
#pragma clang fp contract(off)
      float sn = sinf(yaw * 0.5f), cs = cosf(yaw * 0.5f);
      Q4 q = Q4{m.env_go2_home[3], m.env_go2_home[4], m.env_go2_home[5], m.env_go2_home[6]}, r = Q4{cs, 0.0f * sn, 0.0f * sn, 1.0f * sn};
      Q4 o;
      o.w = q.w * r.w - q.x * r.x - q.y * r.y - q.z * r.z;
      o.x = q.w * r.x + q.x * r.w + q.y * r.z - q.z * r.y;
      o.y = q.w * r.y - q.x * r.z + q.y * r.w + q.z * r.x;
      o.z = q.w * r.z + q.x * r.y - q.y * r.x + q.z * r.w;
      st4(&s.qpos[3], o);
    }
    tf_split<2>(rng0, rng1, bits, lane, k2); rng0 = k2[0][0]; rng1 = k2[0][1];
    float v6 = tf_uniform(k2[1][0], k2[1][1], 6, -0.5f, 0.5f, bits, lane);
    if (lane < 6) s.qvel[lane] = v6;
  }
  WSYNC();
  if (lane < C::NU) s.ctrl[lane] = s.qpos[7 + lane];          // mjx_env.init(..., ctrl = qpos[7:])
  if (lane == 0) { s.xfrc_body = 0; s.acc_body = m.site_bodyid[m.env_ids[0]]; s.xfrc[0] = s.xfrc[1] = s.xfrc[2] = 0.0f; }
  WSYNC();
  float Mrow[C::NV], warm = 0.0f;
  FwdOut<C> f;
  PROF_DECL
  forward<C>(m, hot, s, lane, Mrow, warm, f, a.debug ? a.debug + (size_t)e * RSR_DEBUG_FLOATS : nullptr PROF_PASS);
  WSYNC();
  tf_split<4>(rng0, rng1, bits, lane, ks); rng0 = ks[0][0]; rng1 = ks[0][1];
  {
    float t_pert = rdlane(tf_uniform(ks[1][0], ks[1][1], 1, F[17], F[18], bits, lane), 0);
    float dur = rdlane(tf_uniform(ks[2][0], ks[2][1], 1, F[19], F[20], bits, lane), 0);
    float mag = rdlane(tf_uniform(ks[3][0], ks[3][1], 1, F[21], F[22], bits, lane), 0);
    if (lane == 0) {
      s.ginfo[G2_STEPS_PERT] = rintf(t_pert / F[0]); s.ginfo[G2_PERT_DUR_S] = dur;
      s.ginfo[G2_PERT_DUR] = rintf(dur / F[0]); s.ginfo[G2_PERT_MAG] = mag;
    }
  }
  {
    uint32_t k3[3][2];
    tf_split<3>(rng0, rng1, bits, lane, k3); rng0 = k3[0][0]; rng1 = k3[0][1];
    float uu = rdlane(tf_uniform(k3[1][0], k3[1][1], 1, 0.0f, 1.0f, bits, lane), 0);
    float amp = lane < 3 ? F[10 + lane] : 0.0f;
    float cmd = tf_uniform(k3[2][0], k3[2][1], 3, -amp, amp, bits, lane);
    if (lane < 3) s.ginfo[G2_CMD + lane] = cmd;
    if (lane == 0) {
#pragma clang fp contract(off)
      float t_cmd = -log1pf(-uu) * F[16];
      s.ginfo[G2_STEPS_CMD] = rintf(t_cmd / F[0]);
      s.ginfo[G2_RNG] = __uint_as_float(rng0); s.ginfo[G2_RNG + 1] = __uint_as_float(rng1);
    }
  }
  WSYNC();
  G2Sens sn;
  go2_sensors<C>(m, s, sn);
  go2_accelerometer<C>(m, s, lane, f.qacc, sn);
  go2_obs<C>(m, s, sn, obs_lds, bits, lane, m.env_go2_home[7 + (lane < 12 ? lane : 0)]);
  for (int t = lane; t < GO2_PRIV; t += 64) { float v = go2_priv_elem<C>(m, s, sn, obs_lds, t); rec[L.priv_obs + t] = v; rec[L.f_priv_obs + t] = v; }
  store_pipeline<C>(s, rec, L, lane, warm, 0.0f);
  for (int t = lane; t < C::NINFO; t += 64) rec[L.go2_info + t] = s.ginfo[t];
  for (int t = lane; t < C::OBS; t += 64) { rec[L.obs + t] = obs_lds[t]; rec[L.f_obs + t] = obs_lds[t]; }
  if (lane == 0) {
    rec[L.reward] = 0.0f; rec[L.done] = 0.0f;
    for (int i = 0; i < C::NMET; ++i) rec[L.metrics + i] = 0.0f;
    rec[L.steps] = 0.0f; rec[L.truncation] = 0.0f; rec[L.episode_done] = 0.0f;
    for (int i = 0; i < 2 + C::NMET; ++i) rec[L.episode_metrics + i] = 0.0f;
    int* st = reinterpret_cast<int*>(rec + L.stats);
    st[0] = f.st.niter; st[1] = f.st.ls_total; st[2] = s.ncon; st[3] = s.ncon_drop;
    rec[L.f_time] = 0.0f;
  }
  for (int t = lane; t < C::NQ; t += 64) rec[L.f_qpos + t] = s.qpos[t];
  if (lane < C::NV) { rec[L.f_qvel + lane] = s.qvel[lane]; rec[L.f_warm + lane] = warm; }
  if (lane < C::NU) rec[L.f_ctrl + lane] = s.ctrl[lane];
  for (int t = lane; t < C::NB * 3; t += 64) rec[L.f_xpos + t] = s.xpos[t];
  for (int t = lane; t < C::NS * 3; t += 64) rec[L.f_site_xpos + t] = s.spos[t];
}

// ---------------------------------------------------------------- Go2 step kernel (joystick.py:204-280 + wrappers)
template <class C>
#ifndef RSR_GO2_WAVES_PER_EU
#define RSR_GO2_WAVES_PER_EU 4
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RSR_GO2_WAVES_PER_EU, RSR_GO2_WAVES_PER_EU)))
void go2_step_kernel(const DModel* __restrict__ mp, Layout L, StepArgs a) {
  const DModel& m = *mp;
  const Hot hot = make_hot(m);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem<C>& s = *reinterpret_cast<Smem<C>*>(smem_raw);
  const int e = blockIdx.x, lane = threadIdx.x;
  if (e >= a.n) return;
  float* rec = a.state + (size_t)e * L.rec;
  const bool wrap_episode = m.wrap_flags & 1, wrap_autoreset = (m.wrap_flags & 2) != 0;
  const float* F = m.env_go2f;
  const float dt = F[0];
  uint32_t* bits = reinterpret_cast<uint32_t*>(s.scratch_b());
  float* obs_lds = s.scratch_b() + 64;
  float* rwl = s.scratch_b() + 128;                                   // scaled reward terms staged for the metrics write
  PROF_DECL
  for (int t = lane; t < C::NQ; t += 64) s.qpos[t] = rec[L.qpos + t];
  float warm = 0.0f;
  if (lane < C::NV) { s.qvel[lane] = rec[L.qvel + lane]; warm = rec[L.warm + lane]; }
  float time = rec[L.time];
  load_overrides<C>(m, s, a, e, lane);
  for (int t = lane; t < C::NINFO; t += 64) s.ginfo[t] = rec[L.go2_info + t];
  const float done_prev = rec[L.done];
  float steps = rec[L.steps];
  if (wrap_autoreset && done_prev != 0.0f) steps = 0.0f;
  const float act_in = lane < C::NU ? a.action[(size_t)e * C::NU + lane] : 0.0f;
  WSYNC();
  // ---- perturbation kick (:594-644): half-sine force pulse on the torso, or wait and draw the next direction ----
  if (m.env_go2i[2]) {
    const bool kicking = s.ginfo[G2_SINCE_PERT] >= s.ginfo[G2_STEPS_PERT];      // wave-uniform (LDS)
    if (kicking) {
      WSYNC();
      if (lane == 0) {
#pragma clang fp contract(off)
        float t = s.ginfo[G2_PERT_STEPS] * dt;
        float ph = 3.14159265358979323846f * t;
        float u_t = 0.5f * sinf(ph / s.ginfo[G2_PERT_DUR_S]);
        float f1 = u_t * F[23]; float f2 = f1 * s.ginfo[G2_PERT_MAG];
        float force = f2 / s.ginfo[G2_PERT_DUR_S];
        for (int c = 0; c < 3; ++c) s.ginfo[G2_XFRC + c] = force * s.ginfo[G2_PERT_DIR + c];
        if (s.ginfo[G2_PERT_STEPS] >= s.ginfo[G2_PERT_DUR]) s.ginfo[G2_SINCE_PERT] = 0.0f;
        s.ginfo[G2_PERT_STEPS] += 1.0f;
      }
    } else {
      uint32_t k2[2][2];
      tf_split<2>(__float_as_uint(s.ginfo[G2_RNG]), __float_as_uint(s.ginfo[G2_RNG + 1]), bits, lane, k2);
      float angle = rdlane(tf_uniform(k2[1][0], k2[1][1], 1, 0.0f, 6.2831855f, bits, lane), 0);
      WSYNC();
      if (lane == 0) {
        s.ginfo[G2_RNG] = __uint_as_float(k2[0][0]); s.ginfo[G2_RNG + 1] = __uint_as_float(k2[0][1]);
        float since = s.ginfo[G2_SINCE_PERT] + 1.0f;
        s.ginfo[G2_SINCE_PERT] = since;
        s.ginfo[G2_XFRC] = 0.0f; s.ginfo[G2_XFRC + 1] = 0.0f; s.ginfo[G2_XFRC + 2] = 0.0f;
        if (since >= s.ginfo[G2_STEPS_PERT]) {
          s.ginfo[G2_PERT_STEPS] = 0.0f;
          s.ginfo[G2_PERT_DIR] = cosf(angle); s.ginfo[G2_PERT_DIR + 1] = sinf(angle); s.ginfo[G2_PERT_DIR + 2] = 0.0f;
        }
      }
    }
    WSYNC();
  }
  if (lane == 0) {
    s.acc_body = m.site_bodyid[m.env_ids[0]];
    s.xfrc_body = m.env_go2i[2] ? m.env_ids[10] : 0;
    s.xfrc[0] = s.ginfo[G2_XFRC]; s.xfrc[1] = s.ginfo[G2_XFRC + 1]; s.xfrc[2] = s.ginfo[G2_XFRC + 2];
  }
  WSYNC();
  // ---- action delay FIFO (:207-215) and motor targets (:216) ----
  const int adel = m.env_go2i[0];
  float actual = act_in;
  if (adel > 0) {
    float shifted = 0.0f;
    const int nbuf = adel * C::NU;
    if (lane < C::NU) actual = s.ginfo[G2_ACT_BUF + lane];
    if (lane < nbuf) shifted = s.ginfo[G2_ACT_BUF + C::NU + lane];
    WSYNC();
    if (lane < nbuf) s.ginfo[G2_ACT_BUF + lane] = shifted;
    if (lane < C::NU) s.ginfo[G2_ACT_BUF + nbuf + lane] = act_in;
  }
  if (lane < C::NU) {
#pragma clang fp contract(off)
    float sc = actual * F[1];
    s.ctrl[lane] = m.env_go2_home[7 + lane] + sc;
  }
  WSYNC();
  PROF(PS_LOAD)
  float Mrow[C::NV];
  FwdOut<C> f;
  const int prio_q = prio_quarter(a, e);
  for (int fr = 0; fr < m.n_frames; ++fr) {
#if defined(RSR_PROFILE) || defined(RSR_TIMELINE)
    float* dbg = nullptr;
#else
    float* dbg = (a.debug && fr == m.n_frames - 1) ? a.debug + (size_t)e * RSR_DEBUG_FLOATS : nullptr;
#endif
    const int lane_s = lrec_lane(lane);        // see step_kernel
    prio_substep(a.prio_mode, prio_q, fr);
    forward<C>(m, hot, s, lane_s, Mrow, warm, f, dbg PROF_PASS);
    integrate<C>(m, hot, s, lane_s, Mrow, f PROF_PASS);
    time += hot.timestep;
  }
  // per-joint constants of the epilogue (home pose, soft limits): vector loads take a couple of thousand cycles under load, so
  // they are issued here and consumed after the sensor / FIFO / contact code
  const int jl = lrec_lane(lane < 12 ? lane : 0);      // opaque: not merged with the prologue's load of the same address, which
                                                        // would keep the value live (or spilled) across the whole substep loop
  const float home_l = m.env_go2_home[7 + jl], soft_lo = m.env_go2_soft[jl], soft_hi = m.env_go2_soft[12 + jl];
  // (the Episode wrapper's running sums and the feet of the contact pairs likewise)
  int foot_of_pair = -1;                    // lane p < NP: the foot whose geom pair p holds (pairs are static), or -1
  if (lane < C::NP) {
    const int g1 = m.pair_geom1[lane], g2 = m.pair_geom2[lane], fl = m.env_ids[5];
#pragma unroll
    for (int fi = 0; fi < 4; ++fi) { const int gf = m.env_ids[6 + fi]; if ((g2 == gf && g1 == fl) || (g1 == gf && g2 == fl)) foot_of_pair = fi; }
  }
  const float prev_done = wrap_episode ? rec[L.episode_done] : 0.0f;
  const float em_old = (wrap_episode && lane < C::NMET + 2) ? rec[L.episode_metrics + lane] : 0.0f;
  // ---- sensors of the last forward pass, IMU FIFOs (:220-235) ----
  G2Sens sn;
  go2_sensors<C>(m, s, sn);
  go2_accelerometer<C>(m, s, lane, f.qacc, sn);
  PROF(PS_E_SENS)
  const int idel = m.env_go2i[1];
  if (idel > 0) {
    float v = 0.0f;
    const int nb = idel * 3;
    int base = lane < nb ? G2_GYRO_BUF : (lane < 2 * nb ? G2_LINVEL_BUF : G2_GRAV_BUF);
    int off = lane < nb ? lane : (lane < 2 * nb ? lane - nb : lane - 2 * nb);
    if (lane < 3 * nb) v = s.ginfo[base + 3 + off];
    WSYNC();
    if (lane < 3 * nb) s.ginfo[base + off] = v;
    if (lane < 3) { s.ginfo[G2_GYRO_BUF + nb + lane] = pick3(sn.gyro, lane); s.ginfo[G2_LINVEL_BUF + nb + lane] = pick3(sn.linvel, lane); s.ginfo[G2_GRAV_BUF + nb + lane] = pick3(sn.gravity, lane); }
    WSYNC();
  }
  PROF(PS_E_FIFO)
  // ---- foot contacts (:236-245) ----
  // lane p < NP knows which foot pair p belongs to (the pairs are static); lane i < ncon looks its contact's pair up there
  int contact[4];
  int foot_site[4];
#pragma unroll
  for (int fi = 0; fi < 4; ++fi) foot_site[fi] = m.env_ids[1 + fi];
  {
    const int nc = s.ncon, ci = lane < nc ? lane : 0;
    const int cp = s.cpair[ci]; const float cd = s.cdist[ci];
    const int fo = __shfl(foot_of_pair, cp & 63);
    const int my_foot = (lane < nc && cd < 0.0f) ? fo : -1;
#pragma unroll
    for (int fi = 0; fi < 4; ++fi) contact[fi] = __ballot(my_foot == fi) != 0ull ? 1 : 0;
  }
  int first_contact[4]; float feet_z[4];
  for (int fi = 0; fi < 4; ++fi) {
    bool filt = contact[fi] || s.ginfo[G2_LAST_CONTACT + fi] != 0.0f;
    first_contact[fi] = (s.ginfo[G2_AIR + fi] > 0.0f) && filt;
    feet_z[fi] = s.spos[3 * foot_site[fi] + 2];
  }
  WSYNC();
  if (lane < 4) {
    s.ginfo[G2_AIR + lane] += dt;
    s.ginfo[G2_SWING + lane] = fmaxf(s.ginfo[G2_SWING + lane], s.spos[3 * (lane == 0 ? foot_site[0] : lane == 1 ? foot_site[1] : lane == 2 ? foot_site[2] : foot_site[3]) + 2]);
  }
  WSYNC();
  PROF(PS_E_FEET)
  go2_obs<C>(m, s, sn, obs_lds, bits, lane, home_l);
  PROF(PS_E_OBS)
  float priv[2];                                              // this lane's elements of privileged_state (info as of now)
  {
    const float kick_flag = s.ginfo[G2_SINCE_PERT] >= s.ginfo[G2_STEPS_PERT] ? 1.0f : 0.0f;
    priv[0] = go2_priv_gather<C, true>(m, s, sn, obs_lds, lane, foot_site, kick_flag, home_l);
    priv[1] = go2_priv_gather<C, false>(m, s, sn, obs_lds, lane + 64 < GO2_PRIV ? lane + 64 : GO2_PRIV - 1, foot_site, kick_flag, home_l);
  }
  float done = sn.up[2] < 0.0f ? 1.0f : 0.0f;
  PROF(PS_E_PRIV)
  // ---- rewards (:367-593): the per-joint pieces of the seven 12-term sums are computed by lanes 0..11 and staged; lane 0 adds
  // them up in the reference's order and evaluates the rest of the scalar algebra op by op ----
  float* rstage = s.scratch_b() + 160;                        // [7][12]
  if (lane < 12) {
#pragma clang fp contract(off)
    const float q = s.qpos[7 + lane], dq = q - home_l;
    const float lo_ = q - soft_lo, hi_ = q - soft_hi;
    const float w = (lane % 3 == 2) ? 0.1f : 1.0f;
    const float t = s.aforce[lane];
    const float dd = act_in - s.ginfo[G2_LAST_ACT + lane];
    rstage[lane] = fabsf(dq);
    rstage[12 + lane] = -(lo_ < 0.0f ? lo_ : 0.0f) + (hi_ > 0.0f ? hi_ : 0.0f);
    rstage[24 + lane] = dq * dq * w;
    rstage[36 + lane] = t * t;
    rstage[48 + lane] = fabsf(t);
    rstage[60 + lane] = fabsf(s.qvel[6 + lane]) * fabsf(t);
    rstage[72 + lane] = dd * dd;
  }
  WSYNC();
  float reward = 0.0f;
  if (lane == 0) {
#pragma clang fp contract(off)
    const float* SC = m.env_go2_scales;
    const float* cmd = &s.ginfo[G2_CMD];
    float cmd_norm = sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1] + cmd[2] * cmd[2]);
    float moving = cmd_norm > 0.01f ? 1.0f : 0.0f, still = cmd_norm < 0.01f ? 1.0f : 0.0f;
    float rw[RW_COUNT];
    {
      float e0 = cmd[0] - sn.linvel[0], e1 = cmd[1] - sn.linvel[1];
      rw[RW_TRACK_LIN] = expf(-(e0 * e0 + e1 * e1) / F[8]);
      float ea = cmd[2] - sn.gyro[2];
      rw[RW_TRACK_ANG] = expf(-(ea * ea) / F[8]);
    }
    rw[RW_LIN_VEL_Z] = sn.glin[2] * sn.glin[2];
    rw[RW_ANG_VEL_XY] = sn.gang[0] * sn.gang[0] + sn.gang[1] * sn.gang[1];
    rw[RW_ORIENT] = sn.up[0] * sn.up[0] + sn.up[1] * sn.up[1];
    {
      float sa = 0, lim = 0, pose = 0, s2 = 0, s1 = 0, en = 0, ar = 0;
#pragma unroll 1
      for (int i0 = 0; i0 < 12; i0 += 4) {
#pragma unroll
        for (int i = i0; i < i0 + 4; ++i) {
          sa += rstage[i]; lim += rstage[12 + i]; pose += rstage[24 + i];
          s2 += rstage[36 + i]; s1 += rstage[48 + i]; en += rstage[60 + i]; ar += rstage[72 + i];
        }
      }
      rw[RW_STAND_STILL] = sa * still; rw[RW_DOF_LIMITS] = lim; rw[RW_POSE] = expf(-pose);
      rw[RW_TERM] = done;
      rw[RW_TORQUES] = sqrtf(s2) + s1; rw[RW_ENERGY] = en; rw[RW_ACTION_RATE] = ar;
    }
    {
      float slip = 0, clear = 0, height = 0, air = 0; int nair = 0;
      for (int fi = 0; fi < 4; ++fi) {
        const int sid = foot_site[fi];
        float vx = s.slinvel[3 * sid], vy = s.slinvel[3 * sid + 1];
        float v2 = vx * vx + vy * vy;
        slip += v2 * (float)contact[fi];
        clear += fabsf(feet_z[fi] - F[9]) * sqrtf(sqrtf(v2));
        float err = s.ginfo[G2_SWING + fi] / F[9] - 1.0f;
        height += err * err * (float)first_contact[fi];
        air += (s.ginfo[G2_AIR + fi] - 0.1f) * (float)first_contact[fi];
        nair += !contact[fi];
      }
      rw[RW_FEET_SLIP] = slip * moving; rw[RW_FEET_CLEAR] = clear; rw[RW_FEET_HEIGHT] = height * moving; rw[RW_FEET_AIR] = air * moving;
      rw[RW_ALL_FEET_AIR] = (nair >= 3 ? 1.0f : 0.0f) * moving;
      rw[RW_FEET_OFF_STILL] = (float)nair * still;
      float p1 = 0, p2 = 0;
      for (int i = 0; i < 3; ++i) { float x = s.qpos[7 + 3 + i] - s.qpos[7 + 6 + i], y = s.qpos[7 + i] - s.qpos[7 + 9 + i]; p1 += x * x; p2 += y * y; }
      rw[RW_SYM_GAIT] = (p1 + p2) * moving;
      const float* at = &s.ginfo[G2_AIR]; const float* ct = &s.ginfo[G2_CONTACT_T];
      float la = (at[1] + at[3]) / 2.0f, lc = (ct[1] + ct[3]) / 2.0f, ra = (at[0] + at[2]) / 2.0f, rc = (ct[0] + ct[2]) / 2.0f;
      rw[RW_LR_SYM] = ((la - ra) * (la - ra) + (lc - rc) * (lc - rc)) * moving;
      float fa = (at[0] + at[1]) / 2.0f, fc = (ct[0] + ct[1]) / 2.0f, ba = (at[2] + at[3]) / 2.0f, bc = (ct[2] + ct[3]) / 2.0f;
      rw[RW_FB_SYM] = ((fa - ba) * (fa - ba) + (fc - bc) * (fc - bc)) * moving;
    }
    for (int k = 0; k < RW_COUNT; ++k) { rw[k] = rw[k] * SC[k]; rwl[k] = rw[k]; }
    const int order[RW_COUNT] = {RW_TRACK_LIN, RW_TRACK_ANG, RW_LIN_VEL_Z, RW_ANG_VEL_XY, RW_ORIENT, RW_STAND_STILL, RW_TERM, RW_POSE,
                                 RW_TORQUES, RW_ACTION_RATE, RW_ENERGY, RW_FEET_SLIP, RW_FEET_CLEAR, RW_FEET_HEIGHT, RW_FEET_AIR,
                                 RW_DOF_LIMITS, RW_ALL_FEET_AIR, RW_SYM_GAIT, RW_LR_SYM, RW_FB_SYM, RW_FEET_OFF_STILL};
    float total = 0.0f;
    for (int k = 0; k < RW_COUNT; ++k) total = total + rwl[order[k]];
    reward = clampf(total * dt, 0.0f, 10000.0f);
  }
  reward = rdlane(reward, 0);
  WSYNC();
  PROF(PS_E_REWARD)
  // ---- bookkeeping (:255-277): last actions, command resampling (threefry), timers ----
  if (lane < C::NU) { s.ginfo[G2_LAST_LAST_ACT + lane] = s.ginfo[G2_LAST_ACT + lane]; s.ginfo[G2_LAST_ACT + lane] = act_in; }
  float steps_cmd = s.ginfo[G2_STEPS_CMD] - 1.0f;
  {
    uint32_t rng0 = __float_as_uint(s.ginfo[G2_RNG]), rng1 = __float_as_uint(s.ginfo[G2_RNG + 1]);
    uint32_t k3[3][2], k4[4][2];
    rng0 = (uint32_t)uniform_i((int)rng0); rng1 = (uint32_t)uniform_i((int)rng1);
    tf_split_reg<3>(rng0, rng1, lane, k3);
    tf_split_reg<4>(k3[1][0], k3[1][1], lane, k4);          // sample_command: rng, y_rng, w_rng, z_rng
    float amp = lane < 3 ? F[10 + lane] : 0.0f;
    {
      // the four draws in one evaluation: lanes [0,2) y, [2,4) z, [4,6) w (3 elements each), lane 6 the resampling time (1)
      const int d = lane < 6 ? (lane >> 1) : 3;
      uint32_t k0 = k3[2][0], k1 = k3[2][1];
      if (d == 0) { k0 = k4[1][0]; k1 = k4[1][1]; }
      if (d == 1) { k0 = k4[3][0]; k1 = k4[3][1]; }
      if (d == 2) { k0 = k4[2][0]; k1 = k4[2][1]; }
      WSYNC();
      tf_bits_batched(k0, k1, d < 3 ? 3 : 1, lane < 6 ? (lane & 1) : 0, 3 * d, lane < 7, bits);
      WSYNC();
    }
    const float y = lane < 3 ? uniform_from_bits(bits[lane], -amp, amp) : 0.0f;
    const float uz = lane < 3 ? uniform_from_bits(bits[3 + lane], 0.0f, 1.0f) : 0.0f;
    const float uw = lane < 3 ? uniform_from_bits(bits[6 + lane], 0.0f, 1.0f) : 0.0f;
    const float uu = uniform_from_bits(bits[9], 0.0f, 1.0f);
    WSYNC();
    if (lane < 3 && steps_cmd <= 0.0f) {
#pragma clang fp contract(off)
      float z = uz < F[13 + lane] ? 1.0f : 0.0f, w = uw < 0.5f ? 1.0f : 0.0f;
      float x = s.ginfo[G2_CMD + lane];
      float yz = y * z; float dif = x - yz; float wd = w * dif;
      s.ginfo[G2_CMD + lane] = x - wd;
    }
    if (lane == 0) {
#pragma clang fp contract(off)
      if (done != 0.0f || steps_cmd <= 0.0f) { float t1 = -log1pf(-uu) * F[16]; steps_cmd = rintf(t1 / dt); }
      s.ginfo[G2_STEPS_CMD] = steps_cmd;
      s.ginfo[G2_RNG] = __uint_as_float(k3[0][0]); s.ginfo[G2_RNG + 1] = __uint_as_float(k3[0][1]);
    }
  }
  if (lane < 4) {
#pragma clang fp contract(off)
    float c = (float)(lane == 0 ? contact[0] : (lane == 1 ? contact[1] : (lane == 2 ? contact[2] : contact[3]))), nc = 1.0f - c;   // (no lane-indexed array: scratch)
    s.ginfo[G2_AIR + lane] = (s.ginfo[G2_AIR + lane] + dt) * nc;
    s.ginfo[G2_CONTACT_T + lane] = (s.ginfo[G2_CONTACT_T + lane] + dt) * c;
    s.ginfo[G2_LAST_CONTACT + lane] = c;
    s.ginfo[G2_SWING + lane] *= nc;
  }
  WSYNC();
  if (lane == 0) {
#pragma clang fp contract(off)
    float swing_mean = (((s.ginfo[G2_SWING] + s.ginfo[G2_SWING + 1]) + s.ginfo[G2_SWING + 2]) + s.ginfo[G2_SWING + 3]) / 4.0f;
    rwl[RW_COUNT] = swing_mean;
  }
  WSYNC();
  {
    // metrics and the Episode wrapper's sums: one lane per entry (rwl[0..NMET) = the scaled terms + swing_peak)
    bool over = false;
    float trunc = 0.0f;
    if (wrap_episode) {
      steps += 1.0f;
      over = steps >= (float)m.episode_length;
      trunc = over ? 1.0f - done : 0.0f;
    }
    if (lane < C::NMET) rec[L.metrics + lane] = rwl[lane];
    if (wrap_episode && lane < C::NMET + 2) {
      float* em = rec + L.episode_metrics;
      const float add = lane == 0 ? reward : (lane == 1 ? 1.0f : rwl[lane >= 2 ? lane - 2 : 0]);
      em[lane] = prev_done != 0.0f ? 0.0f : em_old + add;
    }
    if (over) done = 1.0f;
    if (lane == 0) {
      rec[L.reward] = reward;
      if (wrap_episode) { rec[L.truncation] = trunc; rec[L.episode_done] = done; }
      rec[L.steps] = steps;
      rec[L.done] = done;
      int* st = reinterpret_cast<int*>(rec + L.stats);
      st[0] = f.st.niter; st[1] = f.st.ls_total; st[2] = s.ncon; st[3] = s.ncon_drop;
    }
  }
  WSYNC();
  PROF(PS_E_BOOK)
  if (wrap_autoreset && done != 0.0f && lane < 3) s.ginfo[G2_XFRC + lane] = 0.0f;     // xfrc_applied belongs to `data`: back to the first state's zeros
  WSYNC();
  for (int t = lane; t < C::NINFO; t += 64) rec[L.go2_info + t] = s.ginfo[t];       // info is never reset by AutoReset
  if (wrap_autoreset && done != 0.0f) {
    for (int t = lane; t < L.persist_end; t += 64) rec[t] = rec[L.f_qpos + t];
    for (int t = lane; t < C::OBS; t += 64) rec[L.obs + t] = rec[L.f_obs + t];
    for (int t = lane; t < GO2_PRIV; t += 64) rec[L.priv_obs + t] = rec[L.f_priv_obs + t];
  } else {
    store_pipeline<C>(s, rec, L, lane, warm, time);
    for (int t = lane; t < C::OBS; t += 64) rec[L.obs + t] = obs_lds[t];
    rec[L.priv_obs + lane] = priv[0];
    if (lane + 64 < GO2_PRIV) rec[L.priv_obs + lane + 64] = priv[1];
  }
#ifdef RSR_PROFILE
  PROF(PS_EPILOGUE)
  if (a.debug && lane == 0) {
    float* d = a.debug + (size_t)e * RSR_DEBUG_FLOATS + 7200;
    for (int i = 0; i < PS_COUNT; ++i) d[i] = (float)prof_.acc[i];
  }
#endif
#if defined(RSR_PROFILE) || defined(RSR_TIMELINE)
  if (a.debug && lane == 0) prof_timeline(a.debug + (size_t)e * RSR_DEBUG_FLOATS + 7300, prof_rt0_, prof_ct0_);
#endif
}


// ================================================================ Go2 Handstand / Footstand (go2/handstand.py)
// env_ids: 0 imu site, 1 floor geom, 2..13 the twelve unwanted-contact geoms, 14..15 the feet geoms of the contact cost, 16 trunk body.
// env_go2f: ctrl_dt, action_scale, noise level, scales joint_pos / joint_vel / gyro / gravity / linvel, init_from_crouch,
// energy_termination_threshold, z_des, desired forward vector.  env_go2i: joint ids of the pose cost.  env_go2_home: home | pre_recovery
// qpos.  env_go2_soft: soft lower | upper limits.  info block: step at 0, last_act at 4..15, rng at G2_RNG.
enum { HS_STEP = 0, HS_LAST_ACT = 4, HS_PRIV = 94 };
enum { HM_HEIGHT = 0, HM_ORIENT, HM_CONTACT, HM_ACTION_RATE, HM_TERM, HM_DOF_LIMITS, HM_TORQUES, HM_POSE, HM_STAY_STILL, HM_ENERGY, HM_DOF_ACC, HM_COUNT };

// handstand.py:196-245: the 45-dim "state" into obs_lds; five splits of info.rng, draws in the reference's order (gyro, gravity, joint
// angles, joint velocities, linvel)
template <class C>
__device__ __forceinline__ void hs_obs(const DModel& m, Smem<C>& s, const G2Sens& sn, float* obs_lds, uint32_t* bits, int lane) {
#pragma clang fp contract(off)
  const float* F = m.env_go2f;
  uint32_t rng0 = __float_as_uint(s.ginfo[G2_RNG]), rng1 = __float_as_uint(s.ginfo[G2_RNG + 1]);
  rng0 = (uint32_t)uniform_i((int)rng0); rng1 = (uint32_t)uniform_i((int)rng1);
  const float level = F[2];
  const float home_l = m.env_go2_home[7 + (lane < 12 ? lane : 0)];
#pragma unroll
  for (int d = 0; d < 5; ++d) {
    uint32_t k2[2][2];
    tf_split<2>(rng0, rng1, bits, lane, k2); rng0 = k2[0][0]; rng1 = k2[0][1];
    const int n = (d == 2 || d == 3) ? 12 : 3;
    const float u = tf_uniform(k2[1][0], k2[1][1], n, 0.0f, 1.0f, bits, lane);
    if (lane < n) {
      float src, scale; int dst;
      if (d == 0) { src = pick3(sn.gyro, lane); scale = F[5]; dst = 3 + lane; }
      else if (d == 1) { src = pick3(sn.gravity, lane); scale = F[6]; dst = 6 + lane; }
      else if (d == 2) { src = s.qpos[7 + lane]; scale = F[3]; dst = 9 + lane; }
      else if (d == 3) { src = s.qvel[6 + lane]; scale = F[4]; dst = 21 + lane; }
      else { src = pick3(sn.linvel, lane); scale = F[7]; dst = lane; }
      float a = 2.0f * u; float b = a - 1.0f; float c = b * level; float e = c * scale;
      float val = src + e;
      if (d == 2) val = val - home_l;
      obs_lds[dst] = val;
    }
  }
  if (lane < 12) obs_lds[33 + lane] = s.ginfo[HS_LAST_ACT + lane];
  if (lane == 0) { s.ginfo[G2_RNG] = __uint_as_float(rng0); s.ginfo[G2_RNG + 1] = __uint_as_float(rng1); }
  WSYNC();
}

// handstand.py:246-259: element t of obs["privileged_state"] (94)
template <class C>
__device__ __forceinline__ float hs_priv_elem(const DModel& m, const Smem<C>& s, const G2Sens& sn, const float* obs_lds, int t) {
  if (t < 45) return obs_lds[t];
  t -= 45;
  if (t < 3) return pick3(sn.gyro, t);
  if (t < 6) return pick3(sn.accel, t - 3);
  if (t < 9) return pick3(sn.linvel, t - 6);
  if (t < 12) return pick3(sn.gang, t - 9);
  if (t < 24) return s.qpos[7 + t - 12];
  if (t < 36) return s.qvel[6 + t - 24];
  if (t < 48) return s.aforce[t - 36];
  if (t == 48) return s.spos[3 * m.env_ids[0] + 2];
  return 0.0f;
}

// handstand.py:119-160 + wrappers
template <class C>
__global__ __launch_bounds__(64) void hs_reset_kernel(const DModel* __restrict__ mp, Layout L, StepArgs a) {
  const DModel& m = *mp;
  const Hot hot = make_hot(m);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem<C>& s = *reinterpret_cast<Smem<C>*>(smem_raw);
  const int e = blockIdx.x, lane = threadIdx.x;
  if (e >= a.n) return;
  float* rec = a.state + (size_t)e * L.rec;
  const float* F = m.env_go2f;
  uint32_t* bits = reinterpret_cast<uint32_t*>(s.scratch_b());
  float* obs_lds = s.scratch_b() + 64;
  load_overrides<C>(m, s, a, e, lane);
  for (int t = lane; t < C::NINFO; t += 64) s.ginfo[t] = 0.0f;
  uint32_t rng0 = a.keys[2 * e], rng1 = a.keys[2 * e + 1];
  uint32_t k2[2][2];
  tf_split<2>(rng0, rng1, bits, lane, k2); rng0 = k2[0][0]; rng1 = k2[0][1];
  const float ub = rdlane(tf_uniform(k2[1][0], k2[1][1], 1, 0.0f, 1.0f, bits, lane), 0);
  const bool crouch = ub < F[8];                                       // jax.random.bernoulli(key, p)
  const float q_init = m.env_go2_home[(crouch ? C::NQ : 0) + (lane < C::NQ ? lane : 0)];
  if (lane < C::NQ) s.qpos[lane] = q_init;
  if (lane < C::NV) s.qvel[lane] = 0.0f;
  WSYNC();
  {
    tf_split<2>(rng0, rng1, bits, lane, k2); rng0 = k2[0][0]; rng1 = k2[0][1];
    float dxy = tf_uniform(k2[1][0], k2[1][1], 2, -0.5f, 0.5f, bits, lane);
    if (lane < 2) s.qpos[lane] = q_init + dxy;
    tf_split<2>(rng0, rng1, bits, lane, k2); rng0 = k2[0][0]; rng1 = k2[0][1];
    float yaw = rdlane(tf_uniform(k2[1][0], k2[1][1], 1, -3.14f, 3.14f, bits, lane), 0);
    WSYNC();
    if (lane == 0) {
#pragma clang fp contract(off)
      float sn = sinf(yaw * 0.5f), cs = cosf(yaw * 0.5f);
      Q4 q = ld4(&s.qpos[3]), r = Q4{cs, 0.0f * sn, 0.0f * sn, 1.0f * sn};
      Q4 o;
      o.w = q.w * r.w - q.x * r.x - q.y * r.y - q.z * r.z;
      o.x = q.w * r.x + q.x * r.w + q.y * r.z - q.z * r.y;
      o.y = q.w * r.y - q.x * r.z + q.y * r.w + q.z * r.x;
      o.z = q.w * r.z + q.x * r.y - q.y * r.x + q.z * r.w;
      st4(&s.qpos[3], o);
    }
    tf_split<2>(rng0, rng1, bits, lane, k2); rng0 = k2[0][0]; rng1 = k2[0][1];
    float v6 = tf_uniform(k2[1][0], k2[1][1], 6, -0.5f, 0.5f, bits, lane);
    if (lane < 6 && !crouch) s.qvel[lane] = v6;
  }
  WSYNC();
  if (lane < C::NU) s.ctrl[lane] = s.qpos[7 + lane];          // mjx_env.init(..., ctrl = qpos[7:])
  if (lane == 0) {
    s.xfrc_body = 0; s.acc_body = m.site_bodyid[m.env_ids[0]]; s.xfrc[0] = s.xfrc[1] = s.xfrc[2] = 0.0f;
    s.ginfo[G2_RNG] = __uint_as_float(rng0); s.ginfo[G2_RNG + 1] = __uint_as_float(rng1);
  }
  WSYNC();
  float Mrow[C::NV], warm = 0.0f;
  FwdOut<C> f;
  PROF_DECL
  forward<C>(m, hot, s, lane, Mrow, warm, f, a.debug ? a.debug + (size_t)e * RSR_DEBUG_FLOATS : nullptr PROF_PASS);
  WSYNC();
  G2Sens sn;
  go2_sensors<C>(m, s, sn);
  go2_accelerometer<C>(m, s, lane, f.qacc, sn);
  hs_obs<C>(m, s, sn, obs_lds, bits, lane);
  for (int t = lane; t < GO2_PRIV; t += 64) { float v = hs_priv_elem<C>(m, s, sn, obs_lds, t); rec[L.priv_obs + t] = v; rec[L.f_priv_obs + t] = v; }
  store_pipeline<C>(s, rec, L, lane, warm, 0.0f);
  for (int t = lane; t < C::NINFO; t += 64) rec[L.go2_info + t] = s.ginfo[t];
  for (int t = lane; t < C::OBS; t += 64) { rec[L.obs + t] = obs_lds[t]; rec[L.f_obs + t] = obs_lds[t]; }
  if (lane == 0) {
    rec[L.reward] = 0.0f; rec[L.done] = 0.0f;
    for (int i = 0; i < C::NMET; ++i) rec[L.metrics + i] = 0.0f;
    rec[L.steps] = 0.0f; rec[L.truncation] = 0.0f; rec[L.episode_done] = 0.0f;
    for (int i = 0; i < 2 + C::NMET; ++i) rec[L.episode_metrics + i] = 0.0f;
    int* st = reinterpret_cast<int*>(rec + L.stats);
    st[0] = f.st.niter; st[1] = f.st.ls_total; st[2] = s.ncon; st[3] = s.ncon_drop;
    rec[L.f_time] = 0.0f;
  }
  for (int t = lane; t < C::NQ; t += 64) rec[L.f_qpos + t] = s.qpos[t];
  if (lane < C::NV) { rec[L.f_qvel + lane] = s.qvel[lane]; rec[L.f_warm + lane] = warm; }
  if (lane < C::NU) rec[L.f_ctrl + lane] = s.ctrl[lane];
  for (int t = lane; t < C::NB * 3; t += 64) rec[L.f_xpos + t] = s.xpos[t];
  for (int t = lane; t < C::NS * 3; t += 64) rec[L.f_site_xpos + t] = s.spos[t];
}

// handstand.py:161-195 with the rewards :264-342, + wrappers
template <class C>
#ifndef RSR_HS_WAVES_PER_EU
#define RSR_HS_WAVES_PER_EU 3
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RSR_HS_WAVES_PER_EU, RSR_HS_WAVES_PER_EU)))
void hs_step_kernel(const DModel* __restrict__ mp, Layout L, StepArgs a) {
  const DModel& m = *mp;
  const Hot hot = make_hot(m);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem<C>& s = *reinterpret_cast<Smem<C>*>(smem_raw);
  const int e = blockIdx.x, lane = threadIdx.x;
  if (e >= a.n) return;
  float* rec = a.state + (size_t)e * L.rec;
  const bool wrap_episode = m.wrap_flags & 1, wrap_autoreset = (m.wrap_flags & 2) != 0;
  const float* F = m.env_go2f;
  const float dt = F[0];
  uint32_t* bits = reinterpret_cast<uint32_t*>(s.scratch_b());
  float* obs_lds = s.scratch_b() + 64;
  float* rwl = s.scratch_b() + 128;                                   // scaled reward terms staged for the metrics write
  float* qacc_lds = s.scratch_b() + 160;                              // qacc of the last forward pass, for the dof_acc term
  PROF_DECL
  for (int t = lane; t < C::NQ; t += 64) s.qpos[t] = rec[L.qpos + t];
  float warm = 0.0f;
  if (lane < C::NV) { s.qvel[lane] = rec[L.qvel + lane]; warm = rec[L.warm + lane]; }
  float time = rec[L.time];
  load_overrides<C>(m, s, a, e, lane);
  for (int t = lane; t < C::NINFO; t += 64) s.ginfo[t] = rec[L.go2_info + t];
  const float done_prev = rec[L.done];
  float steps = rec[L.steps];
  if (wrap_autoreset && done_prev != 0.0f) steps = 0.0f;
  const float act_in = lane < C::NU ? a.action[(size_t)e * C::NU + lane] : 0.0f;
  if (lane < C::NU) {
#pragma clang fp contract(off)
    float sc = act_in * F[1];
    s.ctrl[lane] = rec[L.ctrl + lane] + sc;                          // motor targets = state.data.ctrl + action * action_scale (:162)
  }
  if (lane == 0) { s.acc_body = m.site_bodyid[m.env_ids[0]]; s.xfrc_body = 0; s.xfrc[0] = s.xfrc[1] = s.xfrc[2] = 0.0f; }
  WSYNC();
  PROF(PS_LOAD)
  float Mrow[C::NV];
  FwdOut<C> f;
  const int prio_q = prio_quarter(a, e);
  for (int fr = 0; fr < m.n_frames; ++fr) {
#if defined(RSR_PROFILE) || defined(RSR_TIMELINE)
    float* dbg = nullptr;
#else
    float* dbg = (a.debug && fr == m.n_frames - 1) ? a.debug + (size_t)e * RSR_DEBUG_FLOATS : nullptr;
#endif
    const int lane_s = lrec_lane(lane);
    prio_substep(a.prio_mode, prio_q, fr);
    forward<C>(m, hot, s, lane_s, Mrow, warm, f, dbg PROF_PASS);
    integrate<C>(m, hot, s, lane_s, Mrow, f PROF_PASS);
    time += hot.timestep;
  }
  // which pairs the termination / contact cost look at (pairs are static): 1 = an unwanted-contact geom, 2 = a foot of the contact cost
  int pair_class = 0;
  if (lane < C::NP) {
    const int g2 = m.pair_geom2[lane];
#pragma unroll
    for (int k = 0; k < 12; ++k) if (g2 == m.env_ids[2 + k]) pair_class = 1;
#pragma unroll
    for (int k = 0; k < 2; ++k) if (g2 == m.env_ids[14 + k]) pair_class = 2;
  }
  const float prev_done = wrap_episode ? rec[L.episode_done] : 0.0f;
  const float em_old = (wrap_episode && lane < C::NMET + 2) ? rec[L.episode_metrics + lane] : 0.0f;
  G2Sens sn;
  go2_sensors<C>(m, s, sn);
  go2_accelerometer<C>(m, s, lane, f.qacc, sn);
  bool unwanted, feet;
  {
    const int nc = s.ncon, ci = lane < nc ? lane : 0;
    const int cp = s.cpair[ci]; const float cd = s.cdist[ci];
    const int cls = __shfl(pair_class, cp & 63);
    const int mine = (lane < nc && cd < 0.0f) ? cls : 0;
    unwanted = __ballot(mine == 1) != 0ull; feet = __ballot(mine == 2) != 0ull;
  }
  if (lane < C::NV) qacc_lds[lane] = f.qacc;
  WSYNC();
  hs_obs<C>(m, s, sn, obs_lds, bits, lane);
  float priv[2];
  priv[0] = hs_priv_elem<C>(m, s, sn, obs_lds, lane);
  priv[1] = hs_priv_elem<C>(m, s, sn, obs_lds, lane + 64 < GO2_PRIV ? lane + 64 : GO2_PRIV - 1);
  float reward = 0.0f, done = 0.0f;
  if (lane == 0) {
#pragma clang fp contract(off)
    const float* SC = m.env_go2_scales; const float* home = m.env_go2_home; const float* soft = m.env_go2_soft;
    const int imu = m.env_ids[0];
    const float torso_height = s.spos[3 * imu + 2];
    float energy = 0.0f;
    for (int i = 0; i < 12; ++i) energy += fabsf(s.aforce[i]) * fabsf(s.qvel[6 + i]);
    done = (sn.up[2] < -0.25f || unwanted || energy > F[9]) ? 1.0f : 0.0f;
    float rw[HM_COUNT];
    {
      float h = torso_height < F[10] ? torso_height : F[10];
      float err = F[10] - h;
      rw[HM_HEIGHT] = expf(-err / 1.0f);
      const float* R = &s.smat[9 * imu];
      float c0 = R[0] * F[11]; float c1 = R[3] * F[12]; float c2 = R[6] * F[13];
      float cd = c0 + c1; float cos_dist = cd + c2;
      float nrm = 0.5f * cos_dist; float nr = nrm + 0.5f;
      rw[HM_ORIENT] = nr * nr;
      rw[HM_CONTACT] = feet ? 1.0f : 0.0f;
      float ar = 0.0f, tq = 0.0f, lim = 0.0f, dacc = 0.0f, pose = 0.0f, en = 0.0f;
      for (int i = 0; i < 12; ++i) {
        float da = a.action[(size_t)e * C::NU + i] - s.ginfo[HS_LAST_ACT + i]; ar += da * da;
        float t = s.aforce[i]; tq += t * t;
        float q = s.qpos[7 + i];
        float lo_ = q - soft[i]; float hi_ = q - soft[12 + i];
        lim += -(lo_ < 0.0f ? lo_ : 0.0f) + (hi_ > 0.0f ? hi_ : 0.0f);
        float qa = qacc_lds[6 + i]; dacc += qa * qa;
        en += fabsf(s.qvel[6 + i]) * fabsf(t);
      }
      for (int k = 0; k < 6; ++k) { const int j = m.env_go2i[k]; float dq = s.qpos[7 + j] - home[7 + j]; pose += dq * dq; }
      rw[HM_ACTION_RATE] = ar; rw[HM_TORQUES] = tq; rw[HM_TERM] = done; rw[HM_DOF_LIMITS] = lim; rw[HM_DOF_ACC] = dacc; rw[HM_POSE] = pose;
      float ss = s.qvel[0] * s.qvel[0] + s.qvel[1] * s.qvel[1];
      rw[HM_STAY_STILL] = ss + s.qvel[5] * s.qvel[5];
      rw[HM_ENERGY] = en;
    }
    for (int k = 0; k < HM_COUNT; ++k) rwl[k] = rw[k] * SC[k];
    const int order[HM_COUNT] = {HM_HEIGHT, HM_ORIENT, HM_CONTACT, HM_ACTION_RATE, HM_TORQUES, HM_TERM, HM_DOF_LIMITS, HM_DOF_ACC, HM_POSE,
                                 HM_STAY_STILL, HM_ENERGY};
    float total = 0.0f;
    for (int k = 0; k < HM_COUNT; ++k) total = total + rwl[order[k]];
    reward = clampf(total * dt, 0.0f, 10000.0f);
  }
  reward = rdlane(reward, 0); done = rdlane(done, 0);
  WSYNC();
  if (lane < C::NU) s.ginfo[HS_LAST_ACT + lane] = act_in;
  if (lane == 0) s.ginfo[HS_STEP] += 1.0f;
  WSYNC();
  {
    bool over = false;
    float trunc = 0.0f;
    if (wrap_episode) {
      steps += 1.0f;
      over = steps >= (float)m.episode_length;
      trunc = over ? 1.0f - done : 0.0f;
    }
    if (lane < C::NMET) rec[L.metrics + lane] = rwl[lane];
    if (wrap_episode && lane < C::NMET + 2) {
      float* em = rec + L.episode_metrics;
      const float add = lane == 0 ? reward : (lane == 1 ? 1.0f : rwl[lane >= 2 ? lane - 2 : 0]);
      em[lane] = prev_done != 0.0f ? 0.0f : em_old + add;
    }
    if (over) done = 1.0f;
    if (lane == 0) {
      rec[L.reward] = reward;
      if (wrap_episode) { rec[L.truncation] = trunc; rec[L.episode_done] = done; }
      rec[L.steps] = steps;
      rec[L.done] = done;
      int* st = reinterpret_cast<int*>(rec + L.stats);
      st[0] = f.st.niter; st[1] = f.st.ls_total; st[2] = s.ncon; st[3] = s.ncon_drop;
    }
  }
  WSYNC();
  for (int t = lane; t < C::NINFO; t += 64) rec[L.go2_info + t] = s.ginfo[t];       // info is never reset by AutoReset
  if (wrap_autoreset && done != 0.0f) {
    for (int t = lane; t < L.persist_end; t += 64) rec[t] = rec[L.f_qpos + t];
    for (int t = lane; t < C::OBS; t += 64) rec[L.obs + t] = rec[L.f_obs + t];
    for (int t = lane; t < GO2_PRIV; t += 64) rec[L.priv_obs + t] = rec[L.f_priv_obs + t];
  } else {
    store_pipeline<C>(s, rec, L, lane, warm, time);
    for (int t = lane; t < C::OBS; t += 64) rec[L.obs + t] = obs_lds[t];
    rec[L.priv_obs + lane] = priv[0];
    if (lane + 64 < GO2_PRIV) rec[L.priv_obs + lane + 64] = priv[1];
  }
#ifdef RSR_PROFILE
  PROF(PS_EPILOGUE)
  if (a.debug && lane == 0) {
    float* d = a.debug + (size_t)e * RSR_DEBUG_FLOATS + 7200;
    for (int i = 0; i < PS_COUNT; ++i) d[i] = (float)prof_.acc[i];
  }
#endif
}

// ---------------------------------------------------------------- launchers of the Go2 kernels
// The Go2 kernels are built as a translation unit of their own (this file with -DRSR_TU_GO2 -fno-slp-vectorize, see
// rsr_mjx_amd/build.py): the SLP vectoriser's packed-fp32 pairing costs them ~3% while it gains the Airbot kernels ~1%.
// Kernel templates are instantiated where they are launched, so each unit compiles only its own kernels.
void launch_go2_reset(int n, hipStream_t st, const DModel* dm, Layout L, StepArgs a, bool hfield);
void launch_go2_step(int n, hipStream_t st, const DModel* dm, Layout L, StepArgs a, bool hfield);
void launch_hs_reset(int n, hipStream_t st, const DModel* dm, Layout L, StepArgs a);
void launch_hs_step(int n, hipStream_t st, const DModel* dm, Layout L, StepArgs a);
int go2_step_occupancy(int kind, bool hfield);
// The T-shape kernels likewise (-DRSR_TU_TSHAPE -fno-slp-vectorize: +3 % for them, measured; the cube kernels keep the vectoriser).
void launch_tshape_reset(int n, hipStream_t st, const DModel* dm, Layout L, StepArgs a);
void launch_tshape_step(int grid, hipStream_t st, const DModel* dm, Layout L, StepArgs a, Sched sc);
int tshape_step_occupancy();
#ifdef RSR_TU_TSHAPE
void launch_tshape_reset(int n, hipStream_t st, const DModel* dm, Layout L, StepArgs a) {
  hipLaunchKernelGGL((reset_kernel<TShapeDims, ENV_TSHAPE>), dim3(n), dim3(64), sizeof(Smem<TShapeDims>), st, dm, L, a);
}
void launch_tshape_step(int grid, hipStream_t st, const DModel* dm, Layout L, StepArgs a, Sched sc) {
  hipLaunchKernelGGL((step_kernel<TShapeDims, ENV_TSHAPE>), dim3(grid), dim3(64), sizeof(Smem<TShapeDims>), st, dm, L, a, sc);
}
int tshape_step_occupancy() {
  int per_cu = 0;
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, step_kernel<TShapeDims, ENV_TSHAPE>, 64, sizeof(Smem<TShapeDims>)) == hipSuccess ? per_cu : 0;
}
#endif
#ifdef RSR_TU_GO2
void launch_hs_reset(int n, hipStream_t st, const DModel* dm, Layout L, StepArgs a) {
  hipLaunchKernelGGL((hs_reset_kernel<HandDims>), dim3(n), dim3(64), sizeof(Smem<HandDims>), st, dm, L, a);
}
void launch_hs_step(int n, hipStream_t st, const DModel* dm, Layout L, StepArgs a) {
  hipLaunchKernelGGL((hs_step_kernel<HandDims>), dim3(n), dim3(64), sizeof(Smem<HandDims>), st, dm, L, a);
}
void launch_go2_reset(int n, hipStream_t st, const DModel* dm, Layout L, StepArgs a, bool hfield) {
  if (hfield) hipLaunchKernelGGL((go2_reset_kernel<Go2Dims>), dim3(n), dim3(64), sizeof(Smem<Go2Dims>), st, dm, L, a);
  else hipLaunchKernelGGL((go2_reset_kernel<Go2FlatDims>), dim3(n), dim3(64), sizeof(Smem<Go2FlatDims>), st, dm, L, a);
}
int go2_step_occupancy(int kind, bool hfield) {       // resident workgroups per CU of the step kernel (kind 1: handstand)
  int per_cu = 0;
  hipError_t e;
  if (kind == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, hs_step_kernel<HandDims>, 64, sizeof(Smem<HandDims>));
  else if (hfield) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, go2_step_kernel<Go2Dims>, 64, sizeof(Smem<Go2Dims>));
  else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, go2_step_kernel<Go2FlatDims>, 64, sizeof(Smem<Go2FlatDims>));
  return e == hipSuccess ? per_cu : 0;
}
void launch_go2_step(int n, hipStream_t st, const DModel* dm, Layout L, StepArgs a, bool hfield) {
  static_assert(sizeof(Smem<Go2FlatDims>) == sizeof(Smem<Go2Dims>), "one LDS image for both Go2 joystick kernels (rsr_model_dims reports it)");
  if (hfield) hipLaunchKernelGGL((go2_step_kernel<Go2Dims>), dim3(n), dim3(64), sizeof(Smem<Go2Dims>), st, dm, L, a);
  else hipLaunchKernelGGL((go2_step_kernel<Go2FlatDims>), dim3(n), dim3(64), sizeof(Smem<Go2FlatDims>), st, dm, L, a);
}
#endif

#if !defined(RSR_TU_GO2) && !defined(RSR_TU_TSHAPE)
// ---------------------------------------------------------------- end-of-rollout metric reduction
// One launch instead of a handful of library reductions: out = {envs, sum of reward, sum of done, mean of the running episode's
// summed reward} over the batch, summed in a fixed order (per-thread strided partial sums, then a binary tree in LDS), so the
// result does not depend on timing.  One workgroup: the batch is a few thousand records and the launch is latency bound.
__global__ __launch_bounds__(1024) void rollout_metrics_kernel(const float* __restrict__ state, Layout L, int n, float* __restrict__ out) {
  __shared__ float red[3][1024];
  const int t = threadIdx.x;
  float r = 0.0f, d = 0.0f, em = 0.0f;
  for (int e = t; e < n; e += 1024) {
    const float* rec = state + (size_t)e * L.rec;
    r += rec[L.reward]; d += rec[L.done]; em += rec[L.episode_metrics];
  }
  red[0][t] = r; red[1][t] = d; red[2][t] = em;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if (t < w) { red[0][t] += red[0][t + w]; red[1][t] += red[1][t + w]; red[2][t] += red[2][t + w]; }
    __syncthreads();
  }
  if (t == 0) { out[0] = (float)n; out[1] = red[0][0]; out[2] = red[1][0]; out[3] = red[2][0] / (float)n; }
}

// ---------------------------------------------------------------- action_repeat > 1 (brax EpisodeWrapper.step: scan of env.step)
// The fused step kernels carry the wrappers for action_repeat = 1, the only value the reference passes (RSR/train.py:81).  For a
// larger value rsr_step runs the step kernels `repeat` times on a copy of the model view with the wrapper flags cleared (plain
// env.step) and these three small kernels carry the wrappers around them: the reward is the sum of the repeats' rewards in order,
// steps and the episode length advance by `repeat`, done / truncation / episode metrics are formed once from the last repeat's
// state, AutoReset restores the first state after that.  One wavefront per env; nothing here is on the reference's hot path.
__global__ __launch_bounds__(64) void repeat_pre_kernel(float* __restrict__ state, Layout L, int n, int wrap_flags, float* __restrict__ racc) {
  const int e = blockIdx.x;
  if (e >= n || threadIdx.x != 0) return;
  float* rec = state + (size_t)e * L.rec;
  if ((wrap_flags & 2) != 0 && rec[L.done] != 0.0f) rec[L.steps] = 0.0f;      // AutoResetWrapper.step pre-step
  racc[e] = 0.0f;
}
__global__ __launch_bounds__(256) void repeat_acc_kernel(const float* __restrict__ state, Layout L, int n, float* __restrict__ racc) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < n) racc[e] += state[(size_t)e * L.rec + L.reward];
}
__global__ __launch_bounds__(64) void repeat_post_kernel(float* __restrict__ state, Layout L, int n, int wrap_flags, int repeat, int episode_length,
                                                        int nmet, int obs_dim, int priv_dim, int xfrc_at, const float* __restrict__ racc) {
  const int e = blockIdx.x, lane = threadIdx.x;
  if (e >= n) return;
  float* rec = state + (size_t)e * L.rec;
  const bool wrap_episode = wrap_flags & 1, wrap_autoreset = (wrap_flags & 2) != 0;
  const float reward = racc[e];
  float done = rec[L.done], steps = rec[L.steps];
  const float prev_done = wrap_episode ? rec[L.episode_done] : 0.0f;
  const float em_old = (wrap_episode && lane < nmet + 2) ? rec[L.episode_metrics + lane] : 0.0f;
  const float met = (lane >= 2 && lane < nmet + 2) ? rec[L.metrics + lane - 2] : 0.0f;
  bool over = false;
  float trunc = 0.0f;
  if (wrap_episode) {
    steps += (float)repeat;
    over = steps >= (float)episode_length;
    trunc = over ? 1.0f - done : 0.0f;
    if (lane < nmet + 2) {
      const float add = lane == 0 ? reward : (lane == 1 ? (float)repeat : met);
      rec[L.episode_metrics + lane] = prev_done != 0.0f ? 0.0f : em_old + add;
    }
  }
  if (over) done = 1.0f;
  __syncthreads();
  if (lane == 0) {
    rec[L.reward] = reward;
    if (wrap_episode) { rec[L.truncation] = trunc; rec[L.episode_done] = done; }
    rec[L.steps] = steps;
    rec[L.done] = done;
  }
  if (wrap_autoreset && done != 0.0f) {
    if (xfrc_at >= 0 && lane < 3) rec[L.go2_info + xfrc_at + lane] = 0.0f;      // data.xfrc_applied goes back with `data`
    for (int t = lane; t < L.persist_end; t += 64) rec[t] = rec[L.f_qpos + t];
    for (int t = lane; t < obs_dim; t += 64) rec[L.obs + t] = rec[L.f_obs + t];
    for (int t = lane; t < priv_dim; t += 64) rec[L.priv_obs + t] = rec[L.f_priv_obs + t];
  }
}
#endif

}  // namespace rsr

#if !defined(RSR_TU_GO2) && !defined(RSR_TU_TSHAPE)
// =====================================================================================
// host side: C ABI
// =====================================================================================
using rsr::DModel;
using rsr::Layout;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(RSR_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

struct blob_entry { char name[40]; int32_t dtype, count, offset, reserved; };

struct rsr_model {
  std::vector<char> blob;
  rsr_dims dims;
  Layout layout;
  bool has_hfield = false;      // any PAIR_HFIELD_SPHERE pair: the Go2 kernels with the height-field narrow phase
  const void* find(const char* name, int* count = nullptr) const {
    const int32_t* h = reinterpret_cast<const int32_t*>(blob.data());
    const blob_entry* e = reinterpret_cast<const blob_entry*>(blob.data() + 16);
    for (int i = 0; i < h[2]; ++i)
      if (std::strncmp(e[i].name, name, 40) == 0) { if (count) *count = e[i].count; return blob.data() + e[i].offset; }
    if (count) *count = 0;
    return nullptr;
  }
  ptrdiff_t offset_of(const char* name) const {
    const void* p = find(name);
    return p ? static_cast<const char*>(p) - blob.data() : -1;
  }
};

struct rsr_batch {
  const rsr_model* model;
  int n, device;
  float* state; bool owns_state;
  char* dblob;
  DModel dm;            // host copy of the device model view
  DModel* dmodel;       // the same struct in device memory (kernels take a pointer: fewer live SGPRs)
  const float *dr_fric, *dr_mass, *dr_damp, *dr_floss;
  const float* dr_ex[5];    // body_ipos, qpos0, dof_armature, actuator_gainprm, actuator_biasprm
  float* debug;
  hipEvent_t ev0, ev1; bool timing; int launches;
  // work-queue dispatch of the Airbot step kernels (rsr_device.hpp: Sched)
  int* sched;           // device: ticket[2], err[2], then flags[n]
  unsigned launch_id;
  int units, step_grid;
  int spin_cap, withhold_env;   // rsr_batch_set_fault_injection (test hook)
  int whole_envs;               // rsr_batch_set_whole_envs: envs stepped as one unit each (-1: all but two resident rounds' worth)
  int prio_policy, prio_slots;  // rsr_batch_set_priority (-1: chosen from the batch size per launch); resident waves of the step kernel
  int action_repeat;            // rsr_batch_set_action_repeat (1: the wrappers fused in the step kernels)
  DModel* dmodel_plain;         // device copy of the model view with the wrapper flags cleared (action_repeat > 1), or null
  float* racc;                  // [n] reward sums of the repeats, or null
};

static Layout make_layout(const rsr_dims& d) {
  Layout L{};
  int o = 0;
  auto take = [&](int n) { int r = o; o += n; return r; };
  L.qpos = take(d.nq); L.qvel = take(d.nv); L.ctrl = take(d.nu); L.warm = take(d.nv); L.time = take(1);
  L.xpos = take(d.nbody * 3); L.site_xpos = take(d.nsite * 3);
  L.persist_end = o;
  L.f_qpos = take(d.nq); L.f_qvel = take(d.nv); L.f_ctrl = take(d.nu); L.f_warm = take(d.nv); L.f_time = take(1);
  L.f_xpos = take(d.nbody * 3); L.f_site_xpos = take(d.nsite * 3);
  L.obs = take(d.obs_dim); L.f_obs = take(d.obs_dim);
  L.reward = take(1); L.done = take(1); L.metrics = take(d.nmetrics);
  L.target_pos = take(3); L.new_cube_pos = take(2); L.site_pos = take(3); L.cube_pos = take(3); L.last_action = take(1);
  L.target_base_pos = take(3); L.target_vertical_pos = take(3); L.target_w = take(1); L.new_T_pos = take(2);
  L.T_pos = take(3); L.xita = take(1);
  const bool go2_family = d.env_kind == rsr::ENV_GO2 || d.env_kind == rsr::ENV_GO2_HANDSTAND;
  L.go2_info = take(go2_family ? 144 : 0);
  L.priv_obs = take(go2_family ? 123 : 0); L.f_priv_obs = take(go2_family ? 123 : 0);
  L.steps = take(1); L.truncation = take(1); L.episode_done = take(1); L.episode_metrics = take(2 + d.nmetrics);
  L.stats = take(4);
  L.rec = (o + 15) & ~15;
  return L;
}

extern "C" const char* rsr_last_error(void) { return g_err.c_str(); }

extern "C" int rsr_model_create(const void* blob, size_t nbytes, rsr_model** out) {
  if (!blob || !out || nbytes < 32) return fail(RSR_ERR_ARG, "rsr_model_create: null or short blob");
  const int32_t* h = static_cast<const int32_t*>(blob);
  if (std::memcmp(blob, "RSRM", 4) != 0 || h[1] != 2 || (size_t)h[3] > nbytes)
    return fail(RSR_ERR_ARG, "rsr_model_create: not an RSRM v2 blob (v2: the lane records carry solimp clamped and with 1 / width: rsr_mjx_amd/model.py impedance_consts)");
  {  // every directory entry must lie inside the blob before anything is read through it (header: magic, version, entry count, bytes)
    const long long nent = h[2];
    if (nent < 0 || 16 + (unsigned long long)nent * sizeof(blob_entry) > nbytes) return fail(RSR_ERR_ARG, "rsr_model_create: entry table exceeds the blob");
    const blob_entry* e = reinterpret_cast<const blob_entry*>(static_cast<const char*>(blob) + 16);
    for (long long i = 0; i < nent; ++i) {
      const bool named = std::memchr(e[i].name, 0, sizeof(e[i].name)) != nullptr;
      if (!named || e[i].count < 0 || e[i].offset < 0 || (e[i].offset & 3) || (unsigned long long)e[i].offset + 4ull * (unsigned long long)e[i].count > nbytes)
        return fail(RSR_ERR_ARG, "rsr_model_create: blob entry " + std::to_string(i) + " has no name terminator or points outside the blob");
    }
  }
  rsr_model* m = new rsr_model();
  m->blob.assign(static_cast<const char*>(blob), static_cast<const char*>(blob) + nbytes);
  int ndims = 0, nei = 0;
  const int* dims = static_cast<const int*>(m->find("dims", &ndims));
  const int* ei = static_cast<const int*>(m->find("env_int", &nei));
  if (!dims || !ei || ndims < 9 || nei < 6) { delete m; return fail(RSR_ERR_ARG, "rsr_model_create: blob lacks dims/env_int"); }
  {  // fields read below or by fill_dmodel without a further check
    static const char* const need[] = {"counts2", "opt_integrator", "opt_timestep", "opt_gravity", "opt_tolerance", "opt_ls_tolerance", "opt_impratio",
                                       "stat_meaninertia", "opt_iterations", "opt_ls_iterations", "opt_disable_eulerdamp", "opt_disable_refsafe",
                                       "pair_condim", "pair_kind", "pair_geom1", "pair_geom2", "geom_size", "eq_active0", "lane_rec", "geom_slot_ids"};
    for (const char* f : need) {
      int cnt = 0;
      if (!m->find(f, &cnt) || (cnt < 1 && std::strcmp(f, "pair_condim") && std::strcmp(f, "pair_kind") && std::strcmp(f, "pair_geom1") && std::strcmp(f, "pair_geom2") && std::strcmp(f, "eq_active0"))) {
        delete m; return fail(RSR_ERR_ARG, std::string("rsr_model_create: blob lacks field ") + f);
      }
    }
    int nc2 = 0, ng = 0; m->find("counts2", &nc2); m->find("opt_gravity", &ng);
    if (nc2 < 4 || ng < 3) { delete m; return fail(RSR_ERR_ARG, "rsr_model_create: counts2 / opt_gravity too short"); }
  }
  rsr_dims& d = m->dims;
  d.nq = dims[0]; d.nv = dims[1]; d.nu = dims[2]; d.nbody = dims[3]; d.njnt = dims[4]; d.ngeom = dims[5];
  d.nsite = dims[6]; d.neq = dims[7]; d.npair = dims[8];
  d.env_kind = ei[0]; d.n_frames = ei[1]; d.episode_length = ei[2]; d.obs_dim = ei[4]; d.nmetrics = ei[5];
  const int* c2 = static_cast<const int*>(m->find("counts2"));
  auto fits = [&](auto dims_tag) {
    using C = decltype(dims_tag);
    return d.nq == C::NQ && d.nv == C::NV && d.nu == C::NU && d.nbody == C::NB && d.njnt == C::NJ && d.ngeom == C::NG &&
           d.nsite == C::NS && d.npair == C::NP && d.neq == C::NEQ && c2 && c2[0] == C::NF && c2[1] == C::NL &&
           d.obs_dim == C::OBS && d.nmetrics == C::NMET;
  };
  bool ok = ((d.env_kind == rsr::ENV_CUBE || d.env_kind == rsr::ENV_AIRBOT_SF) && fits(rsr::CubeDims{})) ||
            (d.env_kind == rsr::ENV_TSHAPE && fits(rsr::TShapeDims{})) || (d.env_kind == rsr::ENV_GO2 && fits(rsr::Go2Dims{})) ||
            (d.env_kind == rsr::ENV_GO2_HANDSTAND && fits(rsr::HandDims{}));
  const bool go2_family = d.env_kind == rsr::ENV_GO2 || d.env_kind == rsr::ENV_GO2_HANDSTAND;
  const int want_condim = go2_family ? 3 : 4;
  if (!ok) { delete m; return fail(RSR_ERR_UNSUPPORTED, "rsr_model_create: model dims / env kind have no compiled kernel (built: Airbot cube, Airbot sf, Airbot T-shape, Go2 joystick, Go2 handstand / footstand)"); }
  if (c2[3] > 1) { delete m; return fail(RSR_ERR_UNSUPPORTED, "rsr_model_create: bodies with more than one joint are not built"); }
  {  // geom slots: as many as the kernel's LDS image keeps, each a geom id, every pair geom among them (model.py: geom_slots)
    int ns = 0, np1 = 0; const int* gs = static_cast<const int*>(m->find("geom_slot_ids", &ns));
    const int want = d.env_kind == rsr::ENV_GO2 ? rsr::Go2Dims::NGA : (d.env_kind == rsr::ENV_GO2_HANDSTAND ? rsr::HandDims::NGA : (d.env_kind == rsr::ENV_TSHAPE ? rsr::TShapeDims::NGA : rsr::CubeDims::NGA));
    bool okg = gs && ns == want;
    for (int i = 0; okg && i < ns; ++i) okg = gs[i] >= 0 && gs[i] < d.ngeom && (want != d.ngeom || gs[i] == i);
    const int* pg1 = static_cast<const int*>(m->find("pair_geom1", &np1)); const int* pg2 = static_cast<const int*>(m->find("pair_geom2"));
    for (int q = 0; okg && q < np1; ++q) {
      bool f1 = false, f2 = false;
      for (int i = 0; i < ns; ++i) { f1 |= gs[i] == pg1[q]; f2 |= gs[i] == pg2[q]; }
      okg = f1 && f2;
    }
    if (!okg) { delete m; return fail(RSR_ERR_UNSUPPORTED, "rsr_model_create: geom_slot_ids do not match the kernel's geom slots (Dims::NGA) or miss a pair geom"); }
  }
  if (!go2_family) {
    // the Airbot kernels factor one kinematic tree per DPP row (Dims::ROWTREE): dof ranges [0, TREE1), [TREE1, TREE2), [TREE2, nv)
    // must be separate trees -- no body chain and no equality constraint may straddle them
    const int t1 = d.env_kind == rsr::ENV_TSHAPE ? rsr::TShapeDims::TREE1 : rsr::CubeDims::TREE1;
    const int t2 = d.env_kind == rsr::ENV_TSHAPE ? rsr::TShapeDims::TREE2 : rsr::CubeDims::TREE2;
    auto trees_of = [&](unsigned mask) {
      const unsigned m0 = (1u << t1) - 1u, m01 = (1u << t2) - 1u;
      return ((mask & m0) != 0u) + ((mask & (m01 & ~m0)) != 0u) + ((mask & ~m01) != 0u);
    };
    int nbm = 0, neq = 0;
    const unsigned* bm = static_cast<const unsigned*>(m->find("body_dofmask", &nbm));
    bool okt = bm != nullptr;
    for (int b = 0; okt && b < nbm; ++b) okt = trees_of(bm[b]) <= 1;
    const int* e1 = static_cast<const int*>(m->find("eq_obj1id", &neq)); const int* e2 = static_cast<const int*>(m->find("eq_obj2id"));
    const int* jd = static_cast<const int*>(m->find("jnt_dofadr"));
    for (int q = 0; okt && e1 && e2 && jd && q < neq; ++q) {
      const bool j1 = e1[q] >= 0 && e1[q] < d.njnt, j2 = e2[q] >= 0 && e2[q] < d.njnt;
      if (j1 && j2) okt = trees_of((1u << jd[e1[q]]) | (1u << jd[e2[q]])) <= 1;
    }
    if (!okt) { delete m; return fail(RSR_ERR_UNSUPPORTED, "rsr_model_create: the Airbot kernels need the arm and the free bodies as separate kinematic trees over fixed dof ranges"); }
  }
  if (go2_family && rsr::Go2Dims::ARROW) {
    // the Go2 kernels factor M and H in block-arrow form (Dims::ARROW): dofs 0..5 are the trunk, every further group of three dofs
    // is a leg, and no body chain and no contact pair may touch two legs
    using G = rsr::Go2Dims;
    auto legs_of = [](unsigned mask) { int n = 0; for (int l = 0; l < G::ALEGS; ++l) n += ((mask >> (G::ANT + G::ALEGN * l)) & 7u) != 0u; return n; };
    int nbm = 0, npm = 0;
    const unsigned* bm = static_cast<const unsigned*>(m->find("body_dofmask", &nbm));
    const unsigned* pm1 = static_cast<const unsigned*>(m->find("pair_mask1", &npm)); const unsigned* pm2 = static_cast<const unsigned*>(m->find("pair_mask2"));
    bool oka = d.nv == G::NV && bm && pm1 && pm2;
    for (int b = 0; oka && b < nbm; ++b) oka = legs_of(bm[b]) <= 1;
    for (int q = 0; oka && q < npm; ++q) oka = legs_of(pm1[q] | pm2[q]) <= 1;
    if (!oka) { delete m; return fail(RSR_ERR_UNSUPPORTED, "rsr_model_create: the Go2 kernels need a trunk of 6 dofs carrying legs of 3 dofs that only couple through the trunk"); }
  }
  int npc = 0; const int* pc = static_cast<const int*>(m->find("pair_condim", &npc));
  for (int i = 0; i < npc; ++i) if (pc[i] != want_condim) { delete m; return fail(RSR_ERR_UNSUPPORTED, "rsr_model_create: contact pairs must all have the condim the kernel is built for (Airbot 4, Go2 3)"); }
  int nea = 0; const int* ea = static_cast<const int*>(m->find("eq_active0", &nea));
  for (int i = 0; i < nea; ++i) if (!ea[i]) { delete m; return fail(RSR_ERR_UNSUPPORTED, "rsr_model_create: inactive equality constraints are not built"); }
  if (static_cast<const int*>(m->find("opt_integrator"))[0] != rsr::INT_IMPLICITFAST &&
      static_cast<const int*>(m->find("opt_integrator"))[0] != rsr::INT_EULER) { delete m; return fail(RSR_ERR_UNSUPPORTED, "rsr_model_create: integrator"); }
  {  // the kernels treat dofs [ISO0, ISO1) as decoupled from the rest: no chain, pair or equality may straddle the range
    const int iso0 = (d.env_kind == rsr::ENV_CUBE || d.env_kind == rsr::ENV_AIRBOT_SF) ? rsr::CubeDims::ISO0 : 0;
    const int iso1 = (d.env_kind == rsr::ENV_CUBE || d.env_kind == rsr::ENV_AIRBOT_SF) ? rsr::CubeDims::ISO1 : 0;
    if (iso1 > iso0) {
      const unsigned iso = ((1u << iso1) - 1u) & ~((1u << iso0) - 1u);
      auto straddles = [&](unsigned mask) { return (mask & iso) && (mask & ~iso); };
      int nb = 0, np1 = 0, np2 = 0, ne1 = 0, ne2 = 0, nj = 0;
      const unsigned* bm = static_cast<const unsigned*>(m->find("body_dofmask", &nb));
      const unsigned* m1 = static_cast<const unsigned*>(m->find("pair_mask1", &np1));
      const unsigned* m2 = static_cast<const unsigned*>(m->find("pair_mask2", &np2));
      const int* e1 = static_cast<const int*>(m->find("eq_obj1id", &ne1));
      const int* e2 = static_cast<const int*>(m->find("eq_obj2id", &ne2));
      const int* jd = static_cast<const int*>(m->find("jnt_dofadr", &nj));
      bool bad = !bm || !m1 || !m2 || np1 != np2 || !jd;
      const unsigned low = (1u << iso0) - 1u;      // the trees before / after the range must be separate too (mass matrix blocks)
      for (int i = 0; !bad && i < nb; ++i) bad = straddles(bm[i]) || ((bm[i] & low) && (bm[i] & ~low));
      for (int i = 0; !bad && i < np1; ++i) bad = straddles(m1[i] | m2[i]);
      for (int i = 0; !bad && e1 && e2 && i < ne1 && i < ne2; ++i) {
        unsigned mk = (e1[i] >= 0 && e1[i] < nj ? 1u << jd[e1[i]] : 0u) | (e2[i] >= 0 && e2[i] < nj ? 1u << jd[e2[i]] : 0u);
        bad = straddles(mk);
      }
      if (bad) { delete m; return fail(RSR_ERR_UNSUPPORTED, "rsr_model_create: the kernel assumes the target body's dofs share no chain, contact pair or equality with other dofs"); }
    }
  }
  {  // height-field pairs: Go2 kernels only, one field, spheres no wider than a grid cell
    int npk = 0, nh = 0, nsz = 0; const int* pk = static_cast<const int*>(m->find("pair_kind", &npk));
    bool any_hf = false;
    for (int i = 0; pk && i < npk; ++i) any_hf |= (pk[i] == rsr::PAIR_HFIELD_SPHERE);
    m->has_hfield = any_hf;
    if (any_hf) {
      const int* hr = static_cast<const int*>(m->find("hfield_nrow", &nh));
      const int* hc = static_cast<const int*>(m->find("hfield_ncol"));
      const float* hs = static_cast<const float*>(m->find("hfield_size", &nsz));
      int nd = 0; m->find("hfield_data", &nd);
      bool okh = d.env_kind == rsr::ENV_GO2 && rsr::Go2Dims::HFIELD && hr && hc && hs && nh == 1 && nsz == 4 && hr[0] >= 3 && hc[0] >= 3 && nd == hr[0] * hc[0];
      if (okh) {
        const float cell = std::fmin(2.0f * hs[0] / (float)(hc[0] - 1), 2.0f * hs[1] / (float)(hr[0] - 1));
        const int* g2 = static_cast<const int*>(m->find("pair_geom2"));
        const float* gs = static_cast<const float*>(m->find("geom_size"));
        int ng2 = 0, ngs = 0; m->find("pair_geom2", &ng2); m->find("geom_size", &ngs);
        for (int i = 0; i < npk; ++i)
          if (pk[i] == rsr::PAIR_HFIELD_SPHERE && (i >= ng2 || g2[i] < 0 || 3 * g2[i] + 2 >= ngs || 2.0f * gs[3 * g2[i]] > cell)) okh = false;
      }
      if (!okh) { delete m; return fail(RSR_ERR_UNSUPPORTED, "rsr_model_create: height-field pairs need the Go2 kernels, exactly one height field of at least 3x3 samples, and spheres no wider than a grid cell"); }
    }
  }
  if (d.env_kind == rsr::ENV_GO2) { using C = rsr::Go2Dims; d.ncon_max = C::NCON; d.nefc_max = C::NEFC; d.lds_bytes = (int32_t)sizeof(rsr::Smem<C>); }
  else if (d.env_kind == rsr::ENV_GO2_HANDSTAND) { using C = rsr::HandDims; d.ncon_max = C::NCON; d.nefc_max = C::NEFC; d.lds_bytes = (int32_t)sizeof(rsr::Smem<C>); }
  else if (d.env_kind == rsr::ENV_TSHAPE) { using C = rsr::TShapeDims; d.ncon_max = C::NCON; d.nefc_max = C::NEFC; d.lds_bytes = (int32_t)sizeof(rsr::Smem<C>); }
  else { using C = rsr::CubeDims; d.ncon_max = C::NCON; d.nefc_max = C::NEFC; d.lds_bytes = (int32_t)sizeof(rsr::Smem<C>); }
  m->layout = make_layout(d);
  d.rec_floats = m->layout.rec;
  *out = m;
  return RSR_OK;
}

extern "C" int rsr_model_dims(const rsr_model* m, rsr_dims* out) {
  if (!m || !out) return fail(RSR_ERR_ARG, "rsr_model_dims: null");
  *out = m->dims;
  return RSR_OK;
}
extern "C" void rsr_model_destroy(rsr_model* m) { delete m; }

static int fill_dmodel(const rsr_model* m, const char* dbase, DModel& dm) {
#define P(T, name) { ptrdiff_t o = m->offset_of(#name); if (o < 0) return fail(RSR_ERR_ARG, "blob lacks field " #name); dm.name = (decltype(dm.name))(dbase + o); }
  P(int, body_parentid) P(int, body_rootid) P(int, body_jntnum) P(int, body_jntadr) P(int, body_dofnum) P(int, body_dofadr) P(int, body_depth)
  P(float, body_pos) P(float, body_quat) P(float, body_ipos) P(float, body_iquat) P(float, body_mass) P(float, body_inertia) P(float, body_invweight0)
  P(int, jnt_type) P(int, jnt_qposadr) P(int, jnt_dofadr) P(int, jnt_bodyid) P(int, jnt_limited) P(int, jnt_actfrclimited)
  P(float, jnt_pos) P(float, jnt_axis) P(float, jnt_range) P(float, jnt_actfrcrange) P(float, jnt_solref) P(float, jnt_solimp) P(float, jnt_margin)
  P(int, dof_bodyid) P(int, dof_jntid)
  P(unsigned, dof_ancmask) P(unsigned, dof_velmask) P(unsigned, body_dofmask) P(unsigned, body_submask)
  P(float, dof_armature) P(float, dof_damping) P(float, dof_frictionloss) P(float, dof_invweight0) P(float, dof_solref) P(float, dof_solimp)
  P(int, geom_bodyid) P(int, geom_priority) P(float, geom_size) P(float, geom_pos) P(float, geom_quat) P(float, geom_friction) P(int, geom_slot_ids)
  P(int, site_bodyid) P(float, site_pos) P(float, site_quat)
  P(int, eq_obj1id) P(int, eq_obj2id) P(int, eq_active0) P(float, eq_data) P(float, eq_solref) P(float, eq_solimp)
  P(int, actuator_trnid) P(int, actuator_ctrllimited) P(int, actuator_forcelimited)
  P(float, actuator_gear) P(float, actuator_gainprm) P(float, actuator_biasprm) P(float, actuator_ctrlrange) P(float, actuator_forcerange)
  P(int, pair_geom1) P(int, pair_geom2) P(int, pair_kind) P(int, pair_condim)
  P(float, pair_solref) P(float, pair_solimp) P(float, pair_margin) P(float, pair_gap)
  P(int, fric_dofs) P(int, limit_jnts) P(float, qpos0)
  P(int, pair_b1) P(int, pair_b2) P(int, pair_root1) P(int, pair_root2) P(unsigned, pair_mask1) P(unsigned, pair_mask2)
  P(float, pair_tw) P(float, pair_incl) P(int, dof_rootid) P(int, dof_jtype) P(int, dof_k) P(int, dof_act) P(int, dof_afl)
  P(float, dof_afrange) P(int, body_jtype) P(int, body_qposadr) P(float, body_jpos) P(float, body_jaxis)
  P(int4, lane_rec)
  { int nrec = 0; m->find("lane_rec", &nrec); if (nrec != rsr::LQ_COUNT * 64 * 4) return fail(RSR_ERR_ARG, "blob field lane_rec has the wrong size (model.py lane_records vs enum LaneQuad)"); }
  P(float, hfield_size) P(float, hfield_data) P(int, hfield_nrow) P(int, hfield_ncol)
  P(int, env_ids) P(float, env_action_scale) P(float, env_ctrl_lo) P(float, env_ctrl_hi) P(float, env_reset) P(float, env_reward)
  if (static_cast<const int*>(m->find("env_int"))[0] == rsr::ENV_GO2 || static_cast<const int*>(m->find("env_int"))[0] == rsr::ENV_GO2_HANDSTAND) {
    P(float, env_go2f) P(float, env_go2_scales) P(float, env_go2_home) P(float, env_go2_soft) P(int, env_go2i)
  } else { dm.env_go2f = dm.env_go2_scales = dm.env_go2_home = dm.env_go2_soft = nullptr; dm.env_go2i = nullptr; }
#undef P
  auto F = [&](const char* n) { return static_cast<const float*>(m->find(n)); };
  auto I = [&](const char* n) { return static_cast<const int*>(m->find(n)); };
  dm.timestep = F("opt_timestep")[0];
  for (int i = 0; i < 3; ++i) dm.gravity[i] = F("opt_gravity")[i];
  dm.tolerance = F("opt_tolerance")[0]; dm.ls_tolerance = F("opt_ls_tolerance")[0]; dm.impratio = F("opt_impratio")[0];
  dm.meaninertia = F("stat_meaninertia")[0];
  dm.iterations = I("opt_iterations")[0]; dm.ls_iterations = I("opt_ls_iterations")[0]; dm.integrator = I("opt_integrator")[0];
  dm.disable_eulerdamp = I("opt_disable_eulerdamp")[0]; dm.disable_refsafe = I("opt_disable_refsafe")[0];
  dm.nfric = I("counts2")[0]; dm.nlimit = I("counts2")[1]; dm.maxdepth = I("counts2")[2];
  { int nc2 = 0; const int* c2 = static_cast<const int*>(m->find("counts2", &nc2)); dm.max_sub = nc2 > 5 ? c2[4] : 32; dm.max_chain = nc2 > 5 ? c2[5] : 32; }
  const int* ei = I("env_int");
  dm.env_kind = ei[0]; dm.n_frames = ei[1]; dm.episode_length = ei[2]; dm.wrap_flags = ei[3];
  return RSR_OK;
}

extern "C" int rsr_batch_create(const rsr_model* m, int num_envs, int hip_device, float* state, rsr_batch** out) {
  if (!m || !out || num_envs <= 0) return fail(RSR_ERR_ARG, "rsr_batch_create: bad argument");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return fail(RSR_ERR_HIP, std::string("rsr_batch_create: no HIP device (this library has no CPU path): hipGetDeviceCount -> ") +
                                 hipGetErrorString(e) + ", count " + std::to_string(ndev));
  if (hip_device < 0 || hip_device >= ndev) return fail(RSR_ERR_ARG, "rsr_batch_create: hip_device out of range");
  HIPCHK(hipSetDevice(hip_device));
  rsr_batch* b = new rsr_batch();
  b->model = m; b->n = num_envs; b->device = hip_device;
  b->dr_fric = b->dr_mass = b->dr_damp = b->dr_floss = nullptr; b->debug = nullptr;
  for (auto& p : b->dr_ex) p = nullptr;
  b->timing = false; b->launches = 0; b->ev0 = b->ev1 = nullptr;
  b->state = state; b->owns_state = false; b->dblob = nullptr;
  if (!state) {
    size_t bytes = (size_t)num_envs * m->layout.rec * sizeof(float);
    if (hipMalloc(&b->state, bytes) != hipSuccess) { delete b; return fail(RSR_ERR_NOMEM, "rsr_batch_create: hipMalloc(state)"); }
    b->owns_state = true;
    (void)hipMemset(b->state, 0, bytes);
  }
  if (hipMalloc(&b->dblob, m->blob.size()) != hipSuccess) { if (b->owns_state) (void)hipFree(b->state); delete b; return fail(RSR_ERR_NOMEM, "rsr_batch_create: hipMalloc(model)"); }
  auto release = [&]() {            // every failure path below frees what has been allocated so far
    if (b->dmodel) (void)hipFree(b->dmodel);
    if (b->sched) (void)hipFree(b->sched);
    if (b->dblob) (void)hipFree(b->dblob);
    if (b->owns_state && b->state) (void)hipFree(b->state);
    delete b;
  };
  b->dmodel = nullptr; b->sched = nullptr;
  { hipError_t ce = hipMemcpy(b->dblob, m->blob.data(), m->blob.size(), hipMemcpyHostToDevice);
    if (ce != hipSuccess) { release(); return fail(RSR_ERR_HIP, std::string("rsr_batch_create: hipMemcpy(model): ") + hipGetErrorString(ce)); } }
  int rc = fill_dmodel(m, b->dblob, b->dm);
  if (rc) { release(); return rc; }
  b->launch_id = 0; b->units = 1; b->step_grid = 0; b->spin_cap = RSR_SPIN_CAP_DEFAULT; b->withhold_env = -1; b->whole_envs = -1;
  b->prio_policy = -1; b->prio_slots = 1;
  b->action_repeat = 1; b->dmodel_plain = nullptr; b->racc = nullptr;
  if (m->dims.env_kind == rsr::ENV_GO2 || m->dims.env_kind == rsr::ENV_GO2_HANDSTAND) {
    int per_cu = rsr::go2_step_occupancy(m->dims.env_kind == rsr::ENV_GO2_HANDSTAND ? 1 : 0, m->has_hfield);
    hipDeviceProp_t prop;
    if (per_cu <= 0 || hipGetDeviceProperties(&prop, hip_device) != hipSuccess) { per_cu = 16; prop.multiProcessorCount = 256; }
    b->prio_slots = per_cu * prop.multiProcessorCount;
    if (const char* pv = std::getenv("RSR_PRIO_MODE")) b->prio_policy = std::atoi(pv);    // diagnostic (tools/ab_bench.py)
  }
  if (m->dims.env_kind != rsr::ENV_GO2 && m->dims.env_kind != rsr::ENV_GO2_HANDSTAND) {
    const size_t sb = (4 + (size_t)num_envs) * sizeof(int);
    if (hipMalloc(&b->sched, sb) != hipSuccess) { b->sched = nullptr; release(); return fail(RSR_ERR_NOMEM, "rsr_batch_create: hipMalloc(sched)"); }
    (void)hipMemset(b->sched, 0, sb);
    // resident waves of the step kernel on this device: the grid of the persistent launch
    int per_cu = 0; hipDeviceProp_t prop;
    hipError_t oe = hipSuccess;
    if (m->dims.env_kind == rsr::ENV_TSHAPE) per_cu = rsr::tshape_step_occupancy();
    else oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rsr::step_kernel<rsr::CubeDims, rsr::ENV_CUBE>, 64, sizeof(rsr::Smem<rsr::CubeDims>));
    if (oe != hipSuccess || hipGetDeviceProperties(&prop, hip_device) != hipSuccess || per_cu <= 0) { per_cu = 8; prop.multiProcessorCount = 256; }
    if (const char* gv = std::getenv("RSR_GRID_PER_CU")) { const int g = std::atoi(gv); if (g > 0 && g < per_cu) per_cu = g; }   // diagnostic: fewer resident waves
    b->step_grid = per_cu * prop.multiProcessorCount;
    const char* ev = std::getenv("RSR_UNITS");
    b->units = ev ? std::atoi(ev) : RSR_DEFAULT_UNITS;
    if (b->units < 1) b->units = 1;
    if (b->units > m->dims.n_frames) b->units = m->dims.n_frames;
    if (b->units > RSR_MAX_UNITS) b->units = RSR_MAX_UNITS;
    if (const char* wv = std::getenv("RSR_WHOLE_ENVS")) b->whole_envs = std::atoi(wv);      // diagnostic (tools/ab_bench.py)
    b->prio_policy = 0;      // the work queue balances itself: rotate / final-set priorities measured at +-0.2 % on the cube and the T-shape
  }
  if (hipMalloc(&b->dmodel, sizeof(DModel)) != hipSuccess) { b->dmodel = nullptr; release(); return fail(RSR_ERR_NOMEM, "rsr_batch_create: hipMalloc(dmodel)"); }
  { hipError_t ce = hipMemcpy(b->dmodel, &b->dm, sizeof(DModel), hipMemcpyHostToDevice);
    if (ce != hipSuccess) { release(); return fail(RSR_ERR_HIP, std::string("rsr_batch_create: hipMemcpy(dmodel): ") + hipGetErrorString(ce)); } }
  *out = b;
  return RSR_OK;
}

extern "C" void rsr_batch_destroy(rsr_batch* b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  if (b->ev0) (void)hipEventDestroy(b->ev0);
  if (b->ev1) (void)hipEventDestroy(b->ev1);
  if (b->dblob) (void)hipFree(b->dblob);
  if (b->dmodel) (void)hipFree(b->dmodel);
  if (b->dmodel_plain) (void)hipFree(b->dmodel_plain);
  if (b->racc) (void)hipFree(b->racc);
  if (b->sched) (void)hipFree(b->sched);
  if (b->owns_state && b->state) (void)hipFree(b->state);
  delete b;
}

extern "C" int rsr_batch_set_dr(rsr_batch* b, const float* geom_friction, const float* body_mass, const float* dof_damping,
                                const float* dof_frictionloss) {
  if (!b) return fail(RSR_ERR_ARG, "rsr_batch_set_dr: null batch");
  b->dr_fric = geom_friction; b->dr_mass = body_mass; b->dr_damp = dof_damping; b->dr_floss = dof_frictionloss;
  return RSR_OK;
}

extern "C" int rsr_batch_set_dr_field(rsr_batch* b, int dr_field, const float* dev_values) {
  if (!b) return fail(RSR_ERR_ARG, "rsr_batch_set_dr_field: null batch");
  switch (dr_field) {
    case RSR_DR_GEOM_FRICTION: b->dr_fric = dev_values; return RSR_OK;
    case RSR_DR_BODY_MASS: b->dr_mass = dev_values; return RSR_OK;
    case RSR_DR_DOF_DAMPING: b->dr_damp = dev_values; return RSR_OK;
    case RSR_DR_DOF_FRICTIONLOSS: b->dr_floss = dev_values; return RSR_OK;
    case RSR_DR_BODY_IPOS: case RSR_DR_QPOS0: case RSR_DR_DOF_ARMATURE: case RSR_DR_ACTUATOR_GAINPRM: case RSR_DR_ACTUATOR_BIASPRM:
      if (b->model->dims.env_kind != rsr::ENV_GO2 && b->model->dims.env_kind != rsr::ENV_GO2_HANDSTAND)
        return fail(RSR_ERR_UNSUPPORTED, "rsr_batch_set_dr_field: this field is per-env only in the Go2 kernels (randomize.py); the Airbot kernels take the four fields of rsr_batch_set_dr");
      b->dr_ex[dr_field - RSR_DR_BODY_IPOS] = dev_values;
      return RSR_OK;
    default: return fail(RSR_ERR_ARG, "rsr_batch_set_dr_field: unknown field");
  }
}

extern "C" int rsr_batch_set_schedule(rsr_batch* b, int units) {
  if (!b || units < 1) return fail(RSR_ERR_ARG, "rsr_batch_set_schedule: bad argument");
  b->units = units > b->model->dims.n_frames ? b->model->dims.n_frames : units;
  if (b->units > RSR_MAX_UNITS) b->units = RSR_MAX_UNITS;
  return RSR_OK;
}

extern "C" int rsr_batch_set_whole_envs(rsr_batch* b, int whole_envs) {
  if (!b || whole_envs > b->n) return fail(RSR_ERR_ARG, "rsr_batch_set_whole_envs: bad argument");
  b->whole_envs = whole_envs < 0 ? -1 : whole_envs;
  return RSR_OK;
}

extern "C" int rsr_batch_set_action_repeat(rsr_batch* b, int repeat) {
  if (!b || repeat < 1 || repeat > 64) return fail(RSR_ERR_ARG, "rsr_batch_set_action_repeat: repeat must be 1 ... 64");
  HIPCHK(hipSetDevice(b->device));
  if (repeat > 1 && !b->dmodel_plain) {
    DModel plain = b->dm;
    plain.wrap_flags = 0;
    if (hipMalloc(&b->dmodel_plain, sizeof(DModel)) != hipSuccess) { b->dmodel_plain = nullptr; return fail(RSR_ERR_NOMEM, "rsr_batch_set_action_repeat: hipMalloc(dmodel)"); }
    if (hipMalloc(&b->racc, (size_t)b->n * sizeof(float)) != hipSuccess) {
      (void)hipFree(b->dmodel_plain); b->dmodel_plain = nullptr; b->racc = nullptr;
      return fail(RSR_ERR_NOMEM, "rsr_batch_set_action_repeat: hipMalloc(reward sums)");
    }
    HIPCHK(hipMemcpy(b->dmodel_plain, &plain, sizeof(DModel), hipMemcpyHostToDevice));
  }
  b->action_repeat = repeat;
  return RSR_OK;
}

extern "C" int rsr_batch_set_priority(rsr_batch* b, int policy) {
  if (!b || policy > rsr::RSR_PRIO_CATCH_UP) return fail(RSR_ERR_ARG, "rsr_batch_set_priority: bad argument");
  b->prio_policy = policy < 0 ? -1 : policy;
  return RSR_OK;
}

extern "C" int rsr_batch_set_fault_injection(rsr_batch* b, int spin_cap, int withhold_env) {
  if (!b) return fail(RSR_ERR_ARG, "rsr_batch_set_fault_injection: null batch");
  b->spin_cap = spin_cap > 0 ? spin_cap : RSR_SPIN_CAP_DEFAULT;
  b->withhold_env = (withhold_env >= 0 && withhold_env < b->n) ? withhold_env : -1;
  return RSR_OK;
}

extern "C" int rsr_batch_check(rsr_batch* b, void* hip_stream, int* handoff_timeouts) {
  if (!b) return fail(RSR_ERR_ARG, "rsr_batch_check: null batch");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(static_cast<hipStream_t>(hip_stream)));
  int err[2] = {0, 0};
  if (b->sched) HIPCHK(hipMemcpy(err, b->sched + 2, sizeof(err), hipMemcpyDeviceToHost));
  if (handoff_timeouts) *handoff_timeouts = err[0];
  if (err[0] > 0)
    return fail(RSR_ERR_HANDOFF, "rsr_batch_check: " + std::to_string(err[0]) + " work-unit hand-off wait(s) timed out since the batch was created (last: env " +
                                     std::to_string(err[1]) + "); those envs' stats[3] read -1 for the step concerned and their state is not to be trusted");
  return RSR_OK;
}

extern "C" int rsr_batch_set_debug(rsr_batch* b, float* dev_buffer) {
  if (!b) return fail(RSR_ERR_ARG, "rsr_batch_set_debug: null batch");
  b->debug = dev_buffer;
  return RSR_OK;
}

static rsr::StepArgs make_args(rsr_batch* b) {
  rsr::StepArgs a{};
  a.state = b->state; a.n = b->n;
  a.dr_geom_friction = b->dr_fric; a.dr_body_mass = b->dr_mass; a.dr_dof_damping = b->dr_damp; a.dr_dof_frictionloss = b->dr_floss;
  a.dr_body_ipos = b->dr_ex[0]; a.dr_qpos0 = b->dr_ex[1]; a.dr_dof_armature = b->dr_ex[2]; a.dr_gainprm = b->dr_ex[3]; a.dr_biasprm = b->dr_ex[4];
  a.debug = b->debug;
  // default: waves that start together take turns; a joystick batch of more than one resident round lets its late starters catch up,
  // except when they are few (up to 5/8 of a round beyond the first: the slots the oldest-first arbitration frees early are worth more
  // than an even finish there -- 5120 / 6144 envs on 4096 slots lose 3.5 / 2 % under either policy, 7168 gain 9 %: DESIGN.md 4)
  const bool rotate = b->model->dims.env_kind == rsr::ENV_GO2_HANDSTAND || b->n <= b->prio_slots;
  const bool few_late = !rotate && (long long)b->n * 8 < (long long)b->prio_slots * 13;
  a.prio_mode = b->prio_policy >= 0 ? b->prio_policy : (rotate ? rsr::RSR_PRIO_ROTATE : (few_late ? rsr::RSR_PRIO_OFF : rsr::RSR_PRIO_CATCH_UP));
  a.prio_slots = b->prio_slots;
  return a;
}

extern "C" int rsr_reset(rsr_batch* b, const uint32_t* keys, void* hip_stream) {
  if (!b || !keys) return fail(RSR_ERR_ARG, "rsr_reset: null argument");
  HIPCHK(hipSetDevice(b->device));
  rsr::StepArgs a = make_args(b);
  a.keys = keys;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  if (b->model->dims.env_kind == rsr::ENV_GO2)
    rsr::launch_go2_reset(b->n, st, b->dmodel, b->model->layout, a, b->model->has_hfield);
  else if (b->model->dims.env_kind == rsr::ENV_GO2_HANDSTAND)
    rsr::launch_hs_reset(b->n, st, b->dmodel, b->model->layout, a);
  else if (b->model->dims.env_kind == rsr::ENV_TSHAPE)
    rsr::launch_tshape_reset(b->n, st, b->dmodel, b->model->layout, a);
  else
    hipLaunchKernelGGL((rsr::reset_kernel<rsr::CubeDims, rsr::ENV_CUBE>), dim3(b->n), dim3(64), sizeof(rsr::Smem<rsr::CubeDims>), st,
                       b->dmodel, b->model->layout, a);
  HIPCHK(hipGetLastError());
  return RSR_OK;
}

extern "C" int rsr_step(rsr_batch* b, const float* action, void* hip_stream) {
  if (!b || !action) return fail(RSR_ERR_ARG, "rsr_step: null argument");
  HIPCHK(hipSetDevice(b->device));
  rsr::StepArgs a = make_args(b);
  a.action = action;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const int repeat = b->action_repeat;
  const Layout& LY = b->model->layout;
  const rsr_dims& dd = b->model->dims;
  const DModel* dmodel = repeat > 1 ? b->dmodel_plain : b->dmodel;      // (repeat > 1: plain env.step, the wrappers around the repeats)
  if (repeat > 1)
    hipLaunchKernelGGL(rsr::repeat_pre_kernel, dim3(b->n), dim3(64), 0, st, b->state, LY, b->n, b->dm.wrap_flags, b->racc);
  for (int rep = 0; rep < repeat; ++rep) {
  if (b->model->dims.env_kind == rsr::ENV_GO2)
    rsr::launch_go2_step(b->n, st, dmodel, b->model->layout, a, b->model->has_hfield);
  else if (b->model->dims.env_kind == rsr::ENV_GO2_HANDSTAND)
    rsr::launch_hs_step(b->n, st, dmodel, b->model->layout, a);
  else {
    ++b->launch_id;
    if ((b->launch_id & 0xFFFFFFu) == 0u) {        // the flags carry 24 bits of the launch number: clear them before the number repeats
      ++b->launch_id;
      HIPCHK(hipMemsetAsync(b->sched + 4, 0, (size_t)b->n * sizeof(int), st));
    }
    // envs stepped as one unit: by default all but two resident rounds' worth (the launch then still drains in short units, with
    // a slack of two resident rounds between the phases of a split env); a batch that fits the resident waves is not split at all
    // (every env has a wave to itself from the start: phases would only add hand-offs; 1024 envs 3.64 -> 4.48 M env-steps/s)
    int n_whole = b->units <= 1 ? b->n : (b->whole_envs >= 0 ? b->whole_envs : (b->n <= b->step_grid ? b->n : (b->n > 2 * b->step_grid ? b->n - 2 * b->step_grid : 0)));
    if (n_whole > b->n) n_whole = b->n;
    rsr::Sched sc{b->sched, b->sched + 2, reinterpret_cast<unsigned*>(b->sched + 4), b->launch_id, b->units, n_whole, b->spin_cap, b->withhold_env};
    const long long work = (long long)n_whole + (long long)b->units * (b->n - n_whole);
    const int grid = (int)(work > b->step_grid ? b->step_grid : work);
    if (b->model->dims.env_kind == rsr::ENV_TSHAPE)
      rsr::launch_tshape_step(grid, st, dmodel, b->model->layout, a, sc);
    else
      hipLaunchKernelGGL((rsr::step_kernel<rsr::CubeDims, rsr::ENV_CUBE>), dim3(grid), dim3(64), sizeof(rsr::Smem<rsr::CubeDims>), st,
                         dmodel, b->model->layout, a, sc);
  }
  if (repeat > 1)
    hipLaunchKernelGGL(rsr::repeat_acc_kernel, dim3((b->n + 255) / 256), dim3(256), 0, st, b->state, LY, b->n, b->racc);
  }
  if (repeat > 1) {
    const bool go2_family = dd.env_kind == rsr::ENV_GO2 || dd.env_kind == rsr::ENV_GO2_HANDSTAND;
    hipLaunchKernelGGL(rsr::repeat_post_kernel, dim3(b->n), dim3(64), 0, st, b->state, LY, b->n, b->dm.wrap_flags, repeat, b->dm.episode_length,
                       dd.nmetrics, dd.obs_dim, go2_family ? rsr::GO2_PRIV : 0, dd.env_kind == rsr::ENV_GO2 ? (int)rsr::G2_XFRC : -1, b->racc);
  }
  HIPCHK(hipGetLastError());
  if (b->timing) b->launches++;
  return RSR_OK;
}

extern "C" int rsr_view(rsr_batch* b, int field_id, void** dev_ptr, int64_t shape[2], int64_t stride[2]) {
  if (!b || !dev_ptr || !shape || !stride) return fail(RSR_ERR_ARG, "rsr_view: null argument");
  const Layout& L = b->model->layout;
  const rsr_dims& d = b->model->dims;
  int off = -1, w = 0;
  switch (field_id) {
    case RSR_F_QPOS: off = L.qpos; w = d.nq; break;
    case RSR_F_QVEL: off = L.qvel; w = d.nv; break;
    case RSR_F_CTRL: off = L.ctrl; w = d.nu; break;
    case RSR_F_QACC_WARMSTART: off = L.warm; w = d.nv; break;
    case RSR_F_TIME: off = L.time; w = 1; break;
    case RSR_F_XPOS: off = L.xpos; w = d.nbody * 3; break;
    case RSR_F_SITE_XPOS: off = L.site_xpos; w = d.nsite * 3; break;
    case RSR_F_OBS: off = L.obs; w = d.obs_dim; break;
    case RSR_F_REWARD: off = L.reward; w = 1; break;
    case RSR_F_DONE: off = L.done; w = 1; break;
    case RSR_F_METRICS: off = L.metrics; w = d.nmetrics; break;
    case RSR_F_INFO_TARGET_POS: off = L.target_pos; w = 3; break;
    case RSR_F_INFO_NEW_CUBE_POS: off = L.new_cube_pos; w = 2; break;
    case RSR_F_INFO_SITE_POS: off = L.site_pos; w = 3; break;
    case RSR_F_INFO_CUBE_POS: off = L.cube_pos; w = 3; break;
    case RSR_F_INFO_LAST_ACTION: off = L.last_action; w = 1; break;
    case RSR_F_INFO_TARGET_BASE_POS: off = L.target_base_pos; w = 3; break;
    case RSR_F_INFO_TARGET_VERTICAL_POS: off = L.target_vertical_pos; w = 3; break;
    case RSR_F_INFO_TARGET_W: off = L.target_w; w = 1; break;
    case RSR_F_INFO_NEW_T_POS: off = L.new_T_pos; w = 2; break;
    case RSR_F_INFO_T_POS: off = L.T_pos; w = 3; break;
    case RSR_F_INFO_XITA: off = L.xita; w = 1; break;
    case RSR_F_INFO_GO2: off = L.go2_info; w = (d.env_kind == rsr::ENV_GO2 || d.env_kind == rsr::ENV_GO2_HANDSTAND) ? 144 : 0; break;
    case RSR_F_INFO_STEPS: off = L.steps; w = 1; break;
    case RSR_F_INFO_TRUNCATION: off = L.truncation; w = 1; break;
    case RSR_F_INFO_EPISODE_DONE: off = L.episode_done; w = 1; break;
    case RSR_F_INFO_EPISODE_METRICS: off = L.episode_metrics; w = 2 + d.nmetrics; break;
    case RSR_F_FIRST_QPOS: off = L.f_qpos; w = d.nq; break;
    case RSR_F_FIRST_QVEL: off = L.f_qvel; w = d.nv; break;
    case RSR_F_FIRST_CTRL: off = L.f_ctrl; w = d.nu; break;
    case RSR_F_FIRST_WARMSTART: off = L.f_warm; w = d.nv; break;
    case RSR_F_FIRST_TIME: off = L.f_time; w = 1; break;
    case RSR_F_FIRST_XPOS: off = L.f_xpos; w = d.nbody * 3; break;
    case RSR_F_FIRST_SITE_XPOS: off = L.f_site_xpos; w = d.nsite * 3; break;
    case RSR_F_FIRST_OBS: off = L.f_obs; w = d.obs_dim; break;
    case RSR_F_PRIVILEGED_OBS: off = L.priv_obs; w = (d.env_kind == rsr::ENV_GO2 || d.env_kind == rsr::ENV_GO2_HANDSTAND) ? 123 : 0; break;
    case RSR_F_FIRST_PRIVILEGED_OBS: off = L.f_priv_obs; w = (d.env_kind == rsr::ENV_GO2 || d.env_kind == rsr::ENV_GO2_HANDSTAND) ? 123 : 0; break;
    case RSR_F_STATS: off = L.stats; w = 4; break;
    default: return fail(RSR_ERR_ARG, "rsr_view: unknown field id");
  }
  *dev_ptr = b->state + off;
  shape[0] = b->n; shape[1] = w;
  stride[0] = L.rec; stride[1] = 1;
  return RSR_OK;
}

extern "C" int rsr_rollout_metrics(rsr_batch* b, float* dev_out, void* hip_stream) {
  if (!b || !dev_out) return fail(RSR_ERR_ARG, "rsr_rollout_metrics: null argument");
  HIPCHK(hipSetDevice(b->device));
  hipLaunchKernelGGL(rsr::rollout_metrics_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(hip_stream), b->state, b->model->layout, b->n, dev_out);
  HIPCHK(hipGetLastError());
  return RSR_OK;
}

extern "C" int rsr_timing_begin(rsr_batch* b, void* hip_stream) {
  if (!b) return fail(RSR_ERR_ARG, "rsr_timing_begin: null batch");
  HIPCHK(hipSetDevice(b->device));
  if (!b->ev0) { HIPCHK(hipEventCreate(&b->ev0)); HIPCHK(hipEventCreate(&b->ev1)); }
  b->launches = 0; b->timing = true;
  HIPCHK(hipEventRecord(b->ev0, static_cast<hipStream_t>(hip_stream)));
  return RSR_OK;
}

extern "C" int rsr_timing_end(rsr_batch* b, void* hip_stream, float* total_ms, int* launches) {
  if (!b || !b->timing || !total_ms || !launches) return fail(RSR_ERR_ARG, "rsr_timing_end: bad argument / timing not begun");
  HIPCHK(hipEventRecord(b->ev1, static_cast<hipStream_t>(hip_stream)));
  HIPCHK(hipEventSynchronize(b->ev1));
  HIPCHK(hipEventElapsedTime(total_ms, b->ev0, b->ev1));
  *launches = b->launches;
  b->timing = false;
  return RSR_OK;
}
#endif  // the host unit
