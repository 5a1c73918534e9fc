/*
 * rsr_oracle.c -- CPU restatement of the reference's env hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * library; the product (rsr_mjx_amd/) never links, imports or calls it.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in third-party packages that are absent from
 * the reference tree and from this image (mujoco-mjx 3.2.4, brax 0.12.1, jax 0.4.29, pinned in
 * reference README.md:40-58); the reference holds no golden vectors for it.
 * This file restates the published MJX algorithm stage by stage and the
 * reference's own env code line by line:
 *   - env prologue / epilogue / obs : ppo_train/airbot_training/cube_env.py:145-229
 *   - env reset                     : ppo_train/airbot_training/cube_env.py:95-143
 *   - wrappers (Episode/AutoReset)  : called at RSR/train.py:224-229 (brax.envs.training.wrap)
 *   - PRNG (threefry2x32)           : jax.random, call sites cube_env.py:97-118
 * What IS pinned: threefry known answers (Random123 / JAX test vectors), the solver against an
 * independent fp64 minimiser, dynamics invariants -- see tests/.
 *
 * Build: see oracle/Makefile.  `-DRSR_REAL=double` gives the fp64 build used by invariant tests.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef RSR_REAL
#define RSR_REAL float
#endif
typedef RSR_REAL real;

#define MINVAL ((real)1e-15)
#define MINIMP ((real)0.0001)
#define MAXIMP ((real)0.9999)

enum { JNT_FREE = 0, JNT_BALL = 1, JNT_SLIDE = 2, JNT_HINGE = 3 };
enum { PAIR_PLANE_BOX = 0, PAIR_BOX_BOX = 1, PAIR_PLANE_SPHERE = 2, PAIR_HFIELD_SPHERE = 3, PAIR_PLANE_CAPSULE = 4, PAIR_PLANE_CYLINDER = 5 };
enum { INT_EULER = 0, INT_IMPLICITFAST = 3 };
enum { ENV_CUBE = 0, ENV_TSHAPE = 1, ENV_AIRBOT_SF = 2, ENV_GO2 = 3, ENV_GO2_HANDSTAND = 4 };

#define NQ_MAX 32
#define NV_MAX 24
#define NU_MAX 12
#define NBODY_MAX 16
#define NJNT_MAX 16
#define NGEOM_MAX 48
#define NSITE_MAX 8
#define NKEY_MAX 8
#define NPAIR_MAX 64
#define NCON_MAX (NPAIR_MAX * 4)
#define NEFC_MAX (1 + 2 * NV_MAX + 6 * NCON_MAX)
#define OBS_MAX 64
#define GO2_INFO 144
#define GO2_PRIV 123   /* obs['privileged_state'] of the Go2 joystick env (joystick.py:341-366) */

/* ------------------------------------------------------------------ blob */
typedef struct { char name[40]; int32_t dtype, count, offset, reserved; } blob_entry;

static const void *blob_find(const void *blob, const char *name, int *count) {
  const int32_t *h = (const int32_t *)blob;
  int n = h[2];
  const blob_entry *e = (const blob_entry *)((const char *)blob + 16);
  for (int i = 0; i < n; i++)
    if (strncmp(e[i].name, name, 40) == 0) {
      if (count) *count = e[i].count;
      return (const char *)blob + e[i].offset;
    }
  if (count) *count = 0;
  return NULL;
}

/* ------------------------------------------------------------------ model */
typedef struct {
  int nq, nv, nu, nbody, njnt, ngeom, nsite, neq, npair;
  const int *body_parentid, *body_rootid, *body_weldid, *body_jntnum, *body_jntadr, *body_dofnum, *body_dofadr;
  const float *body_pos, *body_quat, *body_ipos, *body_iquat, *body_mass, *body_inertia, *body_invweight0;
  const int *jnt_type, *jnt_qposadr, *jnt_dofadr, *jnt_bodyid, *jnt_limited, *jnt_actfrclimited;
  const float *jnt_pos, *jnt_axis, *jnt_range, *jnt_actfrcrange, *jnt_solref, *jnt_solimp, *jnt_margin;
  const int *dof_bodyid, *dof_jntid, *dof_parentid;
  const float *dof_armature, *dof_damping, *dof_frictionloss, *dof_invweight0, *dof_solref, *dof_solimp;
  const int *geom_type, *geom_bodyid, *geom_priority;
  const float *geom_size, *geom_pos, *geom_quat, *geom_friction;
  const int *site_bodyid;
  const float *site_pos, *site_quat;
  const int *eq_obj1id, *eq_obj2id, *eq_active0;
  const float *eq_data, *eq_solref, *eq_solimp;
  const int *actuator_trnid, *actuator_ctrllimited, *actuator_forcelimited;
  const float *actuator_gear, *actuator_gainprm, *actuator_biasprm, *actuator_ctrlrange, *actuator_forcerange;
  const int *pair_geom1, *pair_geom2, *pair_kind, *pair_condim;
  const float *pair_solref, *pair_solimp, *pair_margin, *pair_gap;
  const float *qpos0;
  const float *hfield_size, *hfield_data;   /* one height field at most: size (x, y, z, base), data [nrow*ncol] in [0, 1] */
  const int *hfield_nrow, *hfield_ncol;
  real timestep, gravity[3], tolerance, ls_tolerance, impratio, meaninertia;
  int iterations, ls_iterations, integrator, disable_eulerdamp, disable_refsafe;
  /* env */
  int env_kind, n_frames, episode_length, wrap_flags, obs_dim, nmetrics;
  const int *env_ids;
  const float *env_action_scale, *env_ctrl_lo, *env_ctrl_hi, *env_reset, *env_reward;
  const float *env_go2f, *env_go2_scales, *env_go2_home, *env_go2_soft;
  const int *env_go2i;
  void *blob_copy;
} omodel;

#define F(name) m->name = (const float *)blob_find(b, #name, NULL)
#define I(name) m->name = (const int *)blob_find(b, #name, NULL)

omodel *oracle_model_create(const void *blob, int nbytes) {
  omodel *m = (omodel *)calloc(1, sizeof(omodel));
  void *b = malloc((size_t)nbytes);
  memcpy(b, blob, (size_t)nbytes);
  m->blob_copy = b;
  const int *dims = (const int *)blob_find(b, "dims", NULL);
  m->nq = dims[0]; m->nv = dims[1]; m->nu = dims[2]; m->nbody = dims[3]; m->njnt = dims[4];
  m->ngeom = dims[5]; m->nsite = dims[6]; m->neq = dims[7]; m->npair = dims[8];
  if (m->nq > NQ_MAX || m->nv > NV_MAX || m->nu > NU_MAX || m->nbody > NBODY_MAX || m->njnt > NJNT_MAX ||
      m->ngeom > NGEOM_MAX || m->nsite > NSITE_MAX || m->npair > NPAIR_MAX) {
    free(b); free(m); return NULL;
  }
  I(body_parentid); I(body_rootid); I(body_weldid); I(body_jntnum); I(body_jntadr); I(body_dofnum); I(body_dofadr);
  F(body_pos); F(body_quat); F(body_ipos); F(body_iquat); F(body_mass); F(body_inertia); F(body_invweight0);
  I(jnt_type); I(jnt_qposadr); I(jnt_dofadr); I(jnt_bodyid); I(jnt_limited); I(jnt_actfrclimited);
  F(jnt_pos); F(jnt_axis); F(jnt_range); F(jnt_actfrcrange); F(jnt_solref); F(jnt_solimp); F(jnt_margin);
  I(dof_bodyid); I(dof_jntid); I(dof_parentid);
  F(dof_armature); F(dof_damping); F(dof_frictionloss); F(dof_invweight0); F(dof_solref); F(dof_solimp);
  I(geom_type); I(geom_bodyid); I(geom_priority); F(geom_size); F(geom_pos); F(geom_quat); F(geom_friction);
  I(site_bodyid); F(site_pos); F(site_quat);
  I(eq_obj1id); I(eq_obj2id); I(eq_active0); F(eq_data); F(eq_solref); F(eq_solimp);
  I(actuator_trnid); I(actuator_ctrllimited); I(actuator_forcelimited);
  F(actuator_gear); F(actuator_gainprm); F(actuator_biasprm); F(actuator_ctrlrange); F(actuator_forcerange);
  I(pair_geom1); I(pair_geom2); I(pair_kind); I(pair_condim);
  F(pair_solref); F(pair_solimp); F(pair_margin); F(pair_gap);
  F(qpos0); F(hfield_size); F(hfield_data); I(hfield_nrow); I(hfield_ncol);
  m->timestep = ((const float *)blob_find(b, "opt_timestep", NULL))[0];
  for (int i = 0; i < 3; i++) m->gravity[i] = ((const float *)blob_find(b, "opt_gravity", NULL))[i];
  m->tolerance = ((const float *)blob_find(b, "opt_tolerance", NULL))[0];
  m->ls_tolerance = ((const float *)blob_find(b, "opt_ls_tolerance", NULL))[0];
  m->impratio = ((const float *)blob_find(b, "opt_impratio", NULL))[0];
  m->meaninertia = ((const float *)blob_find(b, "stat_meaninertia", NULL))[0];
  m->iterations = ((const int *)blob_find(b, "opt_iterations", NULL))[0];
  m->ls_iterations = ((const int *)blob_find(b, "opt_ls_iterations", NULL))[0];
  m->integrator = ((const int *)blob_find(b, "opt_integrator", NULL))[0];
  m->disable_eulerdamp = ((const int *)blob_find(b, "opt_disable_eulerdamp", NULL))[0];
  m->disable_refsafe = ((const int *)blob_find(b, "opt_disable_refsafe", NULL))[0];
  const int *ei = (const int *)blob_find(b, "env_int", NULL);
  if (ei) {
    m->env_kind = ei[0]; m->n_frames = ei[1]; m->episode_length = ei[2]; m->wrap_flags = ei[3];
    m->obs_dim = ei[4]; m->nmetrics = ei[5];
  }
  I(env_ids); F(env_action_scale); F(env_ctrl_lo); F(env_ctrl_hi); F(env_reset); F(env_reward);
  F(env_go2f); F(env_go2_scales); F(env_go2_home); F(env_go2_soft); I(env_go2i);
  return m;
}
#undef F
#undef I

void oracle_model_destroy(omodel *m) { if (m) { free(m->blob_copy); free(m); } }

/* ------------------------------------------------------------------ small math */
static inline void v3set(real *o, real a, real b, real c) { o[0] = a; o[1] = b; o[2] = c; }
static inline void v3copy(real *o, const real *a) { o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; }
static inline real v3dot(const real *a, const real *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void v3cross(real *o, const real *a, const real *b) {
  real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
static inline real rsqrt_(real x) { return (real)sqrt((double)x); }
static inline real v3norm(const real *a) { return rsqrt_(v3dot(a, a)); }
static inline void quat_mul(real *o, const real *a, const real *b) {
  real w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  real x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  real y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  real z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}
static inline void quat_to_mat(real *m, const real *q) {
  real w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
/* rotate vector by quaternion (through the rotation matrix, as MJX math.rotate expands to) */
static inline void rotate(real *o, const real *v, const real *q) {
  real m[9]; quat_to_mat(m, q);
  real x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  real y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  real z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static inline void mat_mulv(real *o, const real *m, const real *v) {
  real x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  real y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  real z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static inline void mat_tmulv(real *o, const real *m, const real *v) {
  real x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
  real y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
  real z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static inline void quat_normalize(real *q) {
  real n = rsqrt_(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  real inv = 1 / n; q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
}
static inline void axis_angle_to_quat(real *q, const real *axis, real angle) {
  real s = (real)sin((double)(angle * (real)0.5)), c = (real)cos((double)(angle * (real)0.5));
  q[0] = c; q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
static inline real clampr(real x, real lo, real hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* ------------------------------------------------------------------ per-env transient data */
typedef struct {
  real dist, pos[3], frame[9];
  real friction[5], solref[2], solimp[5], includemargin;
  int body1, body2, condim, pair;
} ocontact;

typedef struct {
  /* state */
  real qpos[NQ_MAX], qvel[NV_MAX], ctrl[NU_MAX], qacc_warmstart[NV_MAX], time;
  /* per-env model overrides (domain randomisation) */
  real geom_friction[NGEOM_MAX * 3], body_mass[NBODY_MAX], dof_damping[NV_MAX], dof_frictionloss[NV_MAX];
  /* the further leaves the Go2 randomisation makes per-env (go2/randomize.py:6-109) */
  real body_ipos[NBODY_MAX * 3], qpos0[NQ_MAX], dof_armature[NV_MAX], gainprm[NU_MAX * 3], biasprm[NU_MAX * 3];
  /* position stage */
  real xpos[NBODY_MAX * 3], xquat[NBODY_MAX * 4], xmat[NBODY_MAX * 9], xipos[NBODY_MAX * 3], ximat[NBODY_MAX * 9];
  real xanchor[NJNT_MAX * 3], xaxis[NJNT_MAX * 3];
  real geom_xpos[NGEOM_MAX * 3], geom_xmat[NGEOM_MAX * 9], site_xpos[NSITE_MAX * 3];
  real site_xmat[NSITE_MAX * 9], site_linvel[NSITE_MAX * 3], site_angvel[NSITE_MAX * 3];   /* sensor sources (Go2) */
  real actuator_force[NU_MAX];
  real subtree_com[NBODY_MAX * 3], cinert[NBODY_MAX * 10], crb[NBODY_MAX * 10], cdof[NV_MAX * 6];
  real M[NV_MAX * NV_MAX], L[NV_MAX * NV_MAX];
  /* velocity stage */
  real cvel[NBODY_MAX * 6], cdof_dot[NV_MAX * 6];
  int xfrc_body; real xfrc_force[3];     /* data.xfrc_applied: one body, force only (the Go2 perturbation kick); body <= 0: none */
  real acc_site[3]; int acc_site_id;      /* accelerometer reading of site acc_site_id (< 0: not computed), from the last forward pass */
  real qfrc_bias[NV_MAX], qfrc_passive[NV_MAX], qfrc_actuator[NV_MAX], qfrc_smooth[NV_MAX], qacc_smooth[NV_MAX];
  /* contacts + constraint rows */
  int ncon, nefc, ne, nf, ncon_overflow;
  ocontact con[NCON_MAX];
  real *efc_J;               /* [NEFC_MAX][nv] */
  real efc_pos[NEFC_MAX], efc_aref[NEFC_MAX], efc_D[NEFC_MAX], efc_R[NEFC_MAX], efc_floss[NEFC_MAX];
  real efc_force[NEFC_MAX];
  /* solver output */
  real qacc[NV_MAX], qfrc_constraint[NV_MAX];
  int solver_niter, ls_total;
} odata;

static odata *odata_alloc(void) {
  odata *d = (odata *)calloc(1, sizeof(odata));
  d->efc_J = (real *)calloc((size_t)NEFC_MAX * NV_MAX, sizeof(real));
  d->acc_site_id = -1;
  return d;
}
static void odata_free(odata *d) { if (d) { free(d->efc_J); free(d); } }

/* global knobs (tests flip them) */
static int g_cull = 1;        /* 1: drop inactive limit rows / separated contacts (result-neutral) */
static int g_ncon_cap = NCON_MAX;
/* Line-search stop rule.  0: MJX's (gradient below gtol, no bracket update, or ls_iterations).  1: MJX's plus the HIP kernel's
 * fp32 noise-floor stop for converging solves (iterations > 1): a bracket end also counts as converged when |derivative| is
 * below g_ls_noise * eps * (sum |linear terms| + 2 |alpha| sum |quadratic terms|), the rounding noise of the derivative sum.
 * 2: as 1, and the sign of the derivative is not asked for (|d| below the floor at either end stops the search); like 1 it
 * leaves single-iteration solves (Go2) on MJX's rule. */
/* g_ls_cycle: exact shortcut of the search's limit cycles.  The bracket update can enter a cycle (a Newton step from `lo`
 * that overshoots is accepted as the new `lo` although its derivative is positive; the roles of the two ends then keep
 * flipping) that only the iteration cap ends.  The loop state is (lo.alpha, hi.alpha) -- every other field is a function of
 * alpha -- so once that pair repeats bit for bit with period P the remaining iterations are known: the search runs
 * (cap - it) mod P more iterations and stops in exactly the state the cap would have left.  Results are bit-identical. */
static int g_ls_cycle = 0;
void oracle_set_ls_cycle(int on) { g_ls_cycle = on; }
static int g_ls_rule = 0;
static double g_ls_noise = 1.0;
static long long g_ls_calls = 0, g_ls_iters = 0, g_ls_hist[64];
void oracle_set_ls_rule(int rule, double noise_eps) { g_ls_rule = rule; g_ls_noise = noise_eps; }
void oracle_ls_counters(long long *out /* calls, iterations, histogram[52] of iterations per call */, int reset) {
  if (out) { out[0] = g_ls_calls; out[1] = g_ls_iters; for (int i = 0; i < 52; i++) out[2 + i] = g_ls_hist[i]; }
  if (reset) { g_ls_calls = 0; g_ls_iters = 0; memset(g_ls_hist, 0, sizeof(g_ls_hist)); }
}
void oracle_set_cull(int c) { g_cull = c; }
void oracle_set_ncon_cap(int c) { g_ncon_cap = c < NCON_MAX ? c : NCON_MAX; }

/* ------------------------------------------------------------------ stage 1: kinematics (Appendix B.1) */
static void kinematics(const omodel *m, odata *d) {
  v3set(d->xpos, 0, 0, 0);
  d->xquat[0] = 1; d->xquat[1] = d->xquat[2] = d->xquat[3] = 0;
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parentid[b];
    real bp[3] = {m->body_pos[3 * b], m->body_pos[3 * b + 1], m->body_pos[3 * b + 2]};
    real bq[4] = {m->body_quat[4 * b], m->body_quat[4 * b + 1], m->body_quat[4 * b + 2], m->body_quat[4 * b + 3]};
    real pos[3], quat[4], t[3];
    rotate(t, bp, &d->xquat[4 * p]);
    for (int k = 0; k < 3; k++) pos[k] = d->xpos[3 * p + k] + t[k];
    quat_mul(quat, &d->xquat[4 * p], bq);
    for (int k = 0; k < m->body_jntnum[b]; k++) {
      int j = m->body_jntadr[b] + k, qa = m->jnt_qposadr[j];
      real jp_[3] = {m->jnt_pos[3 * j], m->jnt_pos[3 * j + 1], m->jnt_pos[3 * j + 2]};
      real ja[3] = {m->jnt_axis[3 * j], m->jnt_axis[3 * j + 1], m->jnt_axis[3 * j + 2]};
      if (m->jnt_type[j] == JNT_FREE) {
        v3copy(&d->xanchor[3 * j], &d->qpos[qa]);
        v3set(&d->xaxis[3 * j], 0, 0, 1);
        v3copy(pos, &d->qpos[qa]);
        for (int c = 0; c < 4; c++) quat[c] = d->qpos[qa + 3 + c];
        quat_normalize(quat);
        for (int c = 0; c < 4; c++) d->qpos[qa + 3 + c] = quat[c];   /* MJX writes the normalised quat back */
      } else {
        real anchor[3], axis[3];
        rotate(anchor, jp_, quat);
        for (int c = 0; c < 3; c++) anchor[c] += pos[c];
        rotate(axis, ja, quat);
        v3copy(&d->xanchor[3 * j], anchor);
        v3copy(&d->xaxis[3 * j], axis);
        if (m->jnt_type[j] == JNT_HINGE) {
          real qloc[4], q2[4], r[3];
          axis_angle_to_quat(qloc, ja, d->qpos[qa] - d->qpos0[qa]);
          quat_mul(q2, quat, qloc);
          for (int c = 0; c < 4; c++) quat[c] = q2[c];
          rotate(r, jp_, quat);
          for (int c = 0; c < 3; c++) pos[c] = anchor[c] - r[c];
        } else { /* slide */
          real dq = d->qpos[qa] - d->qpos0[qa];
          for (int c = 0; c < 3; c++) pos[c] += axis[c] * dq;
        }
      }
    }
    v3copy(&d->xpos[3 * b], pos);
    for (int c = 0; c < 4; c++) d->xquat[4 * b + c] = quat[c];
  }
  for (int b = 0; b < m->nbody; b++) {
    quat_to_mat(&d->xmat[9 * b], &d->xquat[4 * b]);
    real ip[3] = {d->body_ipos[3 * b], d->body_ipos[3 * b + 1], d->body_ipos[3 * b + 2]}, t[3], q[4];
    real iq[4] = {m->body_iquat[4 * b], m->body_iquat[4 * b + 1], m->body_iquat[4 * b + 2], m->body_iquat[4 * b + 3]};
    mat_mulv(t, &d->xmat[9 * b], ip);
    for (int c = 0; c < 3; c++) d->xipos[3 * b + c] = d->xpos[3 * b + c] + t[c];
    quat_mul(q, &d->xquat[4 * b], iq);
    quat_to_mat(&d->ximat[9 * b], q);
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_bodyid[g];
    real gp[3] = {m->geom_pos[3 * g], m->geom_pos[3 * g + 1], m->geom_pos[3 * g + 2]}, t[3], q[4];
    real gq[4] = {m->geom_quat[4 * g], m->geom_quat[4 * g + 1], m->geom_quat[4 * g + 2], m->geom_quat[4 * g + 3]};
    mat_mulv(t, &d->xmat[9 * b], gp);
    for (int c = 0; c < 3; c++) d->geom_xpos[3 * g + c] = d->xpos[3 * b + c] + t[c];
    quat_mul(q, &d->xquat[4 * b], gq);
    quat_to_mat(&d->geom_xmat[9 * g], q);
  }
  for (int s = 0; s < m->nsite; s++) {
    int b = m->site_bodyid[s];
    real sp[3] = {m->site_pos[3 * s], m->site_pos[3 * s + 1], m->site_pos[3 * s + 2]}, t[3];
    mat_mulv(t, &d->xmat[9 * b], sp);
    for (int c = 0; c < 3; c++) d->site_xpos[3 * s + c] = d->xpos[3 * b + c] + t[c];
    real sq[4] = {m->site_quat[4 * s], m->site_quat[4 * s + 1], m->site_quat[4 * s + 2], m->site_quat[4 * s + 3]}, q[4];
    quat_mul(q, &d->xquat[4 * b], sq);
    quat_to_mat(&d->site_xmat[9 * s], q);
  }
}

/* ------------------------------------------------------------------ stage 2: com_pos (Appendix B.2) */
static void com_pos(const omodel *m, odata *d) {
  real pos[NBODY_MAX * 3], mass[NBODY_MAX];
  for (int b = 0; b < m->nbody; b++) {
    mass[b] = d->body_mass[b];
    for (int c = 0; c < 3; c++) pos[3 * b + c] = d->xipos[3 * b + c] * d->body_mass[b];
  }
  for (int b = m->nbody - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    for (int c = 0; c < 3; c++) pos[3 * p + c] += pos[3 * b + c];
    mass[p] += mass[b];
  }
  for (int b = 0; b < m->nbody; b++)
    for (int c = 0; c < 3; c++)
      d->subtree_com[3 * b + c] = mass[b] < MINVAL ? d->xipos[3 * b + c] : pos[3 * b + c] / mass[b];
  for (int b = 0; b < m->nbody; b++) {
    const real *R = &d->ximat[9 * b];
    const real *rc = &d->subtree_com[3 * m->body_rootid[b]];
    real off[3] = {d->xipos[3 * b] - rc[0], d->xipos[3 * b + 1] - rc[1], d->xipos[3 * b + 2] - rc[2]};
    real I0 = m->body_inertia[3 * b], I1 = m->body_inertia[3 * b + 1], I2 = m->body_inertia[3 * b + 2];
    real ms = d->body_mass[b];
    real in[6];
    /* (R diag(I) R^T) upper triangle: xx yy zz xy xz yz */
    in[0] = R[0] * I0 * R[0] + R[1] * I1 * R[1] + R[2] * I2 * R[2];
    in[1] = R[3] * I0 * R[3] + R[4] * I1 * R[4] + R[5] * I2 * R[5];
    in[2] = R[6] * I0 * R[6] + R[7] * I1 * R[7] + R[8] * I2 * R[8];
    in[3] = R[0] * I0 * R[3] + R[1] * I1 * R[4] + R[2] * I2 * R[5];
    in[4] = R[0] * I0 * R[6] + R[1] * I1 * R[7] + R[2] * I2 * R[8];
    in[5] = R[3] * I0 * R[6] + R[4] * I1 * R[7] + R[5] * I2 * R[8];
    /* + mass * (|off|^2 I - off off^T) */
    in[0] += ms * (off[1] * off[1] + off[2] * off[2]);
    in[1] += ms * (off[0] * off[0] + off[2] * off[2]);
    in[2] += ms * (off[0] * off[0] + off[1] * off[1]);
    in[3] -= ms * off[0] * off[1];
    in[4] -= ms * off[0] * off[2];
    in[5] -= ms * off[1] * off[2];
    real *ci = &d->cinert[10 * b];
    for (int c = 0; c < 6; c++) ci[c] = in[c];
    for (int c = 0; c < 3; c++) ci[6 + c] = ms * off[c];
    ci[9] = ms;
  }
  /* cdof: [angular(3), linear(3)] about the root's subtree com */
  for (int j = 0; j < m->njnt; j++) {
    int b = m->jnt_bodyid[j], da = m->jnt_dofadr[j];
    const real *rc = &d->subtree_com[3 * m->body_rootid[b]];
    real off[3] = {rc[0] - d->xanchor[3 * j], rc[1] - d->xanchor[3 * j + 1], rc[2] - d->xanchor[3 * j + 2]};
    if (m->jnt_type[j] == JNT_FREE) {
      for (int a = 0; a < 3; a++) {
        real *cd = &d->cdof[6 * (da + a)];
        for (int c = 0; c < 6; c++) cd[c] = 0;
        cd[3 + a] = 1;
      }
      for (int a = 0; a < 3; a++) {
        real ax[3] = {d->xmat[9 * b + a], d->xmat[9 * b + 3 + a], d->xmat[9 * b + 6 + a]};
        real *cd = &d->cdof[6 * (da + 3 + a)];
        v3copy(cd, ax);
        v3cross(cd + 3, ax, off);
      }
    } else if (m->jnt_type[j] == JNT_HINGE) {
      real *cd = &d->cdof[6 * da];
      v3copy(cd, &d->xaxis[3 * j]);
      v3cross(cd + 3, &d->xaxis[3 * j], off);
    } else {
      real *cd = &d->cdof[6 * da];
      v3set(cd, 0, 0, 0);
      v3copy(cd + 3, &d->xaxis[3 * j]);
    }
  }
}

/* spatial inertia (10-vector) times motion vector */
static inline void inert_mul(real *o, const real *i, const real *v) {
  real ang[3], t[3];
  ang[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2];
  ang[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2];
  ang[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2];
  v3cross(t, i + 6, v + 3);
  for (int c = 0; c < 3; c++) o[c] = ang[c] + t[c];
  v3cross(t, i + 6, v);
  for (int c = 0; c < 3; c++) o[3 + c] = i[9] * v[3 + c] - t[c];
}
static inline void motion_cross(real *o, const real *u, const real *v) {
  real a[3], b1[3], b2[3];
  v3cross(a, u, v);
  v3cross(b1, u + 3, v);
  v3cross(b2, u, v + 3);
  for (int c = 0; c < 3; c++) { o[c] = a[c]; o[3 + c] = b1[c] + b2[c]; }
}
static inline void motion_cross_force(real *o, const real *v, const real *f) {
  real a1[3], a2[3], b[3];
  v3cross(a1, v, f);
  v3cross(a2, v + 3, f + 3);
  v3cross(b, v, f + 3);
  for (int c = 0; c < 3; c++) { o[c] = a1[c] + a2[c]; o[3 + c] = b[c]; }
}

/* dense Cholesky A = L L^T (lower), returns 0 on success */
static int cholesky(real *L, const real *A, int n, int ld) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j <= i; j++) {
      real s = A[i * ld + j];
      for (int k = 0; k < j; k++) s -= L[i * ld + k] * L[j * ld + k];
      if (i == j) { if (!(s > 0)) s = MINVAL; L[i * ld + i] = rsqrt_(s); }
      else L[i * ld + j] = s / L[j * ld + j];
    }
  return 0;
}
static void chol_solve(const real *L, real *x, int n, int ld) {
  for (int i = 0; i < n; i++) {
    real s = x[i];
    for (int k = 0; k < i; k++) s -= L[i * ld + k] * x[k];
    x[i] = s / L[i * ld + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    real s = x[i];
    for (int k = i + 1; k < n; k++) s -= L[k * ld + i] * x[k];
    x[i] = s / L[i * ld + i];
  }
}

/* ------------------------------------------------------------------ stage 3: crb + factor_m (Appendix B.3) */
static void crb_and_factor(const omodel *m, odata *d) {
  int nv = m->nv;
  for (int b = 0; b < m->nbody; b++)
    for (int c = 0; c < 10; c++) d->crb[10 * b + c] = d->cinert[10 * b + c];
  for (int b = m->nbody - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    if (p > 0) for (int c = 0; c < 10; c++) d->crb[10 * p + c] += d->crb[10 * b + c];
  }
  for (int i = 0; i < nv * nv; i++) d->M[i] = 0;
  for (int i = 0; i < nv; i++) {
    real f[6];
    inert_mul(f, &d->crb[10 * m->dof_bodyid[i]], &d->cdof[6 * i]);
    for (int j = i; j >= 0; j = m->dof_parentid[j]) {
      real s = 0;
      for (int c = 0; c < 6; c++) s += f[c] * d->cdof[6 * j + c];
      d->M[i * nv + j] = s;
      d->M[j * nv + i] = s;
    }
    d->M[i * nv + i] += d->dof_armature[i];
  }
  cholesky(d->L, d->M, nv, nv);
}

/* ------------------------------------------------------------------ Jacobian of a world point on a body */
static void jac_point(const omodel *m, const odata *d, const real *point, int body, real *jacp, real *jacr) {
  int nv = m->nv;
  for (int i = 0; i < 3 * nv; i++) { jacp[i] = 0; if (jacr) jacr[i] = 0; }
  if (body <= 0) return;
  const real *rc = &d->subtree_com[3 * m->body_rootid[body]];
  real off[3] = {point[0] - rc[0], point[1] - rc[1], point[2] - rc[2]};
  int b = body;
  while (b > 0 && m->body_dofnum[b] == 0) b = m->body_parentid[b];
  if (b <= 0) return;
  for (int i = m->body_dofadr[b] + m->body_dofnum[b] - 1; i >= 0; i = m->dof_parentid[i]) {
    const real *cd = &d->cdof[6 * i];
    real t[3];
    v3cross(t, cd, off);
    for (int c = 0; c < 3; c++) {
      jacp[c * nv + i] = cd[3 + c] + t[c];
      if (jacr) jacr[c * nv + i] = cd[c];
    }
  }
}

/* ------------------------------------------------------------------ stage 4: collision (Appendix B.4) */
static void make_frame(real *frame, const real *n_in) {
  real a[3], b[3], c[3];
  real nn = v3norm(n_in);
  for (int k = 0; k < 3; k++) a[k] = nn > MINVAL ? n_in[k] / nn : 0;
  if (a[1] > (real)-0.5 && a[1] < (real)0.5) v3set(b, 0, 1, 0); else v3set(b, 0, 0, 1);
  real ab = v3dot(a, b);
  for (int k = 0; k < 3; k++) b[k] -= a[k] * ab;
  real bn = v3norm(b);
  for (int k = 0; k < 3; k++) b[k] = bn > MINVAL ? b[k] / bn : 0;
  v3cross(c, a, b);
  for (int k = 0; k < 3; k++) { frame[k] = a[k]; frame[3 + k] = b[k]; frame[6 + k] = c[k]; }
}

/* pick <=4 of n masked points with approximately maximal area (MJX _manifold_points), 2D version:
 * points p[i] = (x[i], y[i]) in a plane with normal +z.  Returns indices (duplicates possible). */
static void manifold_points(const real *x, const real *y, const int *mask, int n, int *idx) {
  const real NEG = (real)-1e6;
  int a = 0, b = 0, c = 0, dd = 0;
  real best;
  /* a: first valid point */
  best = NEG * 2;
  for (int i = 0; i < n; i++) { real v = mask[i] ? 0 : NEG; if (v > best) { best = v; a = i; } }
  /* b: farthest from a */
  best = NEG * 4;
  for (int i = 0; i < n; i++) {
    real dx = x[a] - x[i], dy = y[a] - y[i];
    real v = dx * dx + dy * dy + (mask[i] ? 0 : NEG);
    if (v > best) { best = v; b = i; }
  }
  /* c: farthest from line ab:  ab_perp = z x (a-b) = (-(a-b).y, (a-b).x) */
  real abx = -(y[a] - y[b]), aby = (x[a] - x[b]);
  best = NEG * 4;
  for (int i = 0; i < n; i++) {
    real v = (real)fabs((double)((x[a] - x[i]) * abx + (y[a] - y[i]) * aby)) + (mask[i] ? 0 : NEG);
    if (v > best) { best = v; c = i; }
  }
  /* d: farthest from line ab on the side opposite to c (MJX takes the point farthest from edges ac/bc,
   * which ties with a or b on exact rectangles and then yields a duplicate; this choice has no such tie).
   * If nothing lies strictly on the other side, d repeats c and is dropped by the caller as a duplicate. */
  real sc = ((x[a] - x[c]) * abx + (y[a] - y[c]) * aby) > 0 ? (real)1 : (real)-1;
  best = 0; dd = c;
  for (int i = 0; i < n; i++) {
    real v = -sc * ((x[a] - x[i]) * abx + (y[a] - y[i]) * aby);
    if (mask[i] && v > best) { best = v; dd = i; }
  }
  idx[0] = a; idx[1] = b; idx[2] = c; idx[3] = dd;
}

typedef struct { real dist, pos[3]; } cpoint;

/* plane (geom1) vs box (geom2): deepest <=4 vertices within a 1 mm skin (MJX plane_convex).
 * Returns number of penetrating contacts written; normal = plane z axis. */
static int plane_box(const real *ppos, const real *pmat, const real *bpos, const real *bmat, const real *size,
                     cpoint *out, real *normal) {
  real n[3] = {pmat[2], pmat[5], pmat[8]};
  v3copy(normal, n);
  real support[8], vx[8], vy[8], vw[8][3];
  /* 2-D coordinates of the vertices in the plane frame for the manifold heuristic */
  real ax[3] = {pmat[0], pmat[3], pmat[6]}, ay[3] = {pmat[1], pmat[4], pmat[7]};
  real smax = (real)-1e30;
  for (int v = 0; v < 8; v++) {
    real loc[3] = {(v & 4) ? size[0] : -size[0], (v & 2) ? size[1] : -size[1], (v & 1) ? size[2] : -size[2]};
    real w[3];
    mat_mulv(w, bmat, loc);
    for (int c = 0; c < 3; c++) { w[c] += bpos[c]; vw[v][c] = w[c]; }
    real rel[3] = {ppos[0] - w[0], ppos[1] - w[1], ppos[2] - w[2]};
    support[v] = v3dot(rel, n);        /* > 0 : vertex below the plane */
    vx[v] = v3dot(w, ax); vy[v] = v3dot(w, ay);
    if (support[v] > smax) smax = support[v];
  }
  if (!(smax > 0)) return 0;
  real thr = smax - (real)1e-3; if (thr < 0) thr = 0;
  int mask[8], idx[4];
  for (int v = 0; v < 8; v++) mask[v] = support[v] > thr;
  manifold_points(vx, vy, mask, 8, idx);
  int k = 0;
  for (int i = 0; i < 4; i++) {
    int dup = 0;
    for (int j = 0; j < i; j++) if (idx[j] == idx[i]) dup = 1;
    if (dup || !mask[idx[i]]) continue;
    real dist = -support[idx[i]];
    out[k].dist = dist;
    for (int c = 0; c < 3; c++) out[k].pos[c] = vw[idx[i]][c] - (real)0.5 * dist * n[c];
    k++;
  }
  return k;
}

/* sphere (geom2) vs plane (geom1) */
static int plane_sphere(const real *ppos, const real *pmat, const real *spos, real radius, cpoint *out, real *normal) {
  real n[3] = {pmat[2], pmat[5], pmat[8]};
  v3copy(normal, n);
  real rel[3] = {spos[0] - ppos[0], spos[1] - ppos[1], spos[2] - ppos[2]};
  real dist = v3dot(rel, n) - radius;
  out[0].dist = dist;
  for (int c = 0; c < 3; c++) out[0].pos[c] = spos[c] - n[c] * (radius + (real)0.5 * dist);
  return 1;
}

/* capsule (geom2) vs plane (geom1): the two end spheres of the segment, +axis end first; both contacts share a frame whose first
 * tangent follows the capsule axis projected into the plane (mujoco-mjx 3.2.x collision_primitive.plane_capsule, restated from
 * its published source: "align contact frames with capsule axis"; not in the reference tree: parity unpinned).  frame: 9 floats. */
static int plane_capsule(const real *ppos, const real *pmat, const real *cpos, const real *cmat, real radius, real halflen,
                         cpoint *out, real *normal, real *frame) {
  real n[3] = {pmat[2], pmat[5], pmat[8]}, axis[3] = {cmat[2], cmat[5], cmat[8]};
  v3copy(normal, n);
  real na = v3dot(n, axis), b[3] = {axis[0] - n[0] * na, axis[1] - n[1] * na, axis[2] - n[2] * na};
  real bn = v3norm(b);
  if (bn < (real)0.5) { if (n[1] > (real)-0.5 && n[1] < (real)0.5) v3set(b, 0, 1, 0); else v3set(b, 0, 0, 1); }
  else for (int k = 0; k < 3; k++) b[k] /= bn;
  real c[3];
  v3cross(c, n, b);
  for (int k = 0; k < 3; k++) { frame[k] = n[k]; frame[3 + k] = b[k]; frame[6 + k] = c[k]; }
  for (int i = 0; i < 2; i++) {
    real sg = i == 0 ? (real)1 : (real)-1, ctr[3];
    for (int k = 0; k < 3; k++) ctr[k] = cpos[k] + sg * axis[k] * halflen;
    real rel[3] = {ctr[0] - ppos[0], ctr[1] - ppos[1], ctr[2] - ppos[2]};
    real dist = v3dot(rel, n) - radius;
    out[i].dist = dist;
    for (int k = 0; k < 3; k++) out[i].pos[k] = ctr[k] - n[k] * (radius + (real)0.5 * dist);
  }
  return 2;
}

/* cylinder (geom2) vs plane (geom1): three points of the rim of the disk that faces the plane -- the deepest one and two at
 * +-120 degrees (scaled sqrt(3)/2 sideways, -1/2 along) -- or, when the cylinder lies parallel to the plane, the deepest rim points
 * of both disks and one side point (mujoco-mjx 3.2.x collision_primitive.plane_cylinder, restated from its published source;
 * parity unpinned).  size = (radius, half length). */
static int plane_cylinder(const real *ppos, const real *pmat, const real *cpos, const real *cmat, real radius, real halflen,
                          cpoint *out, real *normal) {
  real n[3] = {pmat[2], pmat[5], pmat[8]}, axis[3] = {cmat[2], cmat[5], cmat[8]};
  v3copy(normal, n);
  real prjaxis = v3dot(n, axis);
  real sign = prjaxis < 0 ? (real)1 : (real)-1;              /* make the axis point toward the plane */
  for (int k = 0; k < 3; k++) axis[k] *= sign;
  prjaxis *= sign;
  real rel[3] = {cpos[0] - ppos[0], cpos[1] - ppos[1], cpos[2] - ppos[2]};
  real dist0 = v3dot(rel, n);
  real vec[3] = {axis[0] * prjaxis - n[0], axis[1] * prjaxis - n[1], axis[2] * prjaxis - n[2]};
  real len = v3norm(vec);
  if (len < (real)1e-12) for (int k = 0; k < 3; k++) vec[k] = cmat[3 * k] * radius;     /* disk parallel to the plane: the cylinder's x axis */
  else for (int k = 0; k < 3; k++) vec[k] = vec[k] / len * radius;
  real prjvec = v3dot(vec, n);
  for (int k = 0; k < 3; k++) axis[k] *= halflen;
  prjaxis *= halflen;
  real prjvec1 = -prjvec * (real)0.5, vec1[3];
  v3cross(vec1, vec, axis);
  real l1 = v3norm(vec1);
  for (int k = 0; k < 3; k++) vec1[k] = (l1 > 0 ? vec1[k] / l1 : 0) * radius * (real)sqrt(3.0) * (real)0.5;
  real d1 = dist0 + prjaxis + prjvec, d2 = dist0 + prjaxis + prjvec1;
  out[0].dist = d1; out[1].dist = d2; out[2].dist = d2;
  for (int k = 0; k < 3; k++) {
    out[0].pos[k] = cpos[k] + axis[k] + vec[k] - n[k] * d1 * (real)0.5;
    out[1].pos[k] = cpos[k] + axis[k] + vec1[k] + vec[k] * (real)-0.5 - n[k] * d2 * (real)0.5;
    out[2].pos[k] = cpos[k] + axis[k] - vec1[k] + vec[k] * (real)-0.5 - n[k] * d2 * (real)0.5;
  }
  if ((prjaxis < 0 ? -prjaxis : prjaxis) < (real)1e-3) {   /* cylinder parallel to the plane: the second point moves to the other disk */
    real d3 = dist0 - prjaxis + prjvec;
    out[1].dist = d3;
    for (int k = 0; k < 3; k++) out[1].pos[k] = cpos[k] + vec[k] - axis[k] - n[k] * d3 * (real)0.5;
  }
  return 3;
}

/* closest point of triangle abc to p (Ericson, Real-Time Collision Detection 5.1.5: Voronoi regions of the
 * vertices, edges and face) */
static void closest_on_triangle(const real *p, const real *a, const real *b, const real *c, real *q) {
  real ab[3], ac[3], ap[3], bp[3], cp[3];
  for (int k = 0; k < 3; k++) { ab[k] = b[k] - a[k]; ac[k] = c[k] - a[k]; ap[k] = p[k] - a[k]; bp[k] = p[k] - b[k]; cp[k] = p[k] - c[k]; }
  real d1 = v3dot(ab, ap), d2 = v3dot(ac, ap);
  if (d1 <= 0 && d2 <= 0) { v3copy(q, a); return; }
  real d3 = v3dot(ab, bp), d4 = v3dot(ac, bp);
  if (d3 >= 0 && d4 <= d3) { v3copy(q, b); return; }
  real vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) { real v = d1 / (d1 - d3); for (int k = 0; k < 3; k++) q[k] = a[k] + v * ab[k]; return; }
  real d5 = v3dot(ab, cp), d6 = v3dot(ac, cp);
  if (d6 >= 0 && d5 <= d6) { v3copy(q, c); return; }
  real vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) { real w = d2 / (d2 - d6); for (int k = 0; k < 3; k++) q[k] = a[k] + w * ac[k]; return; }
  real va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
    real w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    for (int k = 0; k < 3; k++) q[k] = b[k] + w * (c[k] - b[k]);
    return;
  }
  real den = (real)1 / (va + vb + vc), v = vb * den, w = vc * den;
  for (int k = 0; k < 3; k++) q[k] = a[k] + ab[k] * v + ac[k] * w;
}

/* sphere (geom2) vs height field (geom1): ONE contact at the point of the triangulated surface closest to the
 * sphere centre.  Cells are split along the diagonal (col, row) - (col+1, row+1); elevation = data * size[2].
 * MJX 3.2.4 collides the sphere with the triangular prisms of a sub-grid and keeps the deepest contact; its
 * source is not available here, so this follows the geometry (closest surface point; centre below the surface:
 * perpendicular depth to the triangle above it), not MJX's enumeration -- "parity unpinned", see DESIGN.md.
 * The sphere must not span more than two grid lines per axis (checked at model creation). */
static int hfield_sphere(const real *hpos, const real *hmat, const float *hsize, int nrow, int ncol, const float *data,
                         const real *spos, real radius, cpoint *out, real *normal) {
  real rel[3] = {spos[0] - hpos[0], spos[1] - hpos[1], spos[2] - hpos[2]}, p[3];
  for (int c = 0; c < 3; c++) p[c] = hmat[c] * rel[0] + hmat[3 + c] * rel[1] + hmat[6 + c] * rel[2];
  real sx = hsize[0], sy = hsize[1], sz = hsize[2];
  if (p[0] < -sx || p[0] > sx || p[1] < -sy || p[1] > sy) return 0;
  real dx = (real)2 * sx / (real)(ncol - 1), dy = (real)2 * sy / (real)(nrow - 1);
#define HZ(r_, c_) ((real)data[(r_) * ncol + (c_)] * sz)
  /* the triangle above/below the centre */
  int ci = (int)floor((double)((p[0] + sx) / dx)), ri = (int)floor((double)((p[1] + sy) / dy));
  ci = ci < 0 ? 0 : (ci > ncol - 2 ? ncol - 2 : ci); ri = ri < 0 ? 0 : (ri > nrow - 2 ? nrow - 2 : ri);
  real n[3], q[3], dist;
  {
    real x0 = -sx + dx * (real)ci, y0 = -sy + dy * (real)ri;
    real u = (p[0] - x0) / dx, v = (p[1] - y0) / dy;
    real z00 = HZ(ri, ci), z10 = HZ(ri, ci + 1), z01 = HZ(ri + 1, ci), z11 = HZ(ri + 1, ci + 1);
    real gx, gy, zs;     /* surface gradient and height under the centre */
    if (u >= v) { gx = (z10 - z00) / dx; gy = (z11 - z10) / dy; zs = z00 + (z10 - z00) * u + (z11 - z10) * v; }
    else { gx = (z11 - z01) / dx; gy = (z01 - z00) / dy; zs = z00 + (z11 - z01) * u + (z01 - z00) * v; }
    if (p[2] < zs) {
      real inv = (real)1 / rsqrt_(gx * gx + gy * gy + (real)1);
      n[0] = -gx * inv; n[1] = -gy * inv; n[2] = inv;
      real depth = (zs - p[2]) * inv;
      dist = -depth - radius;
      for (int k = 0; k < 3; k++) q[k] = p[k] + n[k] * depth;
      goto emit;
    }
  }
  {
    int c0 = (int)floor((double)((p[0] - radius + sx) / dx)), r0 = (int)floor((double)((p[1] - radius + sy) / dy));
    c0 = c0 < 0 ? 0 : (c0 > ncol - 3 ? ncol - 3 : c0); r0 = r0 < 0 ? 0 : (r0 > nrow - 3 ? nrow - 3 : r0);
    real best = (real)1e30;
    for (int j = 0; j < 2; j++) for (int i = 0; i < 2; i++) {
      int cc = c0 + i, rr = r0 + j;
      real x0 = -sx + dx * (real)cc, y0 = -sy + dy * (real)rr, x1 = x0 + dx, y1 = y0 + dy;
      real v00[3] = {x0, y0, HZ(rr, cc)}, v10[3] = {x1, y0, HZ(rr, cc + 1)}, v01[3] = {x0, y1, HZ(rr + 1, cc)}, v11[3] = {x1, y1, HZ(rr + 1, cc + 1)};
      for (int t = 0; t < 2; t++) {
        real qq[3];
        if (t == 0) closest_on_triangle(p, v00, v10, v11, qq); else closest_on_triangle(p, v00, v11, v01, qq);
        real d[3] = {p[0] - qq[0], p[1] - qq[1], p[2] - qq[2]};
        real d2 = v3dot(d, d);
        if (d2 < best) { best = d2; v3copy(q, qq); }
      }
    }
    real dn = rsqrt_(best);
    if (dn < (real)1e-12) { n[0] = 0; n[1] = 0; n[2] = 1; }
    else for (int k = 0; k < 3; k++) n[k] = (p[k] - q[k]) / dn;
    dist = dn - radius;
  }
emit:
#undef HZ
  out[0].dist = dist;
  real pl[3];
  for (int k = 0; k < 3; k++) pl[k] = q[k] + n[k] * ((real)0.5 * dist);
  for (int c = 0; c < 3; c++) {
    normal[c] = hmat[3 * c] * n[0] + hmat[3 * c + 1] * n[1] + hmat[3 * c + 2] * n[2];
    out[0].pos[c] = hpos[c] + hmat[3 * c] * pl[0] + hmat[3 * c + 1] * pl[1] + hmat[3 * c + 2] * pl[2];
  }
  return 1;
}

/* box-box: 15-axis SAT, reference-face clipping (Sutherland-Hodgman against the reference rectangle),
 * <=4 manifold points; edge-edge gives one point.  Only penetrating configurations produce contacts
 * (separated pairs contribute nothing to the solve).
 * normal points from box A (geom1) to box B (geom2). */
static int box_box(const real *pa, const real *Ra, const real *sa, const real *pb, const real *Rb, const real *sb,
                   cpoint *out, real *normal) {
  real C[3][3], AC[3][3], t[3], dp[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
  for (int i = 0; i < 3; i++) {
    real ai[3] = {Ra[i], Ra[3 + i], Ra[6 + i]};
    t[i] = v3dot(ai, dp);
    for (int j = 0; j < 3; j++) {
      real bj[3] = {Rb[j], Rb[3 + j], Rb[6 + j]};
      C[i][j] = v3dot(ai, bj);
      AC[i][j] = (real)fabs((double)C[i][j]) + (real)1e-6;
    }
  }
  /* face axes */
  real best_face = (real)-1e30; int face_code = -1;
  for (int i = 0; i < 3; i++) {
    real s = (real)fabs((double)t[i]) - (sa[i] + sb[0] * AC[i][0] + sb[1] * AC[i][1] + sb[2] * AC[i][2]);
    if (s > 0) return 0;
    if (s > best_face) { best_face = s; face_code = i; }
  }
  for (int j = 0; j < 3; j++) {
    real tb = t[0] * C[0][j] + t[1] * C[1][j] + t[2] * C[2][j];
    real s = (real)fabs((double)tb) - (sb[j] + sa[0] * AC[0][j] + sa[1] * AC[1][j] + sa[2] * AC[2][j]);
    if (s > 0) return 0;
    if (s > best_face) { best_face = s; face_code = 3 + j; }
  }
  /* edge axes */
  real best_edge = (real)-1e30; int edge_i = -1, edge_j = -1;
  for (int i = 0; i < 3; i++) {
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
    for (int j = 0; j < 3; j++) {
      int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      real l2 = 1 - C[i][j] * C[i][j];
      if (l2 < (real)1e-6) continue;
      real ra = sa[i1] * AC[i2][j] + sa[i2] * AC[i1][j];
      real rb = sb[j1] * AC[i][j2] + sb[j2] * AC[i][j1];
      real tl = t[i2] * C[i1][j] - t[i1] * C[i2][j];
      real s = ((real)fabs((double)tl) - (ra + rb)) / rsqrt_(l2);
      if (s > 0) return 0;
      if (s > best_edge) { best_edge = s; edge_i = i; edge_j = j; }
    }
  }
  int use_edge = (edge_i >= 0) && (best_edge > (real)0.95 * best_face + (real)1e-6);
  if (use_edge) {
    real ai[3] = {Ra[edge_i], Ra[3 + edge_i], Ra[6 + edge_i]};
    real bj[3] = {Rb[edge_j], Rb[3 + edge_j], Rb[6 + edge_j]};
    real L[3]; v3cross(L, ai, bj);
    real ln = v3norm(L);
    for (int c = 0; c < 3; c++) L[c] /= ln;
    if (v3dot(L, dp) < 0) for (int c = 0; c < 3; c++) L[c] = -L[c];
    /* supporting edges */
    real ea[3] = {pa[0], pa[1], pa[2]}, eb[3] = {pb[0], pb[1], pb[2]};
    for (int k = 0; k < 3; k++) {
      if (k != edge_i) {
        real ak[3] = {Ra[k], Ra[3 + k], Ra[6 + k]};
        real sgn = v3dot(L, ak) > 0 ? (real)1 : (real)-1;
        for (int c = 0; c < 3; c++) ea[c] += sgn * sa[k] * ak[c];
      }
      if (k != edge_j) {
        real bk[3] = {Rb[k], Rb[3 + k], Rb[6 + k]};
        real sgn = v3dot(L, bk) > 0 ? (real)-1 : (real)1;
        for (int c = 0; c < 3; c++) eb[c] += sgn * sb[k] * bk[c];
      }
    }
    /* closest points of lines ea + s*ai, eb + u*bj */
    real r[3] = {eb[0] - ea[0], eb[1] - ea[1], eb[2] - ea[2]};
    real uab = v3dot(ai, bj), q1 = v3dot(ai, r), q2 = -v3dot(bj, r);
    real den = 1 - uab * uab;
    real s = (q1 + uab * q2) / den, u = (uab * q1 + q2) / den;
    s = clampr(s, -sa[edge_i], sa[edge_i]);
    u = clampr(u, -sb[edge_j], sb[edge_j]);
    real qa[3], qb[3];
    for (int c = 0; c < 3; c++) { qa[c] = ea[c] + s * ai[c]; qb[c] = eb[c] + u * bj[c]; }
    real dvec[3] = {qb[0] - qa[0], qb[1] - qa[1], qb[2] - qa[2]};
    real dist = v3dot(dvec, L);
    if (!(dist < 0)) return 0;
    v3copy(normal, L);
    out[0].dist = dist;
    for (int c = 0; c < 3; c++) out[0].pos[c] = (real)0.5 * (qa[c] + qb[c]);
    return 1;
  }
  /* face contact: reference box R, incident box Q */
  int ref_is_a = face_code < 3, k = ref_is_a ? face_code : face_code - 3;
  const real *pr = ref_is_a ? pa : pb, *Rr = ref_is_a ? Ra : Rb, *sr = ref_is_a ? sa : sb;
  const real *pq = ref_is_a ? pb : pa, *Rq = ref_is_a ? Rb : Ra, *sq = ref_is_a ? sb : sa;
  real nref[3] = {Rr[k], Rr[3 + k], Rr[6 + k]};
  real dq[3] = {pq[0] - pr[0], pq[1] - pr[1], pq[2] - pr[2]};
  if (v3dot(nref, dq) < 0) for (int c = 0; c < 3; c++) nref[c] = -nref[c];
  /* incident face: most anti-parallel to nref */
  int mq = 0; real bestd = -1;
  real dots[3];
  for (int j = 0; j < 3; j++) {
    real qj[3] = {Rq[j], Rq[3 + j], Rq[6 + j]};
    dots[j] = v3dot(nref, qj);
    if ((real)fabs((double)dots[j]) > bestd) { bestd = (real)fabs((double)dots[j]); mq = j; }
  }
  real sgn_q = dots[mq] > 0 ? (real)-1 : (real)1;
  int uq = (mq + 1) % 3, vq = (mq + 2) % 3;
  int ur = (k + 1) % 3, vr = (k + 2) % 3;
  real axu[3] = {Rr[ur], Rr[3 + ur], Rr[6 + ur]}, axv[3] = {Rr[vr], Rr[3 + vr], Rr[6 + vr]};
  real o[3];
  for (int c = 0; c < 3; c++) o[c] = pr[c] + nref[c] * sr[k];
  real qm[3] = {Rq[mq], Rq[3 + mq], Rq[6 + mq]}, qu[3] = {Rq[uq], Rq[3 + uq], Rq[6 + uq]}, qv[3] = {Rq[vq], Rq[3 + vq], Rq[6 + vq]};
  real px[2][8], py[2][8], pd[2][8];
  int np_ = 4;
  static const real su[4] = {1, -1, -1, 1}, sv[4] = {1, 1, -1, -1};
  for (int i = 0; i < 4; i++) {
    real w[3];
    for (int c = 0; c < 3; c++)
      w[c] = pq[c] + sgn_q * sq[mq] * qm[c] + su[i] * sq[uq] * qu[c] + sv[i] * sq[vq] * qv[c] - o[c];
    px[0][i] = v3dot(w, axu); py[0][i] = v3dot(w, axv); pd[0][i] = -v3dot(w, nref);
  }
  /* clip against the 4 sides of the reference rectangle */
  int cur = 0;
  for (int side = 0; side < 4; side++) {
    real h = (side < 2) ? sr[ur] : sr[vr];
    real sg = (side & 1) ? (real)-1 : (real)1;
    const real *cx = (side < 2) ? px[cur] : py[cur];
    int nxt = 1 - cur, nn = 0;
    for (int i = 0; i < np_; i++) {
      int i2 = (i + 1 == np_) ? 0 : i + 1;
      real d1 = h - sg * cx[i], d2 = h - sg * cx[i2];   /* >= 0 inside */
      if (d1 >= 0) { px[nxt][nn] = px[cur][i]; py[nxt][nn] = py[cur][i]; pd[nxt][nn] = pd[cur][i]; nn++; }
      if ((d1 >= 0) != (d2 >= 0)) {
        real tt = d1 / (d1 - d2);
        px[nxt][nn] = px[cur][i] + tt * (px[cur][i2] - px[cur][i]);
        py[nxt][nn] = py[cur][i] + tt * (py[cur][i2] - py[cur][i]);
        pd[nxt][nn] = pd[cur][i] + tt * (pd[cur][i2] - pd[cur][i]);
        nn++;
      }
    }
    np_ = nn; cur = nxt;
    if (np_ == 0) return 0;
  }
  int mask[8], idx[4], any = 0;
  for (int i = 0; i < np_; i++) { mask[i] = pd[cur][i] > 0; any |= mask[i]; }
  if (!any) return 0;
  manifold_points(px[cur], py[cur], mask, np_, idx);
  for (int c = 0; c < 3; c++) normal[c] = ref_is_a ? nref[c] : -nref[c];
  int nout = 0;
  for (int i = 0; i < 4; i++) {
    int dup = 0;
    for (int j = 0; j < i; j++) if (idx[j] == idx[i]) dup = 1;
    if (dup || !mask[idx[i]]) continue;
    real x = px[cur][idx[i]], y = py[cur][idx[i]], dep = pd[cur][idx[i]];
    out[nout].dist = -dep;
    for (int c = 0; c < 3; c++) out[nout].pos[c] = o[c] + x * axu[c] + y * axv[c] - (real)0.5 * dep * nref[c];
    nout++;
  }
  return nout;
}

static void collision(const omodel *m, odata *d) {
  d->ncon = 0; d->ncon_overflow = 0;
  for (int p = 0; p < m->npair; p++) {
    int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p];
    cpoint pts[4]; real normal[3], cframe[9]; int n = 0, own_frame = 0;
    real s1[3] = {m->geom_size[3 * g1], m->geom_size[3 * g1 + 1], m->geom_size[3 * g1 + 2]};
    real s2[3] = {m->geom_size[3 * g2], m->geom_size[3 * g2 + 1], m->geom_size[3 * g2 + 2]};
    switch (m->pair_kind[p]) {
      case PAIR_PLANE_BOX:
        n = plane_box(&d->geom_xpos[3 * g1], &d->geom_xmat[9 * g1], &d->geom_xpos[3 * g2], &d->geom_xmat[9 * g2], s2, pts, normal);
        break;
      case PAIR_BOX_BOX:
        n = box_box(&d->geom_xpos[3 * g1], &d->geom_xmat[9 * g1], s1, &d->geom_xpos[3 * g2], &d->geom_xmat[9 * g2], s2, pts, normal);
        break;
      case PAIR_PLANE_SPHERE:
        n = plane_sphere(&d->geom_xpos[3 * g1], &d->geom_xmat[9 * g1], &d->geom_xpos[3 * g2], s2[0], pts, normal);
        break;
      case PAIR_HFIELD_SPHERE:
        n = hfield_sphere(&d->geom_xpos[3 * g1], &d->geom_xmat[9 * g1], m->hfield_size, m->hfield_nrow[0], m->hfield_ncol[0], m->hfield_data,
                          &d->geom_xpos[3 * g2], s2[0], pts, normal);
        break;
      case PAIR_PLANE_CAPSULE:
        n = plane_capsule(&d->geom_xpos[3 * g1], &d->geom_xmat[9 * g1], &d->geom_xpos[3 * g2], &d->geom_xmat[9 * g2], s2[0], s2[1], pts, normal, cframe);
        own_frame = 1;
        break;
      case PAIR_PLANE_CYLINDER:
        n = plane_cylinder(&d->geom_xpos[3 * g1], &d->geom_xmat[9 * g1], &d->geom_xpos[3 * g2], &d->geom_xmat[9 * g2], s2[0], s2[1], pts, normal);
        break;
      default: n = 0;
    }
    real includemargin = m->pair_margin[p] - m->pair_gap[p];
    for (int i = 0; i < n; i++) {
      if (g_cull && !(pts[i].dist - includemargin < 0)) continue;
      if (d->ncon >= g_ncon_cap) { d->ncon_overflow++; continue; }
      ocontact *c = &d->con[d->ncon++];
      c->dist = pts[i].dist; v3copy(c->pos, pts[i].pos);
      if (own_frame) for (int k = 0; k < 9; k++) c->frame[k] = cframe[k]; else make_frame(c->frame, normal);
      c->includemargin = includemargin;
      c->body1 = m->geom_bodyid[g1]; c->body2 = m->geom_bodyid[g2];
      c->condim = m->pair_condim[p]; c->pair = p;
      /* friction: element-wise max unless priorities differ (domain randomisation scales geom_friction) */
      real f[3];
      int p1 = m->geom_priority[g1], p2 = m->geom_priority[g2];
      for (int k = 0; k < 3; k++) {
        real f1 = d->geom_friction[3 * g1 + k], f2 = d->geom_friction[3 * g2 + k];
        f[k] = (p1 == p2) ? (f1 > f2 ? f1 : f2) : (p1 > p2 ? f1 : f2);
      }
      c->friction[0] = f[0]; c->friction[1] = f[0]; c->friction[2] = f[1]; c->friction[3] = f[2]; c->friction[4] = f[2];
      c->solref[0] = m->pair_solref[2 * p]; c->solref[1] = m->pair_solref[2 * p + 1];
      for (int k = 0; k < 5; k++) c->solimp[k] = m->pair_solimp[5 * p + k];
    }
  }
}

/* ------------------------------------------------------------------ stage 5: constraint rows (Appendix B.5, B.10) */
static void kbi(const omodel *m, const real *solref, const real *solimp, real pos, real *k, real *b, real *imp) {
  real timeconst = solref[0], dampratio = solref[1];
  if (!m->disable_refsafe) { real lo = 2 * m->timestep; if (timeconst < lo) timeconst = lo; }
  real dmin = clampr(solimp[0], MINIMP, MAXIMP), dmax = clampr(solimp[1], MINIMP, MAXIMP);
  real width = solimp[2] < MINVAL ? MINVAL : solimp[2];
  real mid = clampr(solimp[3], MINIMP, MAXIMP), power = solimp[4] < 1 ? 1 : solimp[4];
  real kk = 1 / (dmax * dmax * timeconst * timeconst * dampratio * dampratio);
  real bb = 2 / (dmax * timeconst);
  if (solref[0] <= 0) kk = -solref[0] / (dmax * dmax);
  if (solref[1] <= 0) bb = -solref[1] / dmax;
  real x = (real)fabs((double)pos) / width;
  real ia = (1 / (real)pow((double)mid, (double)(power - 1))) * (real)pow((double)x, (double)power);
  real ib = 1 - (1 / (real)pow((double)(1 - mid), (double)(power - 1))) * (real)pow((double)(1 - x), (double)power);
  real y = x < mid ? ia : ib;
  real im = dmin + y * (dmax - dmin);
  im = clampr(im, dmin, dmax);
  if (x > 1) im = dmax;
  *k = kk; *b = bb; *imp = im;
}

static void add_row(const omodel *m, odata *d, const real *J, real pos, real invweight, const real *solref,
                    const real *solimp, real floss) {
  int nv = m->nv, r = d->nefc++;
  real k, b, imp, vel = 0;
  kbi(m, solref, solimp, pos, &k, &b, &imp);
  for (int i = 0; i < nv; i++) { d->efc_J[r * nv + i] = J[i]; vel += J[i] * d->qvel[i]; }
  real R = invweight * (1 - imp) / imp;
  if (R < MINVAL) R = MINVAL;
  d->efc_pos[r] = pos; d->efc_R[r] = R; d->efc_D[r] = 1 / R;
  d->efc_aref[r] = -b * vel - k * imp * pos;
  d->efc_floss[r] = floss;
}

static void make_constraint(const omodel *m, odata *d) {
  int nv = m->nv;
  real J[NV_MAX];
  d->nefc = 0;
  /* equality: joint coupling (cube.xml:170-172) */
  for (int e = 0; e < m->neq; e++) {
    if (!m->eq_active0[e]) continue;
    int j1 = m->eq_obj1id[e], j2 = m->eq_obj2id[e];
    const float *data = &m->eq_data[5 * e];
    real pos1 = d->qpos[m->jnt_qposadr[j1]] - d->qpos0[m->jnt_qposadr[j1]];
    real dif = j2 >= 0 ? d->qpos[m->jnt_qposadr[j2]] - d->qpos0[m->jnt_qposadr[j2]] : 0;
    real pw[5] = {1, dif, dif * dif, dif * dif * dif, dif * dif * dif * dif};
    real poly = 0, deriv = 0;
    for (int k = 0; k < 5; k++) poly += (real)data[k] * pw[k];
    for (int k = 1; k < 5; k++) deriv += (real)data[k] * pw[k - 1] * (real)k;
    for (int i = 0; i < nv; i++) J[i] = 0;
    if (j2 >= 0) J[m->jnt_dofadr[j2]] = -deriv;
    J[m->jnt_dofadr[j1]] = 1;
    real iw = m->dof_invweight0[m->jnt_dofadr[j1]] + (j2 >= 0 ? m->dof_invweight0[m->jnt_dofadr[j2]] : 0);
    real sr[2] = {m->eq_solref[2 * e], m->eq_solref[2 * e + 1]}, si[5];
    for (int k = 0; k < 5; k++) si[k] = m->eq_solimp[5 * e + k];
    add_row(m, d, J, pos1 - poly, iw, sr, si, 0);
  }
  d->ne = d->nefc;
  /* dof friction loss */
  for (int i = 0; i < nv; i++) {
    if (!(m->dof_frictionloss[i] > 0)) continue;
    for (int k = 0; k < nv; k++) J[k] = 0;
    J[i] = 1;
    real sr[2] = {m->dof_solref[2 * i], m->dof_solref[2 * i + 1]}, si[5];
    for (int k = 0; k < 5; k++) si[k] = m->dof_solimp[5 * i + k];
    add_row(m, d, J, 0, m->dof_invweight0[i], sr, si, d->dof_frictionloss[i]);
  }
  d->nf = d->nefc - d->ne;
  /* joint limits */
  for (int j = 0; j < m->njnt; j++) {
    if (!m->jnt_limited[j] || (m->jnt_type[j] != JNT_HINGE && m->jnt_type[j] != JNT_SLIDE)) continue;
    real q = d->qpos[m->jnt_qposadr[j]];
    real dmin = q - (real)m->jnt_range[2 * j], dmax = (real)m->jnt_range[2 * j + 1] - q;
    real pos = (dmin < dmax ? dmin : dmax) - (real)m->jnt_margin[j];
    int active = pos < 0;
    if (g_cull && !active) continue;
    for (int k = 0; k < nv; k++) J[k] = 0;
    J[m->jnt_dofadr[j]] = active ? (dmin < dmax ? (real)1 : (real)-1) : 0;
    real sr[2] = {m->jnt_solref[2 * j], m->jnt_solref[2 * j + 1]}, si[5];
    for (int k = 0; k < 5; k++) si[k] = m->jnt_solimp[5 * j + k];
    add_row(m, d, J, pos, m->dof_invweight0[m->jnt_dofadr[j]], sr, si, 0);
  }
  /* contacts, pyramidal cone */
  real jp1[3 * NV_MAX], jr1[3 * NV_MAX], jp2[3 * NV_MAX], jr2[3 * NV_MAX], diff[6 * NV_MAX];
  for (int ci = 0; ci < d->ncon; ci++) {
    const ocontact *c = &d->con[ci];
    real pos = c->dist - c->includemargin;
    int active = pos < 0;
    jac_point(m, d, c->pos, c->body1, jp1, jr1);
    jac_point(m, d, c->pos, c->body2, jp2, jr2);
    for (int r = 0; r < 3; r++)
      for (int i = 0; i < nv; i++) {
        real sp = 0, sr_ = 0;
        for (int k = 0; k < 3; k++) {
          sp += c->frame[3 * r + k] * (jp2[k * nv + i] - jp1[k * nv + i]);
          sr_ += c->frame[3 * r + k] * (jr2[k * nv + i] - jr1[k * nv + i]);
        }
        diff[r * nv + i] = sp; diff[(3 + r) * nv + i] = sr_;
      }
    real t = m->body_invweight0[2 * c->body1] + m->body_invweight0[2 * c->body2];
    if (c->condim == 1) {
      for (int i = 0; i < nv; i++) J[i] = active ? diff[i] : 0;
      add_row(m, d, J, pos, t, c->solref, c->solimp, 0);
      continue;
    }
    real mu0 = c->friction[0];
    real iw = (t + mu0 * mu0 * t) * 2 * mu0 * mu0 / m->impratio;
    for (int k = 1; k < c->condim; k++) {
      real mu = c->friction[k - 1];
      for (int sgn = 0; sgn < 2; sgn++) {
        for (int i = 0; i < nv; i++) {
          real v = diff[i] + (sgn ? -mu : mu) * diff[k * nv + i];
          J[i] = active ? v : 0;
        }
        add_row(m, d, J, pos, iw, c->solref, c->solimp, 0);
      }
    }
  }
}

/* ------------------------------------------------------------------ stage 6: velocity, rne, actuation (Appendix B.6) */
static void fwd_velocity_actuation(const omodel *m, odata *d) {
  int nv = m->nv;
  /* com_vel */
  for (int c = 0; c < 6; c++) d->cvel[c] = 0;
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parentid[b];
    real cvel[6];
    for (int c = 0; c < 6; c++) cvel[c] = d->cvel[6 * p + c];
    for (int k = 0; k < m->body_jntnum[b]; k++) {
      int j = m->body_jntadr[b] + k, da = m->jnt_dofadr[j];
      if (m->jnt_type[j] == JNT_FREE) {
        for (int a = 0; a < 3; a++)
          for (int c = 0; c < 6; c++) cvel[c] += d->cdof[6 * (da + a) + c] * d->qvel[da + a];
        for (int a = 0; a < 3; a++) {
          for (int c = 0; c < 6; c++) d->cdof_dot[6 * (da + a) + c] = 0;
          motion_cross(&d->cdof_dot[6 * (da + 3 + a)], cvel, &d->cdof[6 * (da + 3 + a)]);
        }
        for (int a = 3; a < 6; a++)
          for (int c = 0; c < 6; c++) cvel[c] += d->cdof[6 * (da + a) + c] * d->qvel[da + a];
      } else {
        motion_cross(&d->cdof_dot[6 * da], cvel, &d->cdof[6 * da]);
        for (int c = 0; c < 6; c++) cvel[c] += d->cdof[6 * da + c] * d->qvel[da];
      }
    }
    for (int c = 0; c < 6; c++) d->cvel[6 * b + c] = cvel[c];
  }
  /* object velocities at the sites (MuJoCo mj_objectVelocity, world frame): the sources of gyro / velocimeter /
   * framelinvel / frameangvel sensors (reference go2_mjx_feetonly.xml:207-230) */
  for (int s_ = 0; s_ < m->nsite; s_++) {
    int b = m->site_bodyid[s_];
    const real *cv = &d->cvel[6 * b], *rc = &d->subtree_com[3 * m->body_rootid[b]];
    real off[3] = {d->site_xpos[3 * s_] - rc[0], d->site_xpos[3 * s_ + 1] - rc[1], d->site_xpos[3 * s_ + 2] - rc[2]}, t[3];
    v3cross(t, cv, off);
    for (int c = 0; c < 3; c++) { d->site_angvel[3 * s_ + c] = cv[c]; d->site_linvel[3 * s_ + c] = cv[3 + c] + t[c]; }
  }
  /* passive */
  for (int i = 0; i < nv; i++) d->qfrc_passive[i] = -d->dof_damping[i] * d->qvel[i];
  /* rne */
  real cacc[NBODY_MAX * 6], cfrc[NBODY_MAX * 6];
  for (int c = 0; c < 3; c++) { cacc[c] = 0; cacc[3 + c] = -m->gravity[c]; }
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parentid[b];
    for (int c = 0; c < 6; c++) cacc[6 * b + c] = cacc[6 * p + c];
    for (int k = 0; k < m->body_dofnum[b]; k++) {
      int i = m->body_dofadr[b] + k;
      for (int c = 0; c < 6; c++) cacc[6 * b + c] += d->cdof_dot[6 * i + c] * d->qvel[i];
    }
  }
  for (int b = 0; b < m->nbody; b++) {
    real f1[6], f2[6], f3[6];
    inert_mul(f1, &d->cinert[10 * b], &cacc[6 * b]);
    inert_mul(f2, &d->cinert[10 * b], &d->cvel[6 * b]);
    motion_cross_force(f3, &d->cvel[6 * b], f2);
    for (int c = 0; c < 6; c++) cfrc[6 * b + c] = f1[c] + f3[c];
  }
  for (int b = m->nbody - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    for (int c = 0; c < 6; c++) cfrc[6 * p + c] += cfrc[6 * b + c];
  }
  for (int i = 0; i < nv; i++) {
    real s = 0;
    for (int c = 0; c < 6; c++) s += d->cdof[6 * i + c] * cfrc[6 * m->dof_bodyid[i] + c];
    d->qfrc_bias[i] = s;
  }
  /* actuation (joint transmission, position servos) */
  for (int i = 0; i < nv; i++) d->qfrc_actuator[i] = 0;
  for (int u = 0; u < m->nu; u++) {
    int j = m->actuator_trnid[u], qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    real gear = m->actuator_gear[u];
    real length = d->qpos[qa] * gear, velocity = d->qvel[da] * gear;
    real ctrl = d->ctrl[u];
    if (m->actuator_ctrllimited[u]) ctrl = clampr(ctrl, m->actuator_ctrlrange[2 * u], m->actuator_ctrlrange[2 * u + 1]);
    real force = d->gainprm[3 * u] * ctrl + d->biasprm[3 * u] + d->biasprm[3 * u + 1] * length + d->biasprm[3 * u + 2] * velocity;
    if (m->actuator_forcelimited[u]) force = clampr(force, m->actuator_forcerange[2 * u], m->actuator_forcerange[2 * u + 1]);
    d->actuator_force[u] = force;
    d->qfrc_actuator[da] += gear * force;
  }
  for (int i = 0; i < nv; i++) {
    int j = m->dof_jntid[i];
    if (m->jnt_actfrclimited[j])
      d->qfrc_actuator[i] = clampr(d->qfrc_actuator[i], m->jnt_actfrcrange[2 * j], m->jnt_actfrcrange[2 * j + 1]);
  }
  /* qacc_smooth */
  for (int i = 0; i < nv; i++) d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i];
  if (d->xfrc_body > 0) {   /* support.xfrc_accumulate: J(xipos[body])^T force, MJX forward.fwd_acceleration */
    real jacp[3 * NV_MAX];
    jac_point(m, d, &d->xipos[3 * d->xfrc_body], d->xfrc_body, jacp, NULL);
    for (int i = 0; i < nv; i++)
      d->qfrc_smooth[i] += jacp[i] * d->xfrc_force[0] + jacp[nv + i] * d->xfrc_force[1] + jacp[2 * nv + i] * d->xfrc_force[2];
  }
  for (int i = 0; i < nv; i++) d->qacc_smooth[i] = d->qfrc_smooth[i];
  chol_solve(d->L, d->qacc_smooth, nv, nv);
}

/* ------------------------------------------------------------------ stage 7: Newton solver (Appendix B.7, B.10) */
typedef struct {
  real qacc[NV_MAX], Ma[NV_MAX], grad[NV_MAX], Mgrad[NV_MAX], search[NV_MAX], qfrc_constraint[NV_MAX];
  real *Jaref, *force; unsigned char *active;
  real gauss, cost, prev_cost;
} sctx;

static void mul_m(const omodel *m, const odata *d, real *o, const real *v) {
  int nv = m->nv;
  for (int i = 0; i < nv; i++) { real s = 0; for (int j = 0; j < nv; j++) s += d->M[i * nv + j] * v[j]; o[i] = s; }
}

static void update_constraint(const omodel *m, odata *d, sctx *c) {
  int nv = m->nv, ne = d->ne, nenf = d->ne + d->nf;
  real cost = 0;
  for (int r = 0; r < d->nefc; r++) {
    real x = c->Jaref[r], D = d->efc_D[r], f = 0;
    int act;
    if (r < ne) act = 1;
    else if (r < nenf) {
      real fl = d->efc_floss[r], rf = d->efc_R[r] * fl;
      if (x <= -rf) { act = 0; f = fl; cost += fl * ((real)-0.5 * rf - x); }
      else if (x >= rf) { act = 0; f = -fl; cost += fl * ((real)-0.5 * rf + x); }
      else act = 1;
    } else act = x < 0;
    if (act) { f = -D * x; cost += (real)0.5 * D * x * x; }
    c->active[r] = (unsigned char)act; c->force[r] = f;
  }
  for (int i = 0; i < nv; i++) c->qfrc_constraint[i] = 0;
  for (int r = 0; r < d->nefc; r++) {
    real f = c->force[r];
    if (f != 0) for (int i = 0; i < nv; i++) c->qfrc_constraint[i] += d->efc_J[r * nv + i] * f;
  }
  real gauss = 0;
  for (int i = 0; i < nv; i++) gauss += (c->Ma[i] - d->qfrc_smooth[i]) * (c->qacc[i] - d->qacc_smooth[i]);
  gauss *= (real)0.5;
  c->gauss = gauss;
  c->prev_cost = c->cost;
  c->cost = cost + gauss;
}

static void update_gradient(const omodel *m, odata *d, sctx *c) {
  int nv = m->nv;
  real H[NV_MAX * NV_MAX], Lh[NV_MAX * NV_MAX];
  for (int i = 0; i < nv; i++) c->grad[i] = c->Ma[i] - d->qfrc_smooth[i] - c->qfrc_constraint[i];
  for (int i = 0; i < nv * nv; i++) H[i] = d->M[i];
  for (int r = 0; r < d->nefc; r++) {
    if (!c->active[r]) continue;
    const real *J = &d->efc_J[r * nv];
    real D = d->efc_D[r];
    for (int i = 0; i < nv; i++) {
      real a = J[i] * D;
      if (a == 0) continue;
      for (int j = 0; j <= i; j++) H[i * nv + j] += a * J[j];
    }
  }
  for (int i = 0; i < nv; i++) for (int j = i + 1; j < nv; j++) H[i * nv + j] = H[j * nv + i];
  cholesky(Lh, H, nv, nv);
  for (int i = 0; i < nv; i++) c->Mgrad[i] = c->grad[i];
  chol_solve(Lh, c->Mgrad, nv, nv);
}

static void sctx_init(const omodel *m, odata *d, sctx *c, const real *qacc, int with_grad) {
  int nv = m->nv;
  for (int i = 0; i < nv; i++) c->qacc[i] = qacc[i];
  mul_m(m, d, c->Ma, qacc);
  for (int r = 0; r < d->nefc; r++) {
    real s = 0;
    for (int i = 0; i < nv; i++) s += d->efc_J[r * nv + i] * qacc[i];
    c->Jaref[r] = s - d->efc_aref[r];
  }
  c->cost = (real)INFINITY; c->prev_cost = 0;
  update_constraint(m, d, c);
  if (with_grad) {
    update_gradient(m, d, c);
    for (int i = 0; i < nv; i++) c->search[i] = -c->Mgrad[i];
  }
}

typedef struct { real alpha, cost, deriv0, deriv1; } lspoint;

static lspoint ls_point(const odata *d, const sctx *c, real alpha, const real *jv, const real *quad, const real *qg) {
  int ne = d->ne, nenf = d->ne + d->nf;
  real q0 = qg[0], q1 = qg[1], q2 = qg[2];
  for (int r = 0; r < d->nefc; r++) {
    real x = c->Jaref[r] + alpha * jv[r];
    const real *q = &quad[3 * r];
    if (r < ne) { q0 += q[0]; q1 += q[1]; q2 += q[2]; }
    else if (r < nenf) {
      real f = d->efc_floss[r], rf = d->efc_R[r] * f;
      if (x <= -rf) { q0 += f * ((real)-0.5 * rf - c->Jaref[r]); q1 += -f * jv[r]; }
      else if (x >= rf) { q0 += f * ((real)-0.5 * rf + c->Jaref[r]); q1 += f * jv[r]; }
      else { q0 += q[0]; q1 += q[1]; q2 += q[2]; }
    } else if (x < 0) { q0 += q[0]; q1 += q[1]; q2 += q[2]; }
  }
  lspoint p;
  p.alpha = alpha;
  p.cost = alpha * alpha * q2 + alpha * q1 + q0;
  p.deriv0 = 2 * alpha * q2 + q1;
  p.deriv1 = 2 * q2 + (q2 == 0 ? MINVAL : 0);
  return p;
}

static int g_ls_trace = -1;      /* RSR_LS_TRACE=1: the bracket of every line-search iteration on stderr (read once) */
static void linesearch(const omodel *m, odata *d, sctx *c, real *jv, real *quad) {
  int nv = m->nv;
  real snorm = 0, mv[NV_MAX];
  for (int i = 0; i < nv; i++) snorm += c->search[i] * c->search[i];
  snorm = rsqrt_(snorm);
  real smag = snorm * m->meaninertia * (real)(nv > 1 ? nv : 1);
  real gtol = m->tolerance * m->ls_tolerance * smag;
  mul_m(m, d, mv, c->search);
  for (int r = 0; r < d->nefc; r++) {
    real s = 0;
    for (int i = 0; i < nv; i++) s += d->efc_J[r * nv + i] * c->search[i];
    jv[r] = s;
    real D = d->efc_D[r], ja = c->Jaref[r];
    quad[3 * r] = (real)0.5 * ja * ja * D; quad[3 * r + 1] = s * ja * D; quad[3 * r + 2] = (real)0.5 * s * s * D;
  }
  real qg[3] = {c->gauss, 0, 0}, sMa = 0, sf = 0, sMv = 0;
  for (int i = 0; i < nv; i++) { sMa += c->search[i] * c->Ma[i]; sf += c->search[i] * d->qfrc_smooth[i]; sMv += c->search[i] * mv[i]; }
  qg[1] = sMa - sf; qg[2] = (real)0.5 * sMv;

  /* rounding-noise scale of the 1-D derivative (rules 1, 2; same sums as rsr_solver.hpp) */
  real n1 = 0, n2 = 0;
  if (g_ls_rule != 0 && m->iterations > 1) {
    for (int r = 0; r < d->nefc; r++) {
      n1 += (real)fabs((double)(jv[r] * c->Jaref[r] * d->efc_D[r]));
      if (r >= d->ne && r < d->ne + d->nf && d->efc_floss[r] > 0) n1 += (real)fabs((double)(d->efc_floss[r] * jv[r]));
      n2 += (real)0.5 * jv[r] * jv[r] * d->efc_D[r];
    }
    for (int i = 0; i < nv; i++) n1 += (real)fabs((double)(c->search[i] * c->Ma[i])) + (real)fabs((double)(c->search[i] * d->qfrc_smooth[i]));
    n2 += (real)fabs((double)qg[2]);
  }
  const real noise = (g_ls_rule != 0 && m->iterations > 1) ? (real)(g_ls_noise * 1.1920929e-7) : 0;

  lspoint p0 = ls_point(d, c, 0, jv, quad, qg);
  lspoint lo = ls_point(d, c, p0.alpha - p0.deriv0 / p0.deriv1, jv, quad, qg), hi;
  if (lo.deriv0 < p0.deriv0) { hi = p0; } else { hi = lo; lo = p0; }
  int swap = 1, it = 0, cap = m->ls_iterations;
  enum { LS_HIST = 8 };
  real hist_lo[LS_HIST], hist_hi[LS_HIST];
  while (1) {
    if (g_ls_cycle && cap == m->ls_iterations) {
      for (int P = 1; P <= LS_HIST && P <= it; P++)
        if (hist_lo[(it - P) % LS_HIST] == lo.alpha && hist_hi[(it - P) % LS_HIST] == hi.alpha) { cap = it + (m->ls_iterations - it) % P; if (getenv("RSR_LS_TRACE")) fprintf(stderr, "cycle P=%d at it=%d\n", P, it); break; }
      hist_lo[it % LS_HIST] = lo.alpha; hist_hi[it % LS_HIST] = hi.alpha;
    }
    if (g_ls_trace < 0) g_ls_trace = getenv("RSR_LS_TRACE") != NULL;
    if (g_ls_trace) fprintf(stderr, "  ls it %d lo a=%.9g d0=%.6g c=%.9g | hi a=%.9g d0=%.6g c=%.9g | p0 d0=%.6g d1=%.6g c=%.9g\n", it, (double)lo.alpha, (double)lo.deriv0, (double)lo.cost, (double)hi.alpha, (double)hi.deriv0, (double)hi.cost, (double)p0.deriv0, (double)p0.deriv1, (double)p0.cost);
    int done = it >= cap;
    done |= !swap;
    real tol_lo = gtol, tol_hi = gtol;
    if (g_ls_rule == 2 && noise > 0) {
      /* the kernel's rule: MJX's signed test for MJX's tolerance, and -- sign-free -- the rounding noise of the derivative's own sum */
      real a = noise * (n1 + 2 * (real)fabs((double)lo.alpha) * n2), b = noise * (n1 + 2 * (real)fabs((double)hi.alpha) * n2);
      done |= ((lo.deriv0 < 0) && (lo.deriv0 > -gtol)) || ((real)fabs((double)lo.deriv0) < a);
      done |= ((hi.deriv0 > 0) && (hi.deriv0 < gtol)) || ((real)fabs((double)hi.deriv0) < b);
    } else {
      if (noise > 0) {       /* rule 1: the noise floor as a larger tolerance of the signed test */
        real a = noise * (n1 + 2 * (real)fabs((double)lo.alpha) * n2), b = noise * (n1 + 2 * (real)fabs((double)hi.alpha) * n2);
        if (a > tol_lo) tol_lo = a;
        if (b > tol_hi) tol_hi = b;
      }
      done |= (lo.deriv0 < 0) && (lo.deriv0 > -tol_lo);
      done |= (hi.deriv0 > 0) && (hi.deriv0 < tol_hi);
    }
    if (done) break;
    lspoint lo_next = ls_point(d, c, lo.alpha - lo.deriv0 / lo.deriv1, jv, quad, qg);
    lspoint hi_next = ls_point(d, c, hi.alpha - hi.deriv0 / hi.deriv1, jv, quad, qg);
    lspoint mid = ls_point(d, c, (real)0.5 * (lo.alpha + hi.alpha), jv, quad, qg);
    int s1 = (lo.deriv0 > 0) || (lo.deriv0 < lo_next.deriv0);
    if (s1) lo = lo_next;
    int s2 = (mid.deriv0 < 0) && (lo.deriv0 < mid.deriv0);
    if (s2) lo = mid;
    int s3 = (hi.deriv0 < 0) || (hi.deriv0 > hi_next.deriv0);
    if (s3) hi = hi_next;
    int s4 = (mid.deriv0 > 0) && (hi.deriv0 > mid.deriv0);
    if (s4) hi = mid;
    swap = s1 | s2 | s3 | s4;
    it++;
  }
  d->ls_total += it;
#pragma omp atomic
  g_ls_calls++;
#pragma omp atomic
  g_ls_iters += it;
#pragma omp atomic
  g_ls_hist[it < 51 ? it : 51]++;
  int improved = (lo.cost < p0.cost) || (hi.cost < p0.cost);
  real alpha = lo.cost < hi.cost ? lo.alpha : hi.alpha;
  if (improved) {
    for (int i = 0; i < nv; i++) { c->qacc[i] += alpha * c->search[i]; c->Ma[i] += alpha * mv[i]; }
    for (int r = 0; r < d->nefc; r++) c->Jaref[r] += alpha * jv[r];
  }
}

static void solve(const omodel *m, odata *d) {
  int nv = m->nv, nefc = d->nefc;
  d->solver_niter = 0; d->ls_total = 0;
  if (nefc == 0) {
    for (int i = 0; i < nv; i++) { d->qacc[i] = d->qacc_smooth[i]; d->qfrc_constraint[i] = 0; d->qacc_warmstart[i] = d->qacc[i]; }
    return;
  }
  real *buf = (real *)malloc(sizeof(real) * (size_t)nefc * 6 + (size_t)nefc);
  sctx c; memset(&c, 0, sizeof(c));
  c.Jaref = buf; c.force = buf + nefc; real *jv = buf + 2 * nefc, *quad = buf + 3 * nefc;
  c.active = (unsigned char *)(buf + 6 * nefc);
  /* warm start: the better of qacc_warmstart and qacc_smooth */
  sctx_init(m, d, &c, d->qacc_warmstart, 0);
  real cost_warm = c.cost;
  sctx_init(m, d, &c, d->qacc_smooth, 0);
  real cost_smooth = c.cost;
  const real *start = cost_warm < cost_smooth ? d->qacc_warmstart : d->qacc_smooth;
  real q0[NV_MAX];
  for (int i = 0; i < nv; i++) q0[i] = start[i];
  sctx_init(m, d, &c, q0, 1);
  real scale = 1 / (m->meaninertia * (real)(nv > 1 ? nv : 1));
  int iter = 0;
  if (getenv("RSR_SOLVER_TRACE")) fprintf(stderr, "   oracle start cost_warm %.9g cost_smooth %.9g cost %.9g\n", (double)cost_warm, (double)cost_smooth, (double)c.cost);
  while (1) {
    if (m->iterations != 1) {
      real gn = 0;
      for (int i = 0; i < nv; i++) gn += c.grad[i] * c.grad[i];
      gn = rsqrt_(gn);
      int done = iter >= m->iterations;
      done |= scale * (c.prev_cost - c.cost) < m->tolerance;
      done |= scale * gn < m->tolerance;
      if (done) break;
    } else if (iter >= 1) break;
    linesearch(m, d, &c, jv, quad);
    update_constraint(m, d, &c);
    if (getenv("RSR_SOLVER_TRACE")) fprintf(stderr, "   oracle iter %d cost %.9g ls_total %d\n", iter, (double)c.cost, d->ls_total);
    update_gradient(m, d, &c);
    for (int i = 0; i < nv; i++) c.search[i] = -c.Mgrad[i];
    iter++;
  }
  d->solver_niter = iter;
  for (int i = 0; i < nv; i++) {
    d->qacc[i] = c.qacc[i]; d->qacc_warmstart[i] = c.qacc[i]; d->qfrc_constraint[i] = c.qfrc_constraint[i];
  }
  for (int r = 0; r < nefc; r++) d->efc_force[r] = c.force[r];
  free(buf);
}

/* ------------------------------------------------------------------ forward + integrate (Appendix B.8) */
/* accelerometer sensor (MuJoCo sensor_acc stage: mj_rnePostConstraint's cacc + mj_objectAcceleration with flg_local):
 * cacc[body] = -gravity at the world, plus cdof_dot*qvel + cdof*qacc over the body's dof chain (com-based spatial
 * acceleration about subtree_com[root]); moved to the site (lin + ang x dif), rotated into the site frame, plus the
 * rotating-frame term w_local x v_local. */
static void sensor_accelerometer(const omodel *m, odata *d) {
  if (d->acc_site_id < 0) return;
  const int sid = d->acc_site_id, body = m->site_bodyid[sid];
  real cacc[6] = {0, 0, 0, -(real)m->gravity[0], -(real)m->gravity[1], -(real)m->gravity[2]};
  int b = body;
  while (b > 0 && m->body_dofnum[b] == 0) b = m->body_parentid[b];
  if (b > 0)
    for (int i = m->body_dofadr[b] + m->body_dofnum[b] - 1; i >= 0; i = m->dof_parentid[i])
      for (int c = 0; c < 6; c++) cacc[c] += d->cdof_dot[6 * i + c] * d->qvel[i] + d->cdof[6 * i + c] * d->qacc[i];
  const real *rc = &d->subtree_com[3 * m->body_rootid[body]], *pos = &d->site_xpos[3 * sid], *R = &d->site_xmat[9 * sid];
  const real *cv = &d->cvel[6 * body];
  real dif[3] = {pos[0] - rc[0], pos[1] - rc[1], pos[2] - rc[2]}, t[3], lin[3], lv[3], al[3], wl[3], vl[3];
  v3cross(t, cacc, dif); for (int c = 0; c < 3; c++) lin[c] = cacc[3 + c] + t[c];
  v3cross(t, cv, dif); for (int c = 0; c < 3; c++) lv[c] = cv[3 + c] + t[c];
  mat_tmulv(al, R, lin); mat_tmulv(wl, R, cv); mat_tmulv(vl, R, lv);
  v3cross(t, wl, vl);
  for (int c = 0; c < 3; c++) d->acc_site[c] = al[c] + t[c];
}

static void forward(const omodel *m, odata *d) {
  kinematics(m, d);
  com_pos(m, d);
  crb_and_factor(m, d);
  collision(m, d);
  make_constraint(m, d);
  fwd_velocity_actuation(m, d);
  solve(m, d);
  sensor_accelerometer(m, d);
}

static void advance(const omodel *m, odata *d, const real *qacc) {
  real dt = m->timestep;
  for (int i = 0; i < m->nv; i++) d->qvel[i] += qacc[i] * dt;
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == JNT_FREE) {
      for (int c = 0; c < 3; c++) d->qpos[qa + c] += dt * d->qvel[da + c];
      real w[3] = {d->qvel[da + 3], d->qvel[da + 4], d->qvel[da + 5]};
      real n = v3norm(w), ax[3] = {0, 0, 0};
      if (n > MINVAL) for (int c = 0; c < 3; c++) ax[c] = w[c] / n;   /* zero axis when |w| = 0 */
      real qr[4], q2[4];
      axis_angle_to_quat(qr, ax, dt * n);
      quat_mul(q2, &d->qpos[qa + 3], qr);
      quat_normalize(q2);
      for (int c = 0; c < 4; c++) d->qpos[qa + 3 + c] = q2[c];
    } else {
      d->qpos[qa] += dt * d->qvel[da];
    }
  }
  d->time += dt;
}

static void step_physics(const omodel *m, odata *d) {
  int nv = m->nv;
  forward(m, d);
  real qacc[NV_MAX];
  for (int i = 0; i < nv; i++) qacc[i] = d->qacc[i];
  int damped = 0;
  for (int i = 0; i < nv; i++) if (d->dof_damping[i] != 0) damped = 1;
  if (m->integrator == INT_IMPLICITFAST || (m->integrator == INT_EULER && !m->disable_eulerdamp && damped)) {
    /* (M + dt*diag(damping)) qacc = qfrc_smooth + qfrc_constraint */
    real A[NV_MAX * NV_MAX], La[NV_MAX * NV_MAX];
    for (int i = 0; i < nv * nv; i++) A[i] = d->M[i];
    for (int i = 0; i < nv; i++) A[i * nv + i] += m->timestep * d->dof_damping[i];
    cholesky(La, A, nv, nv);
    for (int i = 0; i < nv; i++) qacc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
    chol_solve(La, qacc, nv, nv);
  }
  advance(m, d, qacc);
}

/* ------------------------------------------------------------------ threefry2x32 / jax.random (Appendix D) */
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
void oracle_threefry2x32(const uint32_t key[2], const uint32_t ctr[2], uint32_t out[2]) {
  static const int R[8] = {13, 15, 26, 6, 17, 29, 16, 24};
  uint32_t ks[3] = {key[0], key[1], key[0] ^ key[1] ^ 0x1BD11BDAu};
  uint32_t x0 = ctr[0] + ks[0], x1 = ctr[1] + ks[1];
  for (int g = 0; g < 5; g++) {
    const int *r = (g & 1) ? R + 4 : R;
    for (int k = 0; k < 4; k++) { x0 += x1; x1 = rotl32(x1, r[k]); x1 ^= x0; }
    x0 += ks[(g + 1) % 3];
    x1 += ks[(g + 2) % 3] + (uint32_t)(g + 1);
  }
  out[0] = x0; out[1] = x1;
}
/* bits = threefry_2x32(key, iota(n)) with jax's "split the counter array in halves" layout */
static void random_bits(const uint32_t key[2], int n, uint32_t *bits) {
  int half = (n + 1) / 2;
  for (int i = 0; i < half; i++) {
    uint32_t ctr[2] = {(uint32_t)i, (uint32_t)(half + i)}, o[2];
    if (half + i >= n) ctr[1] = 0;     /* odd size: padded with one zero counter */
    oracle_threefry2x32(key, ctr, o);
    bits[i] = o[0];
    if (half + i < n) bits[half + i] = o[1];
  }
}
void oracle_split(const uint32_t key[2], int n, uint32_t *out /* [n][2] */) { random_bits(key, 2 * n, out); }
static inline float bits_to_unit(uint32_t b) {
  uint32_t u = (b >> 9) | 0x3F800000u; float f; memcpy(&f, &u, 4); return f - 1.0f;
}
/* jax.random.uniform(key, (n,), minval, maxval), float32 */
void oracle_uniform(const uint32_t key[2], int n, const float *lo, const float *hi, int lohi_stride, float *out) {
  uint32_t bits[64];
  random_bits(key, n, bits);
  for (int i = 0; i < n; i++) {
    float l = lo[i * lohi_stride], h = hi[i * lohi_stride];
    volatile float scale = h - l;
    volatile float prod = bits_to_unit(bits[i]) * scale;
    float v = prod + l;
    out[i] = v > l ? v : l;
  }
}

/* ------------------------------------------------------------------ batch state (SoA, float32) */
typedef struct {
  int n;
  float *qpos, *qvel, *ctrl, *qacc_warmstart, *time, *xpos, *site_xpos;          /* pipeline_state */
  float *obs, *reward, *done, *metrics;                                          /* outputs */
  float *info_target_pos, *info_new_cube_pos, *info_site_pos, *info_cube_pos, *info_last_action;    /* env info */
  float *info_steps, *info_truncation, *info_episode_done, *info_episode_metrics; /* wrapper info */
  /* T-shape env info (T_shape_env.py:127-134); NULL for the cube envs */
  float *info_target_base_pos, *info_target_vertical_pos, *info_target_w, *info_new_T_pos, *info_T_pos, *info_xita;
  float *info_go2;   /* Go2 joystick info block, GO2_INFO floats per env (layout: enum G2_*); NULL otherwise */
  float *first_qpos, *first_qvel, *first_ctrl, *first_warmstart, *first_time, *first_xpos, *first_site_xpos, *first_obs;
  float *priv_obs, *first_priv_obs;   /* Go2 obs['privileged_state'] (GO2_PRIV floats) and its auto-reset copy; NULL otherwise */
  float *dr_geom_friction, *dr_body_mass, *dr_dof_damping, *dr_dof_frictionloss;  /* NULL = model values */
  float *dr_body_ipos, *dr_qpos0, *dr_dof_armature, *dr_gainprm, *dr_biasprm;      /* Go2 randomize.py leaves; NULL = model values */
  int *stats;   /* [n][4]: solver iterations, line-search iterations, ncon, overflow (last substep) */
} obatch;

static void load_env(const omodel *m, const obatch *s, int e, odata *d) {
  for (int i = 0; i < m->nq; i++) d->qpos[i] = s->qpos[e * m->nq + i];
  for (int i = 0; i < m->nv; i++) { d->qvel[i] = s->qvel[e * m->nv + i]; d->qacc_warmstart[i] = s->qacc_warmstart[e * m->nv + i]; }
  for (int i = 0; i < m->nu; i++) d->ctrl[i] = s->ctrl[e * m->nu + i];
  d->time = s->time[e];
  for (int i = 0; i < m->ngeom * 3; i++) d->geom_friction[i] = s->dr_geom_friction ? s->dr_geom_friction[e * m->ngeom * 3 + i] : m->geom_friction[i];
  for (int i = 0; i < m->nbody; i++) d->body_mass[i] = s->dr_body_mass ? s->dr_body_mass[e * m->nbody + i] : m->body_mass[i];
  for (int i = 0; i < m->nv; i++) {
    d->dof_damping[i] = s->dr_dof_damping ? s->dr_dof_damping[e * m->nv + i] : m->dof_damping[i];
    d->dof_frictionloss[i] = s->dr_dof_frictionloss ? s->dr_dof_frictionloss[e * m->nv + i] : m->dof_frictionloss[i];
    d->dof_armature[i] = s->dr_dof_armature ? s->dr_dof_armature[e * m->nv + i] : m->dof_armature[i];
  }
  for (int i = 0; i < m->nbody * 3; i++) d->body_ipos[i] = s->dr_body_ipos ? s->dr_body_ipos[e * m->nbody * 3 + i] : m->body_ipos[i];
  for (int i = 0; i < m->nq; i++) d->qpos0[i] = s->dr_qpos0 ? s->dr_qpos0[e * m->nq + i] : m->qpos0[i];
  for (int i = 0; i < m->nu * 3; i++) {
    d->gainprm[i] = s->dr_gainprm ? s->dr_gainprm[e * m->nu * 3 + i] : m->actuator_gainprm[i];
    d->biasprm[i] = s->dr_biasprm ? s->dr_biasprm[e * m->nu * 3 + i] : m->actuator_biasprm[i];
  }
}
static void store_pipeline(const omodel *m, obatch *s, int e, const odata *d) {
  for (int i = 0; i < m->nq; i++) s->qpos[e * m->nq + i] = (float)d->qpos[i];
  for (int i = 0; i < m->nv; i++) { s->qvel[e * m->nv + i] = (float)d->qvel[i]; s->qacc_warmstart[e * m->nv + i] = (float)d->qacc_warmstart[i]; }
  for (int i = 0; i < m->nu; i++) s->ctrl[e * m->nu + i] = (float)d->ctrl[i];
  s->time[e] = (float)d->time;
  for (int i = 0; i < m->nbody * 3; i++) s->xpos[e * m->nbody * 3 + i] = (float)d->xpos[i];
  for (int i = 0; i < m->nsite * 3; i++) s->site_xpos[e * m->nsite * 3 + i] = (float)d->site_xpos[i];
  if (s->stats) { s->stats[4 * e] = d->solver_niter; s->stats[4 * e + 1] = d->ls_total; s->stats[4 * e + 2] = d->ncon; s->stats[4 * e + 3] = d->ncon_overflow; }
}

/* env ids layout (env_ids): 0 cube body, 1 target body, 2 endpoint site, 3 cube qposadr, 4 target qposadr,
 * 5 finger qposadr, 6..11 arm joint qposadr */
enum { ID_CUBE = 0, ID_TARGET = 1, ID_SITE = 2, ID_BOXQ = 3, ID_SITEQ = 4, ID_FINGERQ = 5, ID_JOINTQ = 6 };
/* env_reward layout: 0 push_reward_weight, 1 siet_to_box_reward_weight, 2 healthy_reward, 3 endpoint_min_z_pos */
/* env_reset layout: 0 noise, 1..6 arm offsets, 7 finger value, 8..12 ctrl init, 13..15 target min, 16..18 target max,
 * 19..21 cube min, 22..24 cube max, 25..26 new_cube_pos0 */

/* cube_env.py:215-229 */
static void cube_obs(const omodel *m, const float *qpos, const float *site, const float *target_pos, const float *cube,
                     const float *new_cube_pos, float *obs) {
  for (int i = 0; i < 6; i++) obs[i] = qpos[m->env_ids[ID_JOINTQ + i]];
  for (int i = 0; i < 3; i++) obs[6 + i] = site[i];
  for (int i = 0; i < 3; i++) obs[9 + i] = target_pos[i];
  for (int i = 0; i < 3; i++) obs[12 + i] = cube[i];
  for (int i = 0; i < 2; i++) obs[15 + i] = new_cube_pos[i];
  for (int i = 0; i < 3; i++) obs[17 + i] = target_pos[i] - cube[i];
  for (int i = 0; i < 3; i++) obs[20 + i] = cube[i] - site[i];
}

/* ------------------------------------------------------------------ reset (cube_env.py:95-143 + wrappers) */
static void reset_env(const omodel *m, obatch *s, int e, const uint32_t key[2], odata *d) {
  const float *R = m->env_reset;
  uint32_t keys[5][2];
  oracle_split(key, 5, &keys[0][0]);     /* rng, rng1, rng2, rng3, rng4 */
  float lo = -R[0], hi = R[0], u[NQ_MAX];
  oracle_uniform(keys[1], m->nq, &lo, &hi, 0, u);
  float qpos[NQ_MAX], qvel[NV_MAX], ctrl[NU_MAX];
  for (int i = 0; i < m->nq; i++) qpos[i] = m->qpos0[i] + u[i];
  for (int i = 0; i < 6; i++) qpos[m->env_ids[ID_JOINTQ + i]] += R[1 + i];
  qpos[m->env_ids[ID_FINGERQ]] = R[7];
  oracle_uniform(keys[2], m->nv, &lo, &hi, 0, qvel);
  oracle_uniform(keys[3], m->nu, &lo, &hi, 0, u);
  for (int i = 0; i < m->nu; i++) ctrl[i] = R[8 + i] + u[i];
  float target_pos[3], cube_pos[3];
  oracle_uniform(keys[4], 3, &R[13], &R[16], 1, target_pos);
  oracle_uniform(keys[0], 3, &R[19], &R[22], 1, cube_pos);
  for (int i = 0; i < 3; i++) { qpos[m->env_ids[ID_BOXQ] + i] = cube_pos[i]; qpos[m->env_ids[ID_SITEQ] + i] = target_pos[i]; }
  /* pipeline_init: make_data + forward with ctrl = 0, then ctrl replaced (cube_env.py:125-126) */
  s->time[e] = 0;
  for (int i = 0; i < m->nq; i++) s->qpos[e * m->nq + i] = qpos[i];
  for (int i = 0; i < m->nv; i++) { s->qvel[e * m->nv + i] = qvel[i]; s->qacc_warmstart[e * m->nv + i] = 0; }
  for (int i = 0; i < m->nu; i++) s->ctrl[e * m->nu + i] = 0;
  load_env(m, s, e, d);
  forward(m, d);
  for (int i = 0; i < m->nu; i++) d->ctrl[i] = ctrl[i];
  store_pipeline(m, s, e, d);
  int cube = m->env_ids[ID_CUBE], tgt = m->env_ids[ID_TARGET], site = m->env_ids[ID_SITE];
  float *xpos = &s->xpos[e * m->nbody * 3], *sx = &s->site_xpos[e * m->nsite * 3];
  for (int i = 0; i < 3; i++) {
    s->info_target_pos[3 * e + i] = xpos[3 * tgt + i];
    s->info_site_pos[3 * e + i] = sx[3 * site + i];
    s->info_cube_pos[3 * e + i] = xpos[3 * cube + i];
  }
  s->info_new_cube_pos[2 * e] = R[25]; s->info_new_cube_pos[2 * e + 1] = R[26];
  s->info_last_action[e] = 0;
  s->reward[e] = 0; s->done[e] = 0;
  for (int i = 0; i < m->nmetrics; i++) s->metrics[e * m->nmetrics + i] = 0;
  cube_obs(m, &s->qpos[e * m->nq], &sx[3 * site], &s->info_target_pos[3 * e], &xpos[3 * cube],
           &s->info_new_cube_pos[2 * e], &s->obs[e * m->obs_dim]);
  /* EpisodeWrapper.reset / AutoResetWrapper.reset */
  s->info_steps[e] = 0; s->info_truncation[e] = 0; s->info_episode_done[e] = 0;
  for (int i = 0; i < 2 + m->nmetrics; i++) s->info_episode_metrics[e * (2 + m->nmetrics) + i] = 0;
  memcpy(&s->first_qpos[e * m->nq], &s->qpos[e * m->nq], sizeof(float) * (size_t)m->nq);
  memcpy(&s->first_qvel[e * m->nv], &s->qvel[e * m->nv], sizeof(float) * (size_t)m->nv);
  memcpy(&s->first_ctrl[e * m->nu], &s->ctrl[e * m->nu], sizeof(float) * (size_t)m->nu);
  memcpy(&s->first_warmstart[e * m->nv], &s->qacc_warmstart[e * m->nv], sizeof(float) * (size_t)m->nv);
  s->first_time[e] = s->time[e];
  memcpy(&s->first_xpos[e * m->nbody * 3], xpos, sizeof(float) * (size_t)m->nbody * 3);
  memcpy(&s->first_site_xpos[e * m->nsite * 3], sx, sizeof(float) * (size_t)m->nsite * 3);
  memcpy(&s->first_obs[e * m->obs_dim], &s->obs[e * m->obs_dim], sizeof(float) * (size_t)m->obs_dim);
  if ((m->env_kind == ENV_GO2 || m->env_kind == ENV_GO2_HANDSTAND) && s->priv_obs && s->first_priv_obs)
    memcpy(&s->first_priv_obs[(size_t)e * GO2_PRIV], &s->priv_obs[(size_t)e * GO2_PRIV], sizeof(float) * GO2_PRIV);
}

/* ------------------------------------------------------------------ step (cube_env.py:145-213 + wrappers) */
static void step_env(const omodel *m, obatch *s, int e, const float *action, odata *d) {
  const int nq = m->nq, nu = m->nu;
  const int cube = m->env_ids[ID_CUBE], site = m->env_ids[ID_SITE];
  const int wrap_episode = m->wrap_flags & 1, wrap_autoreset = (m->wrap_flags & 2) != 0;
  float *qpos = &s->qpos[e * nq], *xpos = &s->xpos[e * m->nbody * 3], *sx = &s->site_xpos[e * m->nsite * 3];
  float *tp = &s->info_target_pos[3 * e], *ncp = &s->info_new_cube_pos[2 * e];
  /* AutoResetWrapper.step: steps <- 0 where previously done; done cleared */
  if (wrap_autoreset) { if (s->done[e] != 0) s->info_steps[e] = 0; s->done[e] = 0; }
  /* --- prologue (cube_env.py:146-161) --- */
  float act[NU_MAX];
  for (int i = 0; i < nu; i++) {
    volatile float delta = m->env_action_scale[i] * action[e * nu + i];
    act[i] = s->ctrl[e * nu + i] + delta;
  }
  {
    volatile float t1 = 1.57f + qpos[m->env_ids[ID_JOINTQ + 1]];
    volatile float t2 = t1 + qpos[m->env_ids[ID_JOINTQ + 2]];
    act[3] = -t2;
  }
  {
    float dx = tp[0] - xpos[3 * cube], dy = tp[1] - xpos[3 * cube + 1];
    volatile float dxe = dx + 0.00001f;
    float ang = atan2f(dy, dxe);
    volatile float t1 = -ang + act[0];
    act[4] = t1 + 1.5708f;
    if (m->env_kind == ENV_AIRBOT_SF) {   /* test/airbot.py:180-184: hold the wrist target once the cube is within 3 cm */
      float dz = tp[2] - xpos[3 * cube + 2];
      float btd0 = sqrtf(dx * dx + dy * dy + dz * dz);
      if (btd0 < 0.03f) act[4] = s->info_last_action[e];
      s->info_last_action[e] = act[4];
    }
  }
  for (int i = 0; i < nu; i++) act[i] = act[i] < m->env_ctrl_lo[i] ? m->env_ctrl_lo[i] : (act[i] > m->env_ctrl_hi[i] ? m->env_ctrl_hi[i] : act[i]);
  /* --- pipeline_step: n_frames x mjx.step with the same ctrl --- */
  load_env(m, s, e, d);
  for (int i = 0; i < nu; i++) d->ctrl[i] = act[i];
  for (int f = 0; f < m->n_frames; f++) step_physics(m, d);
  store_pipeline(m, s, e, d);
  /* --- epilogue (cube_env.py:164-213); derived quantities are from the last substep's forward pass --- */
  const float *W = m->env_reward;
  float cp[3] = {xpos[3 * cube], xpos[3 * cube + 1], xpos[3 * cube + 2]};
  float sp[3] = {sx[3 * site], sx[3 * site + 1], sx[3 * site + 2]};
  float d0 = tp[0] - cp[0], d1 = tp[1] - cp[1], d2 = tp[2] - cp[2];
  const int sf = m->env_kind == ENV_AIRBOT_SF;
  float box_target_dis = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
  if (box_target_dis < W[4]) box_target_dis = 0.0f;          /* 0.005 (cube_env.py:166) / 0.003 (test/airbot.py:191) */
  float push_reward = 1.0f / (1.0f + 3.0f * box_target_dis);
  push_reward = push_reward * W[0];
  float task_complete = box_target_dis < W[4] ? W[5] : 0.0f;  /* test/airbot.py:196 */
  float old_ncp[2] = {ncp[0], ncp[1]};
  float site_z_reward = sp[2] < 0.82f ? 1.0f : 0.0f;
  float delta_x = tp[0] - cp[0], delta_y = tp[1] - cp[1];
  float angle = atan2f(delta_y, delta_x + 0.00001f);
  float distance = sqrtf(delta_x * delta_x + delta_y * delta_y) + 0.04f;
  float y_ = distance * sinf(angle), x_ = distance * cosf(angle);
  ncp[0] = delta_x - x_ + cp[0];
  ncp[1] = delta_y - y_ + cp[1];
  float e0 = sp[0] - old_ncp[0], e1 = sp[1] - old_ncp[1];
  float s2c = sqrtf(e0 * e0 + e1 * e1);
  s2c = s2c < 0.042f ? 0.0f : s2c - 0.042f;
  float siet = (1.0f - tanhf(5.0f * s2c)) * W[1];
  if (box_target_dis < 0.005f) siet = W[1];
  float hd = sp[2] < W[3] ? 1.0f : 0.0f;
  if (sf && (sp[0] > 1.0f || sp[0] < -0.6f || sp[1] > 0.3f || sp[1] < -0.3f || cp[2] < 0.6f)) hd = 1.0f;   /* test/airbot.py:227-233 */
  float health = W[2] * fabsf(hd - 1.0f);
  float reward = sf ? push_reward + siet + health + task_complete + site_z_reward : push_reward + siet + health + site_z_reward;
  float done = sf ? (box_target_dis < W[4] ? 1.0f : 0.0f) : (cp[2] < 0.6f ? 1.0f : 0.0f);
  reward = reward < -100.0f ? -100.0f : (reward > 100.0f ? 100.0f : reward);
  cube_obs(m, qpos, sp, tp, cp, ncp, &s->obs[e * m->obs_dim]);
  s->metrics[e * m->nmetrics + 0] = push_reward;
  s->metrics[e * m->nmetrics + 2] = siet;
  for (int i = 0; i < 3; i++) { s->info_site_pos[3 * e + i] = sp[i]; s->info_cube_pos[3 * e + i] = cp[i]; }
  s->reward[e] = reward; s->done[e] = done;
  /* --- EpisodeWrapper.step (action_repeat = 1) --- */
  if (wrap_episode) {
    float steps = s->info_steps[e] + 1.0f;
    int over = steps >= (float)m->episode_length;
    float done_env = s->done[e];
    s->info_truncation[e] = over ? 1.0f - done_env : 0.0f;
    s->info_steps[e] = steps;
    float prev_done = s->info_episode_done[e];
    float *em = &s->info_episode_metrics[e * (2 + m->nmetrics)];
    em[0] = prev_done != 0.0f ? 0.0f : em[0] + reward;
    em[1] = prev_done != 0.0f ? 0.0f : em[1] + 1.0f;
    for (int i = 0; i < m->nmetrics; i++) em[2 + i] = prev_done != 0.0f ? 0.0f : em[2 + i] + s->metrics[e * m->nmetrics + i];
    if (over) s->done[e] = 1.0f;
    s->info_episode_done[e] = s->done[e];
  }
  /* --- AutoResetWrapper.step: restore the cached first state where done --- */
  if (wrap_autoreset && s->done[e] != 0) {
    memcpy(qpos, &s->first_qpos[e * nq], sizeof(float) * (size_t)nq);
    memcpy(&s->qvel[e * m->nv], &s->first_qvel[e * m->nv], sizeof(float) * (size_t)m->nv);
    memcpy(&s->ctrl[e * nu], &s->first_ctrl[e * nu], sizeof(float) * (size_t)nu);
    memcpy(&s->qacc_warmstart[e * m->nv], &s->first_warmstart[e * m->nv], sizeof(float) * (size_t)m->nv);
    s->time[e] = s->first_time[e];
    memcpy(xpos, &s->first_xpos[e * m->nbody * 3], sizeof(float) * (size_t)m->nbody * 3);
    memcpy(sx, &s->first_site_xpos[e * m->nsite * 3], sizeof(float) * (size_t)m->nsite * 3);
    memcpy(&s->obs[e * m->obs_dim], &s->first_obs[e * m->obs_dim], sizeof(float) * (size_t)m->obs_dim);
    if ((m->env_kind == ENV_GO2 || m->env_kind == ENV_GO2_HANDSTAND) && s->priv_obs && s->first_priv_obs)
      memcpy(&s->priv_obs[(size_t)e * GO2_PRIV], &s->first_priv_obs[(size_t)e * GO2_PRIV], sizeof(float) * GO2_PRIV);
  }
}

/* ================================================================== T-shape env (T_shape_env.py) */
/* env_ids: 0 T body, 1 target body, 2 endpoint site, 3 T_tail site, 4 T_target_tail site, 5 base_block geom,
 * 6 vertical_block geom, 7 base_target geom, 8 vertical_target geom, 9..14 arm joint qposadr
 * env_reset: 0 noise, 1..6 arm offsets, 7..11 ctrl init, 12..13 new_T_pos0, 14 xita0
 * env_reward: 0 push weight, 1 siet weight, 2 healthy reward, 3 endpoint_min_z */
enum { TID_T = 0, TID_TARGET = 1, TID_SITE = 2, TID_TAIL = 3, TID_TTAIL = 4, TID_GBASE = 5, TID_GVERT = 6, TID_GTBASE = 7,
       TID_GTVERT = 8, TID_JOINTQ = 9 };

/* T_shape_env.py:223-234 */
static void tshape_obs(const omodel *m, const float *qpos, const float *site, const float *gbase, const float *gvert,
                       const float *tb, const float *tv, float xita, const float *newT, float *obs) {
  for (int i = 0; i < 6; i++) obs[i] = qpos[m->env_ids[TID_JOINTQ + i]];
  obs[6] = site[2];
  for (int i = 0; i < 3; i++) { obs[7 + i] = tb[i] - gbase[i]; obs[10 + i] = tv[i] - gvert[i]; }
  obs[13] = xita;
  obs[14] = newT[0] - site[0]; obs[15] = newT[1] - site[1];
}

static void wrappers_reset(const omodel *m, obatch *s, int e) {
  s->info_steps[e] = 0; s->info_truncation[e] = 0; s->info_episode_done[e] = 0;
  for (int i = 0; i < 2 + m->nmetrics; i++) s->info_episode_metrics[e * (2 + m->nmetrics) + i] = 0;
  memcpy(&s->first_qpos[e * m->nq], &s->qpos[e * m->nq], sizeof(float) * (size_t)m->nq);
  memcpy(&s->first_qvel[e * m->nv], &s->qvel[e * m->nv], sizeof(float) * (size_t)m->nv);
  memcpy(&s->first_ctrl[e * m->nu], &s->ctrl[e * m->nu], sizeof(float) * (size_t)m->nu);
  memcpy(&s->first_warmstart[e * m->nv], &s->qacc_warmstart[e * m->nv], sizeof(float) * (size_t)m->nv);
  s->first_time[e] = s->time[e];
  memcpy(&s->first_xpos[e * m->nbody * 3], &s->xpos[e * m->nbody * 3], sizeof(float) * (size_t)m->nbody * 3);
  memcpy(&s->first_site_xpos[e * m->nsite * 3], &s->site_xpos[e * m->nsite * 3], sizeof(float) * (size_t)m->nsite * 3);
  memcpy(&s->first_obs[e * m->obs_dim], &s->obs[e * m->obs_dim], sizeof(float) * (size_t)m->obs_dim);
  if ((m->env_kind == ENV_GO2 || m->env_kind == ENV_GO2_HANDSTAND) && s->priv_obs && s->first_priv_obs)
    memcpy(&s->first_priv_obs[(size_t)e * GO2_PRIV], &s->priv_obs[(size_t)e * GO2_PRIV], sizeof(float) * GO2_PRIV);
}

/* Episode + AutoReset post-step (brax.envs.training.wrap), shared by all envs */
static void wrappers_post(const omodel *m, obatch *s, int e, float reward) {
  const int wrap_episode = m->wrap_flags & 1, wrap_autoreset = (m->wrap_flags & 2) != 0;
  if (wrap_episode) {
    float steps = s->info_steps[e] + 1.0f;
    int over = steps >= (float)m->episode_length;
    float done_env = s->done[e];
    s->info_truncation[e] = over ? 1.0f - done_env : 0.0f;
    s->info_steps[e] = steps;
    float prev_done = s->info_episode_done[e];
    float *em = &s->info_episode_metrics[e * (2 + m->nmetrics)];
    em[0] = prev_done != 0.0f ? 0.0f : em[0] + reward;
    em[1] = prev_done != 0.0f ? 0.0f : em[1] + 1.0f;
    for (int i = 0; i < m->nmetrics; i++) em[2 + i] = prev_done != 0.0f ? 0.0f : em[2 + i] + s->metrics[e * m->nmetrics + i];
    if (over) s->done[e] = 1.0f;
    s->info_episode_done[e] = s->done[e];
  }
  if (wrap_autoreset && s->done[e] != 0) {
    memcpy(&s->qpos[e * m->nq], &s->first_qpos[e * m->nq], sizeof(float) * (size_t)m->nq);
    memcpy(&s->qvel[e * m->nv], &s->first_qvel[e * m->nv], sizeof(float) * (size_t)m->nv);
    memcpy(&s->ctrl[e * m->nu], &s->first_ctrl[e * m->nu], sizeof(float) * (size_t)m->nu);
    memcpy(&s->qacc_warmstart[e * m->nv], &s->first_warmstart[e * m->nv], sizeof(float) * (size_t)m->nv);
    s->time[e] = s->first_time[e];
    memcpy(&s->xpos[e * m->nbody * 3], &s->first_xpos[e * m->nbody * 3], sizeof(float) * (size_t)m->nbody * 3);
    memcpy(&s->site_xpos[e * m->nsite * 3], &s->first_site_xpos[e * m->nsite * 3], sizeof(float) * (size_t)m->nsite * 3);
    memcpy(&s->obs[e * m->obs_dim], &s->first_obs[e * m->obs_dim], sizeof(float) * (size_t)m->obs_dim);
    if ((m->env_kind == ENV_GO2 || m->env_kind == ENV_GO2_HANDSTAND) && s->priv_obs && s->first_priv_obs)
      memcpy(&s->priv_obs[(size_t)e * GO2_PRIV], &s->first_priv_obs[(size_t)e * GO2_PRIV], sizeof(float) * GO2_PRIV);
  }
}

/* T_shape_env.py:98-137 */
static void tshape_reset_env(const omodel *m, obatch *s, int e, const uint32_t key[2], odata *d) {
  const float *R = m->env_reset;
  uint32_t keys[5][2];
  oracle_split(key, 5, &keys[0][0]);
  float lo = -R[0], hi = R[0], u[NQ_MAX];
  oracle_uniform(keys[1], m->nq, &lo, &hi, 0, u);
  float qpos[NQ_MAX], qvel[NV_MAX], ctrl[NU_MAX];
  for (int i = 0; i < m->nq; i++) qpos[i] = m->qpos0[i] + u[i];
  for (int i = 0; i < 6; i++) qpos[m->env_ids[TID_JOINTQ + i]] += R[1 + i];
  oracle_uniform(keys[2], m->nv, &lo, &hi, 0, qvel);
  oracle_uniform(keys[3], m->nu, &lo, &hi, 0, u);
  for (int i = 0; i < m->nu; i++) ctrl[i] = R[7 + i] + u[i];
  s->time[e] = 0;
  for (int i = 0; i < m->nq; i++) s->qpos[e * m->nq + i] = qpos[i];
  for (int i = 0; i < m->nv; i++) { s->qvel[e * m->nv + i] = qvel[i]; s->qacc_warmstart[e * m->nv + i] = 0; }
  for (int i = 0; i < m->nu; i++) s->ctrl[e * m->nu + i] = 0;
  load_env(m, s, e, d);
  forward(m, d);
  for (int i = 0; i < m->nu; i++) d->ctrl[i] = ctrl[i];
  store_pipeline(m, s, e, d);
  const int *id = m->env_ids;
  float gb[3], gv[3];
  for (int i = 0; i < 3; i++) {
    s->info_target_base_pos[3 * e + i] = (float)d->geom_xpos[3 * id[TID_GTBASE] + i];
    s->info_target_vertical_pos[3 * e + i] = (float)d->geom_xpos[3 * id[TID_GTVERT] + i];
    s->info_site_pos[3 * e + i] = (float)d->site_xpos[3 * id[TID_SITE] + i];
    s->info_T_pos[3 * e + i] = (float)d->xpos[3 * id[TID_T] + i];
    gb[i] = (float)d->geom_xpos[3 * id[TID_GBASE] + i]; gv[i] = (float)d->geom_xpos[3 * id[TID_GVERT] + i];
  }
  s->info_target_w[e] = (float)d->xquat[4 * id[TID_TARGET]] * 10.0f;
  s->info_new_T_pos[2 * e] = R[12]; s->info_new_T_pos[2 * e + 1] = R[13];
  s->info_xita[e] = R[14];
  s->reward[e] = 0; s->done[e] = 0;
  for (int i = 0; i < m->nmetrics; i++) s->metrics[e * m->nmetrics + i] = 0;
  tshape_obs(m, &s->qpos[e * m->nq], &s->site_xpos[e * m->nsite * 3 + 3 * id[TID_SITE]], gb, gv, &s->info_target_base_pos[3 * e],
             &s->info_target_vertical_pos[3 * e], s->info_xita[e], &s->info_new_T_pos[2 * e], &s->obs[e * m->obs_dim]);
  wrappers_reset(m, s, e);
}

/* T_shape_env.py:139-221 */
static void tshape_step_env(const omodel *m, obatch *s, int e, const float *action, odata *d) {
  const int nq = m->nq, nu = m->nu;
  const int *id = m->env_ids;
  float *qpos = &s->qpos[e * nq], *sx = &s->site_xpos[e * m->nsite * 3];
  if ((m->wrap_flags & 2) != 0) { if (s->done[e] != 0) s->info_steps[e] = 0; s->done[e] = 0; }
  float act[NU_MAX];
  for (int i = 0; i < nu; i++) {
    volatile float delta = m->env_action_scale[i] * action[e * nu + i];
    act[i] = s->ctrl[e * nu + i] + delta;
  }
  {
    volatile float t1 = 1.57f + qpos[id[TID_JOINTQ + 1]];
    volatile float t2 = t1 + qpos[id[TID_JOINTQ + 2]];
    act[3] = -t2;
  }
  {   /* wrist aims from the end effector at the T's tail site (both from the previous forward pass) */
    float dx = sx[3 * id[TID_TAIL]] - sx[3 * id[TID_SITE]], dy = sx[3 * id[TID_TAIL] + 1] - sx[3 * id[TID_SITE] + 1];
    volatile float dxe = dx + 0.00001f;
    float ang = atan2f(dy, dxe);
    volatile float t1 = -ang + act[0];
    act[4] = t1 + 1.5708f;
  }
  for (int i = 0; i < nu; i++) act[i] = act[i] < m->env_ctrl_lo[i] ? m->env_ctrl_lo[i] : (act[i] > m->env_ctrl_hi[i] ? m->env_ctrl_hi[i] : act[i]);
  load_env(m, s, e, d);
  for (int i = 0; i < nu; i++) d->ctrl[i] = act[i];
  for (int f = 0; f < m->n_frames; f++) step_physics(m, d);
  store_pipeline(m, s, e, d);
  const float *W = m->env_reward;
  float gb[3], gv[3], sp[3], tail[3], ttail[3];
  for (int i = 0; i < 3; i++) {
    gb[i] = (float)d->geom_xpos[3 * id[TID_GBASE] + i]; gv[i] = (float)d->geom_xpos[3 * id[TID_GVERT] + i];
    sp[i] = sx[3 * id[TID_SITE] + i]; tail[i] = sx[3 * id[TID_TAIL] + i]; ttail[i] = sx[3 * id[TID_TTAIL] + i];
  }
  const float *tb = &s->info_target_base_pos[3 * e], *tv = &s->info_target_vertical_pos[3 * e];
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
  float c = dotp / (nb * nt);
  c = c < -1.0f ? -1.0f : (c > 1.0f ? 1.0f : c);
  float xita = acosf(c);
  s->info_xita[e] = xita;
  float push_w = 1.0f / (1.0f + 6.0f * xita);
  float push_reward = (0.1515f * push_base + 0.1515f * push_vert + 0.66f * push_w) * W[0];
  float old_newT[2] = {s->info_new_T_pos[2 * e], s->info_new_T_pos[2 * e + 1]};
  float site_z_reward = sp[2] < 0.83f ? 1.0f : 0.0f;
  float z_dis = fabsf(sp[2] - 0.805f);
  float z_reward = 4.0f / (1.0f + 3.0f * z_dis);
  site_z_reward = site_z_reward + z_reward;
  float delta_x = ttail[0] - tail[0], delta_y = ttail[1] - tail[1];
  float angle = atan2f(delta_y, delta_x + 0.00001f);
  float distance = sqrtf(delta_x * delta_x + delta_y * delta_y) + 0.025f;
  float y_ = distance * sinf(angle), x_ = distance * cosf(angle);
  s->info_new_T_pos[2 * e] = delta_x - x_ + tail[0];
  s->info_new_T_pos[2 * e + 1] = delta_y - y_ + tail[1];
  float e0 = sp[0] - old_newT[0], e1 = sp[1] - old_newT[1];
  float s2c = sqrtf(e0 * e0 + e1 * e1);
  s2c = s2c < 0.02f ? 0.0f : s2c - 0.02f;
  float siet = (1.0f - tanhf(5.0f * s2c)) * W[1];
  float health = W[2] * fabsf((sp[2] < W[3] ? 1.0f : 0.0f) - 1.0f);
  float reward = push_reward + siet + health + site_z_reward;
  float done = s->xpos[e * m->nbody * 3 + 3 * id[TID_T] + 2] < 0.6f ? 1.0f : 0.0f;
  reward = reward < -100.0f ? -100.0f : (reward > 100.0f ? 100.0f : reward);
  tshape_obs(m, qpos, sp, gb, gv, tb, tv, xita, &s->info_new_T_pos[2 * e], &s->obs[e * m->obs_dim]);
  float *met = &s->metrics[e * m->nmetrics];
  met[0] = push_reward; met[1] = siet; met[2] = health; met[4] = site_z_reward;     /* task_complete_reward stays 0 */
  for (int i = 0; i < 3; i++) { s->info_site_pos[3 * e + i] = sp[i]; s->info_T_pos[3 * e + i] = s->xpos[e * m->nbody * 3 + 3 * id[TID_T] + i]; }
  s->reward[e] = reward; s->done[e] = done;
  wrappers_post(m, s, e, reward);
}

/* ================================================================== Go2 joystick env (go2/joystick.py) */
/* info_go2 layout (floats; integers are stored as exact floats, the PRNG key as raw bits) */
enum { G2_CMD = 0, G2_STEPS_CMD = 3, G2_LAST_ACT = 4, G2_LAST_LAST_ACT = 16, G2_AIR = 28, G2_CONTACT_T = 32, G2_LAST_CONTACT = 36,
       G2_SWING = 40, G2_ACT_BUF = 44, G2_GYRO_BUF = 92, G2_LINVEL_BUF = 104, G2_GRAV_BUF = 116, G2_STEPS_PERT = 128,
       G2_PERT_DUR_S = 129, G2_PERT_DUR = 130, G2_SINCE_PERT = 131, G2_PERT_STEPS = 132, G2_PERT_DIR = 133, G2_PERT_MAG = 136,
       G2_RNG = 137, G2_XFRC = 139 /* data.xfrc_applied[torso, :3]; zero again on auto-reset (it is part of `data`) */ };
/* env_go2f: 0 ctrl_dt, 1 action_scale, 2 noise level, 3 n_joint_pos, 4 n_joint_vel, 5 n_gyro, 6 n_gravity, 7 n_linvel,
 *           8 tracking_sigma, 9 max_foot_height, 10..12 cmd_a, 13..15 cmd_b, 16 change_interval, 17..18 kick_wait,
 *           19..20 kick_durations, 21..22 velocity_kick, 23 torso subtree mass (body_subtreemass, joystick.py:104)
 * env_go2i: 0 action delay steps, 1 imu delay steps, 2 pert enable
 * env_ids : 0 imu site, 1..4 feet sites (FR FL RR RL), 5 floor geom, 6..9 feet geoms, 10 torso body
 * env_go2_scales: reward scales in reward_config order (= metrics order); env_go2_home: keyframe "home" qpos;
 * env_go2_soft: soft lower (12) then upper (12) joint limits */
enum { RW_TRACK_LIN = 0, RW_TRACK_ANG, RW_LIN_VEL_Z, RW_ANG_VEL_XY, RW_ORIENT, RW_DOF_LIMITS, RW_POSE, RW_TERM, RW_STAND_STILL,
       RW_TORQUES, RW_ACTION_RATE, RW_ENERGY, RW_FEET_CLEAR, RW_FEET_HEIGHT, RW_FEET_SLIP, RW_FEET_AIR, RW_ALL_FEET_AIR,
       RW_SYM_GAIT, RW_LR_SYM, RW_FB_SYM, RW_FEET_OFF_STILL, RW_COUNT };

static inline void g2_get_key(const float *info, uint32_t key[2]) { memcpy(key, &info[G2_RNG], 8); }
static inline void g2_set_key(float *info, const uint32_t key[2]) { memcpy(&info[G2_RNG], key, 8); }
static float g2_uniform1(const uint32_t key[2], float lo, float hi) { float v; oracle_uniform(key, 1, &lo, &hi, 0, &v); return v; }

typedef struct { float gyro[3], local_linvel[3], gravity[3], upvector[3], global_linvel[3], global_angvel[3], feet_vel[4][3],
                 feet_z[4], act_force[NU_MAX], accel[3]; int contact[4]; } g2_sensors;

static void g2_read_sensors(const omodel *m, const odata *d, g2_sensors *o) {
  const int imu = m->env_ids[0];
  const real *R = &d->site_xmat[9 * imu];
  real t[3], g[3] = {0, 0, -1};
  mat_tmulv(t, R, &d->site_angvel[3 * imu]); for (int c = 0; c < 3; c++) o->gyro[c] = (float)t[c];
  mat_tmulv(t, R, &d->site_linvel[3 * imu]); for (int c = 0; c < 3; c++) o->local_linvel[c] = (float)t[c];
  mat_tmulv(t, R, g); for (int c = 0; c < 3; c++) o->gravity[c] = (float)t[c];
  for (int c = 0; c < 3; c++) {
    o->upvector[c] = (float)R[3 * c + 2];
    o->global_linvel[c] = (float)d->site_linvel[3 * imu + c];
    o->global_angvel[c] = (float)d->site_angvel[3 * imu + c];
  }
  for (int f = 0; f < 4; f++) {
    int sid = m->env_ids[1 + f];
    for (int c = 0; c < 3; c++) o->feet_vel[f][c] = (float)d->site_linvel[3 * sid + c];
    o->feet_z[f] = (float)d->site_xpos[3 * sid + 2];
    o->contact[f] = 0;
  }
  for (int u = 0; u < m->nu; u++) o->act_force[u] = (float)d->actuator_force[u];
  for (int c = 0; c < 3; c++) o->accel[c] = (float)d->acc_site[c];
  /* collision.geoms_colliding(data, foot, floor): a contact of that pair with dist < 0 (_src/collision.py:6-18) */
  for (int i = 0; i < d->ncon; i++) {
    if (!(d->con[i].dist < 0)) continue;
    int g1 = m->pair_geom1[d->con[i].pair], g2 = m->pair_geom2[d->con[i].pair];
    for (int f = 0; f < 4; f++)
      if ((g1 == m->env_ids[5] && g2 == m->env_ids[6 + f]) || (g2 == m->env_ids[5] && g1 == m->env_ids[6 + f])) o->contact[f] = 1;
  }
}

/* joystick.py:284-340: the 48-dim "state" observation; advances info.rng by five splits */
static void g2_obs(const omodel *m, float *info, const float *qpos, const float *qvel, const g2_sensors *sn, float *obs) {
  const float *F = m->env_go2f;
  const int imu_delay = m->env_go2i[1];
  const float *gyro = imu_delay ? &info[G2_GYRO_BUF] : sn->gyro;
  const float *linvel = imu_delay ? &info[G2_LINVEL_BUF] : sn->local_linvel;
  const float *gravity = imu_delay ? &info[G2_GRAV_BUF] : sn->gravity;
  uint32_t rng[2], ks[2][2];
  g2_get_key(info, rng);
  float zero = 0.0f, one = 1.0f, u[12];
  float ngyro[3], ngrav[3], nlin[3], nq[12], nv[12];
#define NOISE(dst, src, n, scale) do { \
    oracle_split(rng, 2, &ks[0][0]); rng[0] = ks[0][0]; rng[1] = ks[0][1]; \
    oracle_uniform(ks[1], n, &zero, &one, 0, u); \
    for (int i_ = 0; i_ < n; i_++) { volatile float a_ = 2.0f * u[i_]; volatile float b_ = a_ - 1.0f; volatile float c_ = b_ * F[2]; \
      volatile float e_ = c_ * (scale); dst[i_] = (src)[i_] + e_; } } while (0)
  NOISE(ngyro, gyro, 3, F[5]);
  NOISE(ngrav, gravity, 3, F[6]);
  NOISE(nlin, linvel, 3, F[7]);
  NOISE(nq, qpos + 7, 12, F[3]);
  NOISE(nv, qvel + 6, 12, F[4]);
#undef NOISE
  g2_set_key(info, rng);
  for (int i = 0; i < 3; i++) { obs[i] = nlin[i]; obs[3 + i] = ngyro[i]; obs[6 + i] = ngrav[i]; obs[45 + i] = info[G2_CMD + i]; }
  for (int i = 0; i < 12; i++) { obs[9 + i] = nq[i] - m->env_go2_home[7 + i]; obs[21 + i] = nv[i]; obs[33 + i] = info[G2_LAST_ACT + i]; }
}

/* joystick.py:123-203 */
/* joystick.py:341-366: obs["privileged_state"] = state + un-noised, un-delayed readings.  Called right after g2_obs. */
static void g2_priv_obs(const omodel *m, const float *info, const float *qpos, const float *qvel, const g2_sensors *sn, const float *obs,
                        float *p) {
  int k = 0;
  for (int i = 0; i < 48; i++) p[k++] = obs[i];
  for (int i = 0; i < 3; i++) p[k++] = sn->gyro[i];
  for (int i = 0; i < 3; i++) p[k++] = sn->accel[i];
  for (int i = 0; i < 3; i++) p[k++] = sn->gravity[i];
  for (int i = 0; i < 3; i++) p[k++] = sn->local_linvel[i];
  for (int i = 0; i < 3; i++) p[k++] = sn->global_angvel[i];
  for (int i = 0; i < 12; i++) p[k++] = qpos[7 + i] - m->env_go2_home[7 + i];
  for (int i = 0; i < 12; i++) p[k++] = qvel[6 + i];
  for (int i = 0; i < 12; i++) p[k++] = sn->act_force[i];
  for (int f = 0; f < 4; f++) p[k++] = info[G2_LAST_CONTACT + f];
  for (int f = 0; f < 4; f++) for (int c = 0; c < 3; c++) p[k++] = sn->feet_vel[f][c];
  for (int f = 0; f < 4; f++) p[k++] = info[G2_AIR + f];
  for (int c = 0; c < 3; c++) p[k++] = info[G2_XFRC + c];
  p[k++] = info[G2_SINCE_PERT] >= info[G2_STEPS_PERT] ? 1.0f : 0.0f;
}

/* joystick.py:594-644 _maybe_apply_perturbation: a half-sine force pulse on the torso every kick_wait_times seconds */
static void g2_maybe_perturb(const omodel *m, float *info) {
  const float *F = m->env_go2f;
  if (info[G2_SINCE_PERT] >= info[G2_STEPS_PERT]) {
    float t = info[G2_PERT_STEPS] * F[0];
    volatile float ph = 3.14159265358979323846f * t;
    float u_t = 0.5f * sinf(ph / info[G2_PERT_DUR_S]);
    volatile float f1 = u_t * F[23]; volatile float f2 = f1 * info[G2_PERT_MAG];
    float force = f2 / info[G2_PERT_DUR_S];
    for (int c = 0; c < 3; c++) info[G2_XFRC + c] = force * info[G2_PERT_DIR + c];
    if (info[G2_PERT_STEPS] >= info[G2_PERT_DUR]) info[G2_SINCE_PERT] = 0.0f;
    info[G2_PERT_STEPS] += 1.0f;
  } else {
    uint32_t rng[2], ks[2][2];
    g2_get_key(info, rng);
    oracle_split(rng, 2, &ks[0][0]);
    g2_set_key(info, ks[0]);
    info[G2_SINCE_PERT] += 1.0f;
    for (int c = 0; c < 3; c++) info[G2_XFRC + c] = 0.0f;
    if (info[G2_SINCE_PERT] >= info[G2_STEPS_PERT]) {
      float angle = g2_uniform1(ks[1], 0.0f, 6.2831855f);
      info[G2_PERT_STEPS] = 0.0f;
      info[G2_PERT_DIR] = cosf(angle); info[G2_PERT_DIR + 1] = sinf(angle); info[G2_PERT_DIR + 2] = 0.0f;
    }
  }
}

static void go2_reset_env(const omodel *m, obatch *s, int e, const uint32_t key[2], odata *d) {
  const float *F = m->env_go2f;
  float *info = &s->info_go2[(size_t)e * GO2_INFO];
  memset(info, 0, sizeof(float) * GO2_INFO);
  uint32_t rng[2] = {key[0], key[1]}, ks[4][2];
  float qpos[NQ_MAX], qvel[NV_MAX];
  for (int i = 0; i < m->nq; i++) qpos[i] = m->env_go2_home[i];
  for (int i = 0; i < m->nv; i++) qvel[i] = 0;
  float lo = -0.5f, hi = 0.5f, u[8];
  oracle_split(rng, 2, &ks[0][0]); rng[0] = ks[0][0]; rng[1] = ks[0][1];
  oracle_uniform(ks[1], 2, &lo, &hi, 0, u);
  qpos[0] += u[0]; qpos[1] += u[1];
  oracle_split(rng, 2, &ks[0][0]); rng[0] = ks[0][0]; rng[1] = ks[0][1];
  float yaw = g2_uniform1(ks[1], -3.14f, 3.14f);
  {
    float sn = sinf(yaw * 0.5f), cs = cosf(yaw * 0.5f);
    float q[4] = {qpos[3], qpos[4], qpos[5], qpos[6]}, r[4] = {cs, 0.0f * sn, 0.0f * sn, 1.0f * sn};
    qpos[3] = q[0] * r[0] - q[1] * r[1] - q[2] * r[2] - q[3] * r[3];
    qpos[4] = q[0] * r[1] + q[1] * r[0] + q[2] * r[3] - q[3] * r[2];
    qpos[5] = q[0] * r[2] - q[1] * r[3] + q[2] * r[0] + q[3] * r[1];
    qpos[6] = q[0] * r[3] + q[1] * r[2] - q[2] * r[1] + q[3] * r[0];
  }
  oracle_split(rng, 2, &ks[0][0]); rng[0] = ks[0][0]; rng[1] = ks[0][1];
  oracle_uniform(ks[1], 6, &lo, &hi, 0, u);
  for (int i = 0; i < 6; i++) qvel[i] = u[i];
  /* mjx_env.init(qpos, qvel, ctrl = qpos[7:]) = make_data + forward (_src/mjx_env.py:30-54) */
  s->time[e] = 0;
  for (int i = 0; i < m->nq; i++) s->qpos[e * m->nq + i] = qpos[i];
  for (int i = 0; i < m->nv; i++) { s->qvel[e * m->nv + i] = qvel[i]; s->qacc_warmstart[e * m->nv + i] = 0; }
  for (int i = 0; i < m->nu; i++) s->ctrl[e * m->nu + i] = qpos[7 + i];
  load_env(m, s, e, d);
  d->acc_site_id = m->env_ids[0]; d->xfrc_body = 0;
  forward(m, d);
  store_pipeline(m, s, e, d);
  oracle_split(rng, 4, &ks[0][0]); rng[0] = ks[0][0]; rng[1] = ks[0][1];
  float t_pert = g2_uniform1(ks[1], F[17], F[18]);
  info[G2_STEPS_PERT] = rintf(t_pert / F[0]);
  info[G2_PERT_DUR_S] = g2_uniform1(ks[2], F[19], F[20]);
  info[G2_PERT_DUR] = rintf(info[G2_PERT_DUR_S] / F[0]);
  info[G2_PERT_MAG] = g2_uniform1(ks[3], F[21], F[22]);
  oracle_split(rng, 3, &ks[0][0]); rng[0] = ks[0][0]; rng[1] = ks[0][1];
  {
    float uu = g2_uniform1(ks[1], 0.0f, 1.0f);
    float t_cmd = -log1pf(-uu) * F[16];
    info[G2_STEPS_CMD] = rintf(t_cmd / F[0]);
    float a_lo[3] = {-F[10], -F[11], -F[12]}, a_hi[3] = {F[10], F[11], F[12]};
    oracle_uniform(ks[2], 3, a_lo, a_hi, 1, &info[G2_CMD]);
  }
  g2_set_key(info, rng);
  for (int i = 0; i < m->nmetrics; i++) s->metrics[e * m->nmetrics + i] = 0;
  g2_sensors sn;
  g2_read_sensors(m, d, &sn);
  g2_obs(m, info, &s->qpos[e * m->nq], &s->qvel[e * m->nv], &sn, &s->obs[e * m->obs_dim]);
  if (s->priv_obs) g2_priv_obs(m, info, &s->qpos[e * m->nq], &s->qvel[e * m->nv], &sn, &s->obs[e * m->obs_dim], &s->priv_obs[(size_t)e * GO2_PRIV]);
  s->reward[e] = 0; s->done[e] = 0;
  wrappers_reset(m, s, e);
}

static float g2_norm3(const float *v) { return sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

/* joystick.py:204-280 with rewards :367-593 and sample_command :645-653 */
static void go2_step_env(const omodel *m, obatch *s, int e, const float *action, odata *d) {
  const float *F = m->env_go2f, *SC = m->env_go2_scales, *home = m->env_go2_home;
  const int nu = m->nu;
  const float dt = F[0];
  float *info = &s->info_go2[(size_t)e * GO2_INFO];
  const float *act_in = &action[e * nu];
  if ((m->wrap_flags & 2) != 0) { if (s->done[e] != 0) s->info_steps[e] = 0; s->done[e] = 0; }
  if (m->env_go2i[2]) g2_maybe_perturb(m, info);
  /* action delay FIFO (:207-215) */
  float actual[NU_MAX];
  const int adel = m->env_go2i[0];
  if (adel > 0) {
    for (int i = 0; i < nu; i++) actual[i] = info[G2_ACT_BUF + i];
    memmove(&info[G2_ACT_BUF], &info[G2_ACT_BUF + nu], sizeof(float) * (size_t)adel * nu);
    for (int i = 0; i < nu; i++) info[G2_ACT_BUF + adel * nu + i] = act_in[i];
  } else for (int i = 0; i < nu; i++) actual[i] = act_in[i];
  load_env(m, s, e, d);
  for (int i = 0; i < nu; i++) { volatile float sc = actual[i] * F[1]; d->ctrl[i] = home[7 + i] + sc; }
  d->acc_site_id = m->env_ids[0];
  d->xfrc_body = m->env_go2i[2] ? m->env_ids[10] : 0;
  for (int c = 0; c < 3; c++) d->xfrc_force[c] = info[G2_XFRC + c];
  for (int f = 0; f < m->n_frames; f++) step_physics(m, d);
  store_pipeline(m, s, e, d);
  const float *qpos = &s->qpos[e * m->nq], *qvel = &s->qvel[e * m->nv];
  g2_sensors sn;
  g2_read_sensors(m, d, &sn);
  /* IMU delay FIFOs (:220-235) */
  const int idel = m->env_go2i[1];
  if (idel > 0) {
    float *bufs[3] = {&info[G2_GYRO_BUF], &info[G2_LINVEL_BUF], &info[G2_GRAV_BUF]};
    const float *cur[3] = {sn.gyro, sn.local_linvel, sn.gravity};
    for (int b = 0; b < 3; b++) {
      memmove(bufs[b], bufs[b] + 3, sizeof(float) * (size_t)idel * 3);
      for (int c = 0; c < 3; c++) bufs[b][idel * 3 + c] = cur[b][c];
    }
  }
  int contact[4], contact_filt[4], first_contact[4];
  for (int f = 0; f < 4; f++) {
    contact[f] = sn.contact[f];
    contact_filt[f] = contact[f] || info[G2_LAST_CONTACT + f] != 0;
    first_contact[f] = (info[G2_AIR + f] > 0.0f) && contact_filt[f];
    info[G2_AIR + f] += dt;
    if (sn.feet_z[f] > info[G2_SWING + f]) info[G2_SWING + f] = sn.feet_z[f];
  }
  g2_obs(m, info, qpos, qvel, &sn, &s->obs[e * m->obs_dim]);
  if (s->priv_obs) g2_priv_obs(m, info, qpos, qvel, &sn, &s->obs[e * m->obs_dim], &s->priv_obs[(size_t)e * GO2_PRIV]);
  float done = sn.upvector[2] < 0.0f ? 1.0f : 0.0f;
  /* ---- reward terms ---- */
  const float *cmd = &info[G2_CMD];
  float cmd_norm = g2_norm3(cmd);
  float moving = cmd_norm > 0.01f ? 1.0f : 0.0f, still = cmd_norm < 0.01f ? 1.0f : 0.0f;
  float rw[RW_COUNT];
  {
    float e0 = cmd[0] - sn.local_linvel[0], e1 = cmd[1] - sn.local_linvel[1];
    rw[RW_TRACK_LIN] = expf(-(e0 * e0 + e1 * e1) / F[8]);
    float ea = cmd[2] - sn.gyro[2];
    rw[RW_TRACK_ANG] = expf(-(ea * ea) / F[8]);
  }
  rw[RW_LIN_VEL_Z] = sn.global_linvel[2] * sn.global_linvel[2];
  rw[RW_ANG_VEL_XY] = sn.global_angvel[0] * sn.global_angvel[0] + sn.global_angvel[1] * sn.global_angvel[1];
  rw[RW_ORIENT] = sn.upvector[0] * sn.upvector[0] + sn.upvector[1] * sn.upvector[1];
  {
    float sa = 0, lim = 0, pose = 0;
    for (int i = 0; i < 12; i++) {
      float q = qpos[7 + i], dq = q - home[7 + i];
      sa += fabsf(dq);
      float lo_ = q - m->env_go2_soft[i], hi_ = q - m->env_go2_soft[12 + i];
      lim += -(lo_ < 0.0f ? lo_ : 0.0f) + (hi_ > 0.0f ? hi_ : 0.0f);
      float w = (i % 3 == 2) ? 0.1f : 1.0f;
      pose += dq * dq * w;
    }
    rw[RW_STAND_STILL] = sa * still;
    rw[RW_DOF_LIMITS] = lim;
    rw[RW_POSE] = expf(-pose);
  }
  rw[RW_TERM] = done;
  {
    float s2 = 0, s1 = 0, en = 0;
    for (int i = 0; i < 12; i++) { float t = sn.act_force[i]; s2 += t * t; s1 += fabsf(t); en += fabsf(qvel[6 + i]) * fabsf(t); }
    rw[RW_TORQUES] = sqrtf(s2) + s1;
    rw[RW_ENERGY] = en;
    float ar = 0;
    for (int i = 0; i < 12; i++) { float dd = act_in[i] - info[G2_LAST_ACT + i]; ar += dd * dd; }
    rw[RW_ACTION_RATE] = ar;
  }
  {
    float slip = 0, clear = 0, height = 0, air = 0; int nair = 0;
    for (int f = 0; f < 4; f++) {
      float vx = sn.feet_vel[f][0], vy = sn.feet_vel[f][1];
      float v2 = vx * vx + vy * vy;
      slip += v2 * (float)contact[f];
      clear += fabsf(sn.feet_z[f] - F[9]) * sqrtf(sqrtf(v2));
      float err = info[G2_SWING + f] / F[9] - 1.0f;
      height += err * err * (float)first_contact[f];
      air += (info[G2_AIR + f] - 0.1f) * (float)first_contact[f];
      nair += !contact[f];
    }
    rw[RW_FEET_SLIP] = slip * moving;
    rw[RW_FEET_CLEAR] = clear;
    rw[RW_FEET_HEIGHT] = height * moving;
    rw[RW_FEET_AIR] = air * moving;
    rw[RW_ALL_FEET_AIR] = (nair >= 3 ? 1.0f : 0.0f) * moving;
    rw[RW_FEET_OFF_STILL] = (float)nair * still;
    float p1 = 0, p2 = 0;
    for (int i = 0; i < 3; i++) { float a = qpos[7 + 3 + i] - qpos[7 + 6 + i], b = qpos[7 + i] - qpos[7 + 9 + i]; p1 += a * a; p2 += b * b; }
    rw[RW_SYM_GAIT] = (p1 + p2) * moving;
    const float *at = &info[G2_AIR], *ct = &info[G2_CONTACT_T];
    float la = (at[1] + at[3]) / 2.0f, lc = (ct[1] + ct[3]) / 2.0f, ra = (at[0] + at[2]) / 2.0f, rc = (ct[0] + ct[2]) / 2.0f;
    rw[RW_LR_SYM] = ((la - ra) * (la - ra) + (lc - rc) * (lc - rc)) * moving;
    float fa = (at[0] + at[1]) / 2.0f, fc = (ct[0] + ct[1]) / 2.0f, ba = (at[2] + at[3]) / 2.0f, bc = (ct[2] + ct[3]) / 2.0f;
    rw[RW_FB_SYM] = ((fa - ba) * (fa - ba) + (fc - bc) * (fc - bc)) * moving;
  }
  for (int k = 0; k < RW_COUNT; k++) { volatile float v = rw[k] * SC[k]; rw[k] = v; }
  /* sum(rewards.values()) in the insertion order of _get_reward's dict (:378-423) */
  static const int order[RW_COUNT] = {RW_TRACK_LIN, RW_TRACK_ANG, RW_LIN_VEL_Z, RW_ANG_VEL_XY, RW_ORIENT, RW_STAND_STILL, RW_TERM, RW_POSE,
                                      RW_TORQUES, RW_ACTION_RATE, RW_ENERGY, RW_FEET_SLIP, RW_FEET_CLEAR, RW_FEET_HEIGHT, RW_FEET_AIR,
                                      RW_DOF_LIMITS, RW_ALL_FEET_AIR, RW_SYM_GAIT, RW_LR_SYM, RW_FB_SYM, RW_FEET_OFF_STILL};
  volatile float total = 0.0f;
  for (int k = 0; k < RW_COUNT; k++) total = total + rw[order[k]];
  float reward = total * dt;
  reward = reward < 0.0f ? 0.0f : (reward > 10000.0f ? 10000.0f : reward);
  /* ---- bookkeeping (:255-277) ---- */
  for (int i = 0; i < 12; i++) { info[G2_LAST_LAST_ACT + i] = info[G2_LAST_ACT + i]; info[G2_LAST_ACT + i] = act_in[i]; }
  info[G2_STEPS_CMD] -= 1.0f;
  uint32_t rng[2], ks[4][2];
  g2_get_key(info, rng);
  oracle_split(rng, 3, &ks[0][0]);
  g2_set_key(info, ks[0]);
  {   /* sample_command(key1, command) */
    uint32_t k4[4][2];
    oracle_split(ks[1], 4, &k4[0][0]);          /* rng, y_rng, w_rng, z_rng */
    float a_lo[3] = {-F[10], -F[11], -F[12]}, a_hi[3] = {F[10], F[11], F[12]}, y[3], uz[3], uw[3], zero = 0.0f, one = 1.0f;
    oracle_uniform(k4[1], 3, a_lo, a_hi, 1, y);
    oracle_uniform(k4[3], 3, &zero, &one, 0, uz);
    oracle_uniform(k4[2], 3, &zero, &one, 0, uw);
    if (info[G2_STEPS_CMD] <= 0.0f)
      for (int i = 0; i < 3; i++) {
        float z = uz[i] < F[13 + i] ? 1.0f : 0.0f, w = uw[i] < 0.5f ? 1.0f : 0.0f;
        volatile float yz = y[i] * z; volatile float dif = info[G2_CMD + i] - yz; volatile float wd = w * dif;
        info[G2_CMD + i] = info[G2_CMD + i] - wd;
      }
  }
  if (done != 0.0f || info[G2_STEPS_CMD] <= 0.0f) {
    float uu = g2_uniform1(ks[2], 0.0f, 1.0f);
    volatile float t1 = -log1pf(-uu) * F[16];
    info[G2_STEPS_CMD] = rintf(t1 / dt);
  }
  float swing_mean = 0;
  for (int f = 0; f < 4; f++) {
    float c = (float)contact[f], nc = 1.0f - c;
    info[G2_AIR + f] = (info[G2_AIR + f] + dt) * nc;
    info[G2_CONTACT_T + f] = (info[G2_CONTACT_T + f] + dt) * c;
    info[G2_LAST_CONTACT + f] = c;
    info[G2_SWING + f] *= nc;
    swing_mean += info[G2_SWING + f];
  }
  float *met = &s->metrics[e * m->nmetrics];
  for (int k = 0; k < RW_COUNT; k++) met[k] = rw[k];
  met[RW_COUNT] = swing_mean / 4.0f;
  s->reward[e] = reward; s->done[e] = done;
  wrappers_post(m, s, e, reward);
  if ((m->wrap_flags & 2) != 0 && s->done[e] != 0) for (int c = 0; c < 3; c++) info[G2_XFRC + c] = 0.0f;   /* data <- first data */
}


/* ================================================================== Go2 Handstand / Footstand (go2/handstand.py)
 * env_ids: 0 imu site, 1 floor geom, 2..13 the twelve "unwanted contact" geoms (:97-113), 14..15 the feet geoms of the contact cost
 * (:115-118), 16 trunk body.  env_go2f: ctrl_dt, action_scale, noise level, scales joint_pos / joint_vel / gyro / gravity / linvel,
 * init_from_crouch, energy_termination_threshold, z_des, desired forward vector (3).  env_go2i: joint_ids (6) of the pose cost.
 * env_go2_scales: the eleven reward scales in config order.  env_go2_home: home qpos (19) | pre_recovery qpos (19).
 * env_go2_soft: soft lower (12) | soft upper (12) joint limits.  info block: step at 0, last_act at 4..15, rng at 137. */
enum { HS_STEP = 0, HS_LAST_ACT = 4, HS_OBS = 45, HS_PRIV = 94 };
enum { HM_HEIGHT = 0, HM_ORIENT, HM_CONTACT, HM_ACTION_RATE, HM_TERM, HM_DOF_LIMITS, HM_TORQUES, HM_POSE, HM_STAY_STILL, HM_ENERGY, HM_DOF_ACC, HM_COUNT };

static int hs_pair_touching(const omodel *m, const odata *d, int geom) {      /* collision.geoms_colliding(data, geom, floor) */
  const int floor_g = m->env_ids[1];
  for (int i = 0; i < d->ncon; i++) {
    if (!(d->con[i].dist < 0)) continue;
    int g1 = m->pair_geom1[d->con[i].pair], g2 = m->pair_geom2[d->con[i].pair];
    if ((g1 == floor_g && g2 == geom) || (g2 == floor_g && g1 == geom)) return 1;
  }
  return 0;
}

/* handstand.py:196-263: obs["state"] (45) and obs["privileged_state"] (94); advances info.rng by five splits */
static void hs_obs(const omodel *m, float *info, const float *qpos, const float *qvel, const g2_sensors *sn, float torso_height,
                   float *obs, float *priv) {
  const float *F = m->env_go2f, *home = m->env_go2_home;
  uint32_t rng[2], ks[2][2];
  g2_get_key(info, rng);
  float zero = 0.0f, one = 1.0f, u[12];
  float ngyro[3], ngrav[3], nlin[3], nq[12], nv[12];
#define NOISE(dst, src, n, scale) do { \
    oracle_split(rng, 2, &ks[0][0]); rng[0] = ks[0][0]; rng[1] = ks[0][1]; \
    oracle_uniform(ks[1], n, &zero, &one, 0, u); \
    for (int i_ = 0; i_ < n; i_++) { volatile float a_ = 2.0f * u[i_]; volatile float b_ = a_ - 1.0f; volatile float c_ = b_ * F[2]; \
      volatile float e_ = c_ * (scale); dst[i_] = (src)[i_] + e_; } } while (0)
  NOISE(ngyro, sn->gyro, 3, F[5]);
  NOISE(ngrav, sn->gravity, 3, F[6]);
  NOISE(nq, qpos + 7, 12, F[3]);
  NOISE(nv, qvel + 6, 12, F[4]);
  NOISE(nlin, sn->local_linvel, 3, F[7]);
#undef NOISE
  g2_set_key(info, rng);
  for (int i = 0; i < 3; i++) { obs[i] = nlin[i]; obs[3 + i] = ngyro[i]; obs[6 + i] = ngrav[i]; }
  for (int i = 0; i < 12; i++) { obs[9 + i] = nq[i] - home[7 + i]; obs[21 + i] = nv[i]; obs[33 + i] = info[HS_LAST_ACT + i]; }
  if (priv) {
    int k = 0;
    for (int i = 0; i < HS_OBS; i++) priv[k++] = obs[i];
    for (int i = 0; i < 3; i++) priv[k++] = sn->gyro[i];
    for (int i = 0; i < 3; i++) priv[k++] = sn->accel[i];
    for (int i = 0; i < 3; i++) priv[k++] = sn->local_linvel[i];
    for (int i = 0; i < 3; i++) priv[k++] = sn->global_angvel[i];
    for (int i = 0; i < 12; i++) priv[k++] = qpos[7 + i];
    for (int i = 0; i < 12; i++) priv[k++] = qvel[6 + i];
    for (int i = 0; i < 12; i++) priv[k++] = sn->act_force[i];
    priv[k++] = torso_height;
    for (; k < GO2_PRIV; k++) priv[k] = 0.0f;
  }
}

/* the sensor block of the joystick env reads feet sites through env_ids[1..4]; this env has none there: the common part only */
static void hs_read_sensors(const omodel *m, const odata *d, g2_sensors *o) {
  const int imu = m->env_ids[0];
  const real *R = &d->site_xmat[9 * imu];
  real t[3], g[3] = {0, 0, -1};
  memset(o, 0, sizeof(*o));
  mat_tmulv(t, R, &d->site_angvel[3 * imu]); for (int c = 0; c < 3; c++) o->gyro[c] = (float)t[c];
  mat_tmulv(t, R, &d->site_linvel[3 * imu]); for (int c = 0; c < 3; c++) o->local_linvel[c] = (float)t[c];
  mat_tmulv(t, R, g); for (int c = 0; c < 3; c++) o->gravity[c] = (float)t[c];
  for (int c = 0; c < 3; c++) {
    o->upvector[c] = (float)R[3 * c + 2];
    o->global_linvel[c] = (float)d->site_linvel[3 * imu + c];
    o->global_angvel[c] = (float)d->site_angvel[3 * imu + c];
  }
  for (int u = 0; u < m->nu; u++) o->act_force[u] = (float)d->actuator_force[u];
  for (int c = 0; c < 3; c++) o->accel[c] = (float)d->acc_site[c];
}

/* handstand.py:119-160 */
static void handstand_reset_env(const omodel *m, obatch *s, int e, const uint32_t key[2], odata *d) {
  const float *F = m->env_go2f;
  float *info = &s->info_go2[(size_t)e * GO2_INFO];
  memset(info, 0, sizeof(float) * GO2_INFO);
  uint32_t rng[2] = {key[0], key[1]}, ks[2][2];
  float qpos[NQ_MAX], qvel[NV_MAX];
  oracle_split(rng, 2, &ks[0][0]); rng[0] = ks[0][0]; rng[1] = ks[0][1];
  const int crouch = g2_uniform1(ks[1], 0.0f, 1.0f) < F[8];            /* jax.random.bernoulli(key, p) = uniform(key) < p */
  for (int i = 0; i < m->nq; i++) qpos[i] = m->env_go2_home[(crouch ? m->nq : 0) + i];
  for (int i = 0; i < m->nv; i++) qvel[i] = 0;
  float lo = -0.5f, hi = 0.5f, u[8];
  oracle_split(rng, 2, &ks[0][0]); rng[0] = ks[0][0]; rng[1] = ks[0][1];
  oracle_uniform(ks[1], 2, &lo, &hi, 0, u);
  qpos[0] += u[0]; qpos[1] += u[1];
  oracle_split(rng, 2, &ks[0][0]); rng[0] = ks[0][0]; rng[1] = ks[0][1];
  float yaw = g2_uniform1(ks[1], -3.14f, 3.14f);
  {
    float sn = sinf(yaw * 0.5f), cs = cosf(yaw * 0.5f);
    float q[4] = {qpos[3], qpos[4], qpos[5], qpos[6]}, r[4] = {cs, 0.0f * sn, 0.0f * sn, 1.0f * sn};
    qpos[3] = q[0] * r[0] - q[1] * r[1] - q[2] * r[2] - q[3] * r[3];
    qpos[4] = q[0] * r[1] + q[1] * r[0] + q[2] * r[3] - q[3] * r[2];
    qpos[5] = q[0] * r[2] - q[1] * r[3] + q[2] * r[0] + q[3] * r[1];
    qpos[6] = q[0] * r[3] + q[1] * r[2] - q[2] * r[1] + q[3] * r[0];
  }
  oracle_split(rng, 2, &ks[0][0]); rng[0] = ks[0][0]; rng[1] = ks[0][1];
  oracle_uniform(ks[1], 6, &lo, &hi, 0, u);
  if (!crouch) for (int i = 0; i < 6; i++) qvel[i] = u[i];
  s->time[e] = 0;
  for (int i = 0; i < m->nq; i++) s->qpos[e * m->nq + i] = qpos[i];
  for (int i = 0; i < m->nv; i++) { s->qvel[e * m->nv + i] = qvel[i]; s->qacc_warmstart[e * m->nv + i] = 0; }
  for (int i = 0; i < m->nu; i++) s->ctrl[e * m->nu + i] = qpos[7 + i];
  load_env(m, s, e, d);
  d->acc_site_id = m->env_ids[0]; d->xfrc_body = 0;
  forward(m, d);
  store_pipeline(m, s, e, d);
  g2_set_key(info, rng);
  for (int i = 0; i < m->nmetrics; i++) s->metrics[e * m->nmetrics + i] = 0;
  g2_sensors sn;
  hs_read_sensors(m, d, &sn);
  hs_obs(m, info, &s->qpos[e * m->nq], &s->qvel[e * m->nv], &sn, (float)d->site_xpos[3 * m->env_ids[0] + 2], &s->obs[e * m->obs_dim],
         s->priv_obs ? &s->priv_obs[(size_t)e * GO2_PRIV] : NULL);
  s->reward[e] = 0; s->done[e] = 0;
  wrappers_reset(m, s, e);
}

/* handstand.py:161-195 with the rewards :264-342 */
static void handstand_step_env(const omodel *m, obatch *s, int e, const float *action, odata *d) {
  const float *F = m->env_go2f, *SC = m->env_go2_scales, *home = m->env_go2_home, *soft = m->env_go2_soft;
  const int nu = m->nu;
  float *info = &s->info_go2[(size_t)e * GO2_INFO];
  const float *act = &action[e * nu];
  if ((m->wrap_flags & 2) != 0) { if (s->done[e] != 0) s->info_steps[e] = 0; s->done[e] = 0; }
  load_env(m, s, e, d);
  for (int i = 0; i < nu; i++) { volatile float sc = act[i] * F[1]; d->ctrl[i] = s->ctrl[e * nu + i] + sc; }     /* state.data.ctrl + action * scale */
  d->acc_site_id = m->env_ids[0]; d->xfrc_body = 0;
  for (int f = 0; f < m->n_frames; f++) step_physics(m, d);
  store_pipeline(m, s, e, d);
  const float *qpos = &s->qpos[e * m->nq], *qvel = &s->qvel[e * m->nv];
  g2_sensors sn;
  hs_read_sensors(m, d, &sn);
  int unwanted = 0, feet = 0;
  for (int k = 0; k < 12; k++) unwanted |= hs_pair_touching(m, d, m->env_ids[2 + k]);
  for (int k = 0; k < 2; k++) feet |= hs_pair_touching(m, d, m->env_ids[14 + k]);
  const float torso_height = (float)d->site_xpos[3 * m->env_ids[0] + 2];
  hs_obs(m, info, qpos, qvel, &sn, torso_height, &s->obs[e * m->obs_dim], s->priv_obs ? &s->priv_obs[(size_t)e * GO2_PRIV] : NULL);
  /* termination (:188-195) */
  float energy = 0;
  for (int i = 0; i < 12; i++) energy += fabsf(sn.act_force[i]) * fabsf(qvel[6 + i]);
  const float done = (sn.upvector[2] < -0.25f || unwanted || energy > F[9]) ? 1.0f : 0.0f;
  /* rewards, unscaled (:264-290) */
  float rw[HM_COUNT];
  {
    float h = torso_height < F[10] ? torso_height : F[10];
    volatile float err = F[10] - h;
    rw[HM_HEIGHT] = expf(-err / 1.0f);
    const real *R = &d->site_xmat[9 * m->env_ids[0]];
    float fwd[3] = {(float)R[0], (float)R[3], (float)R[6]};                 /* site_xmat @ [1, 0, 0] */
    volatile float c0 = fwd[0] * F[11]; volatile float c1 = fwd[1] * F[12]; volatile float c2 = fwd[2] * F[13];
    volatile float cd = c0 + c1; float cos_dist = cd + c2;
    volatile float nrm = 0.5f * cos_dist; float nr = nrm + 0.5f;
    rw[HM_ORIENT] = nr * nr;
    rw[HM_CONTACT] = feet ? 1.0f : 0.0f;
    float ar = 0, tq = 0, lim = 0, dacc = 0, pose = 0, en = 0;
    for (int i = 0; i < 12; i++) {
      volatile float da = act[i] - info[HS_LAST_ACT + i]; ar += da * da;
      tq += sn.act_force[i] * sn.act_force[i];
      float q = qpos[7 + i];
      volatile float lo_ = q - soft[i]; volatile float hi_ = q - soft[12 + i];
      lim += -(lo_ < 0.0f ? lo_ : 0.0f) + (hi_ > 0.0f ? hi_ : 0.0f);
      float qa = (float)d->qacc[6 + i]; dacc += qa * qa;
      en += fabsf(qvel[6 + i]) * fabsf(sn.act_force[i]);
    }
    for (int k = 0; k < 6; k++) { int j = m->env_go2i[k]; volatile float dq = qpos[7 + j] - home[7 + j]; pose += dq * dq; }
    rw[HM_ACTION_RATE] = ar; rw[HM_TORQUES] = tq; rw[HM_TERM] = done; rw[HM_DOF_LIMITS] = lim; rw[HM_DOF_ACC] = dacc; rw[HM_POSE] = pose;
    volatile float ss = qvel[0] * qvel[0] + qvel[1] * qvel[1];
    rw[HM_STAY_STILL] = ss + qvel[5] * qvel[5];
    rw[HM_ENERGY] = en;
  }
  /* scaled, summed in the order of the _get_reward dict, times dt, clipped (:171-176) */
  static const int order[HM_COUNT] = {HM_HEIGHT, HM_ORIENT, HM_CONTACT, HM_ACTION_RATE, HM_TORQUES, HM_TERM, HM_DOF_LIMITS, HM_DOF_ACC, HM_POSE,
                                      HM_STAY_STILL, HM_ENERGY};
  float sc[HM_COUNT], total = 0;
  for (int k = 0; k < HM_COUNT; k++) { volatile float v = rw[k] * SC[k]; sc[k] = v; }
  for (int k = 0; k < HM_COUNT; k++) { volatile float t = total + sc[order[k]]; total = t; }
  volatile float rdt = total * F[0];
  float reward = rdt < 0.0f ? 0.0f : (rdt > 10000.0f ? 10000.0f : rdt);
  info[HS_STEP] += 1.0f;
  for (int i = 0; i < nu; i++) info[HS_LAST_ACT + i] = act[i];
  float *met = &s->metrics[e * m->nmetrics];
  for (int k = 0; k < HM_COUNT; k++) met[k] = sc[k];
  s->reward[e] = reward; s->done[e] = done;
  wrappers_post(m, s, e, reward);
}

/* ------------------------------------------------------------------ exported batch entry points */
int oracle_reset(const omodel *m, obatch *s, const uint32_t *keys, int num_threads) {
  if (m->env_kind != ENV_CUBE && m->env_kind != ENV_AIRBOT_SF && m->env_kind != ENV_TSHAPE && m->env_kind != ENV_GO2 && m->env_kind != ENV_GO2_HANDSTAND) return -1;
#ifdef _OPENMP
  if (num_threads > 0) omp_set_num_threads(num_threads);
#pragma omp parallel
#endif
  {
    odata *d = odata_alloc();
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 8)
#endif
    for (int e = 0; e < s->n; e++) {
      if (m->env_kind == ENV_TSHAPE) tshape_reset_env(m, s, e, &keys[2 * e], d);
      else if (m->env_kind == ENV_GO2) go2_reset_env(m, s, e, &keys[2 * e], d);
      else if (m->env_kind == ENV_GO2_HANDSTAND) handstand_reset_env(m, s, e, &keys[2 * e], d);
      else reset_env(m, s, e, &keys[2 * e], d);
    }
    odata_free(d);
  }
  return 0;
}

int oracle_step(const omodel *m, obatch *s, const float *action, int num_threads) {
  if (m->env_kind != ENV_CUBE && m->env_kind != ENV_AIRBOT_SF && m->env_kind != ENV_TSHAPE && m->env_kind != ENV_GO2 && m->env_kind != ENV_GO2_HANDSTAND) return -1;
#ifdef _OPENMP
  if (num_threads > 0) omp_set_num_threads(num_threads);
#pragma omp parallel
#endif
  {
    odata *d = odata_alloc();
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 8)
#endif
    for (int e = 0; e < s->n; e++) {
      if (m->env_kind == ENV_TSHAPE) tshape_step_env(m, s, e, action, d);
      else if (m->env_kind == ENV_GO2) go2_step_env(m, s, e, action, d);
      else if (m->env_kind == ENV_GO2_HANDSTAND) handstand_step_env(m, s, e, action, d);
      else step_env(m, s, e, action, d);
    }
    odata_free(d);
  }
  return 0;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------ debug: one forward pass, all intermediates */
static odata *g_dbg = NULL;
int oracle_debug_forward(const omodel *m, const real *qpos, const real *qvel, const real *ctrl, const real *warm,
                         int do_step) {
  if (!g_dbg) g_dbg = odata_alloc();
  odata *d = g_dbg;
  for (int i = 0; i < m->nq; i++) d->qpos[i] = qpos[i];
  for (int i = 0; i < m->nv; i++) { d->qvel[i] = qvel[i]; d->qacc_warmstart[i] = warm ? warm[i] : 0; }
  for (int i = 0; i < m->nu; i++) d->ctrl[i] = ctrl[i];
  for (int i = 0; i < m->ngeom * 3; i++) d->geom_friction[i] = m->geom_friction[i];
  for (int i = 0; i < m->nbody; i++) d->body_mass[i] = m->body_mass[i];
  for (int i = 0; i < m->nv; i++) { d->dof_damping[i] = m->dof_damping[i]; d->dof_frictionloss[i] = m->dof_frictionloss[i]; d->dof_armature[i] = m->dof_armature[i]; }
  for (int i = 0; i < m->nbody * 3; i++) d->body_ipos[i] = m->body_ipos[i];
  for (int i = 0; i < m->nq; i++) d->qpos0[i] = m->qpos0[i];
  for (int i = 0; i < m->nu * 3; i++) { d->gainprm[i] = m->actuator_gainprm[i]; d->biasprm[i] = m->actuator_biasprm[i]; }
  d->time = 0;
  d->acc_site_id = m->nsite > 0 ? (m->env_kind == ENV_GO2 ? m->env_ids[0] : 0) : -1;
  if (do_step) step_physics(m, d); else forward(m, d);
  return d->nefc;
}
/* copies a named intermediate (as double) into out; returns element count */
int oracle_debug_get(const omodel *m, const char *name, double *out, int cap) {
  odata *d = g_dbg; if (!d) return -1;
  const real *p = NULL; int n = 0;
#define G(nm, ptr, cnt) if (!strcmp(name, nm)) { p = ptr; n = cnt; }
  G("qpos", d->qpos, m->nq) G("qvel", d->qvel, m->nv) G("xpos", d->xpos, 3 * m->nbody) G("xquat", d->xquat, 4 * m->nbody)
  G("xmat", d->xmat, 9 * m->nbody) G("xipos", d->xipos, 3 * m->nbody) G("ximat", d->ximat, 9 * m->nbody)
  G("geom_xpos", d->geom_xpos, 3 * m->ngeom) G("geom_xmat", d->geom_xmat, 9 * m->ngeom) G("site_xpos", d->site_xpos, 3 * m->nsite)
  G("site_xmat", d->site_xmat, 9 * m->nsite) G("site_linvel", d->site_linvel, 3 * m->nsite) G("site_angvel", d->site_angvel, 3 * m->nsite)
  G("actuator_force", d->actuator_force, m->nu) G("accelerometer", d->acc_site, 3)
  G("subtree_com", d->subtree_com, 3 * m->nbody) G("cinert", d->cinert, 10 * m->nbody) G("cdof", d->cdof, 6 * m->nv)
  G("cvel", d->cvel, 6 * m->nbody) G("cdof_dot", d->cdof_dot, 6 * m->nv)
  G("M", d->M, m->nv * m->nv) G("qfrc_bias", d->qfrc_bias, m->nv) G("qfrc_passive", d->qfrc_passive, m->nv)
  G("qfrc_actuator", d->qfrc_actuator, m->nv) G("qfrc_smooth", d->qfrc_smooth, m->nv) G("qacc_smooth", d->qacc_smooth, m->nv)
  G("qacc", d->qacc, m->nv) G("qfrc_constraint", d->qfrc_constraint, m->nv) G("qacc_warmstart", d->qacc_warmstart, m->nv)
  G("efc_J", d->efc_J, d->nefc * m->nv) G("efc_pos", d->efc_pos, d->nefc) G("efc_aref", d->efc_aref, d->nefc)
  G("efc_D", d->efc_D, d->nefc) G("efc_R", d->efc_R, d->nefc) G("efc_floss", d->efc_floss, d->nefc)
  G("efc_force", d->efc_force, d->nefc)
#undef G
  if (!strcmp(name, "counts")) {
    double v[6] = {(double)d->nefc, (double)d->ne, (double)d->nf, (double)d->ncon, (double)d->solver_niter, (double)d->ls_total};
    for (int i = 0; i < 6 && i < cap; i++) out[i] = v[i];
    return 6;
  }
  if (!strcmp(name, "contacts")) {   /* per contact: dist, pos(3), normal(3), body1, body2, pair */
    int k = 0;
    for (int i = 0; i < d->ncon && k + 10 <= cap; i++) {
      const ocontact *c = &d->con[i];
      out[k++] = c->dist; for (int j = 0; j < 3; j++) out[k++] = c->pos[j];
      for (int j = 0; j < 3; j++) out[k++] = c->frame[j];
      out[k++] = c->body1; out[k++] = c->body2; out[k++] = c->pair;
    }
    return k;
  }
  if (!p) return -1;
  for (int i = 0; i < n && i < cap; i++) out[i] = (double)p[i];
  return n;
}
/* The solver's objective at a given qacc, on the constraint rows of the last oracle_debug_forward: the Gauss term
 * 1/2 (M a - f0).(a - a0) plus the row costs.  Lets a test ask how good a minimiser another implementation's
 * qacc is (in the fp64 build: to fp64 resolution), independently of the path its solver took. */
double oracle_debug_cost(const omodel *m, const real *qacc) {
  const odata *d = g_dbg; if (!d) return -1.0;
  int nv = m->nv, ne = d->ne, nenf = d->ne + d->nf;
  double cost = 0;
  for (int r = 0; r < d->nefc; r++) {
    double x = -(double)d->efc_aref[r], D = d->efc_D[r];
    for (int i = 0; i < nv; i++) x += (double)d->efc_J[r * nv + i] * (double)qacc[i];
    if (r < ne) cost += 0.5 * D * x * x;
    else if (r < nenf) {
      double fl = d->efc_floss[r], rf = (double)d->efc_R[r] * fl;
      if (x <= -rf) cost += fl * (-0.5 * rf - x); else if (x >= rf) cost += fl * (-0.5 * rf + x); else cost += 0.5 * D * x * x;
    } else if (x < 0) cost += 0.5 * D * x * x;
  }
  double gauss = 0;
  for (int i = 0; i < nv; i++) {
    double Ma = 0;
    for (int j = 0; j < nv; j++) Ma += (double)d->M[i * nv + j] * (double)qacc[j];
    gauss += (Ma - (double)d->qfrc_smooth[i]) * ((double)qacc[i] - (double)d->qacc_smooth[i]);
  }
  return cost + 0.5 * gauss;
}
int oracle_real_size(void) { return (int)sizeof(real); }

/* standalone geometry entry points for collision unit tests */
int oracle_box_box(const real *pa, const real *Ra, const real *sa, const real *pb, const real *Rb, const real *sb,
                   real *out /* 4 x (dist,pos3) */, real *normal) {
  cpoint pts[4];
  int n = box_box(pa, Ra, sa, pb, Rb, sb, pts, normal);
  for (int i = 0; i < n; i++) { out[4 * i] = pts[i].dist; for (int c = 0; c < 3; c++) out[4 * i + 1 + c] = pts[i].pos[c]; }
  return n;
}
int oracle_plane_box(const real *pp, const real *pm, const real *bp, const real *bm, const real *size, real *out, real *normal) {
  cpoint pts[4];
  int n = plane_box(pp, pm, bp, bm, size, pts, normal);
  for (int i = 0; i < n; i++) { out[4 * i] = pts[i].dist; for (int c = 0; c < 3; c++) out[4 * i + 1 + c] = pts[i].pos[c]; }
  return n;
}
int oracle_plane_capsule(const real *pp, const real *pm, const real *cp, const real *cm, real radius, real halflen, real *out /* [2][4] dist, pos */,
                         real *normal, real *frame) {
  cpoint pts[4];
  int n = plane_capsule(pp, pm, cp, cm, radius, halflen, pts, normal, frame);
  for (int i = 0; i < n; i++) { out[4 * i] = pts[i].dist; for (int c = 0; c < 3; c++) out[4 * i + 1 + c] = pts[i].pos[c]; }
  return n;
}
int oracle_plane_cylinder(const real *pp, const real *pm, const real *cp, const real *cm, real radius, real halflen, real *out /* [3][4] */, real *normal) {
  cpoint pts[4];
  int n = plane_cylinder(pp, pm, cp, cm, radius, halflen, pts, normal);
  for (int i = 0; i < n; i++) { out[4 * i] = pts[i].dist; for (int c = 0; c < 3; c++) out[4 * i + 1 + c] = pts[i].pos[c]; }
  return n;
}
int oracle_hfield_sphere(const real *hp, const real *hm, const float *hsize, int nrow, int ncol, const float *data, const real *sp,
                         const real *radius, real *out, real *normal) {
  cpoint pts[1];
  int n = hfield_sphere(hp, hm, hsize, nrow, ncol, data, sp, radius[0], pts, normal);
  for (int i = 0; i < n; i++) { out[4 * i] = pts[i].dist; for (int c = 0; c < 3; c++) out[4 * i + 1 + c] = pts[i].pos[c]; }
  return n;
}
