// rsr_device.hpp -- device-side physics for the fused env step, written for gfx950 (CDNA4) only.
//
// Execution model: ONE 64-lane wavefront owns ONE environment for the whole step (prologue,
// n_frames physics substeps, epilogue).  The env's state, kinematic tiles, mass matrix, contact list
// and constraint Jacobian live in LDS for the whole step; per-dof vectors live in lane i's registers
// (lane = dof), per-constraint-row scalars in lane r's registers (lane = row mod 64), the rows of the
// mass matrix / Newton Hessian in lane i's registers for an in-register Cholesky that talks across
// lanes with v_readlane only.  HBM is touched once to load the record and once to store it.
// MFMA is not used: the per-env matrices are 20x20 with ragged active-row counts.
//
// The stages restate MJX's forward/step; the tree recursions of the
// reference (scan.body_tree) are replaced by per-lane loops over precomputed ancestor / subtree
// bitmasks (rsr_mjx_amd/model.py: topology_tables), which removes the serial dependency chains.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace rsr {

#define RSR_MINVAL 1e-15f
#define RSR_MINIMP 0.0001f
#define RSR_MAXIMP 0.9999f

enum { JNT_FREE = 0, JNT_BALL = 1, JNT_SLIDE = 2, JNT_HINGE = 3 };
enum { PAIR_PLANE_BOX = 0, PAIR_BOX_BOX = 1, PAIR_PLANE_SPHERE = 2, PAIR_HFIELD_SPHERE = 3, PAIR_PLANE_CAPSULE = 4, PAIR_PLANE_CYLINDER = 5 };
enum { INT_EULER = 0, INT_IMPLICITFAST = 3 };
enum { ENV_CUBE = 0, ENV_TSHAPE = 1, ENV_AIRBOT_SF = 2, ENV_GO2 = 3, ENV_GO2_HANDSTAND = 4 };

// ---- device view of the model blob (pointers into one device copy of the blob) ----
// Pointers of the model view are global-memory pointers, typed as such in device code: a pointer loaded from a struct in memory
// has no known address space, so the compiler would issue FLAT loads through it -- which count on the LDS counter as well, so
// that every wait for an LDS read also waits for the model loads in flight (and cost 64-bit address arithmetic per lane).
#if defined(__HIP_DEVICE_COMPILE__)
#define RSR_GP(T) const T __attribute__((address_space(1)))*
#else
#define RSR_GP(T) const T*
#endif
using gp_i = RSR_GP(int); using gp_f = RSR_GP(float); using gp_u = RSR_GP(unsigned); using gp_q = RSR_GP(int4);
struct DModel {
  gp_i body_parentid, body_rootid, body_jntnum, body_jntadr, body_dofnum, body_dofadr, body_depth;
  gp_f body_pos, body_quat, body_ipos, body_iquat, body_mass, body_inertia, body_invweight0;
  gp_i jnt_type, jnt_qposadr, jnt_dofadr, jnt_bodyid, jnt_limited, jnt_actfrclimited;
  gp_f jnt_pos, jnt_axis, jnt_range, jnt_actfrcrange, jnt_solref, jnt_solimp, jnt_margin;
  gp_i dof_bodyid, dof_jntid;
  gp_u dof_ancmask, dof_velmask, body_dofmask, body_submask;
  gp_f dof_armature, dof_damping, dof_frictionloss, dof_invweight0, dof_solref, dof_solimp;
  gp_i geom_bodyid, geom_priority;
  gp_f geom_size, geom_pos, geom_quat, geom_friction;
  gp_i geom_slot_ids;      // geom id of every geom slot (the geoms that appear in a contact pair, in geom order)
  gp_i site_bodyid;
  gp_f site_pos, site_quat;
  gp_i eq_obj1id, eq_obj2id, eq_active0;
  gp_f eq_data, eq_solref, eq_solimp;
  gp_i actuator_trnid, actuator_ctrllimited, actuator_forcelimited;
  gp_f actuator_gear, actuator_gainprm, actuator_biasprm, actuator_ctrlrange, actuator_forcerange;
  gp_i pair_geom1, pair_geom2, pair_kind, pair_condim;
  gp_f pair_solref, pair_solimp, pair_margin, pair_gap;
  gp_i fric_dofs, limit_jnts;
  // flattened one-level tables (rsr_mjx_amd/model.py: flattened_tables)
  gp_i pair_b1, pair_b2, pair_root1, pair_root2, dof_rootid, dof_jtype, dof_k, dof_act, dof_afl, body_jtype, body_qposadr;
  gp_u pair_mask1, pair_mask2;
  gp_f pair_tw, pair_incl, dof_afrange, body_jpos, body_jaxis;
  // per-lane constant records [LQ_COUNT][64] of 16-byte quads (rsr_mjx_amd/model.py: lane_records)
  gp_q lane_rec;
  // height field (at most one): size = (x, y, z, base) half extents / elevation scale, data [nrow*ncol] in [0, 1]
  gp_f hfield_size, hfield_data;
  gp_i hfield_nrow, hfield_ncol;
  gp_f qpos0;
  gp_i env_ids;
  gp_f env_action_scale, env_ctrl_lo, env_ctrl_hi, env_reset, env_reward;
  gp_f env_go2f, env_go2_scales, env_go2_home, env_go2_soft;
  gp_i env_go2i;
  float timestep, gravity[3], tolerance, ls_tolerance, impratio, meaninertia;
  int iterations, ls_iterations, integrator, disable_eulerdamp, disable_refsafe;
  int nfric, nlimit, maxdepth;
  int max_sub, max_chain;       // longest body_submask (bodies >= 1) / dof-chain mask, in bits: trip bounds of for_bits4
  int env_kind, n_frames, episode_length, wrap_flags;
};

// The model's scalars of the hot path and the lane-record pointer, read once at kernel start.  The DModel is passed by
// pointer (its ~180 pointers would otherwise sit in SGPRs), so every m.x in a stage is a scalar load from memory with its own
// wait, and every stage's record fetch first loads m.lane_rec; a wave reads these few dozen times per substep.  Held in
// SGPRs (or spilled to VGPR lanes, still ~10x cheaper than the load).
struct Hot {
  gp_q lane_rec;
  float timestep, grav0, grav1, grav2, impratio, tolerance, ls_tolerance, meaninertia;
  int maxdepth, max_sub, max_chain, integrator, disable_eulerdamp, disable_refsafe, iterations, ls_iterations, n_frames;
};
__device__ __forceinline__ Hot make_hot(const DModel& m) {
  Hot h;
  h.lane_rec = m.lane_rec; h.timestep = m.timestep; h.grav0 = m.gravity[0]; h.grav1 = m.gravity[1]; h.grav2 = m.gravity[2];
  h.impratio = m.impratio; h.tolerance = m.tolerance; h.ls_tolerance = m.ls_tolerance; h.meaninertia = m.meaninertia;
  h.maxdepth = m.maxdepth; h.max_sub = m.max_sub; h.max_chain = m.max_chain; h.integrator = m.integrator;
  h.disable_eulerdamp = m.disable_eulerdamp; h.disable_refsafe = m.disable_refsafe; h.iterations = m.iterations;
  h.ls_iterations = m.ls_iterations; h.n_frames = m.n_frames;
  return h;
}

// ---- record layout (floats per env); offsets filled on the host, see rsr_mjx.hip ----
struct Layout {
  int qpos, qvel, ctrl, warm, time, xpos, site_xpos;
  int obs, reward, done, metrics;
  int target_pos, new_cube_pos, site_pos, cube_pos, last_action, steps, truncation, episode_done, episode_metrics;
  int target_base_pos, target_vertical_pos, target_w, new_T_pos, T_pos, xita;     // T-shape env info
  int go2_info;                                                                  // Go2 joystick info block (144 floats)
  int priv_obs, f_priv_obs;                                                      // Go2 obs['privileged_state'] (123 floats) and its first-state copy
  int f_qpos, f_qvel, f_ctrl, f_warm, f_time, f_xpos, f_site_xpos, f_obs;
  int stats;
  int rec;            // floats per env (multiple of 16)
  int persist_end;    // [0, persist_end) = pipeline state restored by auto-reset (qpos..site_xpos)
};

// Work-queue dispatch of a step launch (persistent waves).  An env-step is cut into `units` consecutive groups of physics
// substeps (the first n_whole envs: into one); a wave draws tickets from one counter -- the whole envs first, then
// t = phase * n_split + env over the split ones -- and runs unit `phase` of `env`, handing the pipeline
// state (qpos, qvel, ctrl, qacc_warmstart, time) to the next phase through the env's record.  Tickets are drawn in
// dependency order, so the wave that holds (env, phase - 1) drew its ticket earlier and is running: waits always end.
// Why: with one launch-long unit per env the last waves of a launch run at their full lifetime while most SIMDs are
// already idle (8192 envs = 4 rounds of 2048 slots: ~20 % of the launch); shorter units shorten that drain.
struct Sched {
  int* ticket;          // [2]: counter of launch `launch_id & 1` (tickets from min(gridDim.x, n_whole) on; below that a wave's first
                        //      ticket is its workgroup index); workgroup 0 zeroes the other one
  int* err;             // [2]: hand-off waits that timed out since the batch was created (sticky), and the env of the last one
  unsigned* flags;      // [n]: (launch_id << 8) | phases of the env completed in this launch | RSR_FLAG_ERR
  unsigned launch_id;   // 1, 2, ... per rsr_step of the batch
  int units;            // phases per env-step of the envs that are split (1 = the whole step in one unit), at most RSR_MAX_UNITS
  int n_whole;          // envs [0, n_whole) are stepped as one unit each whatever `units` says (rsr_batch_set_whole_envs)
  int spin_cap;         // bound of the hand-off poll (sleeps of 8 x 64 cycles); test hook: rsr_batch_set_fault_injection
  int withhold_env;     // test hook: phase 0 of this env never publishes its flag (-1: none)
};
// The flag word does not depend on `units` (a stale flag of an earlier launch can never equal a later launch's value, whatever
// rsr_batch_set_schedule did in between), and carries an error bit down the phases of an env: a phase whose wait timed out
// runs on whatever the record holds, publishes with the bit set, and the env's last phase reports stats[3] = -1.
#define RSR_FLAG_ERR 0x80u
#define RSR_MAX_UNITS 64
#define RSR_SPIN_CAP_DEFAULT (1 << 22)

// relaxed agent-scope accesses = global_load / global_store ... sc1: served by / written through to memory, past the CU's L1
// and the XCD's L2, which are not coherent across CUs / XCDs within a launch (MI355X_MICROARCH.md, inter-workgroup visibility)
__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct StepArgs {
  float* state;                 // [N][rec]
  const float* action;          // [N][nu]         (step)
  const uint32_t* keys;         // [N][2]          (reset)
  const float *dr_geom_friction, *dr_body_mass, *dr_dof_damping, *dr_dof_frictionloss;   // [N][...] or null
  // extended set (Go2 randomize.py:6-109; kernels built with Dims::DREX): [N][nbody*3], [N][nq], [N][nv], [N][nu*3], [N][nu*3]
  const float *dr_body_ipos, *dr_qpos0, *dr_dof_armature, *dr_gainprm, *dr_biasprm;
  float* debug;                 // [N][RSR_DEBUG_FLOATS] or null
  int n;
  int prio_mode, prio_slots;    // wave priority schedule of the plain-launch (Go2-family) step kernels: policy, resident waves of the device
};

// ---- wave priority schedule of the plain-launch step kernels (rsr_batch_set_priority) ----
// The SIMD arbitrates between its resident waves by priority, then by age: left alone (all at priority 0) the oldest wave of a SIMD
// runs nearly unimpeded and the youngest takes the leftover issue slots, so waves that start together finish one after another
// (Go2, four per SIMD: 104 / 125 / 147 / 172 us) and the launch ends with SIMDs holding three, two, one wave for a quarter of its
// makespan.  s_setprio at the top of every substep evens the finish times out; results do not depend on it (timing only).
//   RSR_PRIO_ROTATE   (slot + substep) mod 4: the four waves of a SIMD take turns -- for waves that start together (a batch of at
//                     most one resident round);
//   RSR_PRIO_CATCH_UP outside the launch's final resident set: least progress first (3, 2, 1, 0, 0 ... by substep); in the final
//                     set, whose waves start as the slots of the round before fall free: max(that, the quarter of the set the wave
//                     belongs to by index = by start time), so the later a wave starts the longer it keeps the right of way.
enum { RSR_PRIO_OFF = 0, RSR_PRIO_ROTATE = 1, RSR_PRIO_CATCH_UP = 2 };
__device__ __forceinline__ void set_prio(int p) {      // p wave-uniform; s_setprio takes an immediate
  switch (__builtin_amdgcn_readfirstlane(p) & 3) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
  }
}
// quarter of the final resident set env e belongs to (0 = its first starters), or -1 outside the set
__device__ __forceinline__ int prio_quarter(const StepArgs& a, int e) {
  const int first = a.n - a.prio_slots;
  return e >= first ? ((e - first) * 4) / a.prio_slots : -1;
}
__device__ __forceinline__ void prio_substep(int mode, int quarter, int fr) {
  if (mode == RSR_PRIO_OFF) return;
  const int progress = fr >= 3 ? 0 : 3 - fr;
  int p;
  const int rot = ((__builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)) & 3) + fr) & 3;      // HW_ID: wave slot of the SIMD
  if (mode == RSR_PRIO_ROTATE) p = rot;
  else p = quarter > progress ? quarter : progress;
  set_prio(p);
}

// ---- compile-time dimensions of one model family ----
template <int NQ_, int NV_, int NU_, int NB_, int NJ_, int NG_, int NS_, int NP_, int NEQ_, int NF_, int NL_,
          int NCON_, int OBS_, int NMET_, int NEG_ = 0, int EG0_ = 0, int CONDIM_ = 4, int NINFO_ = 0, int ISO0_ = 0, int ISO1_ = 0, bool DREX_ = false, bool HFIELD_ = false,
          bool TALIAS_ = false, int NGA_ = NG_, bool TTAIL_ = false, bool ARROW_ = false, int TREE1_ = 0, int TREE2_ = 0, bool CAPS_ = false>
struct Dims {
  // plane-capsule / plane-cylinder pairs are compiled in (the Go2 model of the Handstand task); a capsule's contacts carry the
  // first tangent of their frame (it follows the capsule axis) from the narrow phase to the constraint rows
  static constexpr bool CAPS = CAPS_;
  // LDS diet of the single-iteration models (Go2: one Hessian per substep, explicit Euler): the transpose / exchange scratch T
  // has no storage of its own -- before the solve it is the dead part of phase A (cinert .. cfrcsum), from the Hessian on it
  // is the mass matrix's storage (M is read for the last time when the Hessian blocks are formed) -- and geom frames /
  // friction are kept for the NGA geoms that appear in a contact pair only (slots; Go2: floor or height field + four feet of
  // 39 geoms).  Together 13.4 KB -> under 10 KB per env: a fourth wave per SIMD.
  static constexpr bool TALIAS = TALIAS_;
  // TTAIL (Go2, instead of TALIAS): from the Hessian on the scratch is the tail of the phase union -- phase A is larger than the
  // Jacobian of a model with few contacts, and the part of it that the Jacobian does not overwrite is dead all the same.  The
  // mass matrix then stays in LDS through the solve and the integration, and a lane reads its row of M where it needs it
  // instead of carrying it in NV registers from the mass-matrix stage to the integrator (at 128 VGPRs those registers were
  // spilled: every M.v product re-read its row from scratch memory).
  static constexpr bool TTAIL = TTAIL_;
  static constexpr bool MROW_LDS = TTAIL_ || (TREE1_ > 0);
  static constexpr int NGA = NGA_;
  static constexpr int CONDIM = CONDIM_;             // all contact pairs of a model share one condim (checked on the host)
  static constexpr int NINFO = NINFO_;               // floats of env info staged in LDS (Go2: 144)
  static constexpr bool XFRC = NINFO_ > 0;           // xfrc_applied on one body and the accelerometer bias are compiled in (Go2)
  static constexpr bool HFIELD = HFIELD_;            // sphere / height-field pairs are compiled in
  static constexpr bool DREX = DREX_;                // per-env body_ipos / qpos0 / armature / actuator gains staged in LDS
  // dofs [ISO0, ISO1) never share a constraint row or a kinematic chain with the others (the Airbot target body: it only
  // touches static geoms), so those entries of M, H and their Cholesky factors are structural zeros; checked on the host.
  static constexpr int ISO0 = ISO0_, ISO1 = ISO1_;
  // Row-blocked factorisation: the Hessian is block diagonal over {dofs outside the isolated range} and {the isolated dofs}; when
  // both blocks have at most 16 dofs each sits in one 16-lane DPP row (block A in lanes 0.., block B in lanes 16..) and the
  // factorisation broadcasts within rows (row_newbcast) instead of through v_readlane and SGPRs.  NCH = columns a lane keeps.
  static constexpr int NISO = ISO1_ - ISO0_;
  static constexpr int NA = NV_ - NISO;
  static constexpr bool ROWCHOL = NA <= 16 && NISO <= 16 && NA >= NISO;
  // Block-arrow factorisation (Go2): a floating trunk (dofs 0 .. ANT) carrying ALEGS legs of ALEGN dofs each; legs couple to the
  // trunk and never to each other -- not in the mass matrix (separate branches of one tree) and not in J^T D J (every
  // constraint row touches the trunk and at most one leg; checked on the host).  Eliminating the legs first leaves no fill
  // between legs, so the four leg blocks are eliminated TOGETHER, one per 16-lane DPP row, each row working on its 9 x 9
  // [leg | trunk] matrix; the four Schur complements are summed onto the trunk, which every row then factors for itself.
  // 36 dependent steps of one DPP fmac per column instead of 153 v_readlane + fma pairs, and 9 + 9 factor registers per lane
  // instead of 18 + 18.  NCH = columns a lane keeps.
  static constexpr bool ARROW = ARROW_;
  static constexpr int ANT = 6, ALEGN = 3, ALEGS = ARROW_ ? (NV_ - 6) / 3 : 0;
  static_assert(!ARROW_ || (NV_ == 6 + 3 * ((NV_ - 6) / 3) && (NV_ - 6) / 3 == 4 && !ROWCHOL), "block-arrow layout: 6 trunk dofs + 4 legs of 3");
  static constexpr int NCH = ROWCHOL ? NA : (ARROW_ ? 9 : NV_);
  // Kinematic trees of the Airbot models (dof ranges [0, TREE1), [TREE1, TREE2), [TREE2, NV); checked on the host): the arm, and one
  // or two free bodies.  The mass matrix never couples them, and the Hessian couples two of them only while a contact pair
  // joins them (the arm pushing the object): otherwise each tree is factored in a 16-lane DPP row of its own, all rows in the same
  // instructions -- max(tree size) pivots instead of NA.  Skipping the coupling steps changes nothing in the arithmetic of the
  // coupled factorisation (they subtract 0 * x), so both layouts give the same bits.
  static constexpr int TREE1 = TREE1_, TREE2 = TREE2_;
  static constexpr bool ROWTREE = ROWCHOL && TREE1_ > 0;
  static constexpr int tree_of(int dof) { return (dof >= TREE1_) + (dof >= TREE2_); }
  static constexpr int tree_base(int t) { return t == 0 ? 0 : (t == 1 ? TREE1_ : TREE2_); }
  static constexpr int tree_size(int t) { return t == 0 ? TREE1_ : (t == 1 ? TREE2_ - TREE1_ : NV_ - TREE2_); }
  static constexpr int NCT = TREE1_ > TREE2_ - TREE1_ ? (TREE1_ > NV_ - TREE2_ ? TREE1_ : NV_ - TREE2_) : (TREE2_ - TREE1_ > NV_ - TREE2_ ? TREE2_ - TREE1_ : NV_ - TREE2_);
  static_assert(!ROWTREE || (NCT <= NCH && NCT <= 16 && TREE2_ >= TREE1_ && TREE2_ <= NV_), "tree rows");
  static constexpr int dof_of_col_a(int c) { return c < ISO0_ ? c : c + NISO; }      // block A column -> dof
  static constexpr int dof_of_col_b(int c) { return ISO0_ + c; }                     // block B column -> dof
  static constexpr bool coupled(int i, int j) { return (i >= ISO0_ && i < ISO1_) == (j >= ISO0_ && j < ISO1_); }
  // with an isolated range the dofs before it, inside it and after it are three separate kinematic trees (checked on the
  // host): the mass matrix alone is block diagonal over them
  static constexpr bool same_tree(int i, int j) { return ISO1_ <= ISO0_ || ((i >= ISO0_) + (i >= ISO1_)) == ((j >= ISO0_) + (j >= ISO1_)); }
  static constexpr int NQ = NQ_, NV = NV_, NU = NU_, NB = NB_, NJ = NJ_, NG = NG_, NS = NS_, NP = NP_;
  static constexpr int NEG = NEG_, EG0 = EG0_;   // geoms whose world position the env epilogue reads: env_ids[EG0 .. EG0+NEG)
  static constexpr int NEQ = NEQ_, NF = NF_, NL = NL_, NCON = NCON_, OBS = OBS_, NMET = NMET_;
  static constexpr int LD = NV_ + 1;                        // padded row stride of M / T: conflict-free row and column reads
  // Row stride of the Jacobian: a multiple of 4 floats so that a lane fetches its row with ds_read_b128, and an odd number of
  // quads so that the 16 lanes of a b128 phase (stride LDJ words) fall on distinct bank quads.
  static constexpr int LDJ = (((NV_ + 3) / 4) % 2 == 1) ? ((NV_ + 3) / 4) * 4 : ((NV_ + 3) / 4 + 1) * 4;
  static constexpr int NPYR = 2 * (CONDIM_ - 1);             // pyramid edges per contact
  static constexpr int NEFC = NEQ_ + NF_ + NL_ + NPYR * NCON_;   // constraint-row capacity (pyramid rows)
  static constexpr int NCHUNK = (NEFC + 63) / 64;           // rows per lane
  // The Jacobian is stored as BASE rows: the sparse rows as they are, and per contact the CONDIM contact-frame
  // rows (normal, tangent 1, tangent 2[, torsion]); the pyramid edges are +-mu combinations of those.
  static constexpr int NBC = CONDIM_;
  static constexpr int NSP = NEQ_ + NF_ + NL_;
  static constexpr int NBASE = NSP + NBC * NCON_;
  static constexpr int NCHB = (NBASE + 63) / 64;
};

// ---- tiny vector helpers ----
struct V3 { float x, y, z; };
struct Q4 { float w, x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ V3 ld3(const float* p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ void st3(float* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
__device__ __forceinline__ Q4 ld4(const float* p) { return Q4{p[0], p[1], p[2], p[3]}; }
__device__ __forceinline__ void st4(float* p, Q4 q) { p[0] = q.w; p[1] = q.x; p[2] = q.y; p[3] = q.z; }
__device__ __forceinline__ Q4 qmul(Q4 a, Q4 b) {
  return Q4{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
struct M33 { float m[9]; };
__device__ __forceinline__ M33 q2m(Q4 q) {
  M33 r;
  float w = q.w, x = q.x, y = q.y, z = q.z;
  r.m[0] = w * w + x * x - y * y - z * z; r.m[1] = 2 * (x * y - w * z); r.m[2] = 2 * (x * z + w * y);
  r.m[3] = 2 * (x * y + w * z); r.m[4] = w * w - x * x + y * y - z * z; r.m[5] = 2 * (y * z - w * x);
  r.m[6] = 2 * (x * z - w * y); r.m[7] = 2 * (y * z + w * x); r.m[8] = w * w - x * x - y * y + z * z;
  return r;
}
__device__ __forceinline__ V3 mulv(const M33& m, V3 v) {
  return V3{m.m[0] * v.x + m.m[1] * v.y + m.m[2] * v.z, m.m[3] * v.x + m.m[4] * v.y + m.m[5] * v.z,
            m.m[6] * v.x + m.m[7] * v.y + m.m[8] * v.z};
}
__device__ __forceinline__ V3 mulv(const float* m, V3 v) {
  return V3{m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z};
}
__device__ __forceinline__ V3 col(const float* m, int c) { return V3{m[c], m[3 + c], m[6 + c]}; }
__device__ __forceinline__ V3 qrot(Q4 q, V3 v) { return mulv(q2m(q), v); }
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

// ---- wave helpers (wave64) ----
// Cross-lane sums use DPP (VALU latency) instead of ds_bpermute (LDS latency): quad xor-1, xor-2, then
// half-row and row mirrors give every lane its 16-lane row sum; the four row sums meet through v_readlane.
// Every lane ends with the same bits (the adds commute), so decisions taken on the result are wave-uniform.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_sum16(float v) {
  v += dpp_mov<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);    // row_half_mirror
  v += dpp_mov<0x140>(v);    // row_mirror
  return v;
}
__device__ __forceinline__ float rdlane(float v, int l);
typedef float v2f __attribute__((ext_vector_type(2)));
// rows 1,3 += lane 15 of the row below; rows 2,3 += lane 31: lane 63 then holds the wave total
// Hand-scheduled: the compiler emits every chain on its own with an s_nop between dependent DPP steps (a DPP read
// needs two wait states after the write) and builds the two row_bcast steps from mov 0 / mov_dpp / add -- 19 issue slots
// per sum.  Written out, a chain is six DPP adds; three chains interleaved fill each other's wait states.
#define RSR_DPP_STEP3(ctrl) \
  "v_add_f32_dpp %0, %0, %0 " ctrl "\n\tv_add_f32_dpp %1, %1, %1 " ctrl "\n\tv_add_f32_dpp %2, %2, %2 " ctrl "\n\t"
#define RSR_DPP_STEP1(ctrl) "v_add_f32_dpp %0, %0, %0 " ctrl "\n\ts_nop 1\n\t"
#define RSR_DPP_FULL " row_mask:0xf bank_mask:0xf bound_ctrl:1"
// the three wave totals, in place (same pairing of the additions as wave_sum: results are bit-identical)
__device__ __forceinline__ void wave_sum3(float& a, float& b, float& c) {
  asm volatile("s_nop 1\n\t"
               RSR_DPP_STEP3("quad_perm:[1,0,3,2]" RSR_DPP_FULL)
               RSR_DPP_STEP3("quad_perm:[2,3,0,1]" RSR_DPP_FULL)
               RSR_DPP_STEP3("row_half_mirror" RSR_DPP_FULL)
               RSR_DPP_STEP3("row_mirror" RSR_DPP_FULL)
               RSR_DPP_STEP3("row_bcast:15 row_mask:0xa bank_mask:0xf")
               RSR_DPP_STEP3("row_bcast:31 row_mask:0xc bank_mask:0xf")
               : "+v"(a), "+v"(b), "+v"(c));
  a = rdlane(a, 63); b = rdlane(b, 63); c = rdlane(c, 63);
}
__device__ __forceinline__ float wave_sum(float v) {
  asm volatile("s_nop 1\n\t"
               RSR_DPP_STEP1("quad_perm:[1,0,3,2]" RSR_DPP_FULL)
               RSR_DPP_STEP1("quad_perm:[2,3,0,1]" RSR_DPP_FULL)
               RSR_DPP_STEP1("row_half_mirror" RSR_DPP_FULL)
               RSR_DPP_STEP1("row_mirror" RSR_DPP_FULL)
               RSR_DPP_STEP1("row_bcast:15 row_mask:0xa bank_mask:0xf")
               "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
               : "+v"(v));
  return rdlane(v, 63);
}
__device__ __forceinline__ float rdlane(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ int rdlane_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Reciprocal, reciprocal square root and square root of the geometry / dynamics stages: the hardware approximation (1 ulp) plus one
// Newton step for the reciprocals -- 3-4 instructions -- instead of the IEEE expansions of `/` and sqrtf() (v_div_scale x 2, v_rcp,
// four fma, v_div_fmas, v_div_fixup: 10-11 dependent instructions per division, ~45 divisions per substep of the cube model).
// Results are within ~1 ulp of the correctly rounded ones: the same order as the fma contraction the kernels already differ from the
// oracle by.  The env algebra (prologue, rewards, observations: the reference's op-by-op JAX-CPU arithmetic) keeps `/` and sqrtf().
// Arguments: finite, non-zero (frcp, frsq: > 0); fsqrt: >= 0, and values below the normal range count as zero (every caller tests
// its norm against RSR_MINVAL or adds a floor).
__device__ __forceinline__ float frcp(float x) { const float r = __builtin_amdgcn_rcpf(x); return __builtin_fmaf(r, __builtin_fmaf(-x, r, 1.0f), r); }
__device__ __forceinline__ float frsq(float x) { const float r = __builtin_amdgcn_rsqf(x); return __builtin_fmaf(r, __builtin_fmaf(-(x * r), 0.5f * r, 0.5f), r); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
// One wave is one workgroup, so a "barrier" only has to order this wave's own LDS accesses.  The LDS executes a wave's
// instructions in order -- a read issued after a write of the same wave observes it -- so a wavefront-scope fence is enough: it
// stops the compiler from moving memory operations across it and emits no instruction.  (__syncthreads() in a one-wave
// workgroup drops the s_barrier but keeps the workgroup-scope fence: s_waitcnt lgkmcnt(0) after every write phase, one LDS
// round trip of idle time per barrier, a few dozen per substep.)  Nothing in the kernels passes data between lanes through
// global memory across a WSYNC: the hand-off between work units has its own waits.
#define WSYNC() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")

// Visits the set bits of `mask` four candidates per trip: f(i, on) with on = false (and i = 0) once the mask is
// exhausted.  The four LDS loads of a trip do not depend on each other, so they are in flight together and a walk over
// n bits costs ceil(n/4) LDS latencies instead of n.  `nbits` is a wave-uniform upper bound of the popcount.
template <class F>
__device__ __forceinline__ void for_bits4(unsigned mask, int nbits, F&& f) {
  for (int t = 0; t < nbits; t += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool on = mask != 0u;
      const int i = on ? __builtin_ctz(mask) : 0;
      mask &= mask - 1u;
      f(i, on);
    }
  }
}

// Same walk in two phases per trip: `load(i)` of all four bits first (returns a value type), then `use(i, on, value)`.
// The scheduler keeps the source order of LDS accesses in this kernel, so a load written next to its use -- or after a
// store that may alias -- waits for its own round trip; gathering the loads in the source is what puts them in flight
// together.
template <class L, class U>
__device__ __forceinline__ void for_bits4_gather(unsigned mask, int nbits, L&& load, U&& use) {
  for (int t = 0; t < nbits; t += 4) {
    int i0, i1, i2, i3; bool o0, o1, o2, o3;
    o0 = mask != 0u; i0 = o0 ? __builtin_ctz(mask) : 0; mask &= mask - 1u;
    o1 = mask != 0u; i1 = o1 ? __builtin_ctz(mask) : 0; mask &= mask - 1u;
    o2 = mask != 0u; i2 = o2 ? __builtin_ctz(mask) : 0; mask &= mask - 1u;
    o3 = mask != 0u; i3 = o3 ? __builtin_ctz(mask) : 0; mask &= mask - 1u;
    auto v0 = load(i0); auto v1 = load(i1); auto v2 = load(i2); auto v3 = load(i3);
    use(i0, o0, v0); use(i1, o1, v1); use(i2, o2, v2); use(i3, o3, v3);
  }
}

// wave timeline (diagnostic build): constant-rate clock (100 MHz) at wave start / end, shader cycles of the wave, hardware id
__device__ __forceinline__ unsigned long long prof_realtime() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
__device__ __forceinline__ void prof_timeline(float* d, unsigned long long rt0, unsigned long long ct0) {
  unsigned long long ct1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ct1)::"memory");
  const unsigned long long rt1 = prof_realtime();
  unsigned* u = reinterpret_cast<unsigned*>(d);
  u[0] = (unsigned)rt0; u[1] = (unsigned)(rt0 >> 32); u[2] = (unsigned)rt1; u[3] = (unsigned)(rt1 >> 32);
  u[4] = (unsigned)(ct1 - ct0);
  u[5] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_REG_HW_ID
  u[6] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));     // HW_REG_XCC_ID
}
// ---- diagnostic build only (-DRSR_PROFILE): per-stage cycle counters, written to the debug buffer ----
#ifdef RSR_PROFILE
enum { PS_LOAD = 0, PS_KIN, PS_COMCRB, PS_COLL, PS_ROWS, PS_SMOOTH, PS_CHOLM, PS_SOLVE_INIT, PS_HESS, PS_LS, PS_UPD,
       PS_INTEG, PS_EPILOGUE, PS_H_PREP, PS_H_SPARSE, PS_H_CONTACT, PS_H_XCHG, PS_H_CHOL, PS_LS_SETUP, PS_X0, PS_X1, PS_X2, PS_X3, PS_X4, PS_X5, PS_X6, PS_X7, PS_L_PREP, PS_L_P0, PS_L_LO, PS_L_ITER, PS_U_JTF, PS_S_COST, PS_S_JTF,
       PS_K_LOAD, PS_K_LOCAL, PS_K_TREE, PS_K_STORE, PS_C_LOAD, PS_C_COM, PS_C_CINERT, PS_C_CRB, PS_F_LOAD, PS_F_VEL, PS_F_FRC, PS_F_SUM,
       PS_E_SENS, PS_E_FIFO, PS_E_FEET, PS_E_OBS, PS_E_PRIV, PS_E_REWARD, PS_E_BOOK, PS_COUNT };
struct Prof { unsigned long long t0; unsigned long long acc[PS_COUNT]; };
__device__ __forceinline__ unsigned long long prof_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);          // nothing is scheduled across a stamp (the asm alone only pins memory operations)
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define PROF_DECL Prof prof_; for (int i_ = 0; i_ < PS_COUNT; ++i_) prof_.acc[i_] = 0; const unsigned long long prof_rt0_ = prof_realtime(); prof_.t0 = prof_now(); const unsigned long long prof_ct0_ = prof_.t0;
#define PROF_ARG , Prof& prof_
#define PROF_PASS , prof_
#define PROF(stage) { unsigned long long t_ = prof_now(); prof_.acc[stage] += t_ - prof_.t0; prof_.t0 = t_; }
#define PROF_VMWAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#elif defined(RSR_TIMELINE)      // start / end stamps only: the wave timeline without the per-stage stamps' overhead
#define PROF_DECL const unsigned long long prof_rt0_ = prof_realtime(); unsigned long long prof_ct0_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(prof_ct0_)::"memory");
#define PROF_ARG
#define PROF_PASS
#define PROF(stage)
#define PROF_VMWAIT()
#else
#define PROF_DECL
#define PROF_ARG
#define PROF_PASS
#define PROF(stage)
#define PROF_VMWAIT()
#endif

// ---- LDS image of one environment ----
// Two phases of a forward pass never overlap in time and share one region (union X):
//   phase A (kinematics .. collision): body/geom frames, spatial inertias, velocity-stage scratch, clip slots
//   phase B (constraint rows .. solve): the constraint Jacobian
// What both phases (or the env epilogue) need stays outside the union.
constexpr int NSLOT = 16;                              // clip-scratch slots handed out to penetrating pairs (models with box pairs)
template <class C>
struct PhaseA {
  float xmat[C::NB * 9], xipos[C::NB * 3], ximat[C::NB * 9];
  float xanchor[C::NJ * 3], xaxis[C::NJ * 3];
  float gpos[C::NGA * 3], gmat[C::NGA * 9];
  float cinert[C::NB * 10], crb[C::NB * 10];
  float cvel[C::NB * 6], cdofdot[C::NV * 6], cfrc[C::NB * 6], cfrcsum[C::NB * 6];
  float clip[(C::CONDIM == 3 ? 1 : NSLOT) * 48];     // condim-3 models here are sphere-only (Go2): no clipping, one dummy slot
};
template <class C>
struct PhaseB {
  alignas(16) float J[(C::NBASE + 1) * C::LDJ];        // base rows (stride LDJ); row NBASE is the null row that pads row groups
};
template <class C>
struct Smem {
  static_assert(C::NB * 10 * 2 + C::NB * 6 >= (C::ROWCHOL ? C::NCH * 22 : C::NV * (C::NV + 1)), "phase A's dead arrays must hold the transpose scratch");
  static_assert(C::TALIAS || C::TTAIL || ((C::NBC * C::NCON > 16 ? 64 : 1) + C::NBASE + 4 + C::NCON * 8 + C::NEFC >= C::NV * (C::NV + 1)), "jtp | bval | wc | rw must hold the Hessian exchange");
  static_assert(!(C::TALIAS && C::TTAIL), "one home for the scratch");
  static_assert(!C::TTAIL || (sizeof(PhaseA<C>) >= sizeof(PhaseB<C>) + sizeof(float) * C::NV * (C::NV + 1) && sizeof(PhaseB<C>) % 16 == 0), "the tail of the phase union must hold the Hessian exchange");
  // state + per-env model overrides
  float qpos[C::NQ], qvel[C::NV], ctrl[C::NU];
  float fric[C::NGA * 3], mass[C::NB], damp[C::NV], floss[C::NV];
  // extended per-env overrides (Dims::DREX; one-element placeholders otherwise)
  float dx_ipos[C::DREX ? C::NB * 3 : 1], dx_qpos0[C::DREX ? C::NQ : 1], dx_arma[C::DREX ? C::NV : 1];
  float dx_gain[C::DREX ? C::NU * 3 : 1], dx_bias[C::DREX ? C::NU * 3 : 1];
  float xpos[C::NB * 3], xquat[C::NB * 4], spos[C::NS * 3];   // also read by the env epilogue
  float egeom[(C::NEG > 0 ? C::NEG : 1) * 3];          // world positions of the env's geoms of interest
  float smat[C::NS * 9], slinvel[C::NS * 3], sangvel[C::NS * 3];   // site frames / object velocities (sensor sources)
  float aforce[C::NU];                                 // actuator_force of the last forward pass
  // Dims::XFRC: external force on one body (data.xfrc_applied, force part), and for the accelerometer the velocity-product
  // part of body acc_body's spatial acceleration ([0, -g] + sum cdof_dot*qvel over its chain) from the last forward pass
  float xfrc[3], accb[6];
  int xfrc_body, acc_body;
  float ginfo[C::NINFO > 0 ? C::NINFO : 1];            // env info block staged in LDS for the whole step (Go2)
  float com[C::NB * 3], cdof[C::NV * 6];               // needed again when the contact Jacobian is built
  float M[C::NV * C::LD];
  // The factorisations' transpose / exchange scratch (NV * LD floats) has no storage of its own:
  //   scratch_a(): before the solve -- the mass-matrix factor's transposes -- phase A's arrays that are dead by then (cinert ..);
  //   scratch_b(): from the Hessian on (block exchange, factor transposes) and for the env prologue / epilogue staging:
  //                Dims::TALIAS: the mass matrix itself (single-iteration models: M is read for the last time when the Hessian
  //                blocks are formed); otherwise the solver's own per-row scratch jtp | bval | wc | rw, none of which is live
  //                across a factorisation (hessian_factor restores the zero words of bval that it overwrites).
  __device__ __forceinline__ float* scratch_a() { return x.a.cinert; }
  __device__ __forceinline__ float* scratch_b() {
    if constexpr (C::TALIAS) return M;
    else if constexpr (C::TTAIL) return reinterpret_cast<float*>(&x) + sizeof(PhaseB<C>) / sizeof(float);
    else return jtp;
  }
  // contacts (active only)
  float cdist[C::NCON], cpos[C::NCON * 3], cnrm[C::NCON * 3];
  float ctan[C::CAPS ? C::NCON * 3 : 0];               // Dims::CAPS: first tangent of the contact's frame, zero = the default frame of the normal
                                                       // (no storage in the other models: their LDS images keep their size)
  int cpair[C::NCON];
  int ncon, ncon_drop, nlim_act;
  int trees_joined;          // Dims::ROWTREE: a contact pair of this substep joins two kinematic trees (the Hessian needs the coupled layout)
  // Dims::ROWTREE, trees not joined: the contacts of each tree, in contact order (a block of the Hessian only visits its own tree's)
  int tree_ncon[4];
  alignas(8) unsigned char tree_con[(C::ROWTREE && C::tree_size(2) > 0) ? 3 * C::NCON : 8];
  int lim_jnt[C::NL > 0 ? C::NL : 1];
  // dof of every friction / limit row (their Jacobian row is a single +-1 there) and the Hessian's diagonal accumulator
  int sdof[C::NSP + 1];
  float dgw[C::NV];
  float bmu[C::NBASE + 4];                             // per base row: friction coefficient of that direction
  // (the next four are contiguous on purpose: scratch_b())
  float jtp[C::NBC * C::NCON > 16 ? 64 : 1];           // J^T f: partial sums of the row groups (lane = group * NV + dof)
  float bval[C::NBASE + 4];                            // per base row: J.v or the force folded onto the base row
  float wc[C::NCON * 8];                               // per contact: arrow-matrix weights of the Hessian (2*NBC-1 used)
  alignas(16) float rw[C::NEFC + 4];                   // per pyramid row: Hessian weight D*active or force; between uses its first words
                                                       // are the broadcast buffer of vec_bcast (b128 reads: hence the alignment)
  union X { PhaseA<C> a; PhaseB<C> b; } x;
};

// model fields that the extended domain randomisation makes per-env
template <class C> __device__ __forceinline__ float mdl_qpos0(const DModel& m, const Smem<C>& s, int i) { if constexpr (C::DREX) return s.dx_qpos0[i]; else return m.qpos0[i]; }
template <class C> __device__ __forceinline__ V3 mdl_ipos(const DModel& m, const Smem<C>& s, int b) { if constexpr (C::DREX) return ld3(&s.dx_ipos[3 * b]); else return ld3(&m.body_ipos[3 * b]); }
template <class C> __device__ __forceinline__ float mdl_armature(const DModel& m, const Smem<C>& s, int i) { if constexpr (C::DREX) return s.dx_arma[i]; else return m.dof_armature[i]; }
template <class C> __device__ __forceinline__ float mdl_gain(const DModel& m, const Smem<C>& s, int k) { if constexpr (C::DREX) return s.dx_gain[k]; else return m.actuator_gainprm[k]; }
template <class C> __device__ __forceinline__ float mdl_bias(const DModel& m, const Smem<C>& s, int k) { if constexpr (C::DREX) return s.dx_bias[k]; else return m.actuator_biasprm[k]; }

// ---- per-lane constant records (model.py lane_records) ----
// A lane is body l, joint l, geom l, site l and dof l at once.  Reading its constants from the per-field tables is a
// scalar pointer load followed by a vector load PER FIELD, each pair a few hundred cycles and mostly serialised by the
// control flow around them; a stage instead fetches the quads it needs at its top: coalesced 16-byte loads, all in
// flight together, one wait.  Lanes past a role's count read zeros.
enum LaneQuad { LQ_B_IDS = 0, LQ_B_POS, LQ_B_QUAT, LQ_B_JPOS, LQ_B_JAX, LQ_B_IQUAT, LQ_B_MISC, LQ_B_INERTIA,
                LQ_J_IDS, LQ_J_BQUAT, LQ_J_BPOS, LQ_J_POSAX, LQ_J_AX, LQ_G_POS, LQ_G_QUAT, LQ_S_POS, LQ_S_QUAT,
                LQ_D_IDS, LQ_D_MASKS, LQ_D_ACT, LQ_D_CTRL, LQ_D_BIAS, LQ_D_FRC,
                LQ_F_0, LQ_F_1, LQ_F_2, LQ_L_0, LQ_L_1, LQ_L_2, LQ_L_3, LQ_P_0, LQ_P_1, LQ_P_2, LQ_P_3, LQ_P_4, LQ_P_5,
                LQ_E_0, LQ_E_1, LQ_E_2, LQ_E_3, LQ_E_4, LQ_COUNT };
// number of lanes that hold data in a quad's row (its role's count); the lanes past it hold zeros
template <class C>
__device__ __forceinline__ constexpr int lq_count(int quad) {
  return quad < LQ_J_IDS ? C::NB : quad < LQ_G_POS ? C::NJ : quad < LQ_S_POS ? C::NGA : quad < LQ_D_IDS ? C::NS : quad < LQ_F_0 ? C::NV
       : quad < LQ_L_0 ? C::NF : quad < LQ_P_0 ? C::NL : quad < LQ_E_0 ? C::NP : C::NEQ;
}
// Lanes past the role's count all read the first zero entry of the row instead of their own: a row then occupies
// (count + 1) * 16 bytes of the CU's vector L1 instead of 1 KB, and the whole table ~14 KB instead of 42 KB (the L1 is 32 KB
// and eight waves in different stages share it).
template <class C>
__device__ __forceinline__ int4 lrec(const Hot& h, int quad, int lane) {
  const int cnt = lq_count<C>(quad) < 63 ? lq_count<C>(quad) : 63;
  return h.lane_rec[quad * 64 + (lane < cnt ? lane : cnt)];
}
// The lane index a stage passes to lrec: `lane` plus a zero the optimiser cannot see through.  The records are loop
// invariant, and hoisted out of the substep loop they would stay live across the solver, where every register is taken:
// they would be spilled there and come back from scratch memory instead.
__device__ __forceinline__ int lrec_lane(int lane) {
  int z;
  asm volatile("s_mov_b32 %0, 0" : "=s"(z));
  return lane + z;
}
// A value the optimiser may not look through (no instruction).  The factorisations compare a lane's position in its row with every
// elimination step; the position is loop invariant, so the ~2 NCH lane masks of a factorisation were computed once per substep, kept
// in SGPR pairs across the Newton loop, spilled there (the solver is short of SGPRs) and read back with two v_readlane per use --
// 136 reads per Newton iteration of the cube kernel -- where the compare itself is one v_cmp into VCC.
__device__ __forceinline__ int opaque_v(int x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ float asf(int x) { return __builtin_bit_cast(float, x); }
__device__ __forceinline__ V3 v3_xyz(int4 r) { return V3{asf(r.x), asf(r.y), asf(r.z)}; }
__device__ __forceinline__ V3 v3_yzw(int4 r) { return V3{asf(r.y), asf(r.z), asf(r.w)}; }
__device__ __forceinline__ Q4 q4_of(int4 r) { return Q4{asf(r.x), asf(r.y), asf(r.z), asf(r.w)}; }

// =====================================================================================
// stage 1: kinematics (MJX smooth.kinematics).  Lane b first builds body b's transform relative to its
// parent, joint included (one sincos for all hinges at once); the tree is then composed level by level,
// each lane fetching its parent's pose with a cross-lane read, so the serial chain is maxdepth short
// register-only steps instead of nbody LDS round trips.  Bodies carry at most one joint (checked on the host).
// =====================================================================================
template <class C>
__device__ __forceinline__ void kinematics(const DModel& m, const Hot& h, Smem<C>& s, int lane PROF_ARG) {
  const int lr = lrec_lane(lane);
  static_assert(C::NB <= 64 && C::NG <= 64 && C::NJ <= 64 && C::NS <= 64, "one lane per body / geom / joint / site");
  // every model constant of the stage, for all of this lane's roles
  const int4 rb_ids = lrec<C>(h, LQ_B_IDS, lr), rb_pos = lrec<C>(h, LQ_B_POS, lr), rb_quat = lrec<C>(h, LQ_B_QUAT, lr);
  const int4 rb_jpos = lrec<C>(h, LQ_B_JPOS, lr), rb_jax = lrec<C>(h, LQ_B_JAX, lr), rb_iquat = lrec<C>(h, LQ_B_IQUAT, lr);
  const int4 rb_misc = lrec<C>(h, LQ_B_MISC, lr);
  const int4 rj_ids = lrec<C>(h, LQ_J_IDS, lr), rj_bquat = lrec<C>(h, LQ_J_BQUAT, lr), rj_bpos = lrec<C>(h, LQ_J_BPOS, lr);
  const int4 rj_posax = lrec<C>(h, LQ_J_POSAX, lr), rj_ax = lrec<C>(h, LQ_J_AX, lr);
  const int4 rg_pos = lrec<C>(h, LQ_G_POS, lr), rg_quat = lrec<C>(h, LQ_G_QUAT, lr);
  const int4 rs_pos = lrec<C>(h, LQ_S_POS, lr), rs_quat = lrec<C>(h, LQ_S_QUAT, lr);
  PROF_VMWAIT(); PROF(PS_K_LOAD)
  const int b = lane < C::NB ? lane : 0;
  const int parent = rb_ids.x, depth = lane < C::NB ? rb_ids.y : -1;
  V3 bp = v3_xyz(rb_pos);
  Q4 bq = q4_of(rb_quat);
  V3 lp = bp; Q4 lq = bq;
  const int jt = lane < C::NB ? rb_ids.z : -1;
  if (jt >= 0) {
    const int qa = rb_ids.w;
    if (jt == JNT_FREE) {
      lp = ld3(&s.qpos[qa]);
      lq = ld4(&s.qpos[qa + 3]);
      float n = fsqrt(lq.w * lq.w + lq.x * lq.x + lq.y * lq.y + lq.z * lq.z);
      if (n < RSR_MINVAL) lq = Q4{1, 0, 0, 0};
      else { float inv = frcp(n); lq = Q4{lq.w * inv, lq.x * inv, lq.y * inv, lq.z * inv}; }
      st4(&s.qpos[qa + 3], lq);                    // MJX writes the normalised quaternion back
    } else {
      V3 jp = v3_xyz(rb_jpos), jax = V3{asf(rb_jpos.w), asf(rb_jax.x), asf(rb_jax.y)};
      float q0;
      if constexpr (C::DREX) q0 = s.dx_qpos0[qa]; else q0 = asf(rb_pos.w);
      float dq = s.qpos[qa] - q0;
      if (jt == JNT_HINGE) {
        float sn, cs;
        sincosf(dq * 0.5f, &sn, &cs);
        Q4 qloc = Q4{cs, jax.x * sn, jax.y * sn, jax.z * sn};
        lq = qmul(bq, qloc);
        lp = bp + qrot(bq, jp - qrot(qloc, jp));
      } else {
        lp = bp + qrot(bq, jax * dq);
      }
    }
  }
  PROF(PS_K_LOCAL)
  // Compose down the tree by pointer jumping: every body holds its pose relative to an ancestor `anc`, and a round composes it
  // with that ancestor's own (pose relative to ITS ancestor) and jumps: anc <- anc of anc.  After r rounds a body is expressed
  // relative to its 2^r-th ancestor, the world being its own ancestor with the identity pose (composing with the identity is
  // exact, so bodies that have arrived keep their bits): ceil(log2(maxdepth)) rounds of cross-lane reads and one composition
  // each instead of maxdepth rounds (cube: 4 instead of 9, 14 of 64 lanes busy either way).  Same products, associated pairwise
  // instead of root-to-leaf: results differ from the sequential composition in the last bit or two.
  V3 pos = lane == 0 ? v3(0, 0, 0) : lp;
  Q4 q = lane == 0 ? Q4{1, 0, 0, 0} : lq;
  int anc = (lane < C::NB && lane > 0) ? parent : 0;
  for (int span = 1; span < h.maxdepth; span *= 2) {
    V3 pp = v3(__shfl(pos.x, anc), __shfl(pos.y, anc), __shfl(pos.z, anc));
    Q4 pq = Q4{__shfl(q.w, anc), __shfl(q.x, anc), __shfl(q.y, anc), __shfl(q.z, anc)};
    const int anc2 = __shfl(anc, anc);
    pos = pp + qrot(pq, pos); q = qmul(pq, q);
    anc = anc2;
  }
  PROF(PS_K_TREE)
  if (lane < C::NB) {
    st3(&s.xpos[3 * b], pos); st4(&s.xquat[4 * b], q);
    M33 R = q2m(q);
#pragma unroll
    for (int c = 0; c < 9; ++c) s.x.a.xmat[9 * b + c] = R.m[c];
    V3 ipos;
    if constexpr (C::DREX) ipos = ld3(&s.dx_ipos[3 * b]); else ipos = V3{asf(rb_jax.z), asf(rb_jax.w), asf(rb_misc.x)};
    st3(&s.x.a.xipos[3 * b], pos + mulv(R, ipos));
    M33 Ri = q2m(qmul(q, q4_of(rb_iquat)));
#pragma unroll
    for (int c = 0; c < 9; ++c) s.x.a.ximat[9 * b + c] = Ri.m[c];
  }
  WSYNC();
  PROF(PS_K_STORE)
  if (lane < C::NJ) {          // joint anchors / axes in the world frame (frame of the body before the joint acts)
    const int j = lane, jb = rj_ids.x, jpar = rj_ids.y;
    if (rj_ids.z == JNT_FREE) {
      st3(&s.x.a.xanchor[3 * j], ld3(&s.xpos[3 * jb])); st3(&s.x.a.xaxis[3 * j], v3(0, 0, 1));
    } else {
      Q4 pq = ld4(&s.xquat[4 * jpar]);
      Q4 qpre = qmul(pq, q4_of(rj_bquat));
      V3 ppre = ld3(&s.xpos[3 * jpar]) + qrot(pq, v3_xyz(rj_bpos));
      st3(&s.x.a.xanchor[3 * j], ppre + qrot(qpre, V3{asf(rj_bpos.w), asf(rj_posax.x), asf(rj_posax.y)}));
      st3(&s.x.a.xaxis[3 * j], qrot(qpre, V3{asf(rj_posax.z), asf(rj_posax.w), asf(rj_ax.x)}));
    }
  }
  if (lane < C::NGA) {          // geom slots: the geoms that appear in a contact pair (all of them unless Dims::NGA says otherwise)
    const int g = lane, gb = rg_pos.x;
    Q4 gq = ld4(&s.xquat[4 * gb]);
    st3(&s.x.a.gpos[3 * g], ld3(&s.xpos[3 * gb]) + mulv(&s.x.a.xmat[9 * gb], v3_yzw(rg_pos)));
    M33 Rg = q2m(qmul(gq, q4_of(rg_quat)));
#pragma unroll
    for (int c = 0; c < 9; ++c) s.x.a.gmat[9 * g + c] = Rg.m[c];
  }
  if (lane < C::NS) {
    const int sb = rs_pos.x;
    st3(&s.spos[3 * lane], ld3(&s.xpos[3 * sb]) + mulv(&s.x.a.xmat[9 * sb], v3_yzw(rs_pos)));
    M33 Rs = q2m(qmul(ld4(&s.xquat[4 * sb]), q4_of(rs_quat)));
#pragma unroll
    for (int c = 0; c < 9; ++c) s.smat[9 * lane + c] = Rs.m[c];
  }
  WSYNC();
}

// spatial inertia (10-vector: xx yy zz xy xz yz, m*off(3), m) times motion vector (ang, lin)
__device__ __forceinline__ void inert_mul(float* o, const float* i, const float* v) {
  V3 p = ld3(i + 6), va = ld3(v), vl = ld3(v + 3);
  V3 t = cross(p, vl);
  o[0] = i[0] * va.x + i[3] * va.y + i[4] * va.z + t.x;
  o[1] = i[3] * va.x + i[1] * va.y + i[5] * va.z + t.y;
  o[2] = i[4] * va.x + i[5] * va.y + i[2] * va.z + t.z;
  V3 u = cross(p, va);
  o[3] = i[9] * vl.x - u.x; o[4] = i[9] * vl.y - u.y; o[5] = i[9] * vl.z - u.z;
}

// dst[b][0..NC) = sum of src[k][0..NC) over the bodies k of `mask`, with four lanes per body (lane = 4 b + g): lane g takes
// the set bits number g, g + 4, g + 8, g + 12 of the mask -- all of its loads in one batch, one LDS round trip for the whole
// subtree -- and the four partial sums meet with two quad-permute adds.  (One lane per (body, component) item walked the
// mask four bits per trip: NB * NC / 64 passes of up to four dependent round trips.)  Needs NB <= 16.
template <class C, int NC>
__device__ __forceinline__ void subtree_sum_quad(const float* src, float* dst, unsigned mask, int lane) {
  static_assert(C::NB <= 16, "four lanes per body and sixteen candidate bits per lane quad");
  const int b = lane >> 2, g = lane & 3;
#pragma unroll
  for (int u = 0; u < 3; ++u) if (u < g) mask &= mask - 1u;          // skip the bits of the lanes before this one
  int k[4]; bool on[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    on[u] = mask != 0u; k[u] = on[u] ? __builtin_ctz(mask) : 0;
    mask &= mask - 1u; mask &= mask - 1u; mask &= mask - 1u; mask &= mask - 1u;      // (0 stays 0)
  }
  float v[4][NC];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int c = 0; c < NC; ++c) v[u][c] = src[NC * k[u] + c];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    float a = (on[0] ? v[0][c] : 0.0f) + (on[1] ? v[1][c] : 0.0f) + (on[2] ? v[2][c] : 0.0f) + (on[3] ? v[3][c] : 0.0f);
    a += dpp_mov<0xB1>(a);          // quad_perm [1,0,3,2]
    a += dpp_mov<0x4E>(a);          // quad_perm [2,3,0,1]
    if (b < C::NB && (c & 3) == g) dst[NC * b + c] = a;
  }
}

// out[0..6) = sum of vec6[i][0..6) * qvel[i] over the dofs i of `mask`, four lanes per body as in subtree_sum_quad; every
// lane of the quad receives the sum
template <class C>
__device__ __forceinline__ void chain_sum_quad(const Smem<C>& s, const float* vec6, unsigned mask, int lane, int max_bits, float (&out)[6]) {
  const int g = lane & 3;
#pragma unroll
  for (int u = 0; u < 3; ++u) if (u < g) mask &= mask - 1u;
  float acc[6] = {0, 0, 0, 0, 0, 0};
  for (int t = 0; t < max_bits; t += 16) {
    int k[4]; bool on[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      on[u] = mask != 0u; k[u] = on[u] ? __builtin_ctz(mask) : 0;
      mask &= mask - 1u; mask &= mask - 1u; mask &= mask - 1u; mask &= mask - 1u;
    }
    float qd[4], v[4][6];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      qd[u] = s.qvel[k[u]];
#pragma unroll
      for (int c = 0; c < 6; ++c) v[u][c] = vec6[6 * k[u] + c];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float q = on[u] ? qd[u] : 0.0f;
#pragma unroll
      for (int c = 0; c < 6; ++c) acc[c] += v[u][c] * q;
    }
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    float a = acc[c];
    a += dpp_mov<0xB1>(a);
    a += dpp_mov<0x4E>(a);
    out[c] = a;
  }
}

// =====================================================================================
// stage 2+3: com_pos, crb, dense mass matrix (MJX smooth.com_pos / crb / make_m)
// =====================================================================================
template <class C>
__device__ __forceinline__ void com_crb_mass(const DModel& m, const Hot& h, Smem<C>& s, int lane PROF_ARG) {
  const int lr = lrec_lane(lane);
  const int4 rb_misc = lrec<C>(h, LQ_B_MISC, lr), rb_inertia = lrec<C>(h, LQ_B_INERTIA, lr);
  const int4 rd_ids = lrec<C>(h, LQ_D_IDS, lr), rd_masks = lrec<C>(h, LQ_D_MASKS, lr);
  PROF_VMWAIT(); PROF(PS_C_LOAD)
  const int max_sub = h.max_sub, max_chain = h.max_chain;
  // subtree mask of body lane / 4 for the four-lanes-per-body composite inertia sum below (body b's mask sits in lane b's
  // record; a cross-lane read needs its source lane active, so it is fetched here in uniform code)
  const int qb = lane >> 2;
  const unsigned crb_mask_q = (unsigned)__shfl(rb_misc.z, qb < C::NB ? qb : 0);
  if (lane < C::NEG) st3(&s.egeom[3 * lane], ld3(&s.x.a.gpos[3 * m.env_ids[C::EG0 + lane]]));
  // subtree centre of mass, four lanes per body (see subtree_sum_quad); the world's subtree is every body
  {
    unsigned mask = qb < C::NB ? crb_mask_q : 0u;
    const int g = lane & 3;
#pragma unroll
    for (int u = 0; u < 3; ++u) if (u < g) mask &= mask - 1u;
    int k[4]; bool on[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      on[u] = mask != 0u; k[u] = on[u] ? __builtin_ctz(mask) : 0;
      mask &= mask - 1u; mask &= mask - 1u; mask &= mask - 1u; mask &= mask - 1u;
    }
    float mk[4]; V3 pk[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { mk[u] = s.mass[k[u]]; pk[u] = ld3(&s.x.a.xipos[3 * k[u]]); }
    float mm = 0; V3 acc = v3(0, 0, 0);
#pragma unroll
    for (int u = 0; u < 4; ++u) { const float w = on[u] ? mk[u] : 0.0f; mm += w; acc = acc + pk[u] * w; }
    mm += dpp_mov<0xB1>(mm); mm += dpp_mov<0x4E>(mm);
    acc.x += dpp_mov<0xB1>(acc.x); acc.x += dpp_mov<0x4E>(acc.x);
    acc.y += dpp_mov<0xB1>(acc.y); acc.y += dpp_mov<0x4E>(acc.y);
    acc.z += dpp_mov<0xB1>(acc.z); acc.z += dpp_mov<0x4E>(acc.z);
    if (g == 0 && qb < C::NB) {
      V3 c = mm < RSR_MINVAL ? ld3(&s.x.a.xipos[3 * qb]) : acc * frcp(mm);
      st3(&s.com[3 * qb], c);
    }
  }
  WSYNC();
  PROF(PS_C_COM)
  if (lane < C::NB) {
    int b = lane;
    const float* R = &s.x.a.ximat[9 * b];
    V3 off = ld3(&s.x.a.xipos[3 * b]) - ld3(&s.com[3 * rb_misc.y]);
    float I0 = asf(rb_inertia.x), I1 = asf(rb_inertia.y), I2 = asf(rb_inertia.z), ms = s.mass[b];
    float* ci = &s.x.a.cinert[10 * b];
    ci[0] = R[0] * I0 * R[0] + R[1] * I1 * R[1] + R[2] * I2 * R[2] + ms * (off.y * off.y + off.z * off.z);
    ci[1] = R[3] * I0 * R[3] + R[4] * I1 * R[4] + R[5] * I2 * R[5] + ms * (off.x * off.x + off.z * off.z);
    ci[2] = R[6] * I0 * R[6] + R[7] * I1 * R[7] + R[8] * I2 * R[8] + ms * (off.x * off.x + off.y * off.y);
    ci[3] = R[0] * I0 * R[3] + R[1] * I1 * R[4] + R[2] * I2 * R[5] - ms * off.x * off.y;
    ci[4] = R[0] * I0 * R[6] + R[1] * I1 * R[7] + R[2] * I2 * R[8] - ms * off.x * off.z;
    ci[5] = R[3] * I0 * R[6] + R[4] * I1 * R[7] + R[5] * I2 * R[8] - ms * off.y * off.z;
    ci[6] = ms * off.x; ci[7] = ms * off.y; ci[8] = ms * off.z; ci[9] = ms;
  }
  // cdof: lane = dof
  if (lane < C::NV) {
    const int i = lane, j = rd_ids.x, b = rd_ids.y, jt = rd_ids.z, k = rd_ids.w;
    V3 off = ld3(&s.com[3 * rd_masks.x]) - ld3(&s.x.a.xanchor[3 * j]);
    V3 ang, lin;
    if (jt == JNT_FREE) {
      if (k < 3) { ang = v3(0, 0, 0); lin = v3(k == 0, k == 1, k == 2); }
      else { ang = col(&s.x.a.xmat[9 * b], k - 3); lin = cross(ang, off); }
    } else if (jt == JNT_HINGE) { ang = ld3(&s.x.a.xaxis[3 * j]); lin = cross(ang, off); }
    else { ang = v3(0, 0, 0); lin = ld3(&s.x.a.xaxis[3 * j]); }
    st3(&s.cdof[6 * i], ang); st3(&s.cdof[6 * i + 3], lin);
  }
  for (int t = lane; t < C::NV * C::LD; t += 64) s.M[t] = 0.0f;
  WSYNC();
  PROF(PS_C_CINERT)
  // composite inertia of body b = sum of cinert over its subtree (the world's is not needed: zero)
  subtree_sum_quad<C, 10>(s.x.a.cinert, s.x.a.crb, (qb == 0 || qb >= C::NB) ? 0u : crb_mask_q, lane);
  WSYNC();
  PROF(PS_C_CRB)
  if (lane < C::NV) {
    int i = lane;
    float f[6];
    inert_mul(f, &s.x.a.crb[10 * rd_ids.y], &s.cdof[6 * i]);
    unsigned mask = (unsigned)rd_masks.y;
    float arma;
    if constexpr (C::DREX) arma = s.dx_arma[i]; else arma = asf(rd_masks.w);
    // Exhausted masks store into the row's padding word M[i][NV] rather than being predicated off (a predicated store
    // drags its loads into the exec region, where they wait one by one).
    const int pad = i * C::LD + C::NV;
    struct C6 { float c[6]; };
    for_bits4_gather(mask, max_chain,
      [&](int j) { C6 r;
#pragma unroll
                   for (int k = 0; k < 6; ++k) r.c[k] = s.cdof[6 * j + k];
                   return r; },
      [&](int j, bool on, const C6& r) {
        float v = f[0] * r.c[0] + f[1] * r.c[1] + f[2] * r.c[2] + f[3] * r.c[3] + f[4] * r.c[4] + f[5] * r.c[5];
        if (j == i) v += arma;
        s.M[on ? i * C::LD + j : pad] = v; s.M[on ? j * C::LD + i : pad] = v;
      });
  }
  WSYNC();
}

// =====================================================================================
// in-register factorisation H = L D L^T (square-root free Cholesky), lane i = row i.  a[] holds row i of H (lower part
// used); on return a[k] holds the UNIT lower factor L[i][k] for k < i and zero elsewhere, lt[k] holds L[k][i] for k > i
// (row i of L^T, fetched through the LDS scratch T) and zero elsewhere, and the return value is 1/D[i] of this lane.
// The zeros let the substitutions run without any lane selects: one v_readlane and one v_fma per step, which is the
// whole dependent chain (the scaled form L L^T needs a multiply by 1/L[k][k] and two selects inside each step).
// =====================================================================================
template <class C, bool MASS_ONLY = false>
__device__ __forceinline__ float chol_factor(float (&a)[C::NV], float (&lt)[C::NV], float* T, int lane) {
  float dinv = 0.0f;
#pragma unroll
  for (int k = 0; k < C::NV; ++k) {
    float piv = rdlane(a[k], k);
    piv = piv > 0.0f ? piv : RSR_MINVAL;
    float r = __builtin_amdgcn_rcpf(piv);        // v_rcp_f32 (1 ulp) + one Newton step
    r = r + r * (1.0f - piv * r);
    if (lane == k) dinv = r;
    const float u = a[k];                        // unscaled column k: u_i = H'[i][k]
    a[k] = (lane > k) ? u * r : 0.0f;            // unit-lower column
    // trailing update a[j] -= L[i][k] * u_j, two columns per instruction where both are coupled to k: the pair of
    // broadcast values sits in an SGPR pair and the row entries in adjacent VGPRs, so the update is one v_pk_fma_f32
    auto cpl = [](int j, int kk) { return MASS_ONLY ? C::same_tree(j, kk) : C::coupled(j, kk); };   // structural zeros fold at compile time
    const v2f akk = {a[k], a[k]};
#pragma unroll
    for (int j = k + 1; j < C::NV; ++j) {
      const bool second_of_pair = (j & 1) == 1 && j - 1 > k && cpl(j - 1, k) && cpl(j, k);     // done together with j - 1
      const bool first_of_pair = (j & 1) == 0 && j + 1 < C::NV && cpl(j, k) && cpl(j + 1, k);
      if (second_of_pair) continue;
      if (first_of_pair) {
        const int j1 = j + 1 < C::NV ? j + 1 : j;
        v2f aj = {a[j], a[j1]};
        const v2f uj = {rdlane(u, j), rdlane(u, j1)};
        aj = __builtin_elementwise_fma(-akk, uj, aj);
        a[j] = aj.x; a[j1] = aj.y;
      } else if (cpl(j, k)) {
        a[j] = __builtin_fmaf(-a[k], rdlane(u, j), a[j]);      // rows i >= j use it; others hold garbage never read
      }
    }
  }
  // transpose through LDS: T[k][i] = L[i][k]
  if (lane < C::NV) {
#pragma unroll
    for (int k = 0; k < C::NV; ++k) T[k * C::LD + lane] = a[k];
  }
  WSYNC();
  if (lane < C::NV) {
#pragma unroll
    for (int k = 0; k < C::NV; ++k) lt[k] = T[lane * C::LD + k];   // lt[k] = L[k][lane] for k > lane, else 0
  } else {
#pragma unroll
    for (int k = 0; k < C::NV; ++k) lt[k] = 0.0f;
  }
  WSYNC();
  return dinv;
}
// solves L D L^T x = b; lane i holds b_i in x and receives x_i.
template <class C>
__device__ __forceinline__ float chol_solve(const float (&a)[C::NV], const float (&lt)[C::NV], float dinv, float x, int lane) {
#pragma unroll
  for (int k = 0; k < C::NV; ++k) x -= a[k] * rdlane(x, k);          // forward, unit lower: lanes <= k hold a[k] = 0
  x *= dinv;
#pragma unroll
  for (int k = C::NV - 1; k >= 0; --k) x -= lt[k] * rdlane(x, k);    // backward, unit upper: lanes >= k hold lt[k] = 0
  return x;
}

// =====================================================================================
// Row-blocked L D L^T (Dims::ROWCHOL).  Lane L = (block L >> 4, row L & 15) holds row `L & 15` of its block: a[c] = H[dof(L)][dof of
// column c of that block].  One factorisation step is: pivot = row_newbcast:k of a[k] (DPP, within each 16-lane row: both blocks
// at once), reciprocal, and ONE v_fmac_f32_dpp per trailing column (a[j] += row_newbcast:j(u) * -a[k]) -- against v_readlane +
// s_nop + v_fma (or a packed pair) per column through SGPRs before.  Same operations on the same operands in the same order:
// the factors are bit-identical to chol_factor's.  Inline asm is opaque to the hazard recogniser, so every DPP read of a
// register an asm statement (or anything else) may just have written is preceded by its two wait states (s_nop 1).
// Lanes outside the blocks (and block B's columns past its size) hold zeros throughout.
// =====================================================================================
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
template <int K> __device__ __forceinline__ float row_bcast(float v) {          // lane K of each 16-lane row, to the whole row
  float r;
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v), "n"(K));
  return r;
}
template <int K, bool WAIT> __device__ __forceinline__ void fmac_row_bcast(float& acc, float src, float mul) {   // acc += bcast_K(src) * -mul
  if constexpr (WAIT) asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(src), "v"(mul), "n"(K));
  else asm volatile("v_fmac_f32_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(src), "v"(mul), "n"(K));
}
template <int K> __device__ __forceinline__ void fmac_self_bcast(float& x, float mul) {                           // x += bcast_K(x) * -mul
  asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, -%1 row_newbcast:%2 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x) : "v"(mul), "n"(K));
}
template <class C> __device__ __forceinline__ int rowchol_dof(int lane) {       // dof of a lane, -1 outside the blocks
  const int r = lane & 15;
  if (lane < 16) return r < C::NA ? (r < C::ISO0 ? r : r + C::NISO) : -1;
  return (lane < 32 && r < C::NISO) ? C::ISO0 + r : -1;
}
template <class C> __device__ __forceinline__ int rowchol_lane(int dof) {       // lane of a dof
  return dof < C::ISO0 ? dof : (dof < C::ISO1 ? 16 + dof - C::ISO0 : dof - C::NISO);
}
// Loads this lane's row from a natural-order [NV][LD] matrix in LDS (entries it does not own come from a zero word), adds
// `diag` to its diagonal, factors.  lt[] = the transposed factor's row, through the LDS scratch `T` (>= NCH * 22 floats).
template <class C, bool MASS_ONLY, bool HAS_DIAG = false>
__device__ __forceinline__ float rowchol_factor(const float* src, float diag, float (&a)[C::NCH], float (&lt)[C::NCH], float* T, int lane) {
  static_assert(C::ROWCHOL, "row-blocked factorisation needs both blocks within 16 lanes");
  const int r = opaque_v(lane & 15), dofl = rowchol_dof<C>(lane);
  const bool in_a = lane < 16 && dofl >= 0, in_b = lane >= 16 && dofl >= 0;
  // One exec region per block with compile-time column offsets (an address select per entry costs a VGPR each and the
  // kernel has none to spare); entries a lane does not own are zero
#pragma unroll
  for (int c = 0; c < C::NCH; ++c) a[c] = 0.0f;
  if (in_a) {
    const float* row = src + dofl * C::LD;
#pragma unroll
    for (int c = 0; c < C::NCH; ++c) a[c] = row[C::dof_of_col_a(c)];
  } else if (in_b) {
    const float* row = src + dofl * C::LD;
#pragma unroll
    for (int c = 0; c < C::NISO; ++c) a[c] = row[C::dof_of_col_b(c)];
  }
  if constexpr (HAS_DIAG) {
#pragma unroll
    for (int c = 0; c < C::NCH; ++c) if (c == r) a[c] += diag;           // (diag is zero for lanes outside the blocks)
  }
  float dinv = 0.0f;
  static_for<0, C::NCH>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    float piv = row_bcast<k>(a[k]);
    piv = piv > 0.0f ? piv : RSR_MINVAL;
    float rcp = __builtin_amdgcn_rcpf(piv);
    rcp = rcp + rcp * (1.0f - piv * rcp);
    if (r == k) dinv = rcp;
    const float u = a[k];
    a[k] = (r > k) ? u * rcp : 0.0f;
    const float lik = a[k];
    bool first = true;
    static_for<k + 1, C::NCH>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      // structural zeros fold at compile time: block A's columns j, k are dofs; the mass matrix alone is block diagonal over
      // the kinematic trees inside block A as well (block B is one tree)
      constexpr bool live = MASS_ONLY ? C::same_tree(C::dof_of_col_a(j), C::dof_of_col_a(k)) : true;
      if constexpr (live) {
        if (first) { fmac_row_bcast<j, true>(a[j], u, lik); first = false; }
        else fmac_row_bcast<j, false>(a[j], u, lik);
      }
    });
  });
  // transpose through LDS: T[c][L] = a[c] = L[row of L][c]; lt[k] of lane (b, i) = L[k][i] = a[i] of lane (b, k)
  constexpr int TS = 22;
  if (lane < TS) {
#pragma unroll
    for (int c = 0; c < C::NCH; ++c) T[c * TS + lane] = a[c];
  }
  WSYNC();
#pragma unroll
  for (int k = 0; k < C::NCH; ++k) lt[k] = 0.0f;
  if (in_a) {
    const float* col = &T[r * TS];
#pragma unroll
    for (int k = 0; k < C::NCH; ++k) lt[k] = col[k];
  } else if (in_b) {
    const float* col = &T[r * TS + 16];
#pragma unroll
    for (int k = 0; k < C::NISO; ++k) lt[k] = col[k];
  }
  WSYNC();
  return dinv;
}
// Solves with the row-blocked factor.  b / result: natural order (lane i = dof i); the two permutations are ds_bpermutes.
template <class C>
__device__ __forceinline__ float rowchol_solve(const float (&a)[C::NCH], const float (&lt)[C::NCH], float dinv, float b, int lane) {
  const int dofl = rowchol_dof<C>(lane);
  float x = __shfl(b, dofl >= 0 ? dofl : 0);
  x = dofl >= 0 ? x : 0.0f;
  static_for<0, C::NCH>([&](auto kc) { constexpr int k = decltype(kc)::value; fmac_self_bcast<k>(x, a[k]); });          // forward: lanes <= k hold a[k] = 0
  x *= dinv;
  static_for<0, C::NCH>([&](auto kc) { constexpr int k = C::NCH - 1 - decltype(kc)::value; fmac_self_bcast<k>(x, lt[k]); });   // backward
  return __shfl(x, lane < C::NV ? rowchol_lane<C>(lane) : 0);
}

// =====================================================================================
// One kinematic tree per DPP row (Dims::ROWTREE): lane (row t, pos p) holds row `p` of tree t's block, a[c] = H[base_t + p][base_t + c].
// Used for the mass-matrix factorisations always, for the Hessian while no contact pair joins two trees.
// =====================================================================================
template <class C> __device__ __forceinline__ int rowtree_dof(int lane) {
  const int t = lane >> 4, p = lane & 15;
  return (t < 3 && p < C::tree_size(t < 3 ? t : 0)) ? C::tree_base(t) + p : -1;
}
template <class C> __device__ __forceinline__ int rowtree_lane(int dof) { return 16 * C::tree_of(dof) + dof - C::tree_base(C::tree_of(dof)); }
template <class C, bool HAS_DIAG = false>
__device__ __forceinline__ float rowtree_factor(const float* src, float diag, float (&a)[C::NCH], float (&lt)[C::NCH], float* T, int lane) {
  static_assert(C::ROWTREE, "per-tree factorisation");
  constexpr int N = C::NCT;
  const int t = lane >> 4, p = opaque_v(lane & 15), dofl = rowtree_dof<C>(lane);
#pragma unroll
  for (int c = 0; c < C::NCH; ++c) a[c] = 0.0f;
  // one exec region per tree with compile-time column offsets (see rowchol_factor)
  static_for<0, 3>([&](auto tc) {
    constexpr int tt = decltype(tc)::value;
    if constexpr (C::tree_size(tt) > 0) {
      if (t == tt && p < C::tree_size(tt)) {
        const float* row = src + (C::tree_base(tt) + p) * C::LD + C::tree_base(tt);
#pragma unroll
        for (int c = 0; c < C::tree_size(tt); ++c) a[c] = row[c];
      }
    }
  });
  if constexpr (HAS_DIAG) {
#pragma unroll
    for (int c = 0; c < N; ++c) if (c == p) a[c] += diag;            // (diag is zero for lanes outside the trees)
  }
  float dinv = 0.0f;
  static_for<0, N>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    float piv = row_bcast<k>(a[k]);
    piv = piv > 0.0f ? piv : RSR_MINVAL;
    float rcp = __builtin_amdgcn_rcpf(piv);
    rcp = rcp + rcp * (1.0f - piv * rcp);
    if (p == k) dinv = rcp;
    const float u = a[k];
    a[k] = (p > k) ? u * rcp : 0.0f;
    const float lik = a[k];
    static_for<k + 1, N>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if constexpr (j == k + 1) fmac_row_bcast<j, true>(a[j], u, lik); else fmac_row_bcast<j, false>(a[j], u, lik);
    });
  });
  // transpose through LDS, per row: T[t][c][p] = a[c]; lt[k] = L[k][p] = a[p] of lane (t, k)
  if (t < 3 && p < N) {
#pragma unroll
    for (int c = 0; c < N; ++c) T[(t * N + c) * N + p] = a[c];
  }
  WSYNC();
#pragma unroll
  for (int k = 0; k < C::NCH; ++k) lt[k] = 0.0f;
  if (t < 3 && p < N) {
    const float* col = &T[(t * N + p) * N];
#pragma unroll
    for (int k = 0; k < N; ++k) lt[k] = col[k];
  }
  WSYNC();
  return dofl >= 0 ? dinv : 0.0f;
}
template <class C>
__device__ __forceinline__ float rowtree_solve(const float (&a)[C::NCH], const float (&lt)[C::NCH], float dinv, float b, int lane) {
  constexpr int N = C::NCT;
  const int dofl = rowtree_dof<C>(lane);
  float x = __shfl(b, dofl >= 0 ? dofl : 0);
  x = dofl >= 0 ? x : 0.0f;
  static_for<0, N>([&](auto kc) { constexpr int k = decltype(kc)::value; fmac_self_bcast<k>(x, a[k]); });
  x *= dinv;
  static_for<0, N>([&](auto kc) { constexpr int k = N - 1 - decltype(kc)::value; fmac_self_bcast<k>(x, lt[k]); });
  return __shfl(x, lane < C::NV ? rowtree_lane<C>(lane) : 0);
}

// =====================================================================================
// Block-arrow L D L^T (Dims::ARROW).  Lane (row = lane >> 4, pos = lane & 15): pos 0..2 = the dofs of leg `row`, pos 3..8 = the
// trunk dofs (a copy per row), pos >= 9 idle.  a[c]: column c of the row's local 9 x 9 matrix (c < 3: the leg, c >= 3: trunk).
// Elimination order legs, then trunk: a different order of the same exact arithmetic than the natural-order factorisation
// (results differ from it in rounding only).
// =====================================================================================
template <class C> __device__ __forceinline__ int arrow_dof(int lane) {           // dof of a lane, -1 for idle lanes
  const int row = lane >> 4, pos = lane & 15;
  return pos < C::ALEGN ? C::ANT + C::ALEGN * row + pos : (pos < C::ANT + C::ALEGN ? pos - C::ALEGN : -1);
}
template <class C> __device__ __forceinline__ int arrow_lane(int dof) {           // lane of a dof (the trunk: its copy in row 0)
  return dof < C::ANT ? C::ALEGN + dof : 16 * ((dof - C::ANT) / C::ALEGN) + (dof - C::ANT) % C::ALEGN;
}
template <int K, int N, class A>
__device__ __forceinline__ float arrow_step(A& a, int pos) {                      // one elimination step of the local matrices
  float piv = row_bcast<K>(a[K]);
  piv = piv > 0.0f ? piv : RSR_MINVAL;
  float rcp = __builtin_amdgcn_rcpf(piv);
  rcp = rcp + rcp * (1.0f - piv * rcp);
  const float u = a[K];
  a[K] = (pos > K) ? u * rcp : 0.0f;
  const float lik = a[K];
  static_for<K + 1, N>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    if constexpr (j == K + 1) fmac_row_bcast<j, true>(a[j], u, lik); else fmac_row_bcast<j, false>(a[j], u, lik);
  });
  return rcp;
}
// src: natural-order [NV][LD] matrix in LDS; diag: added to this lane's diagonal entry (value for the lane's dof; the trunk's is
// taken from row 0 only).  lt[] = row of the transposed local factor, through the LDS scratch T (>= 4 * 81 floats).
template <class C, bool HAS_DIAG = false>
__device__ __forceinline__ float arrow_factor(const float* src, float diag, float (&a)[C::NCH], float (&lt)[C::NCH], float* T, int lane) {
  static_assert(C::ARROW && C::NCH == 9, "block-arrow factorisation");
  constexpr int N = 9, NL = C::ALEGN;
  const int row = lane >> 4, pos = opaque_v(lane & 15), dofl = arrow_dof<C>(lane);
  const bool trunk_copy = pos >= NL && row > 0;          // trunk lanes of rows 1..3: their trunk block starts at zero
#pragma unroll
  for (int c = 0; c < N; ++c) a[c] = 0.0f;
  if (dofl >= 0) {
    const float* r = src + dofl * C::LD;
    // the leg columns: a leg lane's own row; a trunk lane's entries H[trunk][leg] are read as H[leg][trunk] -- in natural order
    // they sit above the diagonal, and the Hessian exchange only writes the lower triangle
    const float* lc = pos < NL ? r + C::ANT + NL * row : src + (C::ANT + NL * row) * C::LD + (pos - NL);
    const int lstride = pos < NL ? 1 : C::LD;
#pragma unroll
    for (int c = 0; c < NL; ++c) a[c] = lc[c * lstride];
    float t[C::ANT];
#pragma unroll
    for (int c = 0; c < C::ANT; ++c) t[c] = r[c];
#pragma unroll
    for (int c = 0; c < C::ANT; ++c) a[NL + c] = trunk_copy ? 0.0f : t[c];
  }
  if constexpr (HAS_DIAG) {
    const float dg = trunk_copy ? 0.0f : diag;
#pragma unroll
    for (int c = 0; c < N; ++c) if (c == pos) a[c] += dg;
  }
  float dinv = 0.0f;
  static_for<0, NL>([&](auto kc) {                        // the four legs, together
    constexpr int k = decltype(kc)::value;
    const float rcp = arrow_step<k, N>(a, pos);
    if (pos == k) dinv = rcp;
  });
  // the legs' Schur complements meet on the trunk: every row ends with the same sum (same bits: the adds commute)
#pragma unroll
  for (int c = NL; c < N; ++c) { float v = a[c]; v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); a[c] = v; }
  static_for<NL, N>([&](auto kc) {                        // the trunk, once per row (identical copies)
    constexpr int k = decltype(kc)::value;
    const float rcp = arrow_step<k, N>(a, pos);
    if (pos == k) dinv = rcp;
  });
  // transpose through LDS, per row: T[row][c][pos] = a[c]; lt[k] = L[k][pos] = a[pos] of lane (row, k)
  if (pos < N) {
#pragma unroll
    for (int c = 0; c < N; ++c) T[(row * N + c) * N + pos] = a[c];
  }
  WSYNC();
#pragma unroll
  for (int k = 0; k < N; ++k) lt[k] = 0.0f;
  if (pos < N) {
    const float* col = &T[(row * N + pos) * N];
#pragma unroll
    for (int k = 0; k < N; ++k) lt[k] = col[k];
  }
  WSYNC();
  return dinv;
}
// Solves with the block-arrow factor.  b / result: natural order (lane i = dof i).
template <class C>
__device__ __forceinline__ float arrow_solve(const float (&a)[C::NCH], const float (&lt)[C::NCH], float dinv, float b, int lane) {
  constexpr int N = 9, NL = C::ALEGN;
  const int row = lane >> 4, pos = lane & 15, dofl = arrow_dof<C>(lane);
  float x = __shfl(b, dofl >= 0 ? dofl : 0);
  x = (dofl >= 0 && !(pos >= NL && row > 0)) ? x : 0.0f;              // the trunk's right-hand side enters once (row 0)
  static_for<0, NL>([&](auto kc) { constexpr int k = decltype(kc)::value; fmac_self_bcast<k>(x, a[k]); });        // legs, forward
  {
    float v = x; v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);      // trunk: b - sum over the legs of L y
    x = pos >= NL ? v : x;
  }
  static_for<NL, N>([&](auto kc) { constexpr int k = decltype(kc)::value; fmac_self_bcast<k>(x, a[k]); });        // trunk, forward
  x *= dinv;
  static_for<0, N>([&](auto kc) { constexpr int k = N - 1 - decltype(kc)::value; fmac_self_bcast<k>(x, lt[k]); });  // backward
  return __shfl(x, lane < C::NV ? arrow_lane<C>(lane) : 0);
}

// A per-dof vector (lane i holds v_i) to every lane: through LDS -- one store, NV/4 uniform-address ds_read_b128 (broadcast reads) --
// instead of NV v_readlane into NV scalar registers.  The products that follow (M.v and J.v of the same vector) then take their
// operands from vector registers: the solver's loops are short of SGPRs (their spills are v_writelane / v_readlane pairs in the
// Newton loop's head), and a readlane needs wait states after the VALU write of its source.  The buffer is the head of the row-weight
// array, which is dead wherever a vector is broadcast (its users rewrite what they read: hessian_factor, jt_force, make_constraint).
template <class C> constexpr int NVP = (C::NV + 3) / 4 * 4;
template <class C>
__device__ __forceinline__ void vec_bcast(Smem<C>& s, int lane, float v, float (&vb)[NVP<C>]) {
  static_assert(NVP<C> <= C::NEFC, "the broadcast buffer is the head of rw");
  WSYNC();
  if (lane < NVP<C>) s.rw[lane] = lane < C::NV ? v : 0.0f;
  WSYNC();
  const float4* p = reinterpret_cast<const float4*>(&s.rw[0]);
#pragma unroll
  for (int k = 0; k < NVP<C> / 4; ++k) { const float4 t = p[k]; vb[4 * k] = t.x; vb[4 * k + 1] = t.y; vb[4 * k + 2] = t.z; vb[4 * k + 3] = t.w; }
  WSYNC();
}
// row . broadcast vector
template <class C>
__device__ __forceinline__ float row_dot(const float (&row)[C::NV], const float (&vb)[NVP<C>]) {
  // even and odd terms in the two halves of one packed accumulator: NV/2 v_pk_fma_f32 instead of NV v_fma_f32 (without
  // fast-math the compiler may not split the serial sum itself)
  static_assert(C::NV % 2 == 0, "paired dot product assumes an even dof count");
  v2f acc = {0.0f, 0.0f};
#pragma unroll
  for (int j = 0; j < C::NV; j += 2) acc = __builtin_elementwise_fma((v2f){row[j], row[j + 1]}, (v2f){vb[j], vb[j + 1]}, acc);
  return acc.x + acc.y;
}

// =====================================================================================
// stage 4: collision (lane = geom pair); penetrating contacts are compacted into the LDS list
// =====================================================================================
__device__ __forceinline__ void make_frame(V3 n, V3& a, V3& b, V3& c) {
  float nn = fsqrt(dot(n, n));
  a = nn > RSR_MINVAL ? n * frcp(nn) : v3(0, 0, 0);
  b = (a.y > -0.5f && a.y < 0.5f) ? v3(0, 1, 0) : v3(0, 0, 1);
  b = b - a * dot(a, b);
  float bn = fsqrt(dot(b, b));
  b = bn > RSR_MINVAL ? b * frcp(bn) : v3(0, 0, 0);
  c = cross(a, b);
}

struct CPts { float dist[4]; V3 pos[4]; V3 n; int cnt; V3 t; };      // t: first tangent of the frame (Dims::CAPS), zero = default

// A pair whose narrow phase found penetration and now needs LDS scratch for its manifold (24 floats for
// plane-box vertex supports, 48 for the two ping-pong polygons of box-box clipping).  Everything the second
// half needs is carried in registers, so only pairs that actually touch claim one of the NSLOT scratch slots.
struct ClipJob {
  int kind;                      // 0 = nothing pending, 1 = plane-box, 2 = box-box face contact
  V3 o, nref, axu, axv;          // reference frame (box-box) / plane point, normal and in-plane axes (plane-box)
  float hu, hv;                  // reference face half extents
  float px[4], py[4], pd[4];     // incident face in the reference frame
  bool flip;                     // reported normal = -nref (reference box is geom2)
  V3 bp, size; const float* bm;  // plane-box: the box
};

__device__ __forceinline__ V3 box_vertex(const float* bm, V3 bp, V3 size, int v) {
  V3 loc = v3((v & 4) ? size.x : -size.x, (v & 2) ? size.y : -size.y, (v & 1) ? size.z : -size.z);
  return mulv(bm, loc) + bp;
}

// pm / bm: the two frames as values (the caller fetched both matrices in one batch); bm_lds: the box frame in LDS for the
// clip stage, which runs later and must not keep a register copy alive
__device__ __forceinline__ void plane_box_sat(V3 pp, const float* pm, V3 bp, const float* bm, const float* bm_lds, V3 size, CPts& out, ClipJob& job) {
  V3 n = col(pm, 2);
  out.n = n; out.cnt = 0; job.kind = 0;
  // deepest vertex's penetration in closed form: the support of the box along -n (the eight vertices evaluated one by one were ~100
  // instructions per substep; the candidates' own supports are evaluated per vertex, one lane each, in collision())
  const float smax = dot(pp - bp, n) + fabsf(dot(col(bm, 0), n) * size.x) + fabsf(dot(col(bm, 1), n) * size.y) + fabsf(dot(col(bm, 2), n) * size.z);
  if (!(smax > 0.0f)) return;
  job.kind = 1; job.o = pp; job.nref = n; job.axu = col(pm, 0); job.axv = col(pm, 1);
  job.bp = bp; job.size = size; job.bm = bm_lds; job.hu = smax;
}
// 15-axis SAT.  Separated pairs return nothing; edge-edge contacts are finished here; face contacts fill `job`.
__device__ __forceinline__ void box_box_sat(V3 pa, const float* Ra, V3 sa_, V3 pb, const float* Rb, V3 sb_, CPts& out, ClipJob& job) {
  out.cnt = 0; job.kind = 0;
  const float sa[3] = {sa_.x, sa_.y, sa_.z}, sb[3] = {sb_.x, sb_.y, sb_.z};
  V3 dp = pb - pa;
  V3 A[3] = {col(Ra, 0), col(Ra, 1), col(Ra, 2)}, B[3] = {col(Rb, 0), col(Rb, 1), col(Rb, 2)};
  float Cm[3][3], AC[3][3], t[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    t[i] = dot(A[i], dp);
#pragma unroll
    for (int j = 0; j < 3; ++j) { Cm[i][j] = dot(A[i], B[j]); AC[i][j] = fabsf(Cm[i][j]) + 1e-6f; }
  }
  float best_face = -1e30f; int face_code = -1; bool sep = false;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float sv = fabsf(t[i]) - (sa[i] + sb[0] * AC[i][0] + sb[1] * AC[i][1] + sb[2] * AC[i][2]);
    sep |= sv > 0.0f;
    if (sv > best_face) { best_face = sv; face_code = i; }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float tb = t[0] * Cm[0][j] + t[1] * Cm[1][j] + t[2] * Cm[2][j];
    float sv = fabsf(tb) - (sb[j] + sa[0] * AC[0][j] + sa[1] * AC[1][j] + sa[2] * AC[2][j]);
    sep |= sv > 0.0f;
    if (sv > best_face) { best_face = sv; face_code = 3 + j; }
  }
  if (sep) return;
  // edge axes: only the index pair of the best axis is tracked in the loop; its edge geometry (the closest edges' centres, nine
  // cross products and sign tests when done per candidate) is built once, after the loop, and only if an edge contact wins
  float best_edge = -1e30f; bool have_edge = false;
  int wi = 0, wj = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      float l2 = 1.0f - Cm[i][j] * Cm[i][j];
      float ra = sa[i1] * AC[i2][j] + sa[i2] * AC[i1][j];
      float rb = sb[j1] * AC[i][j2] + sb[j2] * AC[i][j1];
      float tl = t[i2] * Cm[i1][j] - t[i1] * Cm[i2][j];
      float sv = (fabsf(tl) - (ra + rb)) * frsq(fmaxf(l2, 1e-12f));
      bool ok = !(l2 < 1e-6f);
      sep |= ok && sv > 0.0f;
      if (ok && sv > best_edge) { best_edge = sv; have_edge = true; wi = i; wj = j; }
    }
  }
  if (sep) return;
  if (have_edge && (best_edge > 0.95f * best_face + 1e-6f)) {
    // the winning edges, picked with selects (no dynamic indexing of the register arrays)
    // (the candidates pass through an empty asm as scalars: a select between elements of the register arrays would be rewritten
    // into a load through a selected address, which puts the arrays -- and a store of every element per SAT -- into scratch)
    auto sel3 = [](float a, float b, float c, int k) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c)); return k == 0 ? a : (k == 1 ? b : c); };
    auto pick = [&](const V3 (&v)[3], int k) { return V3{sel3(v[0].x, v[1].x, v[2].x, k), sel3(v[0].y, v[1].y, v[2].y, k), sel3(v[0].z, v[1].z, v[2].z, k)}; };
    auto pickf = [&](const float (&v)[3], int k) { return sel3(v[0], v[1], v[2], k); };
    const int wi1 = wi == 2 ? 0 : wi + 1, wi2 = wi == 0 ? 2 : wi - 1, wj1 = wj == 2 ? 0 : wj + 1, wj2 = wj == 0 ? 2 : wj - 1;
    const V3 eai = pick(A, wi), ebj = pick(B, wj), Ai1 = pick(A, wi1), Ai2 = pick(A, wi2), Bj1 = pick(B, wj1), Bj2 = pick(B, wj2);
    const float ea_half = pickf(sa, wi), eb_half = pickf(sb, wj), sai1 = pickf(sa, wi1), sai2 = pickf(sa, wi2), sbj1 = pickf(sb, wj1), sbj2 = pickf(sb, wj2);
    V3 L = cross(eai, ebj);
    {
      V3 Ls = L;
      if (dot(Ls, dp) < 0.0f) Ls = Ls * -1.0f;
      const V3 ea_c = pa + Ai1 * ((dot(Ls, Ai1) > 0 ? 1.0f : -1.0f) * sai1) + Ai2 * ((dot(Ls, Ai2) > 0 ? 1.0f : -1.0f) * sai2);
      const V3 eb_c = pb + Bj1 * ((dot(Ls, Bj1) > 0 ? -1.0f : 1.0f) * sbj1) + Bj2 * ((dot(Ls, Bj2) > 0 ? -1.0f : 1.0f) * sbj2);
      L = L * frsq(dot(L, L));
      if (dot(L, dp) < 0.0f) L = L * -1.0f;
      V3 r = eb_c - ea_c;
      float uab = dot(eai, ebj), q1 = dot(eai, r), q2 = -dot(ebj, r), den = 1.0f - uab * uab;
      const float rden = frcp(den);          // (den = 1 - cos^2 of the edge pair >= 1e-6: the loop skips near-parallel edges)
      float sp = clampf((q1 + uab * q2) * rden, -ea_half, ea_half);
      float up = clampf((uab * q1 + q2) * rden, -eb_half, eb_half);
      V3 qa = ea_c + eai * sp, qb = eb_c + ebj * up;
      float dist = dot(qb - qa, L);
      if (!(dist < 0.0f)) return;
      out.n = L; out.dist[0] = dist; out.pos[0] = (qa + qb) * 0.5f; out.cnt = 1;
      return;
    }
  }
  // face contact: reference box R, incident box Q; selections are done with compile-time indices + selects
  const bool ref_is_a = face_code < 3;
  const int k = ref_is_a ? face_code : face_code - 3;
  V3 pr = ref_is_a ? pa : pb, pq = ref_is_a ? pb : pa;
  V3 Rr[3], Rq[3]; float sr[3], sq[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    Rr[c] = ref_is_a ? A[c] : B[c]; Rq[c] = ref_is_a ? B[c] : A[c];
    sr[c] = ref_is_a ? sa[c] : sb[c]; sq[c] = ref_is_a ? sb[c] : sa[c];
  }
  V3 nref = k == 0 ? Rr[0] : (k == 1 ? Rr[1] : Rr[2]);
  V3 axu = k == 0 ? Rr[1] : (k == 1 ? Rr[2] : Rr[0]);
  V3 axv = k == 0 ? Rr[2] : (k == 1 ? Rr[0] : Rr[1]);
  float hk = k == 0 ? sr[0] : (k == 1 ? sr[1] : sr[2]);
  float hu = k == 0 ? sr[1] : (k == 1 ? sr[2] : sr[0]);
  float hv = k == 0 ? sr[2] : (k == 1 ? sr[0] : sr[1]);
  if (dot(nref, pq - pr) < 0.0f) nref = nref * -1.0f;
  float d0 = dot(nref, Rq[0]), d1 = dot(nref, Rq[1]), d2 = dot(nref, Rq[2]);
  int mq = 0; float bestd = fabsf(d0);
  if (fabsf(d1) > bestd) { bestd = fabsf(d1); mq = 1; }
  if (fabsf(d2) > bestd) { bestd = fabsf(d2); mq = 2; }
  float dm = mq == 0 ? d0 : (mq == 1 ? d1 : d2);
  float sgn_q = dm > 0 ? -1.0f : 1.0f;
  V3 qm = mq == 0 ? Rq[0] : (mq == 1 ? Rq[1] : Rq[2]);
  V3 qu = mq == 0 ? Rq[1] : (mq == 1 ? Rq[2] : Rq[0]);
  V3 qv = mq == 0 ? Rq[2] : (mq == 1 ? Rq[0] : Rq[1]);
  float sm = mq == 0 ? sq[0] : (mq == 1 ? sq[1] : sq[2]);
  float su_ = mq == 0 ? sq[1] : (mq == 1 ? sq[2] : sq[0]);
  float sv_ = mq == 0 ? sq[2] : (mq == 1 ? sq[0] : sq[1]);
  V3 o = pr + nref * hk;
  const float su[4] = {1, -1, -1, 1}, sv4[4] = {1, 1, -1, -1};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    V3 w = pq + qm * (sgn_q * sm) + qu * (su[i] * su_) + qv * (sv4[i] * sv_) - o;
    job.px[i] = dot(w, axu); job.py[i] = dot(w, axv); job.pd[i] = -dot(w, nref);
  }
  job.kind = 2; job.o = o; job.nref = nref; job.axu = axu; job.axv = axv; job.hu = hu; job.hv = hv; job.flip = !ref_is_a;
}


// closest point of triangle abc to p (Ericson, Real-Time Collision Detection 5.1.5)
__device__ __forceinline__ V3 closest_on_triangle(V3 p, V3 a, V3 b, V3 c) {
  V3 ab = b - a, ac = c - a, ap = p - a;
  float d1 = dot(ab, ap), d2 = dot(ac, ap);
  if (d1 <= 0.0f && d2 <= 0.0f) return a;
  V3 bp = p - b;
  float d3 = dot(ab, bp), d4 = dot(ac, bp);
  if (d3 >= 0.0f && d4 <= d3) return b;
  float vc = d1 * d4 - d3 * d2;
  if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) return a + ab * (d1 * frcp(d1 - d3));
  V3 cp = p - c;
  float d5 = dot(ab, cp), d6 = dot(ac, cp);
  if (d6 >= 0.0f && d5 <= d6) return c;
  float vb = d5 * d2 - d1 * d6;
  if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) return a + ac * (d2 * frcp(d2 - d6));
  float va = d3 * d6 - d5 * d4;
  if (va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f) return b + (c - b) * ((d4 - d3) * frcp((d4 - d3) + (d5 - d6)));
  float den = frcp(va + vb + vc);
  return a + ab * (vb * den) + ac * (vc * den);
}

// sphere (geom2) vs height field (geom1): one contact at the closest point of the triangulated surface
// (cells split along (col,row)-(col+1,row+1)); centre below the surface: perpendicular depth to the triangle above it.
// Same construction as the oracle's hfield_sphere; the sphere spans at most two grid lines per axis (host check).
// In three parts: the pair's lane places the sphere (hfield_place); when its centre is above the surface the eight candidate
// triangles of the 2 x 2 cells around it are searched ONE PER LANE (hfield_search, lane = 8 * pair + triangle) and the closest is
// picked by an arg-min over the pair's eight lanes with the serial scan's tie rule (the first of equal minima) -- the scan of eight
// closest_on_triangle calls by the pair's lane was ~12 % of the rough-terrain kernel's instructions; the pair's lane finishes
// (hfield_finish).  Same arithmetic per triangle, same pick.
struct HfJob { int state; V3 p, q, n; float dist, best; int c0, r0; };      // state: 0 no contact, 1 done (centre below), 2 search pending
__device__ __forceinline__ void hfield_place(const DModel& m, V3 hpos, const float* hmat, V3 spos, float radius, HfJob& j) {
  const int nrow = m.hfield_nrow[0], ncol = m.hfield_ncol[0];
  const float sx = m.hfield_size[0], sy = m.hfield_size[1], sz = m.hfield_size[2];
  const float* __restrict__ data = m.hfield_data;
  V3 rel = spos - hpos;
  V3 p = v3(hmat[0] * rel.x + hmat[3] * rel.y + hmat[6] * rel.z, hmat[1] * rel.x + hmat[4] * rel.y + hmat[7] * rel.z,
            hmat[2] * rel.x + hmat[5] * rel.y + hmat[8] * rel.z);
  j.state = 0; j.p = p;
  if (p.x < -sx || p.x > sx || p.y < -sy || p.y > sy) return;
  const float dx = 2.0f * sx / (float)(ncol - 1), dy = 2.0f * sy / (float)(nrow - 1);      // (wave-uniform: scalar code)
  const float rdx = frcp(dx), rdy = frcp(dy);
  int ci = (int)floorf((p.x + sx) * rdx), ri = (int)floorf((p.y + sy) * rdy);
  ci = ci < 0 ? 0 : (ci > ncol - 2 ? ncol - 2 : ci); ri = ri < 0 ? 0 : (ri > nrow - 2 ? nrow - 2 : ri);
  float x0 = -sx + dx * (float)ci, y0 = -sy + dy * (float)ri;
  float u = (p.x - x0) * rdx, v = (p.y - y0) * rdy;
  float z00 = data[ri * ncol + ci] * sz, z10 = data[ri * ncol + ci + 1] * sz, z01 = data[(ri + 1) * ncol + ci] * sz, z11 = data[(ri + 1) * ncol + ci + 1] * sz;
  float gx, gy, zs;
  if (u >= v) { gx = (z10 - z00) * rdx; gy = (z11 - z10) * rdy; zs = z00 + (z10 - z00) * u + (z11 - z10) * v; }
  else { gx = (z11 - z01) * rdx; gy = (z01 - z00) * rdy; zs = z00 + (z11 - z01) * u + (z01 - z00) * v; }
  if (p.z < zs) {
    float inv = frsq(gx * gx + gy * gy + 1.0f);
    j.n = v3(-gx * inv, -gy * inv, inv);
    float depth = (zs - p.z) * inv;
    j.dist = -depth - radius;
    j.q = p + j.n * depth;
    j.state = 1;
  } else {
    int c0 = (int)floorf((p.x - radius + sx) * rdx), r0 = (int)floorf((p.y - radius + sy) * rdy);
    j.c0 = c0 < 0 ? 0 : (c0 > ncol - 3 ? ncol - 3 : c0); j.r0 = r0 < 0 ? 0 : (r0 > nrow - 3 ? nrow - 3 : r0);
    j.state = 2;
  }
}
// All lanes (uniform control flow).  Lane 8 f + k searches triangle k = (row j, column i, half t) = bits (2, 1, 0) of pair lane f's
// 2 x 2 cells -- the serial scan's order -- and lane f receives the closest point and its squared distance.
template <int NPAIR>
__device__ __forceinline__ void hfield_search(const DModel& m, int lane, HfJob& j) {
  static_assert(8 * NPAIR <= 64, "eight lanes per height-field pair");
  const int f = lane >> 3, k = lane & 7, src = f < NPAIR ? f : 0;
  const int st = __shfl(j.state, src);
  const V3 p = v3(__shfl(j.p.x, src), __shfl(j.p.y, src), __shfl(j.p.z, src));
  const int c0 = __shfl(j.c0, src), r0 = __shfl(j.r0, src);
  const bool work = f < NPAIR && st == 2;
  float d2 = 3.0e38f; V3 q = p;
  if (work) {
    const int ncol = m.hfield_ncol[0], nrow = m.hfield_nrow[0];
    const float sx = m.hfield_size[0], sy = m.hfield_size[1], sz = m.hfield_size[2];
    const float* __restrict__ data = m.hfield_data;
    const float dx = 2.0f * sx / (float)(ncol - 1), dy = 2.0f * sy / (float)(nrow - 1);
    const int cc = c0 + ((k >> 1) & 1), rr = r0 + (k >> 2);
    float xa = -sx + dx * (float)cc, ya = -sy + dy * (float)rr, xb = xa + dx, yb = ya + dy;
    V3 v00 = v3(xa, ya, data[rr * ncol + cc] * sz), v10 = v3(xb, ya, data[rr * ncol + cc + 1] * sz);
    V3 v01 = v3(xa, yb, data[(rr + 1) * ncol + cc] * sz), v11 = v3(xb, yb, data[(rr + 1) * ncol + cc + 1] * sz);
    q = (k & 1) ? closest_on_triangle(p, v00, v11, v01) : closest_on_triangle(p, v00, v10, v11);
    V3 d = p - q; d2 = dot(d, d);
  }
  // arg-min over the eight lanes of the pair, first of equal minima
  float bv = d2; int bi = k;
#pragma unroll
  for (int stp = 0; stp < 3; ++stp) {
    float ov; int oi;
    if (stp == 0) { ov = dpp_mov<0xB1>(bv); oi = __builtin_bit_cast(int, dpp_mov<0xB1>(__builtin_bit_cast(float, bi))); }
    else if (stp == 1) { ov = dpp_mov<0x4E>(bv); oi = __builtin_bit_cast(int, dpp_mov<0x4E>(__builtin_bit_cast(float, bi))); }
    else { ov = dpp_mov<0x141>(bv); oi = __builtin_bit_cast(int, dpp_mov<0x141>(__builtin_bit_cast(float, bi))); }
    const bool take = ov < bv || (ov == bv && oi < bi);
    bv = take ? ov : bv; bi = take ? oi : bi;
  }
  // back to the pair's lane: lane f reads its group's winner
  const int grp = lane < NPAIR ? 8 * lane : 0;
  const int kw = __shfl(bi, grp);
  const float best = __shfl(bv, grp);
  const V3 qw = v3(__shfl(q.x, grp + kw), __shfl(q.y, grp + kw), __shfl(q.z, grp + kw));
  if (lane < NPAIR && j.state == 2) { j.q = qw; j.best = best; }
}
__device__ __forceinline__ bool hfield_finish(V3 hpos, const float* hmat, float radius, const HfJob& j, float& dist, V3& pos, V3& nrm) {
  if (j.state == 0) return false;
  V3 n = j.n, q = j.q;
  dist = j.dist;
  if (j.state == 2) {
    const V3 p = j.p;
    float dn = fsqrt(j.best);
    n = v3(0, 0, 1);
    if (!(dn < 1e-12f)) { const float rdn = frcp(dn); n = v3((p.x - q.x) * rdn, (p.y - q.y) * rdn, (p.z - q.z) * rdn); }
    dist = dn - radius;
  }
  V3 pl = q + n * (0.5f * dist);
  nrm = mulv(hmat, n);
  pos = hpos + mulv(hmat, pl);
  return true;
}

template <class C>
__device__ __forceinline__ void collision(const DModel& m, const Hot& h, Smem<C>& s, int lane PROF_ARG) {
  CPts pts; pts.cnt = 0; pts.t = v3(0, 0, 0);
  ClipJob job; job.kind = 0;
  float incl = 0.0f;
  const int lr = lrec_lane(lane);
  const int4 rp0 = lrec<C>(h, LQ_P_0, lr), rp1 = lrec<C>(h, LQ_P_1, lr), rp2 = lrec<C>(h, LQ_P_2, lr);
  HfJob hf; hf.state = 0;          // (Dims::HFIELD; the other fields are written by hfield_place and read only where state != 0)
  int hf_g1 = 0; float hf_radius = 0.0f; V3 hf_hpos = v3(0, 0, 0);
  if (lane < C::NP) {
    const int g1 = rp0.x, g2 = rp0.y, kind = rp0.z;
    incl = asf(rp0.w);
    const V3 size1 = v3_xyz(rp1), size2 = v3_xyz(rp2);
    V3 p1 = ld3(&s.x.a.gpos[3 * g1]), p2 = ld3(&s.x.a.gpos[3 * g2]);
    // both frames up front: inside the SAT the eighteen loads would be issued next to their uses, a few at a time
    M33 R1, R2;
#pragma unroll
    for (int c = 0; c < 9; ++c) { R1.m[c] = s.x.a.gmat[9 * g1 + c]; R2.m[c] = s.x.a.gmat[9 * g2 + c]; }
    if (kind == PAIR_PLANE_BOX) plane_box_sat(p1, R1.m, p2, R2.m, &s.x.a.gmat[9 * g2], size2, pts, job);
    else if (kind == PAIR_BOX_BOX)
      box_box_sat(p1, R1.m, size1, p2, R2.m, size2, pts, job);
    else if (kind == PAIR_PLANE_SPHERE) {
      V3 n = col(R1.m, 2);
      float r = size2.x;
      float dist = dot(p2 - p1, n) - r;
      pts.n = n; pts.dist[0] = dist; pts.pos[0] = p2 - n * (r + 0.5f * dist); pts.cnt = 1;
    }
    if constexpr (C::CAPS) {
      // mujoco-mjx 3.2.x collision_primitive.plane_capsule / plane_cylinder, as restated in the oracle (same operations in the same order)
      if (kind == PAIR_PLANE_CAPSULE) {
        const V3 n = col(R1.m, 2), axis = col(R2.m, 2);
        const float na = dot(n, axis);
        V3 b = axis - n * na;
        const float bn = fsqrt(dot(b, b));
        if (bn < 0.5f) b = (n.y > -0.5f && n.y < 0.5f) ? v3(0, 1, 0) : v3(0, 0, 1);
        else b = b * frcp(bn);
        const float r = size2.x, hl = size2.y;
        pts.n = n; pts.t = b; pts.cnt = 2;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float sg = i == 0 ? 1.0f : -1.0f;
          const V3 ctr = p2 + (axis * sg) * hl;
          const float dist = dot(ctr - p1, n) - r;
          pts.dist[i] = dist; pts.pos[i] = ctr - n * (r + 0.5f * dist);
        }
      } else if (kind == PAIR_PLANE_CYLINDER) {
        const V3 n = col(R1.m, 2);
        V3 axis = col(R2.m, 2);
        float prjaxis = dot(n, axis);
        const float sign = prjaxis < 0.0f ? 1.0f : -1.0f;
        axis = axis * sign; prjaxis *= sign;
        const float r = size2.x, hl = size2.y;
        const float dist0 = dot(p2 - p1, n);
        V3 vec = axis * prjaxis - n;
        const float len = fsqrt(dot(vec, vec));
        if (len < 1e-12f) vec = col(R2.m, 0) * r;
        else { const float rl = frcp(len); vec = v3(vec.x * rl * r, vec.y * rl * r, vec.z * rl * r); }
        const float prjvec = dot(vec, n);
        axis = axis * hl; prjaxis *= hl;
        const float prjvec1 = -prjvec * 0.5f;
        V3 vec1 = cross(vec, axis);
        const float l1 = fsqrt(dot(vec1, vec1));
        const float s3 = 1.7320508075688772f;
        const float rl1 = l1 > 0.0f ? frcp(l1) : 0.0f;
        vec1 = v3(vec1.x * rl1 * r * s3 * 0.5f, vec1.y * rl1 * r * s3 * 0.5f, vec1.z * rl1 * r * s3 * 0.5f);
        const float d1 = dist0 + prjaxis + prjvec, d2 = dist0 + prjaxis + prjvec1;
        pts.n = n; pts.cnt = 3;
        pts.dist[0] = d1; pts.dist[1] = d2; pts.dist[2] = d2;
        pts.pos[0] = p2 + axis + vec - n * d1 * 0.5f;
        pts.pos[1] = p2 + axis + vec1 + vec * -0.5f - n * d2 * 0.5f;
        pts.pos[2] = p2 + axis - vec1 + vec * -0.5f - n * d2 * 0.5f;
        if (fabsf(prjaxis) < 1e-3f) {          // lying parallel to the plane: the second point moves to the other disk
          const float d3 = dist0 - prjaxis + prjvec;
          pts.dist[1] = d3; pts.pos[1] = p2 + vec - axis - n * d3 * 0.5f;
        }
      }
    }
    if constexpr (C::HFIELD) {
      if (kind == PAIR_HFIELD_SPHERE) { hfield_place(m, p1, &s.x.a.gmat[9 * g1], p2, size2.x, hf); hf_g1 = g1; hf_radius = size2.x; hf_hpos = p1; }
    }
  }
  if constexpr (C::HFIELD) {
    if (__ballot(hf.state == 2) != 0ull) hfield_search<C::NP>(m, lane, hf);       // (wave-uniform: every lane takes part)
    if (hf.state != 0) {
      float dist; V3 pos, n;
      if (hfield_finish(hf_hpos, &s.x.a.gmat[9 * hf_g1], hf_radius, hf, dist, pos, n)) { pts.n = n; pts.dist[0] = dist; pts.pos[0] = pos; pts.cnt = 1; }
    }
  }
  PROF(PS_X0)
  // Manifolds of the touching pairs, eight at a time, EIGHT LANES PER PAIR (lane = 8 * slot + vertex).  The owner lane of a
  // pending pair publishes its job in LDS; the plane-box supports / the Sutherland-Hodgman passes then run with one lane per
  // polygon vertex -- a pass is one batch of loads, a few compares, two ballots for the output positions and one batch of
  // stores, instead of a serial walk over the vertices with a dependent LDS round trip each (that walk was 7 % of the
  // kernel's cycles); the owner lane finishes with the <= 4-point manifold selection.  Same arithmetic per vertex and the
  // same output order (kept vertex, then the edge's intersection), so the contacts are bit-identical.
  if constexpr (C::CONDIM != 3) {
    constexpr int GS = 8;                                    // pairs per round
    float* const poly = s.x.a.clip;                          // [GS][2][24]: ping-pong polygons (x[8], y[8], depth[8])
    float* const par = s.x.a.clip + GS * 48;                 // [GS][32]: job parameters, [30] = final vertex count, [31] = kind
    unsigned long long pend = __ballot(job.kind != 0);
    while (pend) {
      const int rank = __popcll(pend & ((1ull << lane) - 1ull));
      const bool mine = job.kind != 0 && rank < GS;
      if (mine) {
        float* q = par + rank * 32;
        q[31] = (float)job.kind;
        if (job.kind == 1) {
          st3(q, job.bp); st3(q + 3, job.size);
#pragma unroll
          for (int c = 0; c < 9; ++c) q[6 + c] = job.bm[c];
          st3(q + 15, job.o); st3(q + 18, job.nref); st3(q + 21, job.axu); st3(q + 24, job.axv);
          q[27] = job.hu;
        } else {
          q[0] = job.hu; q[1] = job.hv;
          st3(q + 15, job.o); st3(q + 18, job.nref); st3(q + 21, job.axu); st3(q + 24, job.axv);
          float* P0 = poly + rank * 48;
#pragma unroll
          for (int i = 0; i < 4; ++i) { P0[i] = job.px[i]; P0[8 + i] = job.py[i]; P0[16 + i] = job.pd[i]; }
        }
      }
      const int nslots = __popcll(pend) < GS ? __popcll(pend) : GS;
      WSYNC();
      {
        const int slot = lane >> 3, v = lane & 7;
        const float* q = par + slot * 32;
        const int kind = slot < nslots ? (int)q[31] : 0;
        float* P = poly + slot * 48;
        // this lane's point of its pair's candidate set: in-plane coordinates (mx, my) and the third value (plane-box: support of
        // the vertex; box-box: depth of the clipped polygon's vertex), and where it would put a contact (cw: the box vertex itself)
        float mx = 0.0f, my = 0.0f, mz = 0.0f;
        V3 cw = v3(0, 0, 0);
        const V3 jo = ld3(q + 15), jn = ld3(q + 18), jau = ld3(q + 21), jav = ld3(q + 24);      // (kind 0: stale words, unused)
        if (kind == 1) {                                     // plane-box: vertex v's support and in-plane coordinates
          V3 bp = ld3(q), size = ld3(q + 3);
          cw = box_vertex(q + 6, bp, size, v);
          mz = dot(jo - cw, jn); mx = dot(cw, jau); my = dot(cw, jav);
        }
        // box-box: four clipping passes against the reference rectangle
        const bool bb = kind == 2;
        const float hu = bb ? q[0] : 0.0f, hv = bb ? q[1] : 0.0f;
        int np = bb ? 4 : 0, cur = 0;
        const unsigned long long slotbits = 0xFFull << (8 * slot), below = slotbits & ((1ull << lane) - 1ull);
        const bool any_bb = __ballot(bb) != 0ull;
        if (any_bb) {
          for (int side = 0; side < 4; ++side) {
            const float h = (side < 2) ? hu : hv, sg = (side & 1) ? -1.0f : 1.0f;
            const float* Pc = P + cur * 24; float* Qn = P + (1 - cur) * 24;
            const bool act = bb && v < np;
            const int v1 = act ? v : 0, v2 = act ? ((v + 1 == np) ? 0 : v + 1) : 0;
            const float x1 = Pc[v1], y1 = Pc[8 + v1], z1 = Pc[16 + v1], x2 = Pc[v2], y2 = Pc[8 + v2], z2 = Pc[16 + v2];
            const float d1 = h - sg * (side < 2 ? x1 : y1), d2 = h - sg * (side < 2 ? x2 : y2);
            const bool e1 = act && d1 >= 0.0f, e2 = act && ((d1 >= 0.0f) != (d2 >= 0.0f));
            const unsigned long long b1 = __ballot(e1), b2 = __ballot(e2);
            const int pos = __popcll(b1 & below) + __popcll(b2 & below);
            if (e1) { Qn[pos] = x1; Qn[8 + pos] = y1; Qn[16 + pos] = z1; }      // (the reads were from the other buffer)
            if (e2) {
              const float tt = d1 * frcp(d1 - d2);
              const int pi = pos + (e1 ? 1 : 0);
              Qn[pi] = x1 + tt * (x2 - x1); Qn[8 + pi] = y1 + tt * (y2 - y1); Qn[16 + pi] = z1 + tt * (z2 - z1);
            }
            np = bb ? __popcll(b1 & slotbits) + __popcll(b2 & slotbits) : 0;
            cur = 1 - cur;
            WSYNC();
          }
          if (bb && v < np) { mx = P[v]; my = P[8 + v]; mz = P[16 + v]; }      // four passes: the result is back in the first buffer
        }
        // ---- the <= 4-point manifold (oracle manifold_points: first candidate, farthest from it, farthest from their line, farthest
        // on the other side), one lane per candidate: a pass is one value per lane and an arg-max over the pair's eight lanes (three
        // DPP steps on (value, index), the first maximum wins as in the serial scan) instead of a scan over eight register copies
        // by the pair's owner lane, once per kind of pair present.  Same arithmetic per candidate, same picks.
        const int n = kind == 1 ? 8 : np;
        const bool valid = v < n;
        const bool on = valid && (kind == 1 ? mz > fmaxf(q[27] - 1e-3f, 0.0f) : mz > 0.0f);
        const unsigned gm = (unsigned)(__ballot(on) >> (8 * slot)) & 0xFFu;
        const int g0 = lane & ~7;
        const float NEG = -1e6f, pen = on ? 0.0f : NEG;
        auto argmax8 = [&](float val, int& wi) {          // -> index of the first maximum of val over the group's valid lanes
          float bv = valid ? val : -3.0e38f; int bi = v;
#pragma unroll
          for (int st = 0; st < 3; ++st) {
            float ov; int oi;
            if (st == 0) { ov = dpp_mov<0xB1>(bv); oi = __builtin_bit_cast(int, dpp_mov<0xB1>(__builtin_bit_cast(float, bi))); }
            else if (st == 1) { ov = dpp_mov<0x4E>(bv); oi = __builtin_bit_cast(int, dpp_mov<0x4E>(__builtin_bit_cast(float, bi))); }
            else { ov = dpp_mov<0x141>(bv); oi = __builtin_bit_cast(int, dpp_mov<0x141>(__builtin_bit_cast(float, bi))); }
            const bool take = ov > bv || (ov == bv && oi < bi);
            bv = take ? ov : bv; bi = take ? oi : bi;
          }
          wi = bi;
          return bv;
        };
        const int ia = gm ? __builtin_ctz(gm) : 0;
        const float xa = __shfl(mx, g0 + ia), ya = __shfl(my, g0 + ia);
        int ib, ic, id;
        {
          const float dx = xa - mx, dy = ya - my;
          argmax8(dx * dx + dy * dy + pen, ib);
        }
        const float xb = __shfl(mx, g0 + ib), yb = __shfl(my, g0 + ib);
        const float abx = -(ya - yb), aby = (xa - xb);
        const float crs = (xa - mx) * abx + (ya - my) * aby;
        argmax8(fabsf(crs) + pen, ic);
        const float crs_c = __shfl(crs, g0 + ic);
        const float sgn = crs_c > 0 ? 1.0f : -1.0f;
        {
          const float best = argmax8(on ? -sgn * crs : -1.0f, id);
          if (!(best > 0.0f)) id = ic;
        }
        // contacts in the order a, b, c, d without repeats (a, b, c are candidates by construction; the checks are the oracle's)
        const bool use_b = ib != ia && ((gm >> ib) & 1u), use_c = ic != ia && ic != ib && ((gm >> ic) & 1u);
        const bool use_d = id != ia && id != ib && id != ic && ((gm >> id) & 1u);
        const int cnt = gm ? 1 + (use_b ? 1 : 0) + (use_c ? 1 : 0) + (use_d ? 1 : 0) : 0;
        int k = -1;
        if (gm && v == ia) k = 0;
        else if (use_b && v == ib) k = 1;
        else if (use_c && v == ic) k = 1 + (use_b ? 1 : 0);
        else if (use_d && v == id) k = 1 + (use_b ? 1 : 0) + (use_c ? 1 : 0);
        float* const R = P + 24;                             // the second polygon buffer: [4][4] = dist, pos
        if (kind != 0 && k >= 0) {
          const float dist = -mz;
          const V3 cp = kind == 1 ? cw - jn * (0.5f * dist) : jo + jau * mx + jav * my - jn * (0.5f * mz);
          R[4 * k] = dist; st3(&R[4 * k + 1], cp);
        }
        if (kind != 0 && v == 0) par[slot * 32 + 30] = (float)cnt;
      }
      WSYNC();
      if (mine) {
        const float* R = poly + rank * 48 + 24;
        const int cnt = (int)par[rank * 32 + 30];
        float r[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) r[i] = R[i];
        if (job.kind == 2) pts.n = job.flip ? job.nref * -1.0f : job.nref;
#pragma unroll
        for (int i = 0; i < 4; ++i) { pts.dist[i] = r[4 * i]; pts.pos[i] = V3{r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]}; }
        pts.cnt = cnt;
        job.kind = 0;
      }
      WSYNC();
      pend = __ballot(job.kind != 0);
    }
  }
  PROF(PS_X1)
  // keep penetrating contacts only (result-neutral culling), compact in pair order
  int keep = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) if (i < pts.cnt && pts.dist[i] - incl < 0.0f) keep++;
  // exclusive prefix sum of keep (0..4) over the lanes from three ballots, one per bit of the count: no cross-lane data moves
  const unsigned long long below = (1ull << lane) - 1ull;
  const unsigned long long k0 = __ballot(keep & 1), k1 = __ballot(keep & 2), k2 = __ballot(keep & 4);
  const int base = __popcll(k0 & below) + 2 * __popcll(k1 & below) + 4 * __popcll(k2 & below);
  const int total = __popcll(k0) + 2 * __popcll(k1) + 4 * __popcll(k2);
  int w = base;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (!(i < pts.cnt && pts.dist[i] - incl < 0.0f)) continue;
    if (w < C::NCON) {
      s.cdist[w] = pts.dist[i]; st3(&s.cpos[3 * w], pts.pos[i]); st3(&s.cnrm[3 * w], pts.n); s.cpair[w] = lane;
      if constexpr (C::CAPS) st3(&s.ctan[3 * w], pts.t);
    }
    w++;
  }
  if (lane == 0) { s.ncon = total < C::NCON ? total : C::NCON; s.ncon_drop = total > C::NCON ? total - C::NCON : 0; }
  WSYNC();
  PROF(PS_X2)
}

// =====================================================================================
// stage 6: velocity-dependent terms (com_vel, passive, rne, actuation) -> qfrc_smooth (lane = dof)
// =====================================================================================
__device__ __forceinline__ void motion_cross(float* o, const float* u, const float* v) {
  V3 ua = ld3(u), ul = ld3(u + 3), va = ld3(v), vl = ld3(v + 3);
  st3(o, cross(ua, va));
  st3(o + 3, cross(ul, va) + cross(ua, vl));
}

struct Q6 { float qd; float c[6]; };      // one chain dof: its velocity and a spatial vector
template <class C>
__device__ __forceinline__ float smooth_forces(const DModel& m, const Hot& h, Smem<C>& s, int lane, float qvel_i, float ctrl_u PROF_ARG) {
  const int lr = lrec_lane(lane);
  const int4 rb_misc = lrec<C>(h, LQ_B_MISC, lr), rd_ids = lrec<C>(h, LQ_D_IDS, lr), rd_masks = lrec<C>(h, LQ_D_MASKS, lr);
  const int4 rd_act = lrec<C>(h, LQ_D_ACT, lr), rd_ctrl = lrec<C>(h, LQ_D_CTRL, lr), rd_bias = lrec<C>(h, LQ_D_BIAS, lr), rd_frc = lrec<C>(h, LQ_D_FRC, lr);
  PROF_VMWAIT(); PROF(PS_F_LOAD)
  const int max_sub = h.max_sub, max_chain = h.max_chain;
  const float grav0 = h.grav0, grav1 = h.grav1, grav2 = h.grav2;
  const int qb = lane >> 2;            // four lanes per body for the subtree force sums below (see com_crb_mass)
  const unsigned frc_mask_q = (unsigned)__shfl(rb_misc.z, qb < C::NB ? qb : 0);
  const int site_b = lrec<C>(h, LQ_S_POS, lr).x, site_root = __shfl(rb_misc.y, site_b);   // lanes >= NS: body 0
  int xfrc_b = 0, xfrc_root = 0; unsigned xfrc_dofs = 0u;
  if constexpr (C::XFRC) {
    xfrc_b = s.xfrc_body;
    const int src = xfrc_b > 0 && xfrc_b < C::NB ? xfrc_b : 0;
    xfrc_root = __shfl(rb_misc.y, src); xfrc_dofs = (unsigned)__shfl(rb_misc.w, src);
  }
  // cvel[b] = sum over the dofs on b's chain of cdof*qvel (four lanes per body)
  const unsigned dof_mask_raw = (unsigned)__shfl(rb_misc.w, qb < C::NB ? qb : 0);      // unconditional: source lanes must be active
  const unsigned dof_mask_q = qb < C::NB ? dof_mask_raw : 0u;
  {
    float v[6];
    chain_sum_quad<C>(s, s.cdof, dof_mask_q, lane, max_chain, v);
    if ((lane & 3) == 0 && qb < C::NB) {
#pragma unroll
      for (int c = 0; c < 6; ++c) s.x.a.cvel[6 * qb + c] = v[c];
    }
  }
  // cdof_dot[i] = (velocity of the chain before dof i) x cdof[i]
  if (lane < C::NV) {
    unsigned mask = (unsigned)rd_masks.z;
    bool free_trans = (rd_ids.z == JNT_FREE) && (rd_ids.w < 3);
    float v[6] = {0, 0, 0, 0, 0, 0};
    for_bits4_gather(mask, max_chain,
      [&](int i) { Q6 r; r.qd = s.qvel[i];
#pragma unroll
                   for (int c = 0; c < 6; ++c) r.c[c] = s.cdof[6 * i + c];
                   return r; },
      [&](int, bool on, const Q6& r) { float qd = on ? r.qd : 0.0f;
#pragma unroll
                   for (int c = 0; c < 6; ++c) v[c] += r.c[c] * qd; });
    float o[6];
    motion_cross(o, v, &s.cdof[6 * lane]);
#pragma unroll
    for (int c = 0; c < 6; ++c) s.x.a.cdofdot[6 * lane + c] = free_trans ? 0.0f : o[c];
  }
  WSYNC();
  PROF(PS_F_VEL)
  // object velocity at every site, world frame (source of gyro / velocimeter / framelinvel / frameangvel sensors)
  if (lane < C::NS) {
    const int sb = site_b;
    V3 w = ld3(&s.x.a.cvel[6 * sb]), v = ld3(&s.x.a.cvel[6 * sb + 3]);
    V3 off = ld3(&s.spos[3 * lane]) - ld3(&s.com[3 * site_root]);
    st3(&s.sangvel[3 * lane], w); st3(&s.slinvel[3 * lane], v + cross(w, off));
  }
  // cacc[b] = [0, -g] + sum over chain dofs of cdof_dot*qvel ; local force = I*cacc + cvel x* (I*cvel)  (chain sum with
  // four lanes per body; the quad's first lane finishes body b)
  float aq[6];
  chain_sum_quad<C>(s, s.x.a.cdofdot, dof_mask_q, lane, max_chain, aq);
  if ((lane & 3) == 0 && qb < C::NB) {
    float a[6] = {aq[0], aq[1], aq[2], aq[3] - grav0, aq[4] - grav1, aq[5] - grav2};
    if constexpr (C::XFRC) {
      if (qb == s.acc_body) {
#pragma unroll
        for (int c = 0; c < 6; ++c) s.accb[c] = a[c];
      }
    }
    float f1[6], f2[6];
    inert_mul(f1, &s.x.a.cinert[10 * qb], a);
    inert_mul(f2, &s.x.a.cinert[10 * qb], &s.x.a.cvel[6 * qb]);
    V3 va = ld3(&s.x.a.cvel[6 * qb]), vl = ld3(&s.x.a.cvel[6 * qb + 3]), fa = ld3(f2), fl = ld3(f2 + 3);
    V3 ta = cross(va, fa) + cross(vl, fl), tl = cross(va, fl);
    s.x.a.cfrc[6 * qb + 0] = f1[0] + ta.x; s.x.a.cfrc[6 * qb + 1] = f1[1] + ta.y; s.x.a.cfrc[6 * qb + 2] = f1[2] + ta.z;
    s.x.a.cfrc[6 * qb + 3] = f1[3] + tl.x; s.x.a.cfrc[6 * qb + 4] = f1[4] + tl.y; s.x.a.cfrc[6 * qb + 5] = f1[5] + tl.z;
  }
  WSYNC();
  PROF(PS_F_FRC)
  subtree_sum_quad<C, 6>(s.x.a.cfrc, s.x.a.cfrcsum, qb >= C::NB ? 0u : frc_mask_q, lane);
  WSYNC();
  PROF(PS_F_SUM)
  float smooth = 0.0f;
  if (lane < C::NV) {
    int i = lane, b = rd_ids.y;
    float bias = 0;
#pragma unroll
    for (int c = 0; c < 6; ++c) bias += s.cdof[6 * i + c] * s.x.a.cfrcsum[6 * b + c];
    float passive = -s.damp[i] * qvel_i;
    // actuation: joint transmission, at most one actuator per dof (dof_act)
    float act = 0;
    const int u = rd_act.x;
    if (u >= 0) {
      float gear = asf(rd_act.y);
      float length = s.qpos[rd_act.z] * gear, velocity = qvel_i * gear;
      float ctrl = s.ctrl[u];
      if (rd_act.w) ctrl = clampf(ctrl, asf(rd_ctrl.x), asf(rd_ctrl.y));
      float gain0, bias0, bias1, bias2;
      if constexpr (C::DREX) { gain0 = s.dx_gain[3 * u]; bias0 = s.dx_bias[3 * u]; bias1 = s.dx_bias[3 * u + 1]; bias2 = s.dx_bias[3 * u + 2]; }
      else { gain0 = asf(rd_ctrl.z); bias0 = asf(rd_ctrl.w); bias1 = asf(rd_bias.x); bias2 = asf(rd_bias.y); }
      float force = gain0 * ctrl + bias0 + bias1 * length + bias2 * velocity;
      if (rd_bias.z) force = clampf(force, asf(rd_bias.w), asf(rd_frc.x));
      s.aforce[u] = force;
      act = gear * force;
    }
    if (rd_frc.y) act = clampf(act, asf(rd_frc.z), asf(rd_frc.w));
    smooth = passive - bias + act;
    if constexpr (C::XFRC) {     // support.xfrc_accumulate: J(xipos[body])^T force
      const int xb = xfrc_b;
      if (xb > 0 && ((xfrc_dofs >> i) & 1)) {
        V3 off = ld3(&s.x.a.xipos[3 * xb]) - ld3(&s.com[3 * xfrc_root]);
        V3 jp = ld3(&s.cdof[6 * i + 3]) + cross(ld3(&s.cdof[6 * i]), off);
        smooth += jp.x * s.xfrc[0] + jp.y * s.xfrc[1] + jp.z * s.xfrc[2];
      }
    }
  }
  (void)ctrl_u;
  return smooth;
}

// =====================================================================================
// stage 5: constraint rows.  Row order: equality, dof friction, active limits, contacts (6 pyramid
// edges each).  J lives in LDS; the per-row scalars are returned in lane registers (row = lane + 64*c).
// =====================================================================================
// per pyramid row: floss < 0 marks "not a friction row"; D = 0 marks padding.
// Row = base[bn] + mu * base[bk] (mu = 0 and bk = bn for the sparse rows).
struct RowRegs { float aref, D, R, floss, mu; int bn, bk; };

struct Solimp { float v[5]; };
// (stiffness k and damping b of the row come from the lane records: model constants, computed on the host by
// model.stiffness_damping() exactly as this function computed them per row and substep)
__device__ __forceinline__ void kbi(const Hot& m, float k_in, float b_in, const Solimp& sip, float pos, float& k, float& b, float& imp) {
  const float* si = sip.v;
  // (d0, d_width, 1 / width, midpoint, power), clamped on the host: model.impedance_consts()
  const float dmin = si[0], dmax = si[1], rwidth = si[2], mid = si[3], power = si[4];
  k = k_in; b = b_in;
  float x = fabsf(pos) * rwidth;
  float ia, ib;
  if (power == 2.0f) {       // MuJoCo's default; x*x is the correctly rounded square, as pow(x, 2) is
    ia = frcp(mid) * (x * x);
    ib = 1.0f - frcp(1.0f - mid) * ((1.0f - x) * (1.0f - x));
  } else {
    ia = (1.0f / powf(mid, power - 1.0f)) * powf(x, power);
    ib = 1.0f - (1.0f / powf(1.0f - mid, power - 1.0f)) * powf(1.0f - x, power);
  }
  float y = x < mid ? ia : ib;
  imp = clampf(dmin + y * (dmax - dmin), dmin, dmax);
  if (x > 1.0f) imp = dmax;
}

// Returns nefc; rr[] holds the per-row scalars except the velocity part of aref: the caller finishes
// aref -= bcoef * (J.qvel) once the base rows are in LDS (it owns the J.v machinery).
template <class C>
__device__ __forceinline__ int make_constraint(const DModel& m, const Hot& h, Smem<C>& s, int lane, RowRegs (&rr)[C::NCHUNK], float (&bcoef)[C::NCHUNK],
                               int& nbase_out PROF_ARG) {
  constexpr int LD = C::LDJ;      // every LD below strides the Jacobian
  const int lr = lrec_lane(lane);
  const int4 rl0 = lrec<C>(h, LQ_L_0, lr), rl1 = lrec<C>(h, LQ_L_1, lr);
  const int4 re0 = lrec<C>(h, LQ_E_0, lr), re1 = lrec<C>(h, LQ_E_1, lr), re2 = lrec<C>(h, LQ_E_2, lr), re3 = lrec<C>(h, LQ_E_3, lr), re4 = lrec<C>(h, LQ_E_4, lr);
  const int4 rf0 = lrec<C>(h, LQ_F_0, lr);
  const gp_f qpos0_tab = m.qpos0;
  // active joint limits, compacted in slot order (slot = index into limit_jnts; its constants are record LQ_L_*[slot])
  int lim_active = 0;
  if (lane < C::NL) {
    float q = s.qpos[rl0.x];
    float dmin = q - asf(rl0.z), dmax = asf(rl0.w) - q;
    lim_active = (fminf(dmin, dmax) - asf(rl1.x)) < 0.0f;
  }
  unsigned long long bal = __ballot(lim_active);
  int nl = __popcll(bal);
  if (lim_active) s.lim_jnt[__popcll(bal & ((1ull << lane) - 1ull))] = lane;
  const int r_fric = C::NEQ, r_lim = C::NEQ + C::NF, r_con = r_lim + nl;
  const int ncon = s.ncon;
  const int nefc = r_con + C::NPYR * ncon;
  WSYNC();
  // ---- every record this stage reads through a data-dependent index (pair of a contact, slot of an active limit), fetched
  // here in ONE batch: left where they are used -- inside the loops and the per-row branches below -- each fetch was a global
  // round trip of its own behind an LDS read, five to seven in a row per substep.
  //  * lane c < ncon: the pair record of contact c (bodies / dof masks, geom slots, friction rule) -> per-contact table in LDS
  //    (s.wc, free until the Hessian), read by the base-row and friction loops;
  //  * lane < nl: limit record of its compacted slot;
  //  * per constraint row of this lane: the four quads its regulariser needs (A..D, by row type).
  int4 ct3, ct0, ct2, lq0;
  {
    const int p = s.cpair[lane < ncon ? lane : 0];
    const int pi = lane < ncon ? p : 0;
    ct3 = h.lane_rec[LQ_P_3 * 64 + pi]; ct0 = h.lane_rec[LQ_P_0 * 64 + pi]; ct2 = h.lane_rec[LQ_P_2 * 64 + pi];
    const int sl = lane < nl ? s.lim_jnt[lane] : 0;
    lq0 = h.lane_rec[LQ_L_0 * 64 + sl];
  }
  int4 qa[C::NCHUNK], qb[C::NCHUNK], qc[C::NCHUNK], qd[C::NCHUNK];
#pragma unroll
  for (int ch = 0; ch < C::NCHUNK; ++ch) {
    const int r = lane + 64 * ch;
    int base = LQ_F_0, idx = 0, third = 2, fourth = 2;            // (rows without a record: any valid address)
    if (r >= r_fric && r < r_lim) { idx = r - r_fric; }
    else if (r >= r_lim && r < r_con) { base = LQ_L_0; idx = s.lim_jnt[r - r_lim]; fourth = 3; }
    else if (r >= r_con && r < nefc) { base = LQ_P_0; idx = s.cpair[(r - r_con) / C::NPYR]; third = 4; fourth = 5; }
    const gp_q tab = h.lane_rec + base * 64 + idx;
    qa[ch] = tab[0]; qb[ch] = tab[64]; qc[ch] = tab[third * 64]; qd[ch] = tab[fourth * 64];
  }
  float q0_e1 = 0.0f, q0_e0 = 0.0f;
  if constexpr (C::DREX) { if (lane < C::NEQ) { q0_e1 = s.dx_qpos0[re1.y]; q0_e0 = s.dx_qpos0[re0.y]; } }
  else { const int e1 = lane < C::NEQ ? re1.y : 0, e0 = lane < C::NEQ ? re0.y : 0; q0_e1 = qpos0_tab[e1]; q0_e0 = qpos0_tab[e0]; }
  // zero the sparse rows, then poke their entries
  for (int t = lane; t < r_con * LD; t += 64) s.x.b.J[t] = 0.0f;
  if (lane < LD) s.x.b.J[C::NBASE * LD + lane] = 0.0f;      // null row (phase A used this memory)
  if (lane < 4) { s.bval[C::NBASE + lane] = 0.0f; s.bmu[C::NBASE + lane] = 0.0f; }
  for (int t = lane; t < r_con; t += 64) s.bmu[t] = 0.0f;
  int* const ctab = reinterpret_cast<int*>(s.wc);          // [ncon][8]: dof masks 1, 2; bodies 1, 2; geom slots 1, 2; friction rule
  if (lane < ncon) {
    int* t = &ctab[8 * lane];
    t[0] = ct3.x; t[1] = ct3.y; t[2] = ct3.z; t[3] = ct3.w; t[4] = ct0.x; t[5] = ct0.y; t[6] = ct2.w;
  }
  // the contact frame, once per contact (the base-row loop below visits a contact NV times and rebuilt it every time: two square
  // roots and two divisions per visit): the normalised normal replaces the raw one in cnrm, the tangents go to s.rw (free
  // until the Hessian; its last four words, the null row's weights, are not touched)
  static_assert(6 * C::NCON <= C::NEFC, "contact tangents are staged in the row-weight array");
  if (lane < ncon) {
    V3 nn, t1, t2;
    make_frame(ld3(&s.cnrm[3 * lane]), nn, t1, t2);
    if constexpr (C::CAPS) {           // a capsule's frame: (n, b, n x b) with b along the capsule axis, n as the narrow phase gave it
      const V3 b = ld3(&s.ctan[3 * lane]);
      if (b.x != 0.0f || b.y != 0.0f || b.z != 0.0f) { nn = ld3(&s.cnrm[3 * lane]); t1 = b; t2 = cross(nn, b); }
    }
    st3(&s.cnrm[3 * lane], nn); st3(&s.rw[6 * lane], t1); st3(&s.rw[6 * lane + 3], t2);
  }
  if constexpr (C::ROWTREE) {
    // does a contact pair of this substep join two kinematic trees?  (equality rows stay inside a tree: checked on the host)
    constexpr unsigned T0 = (1u << C::TREE1) - 1u, T01 = (1u << C::TREE2) - 1u, TALL = C::NV >= 32 ? 0xFFFFFFFFu : (1u << C::NV) - 1u;
    const unsigned dm = lane < ncon ? (unsigned)(ct3.x | ct3.y) : 0u;
    const int touched = ((dm & T0) != 0u) + ((dm & (T01 & ~T0)) != 0u) + ((dm & (TALL & ~T01)) != 0u);
    const bool joined = __ballot(touched >= 2) != 0ull;
    if (lane == 0) s.trees_joined = joined ? 1 : 0;
    const int mytree = (dm & T0) != 0u ? 0 : ((dm & (T01 & ~T0)) != 0u ? 1 : 2);
    if constexpr (C::tree_size(2) > 0)             // (the per-tree contact walk of the Hessian is used by three-tree models only)
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const unsigned long long in_t = __ballot(lane < ncon && mytree == t);
      if (lane < ncon && mytree == t) s.tree_con[t * C::NCON + __popcll(in_t & ((1ull << lane) - 1ull))] = (unsigned char)lane;
      if (lane == 0) s.tree_ncon[t] = __popcll(in_t);
    }
  }
  WSYNC();
  // equality e = lane (joint coupling): dif = qpos[j2] - qpos0[j2], polynomial data[0..4]
  float eq_dif = 0.0f;
  if (lane < C::NEQ && re1.x) eq_dif = s.qpos[re1.y] - q0_e1;
  if (lane < C::NEQ && re0.x) {
    const int e = lane;
    float deriv = asf(re2.y) + eq_dif * (2.0f * asf(re2.z) + eq_dif * (3.0f * asf(re2.w) + eq_dif * 4.0f * asf(re3.x)));
    if (re1.x) s.x.b.J[e * LD + re1.z] = -deriv;
    s.x.b.J[e * LD + re0.z] = 1.0f;
  }
  if (lane < C::NF) { s.x.b.J[(r_fric + lane) * LD + rf0.x] = 1.0f; s.sdof[r_fric + lane] = rf0.x; }
  if (lane < nl) {
    const int4 q0 = lq0;
    float q = s.qpos[q0.x];
    float dmin = q - asf(q0.z), dmax = asf(q0.w) - q;
    s.x.b.J[(r_lim + lane) * LD + q0.y] = dmin < dmax ? 1.0f : -1.0f;
    s.sdof[r_lim + lane] = q0.y;
  }
  PROF(PS_X3)
  // contact base rows: item (contact c, dof i) fills normal / tangent 1 / tangent 2 / torsion
  for (int t = lane; t < ncon * C::NV; t += 64) {
    int c = t / C::NV, i = t - c * C::NV;
    const int* ct = &ctab[8 * c];
    const unsigned mask1 = (unsigned)ct[0], mask2 = (unsigned)ct[1];
    const int body1 = ct[2], body2 = ct[3];
    const V3 pos = ld3(&s.cpos[3 * c]), nn = ld3(&s.cnrm[3 * c]), t1 = ld3(&s.rw[6 * c]), t2 = ld3(&s.rw[6 * c + 3]);
    V3 ang = ld3(&s.cdof[6 * i]), lin = ld3(&s.cdof[6 * i + 3]);
    float in1 = ((mask1 >> i) & 1) ? 1.0f : 0.0f, in2 = ((mask2 >> i) & 1) ? 1.0f : 0.0f;
    V3 o1 = pos - ld3(&s.com[3 * body1]), o2 = pos - ld3(&s.com[3 * body2]);
    V3 jp = (lin + cross(ang, o2)) * in2 - (lin + cross(ang, o1)) * in1;
    V3 jr = ang * (in2 - in1);
    float* Jr = &s.x.b.J[(r_con + C::NBC * c) * LD + i];
    Jr[0 * LD] = dot(nn, jp); Jr[1 * LD] = dot(t1, jp); Jr[2 * LD] = dot(t2, jp);
    if constexpr (C::NBC > 3) Jr[3 * LD] = dot(nn, jr);
  }
  PROF(PS_X4)
  // friction coefficient of each contact base row (normal: unused)
  for (int t = lane; t < ncon * C::NBC; t += 64) {
    int c = t / C::NBC, k = t - c * C::NBC;
    const int* ct = &ctab[8 * c];
    const int g1 = ct[4], g2 = ct[5], rule = ct[6];       // rule: 0 = max of the two geoms, 1 / 2 = the higher-priority geom's
    float a0 = s.fric[3 * g1], a1 = s.fric[3 * g1 + 1], b0 = s.fric[3 * g2], b1 = s.fric[3 * g2 + 1];
    float f0 = rule == 0 ? fmaxf(a0, b0) : (rule == 1 ? a0 : b0), f1 = rule == 0 ? fmaxf(a1, b1) : (rule == 1 ? a1 : b1);
    s.bmu[r_con + t] = k == 0 ? 0.0f : (k == 3 ? f1 : f0);
  }
  WSYNC();
  PROF(PS_X5)
  // per-row regulariser and reference acceleration (the velocity term J.qvel is added by the caller)
#pragma unroll
  for (int ch = 0; ch < C::NCHUNK; ++ch) {
    int r = lane + 64 * ch;
    RowRegs o{0.0f, 0.0f, 1.0f, -1.0f, 0.0f, C::NBASE, C::NBASE};
    if (r < nefc) {
      float pos = 0, invw = 0, sr0, sr1, fl = -1.0f; Solimp si;
      o.bn = r; o.bk = r;
      const int4 q0 = qa[ch], q1 = qb[ch], q2 = qc[ch], q3 = qd[ch];      // the row's record (fetched above), by row type:
      if (r < r_fric) {                      // equality row r = lane (first chunk)
        float poly = asf(re2.x) + eq_dif * (asf(re2.y) + eq_dif * (asf(re2.z) + eq_dif * (asf(re2.w) + eq_dif * asf(re3.x))));
        pos = s.qpos[re0.y] - q0_e0 - poly;
        invw = asf(re0.w);
        sr0 = asf(re3.y); sr1 = asf(re3.z);
        si.v[0] = asf(re3.w); si.v[1] = asf(re4.x); si.v[2] = asf(re4.y); si.v[3] = asf(re4.z); si.v[4] = asf(re4.w);
      } else if (r < r_lim) {                // friction row: F_0, F_1, F_2 of slot r - r_fric
        invw = asf(q0.y); sr0 = asf(q0.z); sr1 = asf(q0.w);
        si.v[0] = asf(q1.x); si.v[1] = asf(q1.y); si.v[2] = asf(q1.z); si.v[3] = asf(q1.w); si.v[4] = asf(q2.x);
        fl = s.floss[q0.x];
      } else if (r < r_con) {                // active limit: L_0 .. L_3 of the slot from the compaction
        float q = s.qpos[q0.x];
        pos = fminf(q - asf(q0.z), asf(q0.w) - q) - asf(q1.x);
        invw = asf(q1.y); sr0 = asf(q1.z); sr1 = asf(q1.w);
        si.v[0] = asf(q2.x); si.v[1] = asf(q2.y); si.v[2] = asf(q2.z); si.v[3] = asf(q2.w); si.v[4] = asf(q3.x);
      } else {                               // pyramid row of contact c: P_0, P_1, P_4, P_5 of its pair
        int c = (r - r_con) / C::NPYR, e = (r - r_con) - C::NPYR * c;
        pos = s.cdist[c] - asf(q0.w);
        o.bn = r_con + C::NBC * c; o.bk = o.bn + 1 + (e >> 1);
        float f0 = s.bmu[o.bn + 1];
        o.mu = (e & 1) ? -s.bmu[o.bk] : s.bmu[o.bk];
        float tw = asf(q1.w);
        invw = (tw + f0 * f0 * tw) * 2.0f * f0 * f0 * frcp(h.impratio);
        sr0 = asf(q2.x); sr1 = asf(q2.y);
        si.v[0] = asf(q2.z); si.v[1] = asf(q2.w); si.v[2] = asf(q3.x); si.v[3] = asf(q3.y); si.v[4] = asf(q3.z);
      }
      float k, b, imp;
      kbi(h, sr0, sr1, si, pos, k, b, imp);
      float R = fmaxf(invw * (1.0f - imp) * frcp(imp), RSR_MINVAL);
      o.R = R; o.D = frcp(R); o.floss = fl;
      o.aref = -k * imp * pos;        // the caller subtracts b * (J.qvel)
      rr[ch] = o;
      bcoef[ch] = b;
    } else {
      rr[ch] = o;
      bcoef[ch] = 0.0f;
    }
  }
  if (lane == 0) s.nlim_act = nl;
  nbase_out = r_con + C::NBC * ncon;
  return nefc;
}

}  // namespace rsr
