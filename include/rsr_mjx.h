/*
 * rsr_mjx.h -- C ABI of the MI355X-native batched env stepper (librsrmjx.so).
 *
 * The reference has no FFI on this path: the boundary is a Python object protocol traced by JAX
 * (SURVEY.md section 8b).  Each entry point below names the reference interface it stands in for:
 *
 *   rsr_model_create  <- env construction: mujoco.MjModel.from_xml_path + mjcf.load_model
 *                        (reference ppo_train/airbot_training/cube_env.py:37-43).  The MJCF compile
 *                        itself is host Python (rsr_mjx_amd/mjcf.py); the result crosses the ABI as
 *                        one flat "RSRM" blob of named float32/int32 arrays (rsr_mjx_amd/model.py).
 *   rsr_batch_create  <- envs.training.wrap(env, episode_length, action_repeat, randomization_fn)
 *                        + the vmapped state allocation (reference RSR/train.py:224-235)
 *   rsr_batch_set_dr  <- DomainRandomizationVmapWrapper's per-env model leaves
 *                        (reference ppo_train/airbot_training/domain_randomize.py:63-90)
 *   rsr_reset         <- jit(vmap(env.reset))(key_envs)      (RSR/train.py:231-235, cube_env.py:95-143)
 *   rsr_step          <- env.step(state, action) under AutoReset(Episode(Vmap(env)))
 *                        (RSR/train.py:313-319 via acting.generate_unroll; cube_env.py:145-213)
 *   rsr_view          <- reading fields of the returned State pytree (cube_env.py:143, 211-213)
 *
 * Conventions: every function returns 0 on success and a negative code on error;
 * rsr_last_error() returns a thread-local message.  No exceptions cross the ABI.  Handles are opaque
 * and destroyed by the caller.  All device pointers are HIP device pointers on the batch's device.
 * rsr_reset / rsr_step are asynchronous on the given hipStream_t (NULL = default stream).
 * A batch handle is thread-compatible, not thread-safe.  There is no CPU fallback: without a HIP
 * device every compute entry point fails with RSR_ERR_HIP.
 */
#ifndef RSR_MJX_H_
#define RSR_MJX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rsr_model rsr_model;
typedef struct rsr_batch rsr_batch;

enum {
  RSR_OK = 0,
  RSR_ERR_ARG = -1,       /* bad argument / malformed blob */
  RSR_ERR_UNSUPPORTED = -2, /* model dims or env kind the kernels are not built for */
  RSR_ERR_HIP = -3,       /* HIP runtime error (message has the hipError string) */
  RSR_ERR_NOMEM = -4,
  RSR_ERR_HANDOFF = -5    /* rsr_batch_check: a work-unit hand-off inside a step launch timed out */
};

typedef struct rsr_dims {
  int32_t nq, nv, nu, nbody, njnt, ngeom, nsite, neq, npair;
  int32_t obs_dim, nmetrics, n_frames, episode_length, env_kind;
  int32_t rec_floats;        /* floats per env in the persistent state record */
  int32_t ncon_max, nefc_max; /* per-env active-contact / constraint-row capacity of the kernel */
  int32_t lds_bytes;         /* LDS bytes per wavefront (= per env) of the step kernel */
} rsr_dims;

/* Fields of the per-env state record, for rsr_view.  Names follow the reference State pytree. */
enum rsr_field {
  RSR_F_QPOS = 0, RSR_F_QVEL, RSR_F_CTRL, RSR_F_QACC_WARMSTART, RSR_F_TIME, RSR_F_XPOS, RSR_F_SITE_XPOS,
  RSR_F_OBS, RSR_F_REWARD, RSR_F_DONE, RSR_F_METRICS,
  RSR_F_INFO_TARGET_POS, RSR_F_INFO_NEW_CUBE_POS, RSR_F_INFO_SITE_POS, RSR_F_INFO_CUBE_POS,
  RSR_F_INFO_LAST_ACTION,  /* test/airbot.py:159,184 (held wrist target); unused by cube_env */
  /* T-shape env info (T_shape_env.py:127-134); present but unused for the cube envs */
  RSR_F_INFO_TARGET_BASE_POS, RSR_F_INFO_TARGET_VERTICAL_POS, RSR_F_INFO_TARGET_W, RSR_F_INFO_NEW_T_POS, RSR_F_INFO_T_POS,
  RSR_F_INFO_XITA,
  RSR_F_INFO_GO2,          /* Go2 joystick info block, 144 floats (joystick.py:175-196); layout in rsr_mjx_amd/envs/go2.py */
  RSR_F_INFO_STEPS, RSR_F_INFO_TRUNCATION, RSR_F_INFO_EPISODE_DONE, RSR_F_INFO_EPISODE_METRICS,
  RSR_F_FIRST_QPOS, RSR_F_FIRST_QVEL, RSR_F_FIRST_CTRL, RSR_F_FIRST_WARMSTART, RSR_F_FIRST_TIME,
  RSR_F_FIRST_XPOS, RSR_F_FIRST_SITE_XPOS, RSR_F_FIRST_OBS,
  RSR_F_PRIVILEGED_OBS, RSR_F_FIRST_PRIVILEGED_OBS,  /* Go2 obs['privileged_state'], 123 floats (joystick.py:341-366) */
  RSR_F_STATS,             /* int32[4]: solver iters, line-search iters, active contacts, dropped contacts (-1: a hand-off
                            * wait of this env timed out in the last step, see rsr_batch_check) */
  RSR_F_COUNT
};

/* Model + env configuration from an RSRM blob (host memory; copied). */
int rsr_model_create(const void* blob, size_t nbytes, rsr_model** out);
int rsr_model_dims(const rsr_model* m, rsr_dims* out);
void rsr_model_destroy(rsr_model* m);

/* A batch of num_envs env records on HIP device `hip_device`.
 * `state` may be NULL (the library allocates num_envs*rec_floats floats) or a caller-owned device
 * buffer of that size (e.g. a torch tensor) which must outlive the batch. */
int rsr_batch_create(const rsr_model* m, int num_envs, int hip_device, float* state, rsr_batch** out);
void rsr_batch_destroy(rsr_batch* b);

/* Per-env model overrides, device pointers [num_envs, ngeom*3], [num_envs, nbody], [num_envs, nv],
 * [num_envs, nv]; any may be NULL (= the model's value for every env).  Pointers are borrowed. */
int rsr_batch_set_dr(rsr_batch* b, const float* geom_friction, const float* body_mass,
                     const float* dof_damping, const float* dof_frictionloss);

/* One per-env model leaf at a time, covering the Go2 randomisation too (reference
 * ppo_train/go2_training/mujoco_playground/_src/locomotion/go2/randomize.py:6-109: geom_friction, body_ipos, body_mass, qpos0,
 * dof_frictionloss, dof_armature, actuator_gainprm, actuator_biasprm, dof_damping).  Device pointer [num_envs, width] with
 * width = ngeom*3, nbody, nv, nv, nbody*3, nq, nv, nu*3, nu*3 in enum order; NULL restores the model's value.  The last
 * five are built into the Go2 kernels only (RSR_ERR_UNSUPPORTED elsewhere).  Pointers are borrowed. */
enum rsr_dr_field {
  RSR_DR_GEOM_FRICTION = 0, RSR_DR_BODY_MASS, RSR_DR_DOF_DAMPING, RSR_DR_DOF_FRICTIONLOSS,
  RSR_DR_BODY_IPOS, RSR_DR_QPOS0, RSR_DR_DOF_ARMATURE, RSR_DR_ACTUATOR_GAINPRM, RSR_DR_ACTUATOR_BIASPRM,
  RSR_DR_COUNT
};
int rsr_batch_set_dr_field(rsr_batch* b, int dr_field, const float* dev_values);

/* keys: device uint32 [num_envs, 2] (jax.random key data).  Writes every record field. */
int rsr_reset(rsr_batch* b, const uint32_t* keys, void* hip_stream);

/* action: device float32 [num_envs, nu].  One fused launch: AutoReset pre-step, env prologue,
 * n_frames physics substeps, epilogue (reward/done/obs/metrics/info), Episode + AutoReset post-step. */
int rsr_step(rsr_batch* b, const float* action, void* hip_stream);

/* Zero-copy strided view of one record field: element (e, i) is at dev_ptr[e*stride[0] + i*stride[1]]
 * (strides in elements of 4 bytes); shape[0] = num_envs, shape[1] = field width. */
int rsr_view(rsr_batch* b, int field_id, void** dev_ptr, int64_t shape[2], int64_t stride[2]);

/* Scheduling knob of rsr_step, results are bit-identical for every value and for any sequence of values between steps
 * (tests/test_parity_gpu.py::test_schedule_changes_are_bit_identical): the launch runs persistent waves that draw
 * (env, phase) work units from a queue, an env-step being cut into `units` groups of consecutive physics substeps
 * (1 <= units <= min(n_frames, 64); larger values are clamped).  Shorter units shorten the drain at the end of a launch.
 * No counterpart in the reference (XLA schedules its own kernels).  The Go2 kernels ignore it. */
int rsr_batch_set_schedule(rsr_batch* b, int units);

/* Second scheduling knob, results bit-identical for every value as well: the first `whole_envs` envs of the batch are stepped as
 * ONE work unit each whatever `units` says (no hand-off through memory, one ticket), only the rest is cut into `units` phases.
 * Long units first, short units last: the launch still drains in short units while most envs skip the per-unit overhead.
 * -1 (the default): all but twice as many envs as there are resident waves (a phase of a split env then finds the previous
 * one finished two rounds ago instead of waiting for it), and every env of a batch that fits the resident waves (2048 on MI355X);
 * 0: every env is split (the behaviour before round 3). */
int rsr_batch_set_whole_envs(rsr_batch* b, int whole_envs);

/* Third scheduling knob, for the step kernels that run as a plain launch of one wavefront per env (Go2 joystick, Go2 handstand):
 * the hardware priority (s_setprio) a wave asks for at the top of each physics substep.  Timing only: results are bit-identical
 * for every value (tests/test_go2.py::test_wave_priority_policies_are_bit_identical).
 *   0  off: every wave at priority 0; the SIMD then serves its oldest wave first and its waves finish one after another;
 *   1  rotate: (wave slot + substep) mod 4 -- waves that start together take turns and finish together;
 *   2  catch up: least progress first, and the waves of the launch's final resident set keep a floor that grows with their
 *      start time, so that the set finishes together instead of draining one wave per SIMD at a time;
 *  -1  (default) joystick kernels: 1 for a batch of at most one resident round of envs (4096 on MI355X), 0 up to 13/8 of a round
 *      (few late starters: measured), 2 above; handstand: 1.
 * No counterpart in the reference (XLA schedules its own kernels).  The Airbot kernels (work queue) ignore it. */
int rsr_batch_set_priority(rsr_batch* b, int policy);

/* action_repeat of the training wrappers (brax EpisodeWrapper(env, episode_length, action_repeat), applied at RSR/train.py:224-229 and
 * _src/wrapper.py:41-74): one rsr_step = `repeat` env steps with the same action, reward = the sum of their rewards, steps and the
 * episode length advance by `repeat`, done / truncation / the episode metrics are formed once from the last repeat's state, AutoReset
 * acts after that.  1 (the default, and the only value the reference passes: RSR/train.py:81): the wrappers fused in the step kernels;
 * 2 ... 64: the step kernels run as the plain env.step and three small kernels carry the wrappers around the repeats. */
int rsr_batch_set_action_repeat(rsr_batch* b, int repeat);

/* Health of the work queue.  A phase of an env waits for the previous phase's hand-off with a bounded spin; a wait that
 * times out is counted on the device (sticky for the batch's lifetime), the env's later phases inherit the mark, and the
 * env's stats[3] reads -1 after that step.  rsr_batch_check synchronises `hip_stream`, copies the count to
 * *handoff_timeouts (may be NULL) and returns RSR_ERR_HANDOFF if it is not zero.  No counterpart in the reference. */
int rsr_batch_check(rsr_batch* b, void* hip_stream, int* handoff_timeouts);

/* Test hook for the above: `spin_cap` polls before a wait gives up (<= 0: the default, about a second), and phase 0 of env
 * `withhold_env` never publishes its hand-off flag (-1: none), so that env's next phase times out. */
int rsr_batch_set_fault_injection(rsr_batch* b, int spin_cap, int withhold_env);

/* End-of-rollout metric reduction (the reduction in front of the path's one collective: the reference's training loops average
 * the episode metrics of a rollout over the env batch, RSR/train.py:441-447 via acting.Evaluator): one launch writes
 * dev_out[4] = { num_envs, sum of reward, sum of done, mean over envs of the running episode's summed reward }, summed in a
 * fixed order.  dev_out: device float[4]. */
int rsr_rollout_metrics(rsr_batch* b, float* dev_out, void* hip_stream);

/* Optional per-stage dump for parity debugging: device float buffer [num_envs, RSR_DEBUG_FLOATS]
 * filled by the next rsr_step/rsr_reset from the LAST physics forward pass (NULL disables). */
#define RSR_DEBUG_FLOATS 8192
int rsr_batch_set_debug(rsr_batch* b, float* dev_buffer);

/* Average kernel time in ms of the last `rsr_step` launches recorded between rsr_timing_begin/end
 * with HIP events on the stream the kernels were launched on. */
int rsr_timing_begin(rsr_batch* b, void* hip_stream);
int rsr_timing_end(rsr_batch* b, void* hip_stream, float* total_ms, int* launches);

const char* rsr_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RSR_MJX_H_ */
