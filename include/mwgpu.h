/*
 * mwgpu.h -- C ABI of libmwgpu.so, the MI355X-native batched Meta-World runtime.
 *
 * The reference has no FFI for this path: its hot path sits behind the Gymnasium
 * VectorEnv object returned by `metaworld.make_mt_envs` (metaworld/__init__.py:460-513).
 * The entry points below are what a binding for that object needs; each cites the
 * reference interface it replaces.  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions: every function returns 0 on success or a negative code; the message is
 * available through mw_last_error().  One context per GPU, single caller thread, not
 * re-entrant.  All host buffers are owned by the caller; device memory by the library.
 */
#ifndef MWGPU_H
#define MWGPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mw_model mw_model;   /* one compiled MJCF scene (tables built by metaworld_amd/mjcf.py) */
typedef struct mw_ctx mw_ctx;       /* device state of all environments on one GPU */

#define MW_NPROBE 16

typedef struct mw_config {
    int32_t precision;             /* 0 = fp32 state/solver (throughput), 1 = fp64 (parity) */
    int32_t device_id;             /* HIP device ordinal */
    int32_t rank, world_size;      /* data-parallel shard of the env batch (one process per GPU) */
    int32_t max_episode_steps;     /* gymnasium TimeLimit, metaworld/__init__.py:430 */
    int32_t terminate_on_success;  /* AutoTerminateOnSuccessWrapper, metaworld/wrappers.py:207-230 */
    int32_t one_hot;               /* OneHotWrapper, metaworld/wrappers.py:14-32 */
    int32_t num_tasks;             /* one-hot width */
    int32_t full_forward;          /* 0 (product): the final mj_forward of a step (sawyer_xyz_env.py:620) stops after the kinematics unless the
                                    * task's reward reads contact forces (touching_object, :401-440) -- same observations, rewards and state;
                                    * 1: always complete, for callers that read ncon / nefc / efc_force after a step (engine-level tests) */
    int32_t reward_version;        /* SawyerXYZEnv(reward_function_version=...), every sawyer_*_v3.py compute_reward: 1 = the "v1" branches,
                                    * 0 / 2 = "v2" (the reference's default) */
} mw_config;

/* Per-task constants: what the reference keeps in each SawyerXYZEnv subclass
 * (metaworld/envs/sawyer_*_v3.py __init__ + _get_pos_objects/_get_quat_objects). */
typedef struct mw_task {
    int32_t kind;                  /* MT50 id (metaworld/env_dict.py:217-270) selecting reward/reset code */
    int32_t model;                 /* index returned by mw_add_model */
    int32_t onehot_id;             /* position in the benchmark (enumerate order, metaworld/__init__.py:504-506) */
    int32_t probe[MW_NPROBE];      /* frames read by obs/reward, see metaworld_amd/csrc/mw_tasks.hpp */
    int32_t nobj, quat_mode[2];
    int32_t qadr[4], dadr[4], geom[4], reloc[2];
    int32_t partially_observable;  /* SawyerXYZEnv._partially_observable, sawyer_xyz_env.py:217 */
    int32_t max_path_length;       /* SawyerXYZEnv.max_path_length = 500, sawyer_xyz_env.py:153 */
    double hand_init[3], mocap_low[3], mocap_high[3], goal_low[3], goal_high[3];
    double obj_off[2][3];
    double c[15];
} mw_task;

/* ---- model tables (replaces mujoco.MjModel.from_xml_path; reference call: gymnasium MujocoEnv.__init__) ----
 * Fields: the int / real arrays listed in metaworld_amd/pack.py (INT_FIELDS, REAL_FIELDS; names follow mjModel) incl. the
 * support cells of the mesh hulls (mesh_celladr, mesh_cellid: per cube-map cell of directions the ascending list of the vertices
 * that can be the support vertex, metaworld_amd/hullcells.py) from which the runtime derives its one-round-trip support tables.
 * Options: timestep, tolerance (solver tolerance of the context's precision), reset_tolerance (tolerance of the
 * double-precision reset-snapshot build; default = tolerance), meaninertia, gravity_z, iterations, ls_iterations,
 * maxcon, maxefc (contact / constraint-row capacities per environment), nreloc, lanes_per_block (environments per
 * 64-thread workgroup of this model's group, a power of two; 0 = the runtime's choice), step_ms_lpb4 / step_ms_lpb8 (measured
 * time of one late-episode step of this scene at 4 / 8 environments per workgroup: ranks the groups when the runtime chooses). */
mw_model* mw_model_new(void);
int mw_model_set_int(mw_model* m, const char* field, const int32_t* v, int n);
int mw_model_set_real(mw_model* m, const char* field, const double* v, int n);
int mw_model_set_option(mw_model* m, const char* name, double value);
void mw_model_free(mw_model* m);

/* page-locked host memory for the caller's step / reset buffers (optional: any host memory works): into pinned buffers the
 * outputs of mw_step travel as asynchronous DMA copies queued behind the kernel and drained by one synchronisation; into pageable
 * memory every copy is staged and waited for on its own.  No reference counterpart (numpy arrays of a SyncVectorEnv). */
void* mw_alloc_host(size_t bytes);   /* NULL on failure */
void mw_free_host(void* p);

/* ---- context (replaces make_mt_envs / SyncVectorEnv construction, metaworld/__init__.py:460-513) ---- */
int mw_create(const mw_config* cfg, mw_ctx** out);
int mw_add_model(mw_ctx* c, const mw_model* m);                    /* returns model index >= 0 */
int mw_add_task(mw_ctx* c, const mw_task* t, const double* goals /*[ngoals][6] rand_vec*/, int ngoals); /* task index */
int mw_set_envs(mw_ctx* c, const int32_t* env_task /*[n] task index per env*/, int n);
int mw_finalize(mw_ctx* c);   /* allocates device state; runs the faithful reset once per (task, goal) -> snapshots */
/* AutoTerminateOnSuccessWrapper.toggle_terminate_on_success (metaworld/wrappers.py:222-223): takes effect at the next mw_step */
int mw_set_terminate_on_success(mw_ctx* c, int on);
void mw_destroy(mw_ctx* c);
const char* mw_last_error(const mw_ctx* c);
int mw_num_envs(const mw_ctx* c);
int mw_obs_dim(const mw_ctx* c);

/* ---- VectorEnv.reset (SawyerXYZEnv.reset, sawyer_xyz_env.py:664-682; RandomTaskSelectWrapper.reset,
 *      wrappers.py:116-119: goal_idx[i] is the task index drawn by the caller's per-env generator) ---- */
int mw_reset(mw_ctx* c, const uint8_t* mask /*[N] or NULL = all*/, const int32_t* goal_idx /*[N]*/, double* obs_out /*[N][D]*/);

/* ---- VectorEnv.step with SAME_STEP autoreset (SawyerXYZEnv.step, sawyer_xyz_env.py:579-642 + wrapper stack
 *      metaworld/__init__.py:430-454).  next_goal[i] = goal index env i uses if it finishes in this step. ---- */
int mw_step(mw_ctx* c, const float* actions /*[N][4]*/, const int32_t* next_goal /*[N] or NULL*/, double* obs /*[N][D]*/,
            double* reward /*[N]*/, uint8_t* terminated, uint8_t* truncated, uint8_t* success /*[N] each*/,
            float* info /*[N][6] near_object, grasp_success, grasp_reward, in_place_reward, obj_to_target, unscaled_reward; or NULL*/,
            double* final_obs /*[N][D] or NULL*/, double* episode_return /*[N] or NULL*/, int32_t* episode_length /*[N] or NULL*/);

/* gymnasium TimeLimit._elapsed_steps and SawyerXYZEnv.curr_path_length (sawyer_xyz_env.py:596) of every env := elapsed[i]
 * (0 <= elapsed[i] < max_episode_steps): env i truncates and auto-resets after max_episode_steps - elapsed[i] more steps.
 * No reference counterpart (its sub-envs all start at 0 and truncate together); used to stagger the episode phases of a batch. */
int mw_set_episode_phase(mw_ctx* c, const int32_t* elapsed /*[N]*/);

/* ---- throughput path: actions resident in HBM, outputs left in HBM (bench.py) ---- */
int mw_upload_actions(mw_ctx* c, const float* actions /*[nsteps][N][4]*/, int nsteps);
int mw_step_resident(mw_ctx* c, int nsteps, int action_steps, float* kernel_ms /*HIP-event time of the nsteps launches*/);
/* the same loop with the per-step cross-rank bookkeeping gather inside it (mw_comm_init first when world_size > 1): launch k+1
 * overlaps the all-gather of step k on the side stream; returns after both streams have drained */
int mw_step_resident_gather(mw_ctx* c, int nsteps, int action_steps, float* kernel_ms);
/* the same open loop with `steps_per_launch` consecutive steps of every environment inside ONE launch (no reference counterpart:
 * the reference's SyncVectorEnv returns after every step).  Environments are independent, so the final state and outputs equal those
 * of mw_step_resident bit for bit; the batch is synchronised once per launch instead of once per step, which removes the wait for
 * each step's slowest environment.  For rollouts nobody observes between steps (pre-uploaded actions); the output buffers and the
 * bookkeeping record hold the last step. */
int mw_step_resident_fused(mw_ctx* c, int nsteps, int action_steps, int steps_per_launch, float* kernel_ms);
/* RandomTaskSelectWrapper.reset (metaworld/wrappers.py:116-119: every reset of a sub-env draws a new task) INSIDE the resident
 * loop: once a schedule is set, the k-th auto-reset of env i that happens in mw_step_resident / mw_step_resident_gather takes goal
 * goal_schedule[min(k, K-1)][i] (k counts from 0 since this call) instead of the look-ahead goal of the last mw_step / mw_reset.
 * The host draws the rows from the sub-envs' task-selection streams (metaworld_amd/vector_env.py step_resident) and, afterwards,
 * reads how many rows every env consumed with mw_goal_schedule_pos to advance those streams.  NULL / K = 0 clears the schedule. */
int mw_set_goal_schedule(mw_ctx* c, const int32_t* goal_schedule /*[K][N] or NULL*/, int K);
int mw_goal_schedule_pos(mw_ctx* c, int32_t* consumed /*[N] out: auto-resets of env i since mw_set_goal_schedule*/);

/* ---- device-resident boundary (SURVEY.md 8b "outputs_on_device"): the learner's policy runs on the same GPU, so actions and
 *      outputs never visit the host.  Every pointer is a DEVICE pointer the caller owns (a torch tensor's data_ptr()); a NULL
 *      output keeps the context's own buffer.  The call launches on the context's stream and returns after that stream has
 *      drained, so the caller only has to make sure its own writes to `actions` have completed (torch: synchronize the current
 *      stream first).  Semantics are those of mw_step / mw_reset. ---- */
typedef struct mw_device_out {
    double* obs;              /* [N][D] */
    double* reward;           /* [N] */
    uint8_t* flags;           /* [4][N]: terminated, truncated, success, done */
    float* info;              /* [N][6] */
    double* final_obs;        /* [N][D], rows of finished envs only */
    double* episode_return;   /* [N], written for finished envs only */
    int32_t* episode_length;  /* [N], written for finished envs only */
} mw_device_out;
int mw_step_device(mw_ctx* c, const float* actions /*device [N][4]*/, const int32_t* next_goal /*device [N] or NULL = last uploaded*/,
                   const mw_device_out* out /*or NULL*/);
/* the same step ordered against the CALLER's stream (a hipStream_t, e.g. torch.cuda.current_stream().cuda_stream; NULL = the null
 * stream) instead of against the host: the context's stream waits for what the caller has queued so far (its writes to `actions` /
 * `next_goal`), the step is launched, and the caller's stream is made to wait for it -- the call returns at once and whatever the
 * caller queues next on its stream sees the outputs.  The `done` row is also copied to pinned host memory behind the kernel:
 * mw_wait_done blocks until that copy has landed and returns it (valid until the next step), so that the host-side task selection
 * of a finished env (RandomTaskSelectWrapper.reset, wrappers.py:116-119) costs one event wait, not a device-to-host tensor copy. */
int mw_step_device_on(mw_ctx* c, const float* actions /*device [N][4]*/, const int32_t* next_goal /*device [N] or NULL*/,
                      const mw_device_out* out /*or NULL*/, void* caller_stream);
int mw_wait_done(mw_ctx* c, const uint8_t** done_host /*out: pinned host [N], the `done` flags of the last mw_step_device_on*/);
int mw_reset_device(mw_ctx* c, const uint8_t* mask /*device [N] or NULL = all*/, const int32_t* goal_idx /*device [N]*/,
                    double* obs_out /*device [N][D] or NULL*/);

/* ---- scripted policies on the device (metaworld/policies/sawyer_*_v3_policy.py, SawyerXYZPolicy.get_action `policy.py:34-53`;
 *      device code generated from metaworld_amd/policies.py by tools/gen_device_policies.py).  policy_id[i] = index of env i's
 *      task in ALL_V3_ENVIRONMENTS order (`env_dict.py:217-270`); observations must show the goal.
 *      mw_policy_actions: one policy evaluation, host obs in -> host actions out (float32, clipped to [-1, 1]).
 *      mw_policy_rollout: the closed loop of metaworld/evaluation.py:48-103 with the scripted policy as the agent, entirely on
 *      the device: reset every env to goal_schedule[0][i], then nsteps x (policy kernel, step kernel); the k-th auto-reset of
 *      env i takes goal_schedule[min(k, K-1)][i].  Outputs per env: finished episodes, and those in which success was ever 1. ---- */
int mw_policy_actions(mw_ctx* c, const int32_t* policy_id /*[N]*/, const double* obs /*[N][D]*/, float* actions /*[N][4] out*/);
int mw_policy_rollout(mw_ctx* c, const int32_t* policy_id /*[N]*/, const int32_t* goal_schedule /*[K][N]*/, int K, int nsteps,
                      int32_t* episodes /*[N] out or NULL*/, int32_t* successes /*[N] out or NULL*/, float* kernel_ms /*or NULL*/);

/* the same closed loop with `steps_per_launch` (policy, step) pairs of every environment per kernel launch: an environment's policy is
 * evaluated by its own thread between two of its steps -- same episodes and successes, no batch-wide synchronisation inside a launch */
int mw_policy_rollout_fused(mw_ctx* c, const int32_t* policy_id /*[N]*/, const int32_t* goal_schedule /*[K][N]*/, int K, int nsteps,
                            int steps_per_launch, int32_t* episodes /*[N] out or NULL*/, int32_t* successes /*[N] out or NULL*/, float* kernel_ms /*or NULL*/);

/* ---- cross-rank bookkeeping gather (no reference counterpart: the reference's SyncVectorEnv lives in one
 *      process).  Envs are independent, so stepping needs no communication; the one exchange is this 12-byte record per
 *      env and step, all-gathered over RCCL (xGMI) so that every rank sees the global done / success / task-id / episode
 *      statistics that `metaworld/evaluation.py:79-82` reads from `final_info`.  The step kernel writes the record; the
 *      collective runs on a side stream and overlaps the next step's kernel. ---- */
typedef struct mw_bookkeeping {
    uint8_t done;              /* terminated | truncated */
    uint8_t success;           /* info["success"] of this step */
    int16_t task_id;           /* index in ALL_V3_ENVIRONMENTS (metaworld/env_dict.py:217-270) */
    float episode_return;      /* RecordEpisodeStatistics running return (the episode's total where done) */
    int32_t episode_length;    /*   "" length */
} mw_bookkeeping;
int mw_comm_unique_id(uint8_t* id_out /*[128] ncclUniqueId, created on the calling rank*/);
int mw_comm_init(mw_ctx* c, const uint8_t* id /*[128], the same bytes on every rank*/, int rank, int world_size);
/* what the collective library reports for the context's communicator: info[0] = ranks in it (ncclCommCount), info[1] = this rank
 * (ncclCommUserRank), info[2] = 1 if a RCCL communicator exists (0: world size 1, the "gather" is a device copy), info[3] = HIP
 * device of the context.  bench.py prints it in `config`, so that a multi-GPU run shows that RCCL saw N ranks. */
int mw_comm_info(mw_ctx* c, int32_t* info /*[4]*/);
/* all-gather the records of the LAST step; out = [world_size][N] records, a HOST pointer (out_on_device = 0) or a DEVICE
 * pointer (1); NULL keeps the result in the context's own device buffer.  world_size 1 needs no communicator. */
int mw_gather_bookkeeping(mw_ctx* c, mw_bookkeeping* out, int out_on_device);

/* ---- run-time status (SURVEY.md 5 "failure detection"): status[0] = OR of the per-env flags since the last clear
 *      (1 = constraint-row capacity exceeded, 2 = contact capacity exceeded -- rows / contacts were DROPPED, the step differs
 *      from the reference's; 4 = non-finite state, the env was reset: the intent of sawyer_xyz_env.py:603-619),
 *      8 = CANARY: the copies of a redundantly computed value held by the threads that share one environment disagreed -- the
 *      step kernel's own check against silent register corruption; must always be 0),
 *      status[1..4] = number of env-steps that raised flag 1 / 2 / 4 / 8,
 *      status[5] = line searches abandoned because the Newton direction was not a descent direction (informational, no flag
 *      bit: MuJoCo's solver stops the same way; frequent in single precision at the optimum, 0 in fp64 on the bench workload),
 *      status[6..7] reserved. ---- */
#define MW_STATUS_WORDS 8
int mw_status(mw_ctx* c, int32_t* status /*[n]*/, int n /*words the caller has room for: the first min(n, MW_STATUS_WORDS) are written, the rest zeroed*/, int clear);

/* HIP-event time of every launch (= step) of the LAST mw_step_resident / mw_step_resident_gather call, in ms, oldest first; returns
 * the number written (<= cap; at most 8192 are recorded per call).  bench.py reports min / median / max beside the mean. */
int mw_launch_times(mw_ctx* c, float* ms_out /*[cap]*/, int cap);

/* run-time options of a finalized context (no reference counterpart): "split_collision" = 1: every dynamics evaluation runs its
 * narrow phase as batch-wide kernels over (environment, candidate pair) work items between the lane kernels (mid phase ->
 * type-sorted work lists -> narrow phase at several waves per SIMD) instead of inside the one fused step kernel; same contacts in the
 * same order, same results (tests/test_split_collision.py).  0 (default) = the fused kernel.  The split measured 25-37 % slower, so
 * since round 6 its kernels exist only in libraries built with -DMW_SPLIT_COLLISION (libmwgpu_split.so, __graft_entry__.build_gpu_split);
 * the default library accepts 0 and answers 1 with an error (mw_last_error). */
int mw_set_option(mw_ctx* c, const char* name, double value);

/* ---- state access for parity tests (mujoco data.qpos / qvel / mocap_pos, MujocoEnv.set_state) ---- */
int mw_column_size(mw_ctx* c, int env, const char* what);
int mw_read(mw_ctx* c, int env, const char* what, double* out, int n);
int mw_write(mw_ctx* c, int env, const char* what, const double* in, int n);
/* the persistent state of EVERY environment in one call (checkpoints: MujocoEnv.set_state / data.qpos, qvel + the episode block):
 * row i = the mw_column_size(i, "state") reals that mw_read(i, "state") returns, rows `stride` doubles apart */
int mw_get_state(mw_ctx* c, double* out /*[N][stride]*/, int stride);
int mw_set_state(mw_ctx* c, const double* in /*[N][stride]*/, int stride);
int mw_read_int(mw_ctx* c, int env, const char* what, int32_t* out, int n);
int mw_debug(mw_ctx* c, int what /*0 forward, 1 n substeps, 2 resetData, 3 kinematics*/, int n);

#ifdef __cplusplus
}
#endif
#endif
