// mw_abi.inl -- C ABI implementation shared by the device library (mwgpu.hip, prefix mw_) and
// the CPU test harness (tests/host_harness.cpp, prefix mwh_).  Requires `Backend` and MW_API(name).
#include <cstring>

struct mw_model { mw::ModelData d; };
struct mw_ctx {
    mw::Config cfg{};
    std::unique_ptr<mw::ContextBase> impl;
    std::vector<std::shared_ptr<mw::ModelData>> models;
    std::vector<mw::TaskSpec> tasks;
    std::vector<int> env_task;
    std::string error;
};

#define MW_TRY(ctx, body)                                                   \
    try { body; return 0; }                                                 \
    catch (const std::exception& ex) { if (ctx) (ctx)->error = ex.what(); return -1; } \
    catch (...) { if (ctx) (ctx)->error = "unknown error"; return -1; }

extern "C" {

mw_model* MW_API(model_new)(void) { return new mw_model(); }
void MW_API(model_free)(mw_model* m) { delete m; }
int MW_API(model_set_int)(mw_model* m, const char* f, const int32_t* v, int n) { m->d.ints[f].assign(v, v + n); return 0; }
int MW_API(model_set_real)(mw_model* m, const char* f, const double* v, int n) { m->d.reals[f].assign(v, v + n); return 0; }
int MW_API(model_set_option)(mw_model* m, const char* name, double v) {
    std::string k = name;
    if (k == "timestep") m->d.timestep = v; else if (k == "tolerance") m->d.tolerance = v;
    else if (k == "reset_tolerance") m->d.reset_tolerance = v;
    else if (k == "lanes_per_block") m->d.lanes_per_block = (int)v;
    else if (k == "step_ms_lpb4") m->d.step_ms_lpb4 = v; else if (k == "step_ms_lpb8") m->d.step_ms_lpb8 = v;
    else if (k == "meaninertia") m->d.meaninertia = v; else if (k == "gravity_z") m->d.gravity[2] = v;
    else if (k == "iterations") m->d.sz.iterations = (int)v; else if (k == "ls_iterations") m->d.sz.ls_iterations = (int)v;
    else if (k == "maxcon") m->d.sz.maxcon = (int)v; else if (k == "maxefc") m->d.sz.maxefc = (int)v;
    else if (k == "nreloc") m->d.sz.nreloc = (int)v;
    else return -1;
    return 0;
}

void* MW_API(alloc_host)(size_t bytes) { try { return Backend::alloc_host(bytes); } catch (...) { return nullptr; } }
void MW_API(free_host)(void* p) { Backend::free_host(p); }
int MW_API(create)(const mw_config* cfg, mw_ctx** out) {
    mw_ctx* c = new mw_ctx();
    c->cfg.precision = cfg->precision; c->cfg.device_id = cfg->device_id; c->cfg.rank = cfg->rank; c->cfg.world_size = cfg->world_size;
    c->cfg.max_episode_steps = cfg->max_episode_steps; c->cfg.terminate_on_success = cfg->terminate_on_success;
    c->cfg.one_hot = cfg->one_hot; c->cfg.num_tasks = cfg->num_tasks; c->cfg.full_forward = cfg->full_forward; c->cfg.reward_version = cfg->reward_version;
    *out = c;
    MW_TRY(c, Backend::init(cfg->device_id));
}
int MW_API(add_model)(mw_ctx* c, const mw_model* m) {
    try {
        auto d = std::make_shared<mw::ModelData>(m->d);
        d->finalize();
        if (d->sz.maxcon <= 0 || d->sz.maxefc <= 0 || d->sz.iterations <= 0) throw std::runtime_error("model options maxcon/maxefc/iterations not set");
        if (d->sz.ls_iterations <= 0) d->sz.ls_iterations = 50;
        c->models.push_back(d);
        return (int)c->models.size() - 1;
    } catch (const std::exception& ex) { c->error = ex.what(); return -1; }
}
int MW_API(add_task)(mw_ctx* c, const mw_task* t, const double* goals, int ngoals) {
    try {
        if (t->model < 0 || t->model >= (int)c->models.size()) throw std::runtime_error("bad model index");
        mw::TaskSpec s{};
        s.kind = t->kind; s.model = t->model; s.nobj = t->nobj; s.partially_observable = t->partially_observable; s.max_path_length = t->max_path_length;
        for (int k = 0; k < mw::P_COUNT; k++) s.probe[k] = t->probe[k];
        for (int k = 0; k < 2; k++) { s.quat_mode[k] = t->quat_mode[k]; s.reloc[k] = t->reloc[k]; for (int j = 0; j < 3; j++) s.obj_off[k][j] = t->obj_off[k][j]; }
        for (int k = 0; k < 4; k++) { s.qadr[k] = t->qadr[k]; s.dadr[k] = t->dadr[k]; s.geom[k] = t->geom[k]; }
        for (int k = 0; k < 3; k++) { s.hand_init[k] = t->hand_init[k]; s.mocap_low[k] = t->mocap_low[k]; s.mocap_high[k] = t->mocap_high[k]; s.goal_low[k] = t->goal_low[k]; s.goal_high[k] = t->goal_high[k]; }
        for (int k = 0; k < 15; k++) s.c[k] = t->c[k];
        s.c[15] = t->onehot_id;
        if (!goals || ngoals <= 0) throw std::runtime_error("a task needs at least one goal (rand_vec)");
        s.goals.assign(goals, goals + 6 * (size_t)ngoals);
        c->tasks.push_back(s);
        return (int)c->tasks.size() - 1;
    } catch (const std::exception& ex) { c->error = ex.what(); return -1; }
}
int MW_API(set_envs)(mw_ctx* c, const int32_t* env_task, int n) {
    if (!env_task || n <= 0) { c->error = "set_envs: empty env list"; return -1; }
    c->env_task.assign(env_task, env_task + n); return 0;
}
int MW_API(set_terminate_on_success)(mw_ctx* c, int on) {
    c->cfg.terminate_on_success = on ? 1 : 0;
    if (c->impl) c->impl->cfg.terminate_on_success = c->cfg.terminate_on_success;
    return 0;
}
int MW_API(finalize)(mw_ctx* c) {
    MW_TRY(c, {
        for (int t : c->env_task) if (t < 0 || t >= (int)c->tasks.size()) throw std::runtime_error("env refers to unknown task");
        Backend::use(c->cfg.device_id);
        if (c->cfg.precision == 1) c->impl.reset(new mw::Context<double, Backend>());
        else c->impl.reset(new mw::Context<float, Backend>());
        c->impl->cfg = c->cfg; c->impl->models = c->models; c->impl->tasks = c->tasks; c->impl->env_task = c->env_task;
        c->impl->finalize();
    });
}
void MW_API(destroy)(mw_ctx* c) { if (c) { try { Backend::use(c->cfg.device_id); } catch (...) {} delete c; } }
const char* MW_API(last_error)(const mw_ctx* c) { return c ? c->error.c_str() : "null context"; }
int MW_API(num_envs)(const mw_ctx* c) { return (int)c->env_task.size(); }
int MW_API(obs_dim)(const mw_ctx* c) { return 39 + (c->cfg.one_hot ? c->cfg.num_tasks : 0); }

#define MW_NEED_IMPL(c) if (!(c)->impl) throw std::runtime_error("context not finalized"); Backend::use((c)->cfg.device_id)
int MW_API(reset)(mw_ctx* c, const uint8_t* mask, const int32_t* goal_idx, double* obs_out) {
    MW_TRY(c, { MW_NEED_IMPL(c); c->impl->reset(mask, goal_idx, obs_out); });
}
int MW_API(step)(mw_ctx* c, const float* a, const int32_t* ng, double* obs, double* rew, uint8_t* te, uint8_t* tr, uint8_t* su,
                 float* info, double* fo, double* er, int32_t* el) {
    MW_TRY(c, { MW_NEED_IMPL(c); c->impl->step(a, ng, obs, rew, te, tr, su, info, fo, er, el); });
}
int MW_API(upload_actions)(mw_ctx* c, const float* a, int nsteps) { MW_TRY(c, { MW_NEED_IMPL(c); c->impl->upload_actions(a, nsteps); }); }
int MW_API(step_resident)(mw_ctx* c, int nsteps, int asteps, float* ms) { MW_TRY(c, { MW_NEED_IMPL(c); c->impl->step_device_only(nullptr, nsteps, asteps, ms); }); }
int MW_API(step_device)(mw_ctx* c, const float* d_act, const int32_t* d_next_goal, const mw_device_out* out) {
    MW_TRY(c, { MW_NEED_IMPL(c); if (!d_act) throw std::invalid_argument("step_device: actions are required"); c->impl->step_device(d_act, d_next_goal, out); });
}
int MW_API(step_device_on)(mw_ctx* c, const float* d_act, const int32_t* d_next_goal, const mw_device_out* out, void* caller_stream) {
    MW_TRY(c, { MW_NEED_IMPL(c); if (!d_act) throw std::invalid_argument("step_device_on: actions are required"); c->impl->step_device(d_act, d_next_goal, out, true, caller_stream); });
}
int MW_API(wait_done)(mw_ctx* c, const uint8_t** done_host) {
    MW_TRY(c, { MW_NEED_IMPL(c); if (!done_host) throw std::invalid_argument("wait_done: null output"); *done_host = c->impl->wait_done(); });
}
int MW_API(reset_device)(mw_ctx* c, const uint8_t* d_mask, const int32_t* d_goal_idx, double* d_obs) {
    MW_TRY(c, { MW_NEED_IMPL(c); if (!d_goal_idx) throw std::invalid_argument("reset_device: goal_idx is required"); c->impl->reset_device(d_mask, d_goal_idx, d_obs); });
}
int MW_API(policy_actions)(mw_ctx* c, const int32_t* policy_id, const double* obs, float* act) {
    MW_TRY(c, { MW_NEED_IMPL(c); if (!policy_id || !obs || !act) throw std::invalid_argument("policy_actions: null argument"); c->impl->policy_actions(policy_id, obs, act); });
}
int MW_API(policy_rollout)(mw_ctx* c, const int32_t* policy_id, const int32_t* schedule, int K, int nsteps, int32_t* episodes, int32_t* successes, float* ms) {
    MW_TRY(c, { MW_NEED_IMPL(c); if (!policy_id || !schedule) throw std::invalid_argument("policy_rollout: null argument"); c->impl->policy_rollout(policy_id, schedule, K, nsteps, episodes, successes, ms); });
}
int MW_API(policy_rollout_fused)(mw_ctx* c, const int32_t* policy_id, const int32_t* schedule, int K, int nsteps, int per_launch, int32_t* episodes, int32_t* successes, float* ms) {
    MW_TRY(c, { MW_NEED_IMPL(c); if (!policy_id || !schedule) throw std::invalid_argument("policy_rollout_fused: null argument"); if (per_launch < 1) throw std::invalid_argument("policy_rollout_fused: steps_per_launch must be >= 1"); c->impl->policy_rollout(policy_id, schedule, K, nsteps, episodes, successes, ms, per_launch); });
}
int MW_API(step_resident_fused)(mw_ctx* c, int nsteps, int asteps, int per_launch, float* ms) { MW_TRY(c, { MW_NEED_IMPL(c); c->impl->step_fused(nsteps, asteps, per_launch, ms); }); }
int MW_API(step_resident_gather)(mw_ctx* c, int nsteps, int asteps, float* ms) { MW_TRY(c, { MW_NEED_IMPL(c); c->impl->step_resident_gather(nsteps, asteps, ms); }); }
int MW_API(comm_unique_id)(uint8_t* id_out) {
    try { if (!id_out) return -1; Backend::comm_unique_id(id_out); return 0; } catch (...) { return -1; }
}
int MW_API(comm_init)(mw_ctx* c, const uint8_t* id, int rank, int world) {
    MW_TRY(c, { MW_NEED_IMPL(c); if (world > 1 && !id) throw std::invalid_argument("comm_init: id is required when world_size > 1"); c->impl->comm_init(id, rank, world); });
}
int MW_API(comm_info)(mw_ctx* c, int32_t* out) { MW_TRY(c, { MW_NEED_IMPL(c); if (!out) throw std::invalid_argument("comm_info: null output"); c->impl->comm_info(out); }); }
int MW_API(gather_bookkeeping)(mw_ctx* c, mw_bookkeeping* out, int out_on_device) { MW_TRY(c, { MW_NEED_IMPL(c); c->impl->gather_bookkeeping(out, out_on_device); }); }
int MW_API(set_episode_phase)(mw_ctx* c, const int32_t* elapsed) { MW_TRY(c, { MW_NEED_IMPL(c); if (!elapsed) throw std::invalid_argument("set_episode_phase: null argument"); c->impl->set_episode_phase(elapsed); }); }
int MW_API(set_goal_schedule)(mw_ctx* c, const int32_t* schedule, int K) { MW_TRY(c, { MW_NEED_IMPL(c); if (K < 0 || (K > 0 && !schedule)) throw std::invalid_argument("set_goal_schedule: K rows need a schedule"); c->impl->set_goal_schedule(schedule, K); }); }
int MW_API(goal_schedule_pos)(mw_ctx* c, int32_t* out) { MW_TRY(c, { MW_NEED_IMPL(c); if (!out) throw std::invalid_argument("goal_schedule_pos: null output"); c->impl->goal_schedule_pos(out); }); }
int MW_API(status)(mw_ctx* c, int32_t* out, int n, int clear) { MW_TRY(c, { MW_NEED_IMPL(c); if (!out || n < 1) throw std::invalid_argument("status: null output"); c->impl->status(out, n, clear); }); }
int MW_API(launch_times)(mw_ctx* c, float* out, int cap) {
    try { MW_NEED_IMPL(c); if (!out || cap < 0) throw std::invalid_argument("launch_times: null output"); return c->impl->launch_times(out, cap); } catch (const std::exception& ex) { c->error = ex.what(); return -1; }
}
int MW_API(set_option)(mw_ctx* c, const char* name, double value) { MW_TRY(c, { MW_NEED_IMPL(c); if (!name) throw std::invalid_argument("set_option: null name"); c->impl->set_option(name, value); }); }
int MW_API(column_size)(mw_ctx* c, int env, const char* what) {
    try { MW_NEED_IMPL(c); return c->impl->layout_size(env, what); } catch (const std::exception& ex) { c->error = ex.what(); return -1; }
}
int MW_API(get_state)(mw_ctx* c, double* out, int stride) { MW_TRY(c, { MW_NEED_IMPL(c); if (!out) throw std::invalid_argument("get_state: null output"); c->impl->state_all(out, stride, false); }); }
int MW_API(set_state)(mw_ctx* c, const double* in, int stride) { MW_TRY(c, { MW_NEED_IMPL(c); if (!in) throw std::invalid_argument("set_state: null input"); c->impl->state_all(const_cast<double*>(in), stride, true); }); }
int MW_API(read)(mw_ctx* c, int env, const char* what, double* out, int n) { MW_TRY(c, { MW_NEED_IMPL(c); c->impl->read_col(env, what, n, out); }); }
int MW_API(write)(mw_ctx* c, int env, const char* what, const double* in, int n) { MW_TRY(c, { MW_NEED_IMPL(c); c->impl->write_col(env, what, n, in); }); }
int MW_API(read_int)(mw_ctx* c, int env, const char* what, int32_t* out, int n) { MW_TRY(c, { MW_NEED_IMPL(c); c->impl->read_icol(env, what, n, out); }); }
int MW_API(debug)(mw_ctx* c, int what, int n) { MW_TRY(c, { MW_NEED_IMPL(c); c->impl->debug(what, n); }); }

}  // extern "C"
