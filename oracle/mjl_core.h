/*
 * oracle/mjl_core.h -- TEST INFRASTRUCTURE ONLY (CPU oracle, fp64).
 *
 * A plain-C restatement of the slice of the MuJoCo 3.3.0 `mj_step` / `mj_forward`
 * pipeline that the Meta-World reference exercises
 * (reference call sites: metaworld/sawyer_xyz_env.py:595 do_simulation -> mj_step x5,
 *  :620 mj_forward; engine semantics listed in SURVEY.md Appendix C).
 *
 * PARITY UNPINNED: mujoco==3.3.0 (pyproject.toml:28) is a third-party wheel that
 * is not present in /root/reference and cannot be installed here, and the
 * reference's tests hold no golden vectors for this path.
 * The algorithm below is restated from MuJoCo's published documentation
 * (Computation chapter: kinematics, CRB, RNE, soft-constraint model, Newton
 * solver, semi-implicit Euler).  It is pinned only by analytic physics tests
 * and by the reference's own behavioural tests run on top of it.
 *
 * Nothing outside tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * may link or load this code.
 */
#ifndef MJL_CORE_H
#define MJL_CORE_H

#ifdef __cplusplus
extern "C" {
#endif

#define MJL_MAXCON 64
#define MJL_MAXEFC 320
#define MJL_MAXNV 24

enum { MJL_PLANE = 0, MJL_HFIELD, MJL_SPHERE, MJL_CAPSULE, MJL_ELLIPSOID, MJL_CYLINDER, MJL_BOX, MJL_MESH };
enum { MJL_FREE = 0, MJL_BALL, MJL_SLIDE, MJL_HINGE };
enum { MJL_EQUALITY = 0, MJL_LIMIT = 3, MJL_CONTACT_ELLIPTIC = 7 };
enum { MJL_SATISFIED = 0, MJL_QUADRATIC = 1, MJL_CONE = 4 };

typedef struct {
    /* sizes */
    int nq, nv, nbody, njnt, ngeom, nsite, nmesh, nmeshvert, npair, nu, neq;
    /* options */
    double timestep, tolerance, gravity[3], meaninertia;
    int iterations;
    /* bodies */
    int *body_parentid, *body_mocap, *body_dofadr, *body_dofnum, *body_jntadr, *body_jntnum, *body_lastdof, *body_weldid;
    double *body_pos, *body_quat, *body_ipos, *body_iquat, *body_mass, *body_inertia, *body_invweight0;
    /* joints / dofs */
    int *jnt_type, *jnt_bodyid, *jnt_qposadr, *jnt_dofadr, *jnt_limited;
    double *jnt_pos, *jnt_axis, *jnt_range, *jnt_stiffness, *jnt_springref, *jnt_solref, *jnt_solimp, *jnt_margin;
    int *dof_bodyid, *dof_jntid, *dof_parentid;
    double *dof_armature, *dof_damping, *dof_invweight0;
    double *qpos0;
    /* geoms */
    int *geom_type, *geom_bodyid, *geom_meshid, *geom_contype, *geom_conaffinity, *geom_condim, *geom_priority;
    double *geom_size, *geom_pos, *geom_quat, *geom_friction, *geom_solref, *geom_solimp, *geom_solmix,
        *geom_margin, *geom_gap, *geom_rbound;
    int *mesh_vertadr, *mesh_vertnum;
    double *mesh_vert;
    /* OPTIONAL, timing only (bench.py's cpu_baseline; never set by tests or by the parity chain): the support cells of the hulls
       (metaworld_amd/hullcells.py).  When present, support() takes the maximum over the direction cell's list instead of over all
       vertices -- the same answer by construction of the lists -- so that the CPU stand-in is timed with the same acceleration
       the product uses instead of a 324- / 884-vertex scan per support call. */
    int *mesh_celladr, *mesh_cellid;
    int *pair_geom;
    /* sites */
    int *site_bodyid;
    double *site_pos, *site_quat;
    /* actuators, equality */
    int *act_dofid, *act_qposid;
    double *act_kp, *act_ctrlrange;
    int *eq_body1, *eq_body2;
    double *eq_solref, *eq_solimp, *eq_data; /* eq_data: 11 per weld */
} MjlModel;

typedef struct {
    int geom1, geom2, dim, efc_address, exclude;
    double dist, pos[3], frame[9], includemargin, friction[5], solref[2], solimp[5], mu;
} MjlContact;

typedef struct {
    /* state */
    double time, *qpos, *qvel, *qacc_warmstart, *ctrl, mocap_pos[3], mocap_quat[4];
    /* position-dependent */
    double *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis, *geom_xpos, *geom_xmat, *site_xpos, *site_xmat;
    double *cdof;      /* nv x 6: [ang(3), lin(3)] about world origin */
    double *cdof_dot;  /* nv x 6 */
    double *cvel;      /* nbody x 6 */
    double *qM;        /* nv x nv dense */
    double *qL;        /* nv x nv Cholesky of qM */
    /* velocity / force */
    double *qfrc_bias, *qfrc_passive, *qfrc_actuator, *qfrc_smooth, *qacc_smooth, *qfrc_constraint, *qacc;
    /* contacts and constraints */
    int ncon, nefc, ne, nl;
    MjlContact contact[MJL_MAXCON];
    int efc_type[MJL_MAXEFC], efc_id[MJL_MAXEFC], efc_state[MJL_MAXEFC];
    double *efc_J; /* MJL_MAXEFC x nv */
    double efc_pos[MJL_MAXEFC], efc_margin[MJL_MAXEFC], efc_diagApprox[MJL_MAXEFC], efc_R[MJL_MAXEFC],
        efc_D[MJL_MAXEFC], efc_KBIP[MJL_MAXEFC][4], efc_vel[MJL_MAXEFC], efc_aref[MJL_MAXEFC],
        efc_force[MJL_MAXEFC];
    int solver_niter;
    int warning_overflow;
    int max_ncon, max_nefc;   /* high-water marks (sizing the GPU caps) */
} MjlData;

MjlModel* mjl_model_new(void);
void mjl_model_free(MjlModel* m);
/* set a named array (copied); returns 0 ok, -1 unknown name */
int mjl_model_set_int(MjlModel* m, const char* name, const int* v, int n);
int mjl_model_set_real(MjlModel* m, const char* name, const double* v, int n);
int mjl_model_finalize(MjlModel* m);
double* mjl_model_real_ptr(MjlModel* m, const char* name, int* n);

MjlData* mjl_data_new(const MjlModel* m);
void mjl_data_free(MjlData* d);
double* mjl_data_real_ptr(const MjlModel* m, MjlData* d, const char* name, int* n);

void mjl_reset_data(const MjlModel* m, MjlData* d);
void mjl_forward(const MjlModel* m, MjlData* d);
void mjl_step(const MjlModel* m, MjlData* d);
void mjl_step_n(const MjlModel* m, MjlData* d, int n);
/* timing loop of bench.py's cpu_baseline, entirely in C: n_env_steps x (random action in [-1, 1]^4 from a xorshift stream ->
   mocap += 0.01 a clipped to [lo, hi], ctrl = (a3, -a3), 5 x mjl_step, mjl_forward) */
void mjl_bench_env_steps(const MjlModel* m, MjlData* d, int n_env_steps, unsigned long long* rng_state, const double* lo, const double* hi);
/* stages, exposed for unit tests */
void mjl_kinematics(const MjlModel* m, MjlData* d);
void mjl_crb(const MjlModel* m, MjlData* d);
void mjl_rne_bias(const MjlModel* m, MjlData* d);
void mjl_collision(const MjlModel* m, MjlData* d);
void mjl_jac(const MjlModel* m, const MjlData* d, double* jacp, double* jacr, const double point[3], int body);

void mjl_data_info(const MjlData* d, int* out8);
void mjl_data_contact(const MjlData* d, int i, int* iv4, double* rv16);
void mjl_data_efc_int(const MjlData* d, int* type, int* id, int* state);

/* narrow phase entry used by unit tests: returns number of contacts written */
int mjl_collide_pair(const MjlModel* m, const MjlData* d, int g1, int g2, double margin, MjlContact* out, int maxout);

#ifdef __cplusplus
}
#endif
#endif
