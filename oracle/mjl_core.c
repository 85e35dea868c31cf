/*
 * oracle/mjl_core.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, fp64).  See mjl_core.h.
 *
 * Dense, readable restatement of the mj_forward / mj_step stages used by the
 * Meta-World reference:
 *   kinematics -> CRB mass matrix -> Cholesky -> collision -> constraint rows
 *   (weld, joint limits, elliptic contacts) -> bias (RNE) / passive / actuators
 *   -> Newton solver with exact line search -> semi-implicit Euler.
 * "parity unpinned" (no MuJoCo available): see header.
 */
#include "mjl_core.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MINVAL 1e-15
#define MINIMP 0.0001
#define MAXIMP 0.9999
#define IMPRATIO 1.0

/* ------------------------------------------------------------------ vec math */
static inline void v3set(double* r, double a, double b, double c) { r[0] = a; r[1] = b; r[2] = c; }
static inline void v3copy(double* r, const double* a) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
static inline void v3add(double* r, const double* a, const double* b) { r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; }
static inline void v3sub(double* r, const double* a, const double* b) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
static inline void v3addscl(double* r, const double* a, const double* b, double s) { r[0] = a[0] + s * b[0]; r[1] = a[1] + s * b[1]; r[2] = a[2] + s * b[2]; }
static inline double v3dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void v3cross(double* r, const double* a, const double* b) {
    double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    r[0] = x; r[1] = y; r[2] = z;
}
static inline double v3norm(const double* a) { return sqrt(v3dot(a, a)); }
static inline double v3normalize(double* a) {
    double n = v3norm(a);
    if (n < MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; return 0; }
    a[0] /= n; a[1] /= n; a[2] /= n;
    return n;
}
static void qmul(double* r, const double* a, const double* b) {
    double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static void qnormalize(double* q) {
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
static void q2mat(double* m, const double* q) {
    double w = q[0], x = q[1], y = q[2], z = q[3];
    m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
    m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
    m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
static inline void m3mulv(double* r, const double* m, const double* v) {
    double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
    double y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
    double z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
    r[0] = x; r[1] = y; r[2] = z;
}
static void m3mul(double* r, const double* a, const double* b) {
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
    memcpy(r, t, sizeof t);
}

/* ------------------------------------------------------------ model storage */
typedef struct { const char* name; size_t off; int is_int; } Field;
#define FI(n) { #n, offsetof(MjlModel, n), 1 }
#define FR(n) { #n, offsetof(MjlModel, n), 0 }
#include <stddef.h>
static const Field FIELDS[] = {
    FI(body_parentid), FI(body_mocap), FI(body_dofadr), FI(body_dofnum), FI(body_jntadr), FI(body_jntnum),
    FI(body_lastdof), FI(body_weldid), FR(body_pos), FR(body_quat), FR(body_ipos), FR(body_iquat), FR(body_mass),
    FR(body_inertia), FR(body_invweight0), FI(jnt_type), FI(jnt_bodyid), FI(jnt_qposadr), FI(jnt_dofadr),
    FI(jnt_limited), FR(jnt_pos), FR(jnt_axis), FR(jnt_range), FR(jnt_stiffness), FR(jnt_springref), FR(jnt_solref),
    FR(jnt_solimp), FR(jnt_margin), FI(dof_bodyid), FI(dof_jntid), FI(dof_parentid), FR(dof_armature),
    FR(dof_damping), FR(dof_invweight0), FR(qpos0), FI(geom_type), FI(geom_bodyid), FI(geom_meshid),
    FI(geom_contype), FI(geom_conaffinity), FI(geom_condim), FI(geom_priority), FR(geom_size), FR(geom_pos),
    FR(geom_quat), FR(geom_friction), FR(geom_solref), FR(geom_solimp), FR(geom_solmix), FR(geom_margin),
    FR(geom_gap), FR(geom_rbound), FI(mesh_vertadr), FI(mesh_vertnum), FR(mesh_vert), FI(mesh_celladr), FI(mesh_cellid), FI(pair_geom),
    FI(site_bodyid), FR(site_pos), FR(site_quat), FI(act_dofid), FI(act_qposid), FR(act_kp), FR(act_ctrlrange),
    FI(eq_body1), FI(eq_body2), FR(eq_solref), FR(eq_solimp), FR(eq_data),
};
#define NFIELDS ((int)(sizeof(FIELDS) / sizeof(FIELDS[0])))
typedef struct { MjlModel m; int count[sizeof(FIELDS) / sizeof(FIELDS[0])]; } ModelBox;

MjlModel* mjl_model_new(void) {
    ModelBox* b = (ModelBox*)calloc(1, sizeof(ModelBox));
    b->m.timestep = 0.002; b->m.tolerance = 1e-8; b->m.iterations = 100;
    b->m.gravity[2] = -9.81; b->m.meaninertia = 1;
    return &b->m;
}
void mjl_model_free(MjlModel* m) {
    if (!m) return;
    for (int i = 0; i < NFIELDS; i++) free(*(void**)((char*)m + FIELDS[i].off));
    free(m);
}
static int find_field(const char* name) {
    for (int i = 0; i < NFIELDS; i++) if (!strcmp(name, FIELDS[i].name)) return i;
    return -1;
}
int mjl_model_set_int(MjlModel* m, const char* name, const int* v, int n) {
    if (!strcmp(name, "opt_iterations")) { m->iterations = v[0]; return 0; }
    int f = find_field(name);
    if (f < 0 || !FIELDS[f].is_int) return -1;
    int** p = (int**)((char*)m + FIELDS[f].off);
    free(*p);
    *p = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
    memcpy(*p, v, sizeof(int) * n);
    ((ModelBox*)m)->count[f] = n;
    return 0;
}
int mjl_model_set_real(MjlModel* m, const char* name, const double* v, int n) {
    if (!strcmp(name, "opt_timestep")) { m->timestep = v[0]; return 0; }
    if (!strcmp(name, "opt_tolerance")) { m->tolerance = v[0]; return 0; }
    if (!strcmp(name, "gravity")) { v3copy(m->gravity, v); return 0; }
    if (!strcmp(name, "stat_meaninertia")) { m->meaninertia = v[0]; return 0; }
    int f = find_field(name);
    if (f < 0 || FIELDS[f].is_int) return -1;
    double** p = (double**)((char*)m + FIELDS[f].off);
    free(*p);
    *p = (double*)malloc(sizeof(double) * (n > 0 ? n : 1));
    memcpy(*p, v, sizeof(double) * n);
    ((ModelBox*)m)->count[f] = n;
    return 0;
}
static int cnt(const MjlModel* m, const char* name) { return ((const ModelBox*)m)->count[find_field(name)]; }
int mjl_model_finalize(MjlModel* m) {
    m->nq = cnt(m, "qpos0"); m->nv = cnt(m, "dof_bodyid"); m->nbody = cnt(m, "body_parentid");
    m->njnt = cnt(m, "jnt_type"); m->ngeom = cnt(m, "geom_type"); m->nsite = cnt(m, "site_bodyid");
    m->nmesh = cnt(m, "mesh_vertnum"); m->nmeshvert = cnt(m, "mesh_vert") / 3; m->npair = cnt(m, "pair_geom") / 2;
    m->nu = cnt(m, "act_dofid"); m->neq = cnt(m, "eq_body1");
    if (m->nv > MJL_MAXNV) return -2;
    if (!m->eq_data) {
        /* weld default: anchor 0, relpose from qpos0 would go here; the reference overwrites it
           (sawyer_xyz_env.py:133-140) with anchor=0, relpos=0, relquat=(-1,0,0,0), torquescale=5 */
        m->eq_data = (double*)calloc(11 * (m->neq > 0 ? m->neq : 1), sizeof(double));
        for (int i = 0; i < m->neq; i++) { m->eq_data[11 * i + 6] = 1; m->eq_data[11 * i + 10] = 1; }
        ((ModelBox*)m)->count[find_field("eq_data")] = 11 * m->neq;
    }
    return 0;
}
double* mjl_model_real_ptr(MjlModel* m, const char* name, int* n) {
    int f = find_field(name);
    if (f < 0 || FIELDS[f].is_int) return NULL;
    if (n) *n = ((ModelBox*)m)->count[f];
    return *(double**)((char*)m + FIELDS[f].off);
}

/* ------------------------------------------------------------- data storage */
typedef struct { const char* name; size_t off; } DField;
#define DF(n) { #n, offsetof(MjlData, n) }
static const DField DFIELDS[] = {
    DF(qpos), DF(qvel), DF(qacc_warmstart), DF(ctrl), DF(xpos), DF(xquat), DF(xmat), DF(xipos), DF(ximat),
    DF(xanchor), DF(xaxis), DF(geom_xpos), DF(geom_xmat), DF(site_xpos), DF(site_xmat), DF(cdof), DF(cdof_dot),
    DF(cvel), DF(qM), DF(qL), DF(qfrc_bias), DF(qfrc_passive), DF(qfrc_actuator), DF(qfrc_smooth), DF(qacc_smooth),
    DF(qfrc_constraint), DF(qacc), DF(efc_J),
};
#define NDFIELDS ((int)(sizeof(DFIELDS) / sizeof(DFIELDS[0])))
static int dsize(const MjlModel* m, const char* n) {
    int nv = m->nv, nb = m->nbody;
    if (!strcmp(n, "qpos")) return m->nq;
    if (!strcmp(n, "qvel") || !strcmp(n, "qacc_warmstart") || !strcmp(n, "qfrc_bias") || !strcmp(n, "qfrc_passive") ||
        !strcmp(n, "qfrc_actuator") || !strcmp(n, "qfrc_smooth") || !strcmp(n, "qacc_smooth") ||
        !strcmp(n, "qfrc_constraint") || !strcmp(n, "qacc")) return nv;
    if (!strcmp(n, "ctrl")) return m->nu;
    if (!strcmp(n, "xpos") || !strcmp(n, "xipos")) return 3 * nb;
    if (!strcmp(n, "xquat")) return 4 * nb;
    if (!strcmp(n, "xmat") || !strcmp(n, "ximat")) return 9 * nb;
    if (!strcmp(n, "xanchor") || !strcmp(n, "xaxis")) return 3 * m->njnt;
    if (!strcmp(n, "geom_xpos")) return 3 * m->ngeom;
    if (!strcmp(n, "geom_xmat")) return 9 * m->ngeom;
    if (!strcmp(n, "site_xpos")) return 3 * m->nsite;
    if (!strcmp(n, "site_xmat")) return 9 * m->nsite;
    if (!strcmp(n, "cdof") || !strcmp(n, "cdof_dot")) return 6 * nv;
    if (!strcmp(n, "cvel")) return 6 * nb;
    if (!strcmp(n, "qM") || !strcmp(n, "qL")) return nv * nv;
    if (!strcmp(n, "efc_J")) return MJL_MAXEFC * nv;
    return 0;
}
MjlData* mjl_data_new(const MjlModel* m) {
    MjlData* d = (MjlData*)calloc(1, sizeof(MjlData));
    for (int i = 0; i < NDFIELDS; i++) {
        int n = dsize(m, DFIELDS[i].name);
        *(double**)((char*)d + DFIELDS[i].off) = (double*)calloc(n > 0 ? n : 1, sizeof(double));
    }
    mjl_reset_data(m, d);
    return d;
}
void mjl_data_free(MjlData* d) {
    if (!d) return;
    for (int i = 0; i < NDFIELDS; i++) free(*(void**)((char*)d + DFIELDS[i].off));
    free(d);
}
double* mjl_data_real_ptr(const MjlModel* m, MjlData* d, const char* name, int* n) {
    if (!strcmp(name, "mocap_pos")) { if (n) *n = 3; return d->mocap_pos; }
    if (!strcmp(name, "mocap_quat")) { if (n) *n = 4; return d->mocap_quat; }
    if (!strcmp(name, "time")) { if (n) *n = 1; return &d->time; }
    if (!strcmp(name, "efc_force")) { if (n) *n = MJL_MAXEFC; return d->efc_force; }
    if (!strcmp(name, "efc_pos")) { if (n) *n = MJL_MAXEFC; return d->efc_pos; }
    if (!strcmp(name, "efc_aref")) { if (n) *n = MJL_MAXEFC; return d->efc_aref; }
    if (!strcmp(name, "efc_D")) { if (n) *n = MJL_MAXEFC; return d->efc_D; }
    for (int i = 0; i < NDFIELDS; i++)
        if (!strcmp(name, DFIELDS[i].name)) {
            if (n) *n = dsize(m, name);
            return *(double**)((char*)d + DFIELDS[i].off);
        }
    return NULL;
}

/* mj_resetData: qpos<-qpos0, everything else zero, mocap pose <- model default */
void mjl_reset_data(const MjlModel* m, MjlData* d) {
    memcpy(d->qpos, m->qpos0, sizeof(double) * m->nq);
    memset(d->qvel, 0, sizeof(double) * m->nv);
    memset(d->qacc_warmstart, 0, sizeof(double) * m->nv);
    memset(d->qacc, 0, sizeof(double) * m->nv);
    memset(d->ctrl, 0, sizeof(double) * m->nu);
    d->time = 0;
    v3set(d->mocap_pos, 0, 0, 0);
    d->mocap_quat[0] = 1; d->mocap_quat[1] = d->mocap_quat[2] = d->mocap_quat[3] = 0;
    for (int b = 0; b < m->nbody; b++)
        if (m->body_mocap[b]) { v3copy(d->mocap_pos, m->body_pos + 3 * b); memcpy(d->mocap_quat, m->body_quat + 4 * b, 4 * sizeof(double)); }
    d->ncon = d->nefc = 0;
}

/* ------------------------------------------------------------- kinematics */
void mjl_kinematics(const MjlModel* m, MjlData* d) {
    v3set(d->xpos, 0, 0, 0);
    d->xquat[0] = 1; d->xquat[1] = d->xquat[2] = d->xquat[3] = 0;
    q2mat(d->xmat, d->xquat);
    v3set(d->xipos, 0, 0, 0);
    q2mat(d->ximat, d->xquat);
    for (int b = 1; b < m->nbody; b++) {
        int p = m->body_parentid[b];
        double pos[3], quat[4], mat[9], t[3];
        if (m->body_mocap[b]) {
            v3copy(pos, d->mocap_pos);
            memcpy(quat, d->mocap_quat, sizeof quat);
            qnormalize(quat);
        } else {
            m3mulv(t, d->xmat + 9 * p, m->body_pos + 3 * b);
            v3add(pos, d->xpos + 3 * p, t);
            qmul(quat, d->xquat + 4 * p, m->body_quat + 4 * b);
        }
        for (int k = 0; k < m->body_jntnum[b]; k++) {
            int j = m->body_jntadr[b] + k, qa = m->jnt_qposadr[j];
            if (m->jnt_type[j] == MJL_FREE) {
                qnormalize(d->qpos + qa + 3);
                v3copy(pos, d->qpos + qa);
                memcpy(quat, d->qpos + qa + 3, sizeof quat);
                v3copy(d->xanchor + 3 * j, pos);
                v3set(d->xaxis + 3 * j, 0, 0, 1);
            } else {
                q2mat(mat, quat);
                m3mulv(t, mat, m->jnt_pos + 3 * j);
                v3add(d->xanchor + 3 * j, pos, t);
                m3mulv(d->xaxis + 3 * j, mat, m->jnt_axis + 3 * j);
                if (m->jnt_type[j] == MJL_SLIDE) {
                    v3addscl(pos, pos, d->xaxis + 3 * j, d->qpos[qa]);
                } else {
                    double h = 0.5 * d->qpos[qa], s = sin(h), qr[4], qn[4];
                    qr[0] = cos(h); qr[1] = s * m->jnt_axis[3 * j]; qr[2] = s * m->jnt_axis[3 * j + 1]; qr[3] = s * m->jnt_axis[3 * j + 2];
                    qmul(qn, quat, qr);
                    memcpy(quat, qn, sizeof quat);
                    q2mat(mat, quat);
                    m3mulv(t, mat, m->jnt_pos + 3 * j);
                    v3sub(pos, d->xanchor + 3 * j, t);
                }
            }
        }
        qnormalize(quat);
        v3copy(d->xpos + 3 * b, pos);
        memcpy(d->xquat + 4 * b, quat, sizeof quat);
        q2mat(d->xmat + 9 * b, quat);
        m3mulv(t, d->xmat + 9 * b, m->body_ipos + 3 * b);
        v3add(d->xipos + 3 * b, pos, t);
        double iq[4];
        qmul(iq, quat, m->body_iquat + 4 * b);
        q2mat(d->ximat + 9 * b, iq);
    }
    for (int g = 0; g < m->ngeom; g++) {
        int b = m->geom_bodyid[g];
        double t[3], q[4];
        m3mulv(t, d->xmat + 9 * b, m->geom_pos + 3 * g);
        v3add(d->geom_xpos + 3 * g, d->xpos + 3 * b, t);
        qmul(q, d->xquat + 4 * b, m->geom_quat + 4 * g);
        q2mat(d->geom_xmat + 9 * g, q);
    }
    for (int s = 0; s < m->nsite; s++) {
        int b = m->site_bodyid[s];
        double t[3], q[4];
        m3mulv(t, d->xmat + 9 * b, m->site_pos + 3 * s);
        v3add(d->site_xpos + 3 * s, d->xpos + 3 * b, t);
        qmul(q, d->xquat + 4 * b, m->site_quat + 4 * s);
        q2mat(d->site_xmat + 9 * s, q);
    }
    /* motion axes about the world origin: cdof = [angular; linear at origin] */
    for (int j = 0; j < m->njnt; j++) {
        int da = m->jnt_dofadr[j], b = m->jnt_bodyid[j];
        if (m->jnt_type[j] == MJL_FREE) {
            for (int k = 0; k < 3; k++) {
                double* c = d->cdof + 6 * (da + k);
                memset(c, 0, 6 * sizeof(double));
                c[3 + k] = 1;
                double* r = d->cdof + 6 * (da + 3 + k);
                double ax[3] = { d->xmat[9 * b + k], d->xmat[9 * b + 3 + k], d->xmat[9 * b + 6 + k] };
                v3copy(r, ax);
                v3cross(r + 3, d->xpos + 3 * b, ax);
            }
        } else if (m->jnt_type[j] == MJL_SLIDE) {
            double* c = d->cdof + 6 * da;
            v3set(c, 0, 0, 0);
            v3copy(c + 3, d->xaxis + 3 * j);
        } else {
            double* c = d->cdof + 6 * da;
            v3copy(c, d->xaxis + 3 * j);
            v3cross(c + 3, d->xanchor + 3 * j, d->xaxis + 3 * j);
        }
    }
}

/* spatial inertia about world origin: {mass, h[3]=m*c, J[6]=(xx,yy,zz,xy,xz,yz)} */
static void body_spatial_inertia(const MjlModel* m, const MjlData* d, int b, double* I10) {
    double mass = m->body_mass[b];
    const double* c = d->xipos + 3 * b;
    const double* R = d->ximat + 9 * b;
    const double* di = m->body_inertia + 3 * b;
    double Ic[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            Ic[3 * i + j] = R[3 * i] * di[0] * R[3 * j] + R[3 * i + 1] * di[1] * R[3 * j + 1] + R[3 * i + 2] * di[2] * R[3 * j + 2];
    double cc = v3dot(c, c);
    I10[0] = mass;
    I10[1] = mass * c[0]; I10[2] = mass * c[1]; I10[3] = mass * c[2];
    I10[4] = Ic[0] + mass * (cc - c[0] * c[0]);
    I10[5] = Ic[4] + mass * (cc - c[1] * c[1]);
    I10[6] = Ic[8] + mass * (cc - c[2] * c[2]);
    I10[7] = Ic[1] - mass * c[0] * c[1];
    I10[8] = Ic[2] - mass * c[0] * c[2];
    I10[9] = Ic[5] - mass * c[1] * c[2];
}
/* f[ang(3); lin(3)] = I * s, s = [w; v] */
static void inertia_mul(double* f, const double* I, const double* s) {
    const double *w = s, *v = s + 3, *h = I + 1;
    double t[3];
    /* linear momentum p = m v + w x h */
    v3cross(t, w, h);
    f[3] = I[0] * v[0] + t[0]; f[4] = I[0] * v[1] + t[1]; f[5] = I[0] * v[2] + t[2];
    /* angular about origin L = J w + h x v */
    v3cross(t, h, v);
    f[0] = I[4] * w[0] + I[7] * w[1] + I[8] * w[2] + t[0];
    f[1] = I[7] * w[0] + I[5] * w[1] + I[9] * w[2] + t[1];
    f[2] = I[8] * w[0] + I[9] * w[1] + I[6] * w[2] + t[2];
}
static void cross_motion(double* r, const double* v, const double* s) {
    double a[3], b[3], c[3];
    v3cross(a, v, s);
    v3cross(b, v, s + 3);
    v3cross(c, v + 3, s);
    v3copy(r, a);
    v3add(r + 3, b, c);
}
static void cross_force(double* r, const double* v, const double* f) {
    double a[3], b[3], c[3];
    v3cross(a, v, f);
    v3cross(b, v + 3, f + 3);
    v3cross(c, v, f + 3);
    v3add(r, a, b);
    v3copy(r + 3, c);
}

/* composite rigid body: dense symmetric qM (+armature), then dense Cholesky qL (lower) */
static int chol(double* L, const double* A, int n) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++) {
            double s = A[i * n + j];
            for (int k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k];
            if (i == j) {
                if (s < MINVAL) s = MINVAL;
                L[i * n + i] = sqrt(s);
            } else L[i * n + j] = s / L[j * n + j];
        }
    for (int i = 0; i < n; i++)
        for (int j = i + 1; j < n; j++) L[i * n + j] = 0;
    return 0;
}
static void chol_solve(const double* L, double* x, int n) {
    for (int i = 0; i < n; i++) {
        double s = x[i];
        for (int k = 0; k < i; k++) s -= L[i * n + k] * x[k];
        x[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = x[i];
        for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k];
        x[i] = s / L[i * n + i];
    }
}
void mjl_crb(const MjlModel* m, MjlData* d) {
    int nb = m->nbody, nv = m->nv;
    double* crb = (double*)malloc(sizeof(double) * 10 * nb);
    for (int b = 0; b < nb; b++) body_spatial_inertia(m, d, b, crb + 10 * b);
    for (int b = nb - 1; b > 0; b--) {
        int p = m->body_parentid[b];
        for (int k = 0; k < 10; k++) crb[10 * p + k] += crb[10 * b + k];
    }
    memset(d->qM, 0, sizeof(double) * nv * nv);
    for (int i = 0; i < nv; i++) {
        double f[6];
        inertia_mul(f, crb + 10 * m->dof_bodyid[i], d->cdof + 6 * i);
        for (int j = i; j >= 0; j = m->dof_parentid[j]) {
            const double* s = d->cdof + 6 * j;
            double v = s[0] * f[0] + s[1] * f[1] + s[2] * f[2] + s[3] * f[3] + s[4] * f[4] + s[5] * f[5];
            d->qM[i * nv + j] = d->qM[j * nv + i] = v;
        }
        d->qM[i * nv + i] += m->dof_armature[i];
    }
    free(crb);
    chol(d->qL, d->qM, nv);
}

/* body velocities, cdof_dot, and bias forces C(q,v)+g via recursive Newton-Euler */
void mjl_rne_bias(const MjlModel* m, MjlData* d) {
    int nb = m->nbody, nv = m->nv;
    double* cacc = (double*)calloc(6 * nb, sizeof(double));
    double* cfrc = (double*)calloc(6 * nb, sizeof(double));
    memset(d->cvel, 0, sizeof(double) * 6);
    cacc[3] = -m->gravity[0]; cacc[4] = -m->gravity[1]; cacc[5] = -m->gravity[2];
    for (int b = 1; b < nb; b++) {
        int p = m->body_parentid[b];
        double v[6], a[6];
        memcpy(v, d->cvel + 6 * p, sizeof v);
        memcpy(a, cacc + 6 * p, sizeof a);
        for (int k = 0; k < m->body_jntnum[b]; k++) {
            int j = m->body_jntadr[b] + k, da = m->jnt_dofadr[j];
            if (m->jnt_type[j] == MJL_FREE) {
                for (int i = 0; i < 3; i++) {
                    memset(d->cdof_dot + 6 * (da + i), 0, 6 * sizeof(double));
                    for (int c = 0; c < 6; c++) v[c] += d->cdof[6 * (da + i) + c] * d->qvel[da + i];
                }
                double vs[6];
                memcpy(vs, v, sizeof vs);
                for (int i = 3; i < 6; i++) {
                    cross_motion(d->cdof_dot + 6 * (da + i), vs, d->cdof + 6 * (da + i));
                    for (int c = 0; c < 6; c++) {
                        v[c] += d->cdof[6 * (da + i) + c] * d->qvel[da + i];
                        a[c] += d->cdof_dot[6 * (da + i) + c] * d->qvel[da + i];
                    }
                }
            } else {
                cross_motion(d->cdof_dot + 6 * da, v, d->cdof + 6 * da);
                for (int c = 0; c < 6; c++) {
                    v[c] += d->cdof[6 * da + c] * d->qvel[da];
                    a[c] += d->cdof_dot[6 * da + c] * d->qvel[da];
                }
            }
        }
        memcpy(d->cvel + 6 * b, v, sizeof v);
        memcpy(cacc + 6 * b, a, sizeof a);
        double I[10], Ia[6], Iv[6], t[6];
        body_spatial_inertia(m, d, b, I);
        inertia_mul(Ia, I, a);
        inertia_mul(Iv, I, v);
        cross_force(t, v, Iv);
        for (int c = 0; c < 6; c++) cfrc[6 * b + c] = Ia[c] + t[c];
    }
    for (int b = nb - 1; b > 0; b--) {
        int p = m->body_parentid[b];
        for (int c = 0; c < 6; c++) cfrc[6 * p + c] += cfrc[6 * b + c];
    }
    for (int i = 0; i < nv; i++) {
        const double *s = d->cdof + 6 * i, *f = cfrc + 6 * m->dof_bodyid[i];
        d->qfrc_bias[i] = s[0] * f[0] + s[1] * f[1] + s[2] * f[2] + s[3] * f[3] + s[4] * f[4] + s[5] * f[5];
    }
    free(cacc);
    free(cfrc);
}

/* translational / rotational Jacobian (3 x nv each, row-major) of a world point fixed to `body` */
void mjl_jac(const MjlModel* m, const MjlData* d, double* jacp, double* jacr, const double point[3], int body) {
    int nv = m->nv;
    if (jacp) memset(jacp, 0, sizeof(double) * 3 * nv);
    if (jacr) memset(jacr, 0, sizeof(double) * 3 * nv);
    for (int i = m->body_lastdof[body]; i >= 0; i = m->dof_parentid[i]) {
        const double* s = d->cdof + 6 * i;
        double t[3];
        v3cross(t, s, point);
        for (int k = 0; k < 3; k++) {
            if (jacr) jacr[k * nv + i] = s[k];
            if (jacp) jacp[k * nv + i] = s[3 + k] + t[k];
        }
    }
}

/* ------------------------------------------------------------- constraints */
static int add_rows(const MjlModel* m, MjlData* d, int n, int type, int id) {
    if (d->nefc + n > MJL_MAXEFC) { d->warning_overflow++; return -1; }
    int r0 = d->nefc;
    for (int i = 0; i < n; i++) {
        d->efc_type[r0 + i] = type; d->efc_id[r0 + i] = id;
        d->efc_pos[r0 + i] = 0; d->efc_margin[r0 + i] = 0; d->efc_diagApprox[r0 + i] = 0;
        memset(d->efc_J + (size_t)(r0 + i) * m->nv, 0, sizeof(double) * m->nv);
    }
    d->nefc += n;
    return r0;
}

static void make_frame(double* f) {
    v3normalize(f);
    if (v3norm(f + 3) < 0.5) {
        v3set(f + 3, 0, 0, 0);
        if (f[1] < 0.5 && f[1] > -0.5) f[4] = 1; else f[5] = 1;
    }
    double t = v3dot(f, f + 3);
    v3addscl(f + 3, f + 3, f, -t);
    v3normalize(f + 3);
    v3cross(f + 6, f, f + 3);
}

static void make_constraints(const MjlModel* m, MjlData* d) {
    int nv = m->nv;
    double jp1[3 * MJL_MAXNV], jr1[3 * MJL_MAXNV], jp2[3 * MJL_MAXNV], jr2[3 * MJL_MAXNV];
    d->nefc = d->ne = d->nl = 0;
    /* ---- weld equalities (6 rows each) ---- */
    for (int e = 0; e < m->neq; e++) {
        int b1 = m->eq_body1[e], b2 = m->eq_body2[e];
        const double* data = m->eq_data + 11 * e;
        double p1[3], p2[3], t[3], cpos[6];
        m3mulv(t, d->xmat + 9 * b1, data + 3); v3add(p1, d->xpos + 3 * b1, t);
        m3mulv(t, d->xmat + 9 * b2, data + 0); v3add(p2, d->xpos + 3 * b2, t);
        v3sub(cpos, p1, p2);
        mjl_jac(m, d, jp1, jr1, p1, b1);
        mjl_jac(m, d, jp2, jr2, p2, b2);
        double ts = data[10], q[4], q1n[4], q2[4];
        qmul(q, d->xquat + 4 * b1, data + 6);
        q1n[0] = d->xquat[4 * b2]; q1n[1] = -d->xquat[4 * b2 + 1]; q1n[2] = -d->xquat[4 * b2 + 2]; q1n[3] = -d->xquat[4 * b2 + 3];
        qmul(q2, q1n, q);
        cpos[3] = ts * q2[1]; cpos[4] = ts * q2[2]; cpos[5] = ts * q2[3];
        int r0 = add_rows(m, d, 6, MJL_EQUALITY, e);
        if (r0 < 0) continue;
        for (int i = 0; i < nv; i++) {
            for (int k = 0; k < 3; k++) d->efc_J[(size_t)(r0 + k) * nv + i] = jp1[k * nv + i] - jp2[k * nv + i];
            double ax[4] = { 0, jr1[i] - jr2[i], jr1[nv + i] - jr2[nv + i], jr1[2 * nv + i] - jr2[2 * nv + i] };
            double q3[4], q4[4];
            qmul(q3, q1n, ax);
            qmul(q4, q3, q);
            for (int k = 0; k < 3; k++) d->efc_J[(size_t)(r0 + 3 + k) * nv + i] = 0.5 * q4[1 + k] * ts;
        }
        for (int k = 0; k < 6; k++) {
            d->efc_pos[r0 + k] = cpos[k];
            d->efc_diagApprox[r0 + k] = m->body_invweight0[2 * b1 + (k >= 3)] + m->body_invweight0[2 * b2 + (k >= 3)];
        }
        d->ne += 6;
    }
    /* ---- joint limits ---- */
    for (int j = 0; j < m->njnt; j++) {
        if (!m->jnt_limited[j] || m->jnt_type[j] == MJL_FREE) continue;
        double q = d->qpos[m->jnt_qposadr[j]], margin = m->jnt_margin[j];
        for (int side = -1; side <= 1; side += 2) {
            double dist = side * (m->jnt_range[2 * j + (side + 1) / 2] - q);
            if (dist < margin) {
                int r = add_rows(m, d, 1, MJL_LIMIT, j);
                if (r < 0) continue;
                d->efc_J[(size_t)r * nv + m->jnt_dofadr[j]] = -(double)side;
                d->efc_pos[r] = dist; d->efc_margin[r] = margin;
                d->efc_diagApprox[r] = m->dof_invweight0[m->jnt_dofadr[j]];
                d->nl++;
            }
        }
    }
    /* ---- contacts (elliptic cones) ---- */
    for (int c = 0; c < d->ncon; c++) {
        MjlContact* con = d->contact + c;
        con->efc_address = -1;
        if (con->exclude || con->dist >= con->includemargin) continue;
        int b1 = m->geom_bodyid[con->geom1], b2 = m->geom_bodyid[con->geom2];
        int r0 = add_rows(m, d, con->dim, MJL_CONTACT_ELLIPTIC, c);
        if (r0 < 0) continue;
        con->efc_address = r0;
        mjl_jac(m, d, jp1, jr1, con->pos, b1);
        mjl_jac(m, d, jp2, jr2, con->pos, b2);
        for (int k = 0; k < con->dim; k++) {
            const double* ax = con->frame + 3 * (k < 3 ? k : k - 3);
            const double *ja = k < 3 ? jp1 : jr1, *jb = k < 3 ? jp2 : jr2;
            double* row = d->efc_J + (size_t)(r0 + k) * nv;
            for (int i = 0; i < nv; i++)
                row[i] = ax[0] * (jb[i] - ja[i]) + ax[1] * (jb[nv + i] - ja[nv + i]) + ax[2] * (jb[2 * nv + i] - ja[2 * nv + i]);
            d->efc_diagApprox[r0 + k] = m->body_invweight0[2 * b1 + (k >= 3)] + m->body_invweight0[2 * b2 + (k >= 3)];
        }
        d->efc_pos[r0] = con->dist;
        d->efc_margin[r0] = con->includemargin;
    }
}

static double impedance(const double* solimp, double x) {
    double d0 = solimp[0], dw = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
    d0 = fmin(MAXIMP, fmax(MINIMP, d0)); dw = fmin(MAXIMP, fmax(MINIMP, dw));
    mid = fmin(MAXIMP, fmax(MINIMP, mid)); power = fmax(1, power);
    if (width < MINVAL || d0 == dw) return 0.5 * (d0 + dw);
    x = fabs(x) / width;
    if (x >= 1) return dw;
    if (x <= 0) return d0;
    double y;
    if (power == 1) y = x;
    else if (x <= mid) y = pow(x / mid, power) * mid;       /* = x^p / mid^(p-1) */
    else y = 1 - pow((1 - x) / (1 - mid), power) * (1 - mid);
    return d0 + y * (dw - d0);
}

static void make_impedance(const MjlModel* m, MjlData* d) {
    if (d->ncon > d->max_ncon) d->max_ncon = d->ncon;
    if (d->nefc > d->max_nefc) d->max_nefc = d->nefc;
    for (int i = 0; i < d->nefc; i++) {
        const double *ref, *imp;
        int type = d->efc_type[i], id = d->efc_id[i], dim = 1;
        if (type == MJL_EQUALITY) { ref = m->eq_solref + 2 * id; imp = m->eq_solimp + 5 * id; }
        else if (type == MJL_LIMIT) { ref = m->jnt_solref + 2 * id; imp = m->jnt_solimp + 5 * id; }
        else { ref = d->contact[id].solref; imp = d->contact[id].solimp; dim = d->contact[id].dim; }
        double tc = ref[0], dr = ref[1];
        if (tc > 0) tc = fmax(tc, 2 * m->timestep);
        double dmax = fmin(MAXIMP, fmax(MINIMP, imp[1]));
        double K = 1 / fmax(MINVAL, dmax * dmax * tc * tc * dr * dr);
        double B = 2 / fmax(MINVAL, dmax * tc);
        double I = impedance(imp, d->efc_pos[i] - d->efc_margin[i]);
        d->efc_R[i] = fmax(MINVAL, (1 - I) / I * d->efc_diagApprox[i]);
        d->efc_KBIP[i][0] = K; d->efc_KBIP[i][1] = B; d->efc_KBIP[i][2] = I; d->efc_KBIP[i][3] = 0;
        if (type == MJL_CONTACT_ELLIPTIC) {
            MjlContact* con = d->contact + id;
            /* friction rows: regulariser scaled by friction ratios; same damping B, no stiffness (pos=0) */
            d->efc_R[i + 1] = d->efc_R[i] / IMPRATIO;
            for (int j = 1; j < dim - 1; j++)
                d->efc_R[i + 1 + j] = d->efc_R[i + 1] * con->friction[0] * con->friction[0] / (con->friction[j] * con->friction[j]);
            con->mu = con->friction[0] * sqrt(d->efc_R[i + 1] / d->efc_R[i]);
            for (int j = 1; j < dim; j++) {
                d->efc_KBIP[i + j][0] = 0; d->efc_KBIP[i + j][1] = B; d->efc_KBIP[i + j][2] = I; d->efc_KBIP[i + j][3] = 0;
            }
            for (int j = 0; j < dim; j++) d->efc_D[i + j] = 1 / d->efc_R[i + j];
            i += dim - 1;
        } else d->efc_D[i] = 1 / d->efc_R[i];
    }
}

static void reference_accel(const MjlModel* m, MjlData* d) {
    int nv = m->nv;
    for (int i = 0; i < d->nefc; i++) {
        double v = 0;
        const double* row = d->efc_J + (size_t)i * nv;
        for (int k = 0; k < nv; k++) v += row[k] * d->qvel[k];
        d->efc_vel[i] = v;
        d->efc_aref[i] = -d->efc_KBIP[i][1] * v - d->efc_KBIP[i][0] * d->efc_KBIP[i][2] * (d->efc_pos[i] - d->efc_margin[i]);
    }
}

/* ------------------------------------------------------------------ solver */
typedef struct {
    int nv, nefc;
    double jar[MJL_MAXEFC], Jv[MJL_MAXEFC], Ma[MJL_MAXNV], Mv[MJL_MAXNV], grad[MJL_MAXNV], search[MJL_MAXNV];
    double cost, gauss;
} Ctx;

/* constraint cost, forces, states at jar; optional cone Hessian blocks (16 doubles per contact row0) */
static double constraint_update(const MjlModel* m, MjlData* d, const double* jar, double* force, int* state, double* hcone) {
    double cost = 0;
    for (int i = 0; i < d->nefc; i++) {
        int type = d->efc_type[i];
        double D = d->efc_D[i];
        if (type == MJL_EQUALITY) {
            force[i] = -D * jar[i]; cost += 0.5 * D * jar[i] * jar[i]; state[i] = MJL_QUADRATIC;
        } else if (type == MJL_LIMIT) {
            if (jar[i] < 0) { force[i] = -D * jar[i]; cost += 0.5 * D * jar[i] * jar[i]; state[i] = MJL_QUADRATIC; }
            else { force[i] = 0; state[i] = MJL_SATISFIED; }
        } else {
            const MjlContact* con = d->contact + d->efc_id[i];
            int dim = con->dim;
            double mu = con->mu, U[6], N, T = 0;
            U[0] = jar[i] * mu;
            for (int j = 1; j < dim; j++) { U[j] = jar[i + j] * con->friction[j - 1]; T += U[j] * U[j]; }
            T = sqrt(T); N = U[0];
            int st;
            if (N >= mu * T || (T <= 0 && N >= 0)) {
                st = MJL_SATISFIED;
                for (int j = 0; j < dim; j++) force[i + j] = 0;
            } else if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
                st = MJL_QUADRATIC;
                for (int j = 0; j < dim; j++) { force[i + j] = -d->efc_D[i + j] * jar[i + j]; cost += 0.5 * d->efc_D[i + j] * jar[i + j] * jar[i + j]; }
            } else {
                st = MJL_CONE;
                double Dm = D / (mu * mu * (1 + mu * mu)), NmT = N - mu * T;
                cost += 0.5 * Dm * NmT * NmT;
                force[i] = -Dm * NmT * mu;
                for (int j = 1; j < dim; j++) force[i + j] = -force[i] / T * U[j] * con->friction[j - 1];
                if (hcone) {
                    double* H = hcone + 36 * d->efc_id[i];
                    double fri[6];
                    fri[0] = mu;
                    for (int j = 1; j < dim; j++) fri[j] = con->friction[j - 1];
                    H[0] = 1;
                    for (int j = 1; j < dim; j++) H[j] = H[6 * j] = -mu * U[j] / T;
                    double scl = mu * N / (T * T * T), dg = mu * mu - mu * N / T;
                    for (int k = 1; k < dim; k++)
                        for (int j = 1; j < dim; j++) H[6 * k + j] = scl * U[k] * U[j] + (k == j ? dg : 0);
                    for (int k = 0; k < dim; k++)
                        for (int j = 0; j < dim; j++) H[6 * k + j] *= Dm * fri[k] * fri[j];
                }
            }
            for (int j = 0; j < dim; j++) state[i + j] = st;
            i += dim - 1;
        }
    }
    return cost;
}

static void mul_M(const MjlModel* m, const MjlData* d, double* r, const double* v) {
    int nv = m->nv;
    for (int i = 0; i < nv; i++) {
        double s = 0;
        for (int k = 0; k < nv; k++) s += d->qM[i * nv + k] * v[k];
        r[i] = s;
    }
}
static void mul_J(const MjlModel* m, const MjlData* d, double* r, const double* v) {
    int nv = m->nv;
    for (int i = 0; i < d->nefc; i++) {
        double s = 0;
        const double* row = d->efc_J + (size_t)i * nv;
        for (int k = 0; k < nv; k++) s += row[k] * v[k];
        r[i] = s;
    }
}

/* total cost at qacc (Gauss + constraint), filling ctx->jar/Ma and d->efc_force/state/qfrc_constraint */
static void update_constraint(const MjlModel* m, MjlData* d, Ctx* c, double* hcone) {
    int nv = m->nv;
    c->cost = constraint_update(m, d, c->jar, d->efc_force, d->efc_state, hcone);
    for (int k = 0; k < nv; k++) {
        double s = 0;
        for (int i = 0; i < d->nefc; i++) s += d->efc_J[(size_t)i * nv + k] * d->efc_force[i];
        d->qfrc_constraint[k] = s;
    }
    double g = 0;
    for (int k = 0; k < nv; k++) g += (c->Ma[k] - d->qfrc_smooth[k]) * (d->qacc[k] - d->qacc_smooth[k]);
    c->gauss = 0.5 * g;
    c->cost += c->gauss;
}

/* 1-D cost along qacc + alpha*search: returns cost, d1, d2 */
static void line_eval(const MjlModel* m, const MjlData* d, const Ctx* c, const double quadGauss[3], double alpha,
                      double* cost, double* d1, double* d2) {
    double C = alpha * alpha * quadGauss[2] + alpha * quadGauss[1] + quadGauss[0];
    double D1 = 2 * alpha * quadGauss[2] + quadGauss[1], D2 = 2 * quadGauss[2];
    for (int i = 0; i < d->nefc; i++) {
        int type = d->efc_type[i];
        double D = d->efc_D[i];
        double x = c->jar[i] + alpha * c->Jv[i];
        if (type == MJL_EQUALITY || (type == MJL_LIMIT && x < 0)) {
            C += 0.5 * D * x * x; D1 += D * x * c->Jv[i]; D2 += D * c->Jv[i] * c->Jv[i];
        } else if (type == MJL_CONTACT_ELLIPTIC) {
            const MjlContact* con = d->contact + d->efc_id[i];
            int dim = con->dim;
            double mu = con->mu;
            /* N(alpha) = U0 + alpha V0 ; T^2 = UU + 2 alpha UV + alpha^2 VV */
            double U0 = c->jar[i] * mu, V0 = c->Jv[i] * mu, UU = 0, UV = 0, VV = 0;
            for (int j = 1; j < dim; j++) {
                double f = con->friction[j - 1], u = c->jar[i + j] * f, v = c->Jv[i + j] * f;
                UU += u * u; UV += u * v; VV += v * v;
            }
            double N = U0 + alpha * V0;
            double Tsqr = UU + alpha * (2 * UV + alpha * VV);
            if (Tsqr <= 0) {
                if (N < 0) {  /* bottom zone with zero tangential part: quadratic in all rows */
                    for (int j = 0; j < dim; j++) {
                        double xx = c->jar[i + j] + alpha * c->Jv[i + j], DD = d->efc_D[i + j];
                        C += 0.5 * DD * xx * xx; D1 += DD * xx * c->Jv[i + j]; D2 += DD * c->Jv[i + j] * c->Jv[i + j];
                    }
                }
            } else {
                double T = sqrt(Tsqr);
                if (N >= mu * T) {
                    /* top: nothing */
                } else if (mu * N + T <= 0) {
                    for (int j = 0; j < dim; j++) {
                        double xx = c->jar[i + j] + alpha * c->Jv[i + j], DD = d->efc_D[i + j];
                        C += 0.5 * DD * xx * xx; D1 += DD * xx * c->Jv[i + j]; D2 += DD * c->Jv[i + j] * c->Jv[i + j];
                    }
                } else {
                    double Dm = D / (mu * mu * (1 + mu * mu));
                    double N1 = V0, T1 = (UV + alpha * VV) / T;
                    double T2 = VV / T - (UV + alpha * VV) * T1 / (T * T);
                    double NmT = N - mu * T;
                    C += 0.5 * Dm * NmT * NmT;
                    D1 += Dm * NmT * (N1 - mu * T1);
                    D2 += Dm * ((N1 - mu * T1) * (N1 - mu * T1) - NmT * mu * T2);
                }
            }
            i += dim - 1;
        }
    }
    *cost = C; *d1 = D1; *d2 = D2;
}

/* exact line search on a convex piecewise-smooth 1-D function: safeguarded Newton */
static double line_search(const MjlModel* m, MjlData* d, Ctx* c) {
    int nv = m->nv;
    double snorm = 0;
    for (int k = 0; k < nv; k++) snorm += c->search[k] * c->search[k];
    snorm = sqrt(snorm);
    if (snorm < MINVAL) return 0;
    mul_M(m, d, c->Mv, c->search);
    mul_J(m, d, c->Jv, c->search);
    double quadGauss[3] = { c->gauss, 0, 0 };
    for (int k = 0; k < nv; k++) {
        quadGauss[1] += c->search[k] * (c->Ma[k] - d->qfrc_smooth[k]);
        quadGauss[2] += 0.5 * c->search[k] * c->Mv[k];
    }
    double scale = m->meaninertia * (nv > 1 ? nv : 1);
    double gtol = m->tolerance * 0.01 * snorm * scale;
    double c0, d1, d2;
    line_eval(m, d, c, quadGauss, 0, &c0, &d1, &d2);
    if (d1 >= 0 || d2 <= 0) return 0;  /* not a descent direction */
    double lo = 0, hi = -1, a = -d1 / d2, dlo = d1;
    for (int it = 0; it < 100; it++) {
        double ca, da, dda;
        line_eval(m, d, c, quadGauss, a, &ca, &da, &dda);
        if (fabs(da) < gtol) return a;
        if (da < 0) { lo = a; dlo = da; } else hi = a;
        double an = a - da / dda;
        if (hi < 0) {
            if (an <= lo) an = 2 * a + 1e-12;   /* keep expanding until bracketed */
        } else if (!(an > lo && an < hi)) an = 0.5 * (lo + hi);
        if (hi > 0 && (hi - lo) < 1e-16 * (1 + fabs(hi))) return 0.5 * (lo + hi);
        a = an;
    }
    (void)dlo;
    return a;
}

static void solve_newton(const MjlModel* m, MjlData* d) {
    int nv = m->nv, nefc = d->nefc;
    Ctx* c = (Ctx*)calloc(1, sizeof(Ctx));
    double* hcone = (double*)calloc(36 * (d->ncon > 0 ? d->ncon : 1), sizeof(double));
    double* H = (double*)malloc(sizeof(double) * nv * nv);
    double* HL = (double*)malloc(sizeof(double) * nv * nv);
    c->nv = nv; c->nefc = nefc;
    d->solver_niter = 0;
    if (nefc == 0) {
        memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv);
        memset(d->qfrc_constraint, 0, sizeof(double) * nv);
        goto done;
    }
    /* warm start: the better of qacc_warmstart and qacc_smooth */
    {
        double force[MJL_MAXEFC]; int state[MJL_MAXEFC];
        memcpy(d->qacc, d->qacc_warmstart, sizeof(double) * nv);
        mul_J(m, d, c->jar, d->qacc);
        for (int i = 0; i < nefc; i++) c->jar[i] -= d->efc_aref[i];
        double cw = constraint_update(m, d, c->jar, force, state, NULL);
        mul_M(m, d, c->Ma, d->qacc);
        for (int k = 0; k < nv; k++) cw += 0.5 * (c->Ma[k] - d->qfrc_smooth[k]) * (d->qacc[k] - d->qacc_smooth[k]);
        double jb[MJL_MAXEFC];
        mul_J(m, d, jb, d->qacc_smooth);
        for (int i = 0; i < nefc; i++) jb[i] -= d->efc_aref[i];
        double cs = constraint_update(m, d, jb, force, state, NULL);
        if (cw > cs) memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv);
    }
    mul_M(m, d, c->Ma, d->qacc);
    mul_J(m, d, c->jar, d->qacc);
    for (int i = 0; i < nefc; i++) c->jar[i] -= d->efc_aref[i];
    double scale = 1 / (m->meaninertia * (nv > 1 ? nv : 1));
    update_constraint(m, d, c, hcone);
    for (int iter = 0; iter < m->iterations; iter++) {
        /* gradient, Hessian H = M + J' D_active J (+ cone blocks), Newton direction */
        double gnorm = 0;
        for (int k = 0; k < nv; k++) { c->grad[k] = c->Ma[k] - d->qfrc_smooth[k] - d->qfrc_constraint[k]; gnorm += c->grad[k] * c->grad[k]; }
        if (scale * sqrt(gnorm) < m->tolerance) break;
        memcpy(H, d->qM, sizeof(double) * nv * nv);
        for (int i = 0; i < nefc; i++) {
            if (d->efc_state[i] == MJL_QUADRATIC) {
                const double* row = d->efc_J + (size_t)i * nv;
                double D = d->efc_D[i];
                for (int a = 0; a < nv; a++) {
                    if (row[a] == 0) continue;
                    double Da = D * row[a];
                    for (int b = 0; b < nv; b++) H[a * nv + b] += Da * row[b];
                }
            } else if (d->efc_state[i] == MJL_CONE) {
                const MjlContact* con = d->contact + d->efc_id[i];
                int dim = con->dim;
                const double* Hc = hcone + 36 * d->efc_id[i];
                for (int r = 0; r < dim; r++)
                    for (int s = 0; s < dim; s++) {
                        double h = Hc[6 * r + s];
                        if (h == 0) continue;
                        const double *jr_ = d->efc_J + (size_t)(i + r) * nv, *js = d->efc_J + (size_t)(i + s) * nv;
                        for (int a = 0; a < nv; a++) {
                            if (jr_[a] == 0) continue;
                            double ha = h * jr_[a];
                            for (int b = 0; b < nv; b++) H[a * nv + b] += ha * js[b];
                        }
                    }
                i += dim - 1;
            }
        }
        chol(HL, H, nv);
        for (int k = 0; k < nv; k++) c->search[k] = -c->grad[k];
        chol_solve(HL, c->search, nv);
        double alpha = line_search(m, d, c);
        if (alpha == 0) break;
        for (int k = 0; k < nv; k++) { d->qacc[k] += alpha * c->search[k]; c->Ma[k] += alpha * c->Mv[k]; }
        for (int i = 0; i < nefc; i++) c->jar[i] += alpha * c->Jv[i];
        double oldcost = c->cost;
        update_constraint(m, d, c, hcone);
        d->solver_niter = iter + 1;
        if (scale * (oldcost - c->cost) < m->tolerance) break;
    }
done:
    free(c); free(hcone); free(H); free(HL);
}

/* ------------------------------------------------------------------ pipeline */
static void fwd_position(const MjlModel* m, MjlData* d) {
    mjl_kinematics(m, d);
    mjl_crb(m, d);
    mjl_collision(m, d);
    make_constraints(m, d);
    make_impedance(m, d);
}
static void fwd_velocity(const MjlModel* m, MjlData* d) {
    int nv = m->nv;
    reference_accel(m, d);
    for (int i = 0; i < nv; i++) d->qfrc_passive[i] = -m->dof_damping[i] * d->qvel[i];
    for (int j = 0; j < m->njnt; j++)
        if (m->jnt_type[j] != MJL_FREE && m->jnt_stiffness[j] != 0)
            d->qfrc_passive[m->jnt_dofadr[j]] -= m->jnt_stiffness[j] * (d->qpos[m->jnt_qposadr[j]] - m->jnt_springref[j]);
    mjl_rne_bias(m, d);
}
static void fwd_actuation(const MjlModel* m, MjlData* d) {
    memset(d->qfrc_actuator, 0, sizeof(double) * m->nv);
    for (int u = 0; u < m->nu; u++) {
        double c = fmin(m->act_ctrlrange[2 * u + 1], fmax(m->act_ctrlrange[2 * u], d->ctrl[u]));
        d->qfrc_actuator[m->act_dofid[u]] += m->act_kp[u] * (c - d->qpos[m->act_qposid[u]]);
    }
}
static void fwd_acceleration(const MjlModel* m, MjlData* d) {
    int nv = m->nv;
    for (int i = 0; i < nv; i++) {
        d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i];
        d->qacc_smooth[i] = d->qfrc_smooth[i];
    }
    chol_solve(d->qL, d->qacc_smooth, nv);
}
void mjl_forward(const MjlModel* m, MjlData* d) {
    fwd_position(m, d);
    fwd_velocity(m, d);
    fwd_actuation(m, d);
    fwd_acceleration(m, d);
    solve_newton(m, d);
}

static void quat_integrate(double* q, const double* w, double h) {
    double ang = v3norm(w) * h;
    if (ang < MINVAL) return;
    double ax[3] = { w[0], w[1], w[2] };
    v3normalize(ax);
    double s = sin(0.5 * ang), qr[4] = { cos(0.5 * ang), s * ax[0], s * ax[1], s * ax[2] }, qn[4];
    qmul(qn, q, qr);
    memcpy(q, qn, sizeof qn);
    qnormalize(q);
}

void mjl_step(const MjlModel* m, MjlData* d) {
    int nv = m->nv;
    double h = m->timestep;
    mjl_forward(m, d);
    /* semi-implicit Euler with implicit joint damping: (M + h*B) a = f_smooth + f_constraint */
    double qacc[MJL_MAXNV];
    int anydamp = 0;
    for (int i = 0; i < nv; i++) anydamp |= m->dof_damping[i] > 0;
    if (anydamp) {
        double* A = (double*)malloc(sizeof(double) * nv * nv);
        double* L = (double*)malloc(sizeof(double) * nv * nv);
        memcpy(A, d->qM, sizeof(double) * nv * nv);
        for (int i = 0; i < nv; i++) { A[i * nv + i] += h * m->dof_damping[i]; qacc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i]; }
        chol(L, A, nv);
        chol_solve(L, qacc, nv);
        free(A); free(L);
    } else memcpy(qacc, d->qacc, sizeof(double) * nv);
    for (int i = 0; i < nv; i++) d->qvel[i] += h * qacc[i];
    for (int j = 0; j < m->njnt; j++) {
        int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
        if (m->jnt_type[j] == MJL_FREE) {
            for (int k = 0; k < 3; k++) d->qpos[qa + k] += h * d->qvel[da + k];
            quat_integrate(d->qpos + qa + 3, d->qvel + da + 3, h);
        } else d->qpos[qa] += h * d->qvel[da];
    }
    d->time += h;
    memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * nv);
}
void mjl_step_n(const MjlModel* m, MjlData* d, int n) {
    for (int i = 0; i < n; i++) mjl_step(m, d);
}
static double bench_uniform(unsigned long long* s) {          /* xorshift64*, uniform in [-1, 1) */
    unsigned long long x = *s;
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    *s = x;
    return (double)((x * 2685821657736338717ULL) >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}
void mjl_bench_env_steps(const MjlModel* m, MjlData* d, int n, unsigned long long* rng, const double* lo, const double* hi) {
    for (int s = 0; s < n; s++) {
        double a[4];
        for (int k = 0; k < 4; k++) a[k] = bench_uniform(rng);
        for (int k = 0; k < 3; k++) {
            double v = d->mocap_pos[k] + 0.01 * a[k];
            d->mocap_pos[k] = v < lo[k] ? lo[k] : (v > hi[k] ? hi[k] : v);
        }
        d->ctrl[0] = a[3]; d->ctrl[1] = -a[3];
        mjl_step_n(m, d, 5);
        mjl_forward(m, d);
    }
}

/* ------------------------------------------------------------------ accessors for the Python binding */
void mjl_data_info(const MjlData* d, int* out) {
    out[0] = d->ncon; out[1] = d->nefc; out[2] = d->ne; out[3] = d->nl; out[4] = d->solver_niter; out[5] = d->warning_overflow;
    out[6] = d->max_ncon; out[7] = d->max_nefc;
}
void mjl_data_contact(const MjlData* d, int i, int* iv, double* rv) {
    const MjlContact* c = d->contact + i;
    iv[0] = c->geom1; iv[1] = c->geom2; iv[2] = c->dim; iv[3] = c->efc_address;
    rv[0] = c->dist;
    memcpy(rv + 1, c->pos, 3 * sizeof(double));
    memcpy(rv + 4, c->frame, 9 * sizeof(double));
    rv[13] = c->mu;
}
void mjl_data_efc_int(const MjlData* d, int* type, int* id, int* state) {
    for (int i = 0; i < d->nefc; i++) { type[i] = d->efc_type[i]; id[i] = d->efc_id[i]; state[i] = d->efc_state[i]; }
}
