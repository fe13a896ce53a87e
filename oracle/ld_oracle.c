/*
 * ld_oracle.c — CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY:
 * loaded by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
 * product.  Parity is pinned against golden vectors generated from the reference itself
 * (tests/golden/make_golden.py, tests/test_oracle_golden.py).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp; no fast-math — the
 * float32 rounding points below must survive the optimiser).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "ld_oracle.h"

/* ------------------------------------------------------------------------------------------
 * 3-term dot products of the explicit matrix algebra.  numpy hands these tiny products to
 * BLAS, whose summation order and FMA use are implementation details.  The orders below are
 * the ones the fixtures' numpy 2.2.6 / OpenBLAS build uses (found by exhaustive search over
 * association/FMA patterns against np.dot, then pinned by tests/test_oracle_golden.py):
 *   dgemm  C = A.B (both C-contiguous)         fma(a2,b2, fma(a1,b1, a0*b0))
 *   dgemv  with a Fortran-ordered matrix       fma(a2,b2, fma(a1,b1, a0*b0))
 *   dgemv  with a C-ordered matrix             fma(a2,b2, fma(a0,b0, a1*b1))
 * ldo_set_dot_variant(1) switches all three to the plain (a0*b0 + a1*b1) + a2*b2 so that a test
 * can show the pinned orders are what makes the one-step vectors bit-exact.
 * ---------------------------------------------------------------------------------------- */
static int g_dot_variant = 0;
void ldo_set_dot_variant(int v) { g_dot_variant = v; }

static inline double dot3_plain(double a0, double b0, double a1, double b1, double a2, double b2)
{
    return (a0 * b0 + a1 * b1) + a2 * b2;
}

static inline double dot3_gemm(double a0, double b0, double a1, double b1, double a2, double b2)
{
    if (g_dot_variant) return dot3_plain(a0, b0, a1, b1, a2, b2);
    return fma(a2, b2, fma(a1, b1, a0 * b0));
}

#define dot3_gemv_f dot3_gemm          /* Fortran-ordered matrix . vector */

static inline double dot3_gemv_c(double a0, double b0, double a1, double b1, double a2, double b2)
{
    if (g_dot_variant) return dot3_plain(a0, b0, a1, b1, a2, b2);
    return fma(a2, b2, fma(a0, b0, a1 * b1));
}

static void matmul3(const double A[3][3], const double B[3][3], double C[3][3])
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C[i][j] = dot3_gemm(A[i][0], B[0][j], A[i][1], B[1][j], A[i][2], B[2][j]);
}

/* QuadcopterPhysics.calculate_accelerations, physical_model.py:75-122, with every float32
 * rounding point of the reference kept (marked (1)-(4)). */
/* `lab` = the reference's lin_acc_drone_2_lab argument: 0 returns the body-frame pair of :114-115 before Rot^T and
 * Tr are applied (:117-120).  rot_out (may be NULL): self.Rot as the call leaves it (:90), row-major. */
static void accel_faithful_frames(const ld_sim_params *p, const double ang[3], const double u[4],
                                  double lin[3], double rot[3], double *rot_out, int lab)
{
    const double roll = ang[0], pitch = ang[1], yaw = ang[2];
    const double cr = cos(roll), sr = sin(roll), cp = cos(pitch), sp = sin(pitch);
    const double cy = cos(yaw), sy = sin(yaw);
    /* geometry.py:23-43 */
    const double Rx[3][3] = {{1, 0, 0}, {0, cr, -1 * sr}, {0, sr, cr}};
    const double Ry[3][3] = {{cp, 0, sp}, {0, 1, 0}, {-1 * sp, 0, cp}};
    const double Rz[3][3] = {{cy, -1 * sy, 0}, {sy, cy, 0}, {0, 0, 1}};
    double YX[3][3], B[3][3];                 /* B = Rz Ry Rx; Rot = B^T */
    matmul3(Ry, Rx, YX);                      /* np.dot(R_z, np.dot(R_y, R_x)): dgemm */
    matmul3(Rz, YX, B);

    double T[4], R[4];                        /* :93-94 */
    for (int i = 0; i < 4; ++i) { T[i] = u[i] * p->coef_force; R[i] = u[i] * p->coef_moment; }
    const double sumT = ((T[0] + T[1]) + T[2]) + T[3];
    float F[3] = {0.0f, 0.0f, (float)sumT};                        /* (2) :98 */
    const float G32 = (float)(-1 * p->gravity_force);              /* (3) :101 */
    float w32[3];
    for (int i = 0; i < 3; ++i) w32[i] = (float)p->wind[i] * (float)p->coef_wind;   /* :103 */
    for (int i = 0; i < 3; ++i) {             /* Rot[i][k] = B[k][i] */
        const double g = dot3_gemv_f(B[0][i], 0.0, B[1][i], 0.0, B[2][i], (double)G32);
        const double w = dot3_gemv_f(B[0][i], (double)w32[0], B[1][i], (double)w32[1], B[2][i], (double)w32[2]);
        F[i] = (float)((double)F[i] + (g + w));                    /* (2) :104 in-place += */
    }
    /* :107-111 */
    const double L = p->radius_motor * (((T[0] + T[3]) - T[1]) - T[2]);
    const double M = p->radius_motor * (((T[2] + T[3]) - T[0]) - T[1]);
    const double N = (sqrt(2.0) * p->radius_motor) * (((R[1] + R[3]) - R[0]) - R[2]);
    const double mom[3] = {L + p->moments_payload[0], M + p->moments_payload[1], N + p->moments_payload[2]};
    /* :114-115 */
    const float m32 = (float)p->mass_total;
    const float a32[3] = {F[0] / m32, F[1] / m32, F[2] / m32};      /* (4) stays float32 */
    const double b[3] = {mom[0] * p->inv_inertia[0], mom[1] * p->inv_inertia[1], mom[2] * p->inv_inertia[2]};
    if (rot_out)
        for (int i = 0; i < 3; ++i)
            for (int k = 0; k < 3; ++k) rot_out[3 * i + k] = B[k][i];
    if (!lab) {                               /* :117 not taken: the float32 forces / m and the float64 moments / I */
        for (int i = 0; i < 3; ++i) { lin[i] = (double)a32[i]; rot[i] = b[i]; }
        return;
    }
    /* :118  Rot^T . lin_acc */
    for (int i = 0; i < 3; ++i)
        lin[i] = dot3_gemv_c(B[i][0], (double)a32[0], B[i][1], (double)a32[1], B[i][2], (double)a32[2]);
    /* :119-120, geometry.py:62-69: entries computed in float64, stored float32 (1) */
    const double tp = tan(pitch);
    const float Tr[3][3] = {{1.0f, (float)(sr * tp), (float)(cr * tp)},
                            {0.0f, (float)cr, (float)(-sr)},
                            {0.0f, (float)(sr / cp), (float)(cr / cp)}};
    for (int i = 0; i < 3; ++i)
        rot[i] = dot3_gemv_c((double)Tr[i][0], b[0], (double)Tr[i][1], b[1], (double)Tr[i][2], b[2]);
}

static void accel_faithful(const ld_sim_params *p, const double ang[3], const double u[4],
                           double lin[3], double rot[3])
{
    accel_faithful_frames(p, ang, u, lin, rot, NULL, 1);
}

/* ---- float64 instantiation (faithful + closed) ---- */
#define REAL double
#define SUF(name) name##_f64
#define R_SIN sin
#define R_COS cos
#define R_EXP exp
#define R_SQRT sqrt
#define R_FABS fabs
#define R_FMA fma
#define HAS_FAITHFUL 1
#include "ld_oracle_generic.inc"
#undef REAL
#undef SUF
#undef R_SIN
#undef R_COS
#undef R_EXP
#undef R_SQRT
#undef R_FABS
#undef R_FMA
#undef HAS_FAITHFUL

/* ---- float32 instantiation (closed form; the HIP f32 kernel's arithmetic) ---- */
#define REAL float
#define SUF(name) name##_f32
#define R_SIN sinf
#define R_COS cosf
#define R_EXP expf
#define R_SQRT sqrtf
#define R_FABS fabsf
#define R_FMA fmaf
#define HAS_FAITHFUL 0
#include "ld_oracle_generic.inc"
#undef REAL
#undef SUF
#undef R_SIN
#undef R_COS
#undef R_EXP
#undef R_SQRT
#undef R_FABS
#undef R_FMA
#undef HAS_FAITHFUL

static int check_spec(const ld_mlp_spec *spec)
{
    if (!spec) return LD_ERR_NULL;
    if (spec->n_hidden < 1 || spec->n_hidden > LD_MAX_HIDDEN_LAYERS) return LD_ERR_SPEC;
    for (int i = 0; i < spec->n_hidden; ++i)
        if (spec->hidden[i] < 1 || spec->hidden[i] > LD_MAX_WIDTH) return LD_ERR_SPEC;
    return LD_OK;
}

static int64_t param_count(const ld_mlp_spec *spec)
{
    int64_t P = 0;
    int n_in = LD_N_INPUTS;
    for (int i = 0; i < spec->n_hidden; ++i) { P += (int64_t)spec->hidden[i] * (n_in + 1); n_in = spec->hidden[i]; }
    return P + (int64_t)LD_N_OUTPUTS * (n_in + 1);
}

int ldo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int ldo_accelerations(const ld_sim_params *sim, const double angle[3], const double thrust[4],
                      double lin_acc[3], double rot_acc[3], int mode)
{
    if (!sim || !angle || !thrust || !lin_acc || !rot_acc) return LD_ERR_NULL;
    if (mode == LDO_CLOSED_F32) {
        float a[3] = {(float)angle[0], (float)angle[1], (float)angle[2]};
        float u[4] = {(float)thrust[0], (float)thrust[1], (float)thrust[2], (float)thrust[3]};
        float l[3], r[3];
        accel_f32(sim, a, u, l, r, 0);
        for (int i = 0; i < 3; ++i) { lin_acc[i] = l[i]; rot_acc[i] = r[i]; }
    } else {
        accel_f64(sim, angle, thrust, lin_acc, rot_acc, mode == LDO_FAITHFUL_F64);
    }
    return LD_OK;
}

int ldo_accelerations_frames(const ld_sim_params *sim, const double angle[3], const double thrust[4],
                             double lin_acc[3], double rot_acc[3], double rot_matrix[9], int lab_frame)
{
    if (!sim || !angle || !thrust || !lin_acc || !rot_acc) return LD_ERR_NULL;
    accel_faithful_frames(sim, angle, thrust, lin_acc, rot_acc, rot_matrix, lab_frame);
    return LD_OK;
}

int ldo_measure(const ld_sim_params *sim, const ld_episode_params *ep, int64_t n, double *state,
                double *acc, const double *thrust, int64_t *reward, int32_t *crashed, int mode)
{
    if (!sim || !ep || !state || !acc || !thrust || !reward || !crashed) return LD_ERR_NULL;
    for (int64_t i = 0; i < n; ++i) {
        if (crashed[i]) continue;
        if (mode == LDO_CLOSED_F32) {
            float s[12], a[6], u[4];
            for (int k = 0; k < 12; ++k) s[k] = (float)state[i * 12 + k];
            for (int k = 0; k < 6; ++k) a[k] = (float)acc[i * 6 + k];
            for (int k = 0; k < 4; ++k) u[k] = (float)thrust[i * 4 + k];
            crashed[i] = measure_f32(sim, ep, NULL, s, a, u, &reward[i], 0);
            for (int k = 0; k < 12; ++k) state[i * 12 + k] = s[k];
            for (int k = 0; k < 6; ++k) acc[i * 6 + k] = a[k];
        } else {
            crashed[i] = measure_f64(sim, ep, NULL, &state[i * 12], &acc[i * 6], &thrust[i * 4],
                                     &reward[i], mode == LDO_FAITHFUL_F64);
        }
    }
    return LD_OK;
}

int ldo_mlp_forward(const ld_mlp_spec *spec, const double *genes, int64_t n_vec, const double *x,
                    double *y, int mode)
{
    int rc = check_spec(spec);
    if (rc) return rc;
    if (!genes || !x || !y) return LD_ERR_NULL;
    const int64_t P = param_count(spec);
    if (mode == LDO_CLOSED_F32) {
        float *g = (float *)malloc(sizeof(float) * (size_t)P);
        for (int64_t k = 0; k < P; ++k) g[k] = (float)genes[k];
        for (int64_t v = 0; v < n_vec; ++v) {
            float xi[LD_N_INPUTS], yo[LD_N_OUTPUTS];
            for (int k = 0; k < LD_N_INPUTS; ++k) xi[k] = (float)x[v * LD_N_INPUTS + k];
            mlp_f32(spec, g, xi, yo, 0);
            for (int k = 0; k < LD_N_OUTPUTS; ++k) y[v * LD_N_OUTPUTS + k] = yo[k];
        }
        free(g);
    } else {
        for (int64_t v = 0; v < n_vec; ++v)
            mlp_f64(spec, genes, &x[v * LD_N_INPUTS], &y[v * LD_N_OUTPUTS], mode == LDO_FAITHFUL_F64);
    }
    return LD_OK;
}

int ldo_fly(const ld_sim_params *sim, const ld_episode_params *ep, const ld_mlp_spec *spec,
            const double *coins, const double *walls, const double *genes, int64_t n,
            int32_t timesteps, double *state, double *acc, double *thrust, int64_t *reward,
            int32_t *coins_collected, int32_t *flight_time, int32_t *crashed, double *distance,
            double *trace_state, int64_t *trace_reward, int32_t *trace_coins, int n_threads, int mode)
{
    int rc = check_spec(spec);
    if (rc) return rc;
    if (!sim || !ep || !coins || !genes || !state || !acc || !thrust || !reward ||
        !coins_collected || !flight_time || !crashed || !distance) return LD_ERR_NULL;
    if (n < 0 || timesteps < 0 || ep->n_coins < 1 || ep->n_coins > LD_MAX_COINS ||
        ep->n_walls < 0 || ep->n_walls > LD_MAX_WALLS) return LD_ERR_SIZE;
    const int64_t P = param_count(spec);
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 16) num_threads(n_threads)
    for (int64_t i = 0; i < n; ++i) {
        if (mode == LDO_CLOSED_F32)
            fly_one_f32(sim, ep, spec, coins, walls, genes + i * P, P, timesteps, state + i * 12,
                        acc + i * 6, thrust + i * 4, &reward[i], &coins_collected[i], &flight_time[i],
                        &crashed[i], &distance[i], trace_state, trace_reward, trace_coins, n, i, 0);
        else
            fly_one_f64(sim, ep, spec, coins, walls, genes + i * P, P, timesteps, state + i * 12,
                        acc + i * 6, thrust + i * 4, &reward[i], &coins_collected[i], &flight_time[i],
                        &crashed[i], &distance[i], trace_state, trace_reward, trace_coins, n, i,
                        mode == LDO_FAITHFUL_F64);
    }
    return LD_OK;
}

int ldo_open_loop(const ld_sim_params *sim, int32_t timesteps, const double *thrust_seq,
                  double state[12], double acc[6], double *trace_state, int mode)
{
    if (!sim || !thrust_seq || !state || !acc) return LD_ERR_NULL;
    ld_episode_params ep;
    memset(&ep, 0, sizeof ep);
    ep.z_floor = -INFINITY; ep.z_ceiling = INFINITY;
    ep.angle_limit = INFINITY; ep.angle_vel_limit = INFINITY;
    int64_t reward = 0;
    if (mode == LDO_CLOSED_F32) {
        float s[12], a[6];
        for (int k = 0; k < 12; ++k) s[k] = (float)state[k];
        for (int k = 0; k < 6; ++k) a[k] = (float)acc[k];
        for (int32_t t = 0; t < timesteps; ++t) {
            float u[4];
            for (int k = 0; k < 4; ++k) u[k] = (float)thrust_seq[(size_t)t * 4 + k];
            measure_f32(sim, &ep, NULL, s, a, u, &reward, 0);
            if (trace_state) for (int k = 0; k < 12; ++k) trace_state[(size_t)t * 12 + k] = s[k];
        }
        for (int k = 0; k < 12; ++k) state[k] = s[k];
        for (int k = 0; k < 6; ++k) acc[k] = a[k];
    } else {
        for (int32_t t = 0; t < timesteps; ++t) {
            measure_f64(sim, &ep, NULL, state, acc, thrust_seq + (size_t)t * 4, &reward, mode == LDO_FAITHFUL_F64);
            if (trace_state) memcpy(trace_state + (size_t)t * 12, state, sizeof(double) * 12);
        }
    }
    return LD_OK;
}

/* ------------------------------------------------------------------------------------------
 * Config 1: cascaded PID controller.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    double kp, ki, kd, dt, integral, integral_range, set_value, prev_error, prev_cv, lo, hi;
} pid_t_;

static void pid_init(pid_t_ *c, double kp, double ki, double kd, double dt, double sp, double lo, double hi)
{
    c->kp = kp; c->ki = ki; c->kd = kd; c->dt = dt; c->integral = 0; c->integral_range = 2;
    c->set_value = sp; c->prev_error = 0; c->prev_cv = 0; c->lo = lo; c->hi = hi;
}

/* PID.calculate_error_sign, drone/controller/pid.py:77-106 (+ accumulate_error :132-134,
 * check_output :136-142). */
static double pid_calculate(pid_t_ *c, double cv)
{
    const double error = c->set_value - cv;
    if (c->prev_error * error < 0) c->integral = 0;
    if (fabs(error) < c->integral_range) c->integral += error;
    const double diff = (c->prev_cv != 0) ? cv - c->prev_cv : 0.0;
    double out = c->kp * error + c->ki * c->integral * c->dt - c->kd * diff / c->dt;
    c->prev_cv = cv;
    if (error != 0) c->prev_error = error;
    if (out < c->lo) out = c->lo; else if (out > c->hi) out = c->hi;
    return out;
}

static inline double clamp01(double v) { return v > 1 ? 1 : (v < 0 ? 0 : v); }

int ldo_pid_episode(const ld_sim_params *sim, const double gains[18], const double *coins,
                    int32_t n_coins, double coin_radius, int32_t coin_value, int32_t timesteps,
                    double *trace_state, double *trace_thrust, int64_t *trace_score,
                    int32_t *trace_target, int32_t *steps_run, int64_t *final_score)
{
    if (!sim || !gains || !coins || !steps_run || !final_score) return LD_ERR_NULL;
    if (n_coins < 1 || timesteps < 0) return LD_ERR_SIZE;
    const double kp_x = gains[0], kp_y = gains[1], kp_z = gains[2];
    const double ki_x = gains[3], ki_y = gains[4];            /* gains[5] = ki_z is unused: */
    const double kd_x = gains[6], kd_y = gains[7], kd_z = gains[8];
    const double kp_rol = gains[9], kp_pit = gains[10], kp_yaw = gains[11];
    const double ki_rol = gains[12], ki_pit = gains[13], ki_yaw = gains[14];
    const double kd_rol = gains[15], kd_pit = gains[16], kd_yaw = gains[17];
    const double dt = sim->dt, q = M_PI / 4;
    pid_t_ rot[3], lin[3];
    pid_init(&rot[0], kp_rol, ki_rol, kd_rol, dt, 0, -q, q);          /* collect_coins.py:125-156 */
    pid_init(&rot[1], kp_pit, ki_pit, kd_pit, dt, 0, -q, q);
    pid_init(&rot[2], kp_yaw, ki_yaw, kd_yaw, dt, 0, -M_PI / 6, M_PI / 6);
    pid_init(&lin[0], kp_x, ki_x, kd_x, dt, 10, -q, q);               /* :158-188 */
    pid_init(&lin[1], kp_y, ki_y, kd_y, dt, 10, -q, q);
    pid_init(&lin[2], kp_z, ki_y, kd_z, dt, 10, -1, 1);               /* :182 passes ki_y (sic) */

    double s[12] = {0, 0, 10, 0, 0, 0, 0, 0, 0, 0, 0, 0};             /* parameters.py:47-52 */
    double acc[6] = {0, 0, 0, 0, 0, 0};
    double u[4] = {0, 0, 0, 0};
    ld_episode_params ep;
    memset(&ep, 0, sizeof ep);
    ep.z_floor = -INFINITY; ep.z_ceiling = INFINITY; ep.angle_limit = INFINITY; ep.angle_vel_limit = INFINITY;
    int64_t score = timesteps, dummy = 0;
    int32_t target = 0, new_target = 0, t = 0;
    for (; t < timesteps; ++t) {                                      /* :278-364 */
        score -= 1;
        {
            const double *c = coins + (size_t)target * 3;
            const double dx = c[0] - s[0], dy = c[1] - s[1], dz = c[2] - s[2];
            if (sqrt((dx * dx + dy * dy) + dz * dz) < coin_radius) { score += coin_value; new_target = target + 1; }
        }
        if (new_target >= n_coins) break;
        for (int i = 0; i < 3; ++i) { rot[i].set_value = 0; lin[i].set_value = coins[(size_t)new_target * 3 + i]; }
        target = new_target;
        measure_f64(sim, &ep, NULL, s, acc, u, &dummy, 1);
        /* onboard_computer.py:50-93 */
        double lo[3], ro[3];
        for (int i = 0; i < 3; ++i) lo[i] = pid_calculate(&lin[i], s[i]);
        rot[0].set_value = -lo[1];
        rot[1].set_value = lo[0];
        for (int i = 0; i < 3; ++i) ro[i] = pid_calculate(&rot[i], s[6 + i]);
        /* onboard_computer.py:95-118 */
        static const double mix[4][3] = {{1, -1, -1}, {-1, -1, 1}, {-1, 1, -1}, {1, 1, 1}};
        const double base = sim->gravity_force / (4 * sim->coef_force) + lo[2];
        const double cc = cos(s[6]) * cos(s[7]);
        const double ratio = (cc != 0) ? 1 / cc : 1;
        const double base_t = clamp01(base * ratio);
        for (int i = 0; i < 4; ++i) {
            const double want = dot3_gemv_c(mix[i][0], ro[0], mix[i][1], ro[1], mix[i][2], ro[2]);
            u[i] = clamp01(base_t + want);
        }
        if (trace_state) memcpy(trace_state + (size_t)t * 12, s, sizeof s);
        if (trace_thrust) memcpy(trace_thrust + (size_t)t * 4, u, sizeof u);
        if (trace_score) trace_score[t] = score;
        if (trace_target) trace_target[t] = target;
    }
    *steps_run = t;
    *final_score = score;
    return LD_OK;
}
