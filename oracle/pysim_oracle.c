/*
 * pysim_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar fp64 CPU restatement of the reference algorithm for batched rollouts of Pyrado's
 * standalone-Python simulations.  Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of
 * bench.py may load this file's library; the product (simurlacra_amd + libpysimhip.so) never does.
 *
 * PARITY STATUS: the reference's Python could not be executed in the build environment.
 *  - OneMassOscillatorSim is pinned by a trajectory the reference itself produced and ships
 *    (Pyrado/data/perma/time_series/omo_traj_gt.npy, made by Pyrado/scripts/sandbox/time_series/gen_ts_data.py:22-44):
 *    this file reproduces its 501 samples bit for bit through orc_reset / orc_step and through its rollout loop
 *    (tests/test_reference_golden.py) -- dynamics, the order of operations of a step, and the rollout loop.
 *  - For QQube, QCartPole, QBallBalancer and BallOnBeam neither the reference's tests nor its data files hold numbers
 *    for the per-step dynamics: these are "parity unpinned", pinned only by restating the cited source lines expression
 *    by expression (same operand order, no FMA contraction: build with -ffp-contract=off), by a second, independent
 *    NumPy restatement (oracle/pysim_oracle_np.py) that must agree to ~1 ulp, and by hand-computed one-step numbers
 *    (tests/anchors.py).
 *  - The parts the reference's tests pin with numbers are checked against those numbers in tests/test_oracle.py and
 *    tests/test_anchors.py: radial-wrap reward (Pyrado/tests/test_tasks.py:44-61), FinalRewTask yields once (:181-188),
 *    polar sampling (Pyrado/tests/test_spaces.py:145-154).
 *
 * It reads the same descriptors as the product ABI (include/pysimhip.h) with HOST pointers.
 * Paths below are relative to /root/reference/Pyrado/pyrado/.
 */
#include "../include/pysimhip.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_PI 3.141592653589793

typedef struct orc_env {
    int kind, S, O, A, P, simplified;
    double dp[PSH_MAX_DP];
    double c[8];                      /* derived constants, meaning per kind */
    double s_lo[PSH_MAX_STATE], s_hi[PSH_MAX_STATE];
    double a_lo[PSH_MAX_ACT], a_hi[PSH_MAX_ACT];
    double c_max;                     /* ScaledExpQuadrErrRewFcn scale */
    double s[PSH_MAX_STATE];
    double plate[2];                  /* QBB hidden plate angles */
    int curr_step, yielded;
} orc_env;

int orc_env_dims(int kind, int32_t d[5]) {
    switch (kind) {
    case PSH_ENV_QQ: d[0] = 4; d[1] = 6; d[2] = 1; d[3] = 10; d[4] = 4; return 0;
    case PSH_ENV_QCP_SU:
    case PSH_ENV_QCP_ST: d[0] = 4; d[1] = 5; d[2] = 1; d[3] = 14; d[4] = 4; return 0;
    case PSH_ENV_QBB: d[0] = 8; d[1] = 8; d[2] = 2; d[3] = 20; d[4] = 4; return 0;
    case PSH_ENV_BOB: d[0] = 4; d[1] = 4; d[2] = 1; d[3] = 8; d[4] = 4; return 0;
    case PSH_ENV_OMO: d[0] = 2; d[1] = 2; d[2] = 1; d[3] = 3; d[4] = 2; return 0;
    case PSH_ENV_PEND: d[0] = 2; d[1] = 3; d[2] = 1; d[3] = 5; d[4] = 2; return 0;
    case PSH_ENV_QQ_ST: d[0] = 4; d[1] = 4; d[2] = 1; d[3] = 10; d[4] = 4; return 0;
    }
    return -1;
}

/* ---- _calc_constants + _create_spaces ---------------------------------------------------------- */
static void calc_constants_and_spaces(orc_env* e) {
    const double* p = e->dp;
    const double inf = INFINITY;
    switch (e->kind) {
    case PSH_ENV_QQ_ST:
    case PSH_ENV_QQ: { /* pysim/quanser_qube.py:73-90 (constants), :20-36 (spaces); QQubeStabSim :230-255 */
        double g = p[0], Mr = p[3], Lr = p[4], Mp = p[6], Lp = p[7];
        double Jr = Mr * (Lr * Lr) / 12;
        double Jp = Mp * (Lp * Lp) / 12;
        e->c[0] = Jr + Mp * (Lr * Lr);
        e->c[1] = 0.25 * Mp * (Lp * Lp);
        e->c[2] = 0.5 * Mp * Lp * Lr;
        e->c[3] = Jp + e->c[1];
        e->c[4] = 0.5 * Mp * Lp * g;
        double ms[4] = {115. / 180 * ORC_PI, 4 * ORC_PI, 20 * ORC_PI, 20 * ORC_PI};
        for (int i = 0; i < 4; ++i) { e->s_lo[i] = -ms[i]; e->s_hi[i] = ms[i]; }
        if (e->kind == PSH_ENV_QQ_ST) {
            const double stab_thold = 15 / 180. * ORC_PI;
            e->s_lo[0] = -120. / 180 * ORC_PI; e->s_hi[0] = 120. / 180 * ORC_PI;
            e->s_lo[1] = ORC_PI - stab_thold; e->s_hi[1] = ORC_PI + stab_thold;
        }
        e->a_lo[0] = -p[9]; e->a_hi[0] = p[9];
        break;
    }
    case PSH_ENV_PEND: { /* pysim/pendulum.py:17-28 */
        e->s_lo[0] = -4 * ORC_PI; e->s_hi[0] = 4 * ORC_PI; e->s_lo[1] = -4 * ORC_PI; e->s_hi[1] = 4 * ORC_PI;
        e->a_lo[0] = -p[4]; e->a_hi[0] = p[4];
        break;
    }
    case PSH_ENV_QCP_SU:
    case PSH_ENV_QCP_ST: { /* pysim/quanser_cartpole.py:83-93, :41-52, :252-269, :307-318 */
        double m_cart = p[1], l_rail = p[2], eta_g = p[4], K_g = p[5], J_m = p[6], r_mp = p[7];
        double m_pole = p[12], l_pole = p[13];
        e->c[0] = (l_pole * l_pole) * m_pole / 3.;                       /* J_pole */
        e->c[1] = m_cart + (eta_g * (K_g * K_g) * J_m) / (r_mp * r_mp);  /* J_eq */
        const double x_buffer = 0.05;
        if (e->kind == PSH_ENV_QCP_SU) {
            e->s_hi[0] = +l_rail / 2. - x_buffer; e->s_lo[0] = -l_rail / 2. + x_buffer;
            e->s_hi[1] = +4 * ORC_PI; e->s_lo[1] = -4 * ORC_PI;
            e->s_hi[2] = inf; e->s_lo[2] = -inf;
            e->s_hi[3] = inf; e->s_lo[3] = -inf;
        } else {
            const double stab_thold = 15 / 180. * ORC_PI;
            e->s_lo[0] = -l_rail / 2. + x_buffer; e->s_hi[0] = +l_rail / 2. - x_buffer;
            e->s_lo[1] = ORC_PI - stab_thold; e->s_hi[1] = ORC_PI + stab_thold;
            e->s_lo[2] = -l_rail; e->s_hi[2] = +l_rail;
            e->s_lo[3] = -2 * ORC_PI; e->s_hi[3] = +2 * ORC_PI;
        }
        e->a_lo[0] = -8.; e->a_hi[0] = 8.;
        break;
    }
    case PSH_ENV_QBB: { /* pysim/quanser_ball_balancer.py:154-173, :60-76 */
        double m_ball = p[1], r_ball = p[2], l_plate = p[3], r_arm = p[4], K_g = p[5], eta_g = p[6];
        double J_l = p[7], J_m = p[8], k_m = p[9], R_m = p[10], eta_m = p[11], B_eq = p[12];
        e->c[0] = 2. / 5 * m_ball * (r_ball * r_ball);                       /* J_ball */
        e->c[1] = eta_g * (K_g * K_g) * J_m + J_l;                           /* J_eq */
        e->c[2] = 2. * r_arm / l_plate;                                      /* c_kin */
        e->c[3] = eta_g * K_g * eta_m * k_m / R_m;                           /* A_m */
        e->c[4] = eta_g * (K_g * K_g) * eta_m * (k_m * k_m) / R_m + B_eq;    /* B_eq_v */
        e->c[5] = m_ball * (r_ball * r_ball) + e->c[0];                      /* zeta */
        double ms[8] = {ORC_PI / 4., ORC_PI / 4., l_plate / 2., l_plate / 2., 5 * ORC_PI, 5 * ORC_PI, 0.5, 0.5};
        for (int i = 0; i < 8; ++i) { e->s_lo[i] = -ms[i]; e->s_hi[i] = ms[i]; }
        for (int i = 0; i < 2; ++i) { e->a_lo[i] = -3.0; e->a_hi[i] = 3.0; }
        break;
    }
    case PSH_ENV_BOB: { /* pysim/ball_on_beam.py:58-67, :21-45 */
        double g = p[0], m_ball = p[1], r_ball = p[2], m_beam = p[3], l_beam = p[4], d_beam = p[5];
        e->c[0] = 2. / 5 * m_ball * (r_ball * r_ball);                            /* J_ball */
        e->c[1] = 1. / 12 * m_beam * ((l_beam * l_beam) + (d_beam * d_beam));     /* J_beam */
        e->c[2] = m_ball + e->c[0] / (r_ball * r_ball);                           /* zeta_ball */
        double ms[4] = {l_beam / 2., ORC_PI / 4., 10., ORC_PI};
        for (int i = 0; i < 4; ++i) { e->s_lo[i] = -ms[i]; e->s_hi[i] = ms[i]; }
        double max_act = l_beam / 2. * g * 3.;
        e->a_lo[0] = -max_act; e->a_hi[0] = max_act;
        break;
    }
    case PSH_ENV_OMO: { /* pysim/one_mass_oscillator.py:59-65, :28-41 */
        double m = p[0], k = p[1], d = p[2];
        e->c[0] = sqrt(k / m);                 /* omega */
        e->c[1] = d / (2. * sqrt(m * k));      /* zeta */
        e->s_lo[0] = -1.; e->s_hi[0] = 1.; e->s_lo[1] = -10.; e->s_hi[1] = 10.;
        double max_act = 1. * k;
        e->a_lo[0] = -max_act; e->a_hi[0] = max_act;
        break;
    }
    }
}

/* weighted quadratic cost, tasks/reward_functions.py:167-175 */
static double quadr_cost(const psh_task* t, int S, int A, const double* es, const double* ea) {
    double qe[PSH_MAX_STATE], re[PSH_MAX_ACT];
    for (int i = 0; i < S; ++i) {
        double acc = 0.;
        for (int j = 0; j < S; ++j) acc += t->Q[i * S + j] * es[j];
        qe[i] = acc;
    }
    double cs = 0.;
    for (int i = 0; i < S; ++i) cs += es[i] * qe[i];
    for (int i = 0; i < A; ++i) {
        double acc = 0.;
        for (int j = 0; j < A; ++j) acc += t->R[i * A + j] * ea[j];
        re[i] = acc;
    }
    double ca = 0.;
    for (int i = 0; i < A; ++i) ca += ea[i] * re[i];
    return cs + ca;
}

/* ScaledExpQuadrErrRewFcn.reset, tasks/reward_functions.py:249-262; bound_abs_up spaces/base.py:31-34 */
static void task_reset(orc_env* e, const psh_task* t) {
    e->yielded = 0;
    e->c_max = 0.;
    if (t->rew_kind == PSH_REW_SCALED_EXP_QUADR) {
        double om[PSH_MAX_STATE], am[PSH_MAX_ACT];
        for (int i = 0; i < e->S; ++i) om[i] = fmax(fabs(e->s_lo[i]), fabs(e->s_hi[i]));
        for (int i = 0; i < e->A; ++i) am[i] = fmax(fabs(e->a_lo[i]), fabs(e->a_hi[i]));
        double max_cost = quadr_cost(t, e->S, e->A, om, am);
        e->c_max = -1.0 * log(t->min_rew) / max_cost;
    }
}

/* DesStateTask.step_rew tasks/desired_state.py:82-85; RadiallySymmDesStateTask.step_rew :119-128 */
static double step_rew(const orc_env* e, const psh_task* t, const double* state, const double* act) {
    double es[PSH_MAX_STATE], ea[PSH_MAX_ACT];
    for (int i = 0; i < e->S; ++i) es[i] = t->state_des[i] - state[i];
    if (t->radial) {
        for (int i = 0; i < e->S; ++i)
            if (t->radial_mask & (1 << i)) es[i] = fmod(es[i], t->mod[i]);
        for (int i = 0; i < e->S; ++i)
            if (es[i] > ORC_PI) es[i] = 2 * ORC_PI - es[i];
        for (int i = 0; i < e->S; ++i)
            if (es[i] < -ORC_PI) es[i] = -2 * ORC_PI - es[i];
    }
    for (int i = 0; i < e->A; ++i) ea[i] = -act[i];
    switch (t->rew_kind) {
    case PSH_REW_QUADR: return -quadr_cost(t, e->S, e->A, es, ea);
    case PSH_REW_EXP_QUADR: return exp(-quadr_cost(t, e->S, e->A, es, ea));
    case PSH_REW_SCALED_EXP_QUADR: return exp(-e->c_max * quadr_cost(t, e->S, e->A, es, ea));
    case PSH_REW_SWINGUP: { /* tasks/reward_functions.py:299-303 */
        double eth = es[t->idx_th];
        double cost_pole = t->swingup[3] * (eth * eth) + log(eth * eth + t->swingup[4]);
        double cost_cart = fabs(es[t->idx_x]);
        double cost_act = ea[0] * ea[0];
        double dot = (-t->swingup[0]) * cost_pole + (-t->swingup[1]) * cost_cart;
        dot = dot + (-t->swingup[2]) * cost_act;
        return dot + 10;
    }
    }
    return 0.;
}

/* BoxSpace.contains spaces/box.py:107-131: inclusive bounds; NaN is reported through *nan */
static int contains(const double* x, const double* lo, const double* hi, int n, int* nan) {
    int ok = 1;
    for (int i = 0; i < n; ++i) {
        if (isnan(x[i])) *nan = 1;
        if (!(x[i] >= lo[i] && x[i] <= hi[i])) ok = 0;
    }
    return ok;
}

/* FinalRewTask.compute_final_rew tasks/final_reward.py:70-187 (success never happens:
 * tasks/utils.py:4-6, so only the failure branches can be non-zero) */
static double final_rew(orc_env* e, const psh_task* t, int remaining) {
    if (!(t->final_mode & PSH_FIN_ENABLED)) return 0.;
    if (e->yielded) return 0.;
    e->yielded = 1;
    int nan = 0;
    int failed = !contains(e->s, e->s_lo, e->s_hi, e->S, &nan);
    if (!failed) return 0.;
    int ap = (t->final_mode & PSH_FIN_ALWAYS_POSITIVE) != 0, an = (t->final_mode & PSH_FIN_ALWAYS_NEGATIVE) != 0;
    int sd = (t->final_mode & PSH_FIN_STATE_DEPENDENT) != 0, td = (t->final_mode & PSH_FIN_TIME_DEPENDENT) != 0;
    double zero_act[PSH_MAX_ACT] = {0., 0.};
    if (!ap && !an && !sd && !td) return -1. * fabs(t->final_factor);
    if (ap && !an && !sd && !td) return 0.;
    if (!ap && an && !sd && !td) return -1. * fabs(t->final_factor);
    if (ap && !an && sd && !td) return 0.;
    if (!ap && an && sd && !td) return -1. * t->final_factor * fabs(step_rew(e, t, e->s, zero_act));
    if (!ap && !an && sd && !td) return -1. * t->final_factor * fabs(step_rew(e, t, e->s, zero_act));
    if (!ap && !an && sd && td) return -1. * remaining * fabs(step_rew(e, t, e->s, zero_act));
    if (!ap && !an && !sd && td) return -1. * remaining;
    if (ap && !an && !sd && td) return 0.;
    if (!ap && an && !sd && td) return -1. * remaining;
    return NAN; /* the reference raises NotImplementedError for other combinations */
}

/* QBallBalancerKin.__call__ pysim/quanser_ball_balancer.py:413-480; torch.optim.SGD(lr=0.01,
 * momentum=0.9) with the autograd chain of _loss_fcn written out */
double orc_qbb_ik_n(double th, double r, double l_plate, int n_iter, double* tip_out) {
    const double l = l_plate / 2., d = 0.10;
    double t0 = r, t1 = l, b0 = 0., b1 = 0.;
    const double cx = r * cos(th), cy = r * sin(th);
    for (int it = 0; it < n_iter; ++it) {
        double dx1 = t0 - cx, dy1 = t1 - cy;
        double rod = sqrt(dx1 * dx1 + dy1 * dy1);
        double dx2 = t0 - r - l, dy2 = t1 - d;
        double half = sqrt(dx2 * dx2 + dy2 * dy2);
        double g_rod = 2 * (rod - d) / (2 * rod);     /* d(loss)/d(rod^2-argument) through pow and sqrt */
        double g_half = 2 * (half - l) / (2 * half);
        double g0 = g_rod * (2 * dx1) + g_half * (2 * dx2);
        double g1 = g_rod * (2 * dy1) + g_half * (2 * dy2);
        if (it == 0) { b0 = g0; b1 = g1; } else { b0 = 0.9 * b0 + g0; b1 = 0.9 * b1 + g1; }
        t0 = t0 - 0.01 * b0;
        t1 = t1 - 0.01 * b1;
    }
    if (tip_out) { tip_out[0] = t0; tip_out[1] = t1; }
    return ORC_PI / 2. - atan2(r + l - t0, t1 - d);
}

/* num_opt_iter = 100 is the constructor default nobody overrides (pysim/quanser_ball_balancer.py:385) */
double orc_qbb_ik(double th, double r, double l_plate) { return orc_qbb_ik_n(th, r, l_plate, 100, 0); }

/* SimPyEnv.reset pysim/base.py:119-156 (+ QBB reset pysim/quanser_ball_balancer.py:175-195) */
static void env_reset(orc_env* e, const psh_task* t, const double* dp, const double* init, int init_dim) {
    for (int i = 0; i < e->P; ++i) e->dp[i] = dp[i];
    calc_constants_and_spaces(e);
    task_reset(e, t);
    e->curr_step = 0;
    for (int i = 0; i < PSH_MAX_STATE; ++i) e->s[i] = 0.;
    if (e->kind == PSH_ENV_QBB && init_dim == 4) {
        e->s[2] = init[0]; e->s[3] = init[1]; e->s[6] = init[2]; e->s[7] = init[3];
    } else {
        for (int i = 0; i < e->S; ++i) e->s[i] = init[i];
    }
    e->plate[0] = e->plate[1] = 0.;
    if (e->kind == PSH_ENV_QBB && !e->simplified) {
        e->plate[0] = orc_qbb_ik(e->s[0] + e->dp[18], e->dp[4], e->dp[3]);
        e->plate[1] = orc_qbb_ik(e->s[1] + e->dp[19], e->dp[4], e->dp[3]);
    }
}

static void observe(const orc_env* e, double* obs) {
    const double* s = e->s;
    switch (e->kind) {
    case PSH_ENV_QQ: /* pysim/quanser_qube.py:50-54 */
        obs[0] = sin(s[0]); obs[1] = cos(s[0]); obs[2] = sin(s[1]); obs[3] = cos(s[1]); obs[4] = s[2]; obs[5] = s[3];
        break;
    case PSH_ENV_QCP_SU:
    case PSH_ENV_QCP_ST: /* pysim/quanser_cartpole.py:54-55 */
        obs[0] = s[0]; obs[1] = sin(s[1]); obs[2] = cos(s[1]); obs[3] = s[2]; obs[4] = s[3];
        break;
    case PSH_ENV_PEND: /* pysim/pendulum.py:38-39 */
        obs[0] = sin(s[0]); obs[1] = cos(s[0]); obs[2] = s[1];
        break;
    case PSH_ENV_QQ_ST: { /* pysim/quanser_qube.py:257-262: the error to the upright position */
        const double des[4] = {0., ORC_PI, 0., 0.};
        for (int i = 0; i < 4; ++i) obs[i] = des[i] - s[i];
        break;
    }
    default: /* environments/base.py:157-167 */
        for (int i = 0; i < e->S; ++i) obs[i] = s[i];
    }
}

/* _step_dynamics of each simulation; `act` is the clipped action and may be modified (QBB dead zone) */
static void step_dynamics(orc_env* e, double* act, double dt) {
    double* s = e->s;
    const double* p = e->dp;
    switch (e->kind) {
    case PSH_ENV_PEND: { /* pysim/pendulum.py:49-61 */
        double g = p[0], m_pole = p[1], l_pole = p[2], d_pole = p[3];
        double th = s[0], th_dot = s[1];
        double th_ddot = (act[0] - 0.5 * m_pole * g * l_pole * sin(th) - d_pole * th_dot) / (m_pole * (l_pole * l_pole) / 3.);
        s[1] += th_ddot * dt;
        s[0] += s[1] * dt;
        break;
    }
    case PSH_ENV_QQ_ST:
    case PSH_ENV_QQ: { /* pysim/quanser_qube.py:92-142.  All four RK4 stages evaluate the accelerations at the
                          UN-ADVANCED state (:140), so they are computed once. */
        double km = p[2], Rm = p[1], Dr = p[5], Dp = p[8];
        double th = s[0], al = s[1], thd = s[2], ald = s[3], u = act[0];
        (void)th;
        double sa = sin(al);
        double a = e->c[0] + e->c[1] * (sa * sa);
        double b = e->c[2] * cos(al);
        double c = e->c[3];
        double det = a * c - b * b;
        double trq = km * (u - km * thd) / Rm;
        double c0 = e->c[1] * sin(2 * al) * thd * ald - e->c[2] * sin(al) * ald * ald;
        double c1 = -0.5 * e->c[1] * sin(2 * al) * thd * thd + e->c[4] * sin(al);
        double x = trq - Dr * thd - c0;
        double y = -Dp * ald - c1;
        double thdd = (c * x - b * y) / det;
        double aldd = (a * y - b * x) / det;
        double k[4][4];
        k[0][0] = thd; k[0][1] = ald; k[0][2] = thdd; k[0][3] = aldd;
        for (int j = 1; j < 4; ++j) {
            double h = (j <= 2) ? dt / 2. : dt;
            double s2 = s[2] + h * k[j - 1][2];
            double s3 = s[3] + h * k[j - 1][3];
            k[j][0] = s2; k[j][1] = s3; k[j][2] = thdd; k[j][3] = aldd;
        }
        for (int i = 0; i < 4; ++i)
            s[i] += dt / 6 * (k[0][i] + 2 * k[1][i] + 2 * k[2][i] + k[3][i]);
        break;
    }
    case PSH_ENV_QCP_SU:
    case PSH_ENV_QCP_ST: { /* pysim/quanser_cartpole.py:95-124 */
        double g = p[0], eta_m = p[3], eta_g = p[4], K_g = p[5], r_mp = p[7], R_m = p[8], k_m = p[9];
        double B_pole = p[10], B_eq = p[11], m_pole = p[12], l_pole = p[13];
        double th = s[1], x_dot = s[2], th_dot = s[3];
        double force = (eta_g * K_g * eta_m * k_m) / (R_m * r_mp) * (eta_m * act[0] - K_g * k_m * x_dot / r_mp);
        double a11 = m_pole + e->c[1];
        double a12 = m_pole * l_pole * cos(th);
        double a21 = a12;
        double a22 = e->c[0] + m_pole * (l_pole * l_pole);
        double b1 = force - B_eq * x_dot - m_pole * l_pole * sin(th) * (th_dot * th_dot);
        double b2 = -B_pole * th_dot - m_pole * l_pole * g * sin(th);
        /* np.linalg.solve = LAPACK dgesv: LU with partial pivoting, then forward/back substitution */
        if (fabs(a21) > fabs(a11)) {
            double t;
            t = a11; a11 = a21; a21 = t;
            t = a12; a12 = a22; a22 = t;
            t = b1; b1 = b2; b2 = t;
        }
        double l21 = a21 * (1. / a11);
        double u22 = a22 - l21 * a12;
        double y2 = b2 - l21 * b1;
        double x2 = y2 / u22;
        double x1 = (b1 - x2 * a12) / a11;
        s[2] += x1 * dt; s[3] += x2 * dt;
        s[0] += s[2] * dt; s[1] += s[3] * dt;
        break;
    }
    case PSH_ENV_QBB: { /* pysim/quanser_ball_balancer.py:197-271 */
        double g = p[0], m_ball = p[1], r_ball = p[2], c_frict = p[13];
        double Vxp = p[14], Vxn = p[15], Vyp = p[16], Vyn = p[17], off_x = p[18], off_y = p[19];
        double J_ball = e->c[0], J_eq = e->c[1], c_kin = e->c[2], A_m = e->c[3], B_eq_v = e->c[4], zeta = e->c[5];
        if (!e->simplified) {
            if (Vxn <= act[0] && act[0] <= Vxp) act[0] = 0;
            if (Vyn <= act[1] && act[1] <= Vyp) act[1] = 0;
        }
        double th_x = s[0] + off_x, th_y = s[1] + off_y, x = s[2], y = s[3];
        double th_x_dot = s[4], th_y_dot = s[5], x_dot = s[6], y_dot = s[7];
        double th_x_ddot = (A_m * act[0] - B_eq_v * th_x_dot) / J_eq;
        double th_y_ddot = (A_m * act[1] - B_eq_v * th_y_dot) / J_eq;
        double a = e->plate[0], b = e->plate[1];
        double a_dot = c_kin * th_x_dot * cos(th_x) / cos(a);
        double b_dot = c_kin * -th_y_dot * cos(-th_y) / cos(b);
        double a_ddot = 1. / cos(a) * (c_kin * (th_x_ddot * cos(th_x) - (th_x_dot * th_x_dot) * sin(th_x)) + (a_dot * a_dot) * sin(a));
        double b_ddot = 1. / cos(b) * (c_kin * (-th_y_ddot * cos(th_y) - ((-th_y_dot) * (-th_y_dot)) * sin(-th_y)) + (b_dot * b_dot) * sin(b));
        double x_ddot, y_ddot;
        double r2 = r_ball * r_ball;
        if (e->simplified) {
            x_ddot = c_kin * m_ball * g * r2 * sin(th_x) / zeta;
            y_ddot = c_kin * m_ball * g * r2 * sin(th_y) / zeta;
        } else {
            x_ddot = (-c_frict * x_dot * r2 - J_ball * r_ball * a_ddot + m_ball * x * (a_dot * a_dot) * r2 +
                      c_kin * m_ball * g * r2 * sin(th_x)) / zeta;
            y_ddot = (-c_frict * y_dot * r2 - J_ball * r_ball * b_ddot + m_ball * y * ((-b_dot) * (-b_dot)) * r2 +
                      c_kin * m_ball * g * r2 * sin(th_y)) / zeta;
        }
        s[4] += th_x_ddot * dt; s[5] += th_y_ddot * dt; s[6] += x_ddot * dt; s[7] += y_ddot * dt;
        for (int i = 0; i < 4; ++i) s[i] += s[4 + i] * dt;
        e->plate[0] += a_dot * dt; e->plate[1] += b_dot * dt;
        break;
    }
    case PSH_ENV_BOB: { /* pysim/ball_on_beam.py:79-98 */
        double g = p[0], m_ball = p[1], c_frict = p[6], ang_offset = p[7];
        double x = s[0], a = s[1] + ang_offset, x_dot = s[2], a_dot = s[3];
        double zeta_beam = m_ball * (x * x) + e->c[1];
        double x_ddot = (-c_frict * x_dot + m_ball * x * (a_dot * a_dot) - m_ball * g * sin(a)) / e->c[2];
        double a_ddot = (act[0] - 2. * m_ball * x * x_dot * a_dot - m_ball * g * cos(a) * x) / zeta_beam;
        s[2] += x_ddot * dt; s[3] += a_ddot * dt;
        s[0] += s[2] * dt; s[1] += s[3] * dt;
        break;
    }
    case PSH_ENV_OMO: { /* pysim/one_mass_oscillator.py:75-84 */
        double m = p[0], omega = e->c[0], zeta = e->c[1];
        double sd0 = 0. * s[0] + 1. * s[1] + 0. * act[0];
        double sd1 = -(omega * omega) * s[0] + -2. * zeta * omega * s[1] + 1. / m * act[0];
        s[0] = s[0] + sd0 * dt; s[1] = s[1] + sd1 * dt;
        break;
    }
    }
}

/* SimPyEnv.step pysim/base.py:177-201.  act_io: in = raw action, out = the action object the caller
 * still holds after the call (QBB dead zone written through when no clipping copy was made). */
static double env_step(orc_env* e, const psh_task* t, double* act_io, double dt, int max_steps, int inf_steps,
                       int* done, int* st) {
    int remaining = inf_steps ? 0 : max_steps - (e->curr_step + 1);
    double rew = step_rew(e, t, e->s, act_io);
    /* limit_act: environments/base.py:169-176, spaces/box.py:144-148 */
    int nan = 0;
    double a[PSH_MAX_ACT];
    int inside = contains(act_io, e->a_lo, e->a_hi, e->A, &nan);
    for (int i = 0; i < e->A; ++i) a[i] = inside ? act_io[i] : fmin(fmax(act_io[i], e->a_lo[i]), e->a_hi[i]);
    step_dynamics(e, a, dt);
    if (inside) for (int i = 0; i < e->A; ++i) act_io[i] = a[i];
    e->curr_step += 1;
    int failed = !contains(e->s, e->s_lo, e->s_hi, e->S, &nan);
    int d = failed;
    if (!inf_steps && e->curr_step >= max_steps) { d = 1; *st |= PSH_ST_TIMEOUT; }
    if (failed) *st |= PSH_ST_FAILED;
    if (nan) *st |= PSH_ST_NAN;
    if (d) rew += final_rew(e, t, remaining);
    *done = d;
    return rew;
}

/* ---- policies ---------------------------------------------------------------------------------- */
static double nonlin(int k, double x) {
    switch (k) {
    case PSH_NL_TANH: return tanh(x);
    case PSH_NL_RELU: return !(x <= 0.) ? x : 0.;   /* a NaN stays a NaN, as in torch.relu */
    case PSH_NL_SIGMOID: return 1. / (1. + exp(-x));
    }
    return x;
}

static int num_feats(const psh_policy* pol, int O) {
    int n = 0;
    for (int f = 0; f < pol->n_feats; ++f) {
        switch (pol->feats[f]) {
        case PSH_FEAT_CONST: case PSH_FEAT_MULT: n += 1; break;
        case PSH_FEAT_RFF: n += pol->feat_size[f]; break;
        case PSH_FEAT_RBF: n += pol->feat_size[f] * O; break;
        default: n += O;
        }
    }
    return n;
}

int orc_policy_num_param(const psh_policy* pol, int O, int A) {
    if (pol->kind == PSH_POL_LINEAR) return A * num_feats(pol, O);
    int n = 0, last = O;
    for (int l = 0; l < pol->n_hidden; ++l) { n += pol->hidden[l] * last + pol->hidden[l]; last = pol->hidden[l]; }
    return n + A * last + A;
}

/* LinearPolicy.forward policies/linear.py:65-80 + features.py:34-42,90-140; FNN.forward policies/fnn.py:129-151 */
static void policy_forward(const psh_policy* pol, const double* w, int O, int A, const double* obs, double* act) {
    if (pol->kind == PSH_POL_LINEAR) {
        double f[PSH_MAX_LIN_FEAT];
        int n = 0;
        for (int k = 0; k < pol->n_feats; ++k) {
            int id = pol->feats[k];
            if (id == PSH_FEAT_CONST) { f[n++] = 1.; continue; }
            if (id == PSH_FEAT_MULT) { /* MultFeat.__call__ policies/features.py:160-167 */
                double v = obs[pol->feat_arg[k] & 15];
                for (int j = 1; j < pol->feat_size[k]; ++j) v = v * obs[(pol->feat_arg[k] >> (4 * j)) & 15];
                f[n++] = v;
                continue;
            }
            if (id == PSH_FEAT_RFF) { /* RandFourierFeat.__call__ policies/features.py:203-219 */
                const int D = pol->feat_size[k];
                const double* data = pol->feat_data + pol->feat_off[k];
                const double scale = data[0], *shift = data + 1, *freq = data + 1 + D;
                for (int j = 0; j < D; ++j) {
                    double z = 0.;
                    for (int i = 0; i < O; ++i) z += obs[i] * freq[j * O + i];
                    f[n++] = scale * cos(z + shift[j]);
                }
                continue;
            }
            if (id == PSH_FEAT_RBF) { /* RBFFeat.__call__ policies/features.py:278-303, normalize utils/normalizing.py:87-105 */
                const int K = pol->feat_size[k];
                const double* scale = pol->feat_data + pol->feat_off[k], *cen = scale + O;
                double total = 0.;
                for (int i = 0; i < O; ++i) {
                    double norm = 0.;
                    for (int j = 0; j < K; ++j) {
                        double dlt = obs[i] - cen[j * O + i];
                        double e = exp(-scale[i] * (dlt * dlt));
                        f[n + i * K + j] = e;
                        norm += fabs(e);
                    }
                    total += norm;
                    if (pol->feat_arg[k]) { /* every input dimension sums to one */
                        if (!(norm > 1e-8)) norm = 1.;
                        for (int j = 0; j < K; ++j) f[n + i * K + j] /= norm;
                    }
                }
                if (!pol->feat_arg[k]) {
                    if (!(total > 1e-8)) total = 1.;
                    for (int j = 0; j < K * O; ++j) f[n + j] /= total;
                }
                n += K * O;
                continue;
            }
            for (int i = 0; i < O; ++i) {
                double x = obs[i], v = x;
                switch (id) {
                case PSH_FEAT_IDENTITY: v = x; break;
                case PSH_FEAT_SIGN: v = (x > 0.) - (x < 0.); break;
                case PSH_FEAT_ABS: v = fabs(x); break;
                case PSH_FEAT_SQUARED: v = x * x; break;
                case PSH_FEAT_CUBED: v = x * x * x; break;
                case PSH_FEAT_SIG: v = 1. / (1. + exp(-x)); break;
                case PSH_FEAT_BELL: v = exp(-(x * x)); break;
                case PSH_FEAT_SIN: v = sin(x); break;
                case PSH_FEAT_COS: v = cos(x); break;
                case PSH_FEAT_SINSIN: v = sin(x) * sin(x); break;
                case PSH_FEAT_SINCOS: v = sin(x) * cos(x); break;
                }
                f[n++] = v;
            }
        }
        for (int a = 0; a < A; ++a) {
            double acc = 0.;
            for (int j = 0; j < n; ++j) acc += w[a * n + j] * f[j];
            act[a] = acc;
        }
        return;
    }
    double x[PSH_MAX_WIDTH], y[PSH_MAX_WIDTH];
    int last = O;
    for (int i = 0; i < O; ++i) x[i] = obs[i];
    for (int l = 0; l < pol->n_hidden; ++l) {
        int h = pol->hidden[l];
        const double* W = w; const double* b = w + h * last;
        for (int j = 0; j < h; ++j) {
            double acc = 0.;
            for (int k = 0; k < last; ++k) acc += W[j * last + k] * x[k];
            y[j] = nonlin(pol->hidden_nonlin[l], acc + b[j]);
        }
        for (int j = 0; j < h; ++j) x[j] = y[j];
        w += h * last + h;
        last = h;
    }
    const double* W = w; const double* b = w + A * last;
    for (int a = 0; a < A; ++a) {
        double acc = 0.;
        for (int k = 0; k < last; ++k) acc += W[a * last + k] * x[k];
        act[a] = nonlin(pol->out_nonlin, acc + b[a]);
    }
}

int orc_policy_forward(const psh_policy* pol, int O, int A, const double* obs, double* act) {
    policy_forward(pol, pol->params, O, A, obs, act);
    return 0;
}

/* ---- fused environment wrappers (the modules of environment_wrappers), see psh_wrappers in the header ---- */
static void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
static double u01(uint32_t hi, uint32_t lo);

/* n standard normals for (seed, env, step t, stream): Box-Muller pairs, pair p from counter word 3 = stream + 16 p */
static void step_normals(uint64_t seed, uint64_t env, uint32_t t, uint32_t stream, int n, double* out) {
    for (int p = 0; 2 * p < n; ++p) {
        uint32_t ctr[4] = {(uint32_t)env, (uint32_t)(env >> 32), t, stream + 16u * (uint32_t)p};
        uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
        uint32_t o[4];
        philox4x32_10(ctr, key, o);
        double ua = u01(o[0], o[1]), ub = u01(o[2], o[3]);
        double r = sqrt(-2.0 * log(1.0 - ua));
        out[2 * p] = r * cos(2.0 * ORC_PI * ub);
        if (2 * p + 1 < n) out[2 * p + 1] = r * sin(2.0 * ORC_PI * ub);
    }
}

/* two uniforms in [0, 1) for (seed, env, step t, stream) */
static void step_uniforms(uint64_t seed, uint64_t env, uint32_t t, uint32_t stream, double* out) {
    uint32_t ctr[4] = {(uint32_t)env, (uint32_t)(env >> 32), t, stream};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t o[4];
    philox4x32_10(ctr, key, o);
    out[0] = u01(o[0], o[1]);
    out[1] = u01(o[2], o[3]);
}

/* observation-space bounds of the simulation: pysim/quanser_qube.py:25-33, quanser_cartpole.py:45-50,
 * quanser_ball_balancer.py:73, ball_on_beam.py:32, one_mass_oscillator.py:39 */
static void sim_obs_bounds(const orc_env* e, double* lo, double* hi) {
    switch (e->kind) {
    case PSH_ENV_QQ: {
        const double m[6] = {1., 1., 1., 1., INFINITY, INFINITY};
        for (int r = 0; r < 6; ++r) { lo[r] = -m[r]; hi[r] = m[r]; }
        break;
    }
    case PSH_ENV_PEND: { /* pysim/pendulum.py:22,26 */
        const double m[3] = {1., 1., INFINITY};
        for (int r = 0; r < 3; ++r) { lo[r] = -m[r]; hi[r] = m[r]; }
        break;
    }
    case PSH_ENV_QCP_SU:
    case PSH_ENV_QCP_ST: {
        const double m[5] = {e->dp[2] / 2., 1., 1., INFINITY, INFINITY};
        for (int r = 0; r < 5; ++r) { lo[r] = -m[r]; hi[r] = m[r]; }
        break;
    }
    default:
        for (int r = 0; r < e->O; ++r) { lo[r] = e->s_lo[r]; hi[r] = e->s_hi[r]; }
    }
}

/* EnvWrapperObs.reset/step: the simulation's observation through the observation wrappers, innermost first;
 * returns the number of entries the policy sees */
static int wrap_obs(const psh_wrappers* w, const orc_env* e, uint64_t env, uint32_t t, const double* sim_obs, double* out) {
    double o[PSH_MAX_OBS];
    const int O = e->O;
    int keep = (1 << O) - 1;
    for (int r = 0; r < O; ++r) o[r] = sim_obs[r];
    for (int j = w->n_ops - 1; j >= 0; --j) {
        if (w->ops[j] == PSH_WRAP_OBS_NOISE) {
            double n[PSH_MAX_OBS];
            step_normals(w->noise_seed, env, t, 1u, O, n);
            for (int r = 0; r < O; ++r) o[r] = o[r] + (n[r] * w->obs_noise_std[r] + w->obs_noise_mean[r]);
        } else if (w->ops[j] == PSH_WRAP_OBS_NORM) {
            double lo[PSH_MAX_OBS], hi[PSH_MAX_OBS];
            sim_obs_bounds(e, lo, hi);
            for (int r = 0; r < O; ++r) {
                if (!isnan(w->obs_lo[r])) lo[r] = w->obs_lo[r];
                if (!isnan(w->obs_hi[r])) hi[r] = w->obs_hi[r];
                o[r] = (o[r] - lo[r]) / (hi[r] - lo[r]) * 2 - 1;
            }
        } else if (w->ops[j] == PSH_WRAP_OBS_PARTIAL) {
            keep = w->obs_keep_mask;
        }
    }
    int n = 0;
    for (int r = 0; r < O; ++r) if (keep & (1 << r)) out[n++] = o[r];
    return n;
}

static int wrap_obs_dim(const psh_wrappers* w, int O) {
    for (int j = 0; j < w->n_ops; ++j)
        if (w->ops[j] == PSH_WRAP_OBS_PARTIAL) {
            int n = 0;
            for (int r = 0; r < O; ++r) n += (w->obs_keep_mask >> r) & 1;
            return n;
        }
    return O;
}

/* ---- rollout(): sampling/rollout.py:88-275 ------------------------------------------------------- */
static void rollout_one(const psh_rollout_desc* d, int i, const int32_t dims[5]) {
    const int S = dims[0], O = dims[1], A = dims[2], P = dims[3], N = d->n_envs, T = d->max_steps;
    const psh_wrappers* w = &d->wrap;
    const int OP = wrap_obs_dim(w, O);  /* what the policy sees and `obs` records */
    orc_env e;
    memset(&e, 0, sizeof e);
    e.kind = d->env_kind; e.S = S; e.O = O; e.A = A; e.P = P;
    e.simplified = (d->flags & PSH_F_SIMPLIFIED_DYN) != 0;
    double dp[PSH_MAX_DP], init[PSH_MAX_STATE];
    int col = d->dp_index ? d->dp_index[i] : i;
    if (col < 0 || col >= d->ld_dp ||
        (d->policy.policy_index && (d->policy.policy_index[i] < 0 || d->policy.policy_index[i] >= d->policy.n_policies))) {
        /* an index outside its table (the library's own rule, no counterpart in the reference): the env is not run */
        d->ret[i] = 0.; d->length[i] = 0; d->status[i] = PSH_ST_BAD_INDEX;
        return;
    }
    for (int r = 0; r < P; ++r) dp[r] = d->domain_param[(size_t)r * d->ld_dp + col];
    for (int r = 0; r < d->init_dim; ++r) init[r] = d->init_state[(size_t)r * d->ld_init + i];
    env_reset(&e, &d->task, dp, init, d->init_dim);
    /* population mode (parameter_exploration_sampler.py:174-201): policy i / policy_group */
    const size_t prow = d->policy.policy_index ? (size_t)d->policy.policy_index[i]
                                                : (d->policy.policy_group > 0 ? (size_t)(i / d->policy.policy_group) : 0);
    const double* pw = d->policy.params + prow * d->policy.num_param;
    const uint64_t genv = w->env_offset + (uint64_t)i;

    int delay = 0, has_delay = 0, new_array = 0;
    double dq[PSH_MAX_DELAY + 1][PSH_MAX_ACT];
    for (int j = 0; j < w->n_ops; ++j) {
        if (w->ops[j] == PSH_WRAP_ACT_DELAY) has_delay = 1;
        if (w->ops[j] == PSH_WRAP_ACT_NORM || w->ops[j] == PSH_WRAP_ACT_NOISE) new_array = 1;
    }
    if (has_delay) {
        /* ActDelayWrapper environment_wrappers/action_delay.py:46-81: queue pre-filled with `delay` zero actions */
        delay = w->act_delay_row >= 0 ? (int)d->domain_param[(size_t)w->act_delay_row * d->ld_dp + col] : w->act_delay;
        if (delay < 0) delay = 0;
        if (delay > PSH_MAX_DELAY) delay = PSH_MAX_DELAY;
        memset(dq, 0, sizeof dq);
    }

    double sim_obs[PSH_MAX_OBS], obs[PSH_MAX_OBS], act[PSH_MAX_ACT], ret = 0.;
    int st = 0, done = 0, k = 0;
    observe(&e, sim_obs);
    wrap_obs(w, &e, genv, 0u, sim_obs, obs);
    const int stop_on_done = (d->flags & PSH_F_STOP_ON_DONE) != 0;
    while (!(done && stop_on_done) && e.curr_step < T) {
        int nan = 0;
        for (int r = 0; r < OP; ++r) if (isnan(obs[r])) nan = 1;
        if (d->obs) for (int r = 0; r < OP; ++r) d->obs[((size_t)k * OP + r) * N + i] = obs[r];
        if (nan) { st |= PSH_ST_NAN; break; }
        policy_forward(&d->policy, pw, OP, A, obs, act);
        if (d->policy.expl_kind == PSH_EXPL_NORMAL) {
            /* NormalActNoiseExplStrat: Normal(a + mean, std).rsample() = loc + eps * scale
             * (exploration/stochastic_action.py:55, normal_noise.py:100-108) */
            double n[PSH_MAX_ACT];
            step_normals(w->noise_seed, genv, (uint32_t)k, 3u, A, n);
            for (int r = 0; r < A; ++r) act[r] = (act[r] + d->policy.expl_mean[r]) + n[r] * d->policy.expl_scale[r];
        } else if (d->policy.expl_kind == PSH_EXPL_UNIFORM) {
            /* UniformActNoiseExplStrat: Uniform(m - halfspan, m + halfspan).rsample() = low + u * (high - low) */
            double u[2];
            step_uniforms(w->noise_seed, genv, (uint32_t)k, 3u, u);
            for (int r = 0; r < A; ++r) {
                double m = act[r] + d->policy.expl_mean[r];
                double lo = m - d->policy.expl_scale[r], hi = m + d->policy.expl_scale[r];
                act[r] = lo + u[r] * (hi - lo);
            }
        }
        for (int r = 0; r < A; ++r) if (isnan(act[r])) nan = 1;
        if (nan) { st |= PSH_ST_NAN; break; }
        /* EnvWrapperAct.step, outermost wrapper first */
        double inner[PSH_MAX_ACT];
        for (int r = 0; r < A; ++r) inner[r] = act[r];
        for (int j = 0; j < w->n_ops; ++j) {
            if (w->ops[j] == PSH_WRAP_ACT_DELAY && delay > 0) {
                /* _process_act (action_delay.py:66-81): append, then pop the oldest */
                for (int r = 0; r < A; ++r) dq[delay][r] = inner[r];
                for (int r = 0; r < A; ++r) inner[r] = dq[0][r];
                for (int q = 0; q < delay; ++q) for (int r = 0; r < A; ++r) dq[q][r] = dq[q + 1][r];
            } else if (w->ops[j] == PSH_WRAP_ACT_NORM) {
                for (int r = 0; r < A; ++r) inner[r] = e.a_lo[r] + (inner[r] + 1) * (e.a_hi[r] - e.a_lo[r]) / 2;
            } else if (w->ops[j] == PSH_WRAP_ACT_NOISE) {
                double n[PSH_MAX_ACT];
                step_normals(w->noise_seed, genv, (uint32_t)k, 2u, A, n);
                for (int r = 0; r < A; ++r) inner[r] = inner[r] + (n[r] * w->act_noise_std[r] + w->act_noise_mean[r]);
            }
        }
        double rew = env_step(&e, &d->task, inner, d->dt, T, 0, &done, &st);
        /* The simulation may have written into the array it was given (QBB dead zone).  Without an array-creating
         * wrapper that array is the policy's output of `delay` steps ago (ActDelayWrapper queues the array
         * OBJECTS), which the rollout has recorded; with one, the recorded actions stay untouched. */
        if (d->act) for (int r = 0; r < A; ++r) d->act[((size_t)k * A + r) * N + i] = act[r];
        if (d->act && !new_array && k - delay >= 0)
            for (int r = 0; r < A; ++r) d->act[((size_t)(k - delay) * A + r) * N + i] = inner[r];
        if (d->rew) d->rew[(size_t)k * N + i] = rew;
        ret += rew;
        observe(&e, sim_obs);
        ++k;
        wrap_obs(w, &e, genv, (uint32_t)k, sim_obs, obs);
        if (st & PSH_ST_NAN) break;
        if (d->bernoulli_reset > 0. && !(done && stop_on_done) && e.curr_step < T) {
            /* sampling/rollout.py:248-254: stop with probability p after the step (flagged only if nothing else ended it) */
            double u[2];
            step_uniforms(w->noise_seed, genv, (uint32_t)(k - 1), 4u, u);
            if (u[0] < d->bernoulli_reset) { st |= PSH_ST_BERNOULLI; break; }
        }
    }
    if (d->obs) for (int r = 0; r < OP; ++r) d->obs[((size_t)k * OP + r) * N + i] = obs[r];
    d->ret[i] = ret;
    d->length[i] = k;
    d->status[i] = st;
    if (d->final_state) for (int r = 0; r < S; ++r) d->final_state[(size_t)r * N + i] = e.s[r];
}

int orc_rollout(const psh_rollout_desc* d, int n_threads) {
    int32_t dims[5];
    if (orc_env_dims(d->env_kind, dims)) return PSH_ERR_ARG;
    if (d->policy.num_param != orc_policy_num_param(&d->policy, wrap_obs_dim(&d->wrap, dims[1]), dims[2])) return PSH_ERR_ARG;
    (void)n_threads;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 64)
#endif
    for (int i = 0; i < d->n_envs; ++i) rollout_one(d, i, dims);
    return 0;
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- reset / step with caller-owned environment memory ------------------------------------------ */
static int load_env(orc_env* e, const psh_step_desc* d, int i, const int32_t dims[5]) {
    memset(e, 0, sizeof *e);
    e->kind = d->env_kind; e->S = dims[0]; e->O = dims[1]; e->A = dims[2]; e->P = dims[3];
    e->simplified = (d->flags & PSH_F_SIMPLIFIED_DYN) != 0;
    int col = d->dp_index ? d->dp_index[i] : i;
    const int bad = col < 0 || col >= d->ld_dp;  /* outside the table: column 0 is read, the status word says so */
    if (bad) col = 0;
    for (int r = 0; r < e->P; ++r) e->dp[r] = d->domain_param[(size_t)r * d->ld_dp + col];
    return bad ? PSH_ST_BAD_INDEX : 0;
}

int orc_reset(const psh_step_desc* d) {
    int32_t dims[5];
    if (orc_env_dims(d->env_kind, dims)) return PSH_ERR_ARG;
    const int N = d->n_envs;
    for (int i = 0; i < N; ++i) {
        orc_env e;
        const int bad = load_env(&e, d, i, dims);
        double dp[PSH_MAX_DP], init[PSH_MAX_STATE], obs[PSH_MAX_OBS];
        for (int r = 0; r < e.P; ++r) dp[r] = e.dp[r];
        for (int r = 0; r < d->init_dim; ++r) init[r] = d->init_state[(size_t)r * N + i];
        env_reset(&e, &d->task, dp, init, d->init_dim);
        for (int r = 0; r < e.S; ++r) d->state[(size_t)r * N + i] = e.s[r];
        if (d->env_kind == PSH_ENV_QBB) for (int r = 0; r < 2; ++r) d->state[(size_t)(8 + r) * N + i] = e.plate[r];
        d->curr_step[i] = 0;
        d->flags_io[i] = 0;
        observe(&e, obs);
        int nan = 0;
        for (int r = 0; r < e.O; ++r) { d->obs[(size_t)r * N + i] = obs[r]; if (isnan(obs[r])) nan = 1; }
        if (d->status) d->status[i] = (nan ? PSH_ST_NAN : 0) | bad;
    }
    return 0;
}

int orc_step(const psh_step_desc* d) {
    int32_t dims[5];
    if (orc_env_dims(d->env_kind, dims)) return PSH_ERR_ARG;
    const int N = d->n_envs;
    for (int i = 0; i < N; ++i) {
        orc_env e;
        const int bad = load_env(&e, d, i, dims);
        calc_constants_and_spaces(&e);
        task_reset(&e, &d->task);
        for (int r = 0; r < e.S; ++r) e.s[r] = d->state[(size_t)r * N + i];
        if (d->env_kind == PSH_ENV_QBB) for (int r = 0; r < 2; ++r) e.plate[r] = d->state[(size_t)(8 + r) * N + i];
        e.curr_step = d->curr_step[i];
        e.yielded = d->flags_io[i] & 1;
        double act[PSH_MAX_ACT], obs[PSH_MAX_OBS];
        int st = bad, done = 0;
        for (int r = 0; r < e.A; ++r) { act[r] = d->act[(size_t)r * N + i]; if (isnan(act[r])) st |= PSH_ST_NAN; }
        double rew = env_step(&e, &d->task, act, d->dt, d->max_steps, (d->flags & PSH_F_MAX_STEPS_INF) != 0, &done, &st);
        for (int r = 0; r < e.A; ++r) d->act[(size_t)r * N + i] = act[r];
        for (int r = 0; r < e.S; ++r) d->state[(size_t)r * N + i] = e.s[r];
        if (d->env_kind == PSH_ENV_QBB) for (int r = 0; r < 2; ++r) d->state[(size_t)(8 + r) * N + i] = e.plate[r];
        d->curr_step[i] = e.curr_step;
        d->flags_io[i] = e.yielded;
        observe(&e, obs);
        for (int r = 0; r < e.O; ++r) d->obs[(size_t)r * N + i] = obs[r];
        d->rew[i] = rew;
        d->done[i] = done;
        if (d->status) d->status[i] = st;
    }
    return 0;
}

/* single evaluations for known-answer tests */
double orc_step_rew(int env_kind, const psh_task* t, const double* dp, const double* state, const double* act) {
    int32_t dims[5];
    orc_env e;
    memset(&e, 0, sizeof e);
    orc_env_dims(env_kind, dims);
    e.kind = env_kind; e.S = dims[0]; e.O = dims[1]; e.A = dims[2]; e.P = dims[3];
    for (int r = 0; r < e.P; ++r) e.dp[r] = dp[r];
    calc_constants_and_spaces(&e);
    task_reset(&e, t);
    return step_rew(&e, t, state, act);
}

void orc_constants(int env_kind, const double* dp, double* c8, double* s_lo, double* s_hi, double* a_lo, double* a_hi) {
    int32_t dims[5];
    orc_env e;
    memset(&e, 0, sizeof e);
    orc_env_dims(env_kind, dims);
    e.kind = env_kind; e.S = dims[0]; e.O = dims[1]; e.A = dims[2]; e.P = dims[3];
    for (int r = 0; r < e.P; ++r) e.dp[r] = dp[r];
    calc_constants_and_spaces(&e);
    for (int i = 0; i < 8; ++i) c8[i] = e.c[i];
    for (int i = 0; i < e.S; ++i) { s_lo[i] = e.s_lo[i]; s_hi[i] = e.s_hi[i]; }
    for (int i = 0; i < e.A; ++i) { a_lo[i] = e.a_lo[i]; a_hi[i] = e.a_hi[i]; }
}

/* ---- counter-based RNG of the product's on-device sampling (the product's own design; the
 * reference draws from NumPy/torch global generators, so only distributions can agree with it) ---- */
static void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void orc_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox4x32_10(ctr, key, out); }

static double u01(uint32_t hi, uint32_t lo) { /* 53-bit uniform in [0,1) */
    return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6)) * (1.0 / 9007199254740992.0);
}

/* two uniforms for (seed, env, row, stream) */
static void draw2(uint64_t seed, uint64_t env, uint32_t row, uint32_t stream, double* ua, double* ub) {
    uint32_t ctr[4] = {(uint32_t)env, (uint32_t)(env >> 32), row, stream};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t o[4];
    philox4x32_10(ctr, key, o);
    *ua = u01(o[0], o[1]);
    *ub = u01(o[2], o[3]);
}

int orc_randomize(const psh_rand_desc* d) {
    for (int r = 0; r < d->n_rows; ++r)
        for (int i = 0; i < d->n_envs; ++i) d->out[(size_t)r * d->ld_out + i] = d->nominal[r];
    for (int k = 0; k < d->n_distr; ++k) {
        const psh_distr* q = &d->distr[k];
        for (int i = 0; i < d->n_envs; ++i) {
            double ua, ub, x = q->a;
            draw2(d->seed, d->env_offset + (uint64_t)i, (uint32_t)q->row, 0u, &ua, &ub);
            switch (q->kind) {
            case PSH_DISTR_NORMAL: {
                double r = sqrt(-2.0 * log(1.0 - ua)); /* 1-ua in (0,1] */
                x = q->a + q->b * (r * cos(2.0 * ORC_PI * ub));
                break;
            }
            case PSH_DISTR_UNIFORM: x = (q->a - q->b) + ua * ((q->a + q->b) - (q->a - q->b)); break;
            case PSH_DISTR_BERNOULLI: x = (ua < q->c) ? q->b : q->a; break;
            default: x = q->a;
            }
            if (q->kind != PSH_DISTR_CONST) {
                x = fmin(fmax(x, q->clip_lo), q->clip_up);
                if (q->roundint) x = nearbyint(x);
            }
            d->out[(size_t)q->row * d->ld_out + i] = x;
        }
    }
    return 0;
}

/* init_space.sample_uniform(): spaces/box.py:133-142, spaces/polar.py:64-69, spaces/compound.py:51-54 with the
 * per-env init spaces of pysim/quanser_qube.py:26-35, quanser_cartpole.py:261-269,314-318,
 * quanser_ball_balancer.py:66-75, ball_on_beam.py:33-44, one_mass_oscillator.py:33-40 */
int orc_sample_init(const psh_init_desc* d) {
    int32_t dims[5];
    if (orc_env_dims(d->env_kind, dims)) return PSH_ERR_ARG;
    const int I = dims[4];
    for (int i = 0; i < d->n_envs; ++i) {
        int col = d->dp_index ? d->dp_index[i] : i;
        const int bad = d->dp_index && (col < 0 || col >= d->ld_dp);  /* outside the table: a NaN init state */
        if (bad) col = 0;
        double lo[4], hi[4], u[5];
        uint64_t env = d->env_offset + (uint64_t)i;
        for (int r = 0; r < 5; ++r) { double ub; draw2(d->seed, env, (uint32_t)r, 5u, &u[r], &ub); } /* stream 5: initial states */
        switch (d->env_kind) {
        case PSH_ENV_QQ: {
            double m[4] = {2. / 180 * ORC_PI, 1. / 180 * ORC_PI, 0.5 / 180 * ORC_PI, 0.5 / 180 * ORC_PI};
            for (int r = 0; r < 4; ++r) { lo[r] = -m[r]; hi[r] = m[r]; }
            break;
        }
        case PSH_ENV_QCP_SU: {
            double m[4] = {0.03, 1 / 180. * ORC_PI, 0.005, 2 / 180. * ORC_PI};
            for (int r = 0; r < 4; ++r) { lo[r] = -m[r]; hi[r] = m[r]; }
            break;
        }
        case PSH_ENV_QCP_ST: {
            const double off = 8 / 180. * ORC_PI;
            lo[0] = -0.05; hi[0] = 0.05; lo[1] = ORC_PI - off; hi[1] = ORC_PI + off;
            lo[2] = -0.05; hi[2] = 0.05; lo[3] = -8 / 180. * ORC_PI; hi[3] = 8 / 180. * ORC_PI;
            break;
        }
        case PSH_ENV_QBB: {
            double l_plate = d->domain_param[(size_t)3 * d->ld_dp + col];
            lo[0] = 0.75 * l_plate / 2; hi[0] = 0.8 * l_plate / 2; lo[1] = -ORC_PI; hi[1] = ORC_PI;
            lo[2] = -0.05 * 0.5; hi[2] = 0.05 * 0.5; lo[3] = -0.05 * 0.5; hi[3] = 0.05 * 0.5;
            break;
        }
        case PSH_ENV_BOB: {
            double l_beam = d->domain_param[(size_t)4 * d->ld_dp + col];
            int right = u[4] >= 0.5;
            lo[0] = right ? 0.7 * l_beam / 2. : -0.8 * l_beam / 2.;
            hi[0] = right ? 0.8 * l_beam / 2. : -0.7 * l_beam / 2.;
            lo[1] = -5 / 180. * ORC_PI; hi[1] = 5 / 180. * ORC_PI;
            lo[2] = -0.02 * 10.; hi[2] = 0.02 * 10.; lo[3] = -0.02 * ORC_PI; hi[3] = 0.02 * ORC_PI;
            break;
        }
        case PSH_ENV_OMO:
            lo[0] = -0.75 * 1.; hi[0] = -0.65 * 1.; lo[1] = -0.01 * 10.; hi[1] = 0.01 * 10.;
            break;
        case PSH_ENV_PEND: /* SingularStateSpace(zeros), pysim/pendulum.py:23,27 */
            lo[0] = hi[0] = 0.; lo[1] = hi[1] = 0.;
            break;
        case PSH_ENV_QQ_ST: /* pysim/quanser_qube.py:244-245 */
            lo[0] = -5. / 180 * ORC_PI; hi[0] = 5. / 180 * ORC_PI; lo[1] = 175. / 180 * ORC_PI; hi[1] = 185. / 180 * ORC_PI;
            lo[2] = hi[2] = 0.; lo[3] = hi[3] = 0.;
            break;
        }
        double v[4];
        for (int r = 0; r < I; ++r) v[r] = lo[r] + u[r] * (hi[r] - lo[r]);
        if (d->env_kind == PSH_ENV_QBB) {
            double rad = v[0], phi = v[1];
            v[0] = rad * cos(phi); v[1] = rad * sin(phi);
        }
        for (int r = 0; r < I; ++r) d->out[(size_t)r * d->ld_out + i] = bad ? NAN : v[r];
    }
    return 0;
}

/* Bootstrap replications of the mean (Pyrado/pyrado/sampling/bootstrapping.py:47-61 with stat_fcn = np.mean), with the
 * product's draws: draw j of replication r = data[floor(u n)], u the (j & 1)-th uniform of Philox at counter
 * (r, 0, j >> 1, 6).  Summed in draw order. */
void orc_bootstrap_means(int n, int num_reps, const double* data, uint64_t seed, double* out) {
    for (int r = 0; r < num_reps; ++r) {
        double acc = 0.;
        for (int p = 0; 2 * p < n; ++p) {
            double u0, u1;
            draw2(seed, (uint64_t)r, (uint32_t)p, 6u, &u0, &u1);
            int j0 = (int)(u0 * n), j1 = (int)(u1 * n);
            if (j0 >= n) j0 = n - 1;
            if (j1 >= n) j1 = n - 1;
            acc += data[j0];
            if (2 * p + 1 < n) acc += data[j1];
        }
        out[r] = acc / n;
    }
}
