// Device-side simulations: one GPU lane owns one environment instance.  State, the domain parameters a
// step needs and the constants derived from them live in registers for the whole horizon.
//
// Each Env<KIND> follows one SimPyEnv subclass of the reference (paths relative to
// Pyrado/pyrado/environments/pysim/):
//   QQ   quanser_qube.py:20-142,222-271  QCP  quanser_cartpole.py:41-124,252-269,307-318   PEND pendulum.py:17-61
//   QBB  quanser_ball_balancer.py:60-271,413-480   BOB  ball_on_beam.py:21-98   OMO  one_mass_oscillator.py:28-84
// and the step / reward / termination logic follows base.py:119-201 with the tasks of
// Pyrado/pyrado/tasks/{desired_state.py:82-128, reward_functions.py:167-303, final_reward.py:70-187}.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/pysimhip.h"
#include "psh_math.h"

#define PSH_PI 3.141592653589793
#define PSH_DEV __device__ __forceinline__

// column view of an environment-minor [rows, M] table
struct DpCol {
    const double* base;
    long ld;
    long col;
    PSH_DEV double operator[](int r) const { return base[r * ld + col]; }
};

// Column of env i in a domain-parameter table with `ld` columns.  An entry of the caller's dp_index array that points
// outside the table is not followed: column 0 is read instead and *bad is set (such an env is not run).  Without
// dp_index the column is i, which the host side has checked against the table wherever the table is read.
PSH_DEV long dp_column(const int* dp_index, long ld, long i, bool* bad) {
    const long col = dp_index ? (long)dp_index[i] : i;
    *bad = dp_index != nullptr && (unsigned long)col >= (unsigned long)ld;
    return *bad ? 0 : col;
}

template <int KIND> struct Env;

// ------------------------------------------------------------------------------------------------ QQube
// STAB: QQubeStabSim (quanser_qube.py:222-271), same dynamics, narrower state space, observes the error to upright
template <bool STAB> struct EnvQQ {
    static constexpr int S = 4, O = STAB ? 4 : 6, A = 1, P = 10, I = 4, X = 0;
    double s[4];
    double c0, c1, c2, c3, c4, km, Rm, rRm, Dr, Dp, max_volt;
    double sth, cth, sal, cal;  // sin/cos of s[0], s[1]: shared by observe() and the next dynamics()

    PSH_DEV void setup(const DpCol& p) {
        const double g = p[0], Mr = p[3], Lr = p[4], Mp = p[6], Lp = p[7];
        Rm = p[1]; km = p[2]; Dr = p[5]; Dp = p[8]; max_volt = p[9];
        rRm = 1. / Rm;
        const double Jr = Mr * (Lr * Lr) / 12, Jp = Mp * (Lp * Lp) / 12;
        c0 = Jr + Mp * (Lr * Lr);
        c1 = 0.25 * Mp * (Lp * Lp);
        c2 = 0.5 * Mp * Lp * Lr;
        c3 = Jp + c1;
        c4 = 0.5 * Mp * Lp * g;
    }
    PSH_DEV void state_bounds(double* lo, double* hi) const {
        const double m[4] = {115. / 180 * PSH_PI, 4 * PSH_PI, 20 * PSH_PI, 20 * PSH_PI};
        for (int i = 0; i < 4; ++i) { lo[i] = -m[i]; hi[i] = m[i]; }
        if (STAB) {
            const double th = 15 / 180. * PSH_PI;
            lo[0] = -120. / 180 * PSH_PI; hi[0] = 120. / 180 * PSH_PI;
            lo[1] = PSH_PI - th; hi[1] = PSH_PI + th;
        }
    }
    PSH_DEV void act_bounds(double* lo, double* hi) const { lo[0] = -max_volt; hi[0] = max_volt; }
    PSH_DEV void obs_bounds(double* lo, double* hi) const {  // quanser_qube.py:25-33; STAB :252
        if (STAB) { state_bounds(lo, hi); return; }
        const double inf = __builtin_inf();
        for (int i = 0; i < 4; ++i) { lo[i] = -1.; hi[i] = 1.; }
        lo[O - 2] = lo[O - 1] = -inf; hi[O - 2] = hi[O - 1] = inf;
    }
    PSH_DEV void refresh() { psh_sincos(s[0], &sth, &cth); psh_sincos(s[1], &sal, &cal); }
    PSH_DEV void reset(const double* init, int, bool) {
        for (int i = 0; i < 4; ++i) s[i] = init[i];
        refresh();
    }
    PSH_DEV void observe(double* o) const {
        if (STAB) {
            o[0] = 0. - s[0]; o[1] = PSH_PI - s[1]; o[2] = 0. - s[2]; o[3] = 0. - s[3];
        } else {
            o[0] = sth; o[1] = cth; o[2] = sal; o[3] = cal; o[O - 2] = s[2]; o[O - 1] = s[3];
        }
    }
    PSH_DEV void dynamics(double* act, double dt, bool) {
        const double thd = s[2], ald = s[3], u = act[0];
        const double s2al = 2. * sal * cal;  // sin(2 al)
        const double a = c0 + c1 * (sal * sal), b = c2 * cal, c = c3;
        const double det = a * c - b * b;
        const double trq = psh_quot_by(km * (u - km * thd), Rm, rRm);
        const double cc0 = c1 * s2al * thd * ald - c2 * sal * ald * ald;
        const double cc1 = -0.5 * c1 * s2al * thd * thd + c4 * sal;
        const double x = trq - Dr * thd - cc0, y = -Dp * ald - cc1;
        const double rdet = psh_rcp_refined(det);  // one reciprocal serves both quotients
        const double thdd = psh_quot_by(c * x - b * y, det, rdet), aldd = psh_quot_by(a * y - b * x, det, rdet);
        // "RK4" of the reference: every stage re-evaluates the accelerations at the UN-ADVANCED state
        // (quanser_qube.py:140), so only the velocity rows of k1..k3 differ from k0.
        double kv[4][2];
        kv[0][0] = thd; kv[0][1] = ald;
        for (int j = 1; j < 4; ++j) {
            const double h = (j <= 2) ? dt / 2. : dt;
            kv[j][0] = thd + h * thdd;  // k[j-1][2] == thdd for every j
            kv[j][1] = ald + h * aldd;
        }
        const double w = dt / 6;
        s[0] += w * (kv[0][0] + 2 * kv[1][0] + 2 * kv[2][0] + kv[3][0]);
        s[1] += w * (kv[0][1] + 2 * kv[1][1] + 2 * kv[2][1] + kv[3][1]);
        s[2] += w * (thdd + 2 * thdd + 2 * thdd + thdd);
        s[3] += w * (aldd + 2 * aldd + 2 * aldd + aldd);
        refresh();
    }
};

template <> struct Env<PSH_ENV_QQ> : EnvQQ<false> {};
template <> struct Env<PSH_ENV_QQ_ST> : EnvQQ<true> {};

// ---------------------------------------------------------------------------------------------- Pendulum
template <> struct Env<PSH_ENV_PEND> {
    static constexpr int S = 2, O = 3, A = 1, P = 5, I = 2, X = 0;
    double s[2];
    double hmgl, d_pole, tau_max, inertia, r_inertia;  // 0.5 m g l, damping, torque limit, m l^2 / 3 and its reciprocal
    double sth, cth;

    PSH_DEV void setup(const DpCol& p) {
        const double g = p[0], m_pole = p[1], l_pole = p[2];
        d_pole = p[3]; tau_max = p[4];
        hmgl = 0.5 * m_pole * g * l_pole;
        inertia = m_pole * (l_pole * l_pole) / 3.;
        r_inertia = 1. / inertia;
    }
    PSH_DEV void state_bounds(double* lo, double* hi) const { lo[0] = lo[1] = -4 * PSH_PI; hi[0] = hi[1] = 4 * PSH_PI; }
    PSH_DEV void act_bounds(double* lo, double* hi) const { lo[0] = -tau_max; hi[0] = tau_max; }
    PSH_DEV void obs_bounds(double* lo, double* hi) const {
        const double inf = __builtin_inf();
        lo[0] = lo[1] = -1.; hi[0] = hi[1] = 1.; lo[2] = -inf; hi[2] = inf;
    }
    PSH_DEV void refresh() { psh_sincos(s[0], &sth, &cth); }
    PSH_DEV void reset(const double* init, int, bool) { s[0] = init[0]; s[1] = init[1]; refresh(); }
    PSH_DEV void observe(double* o) const { o[0] = sth; o[1] = cth; o[2] = s[1]; }
    PSH_DEV void dynamics(double* act, double dt, bool) {  // pendulum.py:49-61
        const double th_ddot = psh_quot_by(act[0] - hmgl * sth - d_pole * s[1], inertia, r_inertia);
        s[1] += th_ddot * dt;
        s[0] += s[1] * dt;
        refresh();
    }
};

// -------------------------------------------------------------------------------------------- QCartPole
template <int KIND> struct EnvQCP {
    static constexpr int S = 4, O = 5, A = 1, P = 14, I = 4, X = 0;
    double s[4];
    double g, eta_m, K_g_k_m_over_r, B_pole, B_eq, ml, l_rail, f_gain, a11, r_a11, a22, mlg;
    double sth, cth;

    PSH_DEV void setup(const DpCol& p) {
        g = p[0];
        const double m_cart = p[1]; l_rail = p[2]; eta_m = p[3];
        const double eta_g = p[4], K_g = p[5], J_m = p[6], r_mp = p[7], R_m = p[8], k_m = p[9];
        B_pole = p[10]; B_eq = p[11];
        const double m_pole = p[12], l_pole = p[13];
        const double J_pole = (l_pole * l_pole) * m_pole / 3.;
        const double J_eq = m_cart + (eta_g * (K_g * K_g) * J_m) / (r_mp * r_mp);
        f_gain = (eta_g * K_g * eta_m * k_m) / (R_m * r_mp);
        K_g_k_m_over_r = K_g * k_m / r_mp;
        ml = m_pole * l_pole;
        mlg = ml * g;
        a11 = m_pole + J_eq;
        r_a11 = 1. / a11;
        a22 = J_pole + m_pole * (l_pole * l_pole);
    }
    PSH_DEV void state_bounds(double* lo, double* hi) const {
        const double xb = 0.05, inf = __builtin_inf();
        if (KIND == PSH_ENV_QCP_SU) {
            hi[0] = l_rail / 2. - xb; lo[0] = -l_rail / 2. + xb;
            hi[1] = 4 * PSH_PI; lo[1] = -4 * PSH_PI;
            hi[2] = inf; lo[2] = -inf; hi[3] = inf; lo[3] = -inf;
        } else {
            const double th = 15 / 180. * PSH_PI;
            lo[0] = -l_rail / 2. + xb; hi[0] = l_rail / 2. - xb;
            lo[1] = PSH_PI - th; hi[1] = PSH_PI + th;
            lo[2] = -l_rail; hi[2] = l_rail;
            lo[3] = -2 * PSH_PI; hi[3] = 2 * PSH_PI;
        }
    }
    PSH_DEV void act_bounds(double* lo, double* hi) const { lo[0] = -8.; hi[0] = 8.; }
    PSH_DEV void obs_bounds(double* lo, double* hi) const {  // quanser_cartpole.py:45-50
        const double inf = __builtin_inf();
        hi[0] = l_rail / 2.; hi[1] = 1.; hi[2] = 1.; hi[3] = inf; hi[4] = inf;
        for (int i = 0; i < 5; ++i) lo[i] = -hi[i];
    }
    PSH_DEV void refresh() { psh_sincos(s[1], &sth, &cth); }
    PSH_DEV void reset(const double* init, int, bool) {
        for (int i = 0; i < 4; ++i) s[i] = init[i];
        refresh();
    }
    PSH_DEV void observe(double* o) const { o[0] = s[0]; o[1] = sth; o[2] = cth; o[3] = s[2]; o[4] = s[3]; }
    PSH_DEV void dynamics(double* act, double dt, bool) {
        const double x_dot = s[2], th_dot = s[3];
        const double force = f_gain * (eta_m * act[0] - K_g_k_m_over_r * x_dot);
        const double a12 = ml * cth;
        const double b1 = force - B_eq * x_dot - ml * sth * (th_dot * th_dot);
        const double b2 = -B_pole * th_dot - mlg * sth;
        // 2x2 solve.  The reference goes through LAPACK's pivoted LU; a11 = m_pole + J_eq always dominates
        // a21 = m_pole*l_pole*cos(th) for physical parameters, so the elimination order is the same.
        // (a11 is a constant of the environment: its quotients take the stored reciprocal and one correction step)
        const double l21 = psh_quot_by(a12, a11, r_a11);
        const double u22 = a22 - l21 * a12;
        const double x2 = psh_quot(b2 - l21 * b1, u22);
        const double x1 = psh_quot_by(b1 - x2 * a12, a11, r_a11);
        s[2] += x1 * dt; s[3] += x2 * dt;
        s[0] += s[2] * dt; s[1] += s[3] * dt;
        refresh();
    }
};
template <> struct Env<PSH_ENV_QCP_SU> : EnvQCP<PSH_ENV_QCP_SU> {};
template <> struct Env<PSH_ENV_QCP_ST> : EnvQCP<PSH_ENV_QCP_ST> {};

// ----------------------------------------------------------------------------------------- QBallBalancer
// QBallBalancerKin.__call__: 100 steps of SGD(lr 0.01, momentum 0.9) on the two-link constraint loss
PSH_DEV double qbb_ik(double th, double r, double l_plate) {
    const double l = l_plate / 2., d = 0.10;
    double t0 = r, t1 = l, b0 = 0., b1 = 0.;
    double sn, cs;
    sincos(th, &sn, &cs);
    const double cx = r * cs, cy = r * sn;
    for (int it = 0; it < 100; ++it) {
        const double dx1 = t0 - cx, dy1 = t1 - cy;
        const double rod = sqrt(dx1 * dx1 + dy1 * dy1);
        const double dx2 = t0 - r - l, dy2 = t1 - d;
        const double half = sqrt(dx2 * dx2 + dy2 * dy2);
        const double g_rod = 2 * (rod - d) / (2 * rod), g_half = 2 * (half - l) / (2 * half);
        const double g0 = g_rod * (2 * dx1) + g_half * (2 * dx2);
        const double g1 = g_rod * (2 * dy1) + g_half * (2 * dy2);
        if (it == 0) { b0 = g0; b1 = g1; } else { b0 = 0.9 * b0 + g0; b1 = 0.9 * b1 + g1; }
        t0 -= 0.01 * b0;
        t1 -= 0.01 * b1;
    }
    return PSH_PI / 2. - atan2(r + l - t0, t1 - d);
}

template <> struct Env<PSH_ENV_QBB> {
    static constexpr int S = 8, O = 8, A = 2, P = 20, I = 4, X = 2;
    double s[8];
    double plate[2];  // hidden plate angles (forward-Euler state, set by IK at reset)
    double g, m_ball, r_ball, l_plate, r_arm, c_frict, Vxp, Vxn, Vyp, Vyn, off_x, off_y;
    double J_ball, J_eq, r_J_eq, c_kin, A_m, B_eq_v, zeta, r_zeta;

    PSH_DEV void setup(const DpCol& p) {
        g = p[0]; m_ball = p[1]; r_ball = p[2]; l_plate = p[3]; r_arm = p[4];
        const double K_g = p[5], eta_g = p[6], J_l = p[7], J_m = p[8], k_m = p[9], R_m = p[10], eta_m = p[11], B_eq = p[12];
        c_frict = p[13]; Vxp = p[14]; Vxn = p[15]; Vyp = p[16]; Vyn = p[17]; off_x = p[18]; off_y = p[19];
        J_ball = 2. / 5 * m_ball * (r_ball * r_ball);
        J_eq = eta_g * (K_g * K_g) * J_m + J_l;
        c_kin = 2. * r_arm / l_plate;
        A_m = eta_g * K_g * eta_m * k_m / R_m;
        B_eq_v = eta_g * (K_g * K_g) * eta_m * (k_m * k_m) / R_m + B_eq;
        zeta = m_ball * (r_ball * r_ball) + J_ball;
        r_J_eq = 1. / J_eq; r_zeta = 1. / zeta;
    }
    PSH_DEV void state_bounds(double* lo, double* hi) const {
        const double m[8] = {PSH_PI / 4., PSH_PI / 4., l_plate / 2., l_plate / 2., 5 * PSH_PI, 5 * PSH_PI, 0.5, 0.5};
        for (int i = 0; i < 8; ++i) { lo[i] = -m[i]; hi[i] = m[i]; }
    }
    PSH_DEV void act_bounds(double* lo, double* hi) const { lo[0] = lo[1] = -3.0; hi[0] = hi[1] = 3.0; }
    PSH_DEV void obs_bounds(double* lo, double* hi) const { state_bounds(lo, hi); }  // quanser_ball_balancer.py:73
    PSH_DEV void refresh() {}
    PSH_DEV void reset(const double* init, int init_dim, bool simplified) {
        if (init_dim == 4) {
            for (int i = 0; i < 8; ++i) s[i] = 0.;
            s[2] = init[0]; s[3] = init[1]; s[6] = init[2]; s[7] = init[3];
        } else {
            for (int i = 0; i < 8; ++i) s[i] = init[i];
        }
        plate[0] = plate[1] = 0.;
        if (!simplified) {
            plate[0] = qbb_ik(s[0] + off_x, r_arm, l_plate);
            plate[1] = qbb_ik(s[1] + off_y, r_arm, l_plate);
        }
    }
    PSH_DEV void observe(double* o) const { for (int i = 0; i < 8; ++i) o[i] = s[i]; }
    PSH_DEV void dynamics(double* act, double dt, bool simplified) {
        if (!simplified) {  // voltage dead zone, written into the action (quanser_ball_balancer.py:212-215)
            if (Vxn <= act[0] && act[0] <= Vxp) act[0] = 0.;
            if (Vyn <= act[1] && act[1] <= Vyp) act[1] = 0.;
        }
        const double th_x = s[0] + off_x, th_y = s[1] + off_y, x = s[2], y = s[3];
        const double th_x_dot = s[4], th_y_dot = s[5], x_dot = s[6], y_dot = s[7];
        const double th_x_ddot = psh_quot_by(A_m * act[0] - B_eq_v * th_x_dot, J_eq, r_J_eq);
        const double th_y_ddot = psh_quot_by(A_m * act[1] - B_eq_v * th_y_dot, J_eq, r_J_eq);
        double sx, cx, sy, cy, sa, ca, sb, cb;
        psh_sincos(th_x, &sx, &cx); psh_sincos(th_y, &sy, &cy);
        psh_sincos(plate[0], &sa, &ca); psh_sincos(plate[1], &sb, &cb);
        const double rca = psh_rcp_refined(ca), rcb = psh_rcp_refined(cb);  // the plate stays far from vertical
        const double a_dot = psh_quot_by(c_kin * th_x_dot * cx, ca, rca);
        const double b_dot = psh_quot_by(c_kin * -th_y_dot * cy, cb, rcb);  // cos(-th_y) = cos(th_y)
        const double r2 = r_ball * r_ball;
        double x_ddot, y_ddot;
        if (simplified) {
            x_ddot = psh_quot_by(c_kin * m_ball * g * r2 * sx, zeta, r_zeta);
            y_ddot = psh_quot_by(c_kin * m_ball * g * r2 * sy, zeta, r_zeta);
        } else {
            const double a_ddot = rca * (c_kin * (th_x_ddot * cx - (th_x_dot * th_x_dot) * sx) + (a_dot * a_dot) * sa);
            // sin(-th_y) = -sin(th_y)
            const double b_ddot = rcb * (c_kin * (-th_y_ddot * cy - (th_y_dot * th_y_dot) * (-sy)) + (b_dot * b_dot) * sb);
            x_ddot = psh_quot_by(-c_frict * x_dot * r2 - J_ball * r_ball * a_ddot + m_ball * x * (a_dot * a_dot) * r2 +
                                c_kin * m_ball * g * r2 * sx, zeta, r_zeta);
            y_ddot = psh_quot_by(-c_frict * y_dot * r2 - J_ball * r_ball * b_ddot + m_ball * y * (b_dot * b_dot) * r2 +
                                c_kin * m_ball * g * r2 * sy, zeta, r_zeta);
        }
        s[4] += th_x_ddot * dt; s[5] += th_y_ddot * dt; s[6] += x_ddot * dt; s[7] += y_ddot * dt;
        for (int i = 0; i < 4; ++i) s[i] += s[4 + i] * dt;
        plate[0] += a_dot * dt; plate[1] += b_dot * dt;
    }
};

// -------------------------------------------------------------------------------------------- BallOnBeam
template <> struct Env<PSH_ENV_BOB> {
    static constexpr int S = 4, O = 4, A = 1, P = 8, I = 4, X = 0;
    double s[4];
    double g, m_ball, l_beam, c_frict, ang_offset, J_beam, zeta_ball, r_zeta_ball;

    PSH_DEV void setup(const DpCol& p) {
        g = p[0]; m_ball = p[1];
        const double r_ball = p[2], m_beam = p[3]; l_beam = p[4];
        const double d_beam = p[5]; c_frict = p[6]; ang_offset = p[7];
        const double J_ball = 2. / 5 * m_ball * (r_ball * r_ball);
        J_beam = 1. / 12 * m_beam * ((l_beam * l_beam) + (d_beam * d_beam));
        zeta_ball = m_ball + J_ball / (r_ball * r_ball);
        r_zeta_ball = 1. / zeta_ball;
    }
    PSH_DEV void state_bounds(double* lo, double* hi) const {
        const double m[4] = {l_beam / 2., PSH_PI / 4., 10., PSH_PI};
        for (int i = 0; i < 4; ++i) { lo[i] = -m[i]; hi[i] = m[i]; }
    }
    PSH_DEV void act_bounds(double* lo, double* hi) const {
        const double m = l_beam / 2. * g * 3.;
        lo[0] = -m; hi[0] = m;
    }
    PSH_DEV void obs_bounds(double* lo, double* hi) const { state_bounds(lo, hi); }  // ball_on_beam.py:32
    PSH_DEV void refresh() {}
    PSH_DEV void reset(const double* init, int, bool) { for (int i = 0; i < 4; ++i) s[i] = init[i]; }
    PSH_DEV void observe(double* o) const { for (int i = 0; i < 4; ++i) o[i] = s[i]; }
    PSH_DEV void dynamics(double* act, double dt, bool) {
        const double x = s[0], a = s[1] + ang_offset, x_dot = s[2], a_dot = s[3];
        double sa, ca;
        psh_sincos(a, &sa, &ca);
        const double zeta_beam = m_ball * (x * x) + J_beam;
        const double x_ddot = psh_quot_by(-c_frict * x_dot + m_ball * x * (a_dot * a_dot) - m_ball * g * sa, zeta_ball, r_zeta_ball);
        const double a_ddot = psh_quot(act[0] - 2. * m_ball * x * x_dot * a_dot - m_ball * g * ca * x, zeta_beam);
        s[2] += x_ddot * dt; s[3] += a_ddot * dt;
        s[0] += s[2] * dt; s[1] += s[3] * dt;
    }
};

// ------------------------------------------------------------------------------------ OneMassOscillator
template <> struct Env<PSH_ENV_OMO> {
    static constexpr int S = 2, O = 2, A = 1, P = 3, I = 2, X = 0;
    double s[2];
    double k, r_m, a0, a1;  // spring constant (the action bound), 1/m and the second row of the system matrix

    PSH_DEV void setup(const DpCol& p) {
        const double m = p[0], d = p[2];
        k = p[1];
        const double omega = sqrt(k / m), zeta = d / (2. * sqrt(m * k));  // one_mass_oscillator.py:55-57
        r_m = 1. / m;
        a0 = -(omega * omega);
        a1 = -2. * zeta * omega;
    }
    PSH_DEV void state_bounds(double* lo, double* hi) const { lo[0] = -1.; hi[0] = 1.; lo[1] = -10.; hi[1] = 10.; }
    PSH_DEV void act_bounds(double* lo, double* hi) const { lo[0] = -k; hi[0] = k; }
    PSH_DEV void obs_bounds(double* lo, double* hi) const { state_bounds(lo, hi); }  // one_mass_oscillator.py:39
    PSH_DEV void refresh() {}
    PSH_DEV void reset(const double* init, int, bool) { s[0] = init[0]; s[1] = init[1]; }
    PSH_DEV void observe(double* o) const { o[0] = s[0]; o[1] = s[1]; }
    PSH_DEV void dynamics(double* act, double dt, bool) {
        const double sd0 = s[1];
        const double sd1 = a0 * s[0] + a1 * s[1] + r_m * act[0];
        s[0] = s[0] + sd0 * dt;
        s[1] = s[1] + sd1 * dt;
    }
};

// ------------------------------------------------------------------------------- task: reward / done / final
template <int S, int A> PSH_DEV double quadr_cost(const psh_task& t, const double* es, const double* ea) {
    double cs = 0.;
#pragma unroll
    for (int i = 0; i < S; ++i) {
        double acc = 0.;
#pragma unroll
        for (int j = 0; j < S; ++j) acc += t.Q[i * S + j] * es[j];
        cs += es[i] * acc;
    }
    double ca = 0.;
#pragma unroll
    for (int i = 0; i < A; ++i) {
        double acc = 0.;
#pragma unroll
        for (int j = 0; j < A; ++j) acc += t.R[i * A + j] * ea[j];
        ca += ea[i] * acc;
    }
    return cs + ca;
}

// The shape of the task each simulation creates by default (_create_task of the classes cited at the top): which
// reward function, which error components are taken modulo, which final reward.  A kernel built for this shape (FIXT)
// has these decisions made at compile time and takes Q and R as diagonal; the host selects it when the caller's
// psh_task has exactly this shape (task_has_default_shape), the numbers (weights, desired state, factors) still come
// from the descriptor.
template <int KIND> struct TaskShape;
template <> struct TaskShape<PSH_ENV_QQ> { static constexpr int rew = PSH_REW_EXP_QUADR, radial_mask = 2, fin = 0; };
template <> struct TaskShape<PSH_ENV_QCP_SU> { static constexpr int rew = PSH_REW_SWINGUP, radial_mask = 2, fin = PSH_FIN_ENABLED | PSH_FIN_ALWAYS_NEGATIVE; };
template <> struct TaskShape<PSH_ENV_QCP_ST> { static constexpr int rew = PSH_REW_QUADR, radial_mask = 2, fin = PSH_FIN_ENABLED | PSH_FIN_STATE_DEPENDENT | PSH_FIN_TIME_DEPENDENT; };
template <> struct TaskShape<PSH_ENV_QBB> { static constexpr int rew = PSH_REW_SCALED_EXP_QUADR, radial_mask = 0, fin = 0; };
template <> struct TaskShape<PSH_ENV_BOB> { static constexpr int rew = PSH_REW_SCALED_EXP_QUADR, radial_mask = 0, fin = 0; };
template <> struct TaskShape<PSH_ENV_OMO> { static constexpr int rew = PSH_REW_QUADR, radial_mask = 0, fin = PSH_FIN_ENABLED | PSH_FIN_ALWAYS_NEGATIVE; };
template <> struct TaskShape<PSH_ENV_PEND> { static constexpr int rew = PSH_REW_EXP_QUADR, radial_mask = 2, fin = 0; };
template <> struct TaskShape<PSH_ENV_QQ_ST> { static constexpr int rew = PSH_REW_EXP_QUADR, radial_mask = 2, fin = 0; };

template <int KIND> static inline bool task_has_default_shape(const psh_task& t) {
    using Sh = TaskShape<KIND>;
    constexpr int S = Env<KIND>::S, A = Env<KIND>::A;
    if (t.rew_kind != Sh::rew || t.final_mode != Sh::fin) return false;
    if ((t.radial != 0) != (Sh::radial_mask != 0) || (t.radial && t.radial_mask != Sh::radial_mask)) return false;
    if (Sh::rew == PSH_REW_SWINGUP) return t.idx_x == 0 && t.idx_th == 1;
    for (int i = 0; i < S; ++i)
        for (int j = 0; j < S; ++j)
            if (i != j && t.Q[i * S + j] != 0.) return false;
    for (int i = 0; i < A; ++i)
        for (int j = 0; j < A; ++j)
            if (i != j && t.R[i * A + j] != 0.) return false;
    return true;
}

// e^T diag(Q) e + a^T diag(R) a, accumulated in the order of the dense form with its zero terms left out
template <int S, int A> PSH_DEV double quadr_cost_diag(const psh_task& t, const double* es, const double* ea) {
    double cs = 0., ca = 0.;
#pragma unroll
    for (int i = 0; i < S; ++i) cs += es[i] * (t.Q[i * S + i] * es[i]);
#pragma unroll
    for (int i = 0; i < A; ++i) ca += ea[i] * (t.R[i * A + i] * ea[i]);
    return cs + ca;
}

// ScaledExpQuadrErrRewFcn.reset: c_max from the CURRENT (randomized) bounds of this environment
template <int KIND, bool FIXT = false> PSH_DEV double task_c_max(const Env<KIND>& e, const psh_task& t) {
    constexpr int S = Env<KIND>::S, A = Env<KIND>::A;
    if ((FIXT ? TaskShape<KIND>::rew : t.rew_kind) != PSH_REW_SCALED_EXP_QUADR) return 0.;
    double lo[S], hi[S], alo[A], ahi[A], om[S], am[A];
    e.state_bounds(lo, hi);
    e.act_bounds(alo, ahi);
#pragma unroll
    for (int i = 0; i < S; ++i) om[i] = fmax(fabs(lo[i]), fabs(hi[i]));
#pragma unroll
    for (int i = 0; i < A; ++i) am[i] = fmax(fabs(alo[i]), fabs(ahi[i]));
    return -1.0 * psh_log(t.min_rew) / (FIXT ? quadr_cost_diag<S, A>(t, om, am) : quadr_cost<S, A>(t, om, am));
}

template <int KIND, bool FIXT = false>
PSH_DEV double step_rew(const psh_task& t, double c_max, const double* state, const double* act) {
    constexpr int S = Env<KIND>::S, A = Env<KIND>::A;
    using Sh = TaskShape<KIND>;
    double es[S], ea[A];
#pragma unroll
    for (int i = 0; i < S; ++i) es[i] = t.state_des[i] - state[i];
    if (FIXT ? Sh::radial_mask != 0 : t.radial != 0) {
#pragma unroll
        for (int i = 0; i < S; ++i)
            if ((FIXT ? Sh::radial_mask : t.radial_mask) & (1 << i)) es[i] = psh_fmod(es[i], t.mod[i]);
        // the +-pi fold touches EVERY error component (desired_state.py:125-126), velocities included
#pragma unroll
        for (int i = 0; i < S; ++i) {
            // v > pi ? 2 pi - v : v  and then  v < -pi ? -2 pi - v : v, each as one subtraction and a minimum / maximum:
            // for v <= pi the mirror image 2 pi - v is >= pi >= v, for v > pi it is < pi < v, so min(v, 2 pi - v) IS the
            // selected value, bit for bit (two instructions instead of a compare and two 32-bit selects); likewise the
            // maximum with -2 pi - v.  A NaN stays a NaN.
            double v = es[i];
            v = psh_min_raw(v, 2 * PSH_PI - v);
            v = psh_max_raw(v, -2 * PSH_PI - v);
            es[i] = v;
        }
    }
#pragma unroll
    for (int i = 0; i < A; ++i) ea[i] = -act[i];
    const int kind = FIXT ? Sh::rew : t.rew_kind;
    if (kind == PSH_REW_SWINGUP) {
        double eth = es[0], ex = es[0];
#pragma unroll
        for (int i = 0; i < S; ++i) {
            eth = (i == (FIXT ? 1 : t.idx_th)) ? es[i] : eth;
            ex = (i == (FIXT ? 0 : t.idx_x)) ? es[i] : ex;
        }
        const double cost_pole = t.swingup[3] * (eth * eth) + psh_log(eth * eth + t.swingup[4]);
        return -t.swingup[0] * cost_pole - t.swingup[1] * fabs(ex) - t.swingup[2] * (ea[0] * ea[0]) + 10;
    }
    // FIXT takes Q and R as diagonal.  The reference multiplies the full matrices (rewards.py:67-72: err.dot(Q).dot(err)), and
    // an infinite error component times one of their zeros is a NaN there, where the diagonal form says infinity: a cost
    // that is not finite is computed again the reference's way.  An infinite component needs a state that has left its
    // bounds (or an initial state that was never inside them): rare.  For the exponential rewards the test is the one exp
    // needs anyway; the plain quadratic one is corrected by env_step on the step that fails the bounds, which a step that
    // starts from an infinite state always does.
    constexpr bool DIAG = FIXT;
    double cost = DIAG ? quadr_cost_diag<S, A>(t, es, ea) : quadr_cost<S, A>(t, es, ea);
    if (kind == PSH_REW_QUADR) return -cost;
    double x = kind == PSH_REW_EXP_QUADR ? -cost : -c_max * cost;
    if (!(fabs(x) <= PSH_EXP_FAST)) {
        asm volatile("");
        if (DIAG) {
            cost = quadr_cost<S, A>(t, es, ea);
            x = kind == PSH_REW_EXP_QUADR ? -cost : -c_max * cost;
        }
        x = psh_exp_clamp(x);
    }
    return psh_exp_core(x);
}

// BoxSpace.contains on the state: inclusive bounds; a NaN makes the reference raise, here it sets `nan`
template <int KIND> PSH_DEV bool state_inside(const Env<KIND>& e, bool& nan) {
    constexpr int S = Env<KIND>::S;
    double lo[S], hi[S];
    e.state_bounds(lo, hi);
    bool ok = true;
#pragma unroll
    for (int i = 0; i < S; ++i) ok = ok && (e.s[i] >= lo[i]) && (e.s[i] <= hi[i]);
    if (!ok) {  // a NaN fails both comparisons, so only a state outside its bounds needs the test for one
#pragma unroll
        for (int i = 0; i < S; ++i) nan = nan || (e.s[i] != e.s[i]);
    }
    return ok;
}

// FinalRewTask.compute_final_rew for a FAILED state (success never happens in these tasks)
template <int KIND, bool FIXT = false> PSH_DEV double final_rew_failed(const Env<KIND>& e, const psh_task& t, double c_max, int remaining) {
    const int m = FIXT ? TaskShape<KIND>::fin : t.final_mode;
    const bool ap = m & PSH_FIN_ALWAYS_POSITIVE, an = m & PSH_FIN_ALWAYS_NEGATIVE;
    const bool sd = m & PSH_FIN_STATE_DEPENDENT, td = m & PSH_FIN_TIME_DEPENDENT;
    if (ap) return (an ? __builtin_nan("") : 0.);
    if (!sd && !td) return -1. * fabs(t.final_factor);
    if (!sd && td) return -1. * remaining;
    double zero[Env<KIND>::A];
#pragma unroll
    for (int i = 0; i < Env<KIND>::A; ++i) zero[i] = 0.;
    const double r = fabs(step_rew<KIND, FIXT>(t, c_max, e.s, zero));
    if (!td) return -1. * t.final_factor * r;
    return an ? __builtin_nan("") : -1. * remaining * r;
}

// SimPyEnv.step (base.py:177-201).  act_io: in = raw action; out = the action as the caller's array holds it
// after the call (QBB dead zone is written through when the action was inside its bounds, spaces/box.py:144-148).
template <int KIND, bool FIXT = false>
PSH_DEV double env_step(Env<KIND>& e, const psh_task& t, double c_max, int& curr_step, int& yielded, double* act_io,
                        double dt, int max_steps, bool inf_steps, bool simplified, bool& done, int& st) {
    constexpr int A = Env<KIND>::A;
    const int remaining = inf_steps ? 0 : max_steps - (curr_step + 1);
    double rew = step_rew<KIND, FIXT>(t, c_max, e.s, act_io);
    double alo[A], ahi[A], a[A];
    e.act_bounds(alo, ahi);
    // BoxSpace.project (spaces/box.py:144-148): an action inside its bounds is passed on as it is (the caller's array, which
    // only QBallBalancerSim writes to), any other one is clipped; clipping leaves an inside action unchanged
#pragma unroll
    for (int i = 0; i < A; ++i) a[i] = psh_clip_sym(act_io[i], ahi[i]);   // every simulation's action space is symmetric (act_bounds)
    if constexpr (KIND == PSH_ENV_QBB) {
        bool inside = true;
#pragma unroll
        for (int i = 0; i < A; ++i) inside = inside && (act_io[i] >= alo[i]) && (act_io[i] <= ahi[i]);
        e.dynamics(a, dt, simplified);
        if (inside) {
#pragma unroll
            for (int i = 0; i < A; ++i) act_io[i] = a[i];
        }
    } else {
        e.dynamics(a, dt, simplified);
    }
    curr_step += 1;
    bool nan = false;
    const bool failed = !state_inside<KIND>(e, nan);
    const bool timeout = !inf_steps && curr_step >= max_steps;
    const bool d = failed || timeout;
    if (d) {  // the status is only touched on the step that ends something: nothing of it on the common path
        // Two places where the reference multiplies by a zero that the kernels leave out, which matters for an infinite state
        // only (0 * inf is a NaN).  A step that starts from an infinite component ends with a non-finite state, which fails
        // the bounds, so both are dealt with here and cost the other steps nothing; they work from the new state, because
        // keeping the old one alive up to here would cost every step of the small simulations two register copies.
        if constexpr (KIND == PSH_ENV_OMO) {
            // one_mass_oscillator.py:80-85, A.dot(state) + B.dot(act) with A = [[0, 1], [-omega^2, -2 zeta omega]]: an infinite
            // position makes the new position a NaN (0 * inf in A's first row), where x + v dt keeps it infinite.  The new
            // velocity tells that case from an infinite old velocity: -omega^2 * (+-inf) is an infinity, while an infinite
            // velocity cancels against its own damping term into a NaN (then the status holds PSH_ST_NAN either way).
            const double inf = __builtin_inf();
            if (failed && fabs(e.s[0]) == inf && fabs(e.s[1]) == inf) { e.s[0] = __builtin_nan(""); nan = true; }
        }
        if constexpr (FIXT && TaskShape<KIND>::rew == PSH_REW_QUADR) {
            // step_rew, the plain quadratic reward from the diagonals: -inf where the reference's full matrices give a NaN for
            // an infinite error component.  Told apart from a cost that merely overflowed (an error beyond 1e154, or an
            // infinite action: -inf in the reference too) by the new state, which is non-finite after an infinite one.
            // (Not covered: a finite state beyond 1e154 that the step itself takes to infinity.)
            if (failed && rew == -__builtin_inf()) {
                bool finite = true;
#pragma unroll
                for (int i = 0; i < Env<KIND>::S; ++i) finite = finite && fabs(e.s[i]) <= 1.79e308;
                if (!finite) rew = __builtin_nan("");
            }
        }
        if (timeout) st |= PSH_ST_TIMEOUT;
        if (failed) st |= PSH_ST_FAILED;
        if (nan) st |= PSH_ST_NAN;
        if (((FIXT ? TaskShape<KIND>::fin : t.final_mode) & PSH_FIN_ENABLED) && !yielded) {
            yielded = 1;
            if (failed) rew += final_rew_failed<KIND, FIXT>(e, t, c_max, remaining);
        }
    }
    done = d;
    return rew;
}
