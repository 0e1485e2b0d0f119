"""
Hand-computed one-step anchors for the simulations whose dynamics no test of the reference pins with numbers.
Each function evaluates the reference's `_step_dynamics` for ONE concrete (state, action, parameters, dt) with
plain Python floats and `math`, written from the formulas of the cited source lines and independent of both
restatements under oracle/.  The CPU tests hold the oracles to these numbers, the GPU tests the kernels.
"""
import math

import numpy as np


def _fold(e):
    """ RadiallySymmDesStateTask.step_rew, Pyrado/pyrado/tasks/desired_state.py:125-126, for one error component """
    if e > math.pi:
        e = 2*math.pi - e
    if e < -math.pi:
        e = -2*math.pi - e
    return e


def qcartpole_step():
    """ QCartPoleSim._step_dynamics, Pyrado/pyrado/environments/pysim/quanser_cartpole.py:83-124, short pole,
    nominal parameters (:58-81) except a heavier pole; the 2x2 system goes through numpy.linalg.solve as there. """
    g, m_cart, eta_m, eta_g, K_g, J_m = 9.81, 0.38, 0.9, 0.9, 3.71, 3.9e-7
    r_mp, R_m, k_m, B_pole, B_eq, m_pole, l_pole = 6.35e-3, 2.6, 7.67e-3, 0.0024, 5.4, 0.15, 0.3365/2
    dp = [g, m_cart, 0.814, eta_m, eta_g, K_g, J_m, r_mp, R_m, k_m, B_pole, B_eq, m_pole, l_pole]
    x, th, x_dot, th_dot = 0.05, 2.5, -0.3, 1.7
    u, dt = 3.0, 0.002
    J_pole = l_pole**2*m_pole/3.
    J_eq = m_cart + (eta_g*K_g**2*J_m)/r_mp**2
    force = (eta_g*K_g*eta_m*k_m)/(R_m*r_mp)*(eta_m*u - K_g*k_m*x_dot/r_mp)
    A = np.array([[m_pole + J_eq, m_pole*l_pole*math.cos(th)],
                  [m_pole*l_pole*math.cos(th), J_pole + m_pole*l_pole**2]])
    b = np.array([force - B_eq*x_dot - m_pole*l_pole*math.sin(th)*th_dot**2,
                  -B_pole*th_dot - m_pole*l_pole*g*math.sin(th)])
    x_ddot, th_ddot = np.linalg.solve(A, b)
    x_dot_n, th_dot_n = x_dot + x_ddot*dt, th_dot + th_ddot*dt
    nxt = [x + x_dot_n*dt, th + th_dot_n*dt, x_dot_n, th_dot_n]
    # QCartPoleSwingUpSim's task (:320-329): UnderActuatedSwingUpRewFcn (Pyrado/pyrado/tasks/reward_functions.py:299-303)
    # on the error to [0, pi, 0, 0], angle error modulo 2 pi, every component folded into [-pi, pi]
    e = [_fold(0. - x), _fold(math.fmod(math.pi - th, 2*math.pi)), _fold(0. - x_dot), _fold(0. - th_dot)]
    rew = -(1.0*(1.0*e[1]**2 + math.log(e[1]**2 + 0.1)) + 0.2*abs(e[0]) + 1e-3*(-u)**2) + 10
    return dict(dp=dp, state=[x, th, x_dot, th_dot], act=[u], dt=dt, next=nxt, rew=rew,
                obs=[nxt[0], math.sin(nxt[1]), math.cos(nxt[1]), nxt[2], nxt[3]])


def ball_on_beam_step():
    """ BallOnBeamSim._step_dynamics, Pyrado/pyrado/environments/pysim/ball_on_beam.py:58-98 """
    g, m_ball, r_ball, m_beam, l_beam, d_beam, c_frict, ang_offset = 9.81, 0.5, 0.1, 3., 2., 0.1, 0.05, 0.02
    x, al, x_dot, a_dot = 0.4, -0.1, 0.3, 0.25
    u, dt = 1.5, 0.01
    J_ball = 2./5*m_ball*r_ball**2
    J_beam = 1./12*m_beam*(l_beam**2 + d_beam**2)
    zeta_ball = m_ball + J_ball/r_ball**2
    a = al + ang_offset
    zeta_beam = m_ball*x**2 + J_beam
    x_ddot = (-c_frict*x_dot + m_ball*x*a_dot**2 - m_ball*g*math.sin(a))/zeta_ball
    a_ddot = (u - 2.*m_ball*x*x_dot*a_dot - m_ball*g*math.cos(a)*x)/zeta_beam
    x_dot_n, a_dot_n = x_dot + x_ddot*dt, a_dot + a_ddot*dt
    nxt = [x + x_dot_n*dt, al + a_dot_n*dt, x_dot_n, a_dot_n]
    # task (:69-77): ScaledExpQuadrErrRewFcn (Pyrado/pyrado/tasks/reward_functions.py:241-262), c_max from the bounds (:26-27)
    Q, s_max, a_max = [1e5, 1e3, 1e3, 1e2], [l_beam/2., math.pi/4., 10., math.pi], l_beam/2.*g*3.
    c_max = -1.0*math.log(1e-4)/(sum(q*m*m for q, m in zip(Q, s_max)) + a_max*1.*a_max)
    cost = sum(q*v*v for q, v in zip(Q, [x, al, x_dot, a_dot])) + u*1.*u
    return dict(dp=[g, m_ball, r_ball, m_beam, l_beam, d_beam, c_frict, ang_offset], state=[x, al, x_dot, a_dot],
                act=[u], dt=dt, next=nxt, obs=nxt, rew=math.exp(-c_max*cost))


def pendulum_step():
    """ PendulumSim._step_dynamics, Pyrado/pyrado/environments/pysim/pendulum.py:49-61 """
    g, m_pole, l_pole, d_pole, tau_max = 9.81, 1.2, 0.9, 0.05, 3.5
    th, th_dot = 0.7, -1.1
    u, dt = 2.0, 0.02
    th_ddot = (u - 0.5*m_pole*g*l_pole*math.sin(th) - d_pole*th_dot)/(m_pole*l_pole**2/3.)
    th_dot_n = th_dot + th_ddot*dt
    th_n = th + th_dot_n*dt
    # task (:30-36): ExpQuadrErrRewFcn(diag(1, 5e-3), 1e-3) on the error to [pi, 0]; idcs=[1] puts the modulo on the
    # VELOCITY error, and the fold touches both components
    e = [_fold(math.pi - th), _fold(math.fmod(0. - th_dot, 2*math.pi))]
    rew = math.exp(-(1.*e[0]**2 + 5e-3*e[1]**2 + 1e-3*u*u))
    return dict(dp=[g, m_pole, l_pole, d_pole, tau_max], state=[th, th_dot], act=[u], dt=dt, next=[th_n, th_dot_n],
                obs=[math.sin(th_n), math.cos(th_n), th_dot_n], rew=rew)


def qbb_step(simplified: bool):
    """ QBallBalancerSim._calc_constants / _step_dynamics,
    Pyrado/pyrado/environments/pysim/quanser_ball_balancer.py:154-173,197-271, nominal parameters (:131-152) with servo
    offsets; action 0 outside, action 1 inside its dead zone (:212-215); the hidden plate angles are given. """
    g, m_ball, r_ball, l_plate, r_arm, K_g, eta_g = 9.81, 0.003, 0.019625, 0.275, 0.0254, 70., 0.9
    J_l, J_m, k_m, R_m, eta_m, B_eq, c_frict = 5.2822e-5, 4.6063e-7, 0.0077, 2.6, 0.69, 0.015, 0.05
    Vxp, Vxn, Vyp, Vyn, off_x, off_y = 0.28, -0.10, 0.28, -0.074, 0.01, -0.02
    dp = [g, m_ball, r_ball, l_plate, r_arm, K_g, eta_g, J_l, J_m, k_m, R_m, eta_m, B_eq, c_frict, Vxp, Vxn, Vyp, Vyn,
          off_x, off_y]
    s = [0.12, -0.08, 0.03, -0.05, 0.9, -1.3, 0.02, 0.04]
    plate = [0.011, -0.007]
    act, dt = [1.2, 0.2], 0.01
    J_ball = 2./5*m_ball*r_ball**2
    J_eq = eta_g*K_g**2*J_m + J_l
    c_kin = 2.*r_arm/l_plate
    A_m = eta_g*K_g*eta_m*k_m/R_m
    B_eq_v = eta_g*K_g**2*eta_m*k_m**2/R_m + B_eq
    zeta = m_ball*r_ball**2 + J_ball
    u = list(act)
    if not simplified:
        if Vxn <= u[0] <= Vxp:
            u[0] = 0.
        if Vyn <= u[1] <= Vyp:
            u[1] = 0.
    th_x, th_y, x, y = s[0] + off_x, s[1] + off_y, s[2], s[3]
    th_x_dot, th_y_dot, x_dot, y_dot = s[4:]
    th_x_ddot = (A_m*u[0] - B_eq_v*th_x_dot)/J_eq
    th_y_ddot = (A_m*u[1] - B_eq_v*th_y_dot)/J_eq
    a, b = plate
    a_dot = c_kin*th_x_dot*math.cos(th_x)/math.cos(a)
    b_dot = c_kin*-th_y_dot*math.cos(-th_y)/math.cos(b)
    a_ddot = 1./math.cos(a)*(c_kin*(th_x_ddot*math.cos(th_x) - th_x_dot**2*math.sin(th_x)) + a_dot**2*math.sin(a))
    b_ddot = 1./math.cos(b)*(c_kin*(-th_y_ddot*math.cos(th_y) - (-th_y_dot)**2*math.sin(-th_y)) + b_dot**2*math.sin(b))
    if simplified:
        x_ddot = c_kin*m_ball*g*r_ball**2*math.sin(th_x)/zeta
        y_ddot = c_kin*m_ball*g*r_ball**2*math.sin(th_y)/zeta
    else:
        x_ddot = (-c_frict*x_dot*r_ball**2 - J_ball*r_ball*a_ddot + m_ball*x*a_dot**2*r_ball**2
                  + c_kin*m_ball*g*r_ball**2*math.sin(th_x))/zeta
        y_ddot = (-c_frict*y_dot*r_ball**2 - J_ball*r_ball*b_ddot + m_ball*y*(-b_dot)**2*r_ball**2
                  + c_kin*m_ball*g*r_ball**2*math.sin(th_y))/zeta
    vel = [s[4] + th_x_ddot*dt, s[5] + th_y_ddot*dt, s[6] + x_ddot*dt, s[7] + y_ddot*dt]
    pos = [s[i] + vel[i]*dt for i in range(4)]
    # task (:80-90): ScaledExpQuadrErrRewFcn on the pre-step state and the RAW action, bounds of :64-68
    Q, R = [1., 1., 5e3, 5e3, 1e-2, 1e-2, 0.5, 0.5], [1e-2, 1e-2]
    s_max = [math.pi/4., math.pi/4., l_plate/2., l_plate/2., 5*math.pi, 5*math.pi, 0.5, 0.5]
    c_max = -1.0*math.log(1e-4)/(sum(q*m*m for q, m in zip(Q, s_max)) + sum(r*3.*3. for r in R))
    cost = sum(q*v*v for q, v in zip(Q, s)) + sum(r*v*v for r, v in zip(R, act))
    return dict(dp=dp, state=s, plate=plate, act=act, dt=dt, next=pos + vel, rew=math.exp(-c_max*cost),
                plate_next=[a + a_dot*dt, b + b_dot*dt], act_after=u, obs=pos + vel)


def qbb_ik_tips(th: float, r: float, l_plate: float):
    """ The rod tip after one and after two optimizer steps of QBallBalancerKin.rod_tip
    (Pyrado/pyrado/environments/pysim/quanser_ball_balancer.py:438-471): SGD(lr 0.01, momentum 0.9) from
    tip = [r, l]; the first step moves by lr*grad, the second by lr*(0.9*grad_1 + grad_2).  The gradient of
    (rod - d)^2 + (half - l)^2 is written with the chain rule by hand. """
    l, d = l_plate/2., 0.10

    def grad(t0, t1):
        rod = math.sqrt((t0 - r*math.cos(th))**2 + (t1 - r*math.sin(th))**2)
        half = math.sqrt((t0 - r - l)**2 + (t1 - d)**2)
        return (2*(rod - d)*(t0 - r*math.cos(th))/rod + 2*(half - l)*(t0 - r - l)/half,
                2*(rod - d)*(t1 - r*math.sin(th))/rod + 2*(half - l)*(t1 - d)/half)
    g1 = grad(r, l)
    tip1 = (r - 0.01*g1[0], l - 0.01*g1[1])
    g2 = grad(*tip1)
    buf = (0.9*g1[0] + g2[0], 0.9*g1[1] + g2[1])
    tip2 = (tip1[0] - 0.01*buf[0], tip1[1] - 0.01*buf[1])
    return tip1, tip2


def qbb_ik_torch(th: float, r: float, l_plate: float, num_iter: int = 100) -> float:
    """ The inverse kinematics carried out by the libraries the reference hands it to: torch.optim.SGD(lr=0.01,
    momentum=0.9) on a 2-vector with the loss differentiated by autograd, then atan2
    (Pyrado/pyrado/environments/pysim/quanser_ball_balancer.py:413-480).  Independent of the restatements' analytic
    gradient and hand-written momentum update. """
    import torch
    l, d = l_plate/2., 0.10
    ang = torch.tensor(th, dtype=torch.float64)
    tip = torch.tensor([r, l], dtype=torch.float64, requires_grad=True)
    opt = torch.optim.SGD([tip], lr=0.01, momentum=0.9)
    for _ in range(num_iter):
        opt.zero_grad()
        rod = torch.sqrt((tip[0] - r*torch.cos(ang))**2 + (tip[1] - r*torch.sin(ang))**2)
        half = torch.sqrt((tip[0] - r - l)**2 + (tip[1] - d)**2)
        ((rod - d)**2 + (half - l)**2).backward()
        opt.step()
    return float(math.pi/2. - torch.atan2(r + l - tip[0], tip[1] - d).item())
