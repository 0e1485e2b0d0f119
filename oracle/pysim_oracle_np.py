"""
TEST INFRASTRUCTURE, NOT PRODUCT CODE: second, independent CPU restatement of the reference algorithm,
one environment at a time with NumPy, calling the same NumPy routines the reference calls
(np.sin/cos/fmod/exp/log/clip, np.linalg.solve, ndarray.dot).  It exists to cross-check the C
restatement (oracle/pysim_oracle.c) and to generate the golden vectors under tests/golden/.

PARITY STATUS (the reference could not be executed in the build environment): OneMassOscillatorSim is pinned by the
trajectory Pyrado/data/perma/time_series/omo_traj_gt.npy that the reference produced (tests/test_reference_golden.py:
1e-12 here -- this image's BLAS fuses a multiply-add inside ndarray.dot -- and bit for bit in the C restatement);
"parity unpinned" for the per-step dynamics of QQube, QCartPole, QBallBalancer and BallOnBeam (no numbers for them in
the reference's tests or data files); pinned by the reference's own known-answer tests for rewards, final rewards and
spaces (tests/test_oracle.py).

Paths in comments are relative to /root/reference/Pyrado/pyrado/.
"""
import numpy as np

PI = np.pi

# key order = row order of the domain-parameter tables
DP_KEYS = {
    'qq': ['g', 'Rm', 'km', 'Mr', 'Lr', 'Dr', 'Mp', 'Lp', 'Dp', 'max_volt'],
    'qcp': ['g', 'm_cart', 'l_rail', 'eta_m', 'eta_g', 'K_g', 'J_m', 'r_mp', 'R_m', 'k_m', 'B_pole', 'B_eq',
            'm_pole', 'l_pole'],
    'qbb': ['g', 'm_ball', 'r_ball', 'l_plate', 'r_arm', 'K_g', 'eta_g', 'J_l', 'J_m', 'k_m', 'R_m', 'eta_m',
            'B_eq', 'c_frict', 'V_thold_x_pos', 'V_thold_x_neg', 'V_thold_y_pos', 'V_thold_y_neg',
            'offset_th_x', 'offset_th_y'],
    'bob': ['g', 'm_ball', 'r_ball', 'm_beam', 'l_beam', 'd_beam', 'c_frict', 'ang_offset'],
    'omo': ['m', 'k', 'd'],
    'pend': ['g', 'm_pole', 'l_pole', 'd_pole', 'tau_max'],
}
DP_KEYS['qq-st'] = DP_KEYS['qq']
KIND_ID = {'qq': 0, 'qcp-su': 1, 'qcp-st': 2, 'qbb': 3, 'bob': 4, 'omo': 5, 'pend': 6, 'qq-st': 7}


def nominal(kind: str, long: bool = False) -> dict:
    """ get_nominal_domain_param() of each simulation """
    if kind in ('qq', 'qq-st'):  # pysim/quanser_qube.py:61-71
        v = [9.81, 8.4, 0.042, 0.095, 0.085, 5e-6, 0.024, 0.129, 1e-6, 4.]
        return dict(zip(DP_KEYS['qq'], v))
    if kind in ('qcp-su', 'qcp-st'):  # pysim/quanser_cartpole.py:57-81
        m_pole, l_pole = (0.23, 0.641/2) if long else (0.127, 0.3365/2)
        v = [9.81, 0.38, 0.814, 0.9, 0.9, 3.71, 3.9e-7, 6.35e-3, 2.6, 7.67e-3, 0.0024, 5.4, m_pole, l_pole]
        return dict(zip(DP_KEYS['qcp'], v))
    if kind == 'qbb':  # pysim/quanser_ball_balancer.py:95-152 (default thresholds :99-102)
        v = [9.81, 0.003, 0.019625, 0.275, 0.0254, 70., 0.9, 5.2822e-5, 4.6063e-7, 0.0077, 2.6, 0.69, 0.015, 0.05,
             0.28, -0.10, 0.28, -0.074, 0., 0.]
        return dict(zip(DP_KEYS['qbb'], v))
    if kind == 'bob':  # pysim/ball_on_beam.py:47-56
        return dict(zip(DP_KEYS['bob'], [9.81, 0.5, 0.1, 3., 2., 0.1, 0.05, 0.]))
    if kind == 'omo':  # pysim/one_mass_oscillator.py:53-57
        return dict(zip(DP_KEYS['omo'], [1., 30., 0.5]))
    if kind == 'pend':  # pysim/pendulum.py:41-47
        return dict(zip(DP_KEYS['pend'], [9.81, 1., 1., 0.05, 3.5]))
    raise KeyError(kind)


def dp_keys(kind: str):
    return DP_KEYS['qcp'] if kind.startswith('qcp') else DP_KEYS[kind]


# ---------------------------------------------------------------------------------------------- tasks
class Task:
    """ DesStateTask / RadiallySymmDesStateTask (+ FinalRewTask) folded into one object.
    tasks/desired_state.py:65-128, tasks/final_reward.py:66-187, tasks/reward_functions.py:167-303 """

    def __init__(self, state_des, rew_kind, Q=None, R=None, radial_idcs=None, mod=2*PI, final_mode=None,
                 final_factor=1e3, min_rew=1e-4, swingup=(1.0, 0.2, 1e-3, 1.0, 0.1), idx_x=0, idx_th=1):
        self.state_des = np.asarray(state_des, dtype=np.float64)
        self.rew_kind = rew_kind  # 'quadr' | 'exp' | 'scaledexp' | 'swingup'
        self.Q = None if Q is None else np.asarray(Q, dtype=np.float64)
        self.R = None if R is None else np.asarray(R, dtype=np.float64)
        self.radial_idcs = radial_idcs  # None -> DesStateTask
        if radial_idcs is not None:
            self.mod = mod*np.ones(len(radial_idcs))
        self.final_mode = final_mode  # None or dict of the FinalRewMode flags
        self.final_factor = final_factor
        self.min_rew = min_rew
        self.c_costs = np.array(swingup[:3])
        self.c_theta_sq, self.c_theta_log = swingup[3], swingup[4]
        self.idx_x, self.idx_th = idx_x, idx_th
        self.c_max = None
        self.yielded = False
        self.state_lo = self.state_hi = None

    def _cost(self, err_s, err_a):
        return err_s.dot(self.Q.dot(err_s)) + err_a.dot(self.R.dot(err_a))

    def reset(self, state_lo, state_hi, act_lo, act_hi):
        self.state_lo, self.state_hi = state_lo, state_hi
        self.yielded = False
        if self.rew_kind == 'scaledexp':
            obs_max = np.max(np.stack([np.abs(state_lo), np.abs(state_hi)], axis=0), axis=0)
            act_max = np.max(np.stack([np.abs(act_lo), np.abs(act_hi)], axis=0), axis=0)
            max_cost = obs_max.dot(self.Q.dot(obs_max)) + act_max.dot(self.R.dot(act_max))
            self.c_max = -1.0*np.log(self.min_rew)/max_cost

    def step_rew(self, state, act):
        err_state = self.state_des - state
        if self.radial_idcs is not None:
            err_state[self.radial_idcs] = np.fmod(err_state[self.radial_idcs], self.mod)
            err_state[err_state > PI] = 2*PI - err_state[err_state > PI]
            err_state[err_state < -PI] = -2*PI - err_state[err_state < -PI]
        err_a = -act
        if self.rew_kind == 'quadr':
            return -float(self._cost(err_state, err_a))
        if self.rew_kind == 'exp':
            return float(np.exp(-self._cost(err_state, err_a)))
        if self.rew_kind == 'scaledexp':
            return float(np.exp(-self.c_max*self._cost(err_state, err_a)))
        if self.rew_kind == 'swingup':
            cost_pole = self.c_theta_sq*err_state[self.idx_th]**2 + np.log(err_state[self.idx_th]**2 + self.c_theta_log)
            cost_cart = np.abs(err_state[self.idx_x])
            cost_act = float(err_a[0])**2
            return float(-self.c_costs @ np.asarray([cost_pole, cost_cart, cost_act]) + 10)
        raise KeyError(self.rew_kind)

    def has_failed(self, state):
        if np.isnan(state).any():
            raise ValueError('NaN state')
        return not bool(np.all(np.bitwise_and((state >= self.state_lo).astype(int), (state <= self.state_hi).astype(int))))

    def final_rew(self, state, remaining_steps, act_dim):
        m = self.final_mode
        if m is None or self.yielded:
            return 0.
        self.yielded = True
        if not self.has_failed(state):
            return 0.  # success never happens (tasks/utils.py:4-6)
        ap, an = m.get('always_positive', False), m.get('always_negative', False)
        sd, td = m.get('state_dependent', False), m.get('time_dependent', False)
        zero = np.zeros(act_dim)
        if not ap and not sd and not td:
            return -1.*np.abs(self.final_factor)  # default and always_negative
        if ap and not an:
            return 0.
        if sd and not td:
            return -1.*self.final_factor*abs(self.step_rew(state, zero))
        if sd and td and not an:
            return -1.*remaining_steps*abs(self.step_rew(state, zero))
        if td and not sd:
            return -1.*remaining_steps
        raise NotImplementedError


def default_task(kind: str) -> Task:
    if kind == 'qq':  # pysim/quanser_qube.py:38-44
        return Task([0., PI, 0., 0.], 'exp', np.diag([3e-1, 1., 2e-2, 5e-3]), np.diag([4e-3]), radial_idcs=[1])
    if kind == 'qcp-su':  # pysim/quanser_cartpole.py:320-329
        return Task([0., PI, 0., 0.], 'swingup', radial_idcs=[1], final_mode=dict(always_negative=True))
    if kind == 'qcp-st':  # pysim/quanser_cartpole.py:271-280
        return Task([0., PI, 0., 0.], 'quadr', np.diag([5e-0, 1e+1, 1e-2, 1e-2]), np.diag([1e-3]), radial_idcs=[1],
                    final_mode=dict(state_dependent=True, time_dependent=True))
    if kind == 'qbb':  # pysim/quanser_ball_balancer.py:80-90
        return Task(np.zeros(8), 'scaledexp', np.diag([1e0, 1e0, 5e3, 5e3, 1e-2, 1e-2, 5e-1, 5e-1]),
                    np.diag([1e-2, 1e-2]))
    if kind == 'bob':  # pysim/ball_on_beam.py:69-77
        return Task(np.zeros(4), 'scaledexp', np.diag([1e5, 1e3, 1e3, 1e2]), np.eye(1))
    if kind == 'omo':  # pysim/one_mass_oscillator.py:43-51
        return Task(np.zeros(2), 'quadr', np.diag([1e1, 1e-2]), np.diag([1e-6]),
                    final_mode=dict(always_negative=True), final_factor=1e3)
    if kind == 'pend':  # pysim/pendulum.py:30-36 (the radial index is the velocity, as the reference has it)
        return Task([PI, 0.], 'exp', np.diag([1e-0, 5e-3]), np.diag([1e-3]), radial_idcs=[1])
    if kind == 'qq-st':  # pysim/quanser_qube.py:264-271
        return Task([0., PI, 0., 0.], 'exp', np.diag([3., 4., 2., 2.]), np.diag([5e-2]), radial_idcs=[1])
    raise KeyError(kind)


# ---------------------------------------------------------------------------------------- environments
def qbb_ik(th: float, r: float, l_plate: float, num_iter: int = 100) -> float:
    """ QBallBalancerKin.__call__ pysim/quanser_ball_balancer.py:413-480: SGD(lr=0.01, momentum=0.9) on the
    two-constraint loss; the gradient is the chain torch.autograd builds for _loss_fcn. """
    l, d = float(l_plate/2.), 0.10
    tip = np.array([r, l], dtype=np.float64)
    buf = None
    c = np.array([r*np.cos(th), r*np.sin(th)])
    e = np.array([r + l, d])
    for _ in range(num_iter):
        v1 = tip - c
        rod_len = np.sqrt(v1[0]**2 + v1[1]**2)
        v2 = tip - e
        v2[0] = tip[0] - r - l
        half = np.sqrt(v2[0]**2 + v2[1]**2)
        g_rod = 2*(rod_len - d)/(2*rod_len)
        g_half = 2*(half - l)/(2*half)
        grad = g_rod*(2*v1) + g_half*(2*v2)
        buf = grad.copy() if buf is None else 0.9*buf + grad
        tip = tip - 0.01*buf
    return float(PI/2. - np.arctan2(r + l - tip[0], tip[1] - d))


class NpEnv:
    """ One SimPyEnv instance: reset/step as pysim/base.py:119-201 """

    def __init__(self, kind: str, dt: float, max_steps: int, task: Task = None, simplified_dyn: bool = False,
                 long: bool = False):
        self.kind, self.dt, self.max_steps = kind, dt, max_steps
        self.simplified = simplified_dyn
        self.dp = nominal(kind, long)
        self.task = default_task(kind) if task is None else task
        self.state = None
        self.plate_angs = np.zeros(2)
        self.curr_step = 0
        self._update()

    # _calc_constants + _create_spaces + task.reset (pysim/base.py:75-87)
    def _update(self):
        p = self.dp
        k = self.kind
        if k in ('qq', 'qq-st'):
            Jr = p['Mr']*p['Lr']**2/12
            Jp = p['Mp']*p['Lp']**2/12
            c = np.zeros(5)
            c[0] = Jr + p['Mp']*p['Lr']**2
            c[1] = 0.25*p['Mp']*p['Lp']**2
            c[2] = 0.5*p['Mp']*p['Lp']*p['Lr']
            c[3] = Jp + c[1]
            c[4] = 0.5*p['Mp']*p['Lp']*p['g']
            self.c = c
            ms = np.array([115./180*PI, 4*PI, 20*PI, 20*PI])
            self.s_lo, self.s_hi = -ms, ms
            if k == 'qq-st':  # pysim/quanser_qube.py:230-247
                th = 15/180.*PI
                self.s_lo = np.array([-120./180*PI, PI - th, -20*PI, -20*PI])
                self.s_hi = np.array([120./180*PI, PI + th, 20*PI, 20*PI])
            self.a_lo, self.a_hi = np.array([-p['max_volt']]), np.array([p['max_volt']])
        elif k in ('qcp-su', 'qcp-st'):
            self.J_pole = p['l_pole']**2*p['m_pole']/3.
            self.J_eq = p['m_cart'] + (p['eta_g']*p['K_g']**2*p['J_m'])/p['r_mp']**2
            l_rail, xb = p['l_rail'], 0.05
            if k == 'qcp-su':
                self.s_hi = np.array([+l_rail/2. - xb, +4*PI, np.inf, np.inf])
                self.s_lo = np.array([-l_rail/2. + xb, -4*PI, -np.inf, -np.inf])
            else:
                th = 15/180.*PI
                self.s_lo = np.array([-l_rail/2. + xb, PI - th, -l_rail, -2*PI])
                self.s_hi = np.array([+l_rail/2. - xb, PI + th, +l_rail, +2*PI])
            self.a_lo, self.a_hi = np.array([-8.]), np.array([8.])
        elif k == 'qbb':
            self.J_ball = 2./5*p['m_ball']*p['r_ball']**2
            self.J_eq = p['eta_g']*p['K_g']**2*p['J_m'] + p['J_l']
            self.c_kin = 2.*p['r_arm']/p['l_plate']
            self.A_m = p['eta_g']*p['K_g']*p['eta_m']*p['k_m']/p['R_m']
            self.B_eq_v = p['eta_g']*p['K_g']**2*p['eta_m']*p['k_m']**2/p['R_m'] + p['B_eq']
            self.zeta = p['m_ball']*p['r_ball']**2 + self.J_ball
            ms = np.array([PI/4., PI/4., p['l_plate']/2., p['l_plate']/2., 5*PI, 5*PI, 0.5, 0.5])
            self.s_lo, self.s_hi = -ms, ms
            self.a_lo, self.a_hi = -np.array([3.0, 3.0]), np.array([3.0, 3.0])
        elif k == 'bob':
            self.J_ball = 2./5*p['m_ball']*p['r_ball']**2
            self.J_beam = 1./12*p['m_beam']*(p['l_beam']**2 + p['d_beam']**2)
            self.zeta_ball = p['m_ball'] + self.J_ball/p['r_ball']**2
            ms = np.array([p['l_beam']/2., PI/4., 10., PI])
            self.s_lo, self.s_hi = -ms, ms
            ma = np.array([p['l_beam']/2.*p['g']*3.])
            self.a_lo, self.a_hi = -ma, ma
        elif k == 'pend':  # pysim/pendulum.py:17-28
            ms = np.array([4*PI, 4*PI])
            self.s_lo, self.s_hi = -ms, ms
            self.a_lo, self.a_hi = np.array([-p['tau_max']]), np.array([p['tau_max']])
        elif k == 'omo':
            self.omega = np.sqrt(p['k']/p['m'])
            self.zeta = p['d']/(2.*np.sqrt(p['m']*p['k']))
            ms = np.array([1., 10.])
            self.s_lo, self.s_hi = -ms, ms
            ma = np.array([ms[0]*p['k']])
            self.a_lo, self.a_hi = -ma, ma
        self.task.reset(self.s_lo, self.s_hi, self.a_lo, self.a_hi)

    def observe(self, s):
        if self.kind == 'qq':
            return np.array([np.sin(s[0]), np.cos(s[0]), np.sin(s[1]), np.cos(s[1]), s[2], s[3]])
        if self.kind in ('qcp-su', 'qcp-st'):
            return np.array([s[0], np.sin(s[1]), np.cos(s[1]), s[2], s[3]])
        if self.kind == 'pend':  # pysim/pendulum.py:38-39
            return np.array([np.sin(s[0]), np.cos(s[0]), s[1]])
        if self.kind == 'qq-st':  # pysim/quanser_qube.py:257-262
            return (np.array([0., PI, 0., 0.]) - s).copy()
        return s.copy()

    def reset(self, init_state, domain_param=None):
        self.curr_step = 0
        if domain_param is not None:
            self.dp.update(domain_param)
            self._update()
        init_state = np.array(init_state, dtype=np.float64)
        if self.kind == 'qbb' and init_state.shape == (4,):
            s = np.zeros(8)
            s[2:4] = init_state[:2]
            s[6:8] = init_state[2:]
            self.state = s
        else:
            self.state = init_state.copy()
        self.task.reset(self.s_lo, self.s_hi, self.a_lo, self.a_hi)
        if self.kind == 'qbb':
            if self.simplified:
                self.plate_angs = np.zeros(2)
            else:
                self.plate_angs[0] = qbb_ik(self.state[0] + self.dp['offset_th_x'], float(self.dp['r_arm']), self.dp['l_plate'])
                self.plate_angs[1] = qbb_ik(self.state[1] + self.dp['offset_th_y'], float(self.dp['r_arm']), self.dp['l_plate'])
        return self.observe(self.state)

    def _dynamics(self, act):
        p, dt, k = self.dp, self.dt, self.kind
        if k == 'pend':  # pysim/pendulum.py:49-61
            th, th_dot = self.state
            th_ddot = (act - 0.5*p['m_pole']*p['g']*p['l_pole']*np.sin(th) - p['d_pole']*th_dot)/(p['m_pole']*p['l_pole']**2/3.)
            self.state[1] += float(th_ddot[0])*dt
            self.state[0] += self.state[1]*dt
        elif k in ('qq', 'qq-st'):  # pysim/quanser_qube.py:92-142
            def dyn(x, u):
                th, al, thd, ald = x
                a = self.c[0] + self.c[1]*np.sin(al)**2
                b = self.c[2]*np.cos(al)
                c = self.c[3]
                det = a*c - b*b
                trq = p['km']*(u - p['km']*thd)/p['Rm']
                c0 = self.c[1]*np.sin(2*al)*thd*ald - self.c[2]*np.sin(al)*ald*ald
                c1 = -0.5*self.c[1]*np.sin(2*al)*thd*thd + self.c[4]*np.sin(al)
                x_ = trq - p['Dr']*thd - c0
                y_ = -p['Dp']*ald - c1
                return thd, ald, float((c*x_ - b*y_)/det), float((a*y_ - b*x_)/det)
            u = float(act[0])
            thd, ald, thdd, aldd = dyn(self.state, u)
            kk = np.zeros((4, 4))
            kk[0, :] = np.array([thd, ald, thdd, aldd])
            for j in range(1, 4):
                s = self.state + (dt/2. if j <= 2 else dt)*kk[j - 1, :]
                thd, ald, thdd, aldd = dyn(self.state, u)  # un-advanced state, as the reference does
                kk[j, :] = np.array([s[2], s[3], thdd, aldd])
            self.state += dt/6*(kk[0] + 2*kk[1] + 2*kk[2] + kk[3])
        elif k in ('qcp-su', 'qcp-st'):  # pysim/quanser_cartpole.py:95-124
            x, th, x_dot, th_dot = self.state
            force = (p['eta_g']*p['K_g']*p['eta_m']*p['k_m'])/(p['R_m']*p['r_mp'])*(
                    p['eta_m']*float(act[0]) - p['K_g']*p['k_m']*x_dot/p['r_mp'])
            A = np.array([[p['m_pole'] + self.J_eq, p['m_pole']*p['l_pole']*np.cos(th)],
                          [p['m_pole']*p['l_pole']*np.cos(th), self.J_pole + p['m_pole']*p['l_pole']**2]])
            b = np.array([force - p['B_eq']*x_dot - p['m_pole']*p['l_pole']*np.sin(th)*th_dot**2,
                          - p['B_pole']*th_dot - p['m_pole']*p['l_pole']*p['g']*np.sin(th)])
            x_ddot, theta_ddot = np.linalg.solve(A, b)
            self.state[2:] += np.array([x_ddot, theta_ddot])*dt
            self.state[:2] += self.state[2:]*dt
        elif k == 'qbb':  # pysim/quanser_ball_balancer.py:197-271
            if not self.simplified:
                if p['V_thold_x_neg'] <= act[0] <= p['V_thold_x_pos']:
                    act[0] = 0
                if p['V_thold_y_neg'] <= act[1] <= p['V_thold_y_pos']:
                    act[1] = 0
            th_x = self.state[0] + p['offset_th_x']
            th_y = self.state[1] + p['offset_th_y']
            x, y = self.state[2], self.state[3]
            th_x_dot, th_y_dot, x_dot, y_dot = self.state[4], self.state[5], self.state[6], self.state[7]
            th_x_ddot = (self.A_m*act[0] - self.B_eq_v*th_x_dot)/self.J_eq
            th_y_ddot = (self.A_m*act[1] - self.B_eq_v*th_y_dot)/self.J_eq
            a, b = self.plate_angs[0], self.plate_angs[1]
            a_dot = self.c_kin*th_x_dot*np.cos(th_x)/np.cos(a)
            b_dot = self.c_kin*-th_y_dot*np.cos(-th_y)/np.cos(b)
            a_ddot = 1./np.cos(a)*(self.c_kin*(th_x_ddot*np.cos(th_x) - th_x_dot**2*np.sin(th_x)) + a_dot**2*np.sin(a))
            b_ddot = 1./np.cos(b)*(self.c_kin*(-th_y_ddot*np.cos(th_y) - (-th_y_dot)**2*np.sin(-th_y)) + b_dot**2*np.sin(b))
            m_ball, r_ball, g, c_frict = p['m_ball'], p['r_ball'], p['g'], p['c_frict']
            if self.simplified:
                x_ddot = self.c_kin*m_ball*g*r_ball**2*np.sin(th_x)/self.zeta
                y_ddot = self.c_kin*m_ball*g*r_ball**2*np.sin(th_y)/self.zeta
            else:
                x_ddot = (- c_frict*x_dot*r_ball**2 - self.J_ball*r_ball*a_ddot + m_ball*x*a_dot**2*r_ball**2
                          + self.c_kin*m_ball*g*r_ball**2*np.sin(th_x))/self.zeta
                y_ddot = (- c_frict*y_dot*r_ball**2 - self.J_ball*r_ball*b_ddot + m_ball*y*(-b_dot)**2*r_ball**2
                          + self.c_kin*m_ball*g*r_ball**2*np.sin(th_y))/self.zeta
            self.state[4:] += np.array([th_x_ddot, th_y_ddot, x_ddot, y_ddot])*dt
            self.state[:4] += self.state[4:]*dt
            self.plate_angs += np.array([a_dot, b_dot])*dt
        elif k == 'bob':  # pysim/ball_on_beam.py:79-98
            x = self.state[0]
            a = self.state[1] + p['ang_offset']
            x_dot, a_dot = self.state[2], self.state[3]
            zeta_beam = p['m_ball']*x**2 + self.J_beam
            x_ddot = (- p['c_frict']*x_dot + p['m_ball']*x*a_dot**2 - p['m_ball']*p['g']*np.sin(a))/self.zeta_ball
            a_ddot = (float(act[0]) - 2.*p['m_ball']*x*x_dot*a_dot - p['m_ball']*p['g']*np.cos(a)*x)/zeta_beam
            self.state[2:] += np.array([x_ddot, a_ddot])*dt
            self.state[:2] += self.state[2:]*dt
        elif k == 'omo':  # pysim/one_mass_oscillator.py:75-84
            A = np.array([[0, 1], [-self.omega**2, -2.*self.zeta*self.omega]])
            B = np.array([[0], [1./p['m']]])
            state_dot = A.dot(self.state) + B.dot(act).reshape(2, )
            self.state = self.state + state_dot*dt

    def step(self, act, max_steps_inf=False):
        """ pysim/base.py:177-201.  `act` may be modified in place (QBB dead zone), as in the reference. """
        remaining = 0 if max_steps_inf else self.max_steps - (self.curr_step + 1)
        rew = self.task.step_rew(self.state, act)
        inside = bool(np.all((act >= self.a_lo) & (act <= self.a_hi)))
        act_lim = act if inside else np.clip(act, self.a_lo, self.a_hi)
        self._dynamics(act_lim)
        t = self.curr_step*self.dt
        self.curr_step += 1
        done = self.task.has_failed(self.state)
        if not max_steps_inf and self.curr_step >= self.max_steps:
            done = True
        if done:
            rew += self.task.final_rew(self.state, remaining, act.shape[0])
        return self.observe(self.state), rew, done, dict(t=t)


# ------------------------------------------------------------------------------------------- policies
def feat_eval(name, x):  # policies/features.py:90-140
    return {
        'identity': lambda v: v.copy(), 'sign': np.sign, 'abs': np.abs, 'squared': lambda v: np.power(v, 2),
        'cubed': lambda v: np.power(v, 3), 'sig': lambda v: 1./(1. + np.exp(-v)), 'bell': lambda v: np.exp(-np.power(v, 2)),
        'sin': np.sin, 'cos': np.cos, 'sinsin': lambda v: np.sin(v)*np.sin(v), 'sincos': lambda v: np.sin(v)*np.cos(v),
    }[name](x)


class MultFeatNp:
    """ policies/features.py:143-167 """

    def __init__(self, idcs):
        self.idcs = list(idcs)

    def __call__(self, x):
        out = x[self.idcs[0]]
        for i in self.idcs[1:]:
            out = out*x[i]
        return np.array([out])


class RandFourierFeatNp:
    """ policies/features.py:170-219 with the sampled frequencies and shifts given """

    def __init__(self, scale, freq, shift):
        self.scale, self.freq, self.shift = float(scale), np.asarray(freq, float), np.asarray(shift, float)

    def __call__(self, x):
        return self.scale*np.cos(x @ self.freq.T + self.shift)


class RBFFeatNp:
    """ policies/features.py:222-303; normalize() of utils/normalizing.py:87-105 """

    def __init__(self, centers, scale, state_wise=True):
        self.centers, self.scale, self.state_wise = np.asarray(centers, float), scale, state_wise

    @staticmethod
    def _normalize(x, axis):
        norm = np.atleast_1d(np.linalg.norm(x, ord=1, axis=axis))
        norm = np.where(norm > 1e-8, norm, np.ones_like(norm))
        return x/np.expand_dims(norm, axis)

    def __call__(self, x):
        sample = np.exp(-self.scale*np.power(np.tile(x, (self.centers.shape[0], 1)) - self.centers, 2))   # [K, dim]
        if self.state_wise:
            return self._normalize(sample, 0).T.reshape(-1)
        return self._normalize(sample.T.reshape(-1), -1).reshape(-1)


class LinearPolicyNp:
    """ policies/linear.py:65-80; `feats` holds names of the element-wise features or feature objects """

    def __init__(self, feats, W):
        self.feats, self.W = feats, np.asarray(W, dtype=np.float64)

    def __call__(self, obs):
        parts = [np.ones(1) if f == 'const' else (feat_eval(f, obs) if isinstance(f, str) else f(obs)) for f in self.feats]
        return self.W.dot(np.concatenate(parts))


class FNNPolicyNp:
    """ policies/fnn.py:129-151; params in torch parameters() order """

    def __init__(self, in_dim, hidden, out_dim, params, hidden_nonlin='tanh', out_nonlin=None):
        self.layers = []
        params = np.asarray(params, dtype=np.float64)
        last, off = in_dim, 0
        for h in list(hidden) + [out_dim]:
            W = params[off:off + h*last].reshape(h, last)
            off += h*last
            b = params[off:off + h]
            off += h
            self.layers.append((W, b))
            last = h
        assert off == params.size
        self.hn, self.on = hidden_nonlin, out_nonlin

    @staticmethod
    def _nl(name, x):
        if name is None or name == 'none':
            return x
        return {'tanh': np.tanh, 'relu': lambda v: np.maximum(v, 0.), 'sigmoid': lambda v: 1./(1. + np.exp(-v))}[name](x)

    def __call__(self, obs):
        x = obs
        for l, (W, b) in enumerate(self.layers[:-1]):
            x = self._nl(self.hn if isinstance(self.hn, (str, type(None))) else self.hn[l], W.dot(x) + b)
        W, b = self.layers[-1]
        return self._nl(self.on, W.dot(x) + b)


# -------------------------------------------------------------------------------------------- rollout
# ---- environment wrappers, environment_wrappers/*.py, as nested objects like the reference's ------------------
M32 = 0xFFFFFFFF


def philox4x32_10(ctr, key):
    """ Philox4x32-10 (the product's counter-based generator; the reference draws from NumPy's global one) """
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0, p1 = 0xD2511F53*c0, 0xCD9E8D57*c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
        k0, k1 = (k0 + 0x9E3779B9) & M32, (k1 + 0xBB67AE85) & M32
    return c0, c1, c2, c3


def step_normals(seed, env, t, stream, n):
    """ n standard normals for (seed, env, step, stream): Box-Muller pairs, pair p from counter word 3 = stream + 16 p """
    out = np.zeros(n)
    for p in range((n + 1)//2):
        o = philox4x32_10((env & M32, (env >> 32) & M32, t, stream + 16*p), (seed & M32, (seed >> 32) & M32))
        ua = ((o[0] >> 5)*67108864.0 + (o[1] >> 6))*(1.0/9007199254740992.0)
        ub = ((o[2] >> 5)*67108864.0 + (o[3] >> 6))*(1.0/9007199254740992.0)
        r = np.sqrt(-2.0*np.log(1.0 - ua))
        out[2*p] = r*np.cos(2.0*PI*ub)
        if 2*p + 1 < n:
            out[2*p + 1] = r*np.sin(2.0*PI*ub)
    return out


def step_uniforms(seed, env, t, stream):
    o = philox4x32_10((env & M32, (env >> 32) & M32, t, stream), (seed & M32, (seed >> 32) & M32))
    return np.array([((o[0] >> 5)*67108864.0 + (o[1] >> 6))*(1.0/9007199254740992.0),
                     ((o[2] >> 5)*67108864.0 + (o[3] >> 6))*(1.0/9007199254740992.0)])


class ExplPolicyNp:
    """ StochasticActionExplStrat.forward (exploration/stochastic_action.py:42-68) around a policy: Normal
    (normal_noise.py:100-108) or Uniform (uniform_noise.py) action noise; one draw per call, indexed by the call count """

    def __init__(self, policy, kind, mean, scale, seed, env_id):
        self.policy, self.kind, self.seed, self.env_id, self.k = policy, kind, seed, env_id, 0
        self.mean, self.scale = np.asarray(mean, float), np.asarray(scale, float)

    def __call__(self, obs):
        act = np.asarray(self.policy(obs), dtype=np.float64).reshape(-1)
        if self.kind == 'normal':
            act = (act + self.mean) + step_normals(self.seed, self.env_id, self.k, 3, act.shape[0])*self.scale
        else:
            m = act + self.mean
            lo, hi = m - self.scale, m + self.scale
            act = lo + step_uniforms(self.seed, self.env_id, self.k, 3)[:act.shape[0]]*(hi - lo)
        self.k += 1
        return act


class NpWrapper:
    """ EnvWrapper (environment_wrappers/base.py:29-141): forwards everything to the wrapped env """

    def __init__(self, wrapped):
        self.wrapped = wrapped

    def __getattr__(self, name):  # curr_step, max_steps, a_lo, a_hi, state, ...
        return getattr(self.__dict__['wrapped'], name)

    @property
    def sim(self):
        return self.wrapped.sim if isinstance(self.wrapped, NpWrapper) else self.wrapped

    def obs_bounds(self):
        return self.wrapped.obs_bounds()

    def reset(self, init_state, domain_param=None):
        return self.wrapped.reset(init_state, domain_param)

    def step(self, act):
        return self.wrapped.step(act)


def _sim_obs_bounds(env: NpEnv):
    """ observation spaces: pysim/quanser_qube.py:25-33, quanser_cartpole.py:45-50, others = state space """
    if env.kind == 'qq':
        m = np.array([1., 1., 1., 1., np.inf, np.inf])
        return -m, m
    if env.kind in ('qcp-su', 'qcp-st'):
        m = np.array([env.dp['l_rail']/2., 1., 1., np.inf, np.inf])
        return -m, m
    if env.kind == 'pend':
        m = np.array([1., 1., np.inf])
        return -m, m
    return env.s_lo.copy(), env.s_hi.copy()


NpEnv.obs_bounds = _sim_obs_bounds


class NpActNorm(NpWrapper):  # action_normalization.py:10-22
    def step(self, act):
        lb, ub = self.wrapped.a_lo, self.wrapped.a_hi
        return self.wrapped.step(lb + (act + 1)*(ub - lb)/2)


class NpActDelay(NpWrapper):  # action_delay.py:12-81
    def __init__(self, wrapped, delay=0):
        super().__init__(wrapped)
        self.delay, self.queue = delay, []

    def reset(self, init_state, domain_param=None):
        if domain_param is not None:
            self.delay = int(domain_param.get('act_delay', self.delay))
            domain_param = {k: v for k, v in domain_param.items() if k != 'act_delay'}
        self.queue = [np.zeros(self.sim.a_lo.shape)]*self.delay
        return self.wrapped.reset(init_state, domain_param)

    def step(self, act):
        if self.delay != 0:
            self.queue.append(act)
            act = self.queue.pop(0)
        return self.wrapped.step(act)


class NpNoise(NpWrapper):
    def __init__(self, wrapped, mean, std, seed, env_id):
        super().__init__(wrapped)
        self.mean, self.std, self.seed, self.env_id = np.asarray(mean, float), np.asarray(std, float), seed, env_id


class NpActNoise(NpNoise):  # action_noise.py:42-47; step counter = the simulation's curr_step before the step
    def step(self, act):
        noise = step_normals(self.seed, self.env_id, self.sim.curr_step, 2, act.shape[0])*self.std + self.mean
        return self.wrapped.step(act + noise)


class NpObsNoise(NpNoise):  # observation_noise.py:34-39; one draw per observation, indexed by its time step
    def _process(self, obs):
        return obs + (step_normals(self.seed, self.env_id, self.sim.curr_step, 1, obs.shape[0])*self.std + self.mean)

    def reset(self, init_state, domain_param=None):
        return self._process(self.wrapped.reset(init_state, domain_param))

    def step(self, act):
        obs, rew, done, info = self.wrapped.step(act)
        return self._process(obs), rew, done, info


class NpObsNorm(NpWrapper):  # observation_normalization.py:73-86; lo/hi: explicit bounds, NaN = the env's own
    def __init__(self, wrapped, lo=None, hi=None):
        super().__init__(wrapped)
        self.lo, self.hi = lo, hi

    def _process(self, obs):
        lb, ub = self.wrapped.obs_bounds()
        if self.lo is not None:
            lb = np.where(np.isnan(self.lo), lb, self.lo)
        if self.hi is not None:
            ub = np.where(np.isnan(self.hi), ub, self.hi)
        return (obs - lb)/(ub - lb)*2 - 1

    def reset(self, init_state, domain_param=None):
        return self._process(self.wrapped.reset(init_state, domain_param))

    def step(self, act):
        obs, rew, done, info = self.wrapped.step(act)
        return self._process(obs), rew, done, info


class NpObsPartial(NpWrapper):  # observation_partial.py:43-46
    def __init__(self, wrapped, keep):
        super().__init__(wrapped)
        self.keep = np.asarray(keep, dtype=bool)

    def obs_bounds(self):
        lo, hi = self.wrapped.obs_bounds()
        return lo[self.keep], hi[self.keep]

    def reset(self, init_state, domain_param=None):
        return self.wrapped.reset(init_state, domain_param)[self.keep]

    def step(self, act):
        obs, rew, done, info = self.wrapped.step(act)
        return obs[self.keep], rew, done, info


def rollout_np(env: NpEnv, policy, init_state, domain_param=None, stop_on_done=True, act_delay=0, bernoulli_reset=None,
               seed=0, env_id=0):
    """ sampling/rollout.py:88-275 (+ ActDelayWrapper environment_wrappers/action_delay.py:55-81) """
    obs = env.reset(init_state, domain_param)
    queue = [np.zeros(env.a_lo.shape)]*act_delay  # legacy shortcut for a single outermost ActDelayWrapper
    obs_hist, act_hist, rew_hist, t_hist = [], [], [], []
    done = False
    while not (done and stop_on_done) and env.curr_step < env.max_steps:
        if np.isnan(obs).any():
            raise ValueError('NaN observation')
        act = np.asarray(policy(obs), dtype=np.float64).reshape(-1)
        if np.isnan(act).any():
            raise ValueError('NaN action')
        inner = act
        if act_delay != 0:
            queue.append(act)
            inner = queue.pop(0)
        obs_next, rew, done, info = env.step(inner)
        obs_hist.append(obs)
        act_hist.append(act)
        rew_hist.append(rew)
        t_hist.append(info['t'])
        obs = obs_next
        # stochastic reset, sampling/rollout.py:248-254 (one uniform per step from the product's Philox stream 4)
        if bernoulli_reset is not None and step_uniforms(seed, env_id, len(rew_hist) - 1, 4)[0] < bernoulli_reset:
            break
    obs_hist.append(obs)
    return dict(observations=np.stack(obs_hist), actions=np.stack(act_hist), rewards=np.array(rew_hist),
                t=np.array(t_hist), length=len(rew_hist), ret=float(np.sum(rew_hist)), final_state=np.array(env.state).copy())
