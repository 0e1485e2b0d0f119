"""
bootstrap_ci (Pyrado/pyrado/sampling/bootstrapping.py:7-140): nonparametric bootstrap confidence interval of a
statistic of a 1-dim sample, basic or studentized (percentile-t), one- or two-sided, as BayRn and SPOTA use it on the
returns of their evaluation rollouts (Pyrado/pyrado/algorithms/bayrn.py:300-304).

bootstrap_ci runs on the host and resamples with NumPy's global generator exactly like the reference, so a given seed
yields the reference's interval.  bootstrap_ci_device is the same interval for the statistic the callers use (the
mean) with the replications computed on the GPU from a return vector that is already there (psh_bootstrap_means,
counter-based draws: the distribution of the replications agrees with the reference's, the numbers of a seed do not).
"""
from warnings import warn

import numpy as np

from ..utils.exceptions import ValueErr


def bootstrap_ci(data: np.ndarray, stat_fcn, num_reps: int, alpha: float, ci_sides: int, bias_correction: bool = False,
                 studentized: bool = False, seed: int = None):
    assert isinstance(data, np.ndarray)
    assert isinstance(num_reps, int) and num_reps > 0
    assert ci_sides == 1 or ci_sides == 2
    if data.ndim != 1:
        raise NotImplementedError
    stat_emp = stat_fcn(data)
    n = data.shape[0]
    if seed is not None:
        np.random.seed(seed)
    # n draws with replacement per replication: the spread of the statistic depends on the sample size
    data_bs = np.random.choice(data, (n, num_reps), replace=True)
    stat_bs = np.apply_along_axis(stat_fcn, 0, data_bs)
    stat_ret = 2*stat_emp - np.mean(stat_bs) if bias_correction else stat_emp
    return _interval(stat_emp, stat_bs, np.std(data, ddof=0), n, alpha, ci_sides, bias_correction, studentized)


def _interval(stat_emp, stat_bs, std_emp, n, alpha, ci_sides, bias_correction, studentized):
    """ statistic to report and confidence interval from the bootstrap replications (:63-140 of the reference) """
    stat_ret = 2*stat_emp - np.mean(stat_bs) if bias_correction else stat_emp
    # the replications' deviations from the empirical statistic stand in for its deviation from the population value
    delta_bs = stat_bs - stat_emp
    if studentized:
        se_emp = std_emp/np.sqrt(n)
        if se_emp < 1e-9:
            warn('Standard deviation in the empirical data (se_emp) is below 1e-9.', UserWarning)
        se_bs = np.std(stat_bs, ddof=0)/np.sqrt(n)
        if se_bs < 1e-9:
            warn('Standard deviation in the bootstrapped data (se_bs) is below 1e-9. '
                 'Setting confidence interval bounds to infinity.', UserWarning)
            return stat_ret, [-np.inf, np.inf]
        dev, scale = delta_bs/se_bs, se_emp
    else:
        dev, scale = delta_bs, 1.
    if ci_sides == 2:
        lo, up = np.percentile(dev, [100*alpha/2., 100 - 100*alpha/2.])
        return stat_ret, [stat_emp - up*scale, stat_emp - lo*scale]
    if ci_sides == 1:
        lo = np.percentile(dev, 100*alpha)
        return stat_ret, [-np.inf, stat_emp - lo*scale]
    raise ValueErr(given=ci_sides, eq_constraint='1 or 2')


def bootstrap_ci_device(returns, num_reps: int, alpha: float, ci_sides: int, bias_correction: bool = False,
                        studentized: bool = False, seed: int = 0, engine=None):
    """ bootstrap_ci(returns, np.mean, ...) for a return vector on the GPU (e.g. RolloutBatch.device_results['ret']): the
    num_reps x n resampling and the means run on the device, only the num_reps replications come to the host. """
    assert isinstance(num_reps, int) and num_reps > 0
    assert ci_sides == 1 or ci_sides == 2
    if engine is None:
        from ..engine import get_engine
        engine = get_engine()
    returns = engine.f64(returns).reshape(-1)
    n = int(returns.numel())
    stat_bs = engine.bootstrap_means(returns, num_reps, 0 if seed is None else seed).cpu().numpy()
    stat_emp, std_emp = float(returns.mean()), float(returns.std(unbiased=False))
    return _interval(stat_emp, stat_bs, std_emp, n, alpha, ci_sides, bias_correction, studentized)
